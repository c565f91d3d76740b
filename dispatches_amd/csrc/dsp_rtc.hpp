// dsp_rtc.hpp — run-time specialisation of the register-resident PDLP kernel (host code, included by dsp_capi.hip).
//
// The register-resident kernel is a template over the LP's shape: owned columns / rows per lane and the per-slot ELL widths
// of A^T and A (pdlp_solve_kernel<CPL, RPL, LONG, WC, WR, QP>).  The library ships ahead-of-time instantiations for the
// shapes of the reference's flowsheets (DSP_MATREG_SHAPES: no compile latency for them); every OTHER LP that fits the fused
// kernels gets its tight instantiation compiled HERE, at dsp_create, with hiprtc (2-3 s once, then a disk cache), instead of
// dropping to a padded shape or to the LDS-matrix kernel (3-4x slower).  Needs, at run time: libhiprtc.so (dlopen'ed - the
// library loads without it) and the kernel sources (the csrc/ directory next to libdsp_hip.so, or $DSP_KERNEL_SRC).
// Anything missing or failing -> the caller falls back to the padded / generic ahead-of-time kernels; nothing is fatal.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "dsp_device.hpp"

namespace dsp {

struct RtcKernel {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
};

struct RtcApi {
  void *lib = nullptr;
  hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **) = nullptr;
  hiprtcResult (*add_name)(hiprtcProgram, const char *) = nullptr;
  hiprtcResult (*compile)(hiprtcProgram, int, const char **) = nullptr;
  hiprtcResult (*lowered)(hiprtcProgram, const char *, const char **) = nullptr;
  hiprtcResult (*code_size)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*code)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*log_size)(hiprtcProgram, size_t *) = nullptr;
  hiprtcResult (*log)(hiprtcProgram, char *) = nullptr;
  hiprtcResult (*destroy)(hiprtcProgram *) = nullptr;
  bool ok = false;
};

inline const RtcApi &rtc_api() {
  static RtcApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
#define DSP_RTC_SYM(field, sym) *(void **)(&api.field) = dlsym(api.lib, sym)
    DSP_RTC_SYM(create, "hiprtcCreateProgram");
    DSP_RTC_SYM(add_name, "hiprtcAddNameExpression");
    DSP_RTC_SYM(compile, "hiprtcCompileProgram");
    DSP_RTC_SYM(lowered, "hiprtcGetLoweredName");
    DSP_RTC_SYM(code_size, "hiprtcGetCodeSize");
    DSP_RTC_SYM(code, "hiprtcGetCode");
    DSP_RTC_SYM(log_size, "hiprtcGetProgramLogSize");
    DSP_RTC_SYM(log, "hiprtcGetProgramLog");
    DSP_RTC_SYM(destroy, "hiprtcDestroyProgram");
#undef DSP_RTC_SYM
    api.ok = api.create && api.add_name && api.compile && api.lowered && api.code_size && api.code && api.log_size && api.log && api.destroy;
  });
  return api;
}

// csrc/ directory: $DSP_KERNEL_SRC, or "csrc" next to the shared object this code lives in
inline std::string rtc_source_dir() {
  if (const char *e = getenv("DSP_KERNEL_SRC")) return e;
  Dl_info info;
  if (dladdr((const void *)&rtc_source_dir, &info) && info.dli_fname) {
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/csrc";
  }
  return "csrc";
}

// "" = no disk cache: without $DSP_RTC_CACHE and without $HOME there is no private place for it (a world-writable /tmp
// directory would let another user plant code objects that this process loads)
inline std::string rtc_cache_dir() {
  if (const char *e = getenv("DSP_RTC_CACHE")) return e;
  const char *home = getenv("HOME");
  return home && *home ? std::string(home) + "/.cache/dsp_hip" : std::string();
}

// the development switches this library was compiled with, as compiler options for the run-time kernel (same switches on
// both sides; they are part of the layout token and of the cache key)
inline std::vector<std::string> rtc_build_defines() {
  std::vector<std::string> d;
  if (kBuildSwitches & DSP_SW_TRACE) d.push_back("-DDSP_KKT_TRACE");
  if (kBuildSwitches & DSP_SW_PROF) d.push_back("-DDSP_PROF");
  if (kBuildSwitches & DSP_SW_CLOCKS) d.push_back("-DDSP_CLOCKS");
  if (kBuildSwitches & DSP_SW_PULL) d.push_back("-DDSP_LEGACY_PULL");
  if (kBuildSwitches & DSP_SW_NOJUMP) d.push_back("-DDSP_NO_JUMP");
  return d;
}

inline uint64_t rtc_hash_file(const std::string &path, uint64_t h) {
  if (FILE *f = fopen(path.c_str(), "rb")) {
    unsigned char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0)
      for (size_t i = 0; i < n; ++i) { h ^= buf[i]; h *= 1099511628211ull; }
    fclose(f);
  }
  return h;
}

// FNV-1a over the kernel sources: a cached code object is only re-used for the sources it was compiled from
inline uint64_t rtc_source_hash(const std::string &dir) {
  uint64_t h = 1469598103934665603ull;
  for (const char *f : {"/dsp_kernels.hip", "/dsp_device.hpp", "/dsp_wave.hpp", "/dsp_shapes.hpp", "/../../include/dsp_hip.h"})
    h = rtc_hash_file(dir + f, h);
  return h;
}

inline std::string rtc_kernel_expr(int cpl, int rpl, bool lng, unsigned wc, unsigned wr, bool qp) {
  char buf[160];
  snprintf(buf, sizeof buf, "&dsp::pdlp_solve_kernel<%d, %d, %s, %uu, %uu, %s>", cpl, rpl, lng ? "true" : "false", wc, wr,
           qp ? "true" : "false");
  return buf;
}

// Compile (or fetch from the disk cache) the code object of one instantiation.  Returns false (with a reason in *why) when
// run-time compilation is not possible here.  No GPU is needed for this step.
inline bool rtc_build_code(int cpl, int rpl, bool lng, unsigned wc, unsigned wr, bool qp, std::vector<char> *code,
                           std::string *lowered_name, std::string *why) {
  const RtcApi &api = rtc_api();
  if (!api.ok) { *why = "libhiprtc.so not available"; return false; }
  const std::string dir = rtc_source_dir();
  struct stat sb;
  if (stat((dir + "/dsp_kernels.hip").c_str(), &sb) != 0) { *why = "kernel sources not found in " + dir; return false; }
  const std::string expr = rtc_kernel_expr(cpl, rpl, lng, wc, wr, qp);
  char key[240];
  // key: shape, sources, and the library's layout token (structure sizes / offsets, ABI version, development switches)
  snprintf(key, sizeof key, "pdlp_%d_%d_%d_%x_%x_%d_%016llx_%016llx", cpl, rpl, (int)lng, wc, wr, (int)qp,
           (unsigned long long)rtc_source_hash(dir), (unsigned long long)kSolveArgsToken);
  const std::string cdir = rtc_cache_dir(), cpath = cdir + "/" + key + ".hsaco", npath = cdir + "/" + key + ".name";
  // disk cache
  if (FILE *f = cdir.empty() ? nullptr : fopen(cpath.c_str(), "rb")) {
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    code->resize(sz > 0 ? (size_t)sz : 0);
    const bool ok = sz > 0 && fread(code->data(), 1, (size_t)sz, f) == (size_t)sz;
    fclose(f);
    if (ok)
      if (FILE *g = fopen(npath.c_str(), "rb")) {
        char nb[512] = {0};
        const size_t n = fread(nb, 1, sizeof nb - 1, g);
        fclose(g);
        if (n > 0) { *lowered_name = std::string(nb, n); return true; }
      }
  }
  const std::string src = "#include \"dsp_kernels.hip\"\n";
  hiprtcProgram prog = nullptr;
  if (api.create(&prog, src.c_str(), "dsp_rtc_unit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { *why = "hiprtcCreateProgram failed"; return false; }
  api.add_name(prog, expr.c_str());
  const std::string inc = "-I" + dir;
  const std::vector<std::string> defs = rtc_build_defines();
  std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", inc.c_str()};
  for (const std::string &d : defs) opts.push_back(d.c_str());
  const hiprtcResult r = api.compile(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    api.log_size(prog, &ls);
    std::string lg(ls, 0);
    if (ls) api.log(prog, &lg[0]);
    *why = "hiprtc compilation failed: " + lg.substr(0, 2000);
    api.destroy(&prog);
    return false;
  }
  const char *low = nullptr;
  size_t cs = 0;
  if (api.lowered(prog, expr.c_str(), &low) != HIPRTC_SUCCESS || !low || api.code_size(prog, &cs) != HIPRTC_SUCCESS || cs == 0) {
    *why = "hiprtc produced no code";
    api.destroy(&prog);
    return false;
  }
  *lowered_name = low;
  code->resize(cs);
  api.code(prog, code->data());
  api.destroy(&prog);
  // best-effort cache write (atomic rename; a failure just means the next process compiles again); private directory
  if (cdir.empty()) return true;
  if (!getenv("DSP_RTC_CACHE")) mkdir((std::string(getenv("HOME")) + "/.cache").c_str(), 0700);
  mkdir(cdir.c_str(), 0700);
  const std::string tmp = cpath + ".tmp" + std::to_string((long)getpid());
  if (FILE *f = fopen(tmp.c_str(), "wb")) {
    const bool ok = fwrite(code->data(), 1, code->size(), f) == code->size();
    fclose(f);
    if (ok && rename(tmp.c_str(), cpath.c_str()) == 0) {
      if (FILE *g = fopen(npath.c_str(), "wb")) { fwrite(lowered_name->data(), 1, lowered_name->size(), g); fclose(g); }
    } else {
      remove(tmp.c_str());
    }
  }
  return true;
}

// process-wide table of loaded specialisations (per device)
inline bool rtc_get_kernel(int device, int cpl, int rpl, bool lng, unsigned wc, unsigned wr, bool qp, RtcKernel *out, std::string *why) {
  static std::mutex mu;
  static std::map<std::string, RtcKernel> table;
  char key[128];
  snprintf(key, sizeof key, "%d:%d:%d:%d:%x:%x:%d", device, cpl, rpl, (int)lng, wc, wr, (int)qp);
  std::lock_guard<std::mutex> lock(mu);
  auto it = table.find(key);
  if (it != table.end()) { *out = it->second; return out->fn != nullptr; }
  RtcKernel k;
  std::vector<char> code;
  std::string name;
  if (rtc_build_code(cpl, rpl, lng, wc, wr, qp, &code, &name, why)) {
    if (hipModuleLoadData(&k.mod, code.data()) != hipSuccess) { *why = "hipModuleLoadData failed"; k = RtcKernel{}; }
    else if (hipModuleGetFunction(&k.fn, k.mod, name.c_str()) != hipSuccess) { *why = "hipModuleGetFunction failed for " + name; k = RtcKernel{}; }
    else {
      // the code object must have been compiled against THIS library's SolveArgs layout and switches (dsp_device.hpp)
      hipDeviceptr_t tok = nullptr;
      size_t tok_bytes = 0;
      unsigned long long theirs = 0;
      if (hipModuleGetGlobal(&tok, &tok_bytes, k.mod, "dsp_rtc_layout_token") != hipSuccess || tok_bytes != sizeof(theirs) ||
          hipMemcpy(&theirs, tok, sizeof(theirs), hipMemcpyDeviceToHost) != hipSuccess || theirs != kSolveArgsToken) {
        char msg[200];
        snprintf(msg, sizeof msg, "layout token mismatch (library %016llx, code object %016llx): stale libdsp_hip.so or other build "
                 "switches than the kernel sources - run-time kernel refused", (unsigned long long)kSolveArgsToken, theirs);
        *why = msg;
        (void)hipModuleUnload(k.mod);
        k = RtcKernel{};
      }
    }
  }
  table[key] = k;            // failures are remembered too: one attempt per shape and process
  *out = k;
  return k.fn != nullptr;
}

}  // namespace dsp

#!/bin/bash
# Development: VGPRs / scratch / occupancy and the scratch instructions on the loops of a few solve-kernel instantiations, in seconds and
# without a GPU - what a change to a rare block does to the allocation of the common paths (the certificate experiments of round 5:
# profiles/r50_kernel_resources.log).      bash tools/kernel_resources.sh [csrc directory] [tag]
src=${1:-$(cd "$(dirname "$0")/../dispatches_amd/csrc" && pwd)}; tag=${2:-cur}; out=${TMPDIR:-/tmp}/kres; mkdir -p $out
cat > $out/$tag.hip <<EOT
#define DSP_KERNELS_ONLY
#include "$src/dsp_kernels.hip"
namespace dsp {
template __global__ void pdlp_solve_kernel<4, 2, false, 0x1133u, 0x44u, false>(SolveArgs);           // wind + battery 24 h (metric)
template __global__ void pdlp_solve_kernel<7, 4, false, 0x1112333u, 0x2444u, false>(SolveArgs);      // wind + battery 48 h
template __global__ void pdlp_solve_kernel<4, 3, true, 0x1122u, 0x233u, false>(SolveArgs);           // wind + PEM 48 h
template __global__ void pdlp_solve_kernel<3, 2, false, 0x122u, 0x24u, false>(SolveArgs);            // nuclear 24 h
template __global__ void pdlp_solve_kernel<1, 1, false, 0x3u, 0x4u, false>(SolveArgs);               // 4-h hourly LPs
template __global__ void pdlp_solve_kernel<4, 2, false, 0u, 0u, false>(SolveArgs);                   // generic (LDS matrix, with certificates)
}
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $KRES_FLAGS -S --cuda-device-only -I$src -o $out/$tag.s $out/$tag.hip -Rpass-analysis=kernel-resource-usage 2> $out/$tag.res
python - $out/$tag.s $out/$tag.res <<'PY'
import re, sys
src = open(sys.argv[1]).read(); res = open(sys.argv[2]).read()
for blk in res.split("Function Name: ")[1:]:
    name = blk.split("\n")[0].split(" [")[0].strip()
    if "pdlp_solve" not in name: continue
    g = lambda k: int(re.search(k + r": (\d+)", blk).group(1))
    i = src.index("\n" + name + ":"); body = src[i:src.index(".Lfunc_end", i)].split("\n")
    labels = {m.group(1): k for k, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l) or re.search(r"s_branch (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k:
            seg = body[labels[m.group(1)]:k + 1]
            loops.append((len(seg), sum("v_fma_f64" in s for s in seg), sum("scratch_" in s for s in seg)))
    loops = sorted(L for L in loops if L[1] >= 8)[:3]
    scr, occ = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")
    print(f"{name[28:62]:36s} VGPRs {g('VGPRs')} scratch {scr:4d} B/lane occupancy {occ} | scratch instructions {sum('scratch_' in s for s in body):4d} "
          f"| innermost loops (lines, FMAs, scratch instructions) {loops}")
PY

// Test harness for the host-side preparation (dispatches_amd/csrc/dsp_prepare.hpp), built by tests/test_prepare_cpu.py with g++.
// Reads a CSR (int32 m, n, nnz; ptr[m+1]; idx[nnz]; double val[nnz]) and the lane slots (cpl, rpl), reproduces what
// dsp_create does on the host, emulates the kernel's gathers through the register-resident layout + slot map and prints
// one JSON object with the checks.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "../dispatches_amd/csrc/dsp_prepare.hpp"

using namespace dsp;

// y = M v through the SlotELL of M (vectors owned via `own`, gathered vector stored at slot[other position])
static std::vector<double> ell_apply(const HostCSR &M, const SlotELL &E, const SortedLayout &own, const SortedLayout &other,
                                     const std::vector<int32_t> &slot, const std::vector<double> &v) {
  std::vector<double> buf(slot.size(), 0.0), out(M.m, 0.0);
  for (int j = 0; j < M.n; ++j) buf[slot[other.pos[j]]] = v[j];            // the owner lanes' ds_write_b64
  int base = 0;
  for (int q = 0; q < own.slots; ++q) {
    for (int l = 0; l < 64; ++l) {
      double acc = 0.0;
      for (int e = 0; e < E.width[q]; ++e) {
        size_t at = (size_t)(base + e) * 64 + l;
        acc += E.val[at] * buf[E.off[at] / 8u];                            // ds_read_b64 gather + FMA
      }
      int32_t id = own.at[q * 64 + l];
      if (id >= 0) out[id] += acc;
    }
    base += E.width[q];
  }
  for (size_t k = 0; k < E.long_owner_pos.size(); ++k) {
    double acc = 0.0;
    for (int e = 0; e < E.long_len[k]; ++e) acc += E.tail_val[E.long_start[k] + e] * buf[E.tail_off[E.long_start[k] + e] / 8u];
    out[own.at[E.long_owner_pos[k]]] += acc;
  }
  return out;
}

int main(int argc, char **argv) {
  if (argc < 4) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  int cpl = atoi(argv[2]), rpl = atoi(argv[3]);
  int32_t hdr[3];
  if (fread(hdr, 4, 3, f) != 3) return 4;
  HostCSR A;
  A.m = hdr[0]; A.n = hdr[1];
  A.ptr.resize(A.m + 1); A.idx.resize(hdr[2]); A.val.resize(hdr[2]);
  if (fread(A.ptr.data(), 4, A.ptr.size(), f) != A.ptr.size() || fread(A.idx.data(), 4, A.idx.size(), f) != A.idx.size() ||
      fread(A.val.data(), 8, A.val.size(), f) != A.val.size()) return 5;
  fclose(f);
  HostCSR Au = A;
  std::vector<double> dr, dc;
  equilibrate(A, 10, dr, dc, 0);
  // scaled matrix = D_r A D_c
  double scale_err = 0.0;
  for (int i = 0; i < A.m; ++i)
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
      scale_err = std::max(scale_err, std::fabs(A.val[p] - dr[i] * Au.val[p] * dc[A.idx[p]]));
  HostCSR AT = transpose(A);
  double norm2 = spectral_norm(A, AT, 500);
  LaneELL Er = build_lane_ell(A, rpl), Ec = build_lane_ell(AT, cpl);
  SortedLayout Lc = sorted_layout(AT, cpl, Ec.long_owner), Lr = sorted_layout(A, rpl, Er.long_owner);
  SlotELL Sc = build_slot_ell(AT, Lc, Lr, Ec.long_owner), Sr = build_slot_ell(A, Lr, Lc, Er.long_owner);
  auto t0 = std::chrono::steady_clock::now();
  const bool shared = shared_slot_maps(cpl, rpl);
  SlotMap My = optimise_slots(Sc, rpl * 64, 4000, shared), Mx = optimise_slots(Sr, cpl * 64, 4000, shared);
  double search_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  apply_slots(Sc, My.slot);
  apply_slots(Sr, Mx.slot);
  // invariants of the slot maps: bijection within every 32-slot block, store groups distinct mod 16
  int bad_perm = 0, bad_store = 0;
  for (const SlotMap *M : {&Mx, &My}) {
    const auto &s = M->slot;
    for (size_t b = 0; b < s.size() / 32; ++b) {
      unsigned seen = 0;
      for (int i = 0; i < 32; ++i) {
        if ((size_t)(s[b * 32 + i] >> 5) != b) bad_perm++;
        seen |= 1u << (s[b * 32 + i] & 31);
      }
      if (seen != 0xFFFFFFFFu) bad_perm++;
      for (int g = 0; g < 2; ++g) {
        unsigned m16 = 0;
        for (int i = 0; i < 16; ++i) m16 |= 1u << (s[b * 32 + g * 16 + i] & 15);
        if (m16 != 0xFFFFu) bad_store++;
      }
    }
  }
  // shared maps (shapes with more than 8 owned elements per lane): the same map in every 64-position block
  int bad_shared = 0;
  if (shared)
    for (const SlotMap *M : {&Mx, &My})
      for (size_t p = 64; p < M->slot.size(); ++p)
        if (M->slot[p] != M->slot[p & 63] + (int32_t)(p & ~(size_t)63)) bad_shared++;
  // the layouts compute the same products as the CSR
  std::vector<double> x(A.n), y(A.m);
  for (int j = 0; j < A.n; ++j) x[j] = std::sin(0.37 * j + 1.0);
  for (int i = 0; i < A.m; ++i) y[i] = std::cos(0.53 * i + 2.0);
  std::vector<double> ax = ell_apply(A, Sr, Lr, Lc, Mx.slot, x), aty = ell_apply(AT, Sc, Lc, Lr, My.slot, y);
  double err = 0.0;
  for (int i = 0; i < A.m; ++i) {
    double r = 0.0;
    for (int p = A.ptr[i]; p < A.ptr[i + 1]; ++p) r += A.val[p] * x[A.idx[p]];
    err = std::max(err, std::fabs(r - ax[i]));
  }
  for (int j = 0; j < AT.m; ++j) {
    double r = 0.0;
    for (int p = AT.ptr[j]; p < AT.ptr[j + 1]; ++p) r += AT.val[p] * y[AT.idx[p]];
    err = std::max(err, std::fabs(r - aty[j]));
  }
  printf("{\"scale_err\": %.3e, \"norm2\": %.15g, \"product_err\": %.3e, \"bad_perm\": %d, \"bad_store\": %d, "
         "\"conflicts_y\": [%d, %d, %d], \"conflicts_x\": [%d, %d, %d], \"search_ms\": %.1f, \"pack_c\": %u, \"pack_r\": %u, "
         "\"long_c\": %zu, \"long_r\": %zu, \"shared\": %d, \"bad_shared\": %d}\n",
         scale_err, norm2, err, bad_perm, bad_store, My.cost_identity, My.cost_rotation, My.cost_final, Mx.cost_identity,
         Mx.cost_rotation, Mx.cost_final, search_ms, Sc.pack, Sr.pack, Sc.long_owner_pos.size(), Sr.long_owner_pos.size(), (int)shared,
         bad_shared);
  // the scaling vectors for the python side
  for (double d : dr) printf("%.17g ", d);
  printf("\n");
  for (double d : dc) printf("%.17g ", d);
  printf("\n");
  return 0;
}

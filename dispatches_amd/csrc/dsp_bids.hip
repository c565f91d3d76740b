// dsp_bids.hip — the scenario half of the Bidder's bid assembly on the device, gfx950 only.
//
// Reference behaviour (idaes Bidder._assemble_bids as the DISPATCHES double loop drives it; SURVEY.md A.4, golden G2:
// dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:245-250): per hour, the (power, marginal
// price) pairs of all price scenarios, each number rounded with Python's round(x, 2); pairs below p_min dropped; per distinct power the
// highest price kept; sorted by power.  For 4096 scenarios x 24 h that is 98 k exact decimal roundings and 24 sorts - after the 2-ms
// solve the largest single piece of compute_day_ahead_bids (as ~25 tensor launches: workflow/bid_curves.py, which stays as the form for
// batches this kernel does not take and as the CPU-testable statement of the same arithmetic).
//
// ONE launch, one workgroup per hour:
//   * every thread forms its scenarios' pairs from the solution still in HBM (x[s][col], or the one- / two-term power expression of the
//     real-time models, in the operation order of ScenarioBatchModel.expression_values), rounds both numbers to integer cents EXACTLY
//     as round(x, 2) does - the correctly rounded decimal of the double, ties to even: a * 200 as an exact double-double (Dekker /
//     Veltkamp, contraction off: an FMA here would change nothing it computes but the file takes no chances) compared with the odd
//     integer 2 r + 1 - and packs (power cents, price cents) into one signed 64-bit key: power ascending, price DESCENDING;
//   * bitonic sort of the hour's keys in LDS (8 B per scenario: 32 KB at 4096, the 128 KB limit is 16 384 scenarios);
//   * the first key of every run of equal powers carries the run's highest price: flags, one block-wide exclusive scan, and the
//     distinct points go out in order, the count in front of them.
// HBM traffic: B x T x 16 B in (solution columns and prices, read once), <= B x T x 8 B out; the kernel is launch- and sync-latency,
// not bandwidth (about 20 us at 4096 x 24).  What is left for the host are the distinct points of an hour (workflow/bidder.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dsp_hip.h"
#include "dsp_device.hpp"

namespace dsp {

constexpr long long kBidDrop = 0x7fffffffffffffffll;        // key of a pair that takes no part: sorts behind every real key
constexpr long long kBidOff = 1ll << 31;

// round(a, 2) * 100 as an integer, for finite |a| < 2e7 (anything else: 0, and the caller drops the pair).
__device__ __forceinline__ long long bid_cents(double a) {
#pragma clang fp contract(off)
  const double p = a * 200.0;                      // rounded product
  const double c = a * 134217729.0;                // Veltkamp split (2^27 + 1)
  const double hi = c - (c - a);
  const double lo = a - hi;
  const double err = (hi * 200.0 - p) + lo * 200.0;   // exact: a * 200 = p + err
  double r = floor(a * 100.0);                     // the exact floor is r or r +- 1
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (((p - 2.0 * r) + err) < 0.0) r -= 1.0;
    if (((p - (2.0 * r + 2.0)) + err) >= 0.0) r += 1.0;
  }
  const double d = (p - (2.0 * r + 1.0)) + err;    // sign of a * 100 - (r + 1/2), exact
  const bool odd = fmod(r, 2.0) != 0.0;
  if (d > 0.0 || (d == 0.0 && odd)) r += 1.0;      // above the midpoint, or on it with an odd floor: ties to even
  const bool ok = (fabs(a) < 2.0e7);               // false for NaN / inf too
  return ok ? (long long)r : 0ll;
}

__global__ __launch_bounds__(1024) void bid_points_kernel(dsp_bid_request rq, int N) {
#pragma clang fp contract(off)
  extern __shared__ long long keys[];              // [N]
  __shared__ int wave_total[16];
  const int t = blockIdx.x, tid = threadIdx.x, B = rq.B;
  const int c0 = rq.col[t][0], c1 = rq.col[t][1];
  const double v0 = rq.val[t][0], v1 = rq.val[t][1], k0 = rq.constant[t];
  for (int s = tid; s < N; s += 1024) {
    long long key = kBidDrop;
    if (s < B && (!rq.ok || rq.ok[s])) {
      const double *xs = rq.x + (size_t)s * rq.ldx;
      double power = xs[c0];
      if (rq.terms == 1) power = power * v0 + k0;
      else if (rq.terms == 2) power = (power * v0 + xs[c1] * v1) + k0;
      const double price = rq.price[(size_t)s * rq.ldp + t];
      const long long pc = bid_cents(power), cc = bid_cents(price);
      const bool keep = ((double)pc / 100.0 >= rq.p_min) && fabs(power) < INFINITY && fabs(price) < INFINITY;
      if (keep) key = pc * 4294967296ll + ((kBidOff - 1) - cc);      // low half in [0, 2^32): price descending inside a power
    }
    keys[s] = key;
  }
  __syncthreads();
  // bitonic sort, ascending
  const int half = N >> 1;
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < half; i += 1024) {
        const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo + j;
        const long long a = keys[lo], b = keys[hi];
        if ((a > b) == ((lo & k) == 0)) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  // distinct powers, in order: thread `tid` owns the E consecutive keys from tid * E
  const int E = N > 1024 ? N >> 10 : 1;
  const int base = tid * E;
  int mine = 0;
  unsigned flags = 0;
  if (base < N) {
    long long prev = base > 0 ? keys[base - 1] : kBidDrop;
    for (int e = 0; e < E; ++e) {
      const long long key = keys[base + e];
      const bool first = key != kBidDrop && (base + e == 0 || (key >> 32) != (prev >> 32));
      flags |= (unsigned)first << e;
      mine += first;
      prev = key;
    }
  }
  int incl = mine;                                  // inclusive scan over the wave, then over the 16 waves
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    const int v = wave_total[w];
    before += w < wave ? v : 0;
    total += v;
  }
  long long *out = (long long *)rq.out + (size_t)t * (B + 1);
  if (tid == 0) out[0] = total;
  int pos = before + incl - mine;
  if (base < N)
    for (int e = 0; e < E; ++e)
      if (flags >> e & 1u) {
        const long long key = keys[base + e];
        const long long pc = key >> 32, cc = (kBidOff - 1) - (key & 0xffffffffll);
        out[1 + pos++] = pc * 4294967296ll + (cc & 0xffffffffll);        // high half: power cents, low half: price cents (two's complement)
      }
}

hipError_t launch_bid_points(const dsp_bid_request &rq, hipStream_t st) {
  int N = 64;
  while (N < rq.B) N <<= 1;
  const size_t lds = (size_t)N * sizeof(long long);
  if (lds > 64 * 1024) {       // (per call: the attribute belongs to the CURRENT device's code object - a process-wide flag left a second device at 64 KB)
    hipError_t e = hipFuncSetAttribute((const void *)bid_points_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(bid_points_kernel, dim3(rq.T), dim3(1024), lds, st, rq, N);
  return hipGetLastError();
}

}  // namespace dsp

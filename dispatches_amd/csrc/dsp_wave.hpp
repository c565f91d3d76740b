// dsp_wave.hpp — wave64 device helpers shared by the kernels (dsp_kernels.hip, dsp_simplex.hip): single-wave LDS fence,
// VALU (DPP) reductions, LDS access through 32-bit addresses.  gfx950 only.
#pragma once
#ifdef __HIPCC_RTC__     /* run-time compilation (hiprtc, dsp_rtc.cpp): no system headers, the HIP device API is built in */
#ifndef DSP_RTC_TYPES
#define DSP_RTC_TYPES
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t; typedef unsigned long size_t;
#ifndef INFINITY
#define INFINITY (__builtin_inff())
#endif
#ifndef NAN
#define NAN (__builtin_nanf(""))
#endif
#endif
#else
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#endif

namespace dsp {

// ---- wave-level helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
  // Single-wave producer/consumer through LDS: the LDS unit executes one wave's DS ops in order, so only the
  // COMPILER must be kept from reordering the exchange-buffer stores and gathers.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ---- wave64 reductions on the VALU (DPP), not through the LDS crossbar ---------------------------------------------
// Four DPP steps make every lane of a 16-lane row hold its row's total (xor-1, xor-2 quad permutes, half-row mirror,
// row mirror: the operation is commutative, so mirrored partners may be used); the four row totals are then read with
// v_readlane into SGPRs.  ~100 cycles of dependent latency instead of six ds_bpermute round trips (~400), and the
// result is wave-uniform by construction.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

struct OpSum { __device__ __forceinline__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ __forceinline__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ __forceinline__ static double f(double a, double b) { return fmin(a, b); } };

template <class Op, int N>
__device__ __forceinline__ void wave_reduce(double (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = Op::f(a[i], dpp_f64<kDppXor1>(a[i]));
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = Op::f(a[i], dpp_f64<kDppXor2>(a[i]));
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = Op::f(a[i], dpp_f64<kDppHalfMirror>(a[i]));
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = Op::f(a[i], dpp_f64<kDppMirror>(a[i]));
#pragma unroll
  for (int i = 0; i < N; ++i)
    a[i] = Op::f(Op::f(lane_f64(a[i], 0), lane_f64(a[i], 16)), Op::f(lane_f64(a[i], 32), lane_f64(a[i], 48)));
}
__device__ __forceinline__ double wave_sum(double v) { double a[1] = {v}; wave_reduce<OpSum, 1>(a); return a[0]; }
__device__ __forceinline__ double wave_min(double v) { double a[1] = {v}; wave_reduce<OpMin, 1>(a); return a[0]; }
__device__ __forceinline__ double wave_max(double v) { double a[1] = {v}; wave_reduce<OpMax, 1>(a); return a[0]; }
// N sums at once (independent chains interleave)
template <int N>
__device__ __forceinline__ void wave_sums(double (&a)[N]) { wave_reduce<OpSum, N>(a); }

__device__ __forceinline__ double clampd(double v, double lo, double hi) { return fmin(fmax(v, lo), hi); }
// clampd for the hot loop: the bare v_max_f64 / v_min_f64 pair.  fmax / fmin on loop-carried operands make the
// compiler quiet them first (v_max_f64 x, x, x: a 4-cycle FP64 slot each); the operands here are never signalling NaNs.
__device__ __forceinline__ double clampd_bare(double v, double lo, double hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  double t, r;
  asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(lo));
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(t), "v"(hi));
  return r;
#else
  return clampd(v, lo, hi);
#endif
}
__device__ __forceinline__ double finite_or_zero(double v) { return (fabs(v) < INFINITY) ? v : 0.0; }
__device__ __forceinline__ bool is_finite(double v) { return fabs(v) < INFINITY; }

// 64-bit load from a 32-bit LDS byte address (device pass only; the host pass never executes it)
__device__ __forceinline__ double lds_load_f64(uint32_t addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(3))) double *)addr;
#else
  (void)addr;
  return 0.0;
#endif
}

__device__ __forceinline__ void lds_store_f64(uint32_t addr, double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  *(__attribute__((address_space(3))) double *)addr = v;
#else
  (void)addr; (void)v;
#endif
}
}  // namespace dsp

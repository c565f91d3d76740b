// Memory-system probe for the lane form's access pattern (development; standalone):
//   hipcc --offload-arch=gfx950 -O3 -o lane_stream_probe tools/probes/lane_stream_probe.hip && ./lane_stream_probe [groups] [waves]
// A wave walks `rows` consecutive elements of 7 lane-layout arrays ([element][64 lanes] doubles): per element it reads 5 (x, x0, c, y,
// y0), adds them and writes 2 (x', y') - the traffic of one PDHG iteration of the lane form (4 n + 3 m doubles per scenario) with
// none of its arithmetic, LDS or records.  Variants: 8 bytes per lane and load (the kernel's layout) against 16 (two elements per
// lane: a pair-interleaved layout), requests one chunk of 4 elements ahead or not.  Prints GB/s of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { const double *in[5]; double *out[2]; int len, rows, ntile; };

template <int W, bool AHEAD>   // W = doubles per lane and load (1 or 2)
__global__ void __launch_bounds__(256) k_walk(Args a) {
  typedef double V __attribute__((ext_vector_type(W)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = blockIdx.y;
  const int tile = blockIdx.x * 4 + wv;
  if (tile >= a.ntile) return;
  const size_t gb = (size_t)g * a.len * 64;
  const int e0 = tile * a.rows, e1 = min(a.len, e0 + a.rows);
  constexpr int CH = 4 / W;                       // loads per array and chunk of 4 elements
  auto at = [&](const double *p, int e) { return reinterpret_cast<const V *>(p + gb + (size_t)e * 64) + lane; };
  V r[2][5][CH];
  auto load = [&](int s, int e) {
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
      for (int k = 0; k < CH; ++k) r[s][q][k] = *at(a.in[q], min(e + k * W, a.len - W));
  };
  if (AHEAD) load(0, e0);
  int s = 0;
  for (int e = e0; e < e1; e += 4) {
    if (AHEAD) { if (e + 4 < e1) load(s ^ 1, e + 4); } else load(s, e);
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const V xv = r[s][0][k] + r[s][1][k] * r[s][2][k], yv = r[s][3][k] - r[s][4][k];
      if (e + k * W < e1) {
        *const_cast<V *>(at(a.out[0], e + k * W)) = xv;
        *const_cast<V *>(at(a.out[1], e + k * W)) = yv;
      }
    }
    if (AHEAD) s ^= 1;
  }
}

int main(int argc, char **argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 4, waves = argc > 2 ? atoi(argv[2]) : 1024;
  const int len = 52420;                                                                 // (a multiple of 4)
  int rows = (int)(((long)len * G + waves - 1) / waves); rows = (rows + 3) / 4 * 4;
  const int ntile = (len + rows - 1) / rows;
  Args a{}; a.len = len; a.rows = rows; a.ntile = ntile;
  const size_t bytes = (size_t)G * len * 64 * 8;
  for (int q = 0; q < 5; ++q) { double *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); a.in[q] = p; }
  for (int q = 0; q < 2; ++q) { double *p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 0, bytes)); a.out[q] = p; }
  const dim3 grid((ntile + 3) / 4, G), block(256);
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const double total = 7.0 * bytes;
  printf("groups %d (batch %d), %d tiles of %d elements x %d groups = %d waves, %.0f MB per pass\n", G, 64 * G, ntile, rows, G, ntile * G, total / 1e6);
  auto run = [&](const char *name, void (*k)(Args)) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, block, 0, 0, a);
    hipEventRecord(t0, 0);
    const int reps = 30;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, block, 0, 0, a);
    hipEventRecord(t1, 0); hipEventSynchronize(t1);
    float ms = 0; hipEventElapsedTime(&ms, t0, t1);
    printf("  %-42s %8.1f us per pass  %7.0f GB/s\n", name, 1e3 * ms / reps, total / (1e-3 * ms / reps) / 1e9);
    return 0;
  };
  run("8 B per lane, requests with the use", k_walk<1, false>);
  run("8 B per lane, one chunk ahead", k_walk<1, true>);
  run("16 B per lane, requests with the use", k_walk<2, false>);
  run("16 B per lane, one chunk ahead", k_walk<2, true>);
  // reference: a plain float4 copy of the same number of bytes (3.5 arrays read, 3.5 written)
  return 0;
}

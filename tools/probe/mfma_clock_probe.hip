// Does the MFMA shape change what the power management makes of the clock?  Back-to-back bf16 MFMAs on every SIMD
// (two waves each, independent accumulators, operands with random mantissas), 16x16x32 against 32x32x16: same MACs per
// second at the same clock, half the operand-register reads per MAC for the larger shape.  Prints the achieved rate;
// run it beside `rocm-smi --showclocks --showpower` (tools/probe/mfma_clock_probe.sh).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_clock_probe.hip -o tools/probe/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ bf16x8 operand(unsigned seed) {
  bf16x8 v;
  unsigned s = seed * 2654435761u + 12345u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    v[i] = (__bf16)(((int)(s >> 16) - 32768) * (1.0f / 32768.0f));
  }
  return v;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  const unsigned t = blockIdx.x * 512 + threadIdx.x;
  bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = operand(t * 8 + i); b[i] = operand(t * 8 + 4 + i); }
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][3];
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      // the same MACs per trip: 8 x (32 x 32 x 16) = 16 x (16 x 16 x 32)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 3], b[j >> 1], acc[j & 3], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][15];
  }
  if (s == 12345.678f) out[t] = s;
}

int main(int argc, char** argv) {
  const int shape = argc > 1 ? atoi(argv[1]) : 16;
  const double secs = argc > 2 ? atof(argv[2]) : 4.0;
  float* out;
  if (hipMalloc(&out, 256 * 512 * 4) != hipSuccess) return 1;
  const int iters = 20000;
  const double macs = 256.0 * 8 * 16 * 8192.0 * iters;   // per launch: 256 workgroups x 8 waves x 16 MFMAs of 8192 MACs
  auto t0 = std::chrono::steady_clock::now();
  int n = 0;
  double last = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) {
      if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(256), dim3(512), 0, 0, out, iters);
      else hipLaunchKernelGGL(mfma_loop<32>, dim3(256), dim3(512), 0, 0, out, iters);
    }
    hipEventRecord(e1);
    if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    last = 2.0 * macs * 10 / (ms * 1e-3) / 1e12;
    hipEventDestroy(e0); hipEventDestroy(e1);
    n += 10;
  }
  printf("shape %dx%d: %d launches, last batch %.1f TFLOP/s\n", shape, shape, n, last);
  hipFree(out);
  return 0;
}

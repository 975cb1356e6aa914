// Micro-benchmark behind the sampler's row-load phase: how fast can one workgroup per row
// bring a [M, V] fp32 matrix through the CU, as a function of LDS footprint, persistence and prefetch.
// Build: hipcc -O3 --offload-arch=gfx950 row_load_probe.hip -o row_load_probe ; run: ./row_load_probe [M] [V]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ST = 512;

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < ST / 64; ++w) r = fmaxf(r, red[w]);
  return r;
}

// mode bits: 1 = store the row to LDS, 2 = divide by T
template <int MODE>
__global__ __launch_bounds__(ST) void one_row_per_block(const float* __restrict__ x, int V, float T, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* row = (float*)smem;
  __shared__ float red[ST / 64];
  const int tid = threadIdx.x, m = blockIdx.x;
  const float* lr = x + (int64_t)m * V;
  const int v4 = V >> 2;
  float mx = -INFINITY;
  constexpr int U = 8;
  for (int base = tid; base < v4; base += ST * U) {
    float4 buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = base + u * ST; if (i < v4) buf[u] = *(const float4*)(lr + 4 * i); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * ST;
      if (i < v4) {
        float4 l = buf[u];
        if (MODE & 2) { l.x /= T; l.y /= T; l.z /= T; l.w /= T; }
        if (MODE & 1) *(float4*)(row + 4 * i) = l;
        mx = fmaxf(fmaxf(mx, fmaxf(l.x, l.y)), fmaxf(l.z, l.w));
      }
    }
  }
  mx = block_max(mx, red, tid);
  if (tid == 0) out[m] = mx;
}

// persistent: each block walks rows m = blockIdx.x, += gridDim.x; the next row is fetched into registers before the
// current one is written to LDS and reduced
template <int MODE>
__global__ __launch_bounds__(ST) void persistent_prefetch(const float* __restrict__ x, int M, int V, float T, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* row = (float*)smem;
  __shared__ float red[ST / 64];
  const int tid = threadIdx.x;
  const int v4 = V >> 2;
  constexpr int U = 8;  // covers v4 <= 4096
  float4 buf[U];
  int m = blockIdx.x;
  if (m < M) {
    const float* lr = x + (int64_t)m * V;
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = tid + u * ST; if (i < v4) buf[u] = *(const float4*)(lr + 4 * i); }
  }
  for (; m < M; m += gridDim.x) {
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * ST;
      if (i < v4) {
        float4 l = buf[u];
        if (MODE & 2) { l.x /= T; l.y /= T; l.z /= T; l.w /= T; }
        if (MODE & 1) *(float4*)(row + 4 * i) = l;
        mx = fmaxf(fmaxf(mx, fmaxf(l.x, l.y)), fmaxf(l.z, l.w));
      }
    }
    const int mn = m + gridDim.x;
    if (mn < M) {
      const float* lr = x + (int64_t)mn * V;
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = tid + u * ST; if (i < v4) buf[u] = *(const float4*)(lr + 4 * i); }
    }
    mx = block_max(mx, red, tid);
    if (MODE & 4) {  // stand-in for the later phases: a few passes over the LDS row
      float acc = 0.f;
      for (int p = 0; p < 4; ++p) {
        for (int i = tid; i < V; i += ST) acc += __expf(row[i] - mx);
        __syncthreads();
      }
      if (acc == 12345.f) mx = 0.f;
    }
    if (tid == 0) out[m] = mx;
    __syncthreads();
  }
}

template <int MODE>
__global__ __launch_bounds__(ST) void one_row_with_work(const float* __restrict__ x, int V, float T, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* row = (float*)smem;
  __shared__ float red[ST / 64];
  const int tid = threadIdx.x, m = blockIdx.x;
  const float* lr = x + (int64_t)m * V;
  const int v4 = V >> 2;
  float mx = -INFINITY;
  constexpr int U = 8;
  float4 buf[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const int i = tid + u * ST; if (i < v4) buf[u] = *(const float4*)(lr + 4 * i); }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = tid + u * ST;
    if (i < v4) {
      float4 l = buf[u];
      if (MODE & 2) { l.x /= T; l.y /= T; l.z /= T; l.w /= T; }
      *(float4*)(row + 4 * i) = l;
      mx = fmaxf(fmaxf(mx, fmaxf(l.x, l.y)), fmaxf(l.z, l.w));
    }
  }
  mx = block_max(mx, red, tid);
  float acc = 0.f;
  for (int p = 0; p < 4; ++p) {
    for (int i = tid; i < V; i += ST) acc += __expf(row[i] - mx);
    __syncthreads();
  }
  if (acc == 12345.f) mx = 0.f;
  if (tid == 0) out[m] = mx;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16896;
  const int V = argc > 2 ? atoi(argv[2]) : 15080;
  if (V % 4 || V > 16384) { printf("V must be a multiple of 4, <= 16384\n"); return 1; }
  const size_t n = (size_t)M * V;
  float *x, *out;
  CK(hipMalloc(&x, n * 4));
  CK(hipMalloc(&out, (size_t)M * 4));
  std::vector<float> h(n);
  unsigned s = 1;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.f * 8.f - 4.f; }
  CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice));
  const size_t lds_row = (size_t)V * 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-52s %.3f ms  %.0f GB/s\n", name, ms, n * 4 / 1e9 / ms * 1e3);
    fflush(stdout);
  };
#define ATTR(k, b) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(b)))
  ATTR(one_row_per_block<1>, 160 * 1024 - 64); ATTR(one_row_per_block<3>, 160 * 1024 - 64);
  ATTR(one_row_per_block<0>, 160 * 1024 - 64); ATTR(one_row_per_block<2>, 160 * 1024 - 64);
  ATTR(persistent_prefetch<1>, 160 * 1024 - 64); ATTR(persistent_prefetch<3>, 160 * 1024 - 64);
  ATTR(persistent_prefetch<5>, 160 * 1024 - 64); ATTR(persistent_prefetch<7>, 160 * 1024 - 64);
  ATTR(one_row_with_work<1>, 160 * 1024 - 64); ATTR(one_row_with_work<3>, 160 * 1024 - 64);
  const float T = 0.1f;
  timeit("row/block, no LDS, no div (LDS 0)", [&] { one_row_per_block<0><<<M, ST, 0>>>(x, V, T, out); });
  timeit("row/block, no LDS store, no div (LDS 64K reserved)", [&] { one_row_per_block<0><<<M, ST, 65536>>>(x, V, T, out); });
  timeit("row/block, no LDS store, div (LDS 0)", [&] { one_row_per_block<2><<<M, ST, 0>>>(x, V, T, out); });
  timeit("row/block, LDS store, no div (row only)", [&] { one_row_per_block<1><<<M, ST, lds_row>>>(x, V, T, out); });
  timeit("row/block, LDS store, no div (64K+8K as sampler)", [&] { one_row_per_block<1><<<M, ST, lds_row + 8192>>>(x, V, T, out); });
  timeit("row/block, LDS store, div (64K+8K as sampler)", [&] { one_row_per_block<3><<<M, ST, lds_row + 8192>>>(x, V, T, out); });
  for (int g : {256, 512, 768}) {
    char nm[96];
    snprintf(nm, sizeof nm, "persistent prefetch grid %d, LDS store, no div", g);
    timeit(nm, [&] { persistent_prefetch<1><<<g, ST, lds_row + 8192>>>(x, M, V, T, out); });
    snprintf(nm, sizeof nm, "persistent prefetch grid %d, LDS store, div", g);
    timeit(nm, [&] { persistent_prefetch<3><<<g, ST, lds_row + 8192>>>(x, M, V, T, out); });
  }
  timeit("row/block + 4 LDS passes, no div", [&] { one_row_with_work<1><<<M, ST, lds_row + 8192>>>(x, V, T, out); });
  timeit("row/block + 4 LDS passes, div", [&] { one_row_with_work<3><<<M, ST, lds_row + 8192>>>(x, V, T, out); });
  timeit("persistent 512 + 4 LDS passes, no div", [&] { persistent_prefetch<5><<<512, ST, lds_row + 8192>>>(x, M, V, T, out); });
  timeit("persistent 512 + 4 LDS passes, div", [&] { persistent_prefetch<7><<<512, ST, lds_row + 8192>>>(x, M, V, T, out); });
  return 0;
}

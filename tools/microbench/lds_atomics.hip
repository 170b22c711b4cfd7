// tools/microbench/lds_atomics.hip — throughput of LDS atomic adds on gfx950 (f32 / u32 / u64, random addresses in a 96 KB tile).
// Motivation: the band-scatter form of Z = B*Y measured 7.0 ms with ds_add_f32 vs 1.85 ms with plain stores.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o lds_atomics lds_atomics.hip ; run: ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(512) void k(const uint32_t* __restrict__ idx, int n_per_thread, float* out) {
  extern __shared__ char sm[];
  float* tf = (float*)sm;
  unsigned int* tu = (unsigned int*)sm;
  unsigned long long* tl = (unsigned long long*)sm;
  const int words = MODE == 2 ? 12288 : 24576;  // 96 KB
  for (int i = threadIdx.x; i < 24576; i += blockDim.x) tu[i] = 0;
  __syncthreads();
  const uint32_t* my = idx + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
  uint32_t a = my[0], b = my[1], c = my[2], d = my[3];
  for (int it = 0; it < n_per_thread; ++it) {
    a = a * 1664525u + 1013904223u;
    const uint32_t p = (a >> 8) % words;
    if (MODE == 0) atomicAdd(&tf[p], 1.5f);
    else if (MODE == 1) atomicAdd(&tu[p], 3u);
    else if (MODE == 2) atomicAdd(&tl[p], 3ull);
    else tf[p] = 1.5f;  // plain store
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tf[b % 1024] + tf[c % 1024] + tf[d % 1024];
}

int main() {
  const int blocks = 256, threads = 512, n = 4096;
  uint32_t* idx;
  float* out;
  hipMalloc(&idx, (size_t)blocks * threads * 16);
  hipMalloc(&out, blocks * 4);
  hipMemset(idx, 0x5a, (size_t)blocks * threads * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain ds_write_b32"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 98304, 0, idx, n, out);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 98304, 0, idx, n, out);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 98304, 0, idx, n, out);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 98304, 0, idx, n, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double ops = (double)blocks * threads * n;
      if (rep == 1)
        printf("%-20s %8.3f ms  %7.2f G lane-ops/s chip  = %.3f lane-ops/clk/CU (2.4 GHz, %d CUs busy)\n", names[mode], ms, ops / ms / 1e6,
               ops / ms / 1e6 / 2.4 / 256 * 1e0 / 1.0, blocks);
    }
  }
  return 0;
}

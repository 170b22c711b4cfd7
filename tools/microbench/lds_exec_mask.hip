// tools/microbench/lds_exec_mask.hip — what does an LDS gather cost when part of the wave is masked off?
// The padded slots of the sliced-ELL id streams (gram_lds.hip) read a zero row with every lane active.  If the LDS skipped the
// 16-lane passes of a ds_read_b128 whose lanes are all inactive, masking the padded slots off would pay; if an instruction costs
// the same whatever the exec mask, it would not.  One workgroup of 16 waves per CU (160 KB of LDS claimed), every lane walks
// pseudo-random 48-byte rows (2 x b128 + 1 x b64 = the 10-column panel row) under the mask pattern of the variant.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_exec_mask lds_exec_mask.hip ; run: ./lds_exec_mask
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

constexpr int RB = 3412;
constexpr int LDS_BYTES = RB * 48;

// pattern: 0 = all lanes, 1 = lanes 0..15, 2 = lanes 0..31, 3 = lanes 0..47, 4 = every 4th lane (16 lanes, all four quarters),
//          5 = lanes 0..7 of every quarter (32 lanes), 6 = one lane
__device__ inline bool active(int pattern, int lane) {
  switch (pattern) {
    case 0: return true;
    case 1: return lane < 16;
    case 2: return lane < 32;
    case 3: return lane < 48;
    case 4: return (lane & 3) == 0;
    case 5: return (lane & 15) < 8;
    default: return lane == 0;
  }
}

__global__ __launch_bounds__(1024) void gather_k(float* __restrict__ out, int iters, int pattern, int conflict_free) {
  extern __shared__ float4 xs[];
  for (int i = threadIdx.x; i < RB * 3; i += 1024) xs[i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  // planes as in gram_lds.hip: [RB] float4, [RB] float4, [RB] float2
  const float4* p0 = xs;
  const float4* p1 = xs + RB;
  const float2* p2 = reinterpret_cast<const float2*>(xs + 2 * RB);
  uint32_t idx = conflict_free ? (uint32_t)lane : (uint32_t)((threadIdx.x * 2654435761u) % RB);
  const uint32_t step = conflict_free ? 64u : (uint32_t)(1 + 2 * ((threadIdx.x * 40503u) % 1500));
  float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
  float2 a2 = make_float2(0, 0);
  if (active(pattern, lane)) {
    for (int it = 0; it < iters; ++it) {
      const float4 v0 = p0[idx], v1 = p1[idx];
      const float2 v2 = p2[idx];
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y;
      idx += step;
      idx = idx >= RB ? idx - RB : idx;
    }
  }
  out[(size_t)blockIdx.x * 1024 + threadIdx.x] = a0.x + a0.y + a0.z + a0.w + a1.x + a1.y + a1.z + a1.w + a2.x + a2.y;
}

int main() {
  const int nwg = 256, iters = 20000;
  float* out;
  CK(hipMalloc(&out, (size_t)nwg * 1024 * 4));
  CK(hipFuncSetAttribute((const void*)gather_k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[] = {"all 64 lanes", "lanes 0..15", "lanes 0..31", "lanes 0..47", "every 4th lane (16)", "8 of every 16 (32)", "one lane"};
  for (int cf = 0; cf < 2; ++cf)
    for (int pattern = 0; pattern < 7; ++pattern) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gather_k, dim3(nwg), dim3(1024), LDS_BYTES, 0, out, iters, pattern, cf);
        CK(hipGetLastError());
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      // one workgroup per CU: clocks per wave-level row gather = ms * 2.4e6 / (iters * 16 waves)
      printf("%-14s %-22s %8.3f ms   %.1f clk per wave-row (48 B x 64 lanes) at 2.4 GHz\n", cf ? "conflict-free" : "random rows", names[pattern], best,
             best * 2.4e6 / ((double)iters * 16));
    }
  return 0;
}

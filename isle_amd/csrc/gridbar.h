// isle_amd/csrc/gridbar.h — grid-wide barrier for persistent kernels whose workgroups are all resident (grid <= number of CUs,
// one workgroup per CU at most), and the device-scope accesses for the data that crosses workgroups between barriers.
#pragma once
#include <hip/hip_runtime.h>

// sc1 accesses: coherent in memory across the XCDs' L2s without a cache write-back / invalidate
__device__ inline double gb_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gb_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Arrive and wait until `target` arrivals have been counted.  The counter only grows; a launch gets its starting count from
// the host.  Returns false (and raises *abort) after a bounded spin — a workgroup that is not resident would otherwise hang
// the GPU — or when another workgroup has given up.  Every thread of the workgroup must call it.
__device__ inline bool gb_barrier(unsigned int* ctr, unsigned int target, unsigned int* abort) {
  __shared__ unsigned int gb_ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's stores have been issued and acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0, good = 1;
    // (int) difference: the counter may wrap after 4 G arrivals
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 0x3ffu) == 0 && (spins > (1u << 21) || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    gb_ok = good;
  }
  __syncthreads();
  return gb_ok != 0;
}

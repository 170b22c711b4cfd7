// isle_amd/csrc/gridbar.h — grid-wide barriers for persistent kernels whose workgroups are all resident (grid <= number of CUs,
// one workgroup per CU at most), and the device-scope accesses for the data that crosses workgroups between barriers.
//
// Three forms, same contract (every thread of every workgroup calls it; false = gave up, *abort raised); td_persist_k (one barrier per column,
// thousands per launch) takes gbs_barrier, the others under ISLE_TD_BAR = flat | hier.  EVD at n = 2000 / n = 400, round 6:
// gbs 36.4 / 3.50 ms, gbh 37.0 / 3.93, gb 40.7 / 3.69 (profiles/r06_j_evd_probe_three_barriers.log).
//   gb_barrier   one monotonic counter: release once, poll relaxed, acquire once (until round 6 it polled with ACQUIRE loads: an invalidate per poll).
//   gbh_barrier  hierarchical (MI355X_MICROARCH.md, price table row "barrier-xcd": 4.1 us at 256 workgroups against 7.4 for one counter
//                polled with relaxed loads and 13.2 polled with acquire loads — what gb_barrier did until round 6): the workgroups form eight
//                groups by blockIdx % 8 — the dispatcher deals workgroups round-robin over the eight XCDs, so a group is one XCD's
//                workgroups — each with an arrival counter and a generation word on 128-byte lines of their own; the last arriver of a
//                group adds to the top counter, the last of those publishes the top generation, every group's last arriver then publishes
//                its group's generation.  At most 32 pollers per line instead of 256.  Correctness does not rest on the round-robin
//                placement: EVERY workgroup's lane 0 makes its own agent-scope release fence before it arrives and its own agent-scope
//                acquire fence after the release (a group that straddles XCDs only polls a line that is further away).
//   gbs_barrier  sharded (at the end of this file): the eight arrival counters without the two extra hops.
// All counters only grow (signed differences: they may wrap); a launch gets its starting counts from the host (GbHierArgs), or starts
// from a zeroed state block with all bases 0.
#pragma once
#include <hip/hip_runtime.h>

// sc1 accesses: coherent in memory across the XCDs' L2s without a cache write-back / invalidate
__device__ inline double gb_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gb_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr unsigned int GB_SPIN_LIMIT = 1u << 21;  // polls (each behind an s_sleep) before a barrier gives up: ~0.2 s

// Poll *p (relaxed, agent scope) until it has reached `target` (signed difference).  False after the spin limit or when *abort is set.
__device__ inline bool gb_spin_until(unsigned int* p, unsigned int target, unsigned int* abort) {
  unsigned int spins = 0;
  while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 0x3ffu) == 0 && (spins > GB_SPIN_LIMIT || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
      __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}

// Arrive and wait until `target` arrivals have been counted.  The counter only grows; a launch gets its starting count from
// the host.  Returns false (and raises *abort) after a bounded spin — a workgroup that is not resident would otherwise hang
// the GPU — or when another workgroup has given up.  Every thread of the workgroup must call it.
__device__ inline bool gb_barrier(unsigned int* ctr, unsigned int target, unsigned int* abort) {
  __shared__ unsigned int gb_ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // this wave's stores have been issued and acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    // release once, poll relaxed, acquire once (an acquire load per poll invalidates the caches per poll)
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const bool good = gb_spin_until(ctr, target, abort);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    gb_ok = good ? 1u : 0u;
  }
  __syncthreads();
  return gb_ok != 0;
}

// ---- hierarchical form ----------------------------------------------------------------------------------------------------------
constexpr int GBH_GROUPS = 8;
constexpr int GBH_LINE = 32;  // unsigned ints per 128-byte line
// state block (unsigned ints, 128-byte aligned): line g = arrival counter of group g, line 8 = top counter, line 9 = top generation,
// line 10 + g = generation of group g
constexpr int GBH_STATE_WORDS = (2 * GBH_GROUPS + 2) * GBH_LINE;
struct GbHierArgs {
  unsigned int* st;                    // GBH_STATE_WORDS words
  unsigned int cnt_base[GBH_GROUPS];   // arrivals counted on each group's counter before this launch
  unsigned int top_base;               // arrivals counted on the top counter before this launch
  unsigned int gen_base;               // barriers crossed on this state block before this launch
};
__host__ __device__ inline unsigned int gbh_group_size(unsigned int G, unsigned int g) { return g < G ? (G - g + GBH_GROUPS - 1) / GBH_GROUPS : 0u; }

// Barrier number j (1, 2, ... within this launch) of a grid of G workgroups.
__device__ inline bool gbh_barrier(const GbHierArgs& a, unsigned int j, unsigned int* abort) {
  __shared__ unsigned int gbh_ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // every wave's stores have been issued and acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int G = gridDim.x, g = blockIdx.x % GBH_GROUPS;
    const unsigned int ng = gbh_group_size(G, g), ngroups = G < (unsigned)GBH_GROUPS ? G : (unsigned)GBH_GROUPS;
    const unsigned int gen = a.gen_base + j;
    unsigned int* cnt = a.st + g * GBH_LINE;
    unsigned int* top = a.st + GBH_GROUPS * GBH_LINE;
    unsigned int* topgen = top + GBH_LINE;
    unsigned int* mygen = a.st + (GBH_GROUPS + 2 + g) * GBH_LINE;
    bool good = true;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    // (every producer has pushed its own stores to memory before it arrives and every consumer invalidates after it is released, so the
    // counters and generation words themselves need no ordering beyond program order: the asm statements keep the compiler from moving them)
    const unsigned int seen = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    asm volatile("" ::: "memory");
    if (seen == a.cnt_base[g] + j * ng) {  // the group's last arriver
      const unsigned int tseen = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      asm volatile("" ::: "memory");
      if (tseen == a.top_base + j * ngroups) __hip_atomic_store(topgen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else good = gb_spin_until(topgen, gen, abort);
      asm volatile("" ::: "memory");
      // after a time-out the generation stays unpublished: the group's pollers leave through *abort, not as if the barrier had completed
      if (good) __hip_atomic_store(mygen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      good = gb_spin_until(mygen, gen, abort);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    gbh_ok = good ? 1u : 0u;
  }
  __syncthreads();
  return gbh_ok != 0;
}

// ---- sharded form (the default of td_persist_k since round 6) ------------------------------------------------------------------------
// The arrival counters of the hierarchical form without its two extra hops: a workgroup adds to its group's counter (at most 32 adders per
// line) and then polls ALL eight counters at once — lane g of wave 0 polls counter g — until every group has reached its target.  Two round
// trips (add, poll) instead of four (add, top add, top generation, group generation), at the price of 256 pollers per line again (loads, not
// atomics).  Same state block (only the first eight lines are used), same fences, same contract.
__device__ inline bool gbs_barrier(const GbHierArgs& a, unsigned int j, unsigned int* abort) {
  __shared__ unsigned int gbs_ok;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x < 64) {  // wave 0
    const unsigned int G = gridDim.x, lane = threadIdx.x;
    const unsigned int mine = blockIdx.x % GBH_GROUPS;
    if (lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_fetch_add(a.st + mine * GBH_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("" ::: "memory");
    const unsigned int g = lane < (unsigned)GBH_GROUPS ? lane : 0u;
    const unsigned int target = a.cnt_base[g] + j * gbh_group_size(G, g);
    unsigned int* cnt = a.st + g * GBH_LINE;
    unsigned int spins = 0;
    bool good = true;
    for (;;) {
      const bool there = (int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0;
      if (__all(there)) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 0x3ffu) == 0 && (spins > GB_SPIN_LIMIT || __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (lane == 0) __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = false;
        break;
      }
    }
    asm volatile("" ::: "memory");
    if (lane == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      gbs_ok = good ? 1u : 0u;
    }
  }
  __syncthreads();
  return gbs_ok != 0;
}

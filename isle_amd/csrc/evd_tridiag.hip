// isle_amd/csrc/evd_tridiag.hip — small symmetric eigensolver by tridiagonalisation, for arma::eig_sym (-> ssyevd) in
// BlockKs::truncate (block-ks/restarted_block_ks.h:138-187).
//
// The block Jacobi solver of dense.hip needs 11-13 sweeps on Ritz matrices (clustered spectrum: eight or nine sweeps in which
// every column pair still rotates), each sweep a chain of n/16 dependent rounds.  Here, in fp64 throughout:
//   1. Householder tridiagonalisation S = Q T Q^T, one column per step.  n <= 512: td_persist_k, one launch with the matrix
//      resident in the LDS of 32 workgroups and two grid barriers per column (see there).  Larger n — two launches per column: td_step_k (one workgroup:
//      finish p = tau A v from the partial products, w = p - tau/2 (p.v) v, then the next column of the updated matrix and its
//      reflector) and td_update_symv_k (all workgroups: A -= v w^T + w v^T fused with the partial products A v_next of the
//      next step, per column block, summed later in fixed order — no atomics, bitwise reproducible).
//   2. td_bisect_k: eigenvalue i by bisection on the Sturm count, one thread per eigenvalue (ordered for free).
//   3. td_vectors_k: eigenvector of T by twisted factorisation (the core of dlar1v / MRRR: forward L D L^T, backward U D U^T,
//      twist at the smallest |gamma|, one substitution sweep each way), one thread per eigenvector, only the leading nvec.
//   4. td_back_k: Z = Q Z_T, eight eigenvector columns per workgroup held in LDS, reflectors applied in reverse order.
//   5. td_check_k: every vector against its four neighbours in eigenvalue order and its own norm; if orthogonality is worse than
//      1e-6 (clusters tighter than ~1e-9 relative, exact multiplicities) the caller falls back to the Jacobi solver.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "gridbar.h"

namespace {

constexpr int TD_ROWS = 256;       // rows per workgroup of td_update_symv_k
constexpr int TD_NMAX_BACK = 2048; // td_back_k keeps n x 8 doubles in LDS (128 KB)
constexpr int TD_NRB = TD_NMAX_BACK / TD_ROWS;  // row blocks of td_update_symv_k at the largest n (stride of its row-block tickets)

__device__ inline double td_block_sum(double v, double* sh /* >= 16 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = (int)(blockDim.x >> 6);
  for (int w = 0; w < nw; ++w) s += sh[w];  // fixed order
  return s;
}

// One workgroup.  first: set up column 0 only.  Otherwise, for step j:
//   p = tau_j * sum_cb part[cb][:]  (rows j+1..n-1),  w = p - tau_j/2 (p.v_j) v_j        (v_j = A[j+1:, j], v_j[j+1] = 1)
//   column cj = j+1 of A - v w^T - w v^T  ->  d[cj], reflector v_cj into A[cj+1:, cj], e[cj], tau[cj]
template <int TD_MAXPT>  // values per thread: 256 threads x 4 for n <= 1024, x 16 for n <= 4096
__device__ inline void td_step_dev(double* __restrict__ A, int n, int j, int first, int ncb, const double* __restrict__ part,
                                   double* __restrict__ w, double* __restrict__ d, double* __restrict__ e, double* __restrict__ tau) {
  constexpr int TD_T = TD_ROWS;
  __shared__ double sh[16];
  __shared__ double bc[2];
  __shared__ double bw;
  const int t = threadIdx.x;
  double wv[TD_MAXPT], vv[TD_MAXPT];
  double w_cj = 0.0;
  const int cj = first ? 0 : j + 1;
  // the next column is needed only after two block-wide sums: fetch it now, its latency hides behind the partial sums
  double a0[TD_MAXPT];
#pragma unroll
  for (int s = 0; s < TD_MAXPT; ++s) {
    const int i = cj + t + s * TD_T;
    a0[s] = i < n ? A[(size_t)cj * n + i] : 0.0;
  }
  if (!first) {
    const double tj = tau[j];
    const int r0 = j + 1;
    double dot = 0.0;
#pragma unroll
    for (int s = 0; s < TD_MAXPT; ++s) {
      const int i = r0 + t + s * TD_T;
      wv[s] = 0.0;
      vv[s] = 0.0;
      if (i < n) {
        double p = 0.0;
        for (int cb = 0; cb < ncb; ++cb) p += part[(size_t)cb * n + i];  // fixed order
        p *= tj;
        const double v = A[(size_t)j * n + i];
        wv[s] = p;
        vv[s] = v;
        dot = fma(p, v, dot);
      }
    }
    dot = td_block_sum(dot, sh);
    const double a2 = -0.5 * tj * dot;
#pragma unroll
    for (int s = 0; s < TD_MAXPT; ++s) {
      const int i = r0 + t + s * TD_T;
      if (i < n) {
        wv[s] = fma(a2, vv[s], wv[s]);
        w[i] = wv[s];
        if (i == cj) bw = wv[s];
      }
    }
    __syncthreads();
    w_cj = bw;  // w at row j+1 (v_j there is 1)
  }
  // column cj of the updated matrix, rows cj..n-1 (thread-local rows: i = cj + t + s*TD_T; note cj = r0 when !first)
  double cv[TD_MAXPT];
  double nrm2 = 0.0;
#pragma unroll
  for (int s = 0; s < TD_MAXPT; ++s) {
    const int i = cj + t + s * TD_T;
    cv[s] = 0.0;
    if (i < n) {
      double a = a0[s];
      if (!first) a -= vv[s] * w_cj + wv[s];  // v_j[cj] = 1
      cv[s] = a;
      if (i == cj) bc[0] = a;
      if (i == cj + 1) bc[1] = a;
      if (i > cj + 1) nrm2 = fma(a, a, nrm2);
    }
  }
  nrm2 = td_block_sum(nrm2, sh);
  __syncthreads();
  if (t == 0) d[cj] = bc[0];
  if (cj == n - 1) return;
  const double alpha = bc[1];
  double beta = alpha, tv = 0.0, scale = 0.0;
  if (nrm2 > 0.0) {
    beta = -copysign(sqrt(fma(alpha, alpha, nrm2)), alpha);
    tv = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  if (t == 0) {
    e[cj] = beta;
    tau[cj] = tv;
  }
#pragma unroll
  for (int s = 0; s < TD_MAXPT; ++s) {
    const int i = cj + t + s * TD_T;
    if (i < n && i > cj) A[(size_t)cj * n + i] = (i == cj + 1) ? 1.0 : cv[s] * scale;
  }
}

template <int PT>
__global__ __launch_bounds__(TD_ROWS) void td_first_k(double* __restrict__ A, int n, double* __restrict__ w, double* __restrict__ d,
                                                       double* __restrict__ e, double* __restrict__ tau) {
  td_step_dev<PT>(A, n, -1, 1, 0, nullptr, w, d, e, tau);
}

// All workgroups: the block B = rows/cols r0..n-1 of A, r0 = j + 2 (j = -1: no update, r0 = 1).
//   update:  B -= v_j w_j^T + w_j v_j^T     (v_j = A[:, j], w_j = w)
//   part[cb][i] = sum over this column block of B[i][k] * vn[k]     (vn = v_{j+1} = A[:, j+1])
// grid = (row blocks of TD_ROWS, column blocks of CB)
// The workgroups that finish last (ticket counters, see the end of the kernel) reduce the partial products and run step j + 1
// in place: one launch per column.
template <int PT>
__global__ __launch_bounds__(TD_ROWS) void td_update_symv_k(double* __restrict__ A, int n, int j, int update, int CB, double* __restrict__ w,
                                                             double* __restrict__ part, double* __restrict__ psum, double* __restrict__ d,
                                                             double* __restrict__ e, double* __restrict__ tau, unsigned int* __restrict__ tickets,
                                                             unsigned int* __restrict__ tickets_rb) {
  extern __shared__ double cs[];  // 3 x CB: v_j[k], w[k], vn[k]
  __shared__ unsigned int ticket;
  const int r0 = j + 2;
  const int k0 = r0 + blockIdx.y * CB;
  const int k1 = min(n, k0 + CB);
  const int i = r0 + blockIdx.x * TD_ROWS + threadIdx.x;
  for (int kk = threadIdx.x; kk < k1 - k0; kk += TD_ROWS) {
    const int k = k0 + kk;
    cs[kk] = update ? A[(size_t)j * n + k] : 0.0;
    cs[CB + kk] = update ? w[k] : 0.0;
    cs[2 * CB + kk] = A[(size_t)(j + 1) * n + k];
  }
  __syncthreads();
  if (i < n) {
    const double vi = update ? A[(size_t)j * n + i] : 0.0;
    const double wi = update ? w[i] : 0.0;
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) {
      double a = A[(size_t)k * n + i];
      if (update) {
        a -= vi * cs[CB + (k - k0)] + wi * cs[k - k0];
        A[(size_t)k * n + i] = a;
      }
      acc = fma(a, cs[2 * CB + (k - k0)], acc);
    }
    part[(size_t)blockIdx.y * n + i] = acc;
  }
  // Two ticket levels (release / acquire fences at agent scope around each).  The last column block of a ROW block sums that row
  // block's partial products in column-block order into psum — in parallel over the row blocks, instead of one workgroup
  // walking all n x ncb partials; the last row block to finish that then runs the step on the single summed vector.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0)
    ticket = __hip_atomic_fetch_add(&tickets_rb[(size_t)(j + 1) * TD_NRB + blockIdx.x], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != gridDim.y - 1) return;  // uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // this CU's L1 may hold lines other workgroups have rewritten
  if (i < n) {
    double s = 0.0;
    for (unsigned int cb = 0; cb < gridDim.y; ++cb) s += part[(size_t)cb * n + i];  // fixed order
    psum[i] = s;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&tickets[j + 1], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != gridDim.x - 1) return;  // uniform
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  td_step_dev<PT>(A, n, j + 1, 0, 1, psum, w, d, e, tau);
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent form of the tridiagonalisation: ONE launch, G workgroups (32 for n <= 512, up to one per CU for n <= 2048), the
// matrix resident in LDS.  Workgroup g owns the columns c = g (mod G) of the (full, symmetric) trailing matrix.  Every workgroup holds the current
// reflector v (computed redundantly: same data, same order, same bits).  Per column cj:
//   everybody:    p_c = tau * <column c, v> for the owned columns c > cj; the owner of column cj + 1 publishes it  -> grid barrier
//   everybody:    w = p - tau/2 (p.v) v, column cj + 1 of the updated matrix and from it the NEXT reflector (all redundantly),
//                 owned columns -= v w_c + w v_c;  the owner of cj + 1 writes d, e, tau and the reflector for td_back_k
// so a column costs ONE grid barrier instead of a launch (14 us of kernel plus 4-5 us of dependent-launch gap in the chain of
// td_update_symv_k).  What crosses workgroups (p, the published column; double-buffered by the parity of cj) goes through
// device-scope atomics (sc1 accesses: coherent in memory), ordered by the release / acquire of the barrier counter.  A barrier
// gives up after a bounded spin and raises `abort` (a workgroup that is not resident would otherwise hang the GPU); the host
// then runs the launch chain instead.
// ---------------------------------------------------------------------------------------------------------------
// Block sum with ONE barrier: the waves' partial sums go to 16 slots that nobody writes again before another barrier has been crossed
// (td_persist_k has a set per sum), so the barrier in front of the write that td_block_sum needs can go.  Same order of additions.
__device__ inline double td_block_sum1(double v, double* sh16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if (lane == 0) sh16[wave] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = (int)(blockDim.x >> 6);
  for (int w = 0; w < nw; ++w) s += sh16[w];  // fixed order
  return s;
}

#ifdef TD_STAMPS
__device__ unsigned long long td_stamps[16];
#define TD_ST(k) if (g == 0 && t == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); td_stamps[k] += now_ - st_; st_ = now_; }
#else
#define TD_ST(k)
#endif
constexpr int TD_P_GSMALL = 32;   // workgroups for n <= 512 (more only add barrier latency)
constexpr int TD_P_T = 1024;  // 16 waves: the length-n vector phases every workgroup repeats per column are latency chains (256 threads: 35 -> 28 ms at n = 2010)
constexpr int TD_P_MAXPT = TD_NMAX_BACK / TD_P_T;  // column entries per thread
constexpr size_t TD_P_LDS = 160 * 1024 - 512;  // dynamic part: the kernel also has a few static words

__device__ inline double td_ld(const double* p) { return gb_ld(p); }
__device__ inline void td_st(double* p, double v) { gb_st(p, v); }

// Reflector of a column held in LDS (cn[r0 - 1 .. n - 1], cn[r0 - 1] the diagonal entry): the formulas of td_step_dev.  Every
// workgroup runs this on the same data in the same order, so v and tau are bit-identical everywhere without a broadcast.
// Leaves v in vs[r0 .. n - 1]; returns tau; the owner also writes d, e, tau (the reflector itself goes to A behind the next barrier:
// column 0 of A is still being read by the other workgroups when the first reflector is ready).
__device__ inline double td_p_reflector(const double* cn, double* vs, int n, int cj, bool owner, double* __restrict__ d,
                                        double* __restrict__ e, double* __restrict__ tau, double* sh) {
  const int t = threadIdx.x, r0 = cj + 1;
  double nrm2 = 0.0;
  for (int i = r0 + 1 + t; i < n; i += TD_P_T) nrm2 = fma(cn[i], cn[i], nrm2);
  nrm2 = td_block_sum(nrm2, sh);
  const double alpha = cn[r0];
  double beta = alpha, tv = 0.0, scale = 0.0;
  if (nrm2 > 0.0) {
    beta = -copysign(sqrt(fma(alpha, alpha, nrm2)), alpha);
    tv = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  if (owner && t == 0) {
    d[cj] = cn[cj];
    e[cj] = beta;
    tau[cj] = tv;
  }
  for (int i = r0 + t; i < n; i += TD_P_T) vs[i] = (i == r0) ? 1.0 : cn[i] * scale;
  __syncthreads();
  return tv;
}

// The grid barrier (gridbar.h; state block bar.st, zeroed by the host before the launch): sharded counters polled together by default,
// ISLE_TD_BAR=hier the hierarchical form, ISLE_TD_BAR=flat the one-counter form on bar.st[0] (kept to time one against the other)
template <int BAR>  // 0: one counter (gb_barrier), 1: hierarchical (gbh_barrier), 2: sharded counters polled together (gbs_barrier)
__global__ __launch_bounds__(TD_P_T) void td_persist_k(double* __restrict__ A, int n, double* __restrict__ d, double* __restrict__ e,
                                                        double* __restrict__ tau, double* __restrict__ xbuf /* 2 x (p | next column), 4 n */,
                                                        const GbHierArgs bar, unsigned int* __restrict__ abort) {
  extern __shared__ double lds[];  // slab: ncl columns of n | vs n | ws n
  __shared__ double sh[16], shd[16], shn[18], shr[4];  // sh: the first reflector; shd / shn: the two sums of a column (shn[16]: alpha); shr: beta, tau, scale
  const int G = (int)gridDim.x;
  const int g = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ncl = (n + G - 1) / G;
  double* slab = lds;
  double* vs = slab + (size_t)ncl * n;
  double* ws = vs + n;  // w; afterwards the next column (the input of the next reflector)
  for (int lc = 0; lc < ncl; ++lc) {
    const int c = lc * G + g;
    if (c < n)
      for (int i = t; i < n; i += TD_P_T) slab[(size_t)lc * n + i] = A[(size_t)c * n + i];
  }
  // column 0 is read by everybody from the input itself (nothing has been written yet)
  for (int i = t; i < n; i += TD_P_T) ws[i] = A[i];
  __syncthreads();
  double tj = td_p_reflector(ws, vs, n, 0, g == 0, d, e, tau, sh);
  unsigned int phase = 0;
#ifdef TD_STAMPS
  unsigned long long st_ = __builtin_readcyclecounter();
#endif
  for (int cj = 0; cj < n - 1; ++cj) {
    const int r0 = cj + 1;  // first row / column of the trailing block
    double* pbuf = xbuf + (size_t)(cj & 1) * 2 * n;  // double-buffered: a fast workgroup writes step cj + 1 while a slow one still reads step cj
    double* cbuf = pbuf + n;
    // ---- p_c = tau <column c, v> for the owned columns of the trailing block (one wave per column); the owner of column r0 also
    //      publishes that column as it is BEFORE this step's update
    for (int lc = wave; lc < ncl; lc += TD_P_T / 64) {
      const int c = lc * G + g;
      if (c >= r0 && c < n) {  // wave-uniform
        const double* col = slab + (size_t)lc * n;
        double s = 0.0;
#pragma unroll 4
        for (int i = r0 + lane; i < n; i += 64) s = fma(col[i], vs[i], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) td_st(pbuf + c, tj * s);
      }
    }
    if (r0 % G == g) {
      const double* col = slab + (size_t)(r0 / G) * n;
      for (int i = r0 + t; i < n; i += TD_P_T) td_st(cbuf + i, col[i]);
    }
    TD_ST(0)  // symv + publish
    ++phase;
    if (!(BAR == 1 ? gbh_barrier(bar, phase, abort) : BAR == 2 ? gbs_barrier(bar, phase, abort) : gb_barrier(bar.st, phase * (unsigned int)G, abort))) return;
    TD_ST(1)  // grid barrier
    if (cj % G == g)  // the reflector where td_back_k (a later launch) reads it
      for (int i = r0 + t; i < n; i += TD_P_T) A[(size_t)cj * n + i] = vs[i];
    // the next column as published (read below, behind two block-wide sums): asked for now, its round trip to memory runs beside that of p
    double cb[TD_P_MAXPT];
#pragma unroll
    for (int q = 0; q < TD_P_MAXPT; ++q) {
      const int i = r0 + t + q * TD_P_T;
      cb[q] = i < n ? td_ld(cbuf + i) : 0.0;
    }
    // ---- w = p - tau/2 (p.v) v  (redundantly, same order everywhere)
    double dot = 0.0;
    for (int i = r0 + t; i < n; i += TD_P_T) {
      const double p = td_ld(pbuf + i);
      ws[i] = p;
      dot = fma(p, vs[i], dot);
    }
    TD_ST(2)  // loads of p, cb + dot partial
    dot = td_block_sum1(dot, shd);
    TD_ST(3)  // dot block sum
    const double a2 = -0.5 * tj * dot;
    for (int i = r0 + t; i < n; i += TD_P_T) ws[i] = fma(a2, vs[i], ws[i]);  // (the entries this thread wrote above: no barrier in between)
    __syncthreads();
    TD_ST(4)  // w + barrier
    // ---- column r0 of the updated matrix, by everybody, in registers: the input of the next reflector
    double cnr[TD_P_MAXPT];
    {
      const double vr = vs[r0], wr = ws[r0];  // v[r0] = 1
#pragma unroll
      for (int q = 0; q < TD_P_MAXPT; ++q) {
        const int i = r0 + t + q * TD_P_T;
        cnr[q] = i < n ? cb[q] - (vs[i] * wr + ws[i] * vr) : 0.0;
      }
    }
    // ---- rank-2 update of the owned columns
    for (int lc = wave; lc < ncl; lc += TD_P_T / 64) {
      const int c = lc * G + g;
      if (c >= r0 && c < n) {
        double* col = slab + (size_t)lc * n;
        const double vc = vs[c], wc = ws[c];
#pragma unroll 4
        for (int i = r0 + lane; i < n; i += 64) col[i] -= vs[i] * wc + ws[i] * vc;
      }
    }
    TD_ST(5)  // cnr + rank-2 update
    // ---- the next reflector straight from the registers (round 6: the column used to go through LDS behind two barriers, and the sum of its
    // squares had two barriers of its own: eleven workgroup barriers of 16 waves per column, now six).  The formulas of td_p_reflector; thread 0
    // holds the diagonal entry (row r0), thread 1 alpha (row r0 + 1); the sum runs over the rows from r0 + 2.
    if (r0 < n - 1) {
      double nrm2 = 0.0;
#pragma unroll
      for (int q = 0; q < TD_P_MAXPT; ++q) {
        const int i = r0 + t + q * TD_P_T;
        if (i >= r0 + 2 && i < n) nrm2 = fma(cnr[q], cnr[q], nrm2);
      }
      if (t == 1) shn[16] = cnr[0];
      nrm2 = td_block_sum1(nrm2, shn);
      TD_ST(7)  // its barrier also stands between everybody's reads of v, w above and the new v below
      // the square root and the two divisions by ONE wave: sixteen waves doing them side by side share four SIMDs, and the fp64 sequences
      // of the other three waves of a SIMD were 1.5 us of a column's 10.6 (cycle stamps, round 6)
      if (wave == 0) {
        const double alpha = shn[16];
        double beta = alpha, tv = 0.0, scale = 0.0;
        if (nrm2 > 0.0) {
          beta = -copysign(sqrt(fma(alpha, alpha, nrm2)), alpha);
          tv = (beta - alpha) / beta;
          scale = 1.0 / (alpha - beta);
        }
        if (lane == 0) {
          shr[0] = beta;
          shr[1] = tv;
          shr[2] = scale;
        }
      }
      __syncthreads();
      TD_ST(8)
      const double beta = shr[0], tv = shr[1], scale = shr[2];
      if (r0 % G == g && t == 0) {
        d[r0] = cnr[0];
        e[r0] = beta;
        tau[r0] = tv;
      }
#pragma unroll
      for (int q = 0; q < TD_P_MAXPT; ++q) {
        const int i = r0 + t + q * TD_P_T;
        if (i >= r0 + 1 && i < n) vs[i] = (i == r0 + 1) ? 1.0 : cnr[q] * scale;
      }
      __syncthreads();
      tj = tv;
    } else if (r0 % G == g && t == 0) {
      d[n - 1] = cnr[0];
    }
    TD_ST(6)  // reflector
  }
}

// Eigenvalue number idx (ascending) of the tridiagonal (d, e) by multisection on the Sturm count: one wave per eigenvalue, the
// 64 lanes count at 64 interior points of the current interval, which shrinks 65-fold per pass (ten passes instead of
// fifty-odd dependent bisection steps of n divisions each).  Output descending.
// only the nvec largest eigenvalues (lam_desc[0 .. nvec)) are located; the rest of lam_desc is set to 0
__global__ __launch_bounds__(256) void td_bisect_k(const double* __restrict__ d, const double* __restrict__ e, int n, int nvec,
                                                    double* __restrict__ lam_desc) {
  extern __shared__ double sm[];  // d[n], e2[n]
  double* sd = sm;
  double* se2 = sm + n;
  double lo = 1e300, hi = -1e300, emax = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const double di = d[i];
    const double el = i > 0 ? fabs(e[i - 1]) : 0.0, er = i < n - 1 ? fabs(e[i]) : 0.0;
    sd[i] = di;
    se2[i] = i < n - 1 ? e[i] * e[i] : 0.0;
    lo = fmin(lo, di - el - er);
    hi = fmax(hi, di + el + er);
    emax = fmax(emax, er * er);
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_xor(lo, off));
    hi = fmax(hi, __shfl_xor(hi, off));
    emax = fmax(emax, __shfl_xor(emax, off));
  }
  __shared__ double s_lo[4], s_hi[4], s_em[4];
  if ((threadIdx.x & 63) == 0) {
    s_lo[threadIdx.x >> 6] = lo;
    s_hi[threadIdx.x >> 6] = hi;
    s_em[threadIdx.x >> 6] = emax;
  }
  __syncthreads();
  lo = fmin(fmin(s_lo[0], s_lo[1]), fmin(s_lo[2], s_lo[3]));
  hi = fmax(fmax(s_hi[0], s_hi[1]), fmax(s_hi[2], s_hi[3]));
  emax = fmax(fmax(s_em[0], s_em[1]), fmax(s_em[2], s_em[3]));
  const double span = fmax(fabs(lo), fabs(hi));
  lo -= 2.0 * 2.3e-16 * span * n + 1e-300;
  hi += 2.0 * 2.3e-16 * span * n + 1e-300;
  const double pivmin = fmax(2.3e-308 * fmax(1.0, emax), 1e-300);
  const int lane = threadIdx.x & 63;
  for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < n - nvec; idx += gridDim.x * 4)
    if (lane == 0) lam_desc[n - 1 - idx] = 0.0;
  for (int idx = n - nvec + blockIdx.x * 4 + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4) {  // wave-uniform; idx-th smallest
    double a = lo, b = hi;  // invariant: count(a) <= idx < count(b)
    for (int it = 0; it < 16; ++it) {
      const double h = (b - a) * (1.0 / 65.0);
      const double x = a + h * (double)(lane + 1);
      int cnt = 0;
      double q = sd[0] - x;
      if (fabs(q) < pivmin) q = -pivmin;
      cnt += q < 0.0;
      for (int k = 1; k < n; ++k) {
        q = (sd[k] - x) - se2[k - 1] / q;
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += q < 0.0;
      }
      // first lane whose point has more than idx eigenvalues below it (counts are monotone in x up to rounding)
      const unsigned long long above = __ballot(cnt > idx);
      const int f = above ? __ffsll((long long)above) - 1 : 64;
      const double na = f == 0 ? a : a + h * (double)f;          // point of lane f - 1
      const double nb = f == 64 ? b : a + h * (double)(f + 1);   // point of lane f
      const bool stop = !(nb - na < b - a) || (nb - na) <= 4.0 * 2.3e-16 * fmax(fabs(na), fabs(nb));
      a = na;
      b = nb;
      if (stop) break;
    }
    if (lane == 0) lam_desc[n - 1 - idx] = 0.5 * (a + b);
  }
}

// Eigenvector of T for lam_desc[v], v < nvec: twisted factorisation.  One thread per vector; per-thread work arrays
// interleaved over vectors ([i * nvec + v]: coalesced).  Z row-major n x nvec, unit 2-norm.
__global__ __launch_bounds__(64) void td_vectors_k(const double* __restrict__ d, const double* __restrict__ e, int n, const double* __restrict__ lam_desc,
                                                    int nvec, double* __restrict__ Dp /*n x nvec*/, double* __restrict__ Lf /*n x nvec*/,
                                                    double* __restrict__ Z /*n x nvec; also holds U factors during the backward sweep*/,
                                                    int v0, int v1 /*the vectors [v0, v1) of the nvec (several ranks: each takes a range)*/) {
  // One lane per eigenvector; every sweep is a recurrence over the rows.  The operands of a step (d, e, and what an earlier sweep parked
  // in Dp / Lf / Z) do not depend on the recurrence, so they are fetched U rows ahead: fetched inside the step, each row paid a
  // memory latency on top of its division (2.8 ms per call at n = 2010, one wave per CU on 16 CUs).
  constexpr int U = 8;
  const int v = min(v0 + (int)blockIdx.x * 64 + (int)threadIdx.x, v1 - 1);  // lanes beyond the last vector repeat it (the wave max below needs them)
  const bool live = v0 + (int)blockIdx.x * 64 + (int)threadIdx.x < v1;
  const double lam = lam_desc[v];
  double tnorm = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) tnorm = fmax(tnorm, fabs(d[i]) + (i < n - 1 ? fabs(e[i]) : 0.0) + (i > 0 ? fabs(e[i - 1]) : 0.0));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tnorm = fmax(tnorm, __shfl_xor(tnorm, off));
  if (!live) return;
  const double piv = fmax(2.3e-16 * tnorm, 1e-300);
  auto IDX = [&](int i) { return (size_t)i * nvec + v; };
  // forward: T - lam I = L D L^T
  double D = d[0] - lam;
  if (fabs(D) < piv) D = -piv;
  Dp[IDX(0)] = D;
  for (int i0 = 0; i0 < n - 1; i0 += U) {
    double ev[U], dv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u, n - 2);
      ev[u] = e[i];
      dv[u] = d[i + 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u;
      if (i < n - 1) {
        const double l = ev[u] / D;
        Lf[IDX(i)] = l;
        D = (dv[u] - lam) - l * ev[u];
        if (fabs(D) < piv) D = -piv;
        Dp[IDX(i + 1)] = D;
      }
    }
  }
  // backward: T - lam I = U Dm U^T; gamma_i = Dp_i + Dm_i - (d_i - lam)
  double Dm = d[n - 1] - lam;
  if (fabs(Dm) < piv) Dm = -piv;
  double gbest = fabs(Dp[IDX(n - 1)] + Dm - (d[n - 1] - lam));
  int r = n - 1;
  for (int i0 = n - 2; i0 >= 0; i0 -= U) {
    double ev[U], dv[U], pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = max(i0 - u, 0);
      ev[u] = e[i];
      dv[u] = d[i];
      pv[u] = Dp[IDX(i)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 - u;
      if (i >= 0) {
        const double uu = ev[u] / Dm;
        Z[IDX(i)] = uu;  // U factor of row i (used for z_{i+1} = -u_i z_i)
        Dm = (dv[u] - lam) - uu * ev[u];
        if (fabs(Dm) < piv) Dm = -piv;
        const double g = fabs(pv[u] + Dm - (dv[u] - lam));
        if (g < gbest) {
          gbest = g;
          r = i;
        }
      }
    }
  }
  // substitution from the twist
  double nrm = 1.0;
  double z = 1.0;
  for (int i0 = r - 1; i0 >= 0; i0 -= U) {
    double lv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) lv[u] = Lf[IDX(max(i0 - u, 0))];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 - u;
      if (i >= 0) {
        z = -lv[u] * z;
        Dp[IDX(i)] = z;  // Dp is free below the twist: park z there
        nrm = fma(z, z, nrm);
      }
    }
  }
  z = 1.0;
  for (int i0 = r; i0 < n - 1; i0 += U) {
    double uv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) uv[u] = Z[IDX(min(i0 + u, n - 2))];  // the u factors of rows i0 .. i0 + U - 1
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u;
      if (i < n - 1) {
        z = -uv[u] * z;
        Lf[IDX(i + 1)] = z;  // park in Lf (free above the twist)
        nrm = fma(z, z, nrm);
      }
    }
  }
  const double s = 1.0 / sqrt(nrm);
#pragma unroll 8
  for (int i = 0; i < r; ++i) Z[IDX(i)] = Dp[IDX(i)] * s;
  Z[IDX(r)] = s;
#pragma unroll 8
  for (int i = r + 1; i < n; ++i) Z[IDX(i)] = Lf[IDX(i)] * s;
}

// Z (n x nvec row-major, fp64) <- Q Z, Q = H_0 ... H_{n-3} (reflector j: v = A[j+1:, j], tau[j]); result as fp32 col-major n x nvec.
// One workgroup per 8 eigenvectors, tile in LDS.
constexpr int TD_B_T = 1024;  // threads: per reflector every thread walks (n - j) / (TD_B_T / NC) rows twice, a latency chain (256 threads: 9.6 ms per call at n = 2010)
template <int NC>  // eigenvectors per workgroup: 8, or 4 when there are too few of them to give every CU a workgroup
__global__ __launch_bounds__(TD_B_T) void td_back_k(const double* __restrict__ A, const double* __restrict__ tau, int n, const double* __restrict__ Z, int nvec,
                                                    float* __restrict__ out, int v0, int v1 /*the vectors [v0, v1) of the nvec*/) {
  extern __shared__ double zs[];  // n x NC
  constexpr int NWV = TD_B_T / 64;
  __shared__ double red[NWV][NC];
  constexpr int RL = TD_B_T / NC;  // row lanes
  const int c0 = v0 + (int)blockIdx.x * NC;
  const int t = threadIdx.x, c = t & (NC - 1), rl = t / NC, lane = t & 63, wave = t >> 6;
  const bool live = c0 + c < v1;
  for (int i = rl; i < n; i += RL) zs[i * NC + c] = live ? Z[(size_t)i * nvec + c0 + c] : 0.0;
  __syncthreads();
  for (int j = n - 3; j >= 0; --j) {
    const double tj = tau[j];
    if (tj == 0.0) continue;  // uniform
    const double* vj = A + (size_t)j * n;
    double s = 0.0;
    for (int i = j + 1 + rl; i < n; i += RL) s = fma(vj[i], zs[i * NC + c], s);
    // the row lanes of a column inside the wave by butterflies, the waves in LDS: a fixed order
#pragma unroll
    for (int off = NC; off < 64; off <<= 1) s += __shfl_xor(s, off);
    if (lane < NC) red[wave][lane] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NWV; w += 4) tot += (red[w][c] + red[w + 1][c]) + (red[w + 2][c] + red[w + 3][c]);
    tot *= tj;
    for (int i = j + 1 + rl; i < n; i += RL) zs[i * NC + c] = fma(-tot, vj[i], zs[i * NC + c]);
    __syncthreads();
  }
  if (live)
    for (int i = rl; i < n; i += RL) out[(size_t)(c0 + c) * n + i] = (float)zs[i * NC + c];
}

// ---------------------------------------------------------------------------------------------------------------
// The same back-transformation by blocks of four reflectors in compact WY form (round 6): H_j0 H_j1 H_j2 H_j3 = I - V T V^T
// (T upper triangular 4 x 4, LAPACK dlarft 'forward, columnwise'), so a step is  S = V^T Z,  Y = T S,  Z -= V Y  — one reduction and two
// workgroup barriers per FOUR reflectors, and the block's reflector entries are loaded once, all at once (td_back_k: 2.1 us per reflector,
// of which most is the dependent load of the reflector and the reduction: 4.3 ms per call at n = 2010).
// ---------------------------------------------------------------------------------------------------------------
constexpr int TD_WY = 4;
// T of every block: Tb[b][r][s], r <= s.  One workgroup per block: the six inner products of its reflectors, then the recurrence
//   T[i][i] = tau_i,  T[0:i, i] = -tau_i T[0:i, 0:i] (V[:, 0:i]^T v_i).
__global__ __launch_bounds__(256) void td_wy_T_k(const double* __restrict__ A, const double* __restrict__ tau, int n, double* __restrict__ Tb) {
  __shared__ double sh[4][6];
  const int b = blockIdx.x, j0 = TD_WY * b, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nref = n - 2;  // reflectors 0 .. n - 3
  double g[6] = {0, 0, 0, 0, 0, 0};  // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
  for (int i = j0 + 2 + t; i < n; i += 256) {  // row i carries reflector j iff i >= j + 1
    double v[TD_WY];
#pragma unroll
    for (int r = 0; r < TD_WY; ++r) v[r] = (j0 + r < nref && i >= j0 + r + 1) ? A[(size_t)(j0 + r) * n + i] : 0.0;
    g[0] = fma(v[0], v[1], g[0]);
    g[1] = fma(v[0], v[2], g[1]);
    g[2] = fma(v[0], v[3], g[2]);
    g[3] = fma(v[1], v[2], g[3]);
    g[4] = fma(v[1], v[3], g[4]);
    g[5] = fma(v[2], v[3], g[5]);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g[q] += __shfl_xor(g[q], off);
    if (lane == 0) sh[wave][q] = g[q];
  }
  __syncthreads();
  if (t != 0) return;
  double G[TD_WY][TD_WY] = {};
  const int idx[6][2] = {{0, 1}, {0, 2}, {0, 3}, {1, 2}, {1, 3}, {2, 3}};
  for (int q = 0; q < 6; ++q) G[idx[q][0]][idx[q][1]] = (sh[0][q] + sh[1][q]) + (sh[2][q] + sh[3][q]);
  double T[TD_WY][TD_WY] = {};
  for (int i = 0; i < TD_WY; ++i) {
    const double ti = j0 + i < nref ? tau[j0 + i] : 0.0;
    double w[TD_WY];
    for (int r = 0; r < i; ++r) w[r] = -ti * G[r][i];
    for (int r = 0; r < i; ++r) {
      double s = 0.0;
      for (int q = r; q < i; ++q) s = fma(T[r][q], w[q], s);
      T[r][i] = s;
    }
    T[i][i] = ti;
  }
  for (int r = 0; r < TD_WY; ++r)
    for (int s = 0; s < TD_WY; ++s) Tb[(size_t)b * 16 + r * 4 + s] = T[r][s];
}

template <int NC, bool KEEPV, bool DMA>  // eigenvectors per workgroup (as td_back_k); a thread owns column c = t % NC of the rows rl, rl + RL, ... (RL = 1024 / NC):
                   // its entries of Z stay in registers from the first load to the store; KEEPV: so do a block's reflector entries between the
                   // two phases of a step (NC = 4: 8 rows x 4 doubles; at NC = 8 they would spill and are read again, from L2)
__global__ __launch_bounds__(TD_B_T) void td_back_wy_k(const double* __restrict__ A, const double* __restrict__ Tb, int n, const double* __restrict__ Z, int nvec,
                                                       float* __restrict__ out, int v0, int v1 /*the vectors [v0, v1) of the nvec*/) {
  constexpr int NWV = TD_B_T / 64;
  constexpr int RL = TD_B_T / NC;                              // row lanes
  constexpr int RPT = (TD_NMAX_BACK + RL - 1) / RL;            // rows per thread at most (8 at NC = 4)
  __shared__ double red[NWV][TD_WY][NC], ys[TD_WY][NC];
  // two images of a block's four reflector columns (rows 0 .. nrow - 1 of each, nrow = n rounded up to 128 = 1 KiB pieces): block b - 1 is
  // staged by LDS-DMA while block b is worked on, so no step waits for memory (even n: the pieces are 16-byte aligned; odd n reads memory directly)
  extern __shared__ double vb[];
  const int nrow = (n + 127) & ~127;
  constexpr bool dma = DMA;  // even n: the pieces are 16-byte aligned
  const int c0 = v0 + (int)blockIdx.x * NC;
  const int t = threadIdx.x, c = t & (NC - 1), rl = t / NC, lane = t & 63, wave = t >> 6;
  const bool live = c0 + c < v1;
  const int nref0 = n - 2;
  auto stage = [&](int b, int buf) {  // this wave's 1-KiB pieces of block b: piece = (reflector r, chunk ch)
    const int npiece = TD_WY * (nrow / 128);
    for (int pc = wave; pc < npiece; pc += NWV) {
      const int r = pc / (nrow / 128), ch = pc - r * (nrow / 128);
      if (TD_WY * b + r < nref0) {  // wave-uniform
        const char* gp = reinterpret_cast<const char*>(A + (size_t)(TD_WY * b + r) * n) + (size_t)ch * 1024 + (size_t)lane * 16;
        char* lp = reinterpret_cast<char*>(vb + ((size_t)buf * TD_WY + r) * nrow) + (size_t)ch * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp, (__attribute__((address_space(3))) void*)lp, 16, 0, 0);
      }
    }
  };
  double z[RPT];
#pragma unroll
  for (int q = 0; q < RPT; ++q) {
    const int i = rl + q * RL;
    z[q] = (live && i < n) ? Z[(size_t)i * nvec + c0 + c] : 0.0;
  }
  const int nref = n - 2, nblk = (nref + TD_WY - 1) / TD_WY;
  if (dma) {
    stage(nblk - 1, (nblk - 1) & 1);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
  }
  for (int b = nblk - 1; b >= 0; --b) {
    const int j0 = TD_WY * b;
    // the next block's image: its buffer was last read before the previous step's first barrier
    if (dma && b > 0) stage(b - 1, (b - 1) & 1);
    const double* vcur = vb + (size_t)(b & 1) * TD_WY * nrow;
    // the block's reflector entries of this thread's rows (rows above a reflector's support: zero), from the image in LDS
    // the block's T (uniform address: scalar loads), asked for before the reflector entries so that it is there when wave 0 needs it
    double Tr[TD_WY][TD_WY];
#pragma unroll
    for (int r = 0; r < TD_WY; ++r)
#pragma unroll
      for (int q = r; q < TD_WY; ++q) Tr[r][q] = Tb[(size_t)b * 16 + r * 4 + q];
    auto ldv = [&](int q, int r) {
      const int i = rl + q * RL;
      return (i < n && j0 + r < nref && i >= j0 + r + 1) ? (dma ? vcur[(size_t)r * nrow + i] : A[(size_t)(j0 + r) * n + i]) : 0.0;
    };
    double v[KEEPV ? RPT : 1][TD_WY];
    double s[TD_WY] = {0.0, 0.0, 0.0, 0.0};
    if (KEEPV) {
#pragma unroll
      for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int r = 0; r < TD_WY; ++r) v[q][r] = ldv(q, r);
#pragma unroll
      for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int r = 0; r < TD_WY; ++r) s[r] = fma(v[q][r], z[q], s[r]);
    } else {
#pragma unroll
      for (int q = 0; q < RPT; ++q)
#pragma unroll
        for (int r = 0; r < TD_WY; ++r) s[r] = fma(ldv(q, r), z[q], s[r]);
    }
    // the row lanes of a column inside the wave by butterflies, the waves in LDS: a fixed order
#pragma unroll
    for (int r = 0; r < TD_WY; ++r) {
#pragma unroll
      for (int off = NC; off < 64; off <<= 1) s[r] += __shfl_xor(s[r], off);
      if (lane < NC) red[wave][r][lane] = s[r];
    }
    __syncthreads();
    if (t < TD_WY * NC) {  // thread (r, c') of wave 0: S[r][c'] over the waves ...
      const int r = t / NC, cc = t - r * NC;
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < NWV; w += 4) tot += (red[w][r][cc] + red[w + 1][r][cc]) + (red[w + 2][r][cc] + red[w + 3][r][cc]);
      red[0][r][cc] = tot;  // (only this thread reads red[.][r][cc])
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (t < NC) {  // ... then Y = T S, a column per thread
      double S[TD_WY];
#pragma unroll
      for (int r = 0; r < TD_WY; ++r) S[r] = red[0][r][t];
#pragma unroll
      for (int r = 0; r < TD_WY; ++r) {
        double y = 0.0;
#pragma unroll
        for (int q = r; q < TD_WY; ++q) y = fma(Tr[r][q], S[q], y);
        ys[r][t] = y;
      }
    }
    if (dma) __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces of the next block have landed (they had the whole step to)
    __syncthreads();
    double y[TD_WY];
#pragma unroll
    for (int r = 0; r < TD_WY; ++r) y[r] = ys[r][c];
#pragma unroll
    for (int q = 0; q < RPT; ++q)
#pragma unroll
      for (int r = 0; r < TD_WY; ++r) z[q] = fma(-(KEEPV ? v[q][r] : ldv(q, r)), y[r], z[q]);
    // (the next block's first barrier stands between these reads of ys / red[0] and their next writes)
  }
  if (live) {
#pragma unroll
    for (int q = 0; q < RPT; ++q) {
      const int i = rl + q * RL;
      if (i < n) out[(size_t)(c0 + c) * n + i] = (float)z[q];
    }
  }
}

// max over vectors of | <z_c, z_{c+q}> | (q = 1..4) and | |z_c|^2 - 1 |, as the bits of a non-negative float
__global__ __launch_bounds__(256) void td_check_k(const float* __restrict__ Zc, int n, int nvec, unsigned int* __restrict__ worst) {
  __shared__ double sh[16];
  const int c = blockIdx.x;
  double acc[5] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < n; i += 256) {
    const double a = Zc[(size_t)c * n + i];
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (c + q < nvec) acc[q] = fma(a, (double)Zc[(size_t)(c + q) * n + i], acc[q]);
  }
  double dev = 0.0;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const double s = td_block_sum(acc[q], sh);
    if (c + q < nvec) dev = fmax(dev, fabs(q == 0 ? s - 1.0 : s));
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(worst, __float_as_uint((float)dev));
}

// A (fp64, column-major) = the symmetric matrix whose upper triangle is S's
__global__ __launch_bounds__(256) void td_sym_k(const float* __restrict__ S, int n, double* __restrict__ A) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int jj = (int)(idx / n), i = (int)(idx - (size_t)jj * n);
  A[idx] = (i <= jj) ? (double)S[(size_t)jj * n + i] : (double)S[(size_t)i * n + jj];
}
}  // namespace

// evals_host: all n eigenvalues, descending; vecs_dev: the nvec leading eigenvectors (n x nvec col-major fp32).
// Returns 1 (no error) if the result failed the orthogonality check and the caller should use another solver.
int k_tridiag_eig(isle_ctx* c, const float* S_host, int n, float* evals_host, float* vecs_dev, int nvec) {
  if (n < 3 || n > TD_NMAX_BACK || n > TD_ROWS * 16) return 1;
  nvec = std::max(1, std::min(nvec, n));
  const size_t nn = (size_t)n * n;
  // the fp32 matrix goes up once; td_sym_k makes the fp64 working copy from its upper triangle (LAPACK 'U' semantics) on the
  // device — at n = 2000 the host loop and the 32 MB pageable copy it replaced cost ~10 ms per solve
  HIPCHK(c, c->evd_in.reserve(nn));
  HIPCHK(c, hipMemcpyAsync(c->evd_in.p, S_host, nn * sizeof(float), hipMemcpyHostToDevice, c->stream));
  int CB = 16;
  while (CB < 64 && (n + CB - 1) / CB > 32) CB *= 2;
  const int ncb_max = (n + CB - 1) / CB;
  // workspace (doubles): A | part | w | d | e | tau | lam | Dp | Lf | Z ; then the check word
  const int nrb_max = TD_NRB;
  const size_t need = nn + (size_t)ncb_max * n + 7 * (size_t)n + 3 * (size_t)n * nvec + 16 + ((size_t)n * nrb_max + 1) / 2 + 8 + 4 * (size_t)n + 8 +
                      (GBH_STATE_WORDS + GBH_LINE) / 2;  // + the persistent form's barrier state (128-byte aligned)
  HIPCHK(c, c->jacW.reserve(need));
  double* A = c->jacW.p;
  double* part = A + nn;
  double* psum = part + (size_t)ncb_max * n;
  double* w = psum + n;
  double* d = w + n;
  double* e = d + n;
  double* tau = e + n;
  double* lam = tau + n;
  double* Dp = lam + n;
  double* Lf = Dp + (size_t)n * nvec;
  double* Z = Lf + (size_t)n * nvec;
  unsigned int* worst = reinterpret_cast<unsigned int*>(Z + (size_t)n * nvec);
  unsigned int* tickets = worst + 2;  // n + 1 counters, then n x nrb_max row-block counters
  unsigned int* tickets_rb = tickets + n + 2;
  // persistent form: 2 x (p | next column) behind the counters, rounded up to a double boundary
  double* pv = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(tickets_rb + (size_t)n * nrb_max + 2) + 7) & ~(uintptr_t)7);
  unsigned int* bar_state = reinterpret_cast<unsigned int*>((reinterpret_cast<uintptr_t>(pv + 4 * (size_t)n) + 127) & ~(uintptr_t)127);
  ISLECHK(isle_max_lds(c, (const void*)td_back_k<8>, TD_NMAX_BACK * 8 * (int)sizeof(double)));
  ISLECHK(isle_max_lds(c, (const void*)td_back_k<4>, TD_NMAX_BACK * 4 * (int)sizeof(double)));
  ISLECHK(isle_max_lds(c, (const void*)td_back_wy_k<4, true, true>, 2 * TD_WY * TD_NMAX_BACK * (int)sizeof(double)));
  const bool small = n <= TD_ROWS * 4;
  // persistent form: G workgroups, each with its columns (ncl of them) plus v and w in LDS
  int pG = TD_P_GSMALL;
  {
    const int ncl_max = (int)(TD_P_LDS / sizeof(double) / (size_t)n) - 2;
    if (ncl_max >= 1) pG = std::max(pG, (n + ncl_max - 1) / ncl_max);  // what the LDS needs
    else pG = 1 << 30;
  }
  // one workgroup per CU at most: all must be resident.  A time-out at the grid barrier (the GPU is shared with other work) is
  // remembered: later solves of this context go straight to the launch chain instead of paying the time-out again.
  bool persist = pG <= c->num_cus && n <= TD_NMAX_BACK && !c->knob_on(KN_TD_CHAIN) && !c->td_persist_failed;
  if (c->multi()) {  // every rank must take the same form (their roundings differ): the form is agreed, like the bail-out below
    unsigned int no = persist ? 0u : 1u;
    HIPCHK(c, hipMemcpyAsync(tickets + 1, &no, sizeof no, hipMemcpyHostToDevice, c->stream));
    ISLECHK(isle_allreduce(c, tickets + 1, 1, ISLE_DT_U32, true));
    HIPCHK(c, hipMemcpyAsync(&no, tickets + 1, sizeof no, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    persist = no == 0u;
  }
  const size_t p_lds = ((size_t)((n + pG - 1) / pG) * n + 2 * (size_t)n) * sizeof(double);
  for (int attempt = 0; attempt < 2; ++attempt) {
    hipLaunchKernelGGL(td_sym_k, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, c->stream, c->evd_in.p, n, A);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemsetAsync(tau, 0, (size_t)n * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(e, 0, (size_t)n * sizeof(double), c->stream));
    HIPCHK(c, hipMemsetAsync(worst, 0, ((size_t)n + 4 + (size_t)n * nrb_max + 4) * sizeof(unsigned int), c->stream));
    // ---- 1. tridiagonalisation
    if (persist) {
      // one launch, matrix resident in LDS, one grid barrier per column (td_persist_k); bar_state = the barrier's counters (zeroed per
      // launch: all bases 0), tickets[1] = abort
      GbHierArgs bar = {};
      bar.st = bar_state;
      HIPCHK(c, hipMemsetAsync(bar_state, 0, GBH_STATE_WORDS * sizeof(unsigned int), c->stream));
      const char* bk = c->knob(KN_TD_BAR);
      if (bk && !strcmp(bk, "flat")) {
        ISLECHK(isle_max_lds(c, (const void*)td_persist_k<0>, (int)TD_P_LDS));
        hipLaunchKernelGGL(td_persist_k<0>, dim3(pG), dim3(TD_P_T), p_lds, c->stream, A, n, d, e, tau, pv, bar, tickets + 1);
      } else if (bk && !strcmp(bk, "hier")) {
        ISLECHK(isle_max_lds(c, (const void*)td_persist_k<1>, (int)TD_P_LDS));
        hipLaunchKernelGGL(td_persist_k<1>, dim3(pG), dim3(TD_P_T), p_lds, c->stream, A, n, d, e, tau, pv, bar, tickets + 1);
      } else {
        ISLECHK(isle_max_lds(c, (const void*)td_persist_k<2>, (int)TD_P_LDS));
        hipLaunchKernelGGL(td_persist_k<2>, dim3(pG), dim3(TD_P_T), p_lds, c->stream, A, n, d, e, tau, pv, bar, tickets + 1);
      }
      HIPCHK(c, hipGetLastError());
      if (const char* fb = c->knob(KN_TD_FORCE_BAIL_RANK)) {  // test hook: this rank behaves as if its barrier had timed out
        if (atoi(fb) == c->rank) {
          const unsigned int one = 1u;
          HIPCHK(c, hipMemcpyAsync(tickets + 1, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
        }
      }
      // The bail-out depends on timing, and the launch chain rounds differently from the persistent form: in a sharded run every
      // rank must take the same one, or their replicated Ritz data (and then restart decisions, and then the sequence of
      // collectives) drift apart.  One rank bailing out sends all of them to the chain.
      if (c->multi()) ISLECHK(isle_allreduce(c, tickets + 1, 1, ISLE_DT_U32, true));
      unsigned int aborted = 0;
      HIPCHK(c, hipMemcpyAsync(&aborted, tickets + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (!aborted) break;
      if (!c->td_persist_failed)
        fprintf(stderr, "[isle_hip] persistent tridiagonalisation gave up at a grid barrier (n = %d): using the launch chain from now on\n", n);
      c->td_persist_failed = true;
      persist = false;
      continue;
    }
    if (small) hipLaunchKernelGGL((td_first_k<4>), dim3(1), dim3(TD_ROWS), 0, c->stream, A, n, w, d, e, tau);
    else hipLaunchKernelGGL((td_first_k<16>), dim3(1), dim3(TD_ROWS), 0, c->stream, A, n, w, d, e, tau);
    // F_j = update of step j fused with the products of step j + 1 and, in its last workgroup, step j + 1 itself
    for (int j = -1; j <= n - 3; ++j) {
      const int m = n - (j + 2);
      const int ncb = (m + CB - 1) / CB;
      const dim3 g((m + TD_ROWS - 1) / TD_ROWS, ncb);
      if (small)
        hipLaunchKernelGGL((td_update_symv_k<4>), g, dim3(TD_ROWS), 3 * CB * sizeof(double), c->stream, A, n, j, j >= 0 ? 1 : 0, CB, w, part, psum, d, e, tau,
                           tickets, tickets_rb);
      else
        hipLaunchKernelGGL((td_update_symv_k<16>), g, dim3(TD_ROWS), 3 * CB * sizeof(double), c->stream, A, n, j, j >= 0 ? 1 : 0, CB, w, part, psum, d, e, tau,
                           tickets, tickets_rb);
    }
    HIPCHK(c, hipGetLastError());
    break;
  }
  // ---- 2. eigenvalues, 3. eigenvectors of T, 4. back-transformation, 5. check
  hipLaunchKernelGGL(td_bisect_k, dim3((nvec + 3) / 4), dim3(256), 2 * (size_t)n * sizeof(double), c->stream, d, e, n, nvec, lam);
  // Several ranks, ISLE_EVD_SPLIT=1 (round 5; opt-in since round 6): the eigenvectors are independent of each other from here on — a thread per
  // vector in td_vectors_k, a workgroup per 8 (4) vectors in td_back_k — so rank r computes the vectors [r kc, (r + 1) kc) only and the columns
  // are all-gathered: every rank holds the same bits, whoever computed them.  Round 5 made it the default expecting the stage's 5 - 8 ms at
  // n = 2010 to divide by the number of ranks; they do not: both kernels are chains of n dependent steps (a reflector / a row per step) whose
  // length does not depend on how many vectors a rank holds — 250 workgroups or 32, td_back_k takes its 4.2 ms — so the split buys an idle GPU and
  // costs an 8 MB all-gather per EVD that has never run over RCCL.  Default: every rank computes every vector.
  int v0 = 0, v1 = nvec, kc = 0;
  if (c->multi() && c->knob_on(KN_EVD_SPLIT) && !c->knob_zero(KN_EVD_SPLIT)) {
    kc = ((nvec + c->world - 1) / c->world + 7) & ~7;
    if ((size_t)kc * c->world <= (size_t)n) {  // the gathered block fits the caller's n x n array
      v0 = std::min(nvec, c->rank * kc);
      v1 = std::min(nvec, v0 + kc);
    } else {
      kc = 0;
    }
  }
  const int nv = v1 - v0;
  if (nv > 0) {
    hipLaunchKernelGGL(td_vectors_k, dim3((nv + 63) / 64), dim3(64), 0, c->stream, d, e, n, lam, nvec, Dp, Lf, Z, v0, v1);
    // back-transformation by blocks of four reflectors (compact WY): the T factors first (the buffer of the launch chain's partial sums is free by now);
    // ISLE_TD_BACK=seq: reflector by reflector (td_back_k)
    const char* e_back = c->knob(KN_TD_BACK);
    if (e_back && e_back[0] == 's') {
      if ((nvec + 7) / 8 > c->num_cus / 2 && !kc)
        hipLaunchKernelGGL(td_back_k<8>, dim3((nv + 7) / 8), dim3(TD_B_T), (size_t)n * 8 * sizeof(double), c->stream, A, tau, n, Z, nvec, vecs_dev, v0, v1);
      else  // few eigenvectors: four per workgroup, twice the workgroups, half the rows per thread (two per workgroup measured no better)
        hipLaunchKernelGGL(td_back_k<4>, dim3((nv + 3) / 4), dim3(TD_B_T), (size_t)n * 4 * sizeof(double), c->stream, A, tau, n, Z, nvec, vecs_dev, v0, v1);
    } else {
      const int nblk = (n - 2 + TD_WY - 1) / TD_WY;
      double* Tb = part;
      hipLaunchKernelGGL(td_wy_T_k, dim3(nblk), dim3(256), 0, c->stream, A, tau, n, Tb);
      // four eigenvectors per workgroup whatever their number (eight would spill the block's reflector entries: beyond 4 x 256 vectors the
      // workgroups run in two rounds, still ahead of the eight-column sequential form)
      if (n & 1) hipLaunchKernelGGL((td_back_wy_k<4, true, false>), dim3((nv + 3) / 4), dim3(TD_B_T), 0, c->stream, A, Tb, n, Z, nvec, vecs_dev, v0, v1);
      else hipLaunchKernelGGL((td_back_wy_k<4, true, true>), dim3((nv + 3) / 4), dim3(TD_B_T), (size_t)2 * TD_WY * ((n + 127) & ~127) * sizeof(double), c->stream, A, Tb, n, Z, nvec, vecs_dev, v0, v1);
    }
  }
  HIPCHK(c, hipGetLastError());
  if (kc) {
    TimeScope ts(c, ISLE_T_COMM);
    ISLECHK(isle_allgather(c, vecs_dev + (size_t)c->rank * kc * n, vecs_dev, (size_t)kc * n, ISLE_DT_F32));
  }
  hipLaunchKernelGGL(td_check_k, dim3(nvec), dim3(256), 0, c->stream, vecs_dev, n, nvec, worst);
  HIPCHK(c, hipGetLastError());
  std::vector<double> ev(n);
  unsigned int wbits = 0;
  HIPCHK(c, hipMemcpyAsync(ev.data(), lam, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&wbits, worst, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float dev;
  static_assert(sizeof(dev) == sizeof(wbits), "");
  memcpy(&dev, &wbits, sizeof dev);
#ifdef TD_STAMPS
  {
    unsigned long long hs[16];
    hipMemcpyFromSymbol(hs, HIP_SYMBOL(td_stamps), sizeof hs);
    fprintf(stderr, "[td_stamps n=%d] cycles per column: symv %.0f | barrier %.0f | loads+dot %.0f | dot sum %.0f | w %.0f | cnr+rank2 %.0f | reflector: sum %.0f math+barrier %.0f rest %.0f\n", n, hs[0] / (double)n, hs[1] / (double)n, hs[2] / (double)n, hs[3] / (double)n, hs[4] / (double)n, hs[5] / (double)n, hs[7] / (double)n, hs[8] / (double)n, hs[6] / (double)n);
    unsigned long long z[16] = {};
    hipMemcpyToSymbol(HIP_SYMBOL(td_stamps), z, sizeof z);
  }
#endif
  if (c->knob_on(KN_DEBUG_EVD)) fprintf(stderr, "[evd n=%d] tridiagonal solver: nvec %d, worst orthogonality defect %.3g\n", n, nvec, (double)dev);
  if (!(dev <= 1e-6f)) return 1;
  for (int i = 0; i < n; ++i) evals_host[i] = (float)ev[i];
  return 0;
}

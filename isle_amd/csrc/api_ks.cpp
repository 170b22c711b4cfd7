// isle_amd/csrc/api_ks.cpp — the restarted block Krylov-Schur solver behind isle_hip_block_ks / isle_hip_block_ks_dense
// (BlockKs::init / expand / truncate / compute, block-ks/restarted_block_ks.h:62-321; compute_qr, block-ks/ks_utils.h:43-127; the glue of
// src/sparseMatrix.cpp:1195-1240).  The host keeps the small projected matrix H and the restart logic; everything V-sized runs in dense.hip /
// evd_tridiag.hip / gram_lds.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "api_internal.h"

constexpr int KS_AGREE = 40;  // int slots [40, 43) of the expand mailbox: max rank, -min rank, max status over all ranks
__global__ void ks_agree_pack_k(int* meta) {
  if (threadIdx.x == 0) {
    meta[KS_AGREE] = meta[0];
    meta[KS_AGREE + 1] = -meta[0];
    meta[KS_AGREE + 2] = meta[1];
  }
}

namespace {

struct HMat {  // small col-major float matrix on the host (the projected matrix H)
  size_t r = 0, c = 0, ld = 0, cap_c = 0;  // r x c in use inside an ld x cap_c allocation (zero outside what was written)
  std::vector<float> a;
  HMat() {}
  HMat(size_t r_, size_t c_) : r(r_), c(c_), ld(r_), cap_c(c_), a(r_ * c_, 0.f) {}
  // room to grow: the Krylov expansion appends blocks of rows and columns in place (at ncv = 2010 the matrix is 16 MB, and every
  // fresh copy of it cost the host 2 - 6 ms with the GPU idle)
  HMat(size_t r_, size_t c_, size_t cap_r_, size_t cap_c_) : r(r_), c(c_), ld(std::max(r_, cap_r_)), cap_c(std::max(c_, cap_c_)), a(ld * cap_c, 0.f) {}
  float& operator()(size_t i, size_t j) { return a[j * ld + i]; }
  float operator()(size_t i, size_t j) const { return a[j * ld + i]; }
};
HMat hsub(const HMat& m, size_t r0, size_t c0, size_t r1, size_t c1) {  // inclusive bounds (arma submat)
  HMat o(r1 - r0 + 1, c1 - c0 + 1);
  for (size_t j = c0; j <= c1; ++j)
    for (size_t i = r0; i <= r1; ++i) o(i - r0, j - c0) = m(i, j);
  return o;
}
}  // namespace


// ------------------------------------------------------------------------------------------
// Panel QR: rank-revealing CholQR2 on an fp64 Gram matrix.
// utils::compute_qr (block-ks/ks_utils.h:43-127) is MGS in double with one DGKS correction and drops
// columns whose residual norm is < 1e-6.  Cholesky of G = F^T F processed column by column IS that MGS
// in exact arithmetic (pivot_i^2 = residual norm^2 of column i); a second pass restores orthogonality
// to working precision.  F: n x w (device, destroyed).  Q: n x rank written to Qdst.  R: rank x w.
// ------------------------------------------------------------------------------------------
static int dev_qr(isle_ctx* c, float* F, uint64_t n, int w, float* Qdst, std::vector<float>& R, int* rank_out) {
  std::vector<float> Rfull((size_t)w * w, 0.f);
  int rk = 0;
  ISLECHK(k_panel_qr(c, F, n, w, Qdst, Rfull.data(), &rk));
  ISLECHK(agree_i32(c, rk, "the rank of a start / repair block"));
  *rank_out = rk;
  R.assign(Rfull.begin(), Rfull.begin() + (size_t)rk * w);
  return 0;
}

// ------------------------------------------------------------------------------------------
// Restarted block Krylov-Schur
// ------------------------------------------------------------------------------------------
namespace {
struct Ks {
  isle_ctx* c;
  size_t nev, ncv, maxit, blk;
  uint64_t dim;
  float tol;
  HMat H;
  size_t vcols = 0, nconv = 0, n_restarts = 0, last_j = 0;
  long napplies = 0;
  uint64_t seed, draws = 0;
  // The ProdOp plug-in (block-ks/restarted_block_ks.h:18-40): the context's B B^T (MKL_SpSpTrProd, include/matUtils.h:336-365), or a
  // dense symmetric dim x dim matrix on the device (ArmaMatProdOp, block-ks/ks_utils.h:167-182) when dense_A is set.
  const float* dense_A = nullptr;
  const float* start_dev = nullptr;  // optional dim x blk start block (first try of init's draw loop)
  float* Vb() { return c->basis.p; }
  float* col(size_t j) { return c->basis.p + j * dim; }

  int randu(float* F, size_t cols) { return k_randu(c, F, dim * cols, seed + 0x1000 * (++draws)); }

  // orthogonalise F (dim x w) against the first m basis columns, `passes` times; coefficient blocks kept on device
  int ortho(float* F, int w, size_t m, int passes, float* coef_dev = nullptr) {
    HIPCHK(c, c->coef.reserve(3 * (c->basis.cap / dim) * 32));
    float* base = coef_dev ? coef_dev : c->coef.p;
    // Several ranks: the basis is replicated, but nobody needs to orthogonalise ALL rows.  Rank r takes rows [r nloc, (r+1) nloc):
    // its share of V^T F, an all-reduce of the m x w coefficients (80 kB at m = 2000), the update of its rows — for every pass —
    // and at the end the slices of F are all-gathered (4 MB at V = 100k), so every rank again holds the whole, bitwise equal F for
    // the replicated panel QR.  The step is HBM-bound on reading the basis (0.15 s of a 1.15 s C3-shard step): it now divides by
    // the number of ranks at the price of passes + 1 small collectives per step.  Default with several ranks since round 5 (a collective
    // that never completes is turned into ISLE_E_COMM by the watchdog, api.cpp); ISLE_KS_ROWSHARD=0 keeps it replicated.
    const bool shard = c->multi() && !dense_A && !c->knob_zero(KN_KS_ROWSHARD);
    if (!shard) {
      for (int p = 0; p < passes; ++p) {
        float* cf = base + (size_t)p * m * w;
        ISLECHK(k_vtf(c, Vb(), dim, (int)m, F, w, cf));
        ISLECHK(k_update(c, F, dim, w, Vb(), (int)m, cf));
      }
      return 0;
    }
    const uint64_t nloc = (((dim + c->world - 1) / c->world) + 3) & ~3ull;  // rows per rank, a multiple of 4 (16-byte aligned slices)
    const uint64_t r0 = std::min<uint64_t>(dim, (uint64_t)c->rank * nloc), r1 = std::min<uint64_t>(dim, r0 + nloc);
    const uint64_t nl = r1 - r0;
    for (int p = 0; p < passes; ++p) {
      float* cf = base + (size_t)p * m * w;
      ISLECHK(k_vtf(c, Vb() + r0, nl, (int)m, F + r0, w, cf, dim));
      ISLECHK(allreduce_sum<float>(c, cf, m * (size_t)w));
      ISLECHK(k_update(c, F + r0, nl, w, Vb() + r0, (int)m, cf, dim));
    }
    HIPCHK(c, c->ks_gather.reserve((size_t)c->world * nloc * w));
    float* mine = c->ks_gather.p + (size_t)c->rank * nloc * w;
    ISLECHK(k_slice_rows(c, F, dim, w, r0, nl, nloc, mine, true));  // pack my rows (zero padded to nloc)
    {
      TimeScope ts(c, ISLE_T_COMM);
      ISLECHK(isle_allgather(c, mine, c->ks_gather.p, nloc * (size_t)w, ISLE_DT_F32));
    }
    for (int r = 0; r < c->world; ++r) {
      const uint64_t q0 = std::min<uint64_t>(dim, (uint64_t)r * nloc), q1 = std::min<uint64_t>(dim, q0 + nloc);
      if (r != c->rank && q1 > q0) ISLECHK(k_slice_rows(c, F, dim, w, q0, q1 - q0, nloc, c->ks_gather.p + (size_t)r * nloc * w, false));
    }
    return 0;
  }

  // rank repair shared by init (:238-258) and expand (:106-132)
  int repair(size_t& nvecs, size_t target, size_t width) {
    size_t tries = 0;
    while (nvecs < target && tries < 100) {
      tries++;
      ISLECHK(randu(c->Fbuf.p, width));
      ISLECHK(ortho(c->Fbuf.p, (int)width, nvecs, 2));
      std::vector<float> R2;
      int rk2 = 0;
      const size_t room = target - nvecs;
      // Q lands in Tmp first: only `room` columns may be appended
      ISLECHK(dev_qr(c, c->Fbuf.p, dim, (int)width, c->Tmp.p, R2, &rk2));
      const size_t take = std::min<size_t>((size_t)rk2, room);
      if (take) HIPCHK(c, hipMemcpyAsync(col(nvecs), c->Tmp.p, take * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      nvecs += take;
    }
    if (nvecs < target) return isle_fail(c, ISLE_E_NUMERIC, "unable to find new starting basis for Arnoldi expansion");
    return 0;
  }

  int apply(const float* X, float* Z) {
    napplies++;
    if (dense_A) return k_gemm_nn(c, dense_A, dim, (int)dim, X, (int)dim, (int)blk, Z, ISLE_T_GRAM_PASS1);
    return gram_apply_dev(c, X, (int)blk, Z);
  }

  int init() {  // :203-259
    std::vector<float> R;
    int rank = 0;
    bool first = true;
    do {  // :211-218: redrawn until the start block has full rank
      if (first && start_dev) HIPCHK(c, hipMemcpyAsync(c->Fbuf.p, start_dev, dim * blk * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
      else ISLECHK(randu(c->Fbuf.p, blk));
      first = false;
      ISLECHK(dev_qr(c, c->Fbuf.p, dim, (int)blk, col(0), R, &rank));
    } while ((size_t)rank < blk);
    float* V1 = c->Fbuf.p;
    ISLECHK(apply(col(0), V1));
    ISLECHK(ortho(V1, (int)blk, blk, 2));  // H = V^T V1; V1 -= V H; C = V^T V1; H += C; V1 -= V C
    std::vector<float> hc(2 * blk * blk);
    HIPCHK(c, hipMemcpyAsync(hc.data(), c->coef.p, hc.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    ISLECHK(dev_qr(c, V1, dim, (int)blk, col(blk), R, &rank));  // synchronises the stream
    H = HMat(2 * blk, blk, std::max<size_t>(ncv, 2 * blk) + blk, blk + (std::max<size_t>(ncv, 2 * blk) + blk - 2 * blk));
    for (size_t j = 0; j < blk; ++j) {
      for (size_t i = 0; i < blk; ++i) H(i, j) = hc[j * blk + i] + hc[blk * blk + j * blk + i];
      for (int i = 0; i < rank; ++i) H(blk + i, j) = R[j * rank + i];
    }
    vcols = blk + rank;
    if ((size_t)rank < blk) ISLECHK(repair(vcols, 2 * blk, blk - rank));
    vcols = 2 * blk;
    return 0;
  }

  int expand() {  // :62-136
    // H grows by blk rows and columns per step; it is kept in a work matrix of the final size while the loop runs (copying
    // the whole of H at every step cost ~10 ms of host time per solve at ncv = 410, with the GPU idle behind the QR's sync)
    isle_host_mark("expand: entry");
    const size_t cap_r = std::max<size_t>(ncv, H.r) + blk, cap_c = H.c + (cap_r - H.r);
    if (H.ld < cap_r || H.cap_c < cap_c) {  // init() and truncate() allocate with this room, so this copy is the exception
      HMat W(H.r, H.c, cap_r, cap_c);
      for (size_t j = 0; j < H.c; ++j)
        for (size_t i = 0; i < H.r; ++i) W(i, j) = H(i, j);
      H = std::move(W);
    }
    HMat& W = H;  // grows in place: rows hr.. and columns hcn.. are zero until a step writes them
    size_t hr = H.r, hcn = H.c;
    auto shrink = [&]() {
      H.r = hr;
      H.c = hcn;
    };
    // Pipelined: after the QR of step i is enqueued, the operator application and orthogonalisation of step i + 1 are
    // enqueued too (they only need Q on the device, assuming full rank), and the host then waits for an event recorded
    // behind the QR to fold R and the coefficients into H.  Everything the host needs from a step — rank and status of the QR,
    // R, the three coefficient blocks — is written into one device mailbox and comes back as ONE copy (every small copy
    // costs ~20 us of queue time).  A rank-deficient panel (never seen on thresholded matrices) discards the speculative
    // work and repairs, as the synchronous form (ISLE_KS_SYNC=1) does.
    // Round 6: the copy runs on a stream of its own behind an event — in the main stream it held the next step's kernels back for the ~10 us
    // a 160 kB transfer over PCIe takes, 300 times per solve — and the device mailbox is double-buffered by slot, so that the next step's
    // coefficients (written by the speculative orthogonalisation) never land in a mailbox whose copy may still be reading it: a slot is
    // written again only after the host has waited for its copy.
    const bool pipelined = !c->knob_on(KN_KS_SYNC);
    // Passes of block Gram-Schmidt against the basis per step.  The reference makes three (CGS + 2 DGKS, :83-91); the second
    // already leaves coefficients at rounding level ("twice is enough"), so two are made here and the third
    // block of coefficients that the reference adds into H is zero.  ISLE_KS_ORTHO_PASSES=3 restores the reference's count.
    int npass = 2;
    if (const char* e = c->knob(KN_KS_ORTHO_PASSES)) npass = std::max(2, std::min(3, atoi(e)));
    constexpr size_t MB_R = 64, MB_COEF = 64 + 32 * 32;  // mailbox offsets (floats): [meta ints | R | coefficients]
    const size_t mb_floats = MB_COEF + 3 * cap_r * blk;
    HIPCHK(c, c->ks_mail.reserve(2 * mb_floats));
    for (int i = 0; i < 2; ++i) {
      if (!c->ks_ev[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ks_ev[i], hipEventDisableTiming));
      if (!c->ks_ev_ready[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ks_ev_ready[i], hipEventDisableTiming));
    }
    if (!c->copy_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    std::vector<float> host_mail_pageable[2];
    float* host_mail[2];
    for (int i = 0; i < 2; ++i) {
      if (mb_floats * sizeof(float) <= isle_ctx::PIN_MAIL_SLOT) {  // page-locked: the copy below then really is asynchronous
        host_mail[i] = reinterpret_cast<float*>(c->pin + isle_ctx::PIN_MAIL + (size_t)i * isle_ctx::PIN_MAIL_SLOT);
      } else {
        host_mail_pageable[i].resize(mb_floats);
        host_mail[i] = host_mail_pageable[i].data();
      }
    }
    isle_host_mark("expand: work matrix ready");
    float* const mail2[2] = {c->ks_mail.p, c->ks_mail.p + mb_floats};
    bool spec = false;  // apply + ortho of the current step already enqueued
    int slot = 0;
    while (hr < ncv) {
      const size_t m = hr;
      float* F = c->Fbuf.p;
      float* mail = mail2[slot];
      if (!spec) {
        ISLECHK(apply(col(hcn), F));
        ISLECHK(ortho(F, (int)blk, m, npass, mail + MB_COEF));
      }
      spec = false;
      if (m + blk > cap_r || hcn + blk > cap_c) return isle_fail(c, ISLE_E_NUMERIC, "expand: projected matrix outgrew its work space");
      ISLECHK(k_panel_qr_kernels(c, F, dim, (int)blk, col(hcn + blk), reinterpret_cast<int*>(mail), mail + MB_R));
      if (c->multi()) {  // the ranks' verdicts on this block travel with the mailbox (see agree_i32)
        hipLaunchKernelGGL(ks_agree_pack_k, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<int*>(mail));
        HIPCHK(c, hipGetLastError());
        TimeScope ts(c, ISLE_T_COMM);
        ISLECHK(isle_allreduce(c, reinterpret_cast<int*>(mail) + KS_AGREE, 3, ISLE_DT_I32, true));
      }
      float* hm = host_mail[slot];
      HIPCHK(c, hipEventRecord(c->ks_ev_ready[slot], c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ks_ev_ready[slot], 0));
      HIPCHK(c, hipMemcpyAsync(hm, mail, (MB_COEF + (size_t)npass * m * blk) * sizeof(float), hipMemcpyDeviceToHost, c->copy_stream));
      HIPCHK(c, hipEventRecord(c->ks_ev[slot], c->copy_stream));
      const bool more = m + blk < ncv;
      if (pipelined && more) {  // speculate: full rank -> next step works on the blk new columns with m + blk basis vectors
        ISLECHK(apply(col(hcn + blk), F));
        ISLECHK(ortho(F, (int)blk, m + blk, npass, mail2[slot ^ 1] + MB_COEF));  // the other slot's mailbox: its last copy has been waited for
        spec = true;
      }
      HIPCHK(c, hipEventSynchronize(c->ks_ev[slot]));
      const int* meta = reinterpret_cast<const int*>(hm);
      if (c->multi() && meta[KS_AGREE] != -meta[KS_AGREE + 1])
        return isle_fail(c, ISLE_E_COMM, "ranks disagree on the rank of a Krylov block (min %d, max %d): replicated state diverged",
                         -meta[KS_AGREE + 1], meta[KS_AGREE]);
      if (meta[1] || (c->multi() && meta[KS_AGREE + 2]))
        return isle_fail(c, ISLE_E_NUMERIC, "CholQR2: second Gram matrix not positive definite");
      const int rk = meta[0];
      const float* hc = hm + MB_COEF;
      const float* Rfull = hm + MB_R;
      for (size_t j = 0; j < blk; ++j)
        for (size_t i = 0; i < m; ++i) {
          float h = hc[j * m + i];
          h = h + hc[m * blk + j * m + i];
          if (npass > 2) h = h + hc[2 * m * blk + j * m + i];
          W(i, hcn + j) = h;
        }
      for (size_t j = 0; j < blk; ++j)
        for (int i = 0; i < rk; ++i) W(m + i, hcn + j) = Rfull[j * rk + i];
      hr = m + blk;
      hcn += blk;
      if ((size_t)rk < blk) {
        if (spec) {  // the speculative step used columns that are about to be replaced
          HIPCHK(c, hipStreamSynchronize(c->stream));
          spec = false;
          napplies--;
        }
        shrink();  // repair() reads H.r / H.c
        size_t nvecs = H.c + rk;
        ISLECHK(repair(nvecs, H.r, blk - rk));
      }
      slot ^= 1;
    }
    isle_host_mark("expand: loop done");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    isle_host_mark("expand: synchronised");
    shrink();
    isle_host_mark("expand: shrink");
    vcols = H.r;
    return 0;
  }

  int truncate() {  // :138-187
    isle_host_mark("truncate: entry");
    const size_t n = H.c - nconv;
    const size_t keep = nev - nconv;  // only the leading `keep` eigenvectors are used below
    // Everything that crosses the bus here lives in the context's page-locked staging area — [subH n x n | vH n x keep | locked
    // rows of H nconv x n | top nconv x keep], 36 MB at k = 1000: copies from freshly allocated pageable vectors blocked the host
    // (registration with the driver) and made their release slow, with the GPU idle in between.
    HIPCHK(c, c->pin_stage_reserve((n * n + n * keep + nconv * n + nconv * keep) * sizeof(float)));
    float* subH = reinterpret_cast<float*>(c->pin_stage);
    float* vH = subH + n * n;
    float* blkH = vH + n * keep;
    float* top = blkH + nconv * n;
    for (size_t j = 0; j < n; ++j) memcpy(subH + j * n, &H(nconv, nconv + j), n * sizeof(float));  // H(nconv:, nconv:), square
    isle_host_mark("truncate: subH extracted");
    std::vector<float> eH(n);
    HIPCHK(c, c->Wf.reserve(n * n));
    ISLECHK(k_eig_small(c, subH, (int)n, eH.data(), c->Wf.p, (int)keep));
    isle_host_mark("truncate: eig_small returned");
    HIPCHK(c, hipMemcpyAsync(vH, c->Wf.p, n * keep * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    // V = [ V(:, :nconv) | V(:, nconv : ncols-blk) * vH(:, :keep) | V(:, tail blk) ]
    ISLECHK(k_gemm_nn(c, col(nconv), dim, (int)n, c->Wf.p, (int)n, (int)keep, c->Tmp.p));
    // top = H(0:nconv, nconv:) * vH(:, :keep)  (:176-178), the coupling of the locked columns with the rotated block: nconv x n x keep
    // multiply-adds — 0.3 G at k = 1000 with 600 pairs locked, 31 ms of host time with the GPU idle when it was a host loop
    if (nconv > 0) {
      HIPCHK(c, c->ks_top.reserve(nconv * n + nconv * keep));
      for (size_t t = 0; t < n; ++t) memcpy(blkH + t * nconv, &H(0, nconv + t), nconv * sizeof(float));
      HIPCHK(c, hipMemcpyAsync(c->ks_top.p, blkH, nconv * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
      ISLECHK(k_gemm_nn(c, c->ks_top.p, nconv, (int)n, c->Wf.p, (int)n, (int)keep, c->ks_top.p + nconv * n));
      HIPCHK(c, hipMemcpyAsync(top, c->ks_top.p + nconv * n, nconv * keep * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->Fbuf.p, col(vcols - blk), blk * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(col(nconv), c->Tmp.p, keep * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(col(nev), c->Fbuf.p, blk * dim * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    vcols = nev + blk;
    isle_host_mark("truncate: device part synchronised");
    // Transform H (:169-184)
    auto vh = [&](size_t i, size_t j) { return vH[j * n + i]; };
    HMat last = hsub(H, H.r - blk, H.c - blk, H.r - 1, H.c - 1);  // blk x blk
    HMat newrows(blk, keep);
    for (size_t j = 0; j < keep; ++j)
      for (size_t t = 0; t < blk; ++t) {
        const float v = vh(n - blk + t, j);
        for (size_t i = 0; i < blk; ++i) newrows(i, j) += last(i, t) * v;
      }
    const size_t grow_r = std::max<size_t>(ncv, nev + blk) + blk;  // what the next expand() asks for
    HMat Hn(nev + blk, nev, grow_r, nev + (grow_r - (nev + blk)));
    for (size_t j = 0; j < nconv; ++j) {  // locked columns keep their entries (rows < nev from the old H; residual rows too)
      for (size_t i = 0; i < nev; ++i) Hn(i, j) = H(i, j);
      for (size_t i = 0; i < blk; ++i) Hn(nev + i, j) = H(nev + i, j);
    }
    for (size_t j = nconv; j < nev; ++j) {
      Hn(j, j) = eH[j - nconv];
      for (size_t i = 0; i < blk; ++i) Hn(nev + i, j) = newrows(i, j - nconv);
      for (size_t i = 0; i < nconv; ++i) Hn(i, j) = top[(j - nconv) * nconv + i];
    }
    H = std::move(Hn);
    isle_host_mark("truncate: H transformed");
    return 0;
  }

  size_t first_unconverged(bool divide) const {  // :278-293
    for (size_t j = 0; j < H.c; ++j) {
      float s = 0.f;
      for (size_t i = H.r - blk; i < H.r; ++i) s += H(i, j) * H(i, j);
      float nrm = std::sqrt(s);
      if (divide) nrm = nrm / H(j, j);
      if (nrm >= tol) return j;
    }
    return H.c;
  }

  int compute() {  // :261-321
    n_restarts = 0;
    nconv = 0;
    ISLECHK(expand());
    while (n_restarts < maxit) {
      ISLECHK(truncate());
      const size_t j = first_unconverged(true);
      ISLECHK(agree_i32(c, (int)j, "the number of converged Ritz pairs"));
      last_j = j;
      if (j == H.c) {
        nconv = H.c;
        break;
      }
      nconv = j;
      ++n_restarts;
      ISLECHK(expand());
    }
    return 0;
  }
};
}  // namespace

int install_U(isle_ctx* c, const float* Ucm_dev, int k) {
  c->ldk = round4(k);
  HIPCHK(c, c->Ucm.reserve((size_t)c->V * k));
  HIPCHK(c, c->Urm.reserve((size_t)c->V * c->ldk));
  if (Ucm_dev != c->Ucm.p)
    HIPCHK(c, hipMemcpyAsync(c->Ucm.p, Ucm_dev, (size_t)c->V * k * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(c->Urm.p, 0, (size_t)c->V * c->ldk * sizeof(float), c->stream));
  ISLECHK(k_transpose(c, c->Ucm.p, c->V, k, c->V, c->Urm.p, c->ldk));  // compute_U_rowmajor :1223-1231
  c->U_k = k;
  c->P_ready = false;
  c->Pt_ready = false;
  c->Pt2_ready = false;
  c->lift_valid = false;
  c->centers_ready = false;
  return 0;
}

// Shared driver of both eigensolver entries: BlockKs(op, nev, ncv, maxit, blk, tol); init(); compute()  (:190-321).
// ncv and nev need not be multiples of the block size: a decomposition grows by whole blocks until it has AT LEAST ncv
// rows (the reference sizes V for exactly ncv columns and overruns it in that case), so the basis holds up to
// ncv + blk - 1 vectors.
static int ks_solve(isle_ctx* c, Ks& ks, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed, float* evals, int* nconv,
                    int* restarts, int* napplies, int* nconv_ref_rule) {
  if (nev < 1 || blk < 1 || blk > 32 || maxit < 1) return isle_fail(c, ISLE_E_ARG, "bad nev/blk/maxit (nev >= 1, 1 <= blk <= 32, maxit >= 1)");
  ks.c = c;
  ks.nev = nev;
  ks.ncv = ncv;
  ks.maxit = maxit;
  ks.blk = (blk < nev) ? blk : 1;  // block-ks/restarted_block_ks.h:198
  ks.tol = tol;
  ks.seed = seed;
  if ((size_t)ncv < (size_t)nev + 2 * ks.blk || (uint64_t)ncv + ks.blk > ks.dim)
    return isle_fail(c, ISLE_E_ARG, "need nev + 2*blk <= ncv and ncv + blk <= operator dimension (nev=%d ncv=%d blk=%zu dim=%llu)", nev, ncv,
                     ks.blk, (unsigned long long)ks.dim);
  HIPCHK(c, c->basis.reserve((size_t)ks.dim * (ncv + 2 * ks.blk)));
  HIPCHK(c, c->Fbuf.reserve((size_t)ks.dim * ks.blk));
  HIPCHK(c, c->Tmp.reserve((size_t)ks.dim * std::max<size_t>(nev, ks.blk)));
  isle_host_mark("ks_solve: entry");
  ISLECHK(ks.init());
  isle_host_mark("ks_solve: init done");
  ISLECHK(ks.compute());
  isle_host_mark("ks_solve: compute done");
  int rc = 0;
  size_t nc = ks.nconv, nc_ref = ks.nconv;
  if (ks.n_restarts == (size_t)maxit) {
    // The reference recomputes residuals from the EXPANDED H without dividing by the Ritz value (:303-317); the last blk rows of
    // an expanded H are [0 ... 0 R], so that rule reports min(first column of the last block, nev) = nev whatever happened
    // (SURVEY App. C #7).  Here: the count of the last restart's residual test, status ISLE_E_NOCONV, and the same Ritz pairs;
    // the reference's figure is available through nconv_ref_rule.
    nc_ref = ks.first_unconverged(false);
    nc = std::min(ks.last_j, (size_t)nev);
    if (nc < (size_t)nev) rc = ISLE_E_NOCONV;
  }
  nc = std::min(nc, (size_t)nev);
  nc_ref = std::min(nc_ref, (size_t)nev);
  for (int i = 0; i < nev; ++i) evals[i] = ks.H(i, i);  // src/sparseMatrix.cpp:1212-1213
  if (nconv) *nconv = (int)nc;
  if (nconv_ref_rule) *nconv_ref_rule = (int)nc_ref;
  if (restarts) *restarts = (int)ks.n_restarts;
  if (napplies) *napplies = (int)ks.napplies;
  return rc;
}

extern "C" int isle_hip_block_ks(isle_ctx* c, int nev, int ncv, int maxit, int blk, float tol, uint64_t seed, float* evals, int* nconv,
                                 int* restarts, int* napplies) {
  if (!c || !evals) return ISLE_E_ARG;
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "no matrix uploaded");
  ISLECHK(isle_enter(c));
  Ks ks;
  ks.dim = c->V;
  c->band_ready = false;  // the operator (CSR copy) is rebuilt per solve, as in src/sparseMatrix.cpp:1199
  int nc = 0;
  const int rc = ks_solve(c, ks, nev, ncv, maxit, blk, tol, seed, evals, &nc, restarts, napplies, nullptr);
  if (nconv) *nconv = nc;
  if (rc != 0 && rc != ISLE_E_NOCONV) return rc;
  ISLECHK(install_U(c, c->basis.p, nev));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  isle_host_mark("block_ks: U installed, exit");
  if (rc == ISLE_E_NOCONV) return isle_fail(c, rc, "block KS: %d restarts exhausted, %d of %d Ritz pairs converged", maxit, nc, nev);
  return 0;
}

extern "C" int isle_hip_block_ks_dense(isle_ctx* c, const float* A, uint64_t n, int nev, int ncv, int maxit, int blk, float tol,
                                       uint64_t seed, const float* start_block, float* evals, float* U, int* nconv, int* nconv_ref_rule,
                                       int* restarts, int* napplies) {
  if (!c || !A || !evals || n < 2 || n > 46340) return isle_fail(c, ISLE_E_ARG, "block_ks_dense: bad arguments (2 <= n <= 46340)");
  ISLECHK(isle_enter(c));
  if (c->multi()) return isle_fail(c, ISLE_E_ARG, "block_ks_dense: the dense operator is not sharded (single rank only)");
  const int b_eff = (blk < nev) ? blk : 1;
  DevBuf<float> Adev, Sdev;
  HIPCHK(c, Adev.reserve((size_t)n * n));
  HIPCHK(c, hipMemcpy(Adev.p, A, (size_t)n * n * sizeof(float), hipMemcpyHostToDevice));
  Ks ks;
  ks.dim = n;
  ks.dense_A = Adev.p;
  if (start_block && b_eff >= 1) {
    HIPCHK(c, Sdev.reserve((size_t)n * b_eff));
    HIPCHK(c, hipMemcpy(Sdev.p, start_block, (size_t)n * b_eff * sizeof(float), hipMemcpyHostToDevice));
    ks.start_dev = Sdev.p;
  }
  int nc = 0;
  const int rc = ks_solve(c, ks, nev, ncv, maxit, blk, tol, seed, evals, &nc, restarts, napplies, nconv_ref_rule);
  if (nconv) *nconv = nc;
  hipError_t he = hipStreamSynchronize(c->stream);
  if (he == hipSuccess && (rc == 0 || rc == ISLE_E_NOCONV) && U)
    he = hipMemcpy(U, c->basis.p, (size_t)n * nev * sizeof(float), hipMemcpyDeviceToHost);
  HIPCHK(c, he);
  if (rc == ISLE_E_NOCONV) return isle_fail(c, rc, "block KS (dense operator): %d restarts exhausted, %d of %d Ritz pairs converged", maxit, nc, nev);
  return rc;
}

extern "C" int isle_hip_get_U(isle_ctx* c, float* U) {
  if (!c || !U || c->U_k == 0) return isle_fail(c, ISLE_E_ARG, "no U available");
  ISLECHK(isle_enter(c));
  HIPCHK(c, hipMemcpyAsync(U, c->Ucm.p, (size_t)c->V * c->U_k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
extern "C" int isle_hip_set_U(isle_ctx* c, const float* U, int k) {
  if (!c || !U || k < 1 || c->V == 0) return isle_fail(c, ISLE_E_ARG, "set_U: bad arguments");
  ISLECHK(isle_enter(c));
  HIPCHK(c, c->Ucm.reserve((size_t)c->V * k));
  HIPCHK(c, hipMemcpy(c->Ucm.p, U, (size_t)c->V * k * sizeof(float), hipMemcpyHostToDevice));
  ISLECHK(install_U(c, c->Ucm.p, k));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int isle_hip_eig_sym(isle_ctx* c, const float* S, int n, float* evals, float* vecs) {
  if (!c || !S || !evals || !vecs || n < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  HIPCHK(c, c->Wf.reserve((size_t)n * n));
  ISLECHK(k_eig_small(c, S, n, evals, c->Wf.p, n));
  HIPCHK(c, hipMemcpy(vecs, c->Wf.p, (size_t)n * n * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}


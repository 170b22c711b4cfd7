// isle_amd/csrc/api_kmeans.cpp — k-means++ in span(U), Lloyd in span(U), the lift and Lloyd on B behind the C ABI
// (kmeanspp_on_projected_space src/sparseMatrix.cpp:2133-2209, run_lloyds_on_projected_space :2016-2072, left_multiply_by_U_Spectra
// :1438-1450, run_lloyds :1690-1746): the reference's draw schedule, stop rules and iteration structure on the host, the arithmetic in
// kmeans.hip / spmm.hip / dense.hip / gram_lds.hip.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "api_internal.h"

#ifndef ISLE_PROJ_FULL_NUM
// Lloyd in span(U): a full pass instead of the active rows' products when more than NUM / DEN of the documents are active.  A third since round 5
// (a half before): at config 3 the loop takes 187 ms with 1/2, 166 with 1/3 and 1/4, 199 with 1/6, 214 with 1/10 (the full product costs
// 5.5 ns a document, the compaction + tile products of the active rows 15 - 20 ns an active document, and a full pass refreshes every bound)
#define ISLE_PROJ_FULL_NUM 1
#define ISLE_PROJ_FULL_DEN 3
#endif
// ------------------------------------------------------------------------------------------
// k-means in the projected space
// ------------------------------------------------------------------------------------------
// The movers of an iteration (YyMovers): up to ten centres whose movement stands out — more than twice the eleventh largest — and the
// groups' (Yinyang: 8 centres) or tiles' (projected loop: 32) largest movements WITHOUT them.  The same on every rank (replicated inputs).
static void choose_movers(const std::vector<float>& delta, int k, int group, int G, YyMovers* mv, std::vector<float>* gmax_excl,
                          const std::vector<uint32_t>* slot_of_id = nullptr /*regrouped Yinyang groups: centre i sits in group slot_of_id[i] / group*/) {
  mv->n = 0;
  if ((int)delta.size() != k || k <= 16) return;
  std::vector<int> ord(k);
  std::iota(ord.begin(), ord.end(), 0);
  std::partial_sort(ord.begin(), ord.begin() + 11, ord.end(), [&](int a, int b) { return delta[a] > delta[b] || (delta[a] == delta[b] && a < b); });
  const float ref = delta[ord[10]];
  for (int j = 0; j < 10; ++j)
    if (delta[ord[j]] > 2.0f * ref && delta[ord[j]] > 1e-4f) mv->id[mv->n++] = (uint32_t)ord[j];
  if (!mv->n) return;
  mv->ld = 4 * ((mv->n + 3) / 4);
  gmax_excl->assign(G, 0.f);
  for (int i = 0; i < k; ++i) {
    bool is_mover = false;
    for (int j = 0; j < mv->n; ++j) is_mover = is_mover || mv->id[j] == (uint32_t)i;
    const int gi = (int)(slot_of_id ? (*slot_of_id)[i] : (uint32_t)i) / group;
    if (!is_mover) (*gmax_excl)[gi] = std::max((*gmax_excl)[gi], delta[i]);
  }
}

static int ensure_P(isle_ctx* c, int k) {
  if (c->U_k != k) return isle_fail(c, ISLE_E_ARG, "U has %d columns, k = %d (run isle_hip_block_ks / set_U first)", c->U_k, k);
  if (c->P_ready) return 0;
  const size_t D = c->D ? c->D : 1;
  HIPCHK(c, c->P.reserve(D * c->ldk));
  HIPCHK(c, c->pnorm.reserve(D));
  c->Pt_ready = false;
  c->Pt2_ready = false;
  c->Pt2_pos = false;
  // Where the assignment products run with their epilogues inside (a large shard) they read the projection's two bf16 terms in the layout
  // the LDS-DMA product stages (gemm_bf16x2_dma_k).  The grouped projection writes that copy on its way, by POSITION (the products map their
  // rows to documents through dperm): no transposition and no split pass behind the projection (round 5: 15 + 15 ms at config 3).
  const bool want_a2 = c->D && k_gemm_assign_fused_ok(c, c->D, k, k) && !c->knob_zero(KN_GEMM_DMA) &&
                       !(c->knob(KN_GEMM_TERMS) && atoi(c->knob(KN_GEMM_TERMS)) == 3);
  bool a2_done = false;
  if (want_a2) HIPCHK(c, c->Pt2.reserve(k_gemm_split_a_bytes(c->D, k) / sizeof(uint4)));
  ISLECHK(k_spmm_wide_project(c, c->Urm.p, k, c->ldk, c->P.p, c->pnorm.p, want_a2 ? c->Pt2.p : nullptr, &a2_done));
  c->P_ready = true;
  c->P_gen++;
  if (a2_done) {
    c->Pt2_ready = true;
    c->Pt2_pos = true;  // the f32 coordinate-major copy is made when a route asks for it (k_ensure_pt)
  } else if (c->D) {
    TimeScope ts(c, ISLE_T_PROJECT);
    ISLECHK(k_ensure_pt(c));
    if (want_a2) {  // the projection took another route than the grouped one: split the transposed copy
      ISLECHK(k_gemm_split_a(c, c->Pt.p, c->D, k, c->Pt2.p));
      c->Pt2_ready = true;
    }
  }
  return 0;
}

// dst (n x ldk, device) <- P rows of the given GLOBAL doc ids (owner contributes, others zero, then all-reduce)
static int fetch_rows(isle_ctx* c, const uint64_t* ids, int n, float* dst) {
  if (n == 0) return 0;
  const bool multi = c->multi();
  std::vector<uint64_t> local(n);
  for (int i = 0; i < n; ++i) {
    const uint64_t g = ids[i];
    if (g >= c->doc_offset && g < c->doc_offset + c->D) local[i] = g - c->doc_offset;
    else if (multi) local[i] = ~0ull;  // another rank's document: zeros here, the all-reduce brings the row
    else return isle_fail(c, ISLE_E_ARG, "seed doc id %llu out of range", (unsigned long long)g);
  }
  ISLECHK(k_fetch_rows(c, c->P.p, c->ldk, local.data(), n, dst));  // one kernel (the ids travel as arguments), not one copy per row
  if (multi) ISLECHK(allreduce_sum<float>(c, dst, (size_t)n * c->ldk));
  return 0;
}

extern "C" int isle_hip_kmeanspp_projected(isle_ctx* c, int k, const uint64_t* inject, uint64_t rng_seed, uint64_t* seeds_out,
                                           float* C_lowd, float* residual, int* rounds_out) {
  if (!c || !seeds_out || !C_lowd || k < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if ((uint64_t)k > c->D_global) return isle_fail(c, ISLE_E_ARG, "k > number of documents");
  isle_host_mark("kmeanspp: entry");
  ISLECHK(ensure_P(c, k));  // compute_projected_docs_l2sq :2144
  isle_host_mark("kmeanspp: projection enqueued");
  const uint64_t D = c->D, Dg = c->D_global;
  const int ldk = c->ldk;
  const bool multi = c->multi();
  HIPCHK(c, c->min_dist.reserve(D ? D : 1));
  HIPCHK(c, c->cum.reserve(D + 1));
  HIPCHK(c, c->Cdev.reserve((size_t)k * ldk));
  HIPCHK(c, c->gram.reserve(1024));
  HIPCHK(c, c->small.reserve(4096));
  ISLECHK(k_fill_f32(c, c->min_dist.p, D, 3.402823466e+38f));  // FP_MAX :2148
  HostRng rng(rng_seed);
  std::vector<uint64_t> centers;
  const uint64_t first = inject ? inject[0] : (uint64_t)(((size_t)rng.next31() * (size_t)84619573) % (size_t)Dg);  // :2150
  centers.push_back(first);
  ISLECHK(fetch_rows(c, &first, 1, c->Cdev.p));
  int new_added = 1, rounds = 0;
  double grand = 0.0, last_md = 0.0;
  // k > 224 (Lloyd in span(U) keeps tile bounds): the rounds also keep every document's nearest seed and tile minima, so that Lloyd's
  // first assignment — a D x k x k pass against exactly these seeds — need not be computed again (kmeans.hip kmpp_min_dots_track_k)
  const bool track = k > 224 && (k + 31) / 32 <= 32 && !c->knob_zero(KN_KMPP_TRACK);
  c->kmpp_track_k = 0;
  const int maxdraw = 2 + (int)std::ceil(std::sqrt((double)k));
  std::vector<double> dice(maxdraw);
  // page-locked staging for the per-round scalars: [my 2 | tot 2 * world | local maxdraw] doubles, then drawn maxdraw u64
  double* pin_d = reinterpret_cast<double*>(c->pin + isle_ctx::PIN_SMALL);
  if ((size_t)(2 + 2 * c->world + 2 * maxdraw + 42) * 8 > (128u << 10)) return isle_fail(c, ISLE_E_ARG, "k-means++: staging area too small");
  double* my = pin_d;
  double* tot = pin_d + 2;
  double* local = tot + 2 * c->world;
  uint64_t* drawn = reinterpret_cast<uint64_t*>(local + maxdraw);
  while ((int)centers.size() < k) {
    rounds++;
    ISLECHK(k_kmpp_update(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p + (centers.size() - new_added) * (size_t)ldk, new_added,
                          c->min_dist.p, (int)(centers.size() - new_added), track));
    ISLECHK(k_scan_f2d(c, c->min_dist.p, D, c->cum.p));  // :2170-2172 (double, parallel; the reference's is fp32 sequential)
    const int s = (int)centers.size();
    int ndraw = 0;
    for (int cc = 0; cc < 1 + std::sqrt((double)(s - 5 > 0 ? s - 5 : 0)); ++cc) ndraw++;  // :2183 (upper bound on draws)
    ndraw = std::min(ndraw, maxdraw);
    if (!multi && !inject && ndraw <= 40 && !c->knob_on(KN_KMPP_HOST_DICE)) {  // the switch: for the test that holds both forms to the same seeds
      // one rank: the dice are products of the total with host-drawn fractions, so the device can throw them itself — the totals, the
      // dice and their search come back in one copy (search_frac_k), one host round trip per round
      for (int i = 0; i < ndraw; ++i) dice[i] = rng.fraction();  // :2184
      uint64_t* res = drawn + maxdraw;  // page-locked, 42 entries
      // the kernel writes its 42 words straight into the page-locked area (host memory mapped into the device's address space): no
      // copy kernel, and one gap less, between the search and the host's wake-up
      ISLECHK(k_search_frac(c, c->cum.p, D, D > 0 ? c->min_dist.p + (D - 1) : nullptr, dice.data(), ndraw, res));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      memcpy(my, res + 40, 2 * sizeof(double));
      grand = my[0];
      last_md = my[1];
      for (int i = 0; i < ndraw; ++i) drawn[i] = std::min<uint64_t>(res[i], D - 1) + c->doc_offset;
    } else {
      // totals (per rank) -> offsets
      my[0] = my[1] = 0.0;
      ISLECHK(k_pack2(c, c->cum.p + D, D > 0 ? c->min_dist.p + (D - 1) : nullptr, c->gram.p + 200));
      HIPCHK(c, hipMemcpyAsync(my, c->gram.p + 200, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));  // one copy, one round trip for both scalars
      for (int r = 0; r < 2 * c->world; ++r) tot[r] = 0.0;
      if (multi) {
        double* dv = c->gram.p;
        HIPCHK(c, hipMemcpyAsync(dv + 2 * c->world, my, 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        {
          TimeScope ts(c, ISLE_T_COMM);
          ISLECHK(isle_allgather(c, dv + 2 * c->world, dv, 2, ISLE_DT_F64));
        }
        HIPCHK(c, hipMemcpyAsync(tot, dv, 2 * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
      } else {
        tot[0] = my[0];
        tot[1] = my[1];
      }
      grand = 0.0;
      double my_off = 0.0;
      for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) my_off = grand;
        grand += tot[2 * r];
      }
      last_md = tot[2 * (c->world - 1) + 1];
      if (!inject) {
        // all ranks draw the same dice; the owner of the interval searches its local prefix sums
        for (int i = 0; i < ndraw; ++i) {
          dice[i] = grand * rng.fraction();  // :2184
          const double x = dice[i] - my_off;
          const bool mine = (x >= 0.0 && x < my[0]) || (c->world == 1);
          local[i] = mine ? std::min(std::max(x, 0.0), my[0]) : -1.0;
        }
        double* dd = c->gram.p + 64;
        uint64_t* od = (uint64_t*)(c->gram.p + 128);
        if (ndraw <= 16) {
          ISLECHK(k_search_args(c, c->cum.p, D, local, ndraw, od));  // dice as kernel arguments
        } else {
          HIPCHK(c, hipMemcpyAsync(dd, local, ndraw * sizeof(double), hipMemcpyHostToDevice, c->stream));  // `local` outlives the sync below
          ISLECHK(k_search(c, c->cum.p, D, dd, ndraw, od));
        }
        HIPCHK(c, hipMemcpyAsync(drawn, od, ndraw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < ndraw; ++i) {
          if (local[i] < 0.0 || D == 0) drawn[i] = 0;
          else drawn[i] = std::min<uint64_t>(drawn[i], D - 1) + c->doc_offset + 1;  // +1: zero means "not mine"
        }
        if (multi) {
          HIPCHK(c, hipMemcpyAsync(od, drawn, ndraw * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
          ISLECHK(allreduce_sum<uint64_t>(c, od, ndraw));
          HIPCHK(c, hipMemcpyAsync(drawn, od, ndraw * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        for (int i = 0; i < ndraw; ++i) drawn[i] = drawn[i] ? drawn[i] - 1 : 0;
      }
    }
    new_added = 0;
    std::vector<uint64_t> fresh;
    for (int cc = 0; cc < ndraw && (int)centers.size() < k; ++cc) {
      const uint64_t nc = inject ? inject[centers.size()] : drawn[cc];
      if (std::find(centers.begin(), centers.end(), nc) == centers.end()) {  // duplicates skipped, not redrawn :2189
        centers.push_back(nc);
        fresh.push_back(nc);
        new_added++;
      }
    }
    if (new_added) ISLECHK(fetch_rows(c, fresh.data(), new_added, c->Cdev.p + (centers.size() - new_added) * (size_t)ldk));
    if (inject && new_added == 0) return isle_fail(c, ISLE_E_ARG, "injected seeds contain duplicates");
    if (rounds > 100 * k) return isle_fail(c, ISLE_E_NUMERIC, "k-means++ cannot find %d distinct seeds", k);
  }
  isle_host_mark("kmeanspp: rounds done");
  // the last batch of seeds is never folded into min_dist (the loop ends when the k-th seed is drawn, :2163-2207); for Lloyd's first
  // assignment it is folded into a COPY of the distances
  if (track && c->kmpp_track && new_added > 0 && c->kmpp_track_seeds == k - new_added && D > 0) {
    HIPCHK(c, c->kmpp_best.reserve(D));
    HIPCHK(c, hipMemcpyAsync(c->kmpp_best.p, c->min_dist.p, D * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    ISLECHK(k_kmpp_update(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p + (size_t)(k - new_added) * ldk, new_added, c->kmpp_best.p, k - new_added, true));
  }
  // best_centers_coords[c] = U^T b_seed[c]  (:2232-2234)
  const size_t ch_bytes = (size_t)k * ldk * sizeof(float);
  HIPCHK(c, c->pin_stage_reserve(ch_bytes));
  const float* Ch = reinterpret_cast<const float*>(c->pin_stage);
  HIPCHK(c, hipMemcpyAsync(c->pin_stage, c->Cdev.p, ch_bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int cc = 0; cc < k; ++cc) {
    seeds_out[cc] = centers[cc];
    memcpy(C_lowd + (size_t)cc * k, Ch + (size_t)cc * ldk, (size_t)k * sizeof(float));
  }
  if (residual) *residual = (float)(grand - last_md);  // dist_cumul[num_docs - 1]  (:2208; App. C #9)
  if (rounds_out) *rounds_out = rounds;
  if (track && c->kmpp_track && c->kmpp_track_seeds == k) {  // complete: Lloyd may start from it if it is handed exactly these centres
    c->kmpp_C_host.assign(C_lowd, C_lowd + (size_t)k * k);
    c->kmpp_P_gen = c->P_gen;
    c->kmpp_track_k = k;
  }
  isle_host_mark("kmeanspp: exit");
  return 0;
}

extern "C" int isle_hip_get_min_dist(isle_ctx* c, float* out) {
  if (!c || !out) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->D) HIPCHK(c, hipMemcpy(out, c->min_dist.p, c->D * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

// The reference's stop rule (src/sparseMatrix.cpp:2044-2064 / :1718-1738): converged when the cluster
// sizes equal the previous iteration's AND the partition equals the last partition stored on an
// iteration whose sizes matched.
namespace {
struct StopRule {
  isle_ctx* c;
  int k;
  std::vector<long long> prev_sizes;
  bool have_prev = false;
  StopRule(isle_ctx* c_, int k_) : c(c_), k(k_), prev_sizes(k_, 0) {}
  // sizes: GLOBAL cluster sizes of this iteration.  assign: device, local docs.
  int converged(const std::vector<long long>& sizes, const uint32_t* assign, bool* out) {
    bool changed = false;
    for (int i = 0; i < k; ++i)
      if (prev_sizes[i] != sizes[i]) changed = true;
    prev_sizes = sizes;
    if (!changed) {
      if (!have_prev) {
        changed = c->D_global > 0;  // prev_closest_docs are k empty lists
      } else {
        HIPCHK(c, c->flags.reserve(16));
        ISLECHK(k_compare_u32(c, assign, c->assign_prev.p, c->D, c->flags.p));
        ISLECHK(allreduce_sum<int>(c, c->flags.p, 1));
        int* f = reinterpret_cast<int*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10));  // page-locked
        HIPCHK(c, hipMemcpyAsync(f, c->flags.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        changed = *f != 0;
      }
      HIPCHK(c, c->assign_prev.reserve(c->D ? c->D : 1));
      if (c->D) HIPCHK(c, hipMemcpyAsync(c->assign_prev.p, assign, c->D * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
      have_prev = true;
    }
    *out = !changed;
    return 0;
  }
};
}  // namespace

static int fetch_sizes(isle_ctx* c, int k, std::vector<long long>& sizes) {
  ISLECHK(allreduce_sum<int>(c, c->counts.p, k));
  std::vector<int> hv;
  int* h = reinterpret_cast<int*>(c->pin + isle_ctx::PIN_SMALL + (128u << 10));  // page-locked, 64 KB
  if ((size_t)k * sizeof(int) > (64u << 10)) {
    hv.resize(k);
    h = hv.data();
  }
  HIPCHK(c, hipMemcpyAsync(h, c->counts.p, (size_t)k * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  sizes.assign(h, h + k);
  return 0;
}

// The k centre movements of the last update, on the host for choose_movers: through the page-locked area and behind a stream
// synchronisation of its own (the stop rule synchronises only when the cluster sizes stayed the same)
static int fetch_delta(isle_ctx* c, const float* delta_dev, int k, std::vector<float>& out) {
  out.resize(k);
  float* h = reinterpret_cast<float*>(c->pin + isle_ctx::PIN_SMALL + (224u << 10));  // page-locked, 32 KB
  if ((size_t)k * sizeof(float) > (32u << 10)) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out.data(), delta_dev, (size_t)k * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  HIPCHK(c, hipMemcpyAsync(h, delta_dev, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(out.data(), h, (size_t)k * sizeof(float));
  return 0;
}

extern "C" int isle_hip_lloyds_projected(isle_ctx* c, int k, float* C_lowd, int max_reps, int* iters_run, uint32_t* assign_out) {
  if (!c || !C_lowd || k < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  isle_host_mark("lloyds_projected: entry");
  ISLECHK(ensure_P(c, k));  // compute_projected_docs_l2sq :2032
  const uint64_t D = c->D;
  const int ldk = c->ldk;
  HIPCHK(c, c->Cdev.reserve((size_t)k * ldk));
  HIPCHK(c, c->Csum.reserve((size_t)k * ldk));
  HIPCHK(c, c->cnorm.reserve(k));
  HIPCHK(c, c->counts.reserve(k));
  HIPCHK(c, c->assign.reserve(D ? D : 1));
  c->assign_valid = false;
  const size_t ch_bytes = (size_t)k * ldk * sizeof(float);
  HIPCHK(c, c->pin_stage_reserve(ch_bytes));
  float* Ch = reinterpret_cast<float*>(c->pin_stage);
  if (ldk != k) memset(Ch, 0, ch_bytes);
  for (int cc = 0; cc < k; ++cc) memcpy(Ch + (size_t)cc * ldk, C_lowd + (size_t)cc * k, (size_t)k * sizeof(float));
  HIPCHK(c, hipMemcpyAsync(c->Cdev.p, Ch, ch_bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the staging buffer is written again at the end of this call
  isle_host_mark("lloyds_projected: centres uploaded");
  // Hamerly bounds (exact skip of documents whose closest centre provably did not change), as in the sparse Lloyd
  const bool hamerly = !c->knob_on(KN_NO_HAMERLY) && (c->Pt_ready || c->Pt2_ready);
  if (hamerly) {
    HIPCHK(c, c->hub.reserve(D ? D : 1));
    HIPCHK(c, c->hlb.reserve(D ? D : 1));
    HIPCHK(c, c->active.reserve(D + 1));
    HIPCHK(c, c->Pa.reserve((size_t)(D ? D : 1) * ldk));
    HIPCHK(c, c->pna.reserve(D ? D : 1));
    HIPCHK(c, c->Cold.reserve((size_t)k * ldk + k + 8));
  }
  float* delta_dev = hamerly ? c->Cold.p + (size_t)k * ldk : nullptr;
  HamTop* top_dev = hamerly ? reinterpret_cast<HamTop*>(c->Cold.p + (size_t)k * ldk + ((k + 3) & ~3)) : nullptr;
  // k > 224 (more than 7 tiles of 32 centres): one lower bound per tile instead of Hamerly's single one, which prunes nothing at
  // k = 1000 (kmeans.hip PR_TILES, spmm.hip pt_filter_k).  ISLE_PROJ_BOUNDS=hamerly keeps the single bound.
  const int T = (k + 31) / 32, TL = (T + 3) & ~3;
  const char* pbm = c->knob(KN_PROJ_BOUNDS);
  const bool tiles = hamerly && k > 224 && T <= 32 && !(pbm && !strcmp(pbm, "hamerly"));
  float* tmove_dev = nullptr;
  if (tiles) {
    HIPCHK(c, c->ptlb.reserve((size_t)(D ? D : 1) * TL));
    HIPCHK(c, c->pneed.reserve(D ? D : 1));
    HIPCHK(c, c->pcand.reserve(D + 1));
    HIPCHK(c, c->small.reserve(4096));
    tmove_dev = c->small.p;  // T floats
  }
  const bool from_kmpp = tiles && c->kmpp_track_k == k && c->kmpp_P_gen == c->P_gen && c->P_ready && !c->knob_zero(KN_KMPP_TRACK) &&
                         c->kmpp_C_host.size() == (size_t)k * k && memcmp(c->kmpp_C_host.data(), C_lowd, (size_t)k * k * sizeof(float)) == 0;
  c->kmpp_track_k = 0;  // used (the tile minima become bounds in place) or stale
  if (c->knob_on(KN_DEBUG_HAMERLY)) fprintf(stderr, "[projected Lloyd] first assignment %s\n", from_kmpp ? "taken from the k-means++ rounds" : "computed");
  StopRule stop(c, k);
  int it = 0;
  const bool pt_sorted = !c->knob_zero(KN_PT_SORT);  // the active documents are ordered by the tiles they need: the filter walks the documents in their own order
  const char* psm = c->knob(KN_PROJ_SUMS);
  const bool delta_sums = !(psm && !strcmp(psm, "fresh")) && D < (1ull << 31);
  if (delta_sums) HIPCHK(c, c->proj_counted.reserve(D ? D : 1));
  if (c->multi()) HIPCHK(c, c->Csum_local.reserve((size_t)k * ldk));
  std::vector<float> pdelta_host;  // the k centre movements of the last update (tile bounds: choice of the movers)
  isle_host_mark("lloyds_projected: loop starts");
  for (; it < max_reps; ++it) {
    ISLECHK(k_rownorms(c, c->Cdev.p, k, k, ldk, c->cnorm.p));                                          // :1938
    if (tiles) {
      if (it == 0 && from_kmpp) {
        // the centres are the k-means++ seeds and the rounds kept every document's nearest seed, its tile's runner-up and the minimum of
        // every other tile: exactly this assignment (up to the rounding of the two distance evaluations, inside the bounds' slack)
        ISLECHK(k_kmpp_to_tiles(c, D, k, c->pnorm.p, c->cnorm.p, c->kmpp_best.p, c->assign.p, c->hub.p, TL));
      } else if (it == 0) {
        ISLECHK(k_proj_assign_tiles(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->hub.p, c->ptlb.p, TL, nullptr, 0,
                                    nullptr, nullptr, nullptr));                                           // :1947
      } else {
        uint32_t* nact = c->active.p + D;
        // documents are taken grouped by their centre (member lists of the previous iteration): a workgroup of the re-examination
        // then holds neighbours, whose needed tiles coincide
        {  // candidates by the grown upper bounds, then the exact distance to the own centre for those (pt_tighten_k)
          uint32_t* ncand = c->pcand.p + D;
          // movers (see Lloyd on B below): a tile of 32 centres loses its bound to ONE centre that jumped; up to ten such centres are left out
          // of their tiles' movements and bounded by their exact new distances  P_d . c = b_d^T (U c): a thin product of B with U C_m^T
          // (the k-means++ rounds' route), one pass of the pass-1 stream
          YyMovers mv;
          const float* tmove_use = tmove_dev;
          if (!c->knob_zero(KN_YY_MOVERS) && c->gl_mode == 1 && c->band_ready && c->U_k == k && D) {
            std::vector<float> tm;
            choose_movers(pdelta_host, k, 32, T, &mv, &tm);
            if (mv.n) {
              HIPCHK(c, c->yy_gmax2.reserve(std::max(T, 64)));
              HIPCHK(c, hipMemcpyAsync(c->yy_gmax2.p, tm.data(), (size_t)T * sizeof(float), hipMemcpyHostToDevice, c->stream));
              HIPCHK(c, c->Tmp.reserve((size_t)c->V * 32 + (size_t)16 * ldk));
              float* Cm = c->Tmp.p + (size_t)c->V * 32;  // the movers' centres, one row each
              for (int j = 0; j < mv.n; ++j)
                HIPCHK(c, hipMemcpyAsync(Cm + (size_t)j * ldk, c->Cdev.p + (size_t)mv.id[j] * ldk, (size_t)ldk * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
              HIPCHK(c, hipStreamSynchronize(c->stream));  // tm is stack-owned
              HIPCHK(c, c->yy_mdots.reserve((size_t)D * mv.ld));
              ISLECHK(k_gemm_nn(c, c->Ucm.p, c->V, k, Cm, ldk, mv.n, c->Tmp.p, ISLE_T_LLOYD_PROJ));  // W = U C_m^T  (V x n col-major)
              {
                TimeScope ts(c, ISLE_T_LLOYD_PROJ);
                ISLECHK(k_gl_thin(c, c->Tmp.p, mv.n, mv.ld, c->yy_mdots.p, true));  // rows by position: pt_filter_k reads them through dpos
              }
              tmove_use = c->yy_gmax2.p;
            }
          }
          ISLECHK(k_pt_filter(c, pt_sorted ? nullptr : (c->members_valid ? c->members.p : nullptr), c->assign.p, c->hub.p, c->ptlb.p, T, TL, delta_dev, tmove_use,
                              c->pneed.p, c->pcand.p, ncand, mv, c->yy_mdots.p, c->cnorm.p, c->pnorm.p));
          ISLECHK(k_pt_tighten(c, c->P.p, c->pnorm.p, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->pcand.p, ncand, c->hub.p, c->ptlb.p, T, TL,
                               c->pneed.p, c->active.p, nact));
        }
        uint32_t* na_pin = reinterpret_cast<uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 64);  // page-locked
        HIPCHK(c, hipMemcpyAsync(na_pin, nact, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const uint32_t na = *na_pin;
        if (c->knob_on(KN_DEBUG_HAMERLY)) {  // debug only: how many tiles the active documents ask for
          std::vector<uint32_t> act(na), need(D);
          if (na) HIPCHK(c, hipMemcpy(act.data(), c->active.p, na * sizeof(uint32_t), hipMemcpyDeviceToHost));
          if (D) HIPCHK(c, hipMemcpy(need.data(), c->pneed.p, D * sizeof(uint32_t), hipMemcpyDeviceToHost));
          if (na > 256 && !c->knob_zero(KN_PT_SORT))  // the order k_proj_assign_tiles gives the list
            std::stable_sort(act.begin(), act.end(), [&](uint32_t a, uint32_t b) { return need[a] < need[b]; });
          double tiles_sum = 0, union_sum = 0;
          for (uint32_t i = 0; i < na; i += 128) {
            uint32_t u = 0;
            for (uint32_t j = i; j < std::min(na, i + 128); ++j) {
              tiles_sum += __builtin_popcount(need[act[j]]);
              u |= need[act[j]];
            }
            union_sum += __builtin_popcount(u);
          }
          fprintf(stderr, "[tile bounds, projected] iter %d active %u of %llu, tiles per document %.1f, per workgroup (union) %.1f of %d\n", it, na,
                  (unsigned long long)D, na ? tiles_sum / na : 0.0, na ? union_sum / ((na + 127) / 128) : 0.0, T);
        }
        // (measured with the product on the gathered rows of the active documents: handing the full pass over only beyond 3/4 of the documents
        // is slower, 222 against 187 ms at config 3 — gathering 5.4 M rows costs what the product saves, and the full pass refreshes every bound)
        if ((uint64_t)na * ISLE_PROJ_FULL_DEN > (uint64_t)D * ISLE_PROJ_FULL_NUM && k_proj_full_by_gemm(c, D, k))  // most documents are up for re-examination: the full GEMM pass costs less than
          ISLECHK(k_proj_assign_tiles(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->hub.p, c->ptlb.p, TL, nullptr, 0,
                                      nullptr, nullptr, nullptr));  // compacting them and walking their tiles, and refreshes every bound
        else
          ISLECHK(k_proj_assign_tiles(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p, c->hub.p, c->ptlb.p, TL, c->active.p, na,
                                      c->pneed.p, c->Pa.p, c->pna.p));
      }
    } else if (it == 0 || !hamerly) {
      ISLECHK(k_proj_assign(c, c->P.p, c->pnorm.p, D, k, ldk, c->Cdev.p, c->cnorm.p, c->assign.p,
                            hamerly ? c->hub.p : nullptr, hamerly ? c->hlb.p : nullptr));                // :1947
    } else {
      uint32_t* nact = c->active.p + D;
      ISLECHK(k_hamerly_filter(c, nullptr, c->assign.p, c->hub.p, c->hlb.p, delta_dev, top_dev, c->active.p, nact, ISLE_T_LLOYD_PROJ));
      uint32_t* na_pin = reinterpret_cast<uint32_t*>(c->pin + isle_ctx::PIN_SMALL + (192u << 10) + 64);  // page-locked
      HIPCHK(c, hipMemcpyAsync(na_pin, nact, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const uint32_t na = *na_pin;
      if (c->knob_on(KN_DEBUG_HAMERLY)) fprintf(stderr, "[hamerly, projected] iter %d active %u of %llu\n", it, na, (unsigned long long)D);
      ISLECHK(k_proj_assign_active(c, c->P.p, c->pnorm.p, k, ldk, c->Cdev.p, c->cnorm.p, c->active.p, na, c->Pa.p, c->pna.p, c->assign.p,
                                   c->hub.p, c->hlb.p));
    }
    ISLECHK(k_count_sizes(c, c->assign.p, D, k, c->counts.p));
    {
      // sums of the members' rows (:1957-1984): afresh in the first iteration, afterwards from the documents that changed centre
      float* sums = c->multi() ? c->Csum_local.p : c->Csum.p;  // this rank's sums
      bool updated = false;
      if (delta_sums && it > 0) ISLECHK(k_proj_accumulate_delta(c, c->P.p, D, k, ldk, c->assign.p, c->proj_counted.p, sums, c->counts.p, &updated));
      if (!updated) {
        ISLECHK(k_proj_accumulate(c, c->P.p, D, k, ldk, c->assign.p, sums, c->counts.p));
        if (delta_sums && D) HIPCHK(c, hipMemcpyAsync(c->proj_counted.p, c->assign.p, D * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
      } else {
        c->members_valid = false;  // the lists are those of an earlier assignment
      }
      if (c->multi()) HIPCHK(c, hipMemcpyAsync(c->Csum.p, sums, (size_t)k * ldk * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    ISLECHK(allreduce_sum<float>(c, c->Csum.p, (size_t)k * ldk));
    std::vector<long long> sizes;
    ISLECHK(fetch_sizes(c, k, sizes));
    if (hamerly) HIPCHK(c, hipMemcpyAsync(c->Cold.p, c->Cdev.p, (size_t)k * ldk * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    ISLECHK(k_proj_finalize(c, c->Csum.p, c->counts.p, k, ldk, c->Cdev.p));                             // :1988-1992
    if (hamerly && it + 1 < max_reps) {
      ISLECHK(k_rownorms_diff(c, c->Cdev.p, c->Cold.p, k, k, ldk, delta_dev));
      if (tiles) {
        ISLECHK(k_yy_delta(c, delta_dev, k, T, 32, tmove_dev));  // rounded-up movements and their maxima per tile
        ISLECHK(fetch_delta(c, delta_dev, k, pdelta_host));  // ... and a copy for the choice of the movers
      } else {
        ISLECHK(k_ham_delta(c, delta_dev, k, top_dev));  // rounded-up movements and their top two, on the device
      }
    }
    bool conv = false;
    ISLECHK(stop.converged(sizes, c->assign.p, &conv));
    if (conv) {
      ++it;
      break;
    }
  }
  isle_host_mark("lloyds_projected: loop done");
  HIPCHK(c, hipMemcpyAsync(Ch, c->Cdev.p, ch_bytes, hipMemcpyDeviceToHost, c->stream));
  if (assign_out && D) HIPCHK(c, hipMemcpyAsync(assign_out, c->assign.p, D * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int cc = 0; cc < k; ++cc) memcpy(C_lowd + (size_t)cc * k, Ch + (size_t)cc * ldk, (size_t)k * sizeof(float));
  if (iters_run) *iters_run = it;
  isle_host_mark("lloyds_projected: exit");
  return 0;
}

// ------------------------------------------------------------------------------------------
// lift + Lloyd on the sparse matrix
// ------------------------------------------------------------------------------------------
static int install_centers(isle_ctx* c, int ncols) {  // centers_cm (V x ncols) -> centers_rm (V x ld), zero padded
  const int ld = round4(ncols);
  HIPCHK(c, c->centers_rm.reserve((size_t)c->V * ld));
  HIPCHK(c, hipMemsetAsync(c->centers_rm.p, 0, (size_t)c->V * ld * sizeof(float), c->stream));
  ISLECHK(k_transpose(c, c->centers_cm.p, c->V, ncols, c->V, c->centers_rm.p, ld));
  c->centers_ready = true;
  c->centers_k = ncols;
  return 0;
}

extern "C" int isle_hip_lift_centers(isle_ctx* c, const float* in, int ld_in, int ncols, float* centers) {
  if (!c || !in || ncols < 1) return ISLE_E_ARG;
  ISLECHK(isle_enter(c));
  if (c->U_k == 0 || ld_in < c->U_k) return isle_fail(c, ISLE_E_ARG, "lift: need U and ld_in >= k");
  isle_host_mark("lift: entry");
  HIPCHK(c, c->Csum.reserve((size_t)ld_in * ncols));
  {
    const size_t in_bytes = (size_t)ld_in * ncols * sizeof(float);
    HIPCHK(c, c->pin_stage_reserve(in_bytes));
    memcpy(c->pin_stage, in, in_bytes);
    HIPCHK(c, hipMemcpyAsync(c->Csum.p, c->pin_stage, in_bytes, hipMemcpyHostToDevice, c->stream));  // the call synchronises before it returns
  }
  HIPCHK(c, c->centers_cm.reserve((size_t)c->V * ncols));
  ISLECHK(k_gemm_nn(c, c->Ucm.p, c->V, c->U_k, c->Csum.p, ld_in, ncols, c->centers_cm.p, ISLE_T_LIFT));
  ISLECHK(install_centers(c, ncols));
  // the centres lie in span(U): Lloyd on B can take its first assignment from the projection (isle_hip_lloyds_sparse)
  HIPCHK(c, c->lift_C.reserve((size_t)ld_in * ncols));
  HIPCHK(c, hipMemcpyAsync(c->lift_C.p, c->Csum.p, (size_t)ld_in * ncols * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  c->lift_ld = ld_in;
  c->lift_k = ncols;
  c->lift_valid = true;
  if (centers) HIPCHK(c, hipMemcpyAsync(centers, c->centers_cm.p, (size_t)c->V * ncols * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  isle_host_mark("lift: exit");
  return 0;
}

extern "C" int isle_hip_lloyds_sparse(isle_ctx* c, int k, const float* centers_in, float* centers_out, uint32_t* assign, int max_reps,
                                      int* iters_run) {
  if (!c || k < 1) return ISLE_E_ARG;
  if (c->V == 0) return isle_fail(c, ISLE_E_ARG, "no matrix uploaded");
  ISLECHK(isle_enter(c));
  isle_host_mark("lloyds_sparse: entry");
  const uint64_t D = c->D, V = c->V;
  const int ld = round4(k);
  if (centers_in) {
    HIPCHK(c, c->centers_cm.reserve((size_t)V * k));
    HIPCHK(c, hipMemcpy(c->centers_cm.p, centers_in, (size_t)V * k * sizeof(float), hipMemcpyHostToDevice));
    ISLECHK(install_centers(c, k));
    c->lift_valid = false;
  } else if (!c->centers_ready || c->centers_k != k) {
    return isle_fail(c, ISLE_E_ARG, "lloyds_sparse: no device-resident centres for k = %d (call isle_hip_lift_centers)", k);
  }
  HIPCHK(c, c->dnorm.reserve(D ? D : 1));
  HIPCHK(c, c->cnorm.reserve(k));
  HIPCHK(c, c->counts.reserve(k));
  HIPCHK(c, c->assign.reserve(D ? D : 1));
  c->assign_valid = false;
  ISLECHK(k_doc_norms(c, c->dnorm.p));  // :1680-1687
  // Distance bounds: exact accelerations of the assignment step (documents whose bounds prove "unchanged" are skipped).
  // Default: Yinyang group bounds (groups of 8 centres); ISLE_KMEANS_BOUNDS=hamerly|none selects the others.
  const char* bmode = c->knob(KN_KMEANS_BOUNDS);
  const bool nobounds = c->knob_on(KN_NO_HAMERLY) || (bmode && !strcmp(bmode, "none"));
  const bool hamerly = !nobounds;                                   // any bound-based mode
  const bool yinyang = hamerly && !(bmode && !strcmp(bmode, "hamerly"));
  const int G = (k + 7) / 8;
  int yy_mode_env = -1;  // form of the Yinyang iteration: 0 = by document over the row-major centres, 1 = by document over the group-major copy, 2 = by group
  if (const char* e = c->knob(KN_YY_MODE)) yy_mode_env = !strcmp(e, "doc") ? 0 : !strcmp(e, "docg") ? 1 : !strcmp(e, "group") ? 2 : -1;
  if (yinyang) HIPCHK(c, c->yglb.reserve((size_t)(D ? D : 1) * G + 64));
  float* gmax_dev = nullptr;
  HIPCHK(c, c->hub.reserve(D ? D : 1));
  HIPCHK(c, c->hlb.reserve(D ? D : 1));
  HIPCHK(c, c->active.reserve(D + 1));
  HIPCHK(c, c->centers_old.reserve((size_t)V * ld));
  HIPCHK(c, c->Csum.reserve((size_t)2 * k + 16 + G));
  float* delta_dev = c->Csum.p;  // k floats
  gmax_dev = c->Csum.p + 2 * k + 16;  // G floats
  HamTop* top_dev = reinterpret_cast<HamTop*>(c->Csum.p + 2 * k + 12);
  StopRule stop(c, k);
  // first assignment through the projection: only for centres that came from isle_hip_lift_centers with the current U and P, and
  // while the dense product is cheaper than the sparse one: always up to k = 384; beyond, by the measured rates — the D x k x k
  // product runs at ~130 TFLOP/s (rocBLAS), a panel pass of the sparse product takes ~2.8 ps per nonzero (C3 shard, k = 1000: 19 against
  // 44 ms) — and while its D x k scratch can be had (isle_scratch_ok) (ISLE_FIRST_ASSIGN=sparse|projection forces)
  const char* fa = c->knob(KN_FIRST_ASSIGN);
  const double t_dense = 2.0 * (double)D * k * k / 130e12, t_sparse = (double)((k + 7) / 8) * (double)c->nnz * 2.8e-12;
  const bool fused_first = yinyang && k_gemm_assign_fused_ok(c, D, k, k);  // the product's epilogue forms the assignment: no D x k scratch
  const bool dense_pays = k <= 384 || (t_dense < t_sparse && (fused_first || isle_scratch_ok(c, c->dotsT.cap, (double)D * k * sizeof(float))));
  bool via_projection = !centers_in && c->lift_valid && c->lift_k == k && c->U_k == k && c->P_ready && (c->Pt_ready || c->Pt2_ready) && c->ldk == ld &&
                        D > 0 && (dense_pays || (fa && !strcmp(fa, "projection"))) && !(fa && !strcmp(fa, "sparse"));
  if (via_projection && !fused_first && c->dotsT.reserve((size_t)D * k) != hipSuccess) {
    // the route is chosen from sizes alone (isle_scratch_ok), but on a device shared with other work the D x k scratch may still not be
    // had: the sparse product gives the same assignment up to dot-product rounding, so take it instead of failing the call
    (void)hipGetLastError();
    fprintf(stderr, "[isle_hip] lloyds_sparse: no memory for the %.1f GB product of the first assignment; taking the sparse route\n",
            (double)D * k * sizeof(float) / 1e9);
    via_projection = false;
  }
  c->lift_valid = false;  // the centres move below
  int it = 0;
  // the forms that visit documents in member order (docg, group) hold a document's group bounds four per lane: at most 256 groups
  // (k <= 2048); beyond, by document over the row-major centres, whatever ISLE_YY_MODE asks for
  const int yy_mode = !yinyang ? 0 : G > 256 ? 0 : yy_mode_env >= 0 ? yy_mode_env : (G >= 32 ? 2 : 0);
  // Regrouping (round 5).  A Yinyang group's bound is the distance to its CLOSEST member, so one centre every document is near spoils the
  // bound of its whole group — and the centres of small squared norm (the large, diffuse clusters) are near every document that lies
  // far from everything else: with groups of eight consecutive labels those few centres sit in as many groups and an undecided document
  // scans them all (config 3: 10 - 15 groups per active document in the first iterations).  The groups are therefore formed from the
  // centres in the order of their squared norms at the loop's entry (stable: equal norms keep their labels' order), through slot
  // tables (YyMap); labels, ties and everything outside the by-group kernels stay in the centres' own numbering.  Config 3 on one GPU:
  // 137 M -> 39 M (document, group) pairs per step, Lloyd on B 296 -> 199 ms, same partition (tools/yy_probe.py, profiles/r05_d_*).
  // Taken with the by-group form behind the product's first assignment; ISLE_YY_REGROUP=0 keeps groups of consecutive labels.
  const bool regroup = yinyang && yy_mode == 2 && via_projection && fused_first && !c->knob_zero(KN_YY_REGROUP);
  YyMap ymap;
  std::vector<uint32_t> slot_of_id_host;
  std::vector<float> delta_host;  // the k centre movements of the last update (Yinyang: choice of the movers)
  isle_host_mark("lloyds_sparse: loop starts");
  for (; it < max_reps; ++it) {
    {
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      ISLECHK(k_colnorms_rm(c, c->centers_rm.p, V, k, ld, c->cnorm.p));  // :1604
    }
    if (it == 0 && regroup) {  // the slot tables, from the norms just computed (the same on every rank: the centres are replicated)
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      std::vector<float> cnh(k);
      HIPCHK(c, hipMemcpyAsync(c->pin + isle_ctx::PIN_SMALL + (224u << 10), c->cnorm.p, std::min<size_t>((size_t)k * sizeof(float), 32u << 10), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      memcpy(cnh.data(), c->pin + isle_ctx::PIN_SMALL + (224u << 10), (size_t)k * sizeof(float));  // (k <= 2048: by-group form)
      std::vector<uint32_t> id_of_slot((size_t)8 * G, 0xffffffffu);
      std::iota(id_of_slot.begin(), id_of_slot.begin() + k, 0u);
      std::stable_sort(id_of_slot.begin(), id_of_slot.begin() + k, [&](uint32_t a, uint32_t b) { return cnh[a] < cnh[b]; });
      slot_of_id_host.assign(k, 0u);
      for (int s2 = 0; s2 < k; ++s2) slot_of_id_host[id_of_slot[s2]] = (uint32_t)s2;
      ISLECHK(k_yy_map_upload(c, id_of_slot.data(), slot_of_id_host.data(), k, G, &ymap));
      HIPCHK(c, c->yy_cns.reserve((size_t)8 * G));
    }
    if (regroup) {
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      ISLECHK(k_yy_gather_by_slot(c, ymap, k, G, c->cnorm.p, c->yy_cns.p));
    }
    const float* cn_grp = regroup ? c->yy_cns.p : c->cnorm.p;  // the norms as the by-group kernels index them
    if (it == 0 && via_projection) {
      // B^T (U C^T) = (U^T B)^T C^T: the k-wide sparse product of the first assignment (distsq_docs_to_centers, :1494-1550) is a dense
      // D x k x k product on the projection that k-means++ / Lloyd in span(U) left on the device — one MFMA GEMM, a transposition into
      // the doc-major layout and the same distance / bound epilogue (norms of centres and documents are the word-space ones)
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      if (yinyang && fused_first) {  // distances, group bounds and candidates formed inside the product: no D x k matrix in memory
        float* cn_max_dev = c->Csum.p + 2 * k + 8;
        ISLECHK(k_max_f32(c, c->cnorm.p, k, cn_max_dev));
        const float* liftC = c->lift_C.p;
        if (regroup) {  // the product's columns in slot order: its groups of eight columns are the regrouped groups
          HIPCHK(c, c->yy_liftC.reserve((size_t)k * c->lift_ld));
          ISLECHK(k_yy_rows_by_slot(c, ymap, k, c->lift_C.p, c->lift_ld, c->yy_liftC.p));
          liftC = c->yy_liftC.p;
        }
        ISLECHK(k_gemm_assign_yy(c, c->Pt_ready ? c->Pt.p : nullptr, c->P.p, c->ldk, c->pnorm.p, D, k, liftC, c->lift_ld, k, G, cn_grp, c->dnorm.p, cn_max_dev,
                                 c->assign.p, c->hub.p, c->yglb.p, ISLE_T_SPARSE_ASSIGN, c->Pt2_ready ? c->Pt2.p : nullptr,
                                 c->Pt2_ready && c->Pt2_pos ? c->dperm.p : nullptr));
        if (regroup) ISLECHK(k_yy_labels_to_ids(c, ymap, c->assign.p, D));  // columns (slots) -> centres
      } else if (yinyang) {  // assignment and group bounds straight from the column-major product (the projection stays valid)
        HIPCHK(c, c->dotsT.reserve((size_t)D * k));
        ISLECHK(k_ensure_pt(c));
        ISLECHK(k_gemm_nn_assign(c, c->Pt.p, D, k, c->lift_C.p, c->lift_ld, k, c->dotsT.p, ISLE_T_SPARSE_ASSIGN));
        float* cn_max_dev = c->Csum.p + 2 * k + 8;
        ISLECHK(k_max_f32(c, c->cnorm.p, k, cn_max_dev));
        ISLECHK(k_dots_assign_cm(c, c->dotsT.p, k, G, c->cnorm.p, c->dnorm.p, cn_max_dev, c->assign.p, c->hub.p, c->yglb.p));
      } else {
        HIPCHK(c, c->dotsT.reserve((size_t)D * k));
        ISLECHK(k_ensure_pt(c));
        ISLECHK(k_gemm_nn_assign(c, c->Pt.p, D, k, c->lift_C.p, c->lift_ld, k, c->dotsT.p, ISLE_T_SPARSE_ASSIGN));
        c->P_ready = false;  // P now holds the dot products (as with the LDS-banded wide product)
        c->Pt_ready = false;
        c->Pt2_ready = false;
        if (ld != k) HIPCHK(c, hipMemsetAsync(c->P.p, 0, (size_t)D * ld * sizeof(float), c->stream));
        ISLECHK(k_transpose(c, c->dotsT.p, D, (uint64_t)k, D, c->P.p, (uint64_t)ld));
        ISLECHK(k_dots_assign(c, k, ld, c->cnorm.p, c->dnorm.p, c->assign.p, c->hub.p, c->hlb.p, 0));
      }
    } else if (it == 0 || !hamerly) {
      // documents are visited grouped by their previous centre (cache locality of the centre rows); results are order-independent
      ISLECHK(k_spmm_wide_assign(c, c->centers_rm.p, k, ld, c->cnorm.p, c->dnorm.p, c->assign.p,
                                 c->members_valid ? c->members.p : nullptr, nullptr, c->hub.p, yinyang ? c->yglb.p : c->hlb.p,
                                 yinyang ? G : 0));  // :1606
    } else if (yinyang) {
      // all bookkeeping of the Yinyang iteration stays on the device (largest centre norm, movements, group maxima, member
      // offsets): the only host round trip of an iteration is the one the stop rule needs
      float* cn_max_dev = c->Csum.p + 2 * k + 8;
      {
        TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
        ISLECHK(k_max_f32(c, c->cnorm.p, k, cn_max_dev));
      }
      uint32_t* nact = c->active.p + D;
      // large k: the centres also group-major (one 32-byte-row table per group) and the active documents grouped by their own group
      // (the member lists), so that waves running together gather from one table in L2; ISLE_YY_MODE = doc | docg | group picks the form
      // (measured, Lloyd on B per step: C3 shard 176 ms by document -> 112 ms by group, all of config 3 on one GPU 825 -> 588 ms; at C2,
      // G = 25 and a 40 MB table, the three forms are within 10 % of each other and the plain one stays)
      // Visiting order of the filter.  The fused by-group launch takes the documents in their own order since the end of round 5 (its first
      // phase then streams the bounds as one run per workgroup; once that phase kept sixteen loads in flight the member order's gain in the
      // second phase — neighbours share the own group's table — no longer paid for its scattered rows: sparse_assign 177 -> 170 ms at
      // config 3); the other forms keep the member lists' order.  ISLE_YY_ORDER = doc | member forces either.
      const char* yord = c->knob(KN_YY_ORDER);
      const bool by_members = yord ? strcmp(yord, "doc") != 0 : !(yy_mode == 2 && !c->knob_zero(KN_YY_FUSED));
      const uint32_t* order = yy_mode && c->members_valid && by_members ? c->members.p : nullptr;
      if (yy_mode) ISLECHK(k_yy_pack_groups(c, c->centers_rm.p, ld, G, ymap));
      // by group: the bounds are lowered and the active documents tightened in one launch (the D x G bounds read once), ISLE_YY_FUSED=0: in two
      const bool fused = yy_mode == 2 && !c->knob_zero(KN_YY_FUSED);
      // Movers.  A group's bound falls by the LARGEST movement among its eight centres, for every document: one centre that jumped (a
      // cluster of a handful of documents that gained or lost one) takes a whole group's bound away and makes nearly every document
      // active (config 3: 98 % in iterations 3 - 5, because of 1 - 3 centres).  Up to ten centres whose movement stands out (more than
      // twice the eleventh largest) are therefore left out of their groups' maxima and bounded by their exact new distances — one thin
      // pass of the pass-1 stream for b_d . c over all documents (k_yy_filter_tighten).  Exact: min(bound lowered by the other members'
      // movement, distance to the mover) is a lower bound of the group as before.  Same movements on every rank (the centres are all-reduced).
      YyMovers mv;
      const float* gmax_use = gmax_dev;
      // (the movers' distances are a thin product through the pass-1 stream: LDS-banded form only — the gather form, ISLE_GRAM_LDS=0 or a
      // matrix whose rows are not single-valued, keeps every centre inside its group's movement)
      if (fused && !c->knob_zero(KN_YY_MOVERS)) ISLECHK(k_band_build(c));
      if (fused && !c->knob_zero(KN_YY_MOVERS) && c->gl_mode == 1) {
        std::vector<float> gm;
        choose_movers(delta_host, k, 8, G, &mv, &gm, regroup ? &slot_of_id_host : nullptr);
        if (mv.n) {
          HIPCHK(c, c->yy_gmax2.reserve(G));
          HIPCHK(c, hipMemcpyAsync(c->yy_gmax2.p, gm.data(), (size_t)G * sizeof(float), hipMemcpyHostToDevice, c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream));  // gm is stack-owned
          gmax_use = c->yy_gmax2.p;
        }
      }
      if (fused)
        ISLECHK(k_yy_filter_tighten(c, order, c->assign.p, c->hub.p, c->yglb.p, G, delta_dev, gmax_use, c->active.p, nact, c->yy_cg.p, k, ld, cn_grp, c->dnorm.p,
                                    cn_max_dev, mv, c->centers_rm.p, ymap, c->cnorm.p));
      else
        ISLECHK(k_yy_filter(c, order, c->assign.p, c->hub.p, c->yglb.p, G, delta_dev, gmax_dev, c->active.p, nact));
      const bool dbg = c->knob_on(KN_DEBUG_HAMERLY);
      unsigned long long* dbg_dev = nullptr;
      if (dbg && fused) {  // diagnostic only: which bound the active documents reach, and by how much
        unsigned long long hh[16];
        ISLECHK(k_yy_dbg_margins(c, c->active.p, nact, c->assign.p, c->hub.p, c->yglb.p, G, ymap, hh));
        fprintf(stderr, "[yinyang] iter %d active by (ub - smallest group bound) < 1e-6 | 1e-5 | 1e-4 | 1e-3 | 1e-2 | 1e-1 | 1 | more:  own group", it);
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %llu", hh[b]);
        fprintf(stderr, ";  another group");
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %llu", hh[8 + b]);
        fprintf(stderr, "\n");
      }
      if (dbg) {  // diagnostic only: group scans and gathered nonzeros of this iteration
        HIPCHK(c, c->dbg_cnt.reserve(18));
        HIPCHK(c, hipMemsetAsync(c->dbg_cnt.p, 0, 16, c->stream));
        dbg_dev = c->dbg_cnt.p;
      }
      bool done = false;
      unsigned long long npairs = 0;
      if (yy_mode == 2)
        ISLECHK(k_yy2_assign(c, c->yy_cg.p, k, ld, G, cn_grp, c->dnorm.p, cn_max_dev, c->active.p, nact, c->assign.p, c->hub.p, c->yglb.p, &done, &npairs, fused, ymap));
      if (!done)
        ISLECHK(k_yy_scan(c, c->centers_rm.p, yy_mode ? c->yy_cg.p : nullptr, k, ld, G, cn_grp, c->dnorm.p, cn_max_dev, c->active.p, nact, c->assign.p,
                          c->hub.p, c->yglb.p, dbg_dev, ymap));
      if (dbg) {
        uint32_t na = 0;
        unsigned long long cnt[2] = {0, 0};
        HIPCHK(c, hipMemcpy(&na, nact, 4, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(cnt, dbg_dev, 16, hipMemcpyDeviceToHost));
        {
          std::vector<float> dl(k), gm(G), cn(k);
          float cm = 0.f;
          HIPCHK(c, hipMemcpy(dl.data(), delta_dev, k * sizeof(float), hipMemcpyDeviceToHost));
          HIPCHK(c, hipMemcpy(gm.data(), gmax_dev, G * sizeof(float), hipMemcpyDeviceToHost));
          HIPCHK(c, hipMemcpy(cn.data(), c->cnorm.p, k * sizeof(float), hipMemcpyDeviceToHost));
          HIPCHK(c, hipMemcpy(&cm, cn_max_dev, sizeof(float), hipMemcpyDeviceToHost));
          std::vector<float> sd(dl), sc(cn);
          std::sort(sd.begin(), sd.end());
          std::sort(sc.begin(), sc.end());
          std::vector<long long> szs;
          ISLECHK(fetch_sizes(c, k, szs));
          long long smin = szs[0], smax = szs[0], empty = 0;
          for (auto v : szs) { smin = std::min(smin, v); smax = std::max(smax, v); empty += v == 0; }
          {
            int n1 = 0, n2 = 0, n3 = 0;
            for (float v : sd) {
              n1 += v > 0.3f;
              n2 += v > 0.1f;
              n3 += v > 0.03f;
            }
            std::vector<float> gs(gm);
            std::sort(gs.begin(), gs.end());
            int g1 = 0, g2 = 0;
            for (float v : gs) {
              g1 += v > 0.1f;
              g2 += v > 0.03f;
            }
            fprintf(stderr, "[yinyang] iter %d: centres that moved more than 0.3 / 0.1 / 0.03: %d / %d / %d of %d; groups whose largest movement exceeds 0.1 / 0.03: %d / %d of %d\n", it, n1,
                    n2, n3, k, g1, g2, G);
          }
          fprintf(stderr, "[yinyang] iter %d: movement median %.3g max %.3g; |c|^2 median %.3g max %.3g (cn_max %.3g); cluster sizes %lld..%lld, %lld empty\n", it,
                  sd[k / 2], sd[k - 1], sc[k / 2], sc[k - 1], cm, smin, smax, empty);
        }
        if (done)
          fprintf(stderr, "[yinyang] iter %d active %u of %llu; by group: %llu pairs beside the own-group scans (%.1f per active document, of %d)\n", it, na,
                  (unsigned long long)D, npairs, na ? (double)npairs / na : 0.0, G);
        else
          fprintf(stderr, "[yinyang] iter %d active %u of %llu; group scans %llu (%.1f per active document, of %d), gathered nonzeros %llu\n", it, na,
                  (unsigned long long)D, cnt[0], na ? (double)cnt[0] / na : 0.0, G, cnt[1]);
      }
    } else {
      uint32_t* nact = c->active.p + D;
      ISLECHK(k_hamerly_filter(c, c->members_valid ? c->members.p : nullptr, c->assign.p, c->hub.p, c->hlb.p, delta_dev, top_dev, c->active.p,
                               nact));
      ISLECHK(k_spmm_wide_assign(c, c->centers_rm.p, k, ld, c->cnorm.p, c->dnorm.p, c->assign.p, c->active.p, nact, c->hub.p, c->hlb.p));
      if (c->knob_on(KN_DEBUG_HAMERLY)) {
        uint32_t na = 0;
        HIPCHK(c, hipMemcpy(&na, nact, 4, hipMemcpyDeviceToHost));
        fprintf(stderr, "[hamerly] iter %d active %u of %llu\n", it, na, (unsigned long long)D);
      }
    }
    ISLECHK(k_count_sizes(c, c->assign.p, D, k, c->counts.p));
    {  // documents grouped by centre: visiting order of the next assignment, and what the FRESH counting centroid update walks (the
       // first of a run; later ones go by the documents that changed centre).  The fused by-group launch visits the documents in their
       // own order: nothing reads the lists then, and they are not made (a sort of D keys per iteration)
      const char* yord = c->knob(KN_YY_ORDER);
      const bool lists_as_order = !yinyang || !yy_mode || (yord ? strcmp(yord, "doc") != 0 : !(yy_mode == 2 && !c->knob_zero(KN_YY_FUSED)));
      if (it == 0 || lists_as_order || c->gl_mode != 1 || c->knob_on(KN_CENTERS_FRESH)) {
        TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
        ISLECHK(k_member_lists_dev(c, c->assign.p, D, k, c->counts.p));
      } else {
        c->members_valid = false;  // the lists are those of an earlier assignment
      }
    }
    if (hamerly) HIPCHK(c, hipMemcpyAsync(c->centers_old.p, c->centers_rm.p, (size_t)V * ld * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    ISLECHK(k_centers_from_rows(c, c->assign.p, k, ld, c->centers_rm.p, it == 0));                 // :1613-1638
    ISLECHK(allreduce_sum<float>(c, c->centers_rm.p, (size_t)V * ld));
    std::vector<long long> sizes;
    ISLECHK(fetch_sizes(c, k, sizes));
    ISLECHK(k_scale_centers(c, c->centers_rm.p, V, k, ld, c->counts.p));  // :1641-1646
    if (hamerly && it + 1 < max_reps) {  // centre movements for the next filter
      TimeScope ts(c, ISLE_T_SPARSE_ASSIGN);
      ISLECHK(k_colnorms_rm(c, c->centers_rm.p, V, k, ld, delta_dev, c->centers_old.p));
      if (yinyang) {
        ISLECHK(k_yy_delta(c, delta_dev, k, G, 8, gmax_dev, ymap.id_of_slot));  // movements and group maxima stay on the device
        ISLECHK(fetch_delta(c, delta_dev, k, delta_host));  // ... and a copy of the k movements for the choice of the movers
      } else {
        ISLECHK(k_ham_delta(c, delta_dev, k, top_dev));
      }
    }
    bool conv = false;
    ISLECHK(stop.converged(sizes, c->assign.p, &conv));
    if (conv) {
      ++it;
      break;
    }
  }
  isle_host_mark("lloyds_sparse: loop done");
  c->assign_valid = true;  // the partition stays resident for isle_hip_catchwords
  if (assign && D) HIPCHK(c, hipMemcpyAsync(assign, c->assign.p, D * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  if (centers_out) {
    HIPCHK(c, c->centers_cm.reserve((size_t)V * k));
    // row-major (V x ld) -> col-major (V x k): view as a k x V col-major matrix with ld_in = ld
    ISLECHK(k_transpose(c, c->centers_rm.p, k, V, ld, c->centers_cm.p, V));
    HIPCHK(c, hipMemcpyAsync(centers_out, c->centers_cm.p, (size_t)V * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  isle_host_mark("lloyds_sparse: exit");
  if (iters_run) *iters_run = it;
  return 0;
}


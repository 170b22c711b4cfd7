// tools/microbench/lds_band_gather.hip — prototype of an LDS-banded form of Y = B^T X (Gram apply, pass 1).
// Question: the L1/TA path costs ~4 clk per gathered 48-B panel row (seg_gather_k).  How fast is the
// same gather when a band of X (3412 rows x 48 B = 160 KB) sits in LDS, every lane owns one document (register
// accumulators, no reduction) and the row ids of a (document block, band) cell are stored as a sliced-ELL stream
// of u16 band-local ids ([pair-step][lane], 4 B per lane per two nonzeros, fully coalesced)?
// Synthetic input: Poisson(lambda) nonzeros per (document, band), uniform rows.
// build: hipcc -O3 --offload-arch=gfx950 -o lds_band_gather lds_band_gather.hip ; run: ./lds_band_gather [docs] [bands] [lambda]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

constexpr int RB = 3412;       // rows per band; row RB of the LDS image is a zero row (padding target)
constexpr int LDS_BYTES = (RB + 1) * 48;

__device__ inline void add4(float4& a, const float4 b) {
  a.x += b.x;
  a.y += b.y;
  a.z += b.z;
  a.w += b.w;
}

// ids: pair-steps of 64 u32 (two u16 band-local row ids per lane); soff[slice] = first pair-step of a slice,
// slices ordered [workgroup][band][wave][g].  Lane l of wave w, group g owns document ((wg*G + g)*WAVES + w)*64 + l.
template <int G, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void band_gather_k(const float4* __restrict__ X, const uint32_t* __restrict__ ids,
                                                       const uint32_t* __restrict__ soff, float4* __restrict__ Y, int nbands, uint32_t V,
                                                       uint32_t D, int reload) {
  extern __shared__ float4 xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 acc[G][3];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = acc[g][2] = make_float4(0.f, 0.f, 0.f, 0.f);
  const uint32_t PAD = (uint32_t)RB | ((uint32_t)RB << 16);
  for (int band = 0; band < nbands; ++band) {
    __syncthreads();
    if (reload || band == 0) {
    const uint32_t r0 = (uint32_t)band * RB;
    const uint32_t nrow = min((uint32_t)RB, V - r0);
    const float4* src = X + (size_t)r0 * 3;
    const uint32_t n4 = nrow * 3;  // <= 10 * 1024: batches of five loads in flight per thread
    constexpr uint32_t NT = WAVES * 64;
#pragma unroll
    for (int h = 0; h < 2 * (16 / WAVES); ++h) {
      float4 tmp[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) tmp[j] = src[min(threadIdx.x + (h * 5 + j) * NT, n4 - 1)];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint32_t i = threadIdx.x + (h * 5 + j) * NT;
        if (i < n4) xs[i] = tmp[j];
      }
    }
    if (threadIdx.x < 3) xs[RB * 3 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    const size_t sbase = (((size_t)blockIdx.x * nbands + band) * WAVES + w) * G;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t o0 = soff[sbase + g], o1 = soff[sbase + g + 1];
      for (uint32_t t = o0; t < o1; t += 4) {
        uint32_t u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = ids[(size_t)(t + j) * 64 + lane];  // slack behind the array: always in bounds
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t uu = (t + j < o1) ? u[j] : PAD;
          const uint32_t a = (uu & 0xffffu) * 3, b = (uu >> 16) * 3;
          add4(acc[g][0], xs[a]);
          add4(acc[g][1], xs[a + 1]);
          add4(acc[g][2], xs[a + 2]);
          add4(acc[g][0], xs[b]);
          add4(acc[g][1], xs[b + 1]);
          add4(acc[g][2], xs[b + 2]);
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t d = (((size_t)blockIdx.x * G + g) * WAVES + w) * 64 + lane;
    if (d < D) {
      Y[d * 3] = acc[g][0];
      Y[d * 3 + 1] = acc[g][1];
      Y[d * 3 + 2] = acc[g][2];
    }
  }
}

template <int G, int WAVES>
static void run(uint32_t D, int nbands, double lambda, int mode) {
  const uint32_t V = (uint32_t)nbands * RB;
  const uint32_t dpw = 64 * WAVES * G;
  const uint32_t nwg = (D + dpw - 1) / dpw;
  std::mt19937_64 rng(1);
  std::poisson_distribution<int> pois(lambda);
  const size_t nslice = (size_t)nwg * nbands * WAVES * G;
  std::vector<uint32_t> soff(nslice + 1);
  std::vector<uint32_t> ids;
  ids.reserve((size_t)(D * (double)nbands * lambda * 1.3));
  std::vector<float> X((size_t)V * 12);
  for (auto& x : X) x = (float)((rng() >> 40) * (1.0 / (1 << 24)));
  std::vector<double> ref((size_t)D, 0.0);  // sum over the 12 columns, as a checksum per document
  std::vector<float> rowsum(V);
  for (uint32_t r = 0; r < V; ++r) {
    double s = 0;
    for (int j = 0; j < 12; ++j) s += X[(size_t)r * 12 + j];
    rowsum[r] = (float)s;
  }
  size_t nnz = 0, padded = 0;
  std::vector<std::vector<uint16_t>> cell(64);
  for (uint32_t wg = 0; wg < nwg; ++wg)
    for (int band = 0; band < nbands; ++band)
      for (int w = 0; w < WAVES; ++w)
        for (int g = 0; g < G; ++g) {
          const size_t s = (((size_t)wg * nbands + band) * WAVES + w) * G + g;
          soff[s] = (uint32_t)(ids.size() / 64);
          size_t mx = 0;
          for (int l = 0; l < 64; ++l) {
            const size_t d = (((size_t)wg * G + g) * WAVES + w) * 64 + l;
            cell[l].clear();
            if (d < D) {
              const int n = pois(rng);
              for (int i = 0; i < n; ++i) {
                const uint16_t r = mode == 1 ? (uint16_t)((l + 64 * i) % RB) : (uint16_t)(rng() % RB);
                cell[l].push_back(r);
                ref[d] += rowsum[(size_t)band * RB + r];
              }
              nnz += n;
            }
            mx = std::max(mx, cell[l].size());
          }
          const size_t ps = (mx + 1) / 2;
          padded += ps * 2 * 64;
          for (size_t p = 0; p < ps; ++p)
            for (int l = 0; l < 64; ++l) {
              const uint32_t a = 2 * p < cell[l].size() ? cell[l][2 * p] : RB;
              const uint32_t b = 2 * p + 1 < cell[l].size() ? cell[l][2 * p + 1] : RB;
              ids.push_back(a | (b << 16));
            }
        }
  soff[nslice] = (uint32_t)(ids.size() / 64);
  ids.resize(ids.size() + 8 * 64, 0);  // slack for the unconditional loads
  float4 *dX, *dY;
  uint32_t *dids, *dsoff;
  CK(hipMalloc(&dX, X.size() * 4));
  CK(hipMalloc(&dY, (size_t)D * 48));
  CK(hipMalloc(&dids, ids.size() * 4));
  CK(hipMalloc(&dsoff, soff.size() * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsoff, soff.data(), soff.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)band_gather_k<G, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((band_gather_k<G, WAVES>), dim3(nwg), dim3(WAVES * 64), LDS_BYTES, 0, dX, dids, dsoff, dY, nbands, V, D, mode != 2);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  std::vector<float> Y((size_t)D * 12);
  CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (uint32_t d = 0; d < D; d += 97) {
    double s = 0;
    for (int j = 0; j < 12; ++j) s += Y[(size_t)d * 12 + j];
    maxerr = std::max(maxerr, std::fabs(s - ref[d]) / (1.0 + std::fabs(ref[d])));
  }
  printf("mode=%d G=%d W=%d docs=%u bands=%d lambda=%.1f nnz=%zu padded=%.2fx wgs=%u : %.3f ms  (%.2f ps/nnz, %.2f clk/nnz/CU @2.4GHz,256CU)  err=%.2e\n", mode, G, WAVES, D, nbands,
         lambda, nnz, (double)padded / nnz, nwg, best, best * 1e9 / nnz, best * 1e-3 * 2.4e9 * 256 / nnz, maxerr);
  hipFree(dX);
  hipFree(dY);
  hipFree(dids);
  hipFree(dsoff);
}

// ---- v2: one flat id stream per wave over all bands, G documents per lane interleaved at round level
// (round = G u32 per lane = 2 nonzeros for each of the lane's G documents), register ring prefetch of PF rounds.
template <int G> struct UG;
template <> struct UG<2> { typedef uint2 T; };
template <> struct UG<4> { typedef uint4 T; };
__device__ inline uint32_t ug_get(const uint2& u, int g) { return g == 0 ? u.x : u.y; }
__device__ inline uint32_t ug_get(const uint4& u, int g) { return g == 0 ? u.x : g == 1 ? u.y : g == 2 ? u.z : u.w; }

template <int G, int PF>
__global__ __launch_bounds__(1024) void band_gather2_k(const float4* __restrict__ X, const typename UG<G>::T* __restrict__ ids,
                                                        const uint32_t* __restrict__ wstart, const uint16_t* __restrict__ nr,
                                                        float4* __restrict__ Y, int nbands, uint32_t V, uint32_t D) {
  typedef typename UG<G>::T U;
  extern __shared__ float4 xs[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 acc[G][3];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g][0] = acc[g][1] = acc[g][2] = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t wv = (size_t)blockIdx.x * 16 + w;
  const U* p = ids + (size_t)wstart[wv] * 64 + lane;
  U q[PF];
#pragma unroll
  for (int j = 0; j < PF; ++j) q[j] = p[(size_t)j * 64];
  p += (size_t)PF * 64;
  for (int band = 0; band < nbands; ++band) {
    __syncthreads();
    const uint32_t r0 = (uint32_t)band * RB;
    const uint32_t nrow = min((uint32_t)RB, V - r0);
    const float4* src = X + (size_t)r0 * 3;
    const uint32_t n4 = nrow * 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 tmp[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) tmp[j] = src[min(threadIdx.x + (h * 5 + j) * 1024u, n4 - 1)];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const uint32_t i = threadIdx.x + (h * 5 + j) * 1024u;
        if (i < n4) xs[i] = tmp[j];
      }
    }
    if (threadIdx.x < 3) xs[RB * 3 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int n = nr[wv * nbands + band];  // multiple of PF
    for (int r = 0; r < n; r += PF) {
#pragma unroll
      for (int j = 0; j < PF; ++j) {
        const U u = q[j];
        q[j] = *p;  // round (current + PF); slack behind the array keeps it in bounds
        p += 64;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const uint32_t uu = ug_get(u, g);
          const uint32_t a = (uu & 0xffffu) * 3, b = (uu >> 16) * 3;
          add4(acc[g][0], xs[a]);
          add4(acc[g][1], xs[a + 1]);
          add4(acc[g][2], xs[a + 2]);
          add4(acc[g][0], xs[b]);
          add4(acc[g][1], xs[b + 1]);
          add4(acc[g][2], xs[b + 2]);
        }
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t d = ((size_t)blockIdx.x * 16 + w) * 64 * G + (size_t)g * 64 + lane;
    if (d < D) {
      Y[d * 3] = acc[g][0];
      Y[d * 3 + 1] = acc[g][1];
      Y[d * 3 + 2] = acc[g][2];
    }
  }
}

template <int G, int PF>
static void run2(uint32_t D, int nbands, double lambda, int mode) {
  typedef typename UG<G>::T U;
  const uint32_t V = (uint32_t)nbands * RB;
  const uint32_t dpw = 1024 * G;
  const uint32_t nwg = (D + dpw - 1) / dpw;
  std::mt19937_64 rng(1);
  std::poisson_distribution<int> pois(lambda);
  std::vector<uint32_t> wstart((size_t)nwg * 16 + 1);
  std::vector<uint16_t> nr((size_t)nwg * 16 * nbands);
  std::vector<uint32_t> ids;
  ids.reserve((size_t)(D * (double)nbands * lambda * 1.4));
  std::vector<float> X((size_t)V * 12);
  for (auto& x : X) x = (float)((rng() >> 40) * (1.0 / (1 << 24)));
  std::vector<double> ref((size_t)D, 0.0);
  std::vector<float> rowsum(V);
  for (uint32_t r = 0; r < V; ++r) {
    double s = 0;
    for (int j = 0; j < 12; ++j) s += X[(size_t)r * 12 + j];
    rowsum[r] = (float)s;
  }
  size_t nnz = 0, padded = 0;
  std::vector<std::vector<uint16_t>> cell(64 * G);
  for (size_t wv = 0; wv < (size_t)nwg * 16; ++wv) {
    wstart[wv] = (uint32_t)(ids.size() / (64 * G));
    for (int band = 0; band < nbands; ++band) {
      size_t mx = 0;
      for (int g = 0; g < G; ++g)
        for (int l = 0; l < 64; ++l) {
          const size_t d = wv * 64 * G + (size_t)g * 64 + l;
          auto& c = cell[g * 64 + l];
          c.clear();
          if (d < D) {
            const int n = pois(rng);
            for (int i = 0; i < n; ++i) {
              const uint16_t r = mode == 1 ? (uint16_t)((l + 64 * i) % RB) : (uint16_t)(rng() % RB);
              c.push_back(r);
              ref[d] += rowsum[(size_t)band * RB + r];
            }
            nnz += n;
          }
          mx = std::max(mx, c.size());
        }
      size_t rounds = (mx + 1) / 2;
      rounds = (rounds + PF - 1) / PF * PF;
      nr[wv * nbands + band] = (uint16_t)rounds;
      padded += rounds * 2 * 64 * G;
      for (size_t r = 0; r < rounds; ++r)
        for (int l = 0; l < 64; ++l)
          for (int g = 0; g < G; ++g) {
            auto& c = cell[g * 64 + l];
            const uint32_t a = 2 * r < c.size() ? c[2 * r] : RB;
            const uint32_t b = 2 * r + 1 < c.size() ? c[2 * r + 1] : RB;
            ids.push_back(a | (b << 16));
          }
    }
  }
  wstart[(size_t)nwg * 16] = (uint32_t)(ids.size() / (64 * G));
  ids.resize(ids.size() + (size_t)2 * PF * 64 * G, (uint32_t)RB | ((uint32_t)RB << 16));
  float4 *dX, *dY;
  uint32_t *dids, *dws;
  uint16_t* dnr;
  CK(hipMalloc(&dX, X.size() * 4));
  CK(hipMalloc(&dY, (size_t)D * 48));
  CK(hipMalloc(&dids, ids.size() * 4));
  CK(hipMalloc(&dws, wstart.size() * 4));
  CK(hipMalloc(&dnr, nr.size() * 2));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dws, wstart.data(), wstart.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dnr, nr.data(), nr.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)band_gather2_k<G, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((band_gather2_k<G, PF>), dim3(nwg), dim3(1024), LDS_BYTES, 0, dX, (const U*)dids, dws, dnr, dY, nbands, V, D);
    CK(hipGetLastError());
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  std::vector<float> Y((size_t)D * 12);
  CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (uint32_t d = 0; d < D; d += 97) {
    double s = 0;
    for (int j = 0; j < 12; ++j) s += Y[(size_t)d * 12 + j];
    maxerr = std::max(maxerr, std::fabs(s - ref[d]) / (1.0 + std::fabs(ref[d])));
  }
  printf("v2 mode=%d G=%d PF=%d docs=%u bands=%d lambda=%.1f nnz=%zu padded=%.2fx wgs=%u : %.3f ms  (%.2f ps/nnz, %.2f clk/nnz/CU)  err=%.2e\n", mode, G,
         PF, D, nbands, lambda, nnz, (double)padded / nnz, nwg, best, best * 1e9 / nnz, best * 1e-3 * 2.4e9 * 256 / nnz, maxerr);
  hipFree(dX);
  hipFree(dY);
  hipFree(dids);
  hipFree(dws);
  hipFree(dnr);
}

int main(int argc, char** argv) {
  const uint32_t D = argc > 1 ? (uint32_t)atol(argv[1]) : 1000000u;
  const int nbands = argc > 2 ? atoi(argv[2]) : 15;
  const double lambda = argc > 3 ? atof(argv[3]) : 6.6;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  run2<2, 2>(D, nbands, lambda, mode);
  run2<2, 4>(D, nbands, lambda, mode);
  run2<4, 2>(D, nbands, lambda, mode);
  run2<4, 4>(D, nbands, lambda, mode);
  if (argc > 5) return 0;
  run<1, 16>(D, nbands, lambda, mode);
  run<2, 16>(D, nbands, lambda, mode);
  run<4, 16>(D, nbands, lambda, mode);
  run<2, 8>(D, nbands, lambda, mode);
  run<4, 8>(D, nbands, lambda, mode);
  run<8, 8>(D, nbands, lambda, mode);
  return 0;
}

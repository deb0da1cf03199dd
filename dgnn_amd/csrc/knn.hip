// Exact brute-force k-nearest-neighbour graph per point cloud: the step in front of the fused
// EdgeConv (the reference takes it from DGL: KNNGraph, script/train/train_edgeconv.py:4,17,49).
//
// Output convention = what the reference feeds edgeconv_forward (train_edgeconv.py:51-52 and
// fused_edgeconv_kernel.cu:21-27): src_ind[r*k + t] = global id of the t-th nearest point of
// point r inside r's own cloud, r itself included (distance 0), ascending distance, ties ->
// lower id.
//
// Arithmetic is pinned so that the selection is reproducible bit for bit (oracle_knn in
// oracle/dgnn_oracle.c runs the same chain): d(i,j) = fma-chain over f ascending of
// (x_i[f]-x_j[f])^2 in float32.  It is vector-ALU work, not a GEMM: the -2xy^T expansion would
// not give the same bits and cancels badly for near neighbours.
//
// Mapping (gfx950, wave64): one thread owns one query point and keeps it in registers
// (up to 128 dims at a time) together with its sorted candidate list; the block streams the
// cloud's points through LDS in tiles of TJ candidates, every lane reads the same candidate
// element (LDS broadcast, conflict-free), so one 16-byte LDS read feeds 8 vector ops per lane.
#include "../../include/dgnn_hip.h"
#include "common.hpp"

namespace dgnn {
namespace {

constexpr int KNN_TJ = 32;        // candidates per LDS tile
constexpr int KNN_THREADS = 256;  // queries per block

template <int KMAX>
struct TopK {
  float d[KMAX];
  int id[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { d[t] = __builtin_inff(); id[t] = -1; }
  }
  // candidates arrive in ascending id: a strict compare keeps the lower id in front on ties
  __device__ __forceinline__ void push(float dn, int jn) {
    if (dn < d[KMAX - 1]) {
      bool sw = false;   // once the newcomer has found its slot the rest of the list shifts down,
                         // equal distances included (they must stay behind in their old order)
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        sw = sw || dn < d[t];
        const float td = d[t];
        const int ti = id[t];
        d[t] = sw ? dn : td;
        id[t] = sw ? jn : ti;
        dn = sw ? td : dn;
        jn = sw ? ti : jn;
      }
    }
  }
};

// FC: query dims held in registers per chunk (multiple of 4).  SINGLE: dim <= FC, the query is
// loaded once; otherwise the distance chain is carried across chunks through LDS.
template <int KMAX, int FC, bool SINGLE>
__global__ __launch_bounds__(KNN_THREADS) void k_knn(int points, int dim, int k,
                                                     const float *__restrict__ x,
                                                     int *__restrict__ src_ind) {
  __shared__ __align__(16) float tile[KNN_TJ * FC];
  extern __shared__ __align__(16) float carry[];   // [KNN_TJ][KNN_THREADS] when !SINGLE
  const int b = blockIdx.y;
  const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
  const bool active = i < points;
  const float *xc = x + (size_t)b * points * dim;          // this cloud
  const float *xi_ptr = xc + (size_t)(active ? i : 0) * dim;
  const int nchunks = SINGLE ? 1 : (dim + FC - 1) / FC;

  float xi[FC];
  if (SINGLE) {
#pragma unroll
    for (int f = 0; f < FC; ++f) xi[f] = (f < dim) ? xi_ptr[f] : 0.f;
  }
  TopK<KMAX> best;
  best.init();

  for (int j0 = 0; j0 < points; j0 += KNN_TJ) {
    const int jn = min(KNN_TJ, points - j0);
    for (int c = 0; c < nchunks; ++c) {
      const int f0 = c * FC;
      if (!SINGLE) {
#pragma unroll
        for (int f = 0; f < FC; ++f) xi[f] = (f0 + f < dim) ? xi_ptr[f0 + f] : 0.f;
      }
      __syncthreads();                                     // previous tile fully consumed
      for (int e = threadIdx.x; e < KNN_TJ * FC; e += KNN_THREADS) {
        const int j = e / FC, f = e - j * FC;
        tile[e] = (j < jn && f0 + f < dim) ? xc[(size_t)(j0 + j) * dim + f0 + f] : 0.f;
      }
      __syncthreads();
      for (int j = 0; j < jn; ++j) {
        float acc = (SINGLE || c == 0) ? 0.f : carry[j * KNN_THREADS + threadIdx.x];
        const float4 *cj = (const float4 *)(tile + j * FC);
#pragma unroll
        for (int f4 = 0; f4 < FC / 4; ++f4) {
          const float4 v = cj[f4];
          float t;
          // padded dims hold 0 on both sides: fma(0,0,acc) == acc exactly
          t = xi[4 * f4 + 0] - v.x; acc = fmaf(t, t, acc);
          t = xi[4 * f4 + 1] - v.y; acc = fmaf(t, t, acc);
          t = xi[4 * f4 + 2] - v.z; acc = fmaf(t, t, acc);
          t = xi[4 * f4 + 3] - v.w; acc = fmaf(t, t, acc);
        }
        if (SINGLE || c == nchunks - 1) {
          best.push(acc, j0 + j);
        } else {
          carry[j * KNN_THREADS + threadIdx.x] = acc;
        }
      }
    }
  }
  if (active) {
    int *o = src_ind + ((size_t)b * points + i) * k;
    const int base = b * points;
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
      if (t < k) o[t] = base + best.id[t];
  }
}

template <int KMAX>
int launch_knn(int batch, int points, int dim, int k, const float *x, int *src_ind, hipStream_t st) {
  const dim3 grid((unsigned)((points + KNN_THREADS - 1) / KNN_THREADS), (unsigned)batch);
  const dim3 block(KNN_THREADS);
#define DGNN_KNN_CASE(FCV)                                                                       \
  if (dim <= FCV) {                                                                              \
    hipLaunchKernelGGL((k_knn<KMAX, FCV, true>), grid, block, 0, st, points, dim, k, x, src_ind); \
    DGNN_LAUNCH_CHECK();                                                                         \
    return DGNN_OK;                                                                              \
  }
  DGNN_KNN_CASE(4)
  DGNN_KNN_CASE(16)
  DGNN_KNN_CASE(32)
  DGNN_KNN_CASE(64)
  DGNN_KNN_CASE(128)
#undef DGNN_KNN_CASE
  hipLaunchKernelGGL((k_knn<KMAX, 128, false>), grid, block,
                     sizeof(float) * KNN_TJ * KNN_THREADS, st, points, dim, k, x, src_ind);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" int dgnn_knn_graph(int batch, int points, int dim, int k, const float *x, int *src_ind,
                              void *stream) {
  if (batch < 0 || points < 0 || dim <= 0 || k <= 0) return DGNN_ERR_BAD_ARG;
  if (batch == 0 || points == 0) return DGNN_OK;
  if (k > points) return DGNN_ERR_BAD_ARG;          // a cloud cannot supply k distinct points
  if (k > 64 || batch > 65535) return DGNN_ERR_UNSUPPORTED;
  if ((long long)batch * points > (1LL << 31) - 1) return DGNN_ERR_UNSUPPORTED;
  if (!x || !src_ind) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (k <= 8) return launch_knn<8>(batch, points, dim, k, x, src_ind, st);
  if (k <= 16) return launch_knn<16>(batch, points, dim, k, x, src_ind, st);
  if (k <= 24) return launch_knn<24>(batch, points, dim, k, x, src_ind, st);
  if (k <= 32) return launch_knn<32>(batch, points, dim, k, x, src_ind, st);
  if (k <= 48) return launch_knn<48>(batch, points, dim, k, x, src_ind, st);
  return launch_knn<64>(batch, points, dim, k, x, src_ind, st);
}

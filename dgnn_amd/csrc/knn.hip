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
// Mapping (gfx950, wave64): one lane owns one query point and keeps it in registers
// (up to 128 dims at a time) together with its sorted candidate list; a wave streams its share
// of the cloud's points through LDS in tiles, every lane reads the same candidate element
// (LDS broadcast, conflict-free), so one 16-byte LDS read feeds 8 vector ops per lane.
#include <cstdlib>

#include "../../include/dgnn_hip.h"
#include "common.hpp"

namespace dgnn {
namespace {

template <int KMAX, bool MED3>
struct TopK {
  float d[KMAX];
  int id[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { d[t] = __builtin_inff(); id[t] = -1; }
  }
  // candidates arrive in ascending id: a strict compare keeps the lower id in front on ties.
  // The list is sorted, so "newcomer in front of slot t" (dn < d[t]) is monotone in t and every
  // slot can be rewritten on its own: slot t takes the median of (d[t-1], dn, d[t]) -- itself
  // when the newcomer goes behind it, the newcomer at the insertion point, its predecessor after
  // it (equal distances stay behind in their old order) -- and the id follows the same choice.
  // MED3 is the faster form up to 64 dims (3 dims: 0.30 -> 0.26 ms); with 128 query registers
  // its extra temporaries cost more than they save (1.58 -> 1.69 ms): there the chained form runs.
  __device__ __forceinline__ void push(float dn, int jn) {
    if (MED3) {
      push_med3(dn, jn);
    } else if (dn < d[KMAX - 1]) {
      bool sw = false;   // once the newcomer has found its slot the rest of the list shifts down
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
  __device__ __forceinline__ void push_med3(float dn, int jn) {
    if (dn < d[KMAX - 1]) {
      bool front_prev = false;
      float dprev = -__builtin_inff();
      int iprev = -1;
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        const float td = d[t];
        const int ti = id[t];
        const bool front = dn < td;
        d[t] = __builtin_amdgcn_fmed3f(dprev, dn, td);
        id[t] = front ? (front_prev ? iprev : jn) : ti;
        front_prev = front;
        dprev = td;
        iprev = ti;
      }
    }
  }
};

// FC: query dims held in registers per chunk (multiple of 4).  SINGLE: dim <= FC, the query is
// loaded once; otherwise the distance chain is carried across chunks through LDS.
//
// A block owns 64 query points; its four waves each scan a quarter of the cloud's candidates
// for the SAME 64 queries (lane = query) with a private LDS tile, so waves never wait for one
// another while scanning and there are 4x as many waves as queries/64 to hide LDS and VALU
// latency.  The four sorted lists are merged by wave 0 through LDS in part order, which keeps
// the tie rule (a later part only holds higher ids).
constexpr int KNN_MAX_PARTS = DGNN_WAVES_PER_BLOCK;   // 4 waves per block at most

// A candidate enters a lane's sorted list only if it beats the list's last entry, but the
// insertion is a pass over the whole list executed by the wave whenever ANY of its 64 lanes
// needs it -- with 64 independent queries that is nearly every candidate (P = 1-(1-k/j)^64), and
// the insertion (~6 vector ops per list slot) costs more than the distance itself below ~64
// dims.  So accepted candidates are first appended to a per-lane queue of KNN_BUF entries in LDS
// (two stores) and inserted, in arrival order, only when some lane's queue is full: same list,
// same ties, a third of the insertion passes.
constexpr int KNN_BUF = 8;   // 4: 5-15 % slower, 16: the same

static inline int knn_tile(int fc) { return fc >= 128 ? 16 : 32; }    // candidates per LDS tile
static inline size_t knn_lds_bytes(int fc, bool single, int k, int parts) {
  const size_t tile = (size_t)parts * knn_tile(fc) * fc * sizeof(float);
  const size_t carry = single ? 0 : (size_t)parts * knn_tile(fc) * 64 * sizeof(float);
  const size_t queue = (size_t)parts * KNN_BUF * 64 * 2 * sizeof(float);
  const size_t lists = (size_t)(parts - 1) * k * 64 * 2 * sizeof(float);
  return tile + carry + queue > lists ? tile + carry + queue : lists;
}

template <int KMAX, int FC, bool SINGLE>
__global__ __launch_bounds__(64 * KNN_MAX_PARTS)
void k_knn(int points, int dim, int k, int tj,
                                                    const float *__restrict__ x,
                                                    int *__restrict__ src_ind) {
  extern __shared__ __align__(16) float lds[];
  const int part = wave_in_block(), lane = lane_id();
  const int KNN_PARTS = (int)(blockDim.x >> 6);            // waves of this block = parts of the cloud
  float *tile = lds + (size_t)part * tj * FC;                                   // [tj][FC]
  float *carry = lds + (size_t)KNN_PARTS * tj * FC + (size_t)part * tj * 64;    // [tj][64], !SINGLE
  TopK<KMAX, (FC < 128)> best;
  best.init();
  float *qd = lds + (size_t)KNN_PARTS * tj * FC + (SINGLE ? 0 : (size_t)KNN_PARTS * tj * 64) +
              (size_t)part * KNN_BUF * 64 * 2;                                  // [KNN_BUF][64] distances
  int *qi = (int *)(qd + KNN_BUF * 64);                                         // [KNN_BUF][64] ids
  int queued = 0;                                                               // per lane
  auto flush = [&]() {
    for (int t = 0; t < KNN_BUF; ++t) {
      if (!__any(t < queued)) break;
      const bool have = t < queued;
      const float dn = have ? qd[t * 64 + lane] : __builtin_inff();
      const int jn2 = have ? qi[t * 64 + lane] : -1;
      best.push(dn, jn2);
    }
    queued = 0;
  };
  const int b = blockIdx.y;
  const int i = blockIdx.x * 64 + lane;
  const bool active = i < points;
  const float *xc = x + (size_t)b * points * dim;          // this cloud
  const float *xi_ptr = xc + (size_t)(active ? i : 0) * dim;
  const int nchunks = SINGLE ? 1 : (dim + FC - 1) / FC;
  const int per = (points + KNN_PARTS - 1) / KNN_PARTS;
  const int jbeg = uniform_i32(part * per), jend = uniform_i32(min(points, (part + 1) * per));

  float xi[FC];
  if (SINGLE) {
#pragma unroll
    for (int f = 0; f < FC; ++f) xi[f] = (f < dim) ? xi_ptr[f] : 0.f;
  }
  for (int j0 = jbeg; j0 < jend; j0 += tj) {
    const int jn = min(tj, jend - j0);
    for (int c = 0; c < nchunks; ++c) {
      const int f0 = c * FC;
      if (!SINGLE) {
#pragma unroll
        for (int f = 0; f < FC; ++f) xi[f] = (f0 + f < dim) ? xi_ptr[f0 + f] : 0.f;
      }
      wave_lds_sync();                                     // previous tile fully consumed
      for (int e = lane; e < tj * FC; e += 64) {
        const int j = e / FC, f = e - j * FC;
        tile[e] = (j < jn && f0 + f < dim) ? xc[(size_t)(j0 + j) * dim + f0 + f] : 0.f;
      }
      wave_lds_sync();
      // (candidate coordinates through wave-uniform scalar loads from global memory instead of
      // the LDS tile -- the compiler emits s_load_dwordx16 and scalar ALU operands -- measured:
      // 1.14 -> 1.30 ms at 64 dims, 1.70 -> 2.58 ms at 128: the 102 scalar registers cannot hold
      // the next candidate while this one is consumed, so the load latency is exposed)
      // (packed-fp32 v_pk_fma_f32 was measured twice: two candidates per instruction gave no gain
      // at 64 dims and ran 2.6x slower at 128 dims from the doubled query registers; two dims of
      // one candidate per instruction -- even/odd chains -- was 5-8 % slower than this scalar loop)
      // (also measured without gain at 64 / 128 dims: the candidate's dims as DPP row-broadcast
      // operands of the subtraction, 16 dims per register, instead of 16-byte LDS broadcast reads
      // (0.96 -> 0.97 / 1.57 -> 1.73 ms); two candidates per trip as two independent fma chains
      // (0.96 -> 0.94 / 1.57 -> 3.3 ms, registers).  The loop issues one vector instruction per
      // ~6 cycles per SIMD against 4 for unpacked fp32 at two waves per SIMD.)
      for (int j = 0; j < jn; ++j) {
        float acc = (SINGLE || c == 0) ? 0.f : carry[j * 64 + lane];
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
          if (acc < best.d[KMAX - 1]) {
            qd[queued * 64 + lane] = acc;
            qi[queued * 64 + lane] = j0 + j;
            ++queued;
          }
          if (__any(queued == KNN_BUF)) flush();
        } else {
          carry[j * 64 + lane] = acc;
        }
      }
    }
  }
  flush();
  // merge: parts 1..3 publish their lists, wave 0 pushes them in part order
  __syncthreads();                                         // every tile is dead from here on
  float *ld = lds;                                         // [part-1][t][64] distances
  int *li = (int *)(lds + (size_t)(KNN_PARTS - 1) * k * 64);   // same layout, ids
  if (part > 0) {
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
      if (t < k) {
        ld[((size_t)(part - 1) * k + t) * 64 + lane] = best.d[t];
        li[((size_t)(part - 1) * k + t) * 64 + lane] = best.id[t];
      }
  }
  __syncthreads();
  if (part == 0) {
    for (int p = 0; p < KNN_PARTS - 1; ++p)
      for (int t = 0; t < k; ++t) {
        const float dn = ld[((size_t)p * k + t) * 64 + lane];
        const int jn2 = li[((size_t)p * k + t) * 64 + lane];
        if (jn2 >= 0) best.push(dn, jn2);
      }
    if (active) {
      int *o = src_ind + ((size_t)b * points + i) * k;
      const int base = b * points;
#pragma unroll
      for (int t = 0; t < KMAX; ++t)
        if (t < k) o[t] = base + (best.id[t] >= 0 ? best.id[t] : i);   // unfilled (NaN / inf distances): the query itself
    }
  }
}

// More than 64 KiB of dynamic LDS has to be allowed per kernel (k above ~40 in the merge).
static int knn_allow_lds(const void *fn, size_t bytes) {
  if (bytes <= 64 * 1024) return DGNN_OK;
  if (bytes > 160 * 1024) return DGNN_ERR_UNSUPPORTED;
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? DGNN_OK : (int)e;
}

template <int KMAX>
int launch_knn(int batch, int points, int dim, int k, const float *x, int *src_ind, hipStream_t st) {
  const dim3 grid((unsigned)((points + 63) / 64), (unsigned)batch);
  // Parts per query block: each part keeps its own sorted list, so fewer parts mean fewer list
  // insertions in total, but the chip wants its two resident waves per SIMD (2048 waves) and no
  // second round of blocks: the largest of 4 / 2 / 1 parts that keeps query blocks x parts near
  // 2048 (tools/knn_parts_ab.py, 64 dims: 32 x 2048 points 0.89 -> 0.80 ms with 2 parts,
  // 64 x 2048 points 1.71 -> 1.52 ms with 1; 8 x 2048 points want all 4: 0.35 against 1.15 ms).
  const long long qblocks = (long long)((points + 63) / 64) * batch;
  int parts = qblocks <= 640 ? KNN_MAX_PARTS : (qblocks <= 1280 ? 2 : 1);
  if (const char *e = getenv("DGNN_KNN_PARTS")) {       // tuning override: 1, 2 or 4
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4) parts = v;
  }
  const dim3 block(64 * parts);
#define DGNN_KNN_CASE(FCV)                                                                          \
  if (dim <= FCV) {                                                                                 \
    if (int rc = knn_allow_lds((const void *)k_knn<KMAX, FCV, true>, knn_lds_bytes(FCV, true, k, parts)))  \
      return rc;                                                                                    \
    hipLaunchKernelGGL((k_knn<KMAX, FCV, true>), grid, block, knn_lds_bytes(FCV, true, k, parts), st, points, \
                       dim, k, knn_tile(FCV), x, src_ind);                                          \
    DGNN_LAUNCH_CHECK();                                                                            \
    return DGNN_OK;                                                                                 \
  }
  DGNN_KNN_CASE(4)
  DGNN_KNN_CASE(16)
  DGNN_KNN_CASE(32)
  DGNN_KNN_CASE(64)
  DGNN_KNN_CASE(128)
#undef DGNN_KNN_CASE
  if (int rc = knn_allow_lds((const void *)k_knn<KMAX, 128, false>, knn_lds_bytes(128, false, k, parts))) return rc;
  hipLaunchKernelGGL((k_knn<KMAX, 128, false>), grid, block, knn_lds_bytes(128, false, k, parts), st, points, dim,
                     k, knn_tile(128), x, src_ind);
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
  if (k <= 20) return launch_knn<20>(batch, points, dim, k, x, src_ind, st);   // the reference's default k
  if (k <= 24) return launch_knn<24>(batch, points, dim, k, x, src_ind, st);
  if (k <= 32) return launch_knn<32>(batch, points, dim, k, x, src_ind, st);
  if (k <= 40) return launch_knn<40>(batch, points, dim, k, x, src_ind, st);   // its other published k
  if (k <= 48) return launch_knn<48>(batch, points, dim, k, x, src_ind, st);
  return launch_knn<64>(batch, points, dim, k, x, src_ind, st);
}

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

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

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
                                                    int *__restrict__ src_ind,
                                                    const int *__restrict__ block_flags) {
  extern __shared__ __align__(16) float lds[];
  // second stage of the filtered path (below): only the query blocks it could not certify
  if (block_flags != nullptr && block_flags[blockIdx.y * gridDim.x + blockIdx.x] == 0) return;
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
int launch_knn(int batch, int points, int dim, int k, const float *x, int *src_ind, hipStream_t st,
               const int *block_flags = nullptr) {
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
                       dim, k, knn_tile(FCV), x, src_ind, block_flags);                             \
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
                     k, knn_tile(128), x, src_ind, block_flags);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}


// ------------------------------------------------------------------------------------------------
// Filtered path for 17 .. 128 dims: a coarse candidate filter on the matrix cores in front of the
// exact chain.  The ids stay bit-exact: the filter only decides WHICH
// candidates get their exact fma-chain distance, and a query whose result the filter cannot
// certify is redone by k_knn above.
//
//   1. k_knn_split: every point as two bf16 vectors x = xh + xl (+ residual <= 2^-16 |x|), plus
//      nyk = (1 - c) ||x||^2 (float32), per cloud padded to a multiple of 32 points (zero vectors,
//      nyk = +inf).
//   2. k_knn_mfma: a wave owns 32 queries (the B operand of v_mfma_f32_32x32x16_bf16, one query per
//      lane column, held in registers) and walks the cloud in tiles of 32 candidates (A operand,
//      16-byte loads straight from the split arrays in fragment layout: no LDS staging);
//      P = xh.yh + xh.yl + xl.yh accumulates in float32, three MFMAs per 16 dims.  Each lane gets
//      16 candidates of its query per tile and keeps the KL smallest keys
//          key_j = (1 - c) ny_j - 2 P_j       (+ (1 - c) nx = a LOWER BOUND of the exact chain
//                                               distance d_ij: |ny + nx - 2P - d_ij| <= c (nx + ny_j))
//      in a sorted register list (same queue + insertion as the exact kernel); the two lanes of a
//      query (rows 4h .. 4h+3 of every 8) merge at the end.  KL > k leaves slack.
//   3. k_knn_rerank: one lane per query computes the exact chain distance of its KL candidates,
//      keeps the k best by (distance, id), and certifies the result: every candidate outside the
//      list has chain distance >= key_last + (1 - c) nx; if the k-th exact distance is strictly
//      below that bound nobody outside can enter or tie.  Otherwise (ties, duplicates, tiny gaps,
//      non-finite input) the query's 64-query block is flagged and k_knn recomputes it.
// c = 2^-11 covers, with a factor > 2 to spare, the split residual (3.1 * 2^-16 (nx+ny)), the
// float32 accumulation of 3 * dim products (<= 3 * 2^-16 at 128 dims), the float32 norms and the
// rounding of the exact chain itself ((dim + 2) * 2^-24 * d, d <= 2 (nx + ny)).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr float KNN_C = 1.0f / 2048.0f;
constexpr int KNN_MQ = 32;            // queue entries per lane in k_knn_mfma

__device__ __forceinline__ unsigned short bf16_rne(float v) {        // finite or not: NaN stays NaN
  unsigned u = __float_as_uint(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// Distances do not change when a cloud is shifted, the filter's slack c (nx + ny) does: feature
// clouds (after an activation every coordinate is positive) sit far from the origin compared with
// their neighbour distances, and the bound then certifies little (the DGCNN layers at 128 dims spent
// as long in k_knn_redo as in the filter).  The filter therefore works on x' = fl(x - mu) with mu an
// average of the cloud (ANY mu is valid; about 64 points at a fixed stride are averaged).  The exact
// chain keeps the caller's x.  What the rounding of the subtraction adds: x'_i - x'_j differs from
// x_i - x_j by at most 2^-24 (|x'_i| + |x'_j|) per vector, so |d - d'| <= 2^-22 (nx' + ny'): three
// orders of magnitude inside the spare factor of c.
constexpr int KNN_CENTER_SAMPLE = 64;
__global__ __launch_bounds__(256) void k_knn_center(int points, int dim, int dimp, const float *__restrict__ x,
                                                    float *__restrict__ mean) {
  __shared__ float part[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int f = tid % dimp, rp = tid / dimp, rows_par = 256 / dimp;     // dimp is 32, 64 or 128
  const int stride = points > KNN_CENTER_SAMPLE ? points / KNN_CENTER_SAMPLE : 1;
  const int ns = (points + stride - 1) / stride;                        // sampled points
  const float *xc = x + (size_t)b * points * dim;
  float s = 0.f;
  if (f < dim) {
    float s4[4] = {0.f, 0.f, 0.f, 0.f};                                 // four loads in flight
    int i = rp;
    for (; i + 3 * rows_par < ns; i += 4 * rows_par) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s4[u] += xc[(size_t)(i + u * rows_par) * stride * dim + f];
    }
    for (; i < ns; i += rows_par) s4[0] += xc[(size_t)i * stride * dim + f];
    s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
  }
  part[tid] = s;
  __syncthreads();
  if (rp == 0) {
    for (int q = 1; q < rows_par; ++q) s += part[q * dimp + f];
    mean[(size_t)b * dimp + f] = f < dim ? s / (float)ns : 0.f;
  }
}

// one thread per (padded point, group of 8 dims); the thread of group 0 also writes nyk
__global__ void k_knn_split(int points, int pp, int dim, int dimp, const float *__restrict__ x,
                            const float *__restrict__ mean,
                            unsigned short *__restrict__ xh, unsigned short *__restrict__ xl,
                            float *__restrict__ nyk, int *__restrict__ cloud_bad) {
  const int groups = dimp >> 3;
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= (long long)pp * groups) return;
  const int j = (int)(t / groups), g = (int)(t - (long long)j * groups);
  const float *row = x + ((size_t)b * points + (j < points ? j : 0)) * dim;
  unsigned short hi[8], lo[8];
  float vv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int f = g * 8 + e;
    const float v = (j < points && f < dim) ? row[f] - (mean != nullptr ? mean[(size_t)b * dimp + f] : 0.f) : 0.f;
    vv[e] = v;
    hi[e] = bf16_rne(v);
    lo[e] = bf16_rne(v - __uint_as_float((unsigned)hi[e] << 16));
  }
  const size_t at = ((size_t)b * pp + j) * dimp + (size_t)g * 8;
  *(uint4 *)(xh + at) = make_uint4(hi[0] | (unsigned)hi[1] << 16, hi[2] | (unsigned)hi[3] << 16,
                                   hi[4] | (unsigned)hi[5] << 16, hi[6] | (unsigned)hi[7] << 16);
  *(uint4 *)(xl + at) = make_uint4(lo[0] | (unsigned)lo[1] << 16, lo[2] | (unsigned)lo[3] << 16,
                                   lo[4] | (unsigned)lo[5] << 16, lo[6] | (unsigned)lo[7] << 16);
  // |x|^2: every thread squares its 8 dims, the row's threads (4, 8 or 16 consecutive lanes) add up
  float part = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) part = fmaf(vv[e], vv[e], part);
  for (int off = 1; off < groups; off <<= 1) part += __shfl_xor(part, off, 64);
  if (g == 0) {
    float ny = __builtin_inff();
    if (j < points) {
      ny = part * (1.f - KNN_C);
      if (!(ny < __builtin_inff())) cloud_bad[b] = 1;      // inf or NaN: the filter certifies nothing here
    }
    nyk[(size_t)b * pp + j] = ny;
  }
}

template <int KL, int NKK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KL <= 32 ? 2 : 1, KL <= 32 ? 2 : 1)))
void k_knn_mfma(int points, int pp, long long nq, const unsigned short *__restrict__ xh,
                const unsigned short *__restrict__ xl, const float *__restrict__ nyk,
                int *__restrict__ cand, float *__restrict__ key_last, int xcd) {
  extern __shared__ __align__(16) float lds[];
  constexpr int DIMP = NKK * 16;
  constexpr int PER_WAVE = (KL * 32 * 2 > KNN_MQ * 64 * 2) ? KL * 32 * 2 : KNN_MQ * 64 * 2;   // floats
  const int wave = wave_in_block(), lane = lane_id();
  const int r = lane & 31, h = lane >> 5;
  // a cloud's query blocks on one XCD: they all walk the same candidates (DGNN_KNN_XCD=0: launch order)
  const unsigned lid = xcd ? xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y)
                           : blockIdx.y * gridDim.x + blockIdx.x;
  const int b = uniform_i32((int)(lid / gridDim.x)), bx = uniform_i32((int)(lid % gridDim.x));
  const int q0 = uniform_i32((bx * 4 + wave) * 32);
  if (q0 >= points) return;                                  // no block-wide barrier below
  float *qd = lds + (size_t)wave * PER_WAVE;                 // [KNN_MQ][64] keys
  int *qi = (int *)(qd + KNN_MQ * 64);                       // [KNN_MQ][64] ids
  const size_t base = (size_t)b * pp;
  const bf16x8 *gh = (const bf16x8 *)xh, *gl = (const bf16x8 *)xl;   // 8 bf16 = 16 bytes
  // fragment element index of (row, k-step kk, half h): (row * DIMP + 16 kk + 8 h) / 8
  auto frag = [&](size_t row, int kk) { return row * (DIMP / 8) + 2 * kk + h; };
  bf16x8 bh[NKK], bl[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    bh[kk] = gh[frag(base + q0 + r, kk)];
    bl[kk] = gl[frag(base + q0 + r, kk)];
  }
  TopK<KL, true> best;
  best.init();
  int queued = 0;
  float thr = __builtin_inff();
  // One flush site per tile (a tile hands every lane 16 candidates): the queue holds KNN_MQ >= 32
  // entries per lane and is emptied before a tile that might overflow it.  The acceptance
  // threshold `thr` is refreshed after every flush (stale in between = a few more queue entries).
  auto flush = [&]() {
    for (int t = 0; t < KNN_MQ; ++t) {
      if (!__any(t < queued)) break;
      const bool have = t < queued;
      best.push(have ? qd[t * 64 + lane] : __builtin_inff(), have ? qi[t * 64 + lane] : -1);
    }
    queued = 0;
  };
  auto load = [&](bf16x8 (&ah)[NKK], bf16x8 (&al)[NKK], int j0) {
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      ah[kk] = gh[frag(base + j0 + r, kk)];
      al[kk] = gl[frag(base + j0 + r, kk)];
    }
  };
  auto tile = [&](const bf16x8 (&ah)[NKK], const bf16x8 (&al)[NKK], int j0) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk], bh[kk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk], bl[kk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk], bh[kk], acc, 0, 0, 0);
    }
    if (__any(queued > KNN_MQ - 16)) {
      flush();
      // the two lanes of a query share their bound: a key at or above the KL-th smallest key of
      // BOTH lists together cannot be among the KL smallest of the union later either
      // exactly: the KL-th smallest key of the two lists together = min over s of
      // max(A[s-1], B[KL-s-1]) (s entries from this lane's sorted list A, KL - s from the other's B)
      float u = __shfl_xor(best.d[KL - 1], 32, 64);          // s = 0
#pragma unroll
      for (int sidx = 1; sidx < KL; ++sidx) {
        const float bv = __shfl_xor(best.d[KL - sidx - 1], 32, 64);
        u = fminf(u, fmaxf(best.d[sidx - 1], bv));
      }
      thr = fminf(u, best.d[KL - 1]);                         // s = KL
    }
    // register 4g + e = candidate row 8g + 4h + e of the tile, this lane's query column
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 n4 = *(const float4 *)(nyk + base + j0 + 8 * g + 4 * h);
      const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float key = fmaf(-2.f, acc[4 * g + e], nn[e]);
        if (key < thr) {
          qd[queued * 64 + lane] = key;
          qi[queued * 64 + lane] = j0 + 8 * g + 4 * h + e;
          ++queued;
        }
      }
    }
  };
  // two tiles per trip: the fragments of the next tile load while this one multiplies, and the
  // two register sets swap roles without copies
  if constexpr (NKK <= 4) {
    bf16x8 a0h[NKK], a0l[NKK], a1h[NKK], a1l[NKK];
    load(a0h, a0l, 0);
    for (int j0 = 0; j0 < pp; j0 += 64) {
      const bool two = j0 + 32 < pp;
      if (two) load(a1h, a1l, j0 + 32);
      tile(a0h, a0l, j0);
      if (two) {
        if (j0 + 64 < pp) load(a0h, a0l, j0 + 64);
        tile(a1h, a1l, j0 + 32);
      }
    }
  } else {           // 128 dims: one fragment set (two would not fit two waves per SIMD)
    bf16x8 ah[NKK], al[NKK];
    for (int j0 = 0; j0 < pp; j0 += 32) {
      load(ah, al, j0);
      tile(ah, al, j0);
    }
  }
  flush();
  // the two lanes of a query: the upper half hands its list over through LDS
  wave_lds_sync();
  float *md = lds + (size_t)wave * PER_WAVE;                 // [KL][32]
  int *mi = (int *)(md + KL * 32);
  if (h == 1) {
#pragma unroll
    for (int t = 0; t < KL; ++t) {
      md[t * 32 + r] = best.d[t];
      mi[t * 32 + r] = best.id[t];
    }
  }
  wave_lds_sync();
  if (h == 0) {
    for (int t = 0; t < KL; ++t) {
      const int jn = mi[t * 32 + r];
      const float dn = md[t * 32 + r];
      if (__any(jn >= 0)) best.push(jn >= 0 ? dn : __builtin_inff(), jn);
    }
    if (q0 + r < points) {
      const long long gq = (long long)b * points + q0 + r;
#pragma unroll
      for (int t = 0; t < KL; ++t) cand[(size_t)t * nq + gq] = best.id[t];
      key_last[gq] = best.d[KL - 1];
    }
  }
}

// The same pass for clouds of at most 2048 points (the DGCNN sizes), with the sorted list as ONE
// 32-bit word per entry: the lower bound LB = max(0, key + (1 - c) nx) >= 0 orders like its bit
// pattern, its low 11 mantissa bits are replaced by the candidate's index in the cloud (LB rounded
// DOWN by at most 2^-12 LB: still a lower bound), and an insertion is one v_med3_u32 per slot
// instead of five instructions on (key, id) pairs -- the insertion is what this kernel spends its
// time on.  Words at or above the bits of +inf mean "no entry".
constexpr unsigned KNN_PK_IDX = 0x7ffu;
constexpr unsigned KNN_PK_INF = 0x7f800000u;
template <int KL>
struct TopKPacked {
  unsigned v[KL];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < KL; ++t) v[t] = 0xffffffffu;
  }
  __device__ __forceinline__ void push(unsigned x) {
    if (x < v[KL - 1]) {
      unsigned prev = 0u;
#pragma unroll
      for (int t = 0; t < KL; ++t) {       // sorted, prev <= old: the median of (prev, x, old)
        const unsigned old = v[t];
        v[t] = min(max(prev, x), old);
        prev = old;
      }
    }
  }
};

// (Measured and dropped: the block's four waves sharing every candidate tile through a
// double-buffered LDS stage instead of each wave loading its fragments from L2 -- 204 -> 226 us at
// 64 dims: the barriers cost more than the 4x lower L2 traffic saves.)
template <int KL, int NKK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_knn_mfma_packed(int points, int pp, long long nq, const unsigned short *__restrict__ xh,
                       const unsigned short *__restrict__ xl, const float *__restrict__ nyk,
                       int *__restrict__ cand, float *__restrict__ key_last, int xcd) {
  extern __shared__ __align__(16) float lds[];
  constexpr int DIMP = NKK * 16;
  constexpr int PER_WAVE = (KL * 32 > KNN_MQ * 64) ? KL * 32 : KNN_MQ * 64;        // words
  const int wave = wave_in_block(), lane = lane_id();
  const int r = lane & 31, h = lane >> 5;
  // a cloud's query blocks on one XCD: they all walk the same candidates (DGNN_KNN_XCD=0: launch order)
  const unsigned lid = xcd ? xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y)
                           : blockIdx.y * gridDim.x + blockIdx.x;
  const int b = uniform_i32((int)(lid / gridDim.x)), bx = uniform_i32((int)(lid % gridDim.x));
  const int q0 = uniform_i32((bx * 4 + wave) * 32);
  if (q0 >= points) return;                                  // no block-wide barrier below
  unsigned *qv = (unsigned *)lds + (size_t)wave * PER_WAVE;  // [KNN_MQ][64] queued words
  const size_t base = (size_t)b * pp;
  const bf16x8 *gh = (const bf16x8 *)xh, *gl = (const bf16x8 *)xl;
  auto frag = [&](size_t row, int kk) { return row * (DIMP / 8) + 2 * kk + h; };
  bf16x8 bh[NKK], bl[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    bh[kk] = gh[frag(base + q0 + r, kk)];
    bl[kk] = gl[frag(base + q0 + r, kk)];
  }
  const float nxk = nyk[base + q0 + r];                      // (1 - c) |x|^2 of this lane's query
  TopKPacked<KL> best;
  best.init();
  int queued = 0;
  unsigned thr = KNN_PK_INF;                                 // +inf and NaN bounds never enter
  auto flush = [&]() {
    // the next queued word is read while this one is inserted
    unsigned nxt = queued > 0 ? qv[lane] : 0xffffffffu;
    for (int t = 0; t < KNN_MQ; ++t) {
      if (!__any(t < queued)) break;
      const unsigned cur = nxt;
      nxt = (t + 1 < queued) ? qv[(t + 1) * 64 + lane] : 0xffffffffu;
      best.push(cur);
    }
    queued = 0;
  };
  auto load = [&](bf16x8 (&ah)[NKK], bf16x8 (&al)[NKK], int j0) {
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      ah[kk] = gh[frag(base + j0 + r, kk)];
      al[kk] = gl[frag(base + j0 + r, kk)];
    }
  };
  // One fragment set: the loads of the NEXT tile are issued right behind this tile's MFMAs (which
  // have read their operands by then) and land while the selection below runs.
  bf16x8 ah[NKK], al[NKK];
  auto tile = [&](int j0, int jnext) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk], bh[kk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk], bl[kk], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[kk], bh[kk], acc, 0, 0, 0);
    }
    load(ah, al, jnext);
    if (__any(queued > KNN_MQ - 17)) {
      flush();
      // The two lanes of a query share their bound.  The KL-th smallest word of both lists
      // together is at most max(A[s-1], B[KL-s-1]) for every split s; s = 0, KL / 2 and KL (the
      // lists are statistically alike, so the middle split is nearly the exact value; all KL + 1
      // splits cost 24 cross-half moves per flush and gained nothing measurable).
      const unsigned b_last = __shfl_xor(best.v[KL - 1], 32, 64);
      const unsigned b_mid = __shfl_xor(best.v[KL / 2 - 1], 32, 64);
      const unsigned u = min(min(b_last, best.v[KL - 1]), max(best.v[KL - KL / 2 - 1], b_mid));
      thr = min(u, KNN_PK_INF);
    }
    // every word is written to the lane's next queue slot; the slot only counts if it passes
    // (no exec-mask work per candidate); a tile adds at most 16, the flush above left room
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 n4 = *(const float4 *)(nyk + base + j0 + 8 * g + 4 * h);
      const float nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lb = fmaxf(fmaf(-2.f, acc[4 * g + e], nn[e]) + nxk, 0.f);
        const unsigned w = (__float_as_uint(lb) & ~KNN_PK_IDX) | (unsigned)(j0 + 8 * g + 4 * h + e);
        qv[queued * 64 + lane] = w;
        queued += (w < thr) ? 1 : 0;
      }
    }
  };
  // (Measured without effect: waves of a cloud starting at different tiles so that they do not ask
  // the same L2 lines at the same time; a second fragment set loaded a whole tile ahead.  What grows
  // with the dimension is the fragment traffic through each CU's vector memory pipe: 16 KB per wave
  // and tile at 128 dims.)
  load(ah, al, 0);
  for (int j0 = 0; j0 < pp; j0 += 32) tile(j0, j0 + 32 < pp ? j0 + 32 : 0);
  flush();
  wave_lds_sync();
  unsigned *mv = qv;                                         // [KL][32], over the (empty) queue
  if (h == 1) {
#pragma unroll
    for (int t = 0; t < KL; ++t) mv[t * 32 + r] = best.v[t];
  }
  wave_lds_sync();
  if (h == 0) {
    for (int t = 0; t < KL; ++t) best.push(mv[t * 32 + r]);
    if (q0 + r < points) {
      const long long gq = (long long)b * points + q0 + r;
#pragma unroll
      for (int t = 0; t < KL; ++t)
        cand[(size_t)t * nq + gq] = best.v[t] < KNN_PK_INF ? (int)(best.v[t] & KNN_PK_IDX) : -1;
      // k_knn_rerank adds (1 - c) nx back: hand it the bound in the unpacked kernel's form
      key_last[gq] = best.v[KL - 1] < KNN_PK_INF ? __uint_as_float(best.v[KL - 1] & ~KNN_PK_IDX) - nxk
                                                 : __builtin_inff();
    }
  }
}

// exact (distance, id) order for candidates that arrive in any order
template <int KMAX>
struct TopKLex {
  float d[KMAX];
  int id[KMAX];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { d[t] = __builtin_inff(); id[t] = -1; }
  }
  __device__ __forceinline__ void push(float dn, int jn) {
    bool sw = false;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      const float td = d[t];
      const int ti = id[t];
      sw = sw || dn < td || (dn == td && jn >= 0 && ti >= 0 && jn < ti);
      d[t] = sw ? dn : td;
      id[t] = sw ? jn : ti;
      dn = sw ? td : dn;
      jn = sw ? ti : jn;
    }
  }
};

// A block = 64 queries (lane = query) x 4 waves; wave w evaluates the exact chain for list entries
// w, w + 4, ...  A lane reading its own candidate row straight from memory makes every load
// instruction touch 64 different lines: instead the wave copies the 64 rows of one list entry, 32
// dims at a time, into a private LDS tile with coalesced 16-byte loads (8 rows x 128 B per
// instruction; the next chunk's loads are in flight while this one is consumed) and every lane
// reads ITS row from the tile.  The chain is carried across the chunks in a register, so the order
// of operations is the exhaustive kernel's.  Selection is by RANK, not by insertion: entry t goes
// to output slot #{u : (d_u, id_u) < (d_t, id_t)} -- kl^2 compares spread over the four waves.
constexpr int KNN_CH = 32;                // dims per staged chunk
constexpr int KNN_RS = KNN_CH + 4;        // tile row stride (floats): 16-byte reads of 16 lanes hit 16 bank groups
constexpr int KNN_REDO_MAX = 4;           // flagged queries per 64-query block the wave-per-query pass takes

// dims [f0, f0 + 32) of the rows `row_of_lane` (lane r names row r of the tile): global -> registers
__device__ __forceinline__ void knn_fetch(float (&regs)[32], const float *xc, int row_of_lane, int dim, int f0,
                                          bool vec4, int lane) {
  if (vec4) {
    const int p = lane & 7;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = __shfl(row_of_lane, (lane >> 3) + 8 * m, 64);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f0 + 4 * p < dim) v = *(const float4 *)(xc + (size_t)j * dim + f0 + 4 * p);
      regs[4 * m + 0] = v.x; regs[4 * m + 1] = v.y; regs[4 * m + 2] = v.z; regs[4 * m + 3] = v.w;
    }
  } else {
    const int e = lane & 31;
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const int j = __shfl(row_of_lane, (lane >> 5) + 2 * m, 64);
      regs[m] = (f0 + e < dim) ? xc[(size_t)j * dim + f0 + e] : 0.f;
    }
  }
}
// registers -> tile[64][KNN_RS]
__device__ __forceinline__ void knn_put(float *tile, const float (&regs)[32], bool vec4, int lane) {
  if (vec4) {
    const int p = lane & 7;
#pragma unroll
    for (int m = 0; m < 8; ++m)
      *(float4 *)(tile + ((lane >> 3) + 8 * m) * KNN_RS + 4 * p) =
          make_float4(regs[4 * m + 0], regs[4 * m + 1], regs[4 * m + 2], regs[4 * m + 3]);
  } else {
    const int e = lane & 31;
#pragma unroll
    for (int m = 0; m < 32; ++m) tile[((lane >> 5) + 2 * m) * KNN_RS + e] = regs[m];
  }
}

template <int FC>
__global__ __launch_bounds__(256)
void k_knn_rerank(int points, int pp, int dim, int k, int kl, long long nq, const float *__restrict__ x,
                  const float *__restrict__ nyk, const int *__restrict__ cand, float *__restrict__ key_last,
                  const int *__restrict__ cloud_bad, int *__restrict__ src_ind, int *__restrict__ redo_flags,
                  int *__restrict__ block_flags, int *__restrict__ query_flags, int redo_max, int xcd) {
  extern __shared__ __align__(16) float lds[];
  const int lane = lane_id(), wave = wave_in_block();
  // as in the filter kernels: a cloud's blocks share an XCD's L2 (their candidate rows overlap)
  const unsigned lid = xcd ? xcd_contiguous_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y)
                           : blockIdx.y * gridDim.x + blockIdx.x;
  const int b = uniform_i32((int)(lid / gridDim.x)), bx = uniform_i32((int)(lid % gridDim.x));
  float *tile = lds + (size_t)wave * 64 * KNN_RS;             // wave-private [64][KNN_RS]
  float *dl = lds + (size_t)4 * 64 * KNN_RS;                  // [kl][64] exact distances (+inf = no entry)
  int *il = (int *)(dl + (size_t)kl * 64);                    // [kl][64] ids
  float *dks = (float *)(il + (size_t)kl * 64);               // [64] k-th exact distance
  const int i = bx * 64 + lane;
  const bool active = i < points;
  const float *xc = x + (size_t)b * points * dim;
  const long long gq = (long long)b * points + (active ? i : 0);
  const bool vec4 = (dim & 3) == 0 && ((size_t)x & 15) == 0;
  constexpr int NCH = FC / KNN_CH;
  const int nch = (dim + KNN_CH - 1) / KNN_CH;                // chunks that hold data
  float regs[32];
  float xi[FC];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c < nch) {
      knn_fetch(regs, xc, active ? i : 0, dim, c * KNN_CH, vec4, lane);
      knn_put(tile, regs, vec4, lane);
      wave_lds_sync();
#pragma unroll
      for (int f4 = 0; f4 < KNN_CH / 4; ++f4) {
        const float4 v = *(const float4 *)(tile + lane * KNN_RS + 4 * f4);
        xi[c * KNN_CH + 4 * f4 + 0] = v.x; xi[c * KNN_CH + 4 * f4 + 1] = v.y;
        xi[c * KNN_CH + 4 * f4 + 2] = v.z; xi[c * KNN_CH + 4 * f4 + 3] = v.w;
      }
      wave_lds_sync();
    } else {
#pragma unroll
      for (int f = 0; f < KNN_CH; ++f) xi[c * KNN_CH + f] = 0.f;
    }
  }
  // the wave's list entries t = wave, wave + 4, ...; chunks in (t, c) order, one fetch ahead
  int jt = (active && wave < kl) ? cand[(size_t)wave * nq + gq] : -1;
  if (wave < kl) knn_fetch(regs, xc, jt >= 0 ? jt : 0, dim, 0, vec4, lane);
  for (int t = wave; t < kl; t += 4) {
    const int tn = t + 4;
    const int jn = (active && tn < kl) ? cand[(size_t)tn * nq + gq] : -1;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c < nch) {
        knn_put(tile, regs, vec4, lane);
        wave_lds_sync();
        if (c + 1 < nch) knn_fetch(regs, xc, jt >= 0 ? jt : 0, dim, (c + 1) * KNN_CH, vec4, lane);
        else if (tn < kl) knn_fetch(regs, xc, jn >= 0 ? jn : 0, dim, 0, vec4, lane);
#pragma unroll
        for (int f4 = 0; f4 < KNN_CH / 4; ++f4) {
          const float4 v = *(const float4 *)(tile + lane * KNN_RS + 4 * f4);
          float u;       // padded dims hold 0 on both sides: fma(0,0,acc) == acc exactly
          u = xi[c * KNN_CH + 4 * f4 + 0] - v.x; acc = fmaf(u, u, acc);
          u = xi[c * KNN_CH + 4 * f4 + 1] - v.y; acc = fmaf(u, u, acc);
          u = xi[c * KNN_CH + 4 * f4 + 2] - v.z; acc = fmaf(u, u, acc);
          u = xi[c * KNN_CH + 4 * f4 + 3] - v.w; acc = fmaf(u, u, acc);
        }
        wave_lds_sync();
      }
    }
    const bool ok = jt >= 0 && acc < __builtin_inff();          // NaN / inf: never a neighbour
    dl[t * 64 + lane] = ok ? acc : __builtin_inff();
    il[t * 64 + lane] = ok ? jt : -1;
    jt = jn;
  }
  __syncthreads();
  // rank of every entry among the kl of its query; entries without a candidate (d = +inf) rank
  // behind all others in list order, so slots 0 .. k-1 are each written exactly once
  {
    int *o = src_ind + ((size_t)b * points + (active ? i : 0)) * k;
    const int gbase = b * points;
    for (int t = wave; t < kl; t += 4) {
      const float dt = dl[t * 64 + lane];
      const int idt = il[t * 64 + lane];
      int rank = 0;
      for (int u = 0; u < kl; ++u) {
        const float du = dl[u * 64 + lane];
        const int idu = il[u * 64 + lane];
        const bool less = du < dt || (du == dt && (idt >= 0 ? idu < idt : u < t));
        rank += less ? 1 : 0;
      }
      if (active && rank < k) o[rank] = gbase + (idt >= 0 ? idt : i);   // unfilled: the query itself
      if (rank == k - 1) dks[lane] = dt;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  bool safe = true;
  if (active) {
    const float dk = dks[lane];
    const float kl_key = key_last[gq], nx = nyk[(size_t)b * pp + i];
    if (kl_key == __builtin_inff()) {
      safe = true;                      // the list never filled: every candidate with a finite key is in it
    } else {
      const float bound = kl_key + nx;
      const float slack = 3.8e-6f * (fabsf(kl_key) + nx);          // 2^-18: rounding of key and bound
      safe = dk < bound - slack;                                   // false for NaN
    }
    if (cloud_bad[b] != 0) safe = false;
    query_flags[gq] = safe ? 0 : 1;
    key_last[gq] = dk;                  // for k_knn_redo: every true neighbour is at most this far
  }
  const int nbad = __popcll(__ballot(!safe));
  if (lane == 0) {
    const size_t blk = (size_t)b * gridDim.x + bx;
    redo_flags[blk] = (nbad > 0 && nbad <= redo_max) ? 1 : 0;      // few: k_knn_redo, query by query
    block_flags[blk] = nbad > redo_max ? 1 : 0;                    // many (ties everywhere): k_knn
  }
}

// The few queries the filter could not certify, one BLOCK pass per query: the list's k-th exact
// distance bounds every true neighbour, so the block's 256 lanes evaluate the exact chain for all of the
// cloud's candidates, keep those within the bound (k plus a handful), and rank them by (distance,
// id).  More survivors than the buffer holds (a cloud of exact ties) hands the block to k_knn.
constexpr int KNN_REDO_CAP = 1024;
template <int FC>
__global__ __launch_bounds__(256)
void k_knn_redo(int points, int dim, int k, const float *__restrict__ x, const int *__restrict__ redo_flags,
                const int *__restrict__ query_flags, const float *__restrict__ dk_bound,
                int *__restrict__ block_flags, int *__restrict__ src_ind) {
  __shared__ float sd[KNN_REDO_CAP];
  __shared__ int sj[KNN_REDO_CAP];
  __shared__ int cnt;
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (redo_flags[blk] == 0) return;
  constexpr int NT = 256;
  const int tid = threadIdx.x, b = blockIdx.y;
  const float *xc = x + (size_t)b * points * dim;
  const bool vec4 = (dim & 3) == 0 && ((size_t)x & 15) == 0;
  // the block's 64 flags in one load (one dependent load per query cost 0.5 us each: most of this
  // kernel's time when a handful of queries is flagged)
  __shared__ unsigned long long flagged;
  if (tid < 64) {
    const int i = blockIdx.x * 64 + tid;
    const unsigned long long m = __ballot(i < points && query_flags[(size_t)b * points + min(i, points - 1)] != 0);
    if (tid == 0) flagged = m;
  }
  __syncthreads();
  unsigned long long todo = flagged;
  while (todo != 0) {
    const int q = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int i = blockIdx.x * 64 + q;
    const size_t gq = (size_t)b * points + i;
    if (tid == 0) cnt = 0;
    __syncthreads();
    const float dk = dk_bound[gq];
    float xi[FC];
#pragma unroll
    for (int f = 0; f < FC; ++f) xi[f] = (f < dim) ? xc[(size_t)i * dim + f] : 0.f;
    // two candidates per trip, their row loads in flight together (the pass is latency-bound)
    for (int c = tid; c < points; c += 2 * NT) {
      const int c2 = c + NT;
      const bool two = c2 < points;
      const float *cj = xc + (size_t)c * dim, *cj2 = xc + (size_t)(two ? c2 : c) * dim;
      float acc = 0.f, acc2 = 0.f;
      if (vec4) {
        float4 v[FC / 4], v2[FC / 4];
#pragma unroll
        for (int f4 = 0; f4 < FC / 4; ++f4) {
          if (4 * f4 < dim) {
            v[f4] = *(const float4 *)(cj + 4 * f4);
            v2[f4] = *(const float4 *)(cj2 + 4 * f4);
          }
        }
#pragma unroll
        for (int f4 = 0; f4 < FC / 4; ++f4) {
          if (4 * f4 < dim) {
            float u;
            u = xi[4 * f4 + 0] - v[f4].x; acc = fmaf(u, u, acc);
            u = xi[4 * f4 + 1] - v[f4].y; acc = fmaf(u, u, acc);
            u = xi[4 * f4 + 2] - v[f4].z; acc = fmaf(u, u, acc);
            u = xi[4 * f4 + 3] - v[f4].w; acc = fmaf(u, u, acc);
            u = xi[4 * f4 + 0] - v2[f4].x; acc2 = fmaf(u, u, acc2);
            u = xi[4 * f4 + 1] - v2[f4].y; acc2 = fmaf(u, u, acc2);
            u = xi[4 * f4 + 2] - v2[f4].z; acc2 = fmaf(u, u, acc2);
            u = xi[4 * f4 + 3] - v2[f4].w; acc2 = fmaf(u, u, acc2);
          }
        }
      } else {
#pragma unroll
        for (int f = 0; f < FC; ++f) {
          if (f < dim) {
            float u = xi[f] - cj[f];
            acc = fmaf(u, u, acc);
            u = xi[f] - cj2[f];
            acc2 = fmaf(u, u, acc2);
          }
        }
      }
      if (acc <= dk && acc < __builtin_inff()) {              // NaN / inf: never a neighbour
        const int slot = atomicAdd(&cnt, 1);
        if (slot < KNN_REDO_CAP) { sd[slot] = acc; sj[slot] = c; }
      }
      if (two && acc2 <= dk && acc2 < __builtin_inff()) {
        const int slot = atomicAdd(&cnt, 1);
        if (slot < KNN_REDO_CAP) { sd[slot] = acc2; sj[slot] = c2; }
      }
    }
    __syncthreads();
    const int m = cnt;
    if (m > KNN_REDO_CAP) {              // block-uniform: the exhaustive kernel takes the whole block
      if (tid == 0) block_flags[blk] = 1;
      return;
    }
    int *o = src_ind + gq * k;
    const int gbase = b * points;
    for (int t = tid; t < m; t += NT) {
      const float dt = sd[t];
      const int jt = sj[t];
      int rank = 0;
      for (int u = 0; u < m; ++u) {
        const float du = sd[u];
        const int ju = sj[u];
        rank += (du < dt || (du == dt && ju < jt)) ? 1 : 0;
      }
      if (rank < k) o[rank] = gbase + jt;
    }
    for (int t = m + tid; t < k; t += NT) o[t] = gbase + i;     // fewer finite candidates than k
    __syncthreads();
  }
}

struct KnnWs {
  size_t off_xh, off_xl, off_nyk, off_cand, off_key, off_bad, off_flags, off_qflags, off_rflags, off_mean, total;
  int pp, dimp, kl;
};
// list length of the filter: k + at least 4 spare candidates.  Cost grows with KL^2 (accepted
// candidates ~ KL ln(N / KL), each a pass over KL slots); a thin margin only means more query
// blocks handed to the exhaustive kernel on data with near-ties.
static inline int knn_filter_list(int k) { return k <= 20 ? 24 : (k <= 28 ? 32 : (k <= 44 ? 48 : (k <= 60 ? 64 : 0))); }
static KnnWs knn_ws_layout(int batch, int points, int dim, int k) {
  KnnWs w = {};
  w.kl = knn_filter_list(k);
  w.pp = (points + 31) / 32 * 32;
  w.dimp = dim <= 32 ? 32 : (dim <= 64 ? 64 : 128);     // k-steps of 16: 2, 4 or 8 per tile
  const size_t nq = (size_t)batch * points, np = (size_t)batch * w.pp;
  size_t o = 0;
  auto seg = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
  w.off_xh = seg(np * w.dimp * 2);
  w.off_xl = seg(np * w.dimp * 2);
  w.off_nyk = seg(np * 4);
  w.off_cand = seg(nq * (size_t)w.kl * 4);
  w.off_key = seg(nq * 4);
  w.off_bad = seg((size_t)batch * 4);
  w.off_flags = seg((size_t)batch * ((points + 63) / 64) * 4);
  w.off_qflags = seg(nq * 4);
  w.off_rflags = seg((size_t)batch * ((points + 63) / 64) * 4);
  w.off_mean = seg((size_t)batch * w.dimp * 4);
  w.total = o;
  return w;
}
// The filter pays when the exact chain dominates (more than 16 dims) and the problem fills the
// chip with 32-query waves; everything else stays on the exact kernel alone.
static bool knn_filter_applies(int batch, int points, int dim, int k) {
  if (env_int("DGNN_KNN_FILTER", 1) == 0) return false;
  if (dim <= 16 || dim > 128 || knn_filter_list(k) == 0) return false;
  // long lists as (key, id) pairs run one wave per SIMD: there the chain must dominate; packed words
  // (clouds of at most 2048 points, lists up to 48) keep two
  const bool packed_ok = points <= 2048 && knn_filter_list(k) <= 48 && env_int("DGNN_KNN_PACKED", 1) != 0;
  if (knn_filter_list(k) > 32 && dim <= 64 && !packed_ok) return false;
  if (points < 2 * knn_filter_list(k)) return false;
  return (long long)batch * ((points + 31) / 32) >= env_int("DGNN_KNN_FILTER_MIN_WAVES", 512);
}

template <int KL>
static int launch_knn_mfma_packed(int nkk, dim3 grid, hipStream_t st, int points, int pp, long long nq,
                                  const unsigned short *xh, const unsigned short *xl, const float *nyk, int *cand,
                                  float *key_last) {
  const size_t queue = (size_t)4 * ((KL * 32 > KNN_MQ * 64) ? KL * 32 : KNN_MQ * 64) * sizeof(unsigned);
#define DGNN_KNN_MFMA_CASE(N)                                                                              \
  if (nkk == N) {                                                                                          \
    const size_t lds = queue;                                                                              \
    hipLaunchKernelGGL((k_knn_mfma_packed<KL, N>), grid, dim3(256), lds, st, points, pp, nq, xh, xl, nyk,  \
                       cand, key_last, env_int("DGNN_KNN_XCD", 1));                                        \
    return DGNN_OK;                                                                                        \
  }
  DGNN_KNN_MFMA_CASE(2)
  DGNN_KNN_MFMA_CASE(4)
  DGNN_KNN_MFMA_CASE(8)
#undef DGNN_KNN_MFMA_CASE
  return DGNN_ERR_UNSUPPORTED;
}

template <int KL>
static int launch_knn_mfma(int nkk, dim3 grid, size_t lds, hipStream_t st, int points, int pp, long long nq,
                           const unsigned short *xh, const unsigned short *xl, const float *nyk, int *cand,
                           float *key_last) {
#define DGNN_KNN_MFMA_CASE(N)                                                                              \
  if (nkk == N) {                                                                                          \
    if (int rc = knn_allow_lds((const void *)k_knn_mfma<KL, N>, lds)) return rc;                           \
    hipLaunchKernelGGL((k_knn_mfma<KL, N>), grid, dim3(256), lds, st, points, pp, nq, xh, xl, nyk, cand,   \
                       key_last, env_int("DGNN_KNN_XCD", 1));                                              \
    return DGNN_OK;                                                                                        \
  }
  DGNN_KNN_MFMA_CASE(2)
  DGNN_KNN_MFMA_CASE(4)
  DGNN_KNN_MFMA_CASE(8)
#undef DGNN_KNN_MFMA_CASE
  return DGNN_ERR_UNSUPPORTED;
}

static int launch_knn_rerank(int batch, int points, int pp, int dim, int k, int kl, const float *x, const float *nyk,
                             const int *cand, float *key_last, const int *bad, int *src_ind, int *rflags, int *flags,
                             int *qflags, hipStream_t st) {
  const dim3 grid((unsigned)((points + 63) / 64), (unsigned)batch);
  const long long nq = (long long)batch * points;
  const size_t lds = ((size_t)4 * 64 * KNN_RS + (size_t)kl * 64 * 2 + 64) * sizeof(float);
#define DGNN_KNN_RR_CASE(FCV)                                                                            \
  if (dim <= FCV) {                                                                                      \
    if (int rc = knn_allow_lds((const void *)k_knn_rerank<FCV>, lds)) return rc;                         \
    hipLaunchKernelGGL((k_knn_rerank<FCV>), grid, dim3(256), lds, st, points, pp, dim, k, kl, nq, x, nyk, cand, \
                       key_last, bad, src_ind, rflags, flags, qflags, env_int("DGNN_KNN_REDO_MAX", KNN_REDO_MAX), \
                       env_int("DGNN_KNN_XCD", 1));                                                      \
    DGNN_LAUNCH_CHECK();                                                                                 \
    return DGNN_OK;                                                                                      \
  }
  DGNN_KNN_RR_CASE(32)
  DGNN_KNN_RR_CASE(64)
  DGNN_KNN_RR_CASE(128)
#undef DGNN_KNN_RR_CASE
  return DGNN_ERR_UNSUPPORTED;
}

static int launch_knn_redo(int batch, int points, int dim, int k, const float *x, const int *rflags, const int *qflags,
                           const float *dk_bound, int *flags, int *src_ind, hipStream_t st) {
  const dim3 grid((unsigned)((points + 63) / 64), (unsigned)batch);
  if (dim <= 32)
    hipLaunchKernelGGL((k_knn_redo<32>), grid, dim3(256), 0, st, points, dim, k, x, rflags, qflags, dk_bound, flags, src_ind);
  else if (dim <= 64)
    hipLaunchKernelGGL((k_knn_redo<64>), grid, dim3(256), 0, st, points, dim, k, x, rflags, qflags, dk_bound, flags, src_ind);
  else
    hipLaunchKernelGGL((k_knn_redo<128>), grid, dim3(256), 0, st, points, dim, k, x, rflags, qflags, dk_bound, flags, src_ind);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

template <int KMAX>
static int knn_filtered(int batch, int points, int dim, int k, const float *x, int *src_ind, unsigned char *ws,
                        hipStream_t st) {
  const KnnWs w = knn_ws_layout(batch, points, dim, k);
  unsigned short *xh = (unsigned short *)(ws + w.off_xh), *xl = (unsigned short *)(ws + w.off_xl);
  float *nyk = (float *)(ws + w.off_nyk), *key_last = (float *)(ws + w.off_key);
  int *cand = (int *)(ws + w.off_cand), *bad = (int *)(ws + w.off_bad), *flags = (int *)(ws + w.off_flags);
  const long long nq = (long long)batch * points;
  if (hipError_t e = hipMemsetAsync(bad, 0, (size_t)batch * 4, st); e != hipSuccess) return (int)e;
  {
    float *mean = env_int("DGNN_KNN_CENTER", 1) != 0 ? (float *)(ws + w.off_mean) : nullptr;
    if (mean != nullptr) {
      hipLaunchKernelGGL(k_knn_center, dim3((unsigned)batch), dim3(256), 0, st, points, dim, w.dimp, x, mean);
      DGNN_LAUNCH_CHECK();
    }
    const long long threads = (long long)w.pp * (w.dimp / 8);
    hipLaunchKernelGGL(k_knn_split, dim3((unsigned)((threads + 255) / 256), (unsigned)batch), dim3(256), 0, st,
                       points, w.pp, dim, w.dimp, x, (const float *)mean, xh, xl, nyk, bad);
    DGNN_LAUNCH_CHECK();
  }
  {
    const dim3 grid((unsigned)((points + 127) / 128), (unsigned)batch);
    const size_t per_wave = (size_t)((w.kl * 32 * 2 > KNN_MQ * 64 * 2) ? w.kl * 32 * 2 : KNN_MQ * 64 * 2) * 4;
    int rc;
    const bool packed = w.pp <= 2048 && w.kl <= 48 && env_int("DGNN_KNN_PACKED", 1) != 0;
    if (packed && w.kl == 24) rc = launch_knn_mfma_packed<24>(w.dimp / 16, grid, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    else if (packed && w.kl == 32) rc = launch_knn_mfma_packed<32>(w.dimp / 16, grid, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    else if (packed) rc = launch_knn_mfma_packed<48>(w.dimp / 16, grid, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    else if (w.kl == 24) rc = launch_knn_mfma<24>(w.dimp / 16, grid, 4 * per_wave, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    else if (w.kl == 32) rc = launch_knn_mfma<32>(w.dimp / 16, grid, 4 * per_wave, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    else if (w.kl == 48) rc = launch_knn_mfma<48>(w.dimp / 16, grid, 4 * per_wave, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    else rc = launch_knn_mfma<64>(w.dimp / 16, grid, 4 * per_wave, st, points, w.pp, nq, xh, xl, nyk, cand, key_last);
    if (rc) return rc;
    DGNN_LAUNCH_CHECK();
  }
  int *qflags = (int *)(ws + w.off_qflags), *rflags = (int *)(ws + w.off_rflags);
  if (int rc = launch_knn_rerank(batch, points, w.pp, dim, k, w.kl, x, nyk, cand, key_last, bad, src_ind, rflags, flags,
                                 qflags, st))
    return rc;
  // second stage: blocks with a few uncertified queries, query by query; then blocks with many
  if (int rc = launch_knn_redo(batch, points, dim, k, x, rflags, qflags, key_last, flags, src_ind, st)) return rc;
  return launch_knn<KMAX>(batch, points, dim, k, x, src_ind, st, flags);            // flagged query blocks only
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" size_t dgnn_knn_graph_workspace_bytes(int batch, int points, int dim, int k) {
  if (batch <= 0 || points <= 0 || dim <= 0 || k <= 0 || k > points) return 0;
  if (!knn_filter_applies(batch, points, dim, k)) return 0;
  return knn_ws_layout(batch, points, dim, k).total;
}

extern "C" int dgnn_knn_graph(int batch, int points, int dim, int k, const float *x, int *src_ind,
                              void *stream) {
  return dgnn_knn_graph_ws(batch, points, dim, k, x, src_ind, nullptr, 0, stream);
}

extern "C" int dgnn_knn_graph_ws(int batch, int points, int dim, int k, const float *x, int *src_ind,
                                 void *workspace, size_t workspace_bytes, void *stream) {
  if (batch < 0 || points < 0 || dim <= 0 || k <= 0) return DGNN_ERR_BAD_ARG;
  if (batch == 0 || points == 0) return DGNN_OK;
  if (k > points) return DGNN_ERR_BAD_ARG;          // a cloud cannot supply k distinct points
  if (k > 64 || batch > 65535) return DGNN_ERR_UNSUPPORTED;
  if ((long long)batch * points > (1LL << 31) - 1) return DGNN_ERR_UNSUPPORTED;
  if (!x || !src_ind) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (workspace != nullptr && knn_filter_applies(batch, points, dim, k)) {
    if (workspace_bytes < knn_ws_layout(batch, points, dim, k).total || ((size_t)workspace & 255) != 0)
      return DGNN_ERR_BAD_ARG;
    unsigned char *ws = (unsigned char *)workspace;
    if (k <= 20) return knn_filtered<20>(batch, points, dim, k, x, src_ind, ws, st);
    if (k <= 24) return knn_filtered<24>(batch, points, dim, k, x, src_ind, ws, st);
    if (k <= 32) return knn_filtered<32>(batch, points, dim, k, x, src_ind, ws, st);
    if (k <= 40) return knn_filtered<40>(batch, points, dim, k, x, src_ind, ws, st);
    if (k <= 48) return knn_filtered<48>(batch, points, dim, k, x, src_ind, ws, st);
    return knn_filtered<64>(batch, points, dim, k, x, src_ind, ws, st);
  }
  if (k <= 8) return launch_knn<8>(batch, points, dim, k, x, src_ind, st);
  if (k <= 16) return launch_knn<16>(batch, points, dim, k, x, src_ind, st);
  if (k <= 20) return launch_knn<20>(batch, points, dim, k, x, src_ind, st);   // the reference's default k
  if (k <= 24) return launch_knn<24>(batch, points, dim, k, x, src_ind, st);
  if (k <= 32) return launch_knn<32>(batch, points, dim, k, x, src_ind, st);
  if (k <= 40) return launch_knn<40>(batch, points, dim, k, x, src_ind, st);   // its other published k
  if (k <= 48) return launch_knn<48>(batch, points, dim, k, x, src_ind, st);
  return launch_knn<64>(batch, points, dim, k, x, src_ind, st);
}

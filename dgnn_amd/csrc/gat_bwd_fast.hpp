// Neighbour-grouped fused GAT backward for gfx950: ONE gather pass over the CSC.
//
// The reference backward makes three passes with two per-edge 1 KiB gathers
// (mhspmm_backward_kernel :712-763 gathers grad rows over the CSC; mhsddmm :814-890 gathers
// in_feat rows over the CSR with a binary search per edge and writes grad_edge_csr [nnz,h];
// fused_backward_kernel :892-968 re-gathers attn_col twice and scatters E*H float atomics).
// Here, for a CSC column c (a source node) every quantity of the edge (r_i -> c) is available
// while grad[r_i] is in registers:
//   g_i = <grad[r_i,h,:], in_feat[c,h,:]>           in_feat[c] is the column's OWN row
//   S_r = sum_j a_j g_j = <grad[r,h,:], out[r,h,:]>  one dense dot per row (k_bwd_rowstat)
//   d_i = a_i (g_i - S_{r_i}) * (e_i<0 ? slope : 1)
//   grad_feat[c]     = sum_i w_i grad[r_i]           (w = a * mask/(1-p))
//   grad_attn_col[c] = sum_i d_i                     plain sum, no atomics
//   dE[permute[i]]   = d_i                           16-byte scatter, summed per row afterwards
// so the second 1 KiB gather, the [nnz,h] grad_edge_csr round trip, the binary searches and
// all float atomics disappear.
//
// Work decomposition: fixed neighbour groups of L CSC entries per wave64 (see gather_fast.hpp).
// The column's own in_feat row travels through the SAME software-pipelined load stream as the
// gathered grad rows (a "U item" in front of the column's edges), so a switch of column never
// stalls the pipeline.
#pragma once
#include <type_traits>
#include "common.hpp"
#include "gather_fast.hpp"

namespace dgnn {

#define DGNN_BWD_UFLAG 0x80000000u   // stream item = the column's own in_feat row
#define DGNN_BWD_END 0x40000000u     // stream item = last edge of its column inside the group
#define DGNN_BWD_IDMASK 0x3fffffffu

struct BwdParams {
  int n, m, nnz, H, F, HF, L, T;
  int steady;          // narrow rows: unguarded steady-state gather stream (STEADY instantiation)
  int fv_const;        // 0: the packed stream keeps run-time lane widths (DGNN_BWD_PACK_CONST=0, A/B runs)
  int pack_s;          // > 0: lanes per row of the PACKED stream (H * fv <= 32 lanes: 64 / pack_s edges of a
                       // column per load step), see the stream phase
  int vec;             // floats per lane load: 4 (F % 4 == 0, 16-byte aligned rows) or 1
  int fv;              // lanes per head: power of two >= ceil(F/vec), <= 64
  int hpp;             // heads per lane pass = 64 / fv; ceil(H / hpp) passes walk the stream
  int slot_stride;     // HF + H rounded up to 4
  float attn_drop, slope;
  const int *col_ptr, *row_ind, *permute;
  const float *attn_row, *attn_col, *edge_max, *edge_sum, *edge_mask;
  const float *in_feat, *grad, *out_feat;
  DropRng rng;         // dropout mask regenerated from (seed, offset) instead of edge_mask
  // optional pair from the forward (GatherParams::out_neg / a_neg): with it k_bwd_rowstat writes
  //   grad_attn_row[r,h] = (1 - slope) (S[r,h] * a_neg[r,h] - <grad[r,h,:], out_neg[r,h,:]>)
  // itself, and the column pass neither scatters d_i to dE nor is followed by the per-row sums
  const float *out_neg, *a_neg;
  float *grad_attn_row;
  const float *S_in;   // optional precomputed row dots [m,H]
  float4 *rowstat;     // [m][H] {attn_row, edge_max, edge_sum, S}
  float *dE;           // [nnz][H] in CSR order
  float *grad_feat, *grad_attn_col;
  int *task_col0;
  float *slots;        // [2T][slot_stride]
  // GMM mode (BWD_GMM): the edge weight is the Gaussian mixture weight of the edge's pseudo
  // coordinates (CSR order, fetched through permute); dE receives the raw dots = grad_gauss
  const float *pseudo, *mu, *inv_sigma;
  int dim;
  int grad_sum;        // BWD_GMM: grad is [m,F], the gradient of the output SUMMED over the kernels
                       // (every kernel's lanes read the same row slice): gmmconv_layer.py:58 folded in
  int hreg;            // BWD_GMM with grad_sum, F <= 64: an entry takes pack_s = max(4, F/4) lanes and the
                       // kernels' slices and accumulators sit in registers (see run_hreg)
};

enum { BWD_GAT = 0, BWD_GMM = 1 };

static inline int bwd_group_size(int H, int nnz, bool narrow = false, int slots = 0) {
  // narrow rows: see narrow_rows() in gather_fast.hpp (4 x 16 backward 9.4 -> 7.2 ms with 128)
  int dflt = narrow ? (H <= 4 ? 128 : 64) : (H <= 4 ? 256 : 128);
  // round 3 sweep (profiles/r3s_group_size_sweep.txt): one head of 12 .. 16 float4 slots (1 x 48, 1 x 64)
  // gains 4-5 % with 256; everything else keeps its size (two and four heads lose 10-35 % with 256)
  if (narrow && H == 1 && slots >= 12 && slots <= 16 && env_int("DGNN_FWD_L_RULE", 1) != 0) dflt = 256;
  const int v = env_int("DGNN_BWD_L", 0);      // tuning override: 64, 128 or 256
  if (!(v == 64 || v == 128 || v == 256)) return shrink_group_for_small_graphs(dflt, nnz, 64);
  return (H > 4 && v > 128) ? 128 : v;
}
static inline size_t bwd_lds_bytes(int H, int L, bool hreg = false) {
  // per wave: cell float2 [L][H] + kof[L] + colk[L] + dq[L] + stream[2L] (hreg: stream[(H + 1) L])
  return (size_t)DGNN_WAVES_PER_BLOCK * ((size_t)L * H * 2 + (size_t)L * 5 + (hreg ? (size_t)L * (H - 1) : 0)) * sizeof(float);
}

// Lane -> (head, feature slice) map shared by the backward kernels: in the pass that starts at
// head hb, head hb+g owns the aligned group of fv lanes [g*fv, (g+1)*fv); lane q of the group
// holds features [q*VEC, q*VEC+VEC) of the head, lanes past F idle.  Any (H <= 8, F) fits: a
// 4 x 64 row is one pass, 1 x 47 one pass of scalar lanes, 4 x 41 four passes of one head each;
// a pass gathers only its own heads' slice of every row, so no byte is read twice, and the
// per-head reductions stay inside aligned power-of-two lane groups (DPP).
struct BwdLane {
  bool active, leader;
  int head;
  int dup;             // float4 lanes, F % 4 != 0: leading components of the head's last slot that repeat
                       // its predecessor (common.hpp slot4_dup): dot products skip them
  size_t off;          // float offset of the lane's slice inside a feature row
};
template <int VEC>
__device__ __forceinline__ BwdLane bwd_lane(const BwdParams &P, int lane, int hb) {
  BwdLane m;
  const int head = hb + lane / P.fv, q = lane % P.fv;
  m.active = head < P.H && q * VEC < P.F;
  m.head = m.active ? head : 0;
  const int qo = (VEC == 4) ? slot4_off(q, P.F) : q;
  m.off = m.active ? (size_t)head * P.F + (size_t)qo : 0;
  m.dup = (VEC == 4 && m.active) ? slot4_dup(q, P.F) : 0;
  m.leader = m.active && q == 0;
  return m;
}
// the operand of a dot product: the repeated components of a head's last slot count once
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type dot_operand(const typename VecT<VEC>::type &v, int dup);
template <>
__device__ __forceinline__ float4 dot_operand<4>(const float4 &v, int dup) { return slot4_mask(v, dup); }
template <>
__device__ __forceinline__ float dot_operand<1>(const float &v, int) { return v; }
template <int VEC>
__device__ __forceinline__ float vec_dot(const typename VecT<VEC>::type &a, const typename VecT<VEC>::type &b);
template <>
__device__ __forceinline__ float vec_dot<4>(const float4 &a, const float4 &b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}
template <>
__device__ __forceinline__ float vec_dot<1>(const float &a, const float &b) { return a * b; }
template <int VEC>
__device__ __forceinline__ void vec_add(typename VecT<VEC>::type &a, const typename VecT<VEC>::type &b);
template <>
__device__ __forceinline__ void vec_add<4>(float4 &a, const float4 &b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
template <>
__device__ __forceinline__ void vec_add<1>(float &a, const float &b) { a += b; }

// Per destination row: S[r,h] = <grad[r,h,:], out[r,h,:]> and the packed statistics the CSC pass
// gathers with one 16-byte load per (edge, head).  One wave per row.
template <int VEC>
__global__ __launch_bounds__(DGNN_BLOCK) void k_bwd_rowstat(BwdParams P) {
  using vec_t = typename VecT<VEC>::type;
  const int r = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (r >= P.m) return;
  for (int hb = 0; hb < P.H; hb += P.hpp) {
    const BwdLane ln = bwd_lane<VEC>(P, lane_id(), hb);
    float s = 0.f, gn = 0.f;
    vec_t g = VecT<VEC>::zero();
    if (ln.active && (P.S_in == nullptr || P.out_neg != nullptr))
      g = dot_operand<VEC>(VecT<VEC>::load(P.grad + (size_t)r * P.HF + ln.off), ln.dup);
    if (P.S_in == nullptr) {
      if (ln.active) {
        const vec_t o = VecT<VEC>::load(P.out_feat + (size_t)r * P.HF + ln.off);
        s = vec_dot<VEC>(g, o);
      }
      s = group_reduce(s, P.fv, OpSum());
    } else if (ln.active) {
      s = P.S_in[(size_t)r * P.H + ln.head];
    }
    if (P.out_neg != nullptr) {
      if (ln.active) {
        const vec_t o = VecT<VEC>::load(P.out_neg + (size_t)r * P.HF + ln.off);
        gn = vec_dot<VEC>(g, o);
      }
      gn = group_reduce(gn, P.fv, OpSum());
    }
    if (ln.leader) {
      const size_t i = (size_t)r * P.H + ln.head;
      P.rowstat[i] = make_float4(P.attn_row[i], P.edge_max[i], P.edge_sum[i], s);
      if (P.out_neg != nullptr) P.grad_attn_row[i] = (1.f - P.slope) * fmaf(s, P.a_neg[i], -gn);
    }
  }
}

// The same for rows of at most 32 float4 slots: S = lanes per row (a power of two >= H * fv),
// 64 / S rows per load step and four steps per wave with all their loads issued before the first
// reduction.  One wave per row left most lanes idle and the pass bound by waves launched
// (profiles/r2d_*gatneg_1x64_products: 0.96 ms for 2.2 GB).
__global__ __launch_bounds__(DGNN_BLOCK) void k_bwd_rowstat_packed(BwdParams P, int S) {
  constexpr int STEPS = 4;
  const int G = DGNN_WAVE / S;
  const int lane = lane_id();
  const int sl = lane % S, gi = lane / S;
  const long long r0 = (long long)(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block()) * G * STEPS;
  if (r0 >= P.m) return;
  const BwdLane ln = bwd_lane<4>(P, sl, 0);
  const bool need_g = P.S_in == nullptr || P.out_neg != nullptr;
  float4 g[STEPS], o[STEPS], on[STEPS];
  bool ok[STEPS];
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const long long r = r0 + k * G + gi;
    ok[k] = ln.active && r < P.m;
    g[k] = o[k] = on[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok[k]) {
      const size_t at = (size_t)r * P.HF + ln.off;
      if (need_g) g[k] = slot4_mask(ld4u(P.grad + at), ln.dup);
      if (P.S_in == nullptr) o[k] = ld4u(P.out_feat + at);
      if (P.out_neg != nullptr) on[k] = ld4u(P.out_neg + at);
    }
  }
#pragma unroll
  for (int k = 0; k < STEPS; ++k) {
    const long long r = r0 + k * G + gi;
    float s, gn = 0.f;
    if (P.S_in == nullptr) s = group_reduce(vec_dot<4>(g[k], o[k]), P.fv, OpSum());
    else s = ok[k] ? P.S_in[(size_t)r * P.H + ln.head] : 0.f;
    if (P.out_neg != nullptr) gn = group_reduce(vec_dot<4>(g[k], on[k]), P.fv, OpSum());
    if (ok[k] && ln.leader) {
      const size_t i = (size_t)r * P.H + ln.head;
      P.rowstat[i] = make_float4(P.attn_row[i], P.edge_max[i], P.edge_sum[i], s);
      if (P.out_neg != nullptr) P.grad_attn_row[i] = (1.f - P.slope) * fmaf(s, P.a_neg[i], -gn);
    }
  }
}

// Column bookkeeping: empty columns are finished here, every group learns its first column.
__global__ void k_bwd_prep_cols(BwdParams P) {
  for (long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x; c < P.n;
       c += (long long)gridDim.x * blockDim.x) {
    const int lb = P.col_ptr[c], hb = P.col_ptr[c + 1];
    if (lb == hb) {
      float *o = P.grad_feat + (size_t)c * P.HF;
      for (int i = 0; i < P.HF; ++i) o[i] = 0.f;
      if (P.grad_attn_col != nullptr)
        for (int h = 0; h < P.H; ++h) P.grad_attn_col[(size_t)c * P.H + h] = 0.f;
    } else {
      for (int t = (lb + P.L - 1) / P.L; t <= (hb - 1) / P.L; ++t) P.task_col0[t] = (int)c;
    }
  }
}

// HC = heads rounded up to 1, 2, 4 or 8 (compile time; P.H is the real count), EPL = edges per
// lane; the group holds L = 64*EPL CSC entries.
// The gathered row statistics stay in registers between staging and the coefficient pass, so
// LDS keeps only 8 B per (edge, head): {signed weight, B} with d written over B afterwards.
// BMODE = BWD_GAT: the fused GAT backward above.  BWD_GMM: the same single pass for the fused
// GMMConv backward (replaces mhsddmm + mhtranspose + gmmconv on the CSC,
// dgNN/operators/fused_gmmconv.py:21-24): grad_node_feat[c,k,:] = sum_i w_ik grad[r_i,k,:] with
// w_ik = Gaussian weight of edge i, and grad_gauss[permute[i],k] = <grad[r_i,k,:], node_feat[c,k,:]>.
// STEADY: instantiation for narrow rows (fewer than 64 float4 slots), bound by instructions per
// edge: adds an unguarded steady-state form of the gather stream (see gather_fast.hpp).
// PK (STEADY only) = which gather stream the instantiation holds, chosen by the host
// (bwd_stream_kind in gather_bwd.hip) so that a kernel carries ONE stream instead of all of them:
//   0  the one-entry-per-step streams (steady / guarded)
//   1  the packed stream with run-time lane widths (and the GMM form with the kernels in registers)
//   2, 4, 8, 16  the packed stream with PK lanes per head as a compile-time constant
// Until round 3 every narrow kernel held all of these side by side and the register allocation was
// the union of their lane constants: 102 VGPRs = four waves per SIMD at one head; one stream per
// kernel takes 65 = seven waves, and these kernels are bound by the latencies a wave cannot hide.
#ifdef DGNN_BWD_WPE      // A/B builds: cap the waves per SIMD of the column gather (tools/ab_shapes.sh)
#define DGNN_BWD_OCC __attribute__((amdgpu_waves_per_eu(1, DGNN_BWD_WPE)))
#else
#define DGNN_BWD_OCC
#endif
template <int HC, int EPL, int VEC, int BMODE, bool STEADY, int PK = 0>
__global__ __launch_bounds__(DGNN_BLOCK) DGNN_BWD_OCC void k_bwd_col_gather(BwdParams P) {
  static_assert(PK == 0 || (STEADY && VEC == 4), "the packed streams exist in the narrow float4 instantiations only");
  using V = VecT<VEC>;
  using vec_t = typename V::type;
  extern __shared__ __align__(16) float lds_raw[];
  constexpr int L = DGNN_WAVE * EPL;
  const int H = P.H;
  const int wib = wave_in_block();
  const int lane = lane_id();
  const int t = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wib);
  if (t >= P.T) return;
  DGNN_PHASE_DECL();
  if (BMODE == BWD_GAT) drop_resolve(P.rng);
  // (hreg: the stream holds HC U steps per column instead of one: (HC + 1) L words, bwd_lds_bytes)
  const size_t per_wave_floats = (size_t)L * HC * 2 + (size_t)L * 5 + ((BMODE == BWD_GMM && P.hreg) ? (size_t)L * (HC - 1) : 0);
  float2 *cell = (float2 *)(lds_raw + (size_t)wib * per_wave_floats);  // [L][HC] {±w, B -> d}
  unsigned *kof = (unsigned *)(cell + (size_t)L * HC);                  // [L] column ordinal | END
  int *colk = (int *)(kof + L);                                         // [L] column id of ordinal k
  int *dq = colk + L;                                                   // [L] flush destination of ordinal k
  unsigned *stream = (unsigned *)(dq + L);                              // [2L] load stream

  const int e0 = t * L;
  const int e1 = min(e0 + L, P.nnz);
  const int cnt = e1 - e0;

  // ---------------------------------------------------------------- phase A: stage row data (registers)
  // Wide rows fetch the packed row records here, ahead of the column directory whose latencies they
  // overlap, and hold them in registers until the coefficient pass.  Narrow single-head rows fetch
  // them in the coefficient pass itself (57 VGPRs = eight waves per SIMD instead of 65 = seven).
  // With more heads the exposed latency costs more than the waves gained (measured, products shape,
  // backward call without / with the early fetch: 8 x 8 5.36 / 5.12 ms, 2 x 64 8.04 / 7.14, 2 x 32
  // 4.33 / 4.03), so those keep it.  DGNN_BWD_PREFETCH_NARROW=1 / 0 at build time forces it on / off
  // for every narrow kernel (A/B runs).
#ifdef DGNN_BWD_PREFETCH_NARROW
  constexpr bool PREFETCH = !STEADY || DGNN_BWD_PREFETCH_NARROW != 0;
#else
  constexpr bool PREFETCH = !STEADY || HC >= 2;
#endif
  int rid[EPL];
  float4 st[EPL][HC];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int p = j * DGNN_WAVE + lane;
    rid[j] = (p < cnt) ? P.row_ind[e0 + p] : 0;
  }
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int p = j * DGNN_WAVE + lane;
#pragma unroll
    for (int h = 0; h < HC; ++h)
      st[j][h] = (PREFETCH && BMODE == BWD_GAT && p < cnt && h < H) ? P.rowstat[(size_t)rid[j] * H + h]
                                                                   : make_float4(0.f, 0.f, 1.f, 0.f);
  }

  DGNN_PHASE_MARK(1, 0);
  // ---------------------------------------------------------------- columns of the group
  int K = 0;   // non-empty columns seen so far
  constexpr bool packed = PK != 0;       // the host launches PK != 0 exactly when P.pack_s > 0
  // packed stream: the load steps of the group in `stream`: per column one U step (the column's
  // own row: column id | UFLAG) followed by one step per (up to) G = 64 / pack_s consecutive
  // entries of that column: first entry (10 bits) | entries - 1 (4 bits) | DGNN_PACK_END on the
  // column's last step.  Steps never span columns.
  int nsteps = 0;
  {
    int cbase = uniform_i32(P.task_col0[t]);
    while (true) {
      const int c = cbase + lane;
      int lo = 0x7fffffff, hi = 0x7fffffff;
      if (c < P.n) {
        lo = P.col_ptr[c];
        hi = P.col_ptr[c + 1];
      }
      const int a = max(lo, e0) - e0;
      const int b = (hi < e1 ? hi : e1) - e0;
      const bool nonempty = (c < P.n) && (lo < e1) && (b > a);
      const unsigned long long bal = __ballot(nonempty);
      const int k = K + __popcll(bal & ((1ull << lane) - 1ull));
      if (nonempty) {
        const bool whole = (lo >= e0) && (hi <= e1);
        const int slot = (lo < e0) ? 2 * t : 2 * t + 1;
        colk[k] = c;
        dq[k] = whole ? c : -(slot + 1);
        if (!packed) stream[a + k] = (unsigned)c | DGNN_BWD_UFLAG;
        for (int p = a; p < b; ++p) kof[p] = (unsigned)k | (p == b - 1 ? DGNN_BWD_END : 0u);
      }
      if constexpr (STEADY) {
        if (packed) {
          const int G = DGNN_WAVE / P.pack_s;
          const int NU = (BMODE == BWD_GMM && P.hreg) ? H : 1;      // U steps per column
          const int mine = nonempty ? NU + (b - a + G - 1) / G : 0;
          int incl = mine;                                  // inclusive prefix sum over the lanes
#pragma unroll
          for (int off = 1; off < DGNN_WAVE; off <<= 1) {
            const int up = __shfl_up(incl, off, DGNN_WAVE);
            if (lane >= off) incl += up;
          }
          if (nonempty) {
            int pos = nsteps + incl - mine;
            for (int q = 0; q < NU; ++q) stream[pos++] = (unsigned)c | DGNN_BWD_UFLAG;
            for (int p0 = a; p0 < b; p0 += G)
              stream[pos++] = (unsigned)p0 | ((unsigned)(min(G, b - p0) - 1) << 10) | (p0 + G >= b ? DGNN_PACK_END : 0u);
          }
          nsteps += bcast_i32(incl, DGNN_WAVE - 1);
        }
      }
      K += __popcll(bal);
      const bool reached_end = (c >= P.n - 1) || (hi >= e1);
      if (__ballot(reached_end) != 0ull) break;
      cbase += DGNN_WAVE;
    }
  }
  wave_lds_sync();
  DGNN_PHASE_MARK(1, 1);

  // ---------------------------------------------------------------- per-edge coefficients
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int p = j * DGNN_WAVE + lane;
    if (p < cnt) {
      const unsigned kk = kof[p];
      const int k = (int)(kk & DGNN_BWD_IDMASK);
      const int c = colk[k];
      if (packed) kof[p] = (unsigned)rid[j];          // the stream gathers grad[kof[p]]; the ordinal is dead now
      else stream[p + k + 1] = (unsigned)rid[j] | (kk & DGNN_BWD_END);
      if (BMODE == BWD_GMM) {
        const size_t pos = (size_t)P.permute[e0 + p];
#pragma unroll
        for (int h = 0; h < HC; ++h) {
          if (h >= H) break;
          const float w = gmm_weight(P.dim, P.pseudo + pos * P.dim, P.mu + (size_t)h * P.dim,
                                     P.inv_sigma + (size_t)h * P.dim);
          cell[(size_t)p * HC + h] = make_float2(w, 0.f);
        }
        continue;
      }
      const bool drop = (P.edge_mask != nullptr) || P.rng.on;
      const int pos = drop ? P.permute[e0 + p] : 0;
      float ac[HC];
      if (HC == 4 && H == 4) {
        const float4 a4 = *(const float4 *)(P.attn_col + (size_t)c * 4);
        ac[0] = a4.x; ac[1 % HC] = a4.y; ac[2 % HC] = a4.z; ac[3 % HC] = a4.w;
      } else {
#pragma unroll
        for (int h = 0; h < HC; ++h) ac[h] = (h < H) ? P.attn_col[(size_t)c * H + h] : 0.f;
      }
#pragma unroll
      for (int h = 0; h < HC; ++h) {
        if (h >= H) break;
        const float4 s4 = PREFETCH ? st[j][h] : P.rowstat[(size_t)rid[j] * H + h];   // {attn_row, edge_max, edge_sum, S}
        const float e = leaky_relu(s4.x + ac[h], P.slope);
        const float a = div_pos(exp_nonpos(e - s4.y), s4.z);    // :747
        float w = a;
        if (drop) {
          const float mk = P.rng.on ? drop_uniform(P.rng, (unsigned long long)pos * H + h)
                                    : P.edge_mask[(size_t)pos * H + h];
          if (!(mk > P.attn_drop)) w = 0.f;                                  // :738
          w = w / (1 - P.attn_drop);                                         // :749 (float divide)
        }
        const bool neg = e < 0.f;                                            // :953
        const float B = a * s4.w * (neg ? P.slope : 1.f);
        cell[(size_t)p * HC + h] = make_float2(neg ? -w : w, B);
      }
    }
  }
  wave_lds_sync();
  DGNN_PHASE_MARK(1, 2);

  // ---------------------------------------------------------------- pipelined gather stream
  const int items = cnt + K;
  for (int hb = 0; hb < H; hb += P.hpp) {   // one pass per block of heads (a single pass when H*fv <= 64)
    const BwdLane ln = bwd_lane<VEC>(P, lane, hb);
    const bool active = ln.active;
    const int head = ln.head;
    const size_t lane_off = ln.off;
    const bool head_leader = ln.leader;
    // grad rows: [m,H,F] like the feature rows, or [m,F] shared by all kernels (BWD_GMM, grad_sum)
    const bool gsum = (BMODE == BWD_GMM) && P.grad_sum != 0;
    const size_t gstride = gsum ? (size_t)P.F : (size_t)P.HF;
    const size_t goff = gsum ? lane_off - (size_t)head * P.F : lane_off;
    vec_t u = V::zero();
    vec_t acc = V::zero();
    float dsum = 0.f;
    int k = -1;
    vec_t ring[DGNN_FAST_U];
    bool took_steady = false;
    if constexpr (STEADY && VEC == 4) {
      if (packed) {
        // ---- packed stream (H * fv <= 16 lanes per row): the lanes form G = 64 / S groups; an
        // edge step gathers the grad rows of G consecutive entries of ONE column (group g: entry
        // p0 + g), each group keeps its own partial sums, and the column's last step adds them
        // across groups.  The column's own row (U step) is loaded by every group alike.
        //
        // The lanes per head (fv) and per row (S = max(4, HC * fv): heads are padded to HC) are
        // compile-time constants of the runner: with run-time widths the per-entry dot reduction
        // was a chain of six wave-uniform branches and the column-end sum a loop of shuffles --
        // 11 scalar instructions per EDGE at 1 x 64 and K = 3 x 16, as many cycles as the vector work
        // (profiles/r2h_pmc_sq_gmmsum16_reddit.csv).  FVC = 0 keeps the run-time widths (shapes
        // whose row width is not max(4, HC * fv)).
        took_steady = true;
        // ---- BWD_GMM with the summed gradient: every kernel reads the SAME grad slice, so an entry
        // takes SC = max(4, F/4) lanes (not kernels x F/4), a step 64 / SC entries, and the lane keeps
        // the column's feature slice and the accumulator of EVERY kernel in registers: one grad load
        // per entry instead of one per kernel, and (K = 3 x 64) four entries per step instead of one.
        // The column's own rows arrive as H consecutive U steps (kernel 0, 1, ...).
        bool ran_hreg = false;
        if constexpr (BMODE == BWD_GMM && PK == 1) {
          if (P.hreg) {
            ran_hreg = true;
            auto run_hreg = [&](auto scc) {
              constexpr int SC = decltype(scc)::value;
              const int g = lane / SC, qq = lane % SC;
              const bool act = qq * 4 < P.F;
              const size_t qoff = act ? (size_t)qq * 4 : 0;
              float4 uh[HC], acch[HC];
              float4 ring[DGNN_PACK_U];
#pragma unroll
              for (int h = 0; h < HC; ++h) uh[h] = acch[h] = make_float4(0.f, 0.f, 0.f, 0.f);
              int iu = 0, cu = 0;       // kernel of the next U step: issue side, consume side (wave-uniform)
              auto issue = [&](int j, float4 &x) {
                const unsigned it = (unsigned)uniform_i32((int)stream[j]);
                if (it & DGNN_BWD_UFLAG) {
                  x = ld4u(P.in_feat + (size_t)(it & DGNN_BWD_IDMASK) * P.HF + (size_t)iu * P.F + qoff);
                  iu = (iu + 1 == H) ? 0 : iu + 1;
                } else {
                  const int p0 = (int)(it & 1023u), n = (int)((it >> 10) & 15u) + 1;
                  const unsigned r = kof[p0 + (g < n ? g : 0)];
                  x = ld4u(P.grad + (size_t)r * P.F + qoff);
                }
              };
              auto consume = [&](const float4 &x, int j) {
                const unsigned it = (unsigned)uniform_i32((int)stream[j]);
                if (it & DGNN_BWD_UFLAG) {
#pragma unroll
                  for (int h = 0; h < HC; ++h)
                    if (cu == h) uh[h] = x;
                  if (cu == 0) ++k;
                  cu = (cu + 1 == H) ? 0 : cu + 1;
                  return;
                }
                const int p0 = (int)(it & 1023u), n = (int)((it >> 10) & 15u) + 1;
                const bool live = g < n;
                float2 *cp = cell + (size_t)(p0 + (live ? g : 0)) * HC;
#pragma unroll
                for (int h = 0; h < HC; ++h) {
                  if (h >= H) break;
                  const float w = live ? fabsf(cp[h].x) : 0.f;
                  float dot = act ? vec_dot<4>(x, uh[h]) : 0.f;
                  dot += dpp_f32<DGNN_DPP_QUAD_XOR1>(dot);
                  dot += dpp_f32<DGNN_DPP_QUAD_XOR2>(dot);
                  if constexpr (SC >= 8) dot += dpp_f32<DGNN_DPP_ROW_HALF_MIRROR>(dot);
                  if constexpr (SC >= 16) dot += dpp_f32<DGNN_DPP_ROW_MIRROR>(dot);
                  V::fma(acch[h], w, x);
                  if (qq == 0 && live) cp[h].y = dot;
                }
                if (it & DGNN_PACK_END) {
                  const int dst = uniform_i32(dq[k]);
                  float *base = dst >= 0 ? P.grad_feat + (size_t)dst * P.HF
                                         : P.slots + (size_t)(-(dst + 1)) * P.slot_stride;
#pragma unroll
                  for (int h = 0; h < HC; ++h) {
                    if (h >= H) break;
#pragma unroll
                    for (int off = SC; off < DGNN_WAVE; off <<= 1) {
                      acch[h].x += __shfl_xor(acch[h].x, off, DGNN_WAVE);
                      acch[h].y += __shfl_xor(acch[h].y, off, DGNN_WAVE);
                      acch[h].z += __shfl_xor(acch[h].z, off, DGNN_WAVE);
                      acch[h].w += __shfl_xor(acch[h].w, off, DGNN_WAVE);
                    }
                    if (g == 0 && act) st4u(base + (size_t)h * P.F + qoff, acch[h]);
                    acch[h] = make_float4(0.f, 0.f, 0.f, 0.f);
                  }
                }
              };
              int j = 0;
              if (nsteps >= 2 * DGNN_PACK_U) {
#pragma unroll
                for (int i = 0; i < DGNN_PACK_U; ++i) {
                  issue(i, ring[i]);
                  __builtin_amdgcn_sched_barrier(0);
                }
                for (; j + 2 * DGNN_PACK_U <= nsteps; j += DGNN_PACK_U) {
#pragma unroll
                  for (int i = 0; i < DGNN_PACK_U; ++i) {
                    consume(ring[i], j + i);
                    issue(j + i + DGNN_PACK_U, ring[i]);
                    __builtin_amdgcn_sched_barrier(0);
                  }
                }
#pragma unroll
                for (int i = 0; i < DGNN_PACK_U; ++i) {
                  const float4 x = ring[i];
                  if (j + i + DGNN_PACK_U < nsteps) issue(j + i + DGNN_PACK_U, ring[i]);
                  consume(x, j + i);
                }
                j += DGNN_PACK_U;
#pragma unroll
                for (int i = 0; i < DGNN_PACK_U; ++i)
                  if (j + i < nsteps) consume(ring[i], j + i);
              } else {
                for (; j < nsteps; ++j) {
                  float4 x;
                  issue(j, x);
                  consume(x, j);
                }
              }
            };
            if (P.pack_s == 16) run_hreg(std::integral_constant<int, 16>{});
            else if (P.pack_s == 8) run_hreg(std::integral_constant<int, 8>{});
            else run_hreg(std::integral_constant<int, 4>{});
          }
        }
        auto run_packed = [&](auto fvc) {
        constexpr int FVC = decltype(fvc)::value;
        constexpr int SC = FVC == 0 ? 0 : (HC * FVC > 4 ? HC * FVC : 4);
        const int fv = FVC == 0 ? P.fv : FVC;
        const int S = FVC == 0 ? P.pack_s : SC;
        const int g = lane / S, sl = lane % S;
        const int hd = sl / fv, qq = sl % fv;
        const bool act = hd < H && qq * 4 < P.F;
        const int phead = act ? hd : 0;
        const int pq = act ? slot4_off(qq, P.F) : 0;           // the head's last slot may overlap its predecessor
        const int pdup = act ? slot4_dup(qq, P.F) : 0;
        const size_t poff = act ? (size_t)phead * P.F + (size_t)pq : 0;
        const bool pleader = act && qq == 0;
        const size_t pgoff = (BMODE == BWD_GMM && P.grad_sum) ? (size_t)pq : poff;
        float4 ring[DGNN_PACK_U];
        auto issue = [&](int j, float4 &x) {
          const unsigned it = (unsigned)uniform_i32((int)stream[j]);
          const bool isu = (it & DGNN_BWD_UFLAG) != 0u;
          const int p0 = isu ? 0 : (int)(it & 1023u), n = isu ? 0 : (int)((it >> 10) & 15u) + 1;
          const int pe = p0 + (g < n ? g : 0);
          const unsigned r = isu ? (it & DGNN_BWD_IDMASK) : kof[pe];
          x = ld4u(isu ? P.in_feat + (size_t)r * P.HF + poff : P.grad + (size_t)r * gstride + pgoff);
        };
        auto consume = [&](const float4 &x, int j) {
          const unsigned it = (unsigned)uniform_i32((int)stream[j]);
          if (it & DGNN_BWD_UFLAG) {
            u = (P.F & 3) ? slot4_mask(x, pdup) : x;      // (wave-uniform: aligned widths skip the mask)
            ++k;
            return;
          }
          const int p0 = (int)(it & 1023u), n = (int)((it >> 10) & 15u) + 1;
          const bool live = g < n;
          const int pe = p0 + (live ? g : 0);
          float2 *cp = cell + (size_t)pe * HC + phead;
          const float2 wb = *cp;
          const float w = live ? fabsf(wb.x) : 0.f;
          const float slp = (__float_as_uint(wb.x) & 0x80000000u) ? P.slope : 1.f;
          float dot = act ? vec_dot<4>(x, u) : 0.f;
          if constexpr (FVC == 0) {
            dot = group_reduce(dot, fv, OpSum());
          } else {               // at most 16 lanes per head here: the four DPP steps, no branch
            if constexpr (FVC >= 2) dot += dpp_f32<DGNN_DPP_QUAD_XOR1>(dot);
            if constexpr (FVC >= 4) dot += dpp_f32<DGNN_DPP_QUAD_XOR2>(dot);
            if constexpr (FVC >= 8) dot += dpp_f32<DGNN_DPP_ROW_HALF_MIRROR>(dot);
            if constexpr (FVC >= 16) dot += dpp_f32<DGNN_DPP_ROW_MIRROR>(dot);
          }
          const float d = live ? ((BMODE == BWD_GMM) ? dot : fmaf(slp * w, dot, -wb.y)) : 0.f;
          V::fma(acc, w, x);
          dsum += d;
          if (pleader && live) cp->y = d;
          if (it & DGNN_PACK_END) {
            if constexpr (FVC == 0) {
              for (int off = S; off < DGNN_WAVE; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off, DGNN_WAVE);
                acc.y += __shfl_xor(acc.y, off, DGNN_WAVE);
                acc.z += __shfl_xor(acc.z, off, DGNN_WAVE);
                acc.w += __shfl_xor(acc.w, off, DGNN_WAVE);
                dsum += __shfl_xor(dsum, off, DGNN_WAVE);
              }
            } else {
#pragma unroll
              for (int off = SC; off < DGNN_WAVE; off <<= 1) {
                acc.x += __shfl_xor(acc.x, off, DGNN_WAVE);
                acc.y += __shfl_xor(acc.y, off, DGNN_WAVE);
                acc.z += __shfl_xor(acc.z, off, DGNN_WAVE);
                acc.w += __shfl_xor(acc.w, off, DGNN_WAVE);
                dsum += __shfl_xor(dsum, off, DGNN_WAVE);
              }
            }
            const int dst = uniform_i32(dq[k]);
            if (g == 0) {
              if (dst >= 0) {
                if (act) st4u(P.grad_feat + (size_t)dst * P.HF + poff, acc);
                if (BMODE == BWD_GAT && pleader) P.grad_attn_col[(size_t)dst * H + phead] = dsum;
              } else {
                float *sp = P.slots + (size_t)(-(dst + 1)) * P.slot_stride;
                if (act) st4u(sp + poff, acc);
                if (BMODE == BWD_GAT && pleader) sp[P.HF + phead] = dsum;
              }
            }
            acc = V::zero();
            dsum = 0.f;
          }
        };
        int j = 0;
        if (nsteps >= 2 * DGNN_PACK_U) {
#pragma unroll
          for (int i = 0; i < DGNN_PACK_U; ++i) {
            issue(i, ring[i]);
            __builtin_amdgcn_sched_barrier(0);
          }
          for (; j + 2 * DGNN_PACK_U <= nsteps; j += DGNN_PACK_U) {
#pragma unroll
            for (int i = 0; i < DGNN_PACK_U; ++i) {
              consume(ring[i], j + i);
              issue(j + i + DGNN_PACK_U, ring[i]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int i = 0; i < DGNN_PACK_U; ++i) {
            const float4 x = ring[i];
            if (j + i + DGNN_PACK_U < nsteps) issue(j + i + DGNN_PACK_U, ring[i]);
            consume(x, j + i);
          }
          j += DGNN_PACK_U;
#pragma unroll
          for (int i = 0; i < DGNN_PACK_U; ++i)
            if (j + i < nsteps) consume(ring[i], j + i);
        } else {
          for (; j < nsteps; ++j) {
            float4 x;
            issue(j, x);
            consume(x, j);
          }
        }
        };   // run_packed
        if constexpr (PK >= 2) {
          run_packed(std::integral_constant<int, PK>{});
        } else {
          if (!ran_hreg) run_packed(std::integral_constant<int, 0>{});
        }
      }
    }
    if constexpr (STEADY && PK == 0) {
      if (!took_steady && P.steady && items >= 2 * DGNN_FAST_U) {
        took_steady = true;
        // one item of the stream, its row already in x
        auto consume = [&](const vec_t &x, unsigned v, int it) {
          if (v & DGNN_BWD_UFLAG) {
            u = dot_operand<VEC>(x, ln.dup);
            ++k;
          } else {
            const int p = it - (k + 1);
            float2 *cp = cell + (size_t)p * HC + head;
            const float2 wb = *cp;
            const float w = fabsf(wb.x);
            const float sl = (__float_as_uint(wb.x) & 0x80000000u) ? P.slope : 1.f;
            float dot = active ? vec_dot<VEC>(x, u) : 0.f;
            dot = group_reduce(dot, P.fv, OpSum());
            const float d = (BMODE == BWD_GMM) ? dot : fmaf(sl * w, dot, -wb.y);
            V::fma(acc, w, x);
            dsum += d;
            if (head_leader) cp->y = d;
            if (v & DGNN_BWD_END) {
              const int dst = uniform_i32(dq[k]);
              if (dst >= 0) {
                if (active) V::store(P.grad_feat + (size_t)dst * P.HF + lane_off, acc);
                if (BMODE == BWD_GAT && head_leader) P.grad_attn_col[(size_t)dst * H + head] = dsum;
              } else {
                float *sp = P.slots + (size_t)(-(dst + 1)) * P.slot_stride;
                if (active) V::store(sp + lane_off, acc);
                if (BMODE == BWD_GAT && head_leader) sp[P.HF + head] = dsum;
              }
              acc = V::zero();
              dsum = 0.f;
            }
          }
        };
        auto row_of = [&](unsigned v) -> const float * {
          return (v & DGNN_BWD_UFLAG) ? P.in_feat + (size_t)(v & DGNN_BWD_IDMASK) * P.HF + lane_off
                                      : P.grad + (size_t)(v & DGNN_BWD_IDMASK) * gstride + goff;
        };
        // unguarded fill and loop: loads issued unconditionally (idle lanes re-read slot 0 of the
        // row), a slot consumed before it is refilled
#pragma unroll
        for (int i = 0; i < DGNN_FAST_U; ++i) {
          ring[i] = V::load(row_of((unsigned)uniform_i32((int)stream[i])));
          __builtin_amdgcn_sched_barrier(0);
        }
        int j = 0;
        for (; j + 2 * DGNN_FAST_U <= items; j += DGNN_FAST_U) {
#pragma unroll
          for (int i = 0; i < DGNN_FAST_U; ++i) {
            const int it = j + i;
            const unsigned v = (unsigned)uniform_i32((int)stream[it]);
            const unsigned nv = (unsigned)uniform_i32((int)stream[it + DGNN_FAST_U]);
            consume(ring[i], v, it);
            ring[i] = V::load(row_of(nv));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        for (; j < items; j += DGNN_FAST_U) {
#pragma unroll
          for (int i = 0; i < DGNN_FAST_U; ++i) {
            const int it = j + i;
            if (it < items) {
              const vec_t x = ring[i];
              const unsigned v = (unsigned)uniform_i32((int)stream[it]);
              if (it + DGNN_FAST_U < items)
                ring[i] = V::load(row_of((unsigned)uniform_i32((int)stream[it + DGNN_FAST_U])));
              consume(x, v, it);
            }
          }
        }
      }
    }
    if constexpr (PK == 0) {
    if (!took_steady) {
  #pragma unroll
    for (int i = 0; i < DGNN_FAST_U; ++i) {
      ring[i] = V::zero();
      if (i < items) {
        const unsigned v = (unsigned)uniform_i32((int)stream[i]);
        const bool isu = (v & DGNN_BWD_UFLAG) != 0u;
        if (active) ring[i] = V::load((isu ? P.in_feat : P.grad) + (size_t)(v & DGNN_BWD_IDMASK) * (isu ? (size_t)P.HF : gstride) +
                                      (isu ? lane_off : goff));
      }
    }
    for (int j = 0; j < items; j += DGNN_FAST_U) {
  #pragma unroll
      for (int i = 0; i < DGNN_FAST_U; ++i) {
        const int it = j + i;
        if (it < items) {
          const vec_t x = ring[i];
          const unsigned v = (unsigned)uniform_i32((int)stream[it]);
          if (it + DGNN_FAST_U < items) {
            const unsigned nv = (unsigned)uniform_i32((int)stream[it + DGNN_FAST_U]);
            const bool isu = (nv & DGNN_BWD_UFLAG) != 0u;
            if (active) ring[i] = V::load((isu ? P.in_feat : P.grad) + (size_t)(nv & DGNN_BWD_IDMASK) * (isu ? (size_t)P.HF : gstride) +
                                          (isu ? lane_off : goff));
          }
          if (v & DGNN_BWD_UFLAG) {
            u = dot_operand<VEC>(x, ln.dup);
            ++k;
          } else {
            const int p = it - (k + 1);
            float2 *cp = cell + (size_t)p * HC + head;
            const float2 wb = *cp;
            const float w = fabsf(wb.x);
            const float sl = (__float_as_uint(wb.x) & 0x80000000u) ? P.slope : 1.f;
            float dot = active ? vec_dot<VEC>(x, u) : 0.f;
            dot = group_reduce(dot, P.fv, OpSum());
            const float d = (BMODE == BWD_GMM) ? dot : fmaf(sl * w, dot, -wb.y);
            V::fma(acc, w, x);
            dsum += d;
            if (head_leader) cp->y = d;     // B is dead after this read: keep d in its place
            if (v & DGNN_BWD_END) {
              const int dst = uniform_i32(dq[k]);
              if (dst >= 0) {
                if (active) V::store(P.grad_feat + (size_t)dst * P.HF + lane_off, acc);
                if (BMODE == BWD_GAT && head_leader) P.grad_attn_col[(size_t)dst * H + head] = dsum;
              } else {
                float *sp = P.slots + (size_t)(-(dst + 1)) * P.slot_stride;
                if (active) V::store(sp + lane_off, acc);
                if (BMODE == BWD_GAT && head_leader) sp[P.HF + head] = dsum;
              }
              acc = V::zero();
              dsum = 0.f;
            }
          }
        }
      }
    }
    }
    }
  }
  wave_lds_sync();
  DGNN_PHASE_MARK(1, 3);

  // ---------------------------------------------------------------- d_i to CSR order
  if (P.dE == nullptr) return;      // grad_attn_row came out of the row pass (out_neg / a_neg given)
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int p = j * DGNN_WAVE + lane;
    if (p < cnt) {
      const size_t pos = (size_t)P.permute[e0 + p];
      if (HC == 4 && H == 4) {
        *(float4 *)(P.dE + pos * 4) = make_float4(cell[(size_t)p * HC + 0].y, cell[(size_t)p * HC + 1 % HC].y,
                                                  cell[(size_t)p * HC + 2 % HC].y, cell[(size_t)p * HC + 3 % HC].y);
      } else {
#pragma unroll
        for (int h = 0; h < HC; ++h)
          if (h < H) P.dE[pos * H + h] = cell[(size_t)p * HC + h].y;
      }
    }
  }
}

// Merge the partial slots of columns cut by a group boundary (plain sums, fixed order).
#ifndef DGNN_FIX_C
#define DGNN_FIX_C 16
#endif
template <int VEC>
__global__ __launch_bounds__(DGNN_BLOCK) void k_bwd_fixup(BwdParams P, int phase) {
  using V = VecT<VEC>;
  using vec_t = typename V::type;
  const int t = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block()) + 1;
  if (t >= P.T) return;
  const int L = P.L, H = P.H;
  const int c = uniform_i32(P.task_col0[t]);
  if (phase == 0) {        // cheap necessary condition first (see k_fast_fixup)
    const int back = uniform_i32(P.task_col0[max(t - (DGNN_FIX_C - 1), 0)]);
    const int ahead = uniform_i32(P.task_col0[min(t + DGNN_FIX_C - 2, P.T - 1)]);
    if (back != c && ahead != c) return;
  }
  const int cs = uniform_i32(P.col_ptr[c]), ce = uniform_i32(P.col_ptr[c + 1]);
  const long long edge = (long long)t * L;
  if (cs >= edge) return;
  const int t0 = cs / L + 1;                 // the first boundary inside the column owns its merge
  const int t_last = (ce - 1) / L;
  const int npieces = t_last - t0 + 2;
  // hub columns (more than DGNN_FIX_C pieces) in two launches, like k_fast_fixup: phase 0 sums
  // groups of DGNN_FIX_C pieces into the first of each group, phase 1 sums those
  const bool two_level = npieces > DGNN_FIX_C;
  int first = 0, cnt = npieces, stride = 1;
  if (phase == 0) {
    if (!two_level) return;
    const int d = t - t0;
    int k;
    if (d == 0) k = 0;
    else if ((d + 1) % DGNN_FIX_C == 0) k = (d + 1) / DGNN_FIX_C;
    else return;
    first = k * DGNN_FIX_C;
    cnt = min(DGNN_FIX_C, npieces - first);
    if (cnt <= 1) return;
  } else {
    if (t != t0) return;
    if (two_level) {
      stride = DGNN_FIX_C;
      cnt = (npieces + DGNN_FIX_C - 1) / DGNN_FIX_C;
    }
  }
  auto slot_of = [&](int i) -> float * {
    const int pi = first + i * stride;
    const int s = (pi == 0) ? 2 * (t0 - 1) + 1 : 2 * (t0 + pi - 1);
    return P.slots + (size_t)s * P.slot_stride;
  };
  for (int hb = 0; hb < H; hb += P.hpp) {
    const BwdLane ln = bwd_lane<VEC>(P, lane_id(), hb);
    vec_t acc = V::zero();
    float dsum = 0.f;
    // loads of FB pieces in flight together, sums in piece order; the usual two pieces as a batch of two
    const bool want_d = P.grad_attn_col != nullptr;
    auto merge = [&](auto fbc) {
      constexpr int FB = decltype(fbc)::value;
      for (int i0 = 0; i0 < cnt; i0 += FB) {
        vec_t v[FB];
        float d[FB];
#pragma unroll
        for (int j = 0; j < FB; ++j) {
          const float *sp = slot_of(min(i0 + j, cnt - 1));
          v[j] = V::load(sp + ln.off);
          d[j] = want_d ? sp[P.HF + ln.head] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < FB; ++j)
          if (i0 + j < cnt) {
            vec_add<VEC>(acc, v[j]);
            dsum += d[j];
          }
      }
    };
    if (cnt <= 2) merge(std::integral_constant<int, 2>{});
    else merge(std::integral_constant<int, 8>{});
    if (phase == 0) {
      float *sp = slot_of(0);
      if (ln.active) {
        V::store(sp + ln.off, acc);
        if (ln.leader && want_d) sp[P.HF + ln.head] = dsum;
      }
    } else if (ln.active) {
      V::store(P.grad_feat + (size_t)c * P.HF + ln.off, acc);
      if (ln.leader && P.grad_attn_col != nullptr) P.grad_attn_col[(size_t)c * H + ln.head] = dsum;
    }
  }
}

}  // namespace dgnn

// Neighbour-grouped row-gather engine for gfx950 (wave64): the fast path of the fused GAT
// forward, GE-SpMM and fused GMMConv forward.
//
// Work decomposition (what replaces the reference's one-block-per-(row,head) launch,
// fused_gatconv_kernel.cu:344, and its experimental tile scheduler, :1153-1237):
//   * the nnz array is cut into fixed NEIGHBOUR GROUPS of L consecutive edges; one wave64
//     owns one group, whatever rows it spans.  Short rows are packed many-per-wave, a hub
//     row of 100k neighbours is spread over hundreds of waves: every wave gathers the same
//     number of feature rows, so there is no tail and no host-side scheduler.
//   * phase A  the wave stages its L column ids and their per-head score terms in LDS
//              (one coalesced index load + one 16-byte gather per edge, once).
//   * phase B  per row segment inside the group: LeakyReLU scores, max / sum reductions with
//              DPP row ops, softmax weights written back to LDS in place.  A row that lies
//              wholly inside the group is finished here exactly like the reference
//              (w = exp(e-max)/sum per edge); a row cut by a group boundary keeps
//              un-normalised exp(e-m_piece) and leaves (m, s, acc) in a partial slot.
//   * phase C  ONE flat, software-pipelined stream over the group's edges: U feature-row
//              gathers always in flight per wave (each a fully coalesced 16 B/lane load:
//              64 lanes x float4 = one 1 KiB [H=4,F=64] row), accumulate, and flush at row
//              ends (flag bit in the staged column id).  No atomics anywhere.
//   * a tiny fix-up kernel merges the partial slots of cut rows in a fixed order (online
//              softmax merge), so results are deterministic.
#pragma once
#include <type_traits>
#include <cstdlib>

#include "common.hpp"

namespace dgnn {

#ifndef DGNN_FAST_U
#define DGNN_FAST_U 3          // feature-row gathers in flight per wave (swept on MI355X: 1..16)
#endif
#ifndef DGNN_PACK_U
#define DGNN_PACK_U 3          // load steps in flight per wave in the PACKED streams (narrow rows, both gather kernels)
#endif
#ifndef DGNN_FWD_ROWCACHE
#define DGNN_FWD_ROWCACHE 1
#endif
#define DGNN_FAST_HMAX 8
#define DGNN_ROWEND 0x80000000u

enum { MODE_GAT = 0, MODE_VAL = 1, MODE_GMM = 2 };

struct GatherParams {
  int m, nnz, H, F, HF, L, T;        // rows, edges, heads, feat, H*F, group size, #groups
  int rowv;                          // vector slots per feature row: H * spf
  int vec4;                          // != 0: float4 lanes (F >= 4), else scalar lanes
  int spf;                           // vector slots per head: ceil(F / 4) for float4 lanes (any F >= 4: the last
                                     // slot of an F that is no multiple of four overlaps its predecessor,
                                     // common.hpp slot4_off), F for scalar lanes
  int slot_stride;                   // floats per partial slot (HF + 2H rounded up to 4)
  int dim;                           // GMM pseudo dimension
  int sum_heads;                     // != 0: out is [m,F] = the sum over the H heads / kernels (GMMConv layer:
                                     // gmmconv_layer.py:58 `.sum(1)`), folded into the store; rows of <= 64 float4 slots
  int steady;                        // narrow rows: unguarded steady-state gather loop (phase C)
  int batch_rows;                    // != 0: the softmax phase takes up to four short rows per pass
  int pack_s;                        // > 0: lanes per row of the PACKED gather (rows of <= 32 float4 slots:
                                     // 64 / pack_s edges of a row per load instruction), see phase C
  float attn_drop, slope;
  const int *row_ptr, *col_ind;
  const float *attn_row, *attn_col;  // GAT
  const float *edge_mask;            // GAT dropout mask or NULL
  DropRng rng;                       // or: the mask regenerated from (seed, offset)
  const float *edge_val;             // SpMM values [nnz,H]
  const float *pseudo, *mu, *inv_sigma;  // GMM
  const float *in_feat;
  float *edge_max, *edge_sum;        // GAT stats or NULL
  float *out;
  // GAT, optional pair (the NEG instantiation): what lets the backward compute grad_attn_row per
  // row, without the [nnz,H] edge-gradient round trip.  With sl_j = (e_j < 0 ? slope : 1):
  //   grad_attn_row = sum_j a_j (g_j - S) sl_j = (1 - slope) (S * A_neg - <grad, out_neg>)
  // because sum_j a_j g_j sl_j = S - (1-slope) sum_{e_j<0} a_j g_j and sum_j a_j sl_j = 1 - (1-slope) A_neg.
  float *out_neg;                    // [m,H,F]  sum over the edges with e_j < 0 of w_j * in_feat[c_j]
  float *a_neg;                      // [m,H]    sum over the edges with e_j < 0 of the softmax weight (no mask)
  float *out2;                       // optional second copy of `out` (NEG instantiations only): the copy the
                                     // reference-signature path keeps for the backward (dgnn_gat_forward_saved),
                                     // written by the same lanes as `out` -- one more store, no copy pass
  int slot_neg;                      // float offset of the out_neg partial inside a slot
  int *task_row0;                    // [T] row that contains edge t*L
  float *slots;                      // [2T][slot_stride]
};

static inline int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return (e != nullptr && e[0] != 0) ? atoi(e) : dflt;
}
// A wave walks its group with only a few loads in flight, so a group costs ~L memory latencies
// whatever the graph: small graphs get smaller groups until there are enough waves to fill
// the 256 CUs (a Cora-sized graph with L = 512 would run on 26 waves).
static inline int shrink_group_for_small_graphs(int L, int nnz, int min_L) {
  while (L > min_L && (long long)nnz / L < 8192) L >>= 1;
  return L;
}
// Rows below 64 float4 slots (1 x 64, 4 x 16, 8 x 8 ...) are bound by latency and instruction
// issue, not bandwidth: they want occupancy, i.e. at most ~6 KiB of LDS per wave (swept on the
// products and Reddit shapes, tools/sweep_narrow.sh: 4 x 16 forward 6.9 -> 4.5 ms with 256).
static inline bool narrow_rows(int H, int F) { return F >= 4 && H * ((F + 3) / 4) < DGNN_WAVE; }
// Tuning override (multiples of 16): DGNN_FWD_L.
static inline int fast_group_size(int H, int nnz, bool narrow = false, int slots = 0) {
  int dflt = H <= 4 ? 512 : 256;
  // narrow rows, re-swept with the packed gather (tools/sweep_narrow.sh, round 2): one head 256
  // (products shape 1 x 16: 2.5 -> 2.0 ms, 1 x 48: 3.1 -> 2.8; Reddit shape +2 %), more heads 128
  // (2 x 32: 3.9 -> 3.6, 4 x 16: 4.1 -> 3.8)
  if (narrow) dflt = H == 1 ? 256 : 128;
  // Round 3, swept again on the one-stream-per-kernel instantiations (profiles/r3s_group_size_sweep.txt,
  // products shape, scrambled labels): with seven to eight waves per SIMD from registers alone the groups
  // can be longer -- fewer cut rows to merge, fewer group prologues -- as long as a wave's LDS stage stays
  // small: a budget of staged floats that grows with the row's float4 slots (wider rows hide more),
  // divided by the floats staged per edge, to the nearest multiple of 128.  1 x 16 / 1 x 32: 384 (-4 %),
  // 1 x 48 / 1 x 64: 512 (-5 %), 2 x 16 / 4 x 8: 256 (-13 %), 2 x 32: 384 (-10 %), 2 x 64: 512 (-7 %),
  // 4 x 16: 256 (-7 %), 4 x 32: 384 (-4 %); eight heads stay at 128 (8 x 8 at 256: +45 %, its softmax
  // phase needs the waves).  DGNN_FWD_L_RULE=0: the round-2 sizes.
  if (narrow && slots > 0 && H <= 4 && env_int("DGNN_FWD_L_RULE", 1) != 0) {
    const float budget = slots <= 8 ? 1300.f : (slots <= 24 ? 1700.f : 2400.f);
    int l = (int)(budget / ((float)H + 2.5f) / 128.f + 0.5f) * 128;
    dflt = l < 128 ? 128 : (l > 512 ? 512 : l);
  }
  const int v = env_int("DGNN_FWD_L", 0);
  if (v >= 16 && v <= 1024 && v % 16 == 0) return v;
  // down to 16 edges per wave: a Cora-sized forward 39 -> 19 us (a group is a chain of dependent
  // latencies: its rows' softmax passes, then its edges four at a time)
  return shrink_group_for_small_graphs(dflt, nnz, 16);
}
__host__ __device__ static inline size_t fast_wave_lds_floats(int H, int L, bool packed) {
  // per wave: weights [L*H] + column ids [L] + flush queue [L] (+ packed gather: step list, 16 bit x L)
  return (size_t)L * (H + 2) + (packed ? (size_t)L / 2 : 0);
}
static inline size_t fast_lds_bytes(int H, int L, bool packed = false) {
  return (size_t)DGNN_WAVES_PER_BLOCK * fast_wave_lds_floats(H, L, packed) * sizeof(float);
}
// Packed gather (rows of at most 32 float4 slots): lanes per row = the power of two covering the
// row's slots, at least 4 (so at most 16 edges share one load instruction); 0 = not applicable.
static inline int fast_pack_lanes(int rowv, bool vec4) {
  // two edges per step (rows of 17..32 slots) measured SLOWER than one (2 x 64 forward 5.2 -> 5.6 ms
  // on the products shape): packing starts at four edges per step.  DGNN_PACK_MAX_S: tuning runs.
  if (!vec4 || rowv > env_int("DGNN_PACK_MAX_S", 16)) return 0;
  int s = 4;
  while (s < rowv) s <<= 1;
  return s;
}
#define DGNN_PACK_END 0x4000u      // step list entry: last step of its row segment

#ifdef DGNN_TU_FORWARD   // non-template kernel: compiled by the forward translation unit only (gather.hip)
// Row bookkeeping: empty rows are finished here (out = 0, edge_max = -1e38, edge_sum = 0:
// fused_gatconv_kernel.cu:45,66,90) and every group learns the row its first edge belongs to.
__global__ void k_fast_prep_rows(GatherParams P) {
  for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < P.m;
       r += (long long)gridDim.x * blockDim.x) {
    const int lb = P.row_ptr[r], hb = P.row_ptr[r + 1];
    if (lb == hb) {
      const int width = P.sum_heads ? P.F : P.HF;
      float *o = P.out + (size_t)r * width;
      for (int i = 0; i < width; ++i) o[i] = 0.f;
      if (P.out2 != nullptr)
        for (int i = 0; i < P.HF; ++i) P.out2[(size_t)r * P.HF + i] = 0.f;
      if (P.out_neg != nullptr) {
        for (int i = 0; i < P.HF; ++i) P.out_neg[(size_t)r * P.HF + i] = 0.f;
        for (int h = 0; h < P.H; ++h) P.a_neg[(size_t)r * P.H + h] = 0.f;
      }
      if (P.edge_max != nullptr)
        for (int h = 0; h < P.H; ++h) {
          P.edge_max[(size_t)r * P.H + h] = DGNN_NEG_INIT;
          P.edge_sum[(size_t)r * P.H + h] = 0.f;
        }
    } else {
      for (int t = (lb + P.L - 1) / P.L; t <= (hb - 1) / P.L; ++t) P.task_row0[t] = (int)r;
    }
  }
}
#endif

template <int VEC>
struct VecT;
template <>
struct VecT<4> {
  using type = float4;
  static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  // feature rows: 16 bytes per lane at any dword-aligned address (common.hpp: ld4u)
  static __device__ __forceinline__ float4 load(const float *p) { return ld4u(p); }
  static __device__ __forceinline__ void store(float *p, const float4 &v) { st4u(p, v); }
  static __device__ __forceinline__ void fma(float4 &a, float w, const float4 &x) {
    a.x = fmaf(w, x.x, a.x); a.y = fmaf(w, x.y, a.y); a.z = fmaf(w, x.z, a.z); a.w = fmaf(w, x.w, a.w);
  }
  static __device__ __forceinline__ void scale_add(float4 &a, float s, const float4 &x) { fma(a, s, x); }
  static __device__ __forceinline__ void mul(float4 &a, float s) { a.x *= s; a.y *= s; a.z *= s; a.w *= s; }
  static __device__ __forceinline__ void div(float4 &a, float s) { a.x /= s; a.y /= s; a.z /= s; a.w /= s; }
};
template <>
struct VecT<1> {
  using type = float;
  static __device__ __forceinline__ float zero() { return 0.f; }
  static __device__ __forceinline__ float load(const float *p) { return *p; }
  static __device__ __forceinline__ void store(float *p, const float &v) { *p = v; }
  static __device__ __forceinline__ void fma(float &a, float w, const float &x) { a = fmaf(w, x, a); }
  static __device__ __forceinline__ void scale_add(float &a, float s, const float &x) { a = fmaf(s, x, a); }
  static __device__ __forceinline__ void mul(float &a, float s) { a *= s; }
  static __device__ __forceinline__ void div(float &a, float s) { a /= s; }
};

// NS = vector slots per lane: a lane pass covers 64*NS slots of the feature row, so rows wider
// than 64 slots (8 x 64 as float4, 3 x 41 as scalars ...) are still gathered in ONE walk over the
// group's edges instead of one walk per 64 slots.
// HT floats (one per head) moved as the widest vectors that fit: 1 -> scalar, 2 -> float2,
// 8 -> two float4.  Works on global and LDS pointers alike.
template <int HT>
__device__ __forceinline__ void ld_heads(float (&d)[HT], const float *p) {
  if (HT % 4 == 0) {
#pragma unroll
    for (int i = 0; i < HT / 4; ++i) {
      const float4 v = ((const float4 *)p)[i];
      d[4 * i] = v.x; d[(4 * i + 1) % HT] = v.y; d[(4 * i + 2) % HT] = v.z; d[(4 * i + 3) % HT] = v.w;
    }
  } else if (HT == 2) {
    const float2 v = *(const float2 *)p;
    d[0] = v.x; d[1 % HT] = v.y;
  } else {
#pragma unroll
    for (int i = 0; i < HT; ++i) d[i] = p[i];
  }
}
template <int HT>
__device__ __forceinline__ void st_heads(float *p, const float (&d)[HT]) {
  if (HT % 4 == 0) {
#pragma unroll
    for (int i = 0; i < HT / 4; ++i)
      ((float4 *)p)[i] = make_float4(d[4 * i], d[(4 * i + 1) % HT], d[(4 * i + 2) % HT], d[(4 * i + 3) % HT]);
  } else if (HT == 2) {
    *(float2 *)p = make_float2(d[0], d[1 % HT]);
  } else {
#pragma unroll
    for (int i = 0; i < HT; ++i) p[i] = d[i];
  }
}

// One edge into the accumulator(s).  NEG: the staged weight carries the sign bit of its score; the
// plain aggregate takes |w|, the negative-score aggregate takes |w| only where the bit is set.
// `an` sums the negative-score weights themselves (a_neg when no dropout is active: then the staged
// weight IS the softmax weight; under dropout phase B reduces the unmasked weights instead).
template <bool NEG, int VEC>
__device__ __forceinline__ void acc_edge(typename VecT<VEC>::type &acc, typename VecT<VEC>::type &accn, float &an,
                                         float w, const typename VecT<VEC>::type &x) {
  if constexpr (NEG) {
    const float wa = fabsf(w);
    const float wn = (__float_as_uint(w) & 0x80000000u) ? wa : 0.f;
    VecT<VEC>::fma(acc, wa, x);
    VecT<VEC>::fma(accn, wn, x);
    an += wn;
  } else {
    VecT<VEC>::fma(acc, w, x);
  }
}

// Vector slot -> (head, float offset inside the feature row).  Float4 lanes: head h owns slots
// [h*spf, (h+1)*spf), slot q of a head the floats slot4_off(q, F) .. +3 of it (for F % 4 == 0 this
// is slot * 4); scalar lanes: slot = float.
template <int VEC>
__device__ __forceinline__ void fwd_slot_map(const GatherParams &P, int vslot, int &head, size_t &off, bool &first) {
  if (VEC == 4) {
    head = vslot / P.spf;
    const int q = vslot - head * P.spf;
    off = (size_t)head * P.F + (size_t)slot4_off(q, P.F);
    first = q == 0;
  } else {
    head = vslot / P.F;
    off = (size_t)vslot;
    first = vslot == head * P.F;
  }
}

// Store of a finished WHOLE row when the output is the sum over heads (P.sum_heads; float4 lanes, one
// slot per lane): the lane of head 0 that owns feature slot fs adds the accumulators of the lanes
// that own the same slot of heads 1 .. H-1 (they sit F/4 lanes apart) and writes out[row, fs].
__device__ __forceinline__ void store_row_summed(const GatherParams &P, int row, bool active, int head, size_t lane_off,
                                                 float4 acc) {
  const int lane = lane_id();
  const int per_head = P.F / 4;
  float4 tot = acc;
  for (int h = 1; h < P.H; ++h) {
    const int src = lane + h * per_head;
    tot.x += __shfl(acc.x, src, DGNN_WAVE);
    tot.y += __shfl(acc.y, src, DGNN_WAVE);
    tot.z += __shfl(acc.z, src, DGNN_WAVE);
    tot.w += __shfl(acc.w, src, DGNN_WAVE);
  }
  if (active && head == 0) *(float4 *)(P.out + (size_t)row * P.F + lane_off) = tot;
}

// HT = compile-time head count of the all-heads-at-once softmax phase (1, 2, 8 heads; 4 for narrow rows; 3, 5, 6, 7
// for rows of one lane pass, round 3: 6 x 8 forward 6.2 -> 3.6 ms) -- 0: run-time H; wide 4-head rows keep their own
// float4 path below.
// STEADY: the instantiation for narrow rows (fewer than 64 float4 slots), whose gather loop is
// bound by instructions per edge: it adds an unguarded steady-state loop (phase C).  Wide rows
// keep the kernel without it (the extra path costs the 4 x 64 kernel 10 % through registers).
// NEG (GAT only, float4 rows): also produce out_neg / a_neg (GatherParams); the weight staged in LDS
// carries the sign bit of its score (weights are >= 0, so the bit is free).
// PS (STEADY only) = lanes per row of the PACKED phase C as a compile-time constant (4, 8 or 16: rows
// of at most that many float4 slots), 0 = the one-edge-per-step streams.  The host launches PS ==
// P.pack_s; a kernel then carries ONE form of phase C instead of all of them (registers and code
// of the others cost the narrow kernels occupancy, see k_bwd_col_gather's PK).
template <int MODE, int VEC, int NS, int HT, bool STEADY, bool NEG = false, int PS = 0>
__global__ __launch_bounds__(DGNN_BLOCK) void k_fast_row_gather(GatherParams P) {
  static_assert(PS == 0 || (STEADY && NS == 1 && VEC == 4), "the packed phase C exists in the narrow float4 instantiations only");
  constexpr int HTA = HT > 0 ? HT : 1;     // array extent that is legal for HT == 0 too
#ifndef DGNN_FWD_HTB
#define DGNN_FWD_HTB 2                     // probe builds: 8 x 8 forward 110 VGPRs at once, 86 in fours, 71 in twos
#endif
  // heads per block of the softmax phase: blocks must tile the heads exactly (3, 5, 7 heads: all three at once / one by one)
  constexpr int HTB = HTA <= DGNN_FWD_HTB ? HTA : (HTA % DGNN_FWD_HTB == 0 ? DGNN_FWD_HTB : (HTA <= 4 ? HTA : 1));
  static_assert(HTA % HTB == 0, "head blocks tile the heads");
  using V = VecT<VEC>;
  using vec_t = typename V::type;
  extern __shared__ __align__(16) float lds_raw[];
  const int wib = wave_in_block();
  const int lane = lane_id();
  const int t = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wib);
  if (t >= P.T) return;
  DGNN_PHASE_DECL();
  if (MODE == MODE_GAT) drop_resolve(P.rng);
  const int L = P.L, H = P.H;
  constexpr bool packed = PS != 0;        // the host launches PS == P.pack_s
  float *sc = lds_raw + (size_t)wib * fast_wave_lds_floats(H, L, packed);   // [L][H] score terms -> weights
  unsigned *cidx = (unsigned *)(sc + (size_t)L * H);      // [L] column id | ROWEND
  int *dq = (int *)(cidx + L);                            // [L] flush destinations, in row order
  // packed gather: the group's load steps, one per (up to) G = 64 / pack_s consecutive edges of ONE
  // row: first edge (10 bits) | edges - 1 (4 bits) | DGNN_PACK_END
  unsigned short *steps = (unsigned short *)(dq + L);
  int nsteps = 0;

  const int e0 = t * L;
  const int e1 = min(e0 + L, P.nnz);
  const int cnt = e1 - e0;

  // The group's first row, its start and the row cache of phase B are fetched HERE: three dependent
  // memory latencies (task_row0 -> row_ptr -> the cache line of row ends) that would otherwise
  // start after phase A's waits now run under them.
  const int r_first = uniform_i32(P.task_row0[t]);
  const int rs_first = uniform_i32(P.row_ptr[r_first]);
#if DGNN_FWD_ROWCACHE
  constexpr bool ARC = MODE == MODE_GAT && HT > 0 && HT <= 4;    // attn_row of the cached rows in registers too
  int rp_first = P.row_ptr[min(r_first + 1 + lane, P.m)];
  float4 ar_first = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == MODE_GAT && HT == 0 && H == 4) ar_first = *(const float4 *)(P.attn_row + (size_t)min(r_first + lane, P.m - 1) * 4);
  float arc[HTA];
#pragma unroll
  for (int h = 0; h < HTA; ++h) arc[h] = 0.f;
  if constexpr (ARC) ld_heads<HTA>(arc, P.attn_row + (size_t)min(r_first + lane, P.m - 1) * HTA);
#endif

  // ---------------------------------------------------------------- phase A: stage
  for (int p = lane; p < cnt; p += DGNN_WAVE) {
    const int c = P.col_ind[e0 + p];
    cidx[p] = (unsigned)c;
    if (MODE == MODE_GAT) {
      if (HT > 0) {
        float tmp[HTA];
        ld_heads<HTA>(tmp, P.attn_col + (size_t)c * HTA);
        st_heads<HTA>(sc + (size_t)p * HTA, tmp);
      } else if (H == 4) {
        *(float4 *)(sc + (size_t)p * 4) = *(const float4 *)(P.attn_col + (size_t)c * 4);
      } else {
        for (int h = 0; h < H; ++h) sc[(size_t)p * H + h] = P.attn_col[(size_t)c * H + h];
      }
    } else if (MODE == MODE_VAL) {
      for (int h = 0; h < H; ++h) sc[(size_t)p * H + h] = P.edge_val[(size_t)(e0 + p) * H + h];
    } else {
      for (int h = 0; h < H; ++h)
        sc[(size_t)p * H + h] = gmm_weight(P.dim, P.pseudo + (size_t)(e0 + p) * P.dim,
                                           P.mu + (size_t)h * P.dim, P.inv_sigma + (size_t)h * P.dim);
    }
  }
  wave_lds_sync();
  DGNN_PHASE_MARK(0, 0);

  // ---------------------------------------------------------------- phase B: rows of the group
  // a_neg: without dropout the staged weight is the softmax weight itself and phase C sums it
  // along the way (one add per step); under dropout the staged weight is masked, so the unmasked
  // sum is reduced here (one more wave reduction per head and row)
  const bool an_in_b = NEG && (P.edge_mask != nullptr || P.rng.on);
  const bool batch4 = P.batch_rows != 0;
  int nflush = 0;
  {
    int r = r_first;
    int rs = rs_first;
#if DGNN_FWD_ROWCACHE
    // Row ends (and, for four heads, the rows' own score terms) of 64 consecutive rows are
    // fetched with ONE coalesced load per lane and handed out by v_readlane: a group of short
    // rows no longer pays two dependent memory latencies per row.
    int rbase = r;
    int rp_c = rp_first;
    float4 ar_c = ar_first;
#endif
    while (true) {
#if DGNN_FWD_ROWCACHE
      if (r - rbase == DGNN_WAVE) {
        rbase = r;
        rp_c = P.row_ptr[min(rbase + 1 + lane, P.m)];
        if (MODE == MODE_GAT && HT == 0 && H == 4) ar_c = *(const float4 *)(P.attn_row + (size_t)min(rbase + lane, P.m - 1) * 4);
        if constexpr (ARC) ld_heads<HTA>(arc, P.attn_row + (size_t)min(rbase + lane, P.m - 1) * HTA);
      }
      const int rk = r - rbase;
      const int re = bcast_i32(rp_c, rk);
#else
      const int re = uniform_i32(P.row_ptr[r + 1]);
#endif
      const int a = max(rs, e0) - e0, b = min(re, e1) - e0;
#if DGNN_FWD_ROWCACHE
      // ---- up to four SHORT rows in one pass (rows of at most 16 edges, each whole inside the
      // group): row q sits in DPP row q (lanes 16q .. 16q+15), so max and sum are the 16-lane DPP
      // reductions alone, four rows at a time.  On the products shape 82 % of the rows have at most
      // 16 edges and a pass per row made the softmax phase 40 % of a narrow forward's instructions.
      // Same lanes, same reduction tree inside a row as the one-row pass: identical bits.
      if constexpr (MODE == MODE_GAT && HT > 0) {
        int nb = 0;                       // rows r .. r+nb-1 qualify
        int ends[4] = {re, re, re, re};
        if (batch4 && rs >= e0 && rk + 3 < DGNN_WAVE && r + 3 < P.m) {
          int prev = rs;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int eq = bcast_i32(rp_c, rk + q);
            const bool ok = nb == q && eq > prev && eq - prev <= 16 && eq <= e1;
            if (ok) { ends[q] = eq; nb = q + 1; prev = eq; }
          }
        }
        if (nb >= 2) {
          const int qd = lane >> 4, li = lane & 15;
          const int rstart = qd == 0 ? rs : (qd == 1 ? ends[0] : (qd == 2 ? ends[1] : ends[2]));
          const int rend = qd == 0 ? ends[0] : (qd == 1 ? ends[1] : (qd == 2 ? ends[2] : ends[3]));
          const bool mine = qd < nb;
          const int p = rstart - e0 + li;
          const bool live = mine && (e0 + p) < rend;
          const int rq = r + (mine ? qd : 0);
          // heads in blocks of at most four: eight heads at once kept ~64 values per lane alive (arow, e,
          // mx, pe, sm, an, w, mask) and the 8 x 8 kernel at 110 VGPRs = four waves per SIMD
#pragma unroll
          for (int h0 = 0; h0 < HTA; h0 += HTB) {
            float arow[HTB], e[HTB], mx[HTB], pe[HTB], sm[HTB];
            if constexpr (ARC) {      // row rq's terms sit in lane rk + qd of the row cache: a cross-lane read, no memory latency
#pragma unroll
              for (int h = 0; h < HTB; ++h) arow[h] = __shfl(arc[h0 + h], rk + (mine ? qd : 0), DGNN_WAVE);
            } else {
              ld_heads<HTB>(arow, P.attn_row + (size_t)rq * HTA + h0);
            }
#pragma unroll
            for (int h = 0; h < HTB; ++h) e[h] = DGNN_NEG_INIT;
            if (live) {
              float c[HTB];
              ld_heads<HTB>(c, sc + (size_t)p * HTA + h0);
#pragma unroll
              for (int h = 0; h < HTB; ++h) e[h] = leaky_relu(arow[h] + c[h], P.slope);
            }
#pragma unroll
            for (int h = 0; h < HTB; ++h) mx[h] = row16_reduce(e[h], OpMax());
#pragma unroll
            for (int h = 0; h < HTB; ++h) pe[h] = live ? exp_nonpos(e[h] - mx[h]) : 0.f;
#pragma unroll
            for (int h = 0; h < HTB; ++h) sm[h] = row16_reduce(pe[h], OpSum());
            float an[HTB];
            if constexpr (NEG) {
              if (an_in_b) {
#pragma unroll
                for (int h = 0; h < HTB; ++h)
                  an[h] = div_pos(row16_reduce((live && e[h] < 0.f) ? pe[h] : 0.f, OpSum()), sm[h]);
              }
            }
            if (live) {
              float w[HTB];
#pragma unroll
              for (int h = 0; h < HTB; ++h) w[h] = div_pos(pe[h], sm[h]);                        // :102
              if (P.edge_mask != nullptr || P.rng.on) {
                float mk[HTB];
                if (P.rng.on) {
#pragma unroll
                  for (int h = 0; h < HTB; ++h) mk[h] = drop_uniform(P.rng, (unsigned long long)(e0 + p) * HTA + h0 + h);
                } else {
                  ld_heads<HTB>(mk, P.edge_mask + (size_t)(e0 + p) * HTA + h0);
                }
#pragma unroll
                for (int h = 0; h < HTB; ++h) {
                  if (!(mk[h] > P.attn_drop)) w[h] = 0.f;                                      // :95
                  w[h] = drop_scale(w[h], P.attn_drop);                                        // :104
                }
              }
              if constexpr (NEG) {
#pragma unroll
                for (int h = 0; h < HTB; ++h)
                  if (e[h] < 0.f) w[h] = -w[h];
              }
              st_heads<HTB>(sc + (size_t)p * HTA + h0, w);
            }
            if (mine && li == 0) {
              if (P.edge_max != nullptr) {
                st_heads<HTB>(P.edge_max + (size_t)rq * HTA + h0, mx);
                st_heads<HTB>(P.edge_sum + (size_t)rq * HTA + h0, sm);
              }
              if constexpr (NEG) {
                if (an_in_b) st_heads<HTB>(P.a_neg + (size_t)rq * HTA + h0, an);
              }
            }
          }
          if (live && e0 + p == rend - 1) cidx[p] |= DGNN_ROWEND;
          if (mine && li == 0) dq[nflush + qd] = rq;
          if constexpr (STEADY) {
            if (packed) {
              constexpr int G = DGNN_WAVE / (PS > 0 ? PS : DGNN_WAVE);
              int off = nsteps, myoff = nsteps, myns = 0;
              int prev = rs;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if (q < nb) {
                  const int ns = (ends[q] - prev + G - 1) / G;
                  if (q == qd) { myoff = off; myns = ns; }
                  off += ns;
                  prev = ends[q];
                }
              }
              if (mine && li < myns) {
                const int p0 = rstart - e0 + li * G;
                steps[myoff + li] = (unsigned short)((unsigned)p0 | ((unsigned)(min(G, rend - e0 - p0) - 1) << 10) |
                                                     (li == myns - 1 ? DGNN_PACK_END : 0u));
              }
              nsteps = off;
            }
          }
          nflush += nb;
          const int last_end = nb == 2 ? ends[1] : (nb == 3 ? ends[2] : ends[3]);
          r += nb - 1;
          if (last_end >= e1 || r + 1 >= P.m) break;
          rs = last_end;
          ++r;
          continue;
        }
      }
#endif
      if (b > a) {
        const bool whole = (rs >= e0) && (re <= e1);
        const int slot = (rs < e0) ? 2 * t : 2 * t + 1;
        float *slot_ptr = P.slots + (size_t)slot * P.slot_stride;
        if (MODE == MODE_GAT && HT > 0 && (b - a) <= DGNN_WAVE) {
          // the whole segment fits one wave pass: all HT heads at once (lane = edge)
          const int p = a + lane;
          const bool live = p < b;
#pragma unroll
          for (int h0 = 0; h0 < HTA; h0 += HTB) {       // heads in blocks of at most four (registers, see above)
            float arow[HTB], e[HTB], mx[HTB], pe[HTB], sm[HTB];
            if constexpr (ARC) {
#pragma unroll
              for (int h = 0; h < HTB; ++h) arow[h] = bcast_f32(arc[h0 + h], rk);
            } else {
              ld_heads<HTB>(arow, P.attn_row + (size_t)r * HTA + h0);
            }
#pragma unroll
            for (int h = 0; h < HTB; ++h) e[h] = DGNN_NEG_INIT;
            if (live) {
              float c[HTB];
              ld_heads<HTB>(c, sc + (size_t)p * HTA + h0);
#pragma unroll
              for (int h = 0; h < HTB; ++h) e[h] = leaky_relu(arow[h] + c[h], P.slope);
            }
#pragma unroll
            for (int h = 0; h < HTB; ++h) mx[h] = wave_max(e[h]);
#pragma unroll
            for (int h = 0; h < HTB; ++h) pe[h] = live ? exp_nonpos(e[h] - mx[h]) : 0.f;
#pragma unroll
            for (int h = 0; h < HTB; ++h) sm[h] = wave_sum(pe[h]);
            float an[HTB];
            if constexpr (NEG) {
              if (an_in_b) {
#pragma unroll
                for (int h = 0; h < HTB; ++h) {
                  an[h] = wave_sum((live && e[h] < 0.f) ? pe[h] : 0.f);
                  if (whole) an[h] = div_pos(an[h], sm[h]);
                }
              }
            }
            if (live) {
              float w[HTB];
#pragma unroll
              for (int h = 0; h < HTB; ++h) w[h] = whole ? div_pos(pe[h], sm[h]) : pe[h];      // :102
              if (P.edge_mask != nullptr || P.rng.on) {
                float mk[HTB];
                if (P.rng.on) {
#pragma unroll
                  for (int h = 0; h < HTB; ++h) mk[h] = drop_uniform(P.rng, (unsigned long long)(e0 + p) * HTA + h0 + h);
                } else {
                  ld_heads<HTB>(mk, P.edge_mask + (size_t)(e0 + p) * HTA + h0);
                }
#pragma unroll
                for (int h = 0; h < HTB; ++h) {
                  if (!(mk[h] > P.attn_drop)) w[h] = 0.f;                                      // :95
                  w[h] = drop_scale(w[h], P.attn_drop);                                        // :104
                }
              }
              if constexpr (NEG) {
#pragma unroll
                for (int h = 0; h < HTB; ++h)
                  if (e[h] < 0.f) w[h] = -w[h];
              }
              st_heads<HTB>(sc + (size_t)p * HTA + h0, w);
            }
            if (lane == 0) {
              if (whole) {
                if (P.edge_max != nullptr) {
                  st_heads<HTB>(P.edge_max + (size_t)r * HTA + h0, mx);
                  st_heads<HTB>(P.edge_sum + (size_t)r * HTA + h0, sm);
                }
                if constexpr (NEG) {
                  if (an_in_b) st_heads<HTB>(P.a_neg + (size_t)r * HTA + h0, an);
                }
              } else {
                st_heads<HTB>(slot_ptr + P.HF + h0, mx);
                st_heads<HTB>(slot_ptr + P.HF + HTA + h0, sm);
                if constexpr (NEG) {
                  if (an_in_b) st_heads<HTB>(slot_ptr + P.HF + 2 * HTA + h0, an);
                }
              }
            }
          }
        } else if (MODE == MODE_GAT && H == 4 && (b - a) <= DGNN_WAVE) {
          // common case: the whole segment fits one wave pass, all four heads at once
          const int p = a + lane;
          const bool live = p < b;
#if DGNN_FWD_ROWCACHE
          const float4 ar4 = make_float4(bcast_f32(ar_c.x, rk), bcast_f32(ar_c.y, rk), bcast_f32(ar_c.z, rk),
                                         bcast_f32(ar_c.w, rk));
#else
          const float4 ar4 = *(const float4 *)(P.attn_row + (size_t)r * 4);
#endif
          float4 e4 = make_float4(DGNN_NEG_INIT, DGNN_NEG_INIT, DGNN_NEG_INIT, DGNN_NEG_INIT);
          if (live) {
            const float4 c4 = *(const float4 *)(sc + (size_t)p * 4);
            e4.x = leaky_relu(ar4.x + c4.x, P.slope);
            e4.y = leaky_relu(ar4.y + c4.y, P.slope);
            e4.z = leaky_relu(ar4.z + c4.z, P.slope);
            e4.w = leaky_relu(ar4.w + c4.w, P.slope);
          }
          float4 mx;
          mx.x = wave_max(e4.x); mx.y = wave_max(e4.y); mx.z = wave_max(e4.z); mx.w = wave_max(e4.w);
          float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
          if (live) {
            pe.x = exp_nonpos(e4.x - mx.x); pe.y = exp_nonpos(e4.y - mx.y);
            pe.z = exp_nonpos(e4.z - mx.z); pe.w = exp_nonpos(e4.w - mx.w);
          }
          float4 sm;
          sm.x = wave_sum(pe.x); sm.y = wave_sum(pe.y); sm.z = wave_sum(pe.z); sm.w = wave_sum(pe.w);
          float4 an4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (NEG) {
            if (an_in_b) {
              an4.x = wave_sum(e4.x < 0.f ? pe.x : 0.f); an4.y = wave_sum(e4.y < 0.f ? pe.y : 0.f);
              an4.z = wave_sum(e4.z < 0.f ? pe.z : 0.f); an4.w = wave_sum(e4.w < 0.f ? pe.w : 0.f);
              if (whole) { an4.x = div_pos(an4.x, sm.x); an4.y = div_pos(an4.y, sm.y); an4.z = div_pos(an4.z, sm.z); an4.w = div_pos(an4.w, sm.w); }
            }
          }
          if (live) {
            float4 w4 = pe;
            if (whole) { w4.x = div_pos(w4.x, sm.x); w4.y = div_pos(w4.y, sm.y); w4.z = div_pos(w4.z, sm.z); w4.w = div_pos(w4.w, sm.w); }   // :102
            if (P.edge_mask != nullptr || P.rng.on) {
              const float4 mk = P.rng.on ? drop_uniform4(P.rng, (unsigned long long)(e0 + p))
                                         : *(const float4 *)(P.edge_mask + (size_t)(e0 + p) * 4);
              if (!(mk.x > P.attn_drop)) w4.x = 0.f;                                 // :95
              if (!(mk.y > P.attn_drop)) w4.y = 0.f;
              if (!(mk.z > P.attn_drop)) w4.z = 0.f;
              if (!(mk.w > P.attn_drop)) w4.w = 0.f;
              w4.x = drop_scale(w4.x, P.attn_drop); w4.y = drop_scale(w4.y, P.attn_drop);  // :104
              w4.z = drop_scale(w4.z, P.attn_drop); w4.w = drop_scale(w4.w, P.attn_drop);
            }
            if constexpr (NEG) {
              if (e4.x < 0.f) w4.x = -w4.x;
              if (e4.y < 0.f) w4.y = -w4.y;
              if (e4.z < 0.f) w4.z = -w4.z;
              if (e4.w < 0.f) w4.w = -w4.w;
            }
            *(float4 *)(sc + (size_t)p * 4) = w4;
          }
          if (lane == 0) {
            if (whole) {
              if (P.edge_max != nullptr) {
                *(float4 *)(P.edge_max + (size_t)r * 4) = mx;
                *(float4 *)(P.edge_sum + (size_t)r * 4) = sm;
              }
              if constexpr (NEG) {
                if (an_in_b) *(float4 *)(P.a_neg + (size_t)r * 4) = an4;
              }
            } else {
              *(float4 *)(slot_ptr + P.HF) = mx;
              *(float4 *)(slot_ptr + P.HF + 4) = sm;
              if constexpr (NEG) {
                if (an_in_b) *(float4 *)(slot_ptr + P.HF + 8) = an4;
              }
            }
          }
        } else if (MODE == MODE_GAT) {
          for (int h = 0; h < H; ++h) {
            const float arh = P.attn_row[(size_t)r * H + h];
            float mx = DGNN_NEG_INIT;
            for (int p = a + lane; p < b; p += DGNN_WAVE)
              mx = fmaxf(mx, leaky_relu(arh + sc[(size_t)p * H + h], P.slope));
            mx = wave_max(mx);
            float sm = 0.f, an = 0.f;
            for (int p = a + lane; p < b; p += DGNN_WAVE) {
              const float ev = leaky_relu(arh + sc[(size_t)p * H + h], P.slope);
              const float pe = exp_nonpos(ev - mx);
              sc[(size_t)p * H + h] = (NEG && ev < 0.f) ? -pe : pe;      // NEG: sign bit = (score < 0)
              sm += pe;
              if (NEG && ev < 0.f) an += pe;
            }
            sm = wave_sum(sm);
            if constexpr (NEG) {
              if (an_in_b) {
                an = wave_sum(an);
                if (whole) an = div_pos(an, sm);
              }
            }
            for (int p = a + lane; p < b; p += DGNN_WAVE) {
              float w = sc[(size_t)p * H + h];
              const bool ng = NEG && (__float_as_uint(w) & 0x80000000u) != 0u;
              if constexpr (NEG) w = fabsf(w);
              if (whole) w = div_pos(w, sm);               // exp(e-max)/sum per edge, :102
              if (P.edge_mask != nullptr || P.rng.on) {
                const float mk = P.rng.on ? drop_uniform(P.rng, (unsigned long long)(e0 + p) * H + h)
                                          : P.edge_mask[(size_t)(e0 + p) * H + h];
                if (!(mk > P.attn_drop)) w = 0.f;   // :95
                w = drop_scale(w, P.attn_drop);                                       // :104
              }
              sc[(size_t)p * H + h] = ng ? -w : w;
            }
            if (lane == 0) {
              if (whole) {
                if (P.edge_max != nullptr) {
                  P.edge_max[(size_t)r * H + h] = mx;
                  P.edge_sum[(size_t)r * H + h] = sm;
                }
                if constexpr (NEG) {
                  if (an_in_b) P.a_neg[(size_t)r * H + h] = an;
                }
              } else {
                slot_ptr[P.HF + h] = mx;
                slot_ptr[P.HF + H + h] = sm;
                if constexpr (NEG) {
                  if (an_in_b) slot_ptr[P.HF + 2 * H + h] = an;
                }
              }
            }
          }
        }
        if (lane == 0) {
          cidx[b - 1] |= DGNN_ROWEND;
          dq[nflush] = whole ? r : -(slot + 1);
        }
        ++nflush;
        if constexpr (STEADY) {
          if (packed) {
            constexpr int G = DGNN_WAVE / (PS > 0 ? PS : DGNN_WAVE);
            const int ns = (b - a + G - 1) / G;
            for (int j = lane; j < ns; j += DGNN_WAVE) {
              const int p0 = a + j * G;
              steps[nsteps + j] = (unsigned short)((unsigned)p0 | ((unsigned)(min(G, b - p0) - 1) << 10) |
                                                   (j == ns - 1 ? DGNN_PACK_END : 0u));
            }
            nsteps += ns;
          }
        }
      }
      if (re >= e1 || r + 1 >= P.m) break;   // second test: never walk past the last row
      rs = re;
      ++r;
    }
  }
  wave_lds_sync();
  DGNN_PHASE_MARK(0, 1);

  // ---------------------------------------------------------------- phase C, packed (narrow rows)
  // A row of S <= 32 float4 slots leaves 64 - S lanes of a one-edge-per-step walk idle and the
  // walk bound by instructions per EDGE (~30 VALU + ~20 SALU, profiles/r2a_pmc_sq_gat_1x64_*).
  // Here the wave's lanes form G = 64 / S groups; one step loads G consecutive edges of ONE row
  // (group g: edge p0 + g, lane s of the group: slot s), every group keeps its own partial sum,
  // and at the row's last step the G partials are added across groups (xor shuffles) and stored.
  // Steps never span rows (the step list is built in phase B), so no lane ever mixes rows; idle
  // groups of a row's last step re-read the step's first edge with weight 0.
  if constexpr (PS != 0) {
    {
      constexpr int S = PS;
      const int g = lane / S, sl = lane % S;
      const bool active = sl < P.rowv;
      int head;
      size_t lane_off;
      bool first_of_head;
      fwd_slot_map<4>(P, active ? sl : 0, head, lane_off, first_of_head);
      const float *wbase = sc + head;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accn = acc;
      float an = 0.f;
      const bool an_leader = active && g == 0 && first_of_head;
      float4 ring[DGNN_PACK_U];
      float wring[DGNN_PACK_U];
      int q = 0;
      auto issue = [&](int j, float4 &x, float &w) {
        const unsigned it = (unsigned)uniform_i32((int)steps[j]);
        const int p0 = (int)(it & 1023u), n = (int)((it >> 10) & 15u) + 1;
        const bool live = g < n;
        const int pe = p0 + (live ? g : 0);
        const unsigned c = cidx[pe] & ~DGNN_ROWEND;
        w = live ? wbase[(size_t)pe * H] : 0.f;
        x = ld4u(P.in_feat + (size_t)c * P.HF + lane_off);
      };
      auto consume = [&](const float4 &x, float w, int j) {
        acc_edge<NEG, 4>(acc, accn, an, w, x);
        if (uniform_i32((int)steps[j]) & (int)DGNN_PACK_END) {
#pragma unroll
          for (int off = S; off < DGNN_WAVE; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off, DGNN_WAVE);
            acc.y += __shfl_xor(acc.y, off, DGNN_WAVE);
            acc.z += __shfl_xor(acc.z, off, DGNN_WAVE);
            acc.w += __shfl_xor(acc.w, off, DGNN_WAVE);
            if constexpr (NEG) {
              accn.x += __shfl_xor(accn.x, off, DGNN_WAVE);
              accn.y += __shfl_xor(accn.y, off, DGNN_WAVE);
              accn.z += __shfl_xor(accn.z, off, DGNN_WAVE);
              accn.w += __shfl_xor(accn.w, off, DGNN_WAVE);
              an += __shfl_xor(an, off, DGNN_WAVE);
            }
          }
          const int d = uniform_i32(dq[q]);
          ++q;
          if (MODE != MODE_GAT && P.sum_heads && d >= 0) {
            store_row_summed(P, d, active && g == 0, head, lane_off, acc);
          } else {
            float *dst = (d >= 0) ? P.out + (size_t)d * P.HF : P.slots + (size_t)(-(d + 1)) * P.slot_stride;
            if (active && g == 0) st4u(dst + lane_off, acc);
            if constexpr (NEG) {
              if (P.out2 != nullptr && d >= 0 && active && g == 0) st4u(P.out2 + (size_t)d * P.HF + lane_off, acc);
              float *dn = (d >= 0) ? P.out_neg + (size_t)d * P.HF
                                   : P.slots + (size_t)(-(d + 1)) * P.slot_stride + P.slot_neg;
              if (active && g == 0) st4u(dn + lane_off, accn);
              accn = make_float4(0.f, 0.f, 0.f, 0.f);
              if (!an_in_b && an_leader) {
                if (d >= 0) P.a_neg[(size_t)d * H + head] = an;
                else P.slots[(size_t)(-(d + 1)) * P.slot_stride + P.HF + 2 * H + head] = an;
              }
              an = 0.f;
            }
          }
          acc = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      int j = 0;
      if (nsteps >= 2 * DGNN_PACK_U) {
#pragma unroll
        for (int k = 0; k < DGNN_PACK_U; ++k) {
          issue(k, ring[k], wring[k]);
          __builtin_amdgcn_sched_barrier(0);
        }
        for (; j + 2 * DGNN_PACK_U <= nsteps; j += DGNN_PACK_U) {
#pragma unroll
          for (int k = 0; k < DGNN_PACK_U; ++k) {
            consume(ring[k], wring[k], j + k);
            issue(j + k + DGNN_PACK_U, ring[k], wring[k]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // the ring holds steps j .. j+U-1; at most U-1 further steps remain
#pragma unroll
        for (int k = 0; k < DGNN_PACK_U; ++k) {
          const float4 x = ring[k];
          const float w = wring[k];
          if (j + k + DGNN_PACK_U < nsteps) issue(j + k + DGNN_PACK_U, ring[k], wring[k]);
          consume(x, w, j + k);
        }
        j += DGNN_PACK_U;
#pragma unroll
        for (int k = 0; k < DGNN_PACK_U; ++k)
          if (j + k < nsteps) consume(ring[k], wring[k], j + k);
      } else {
        for (; j < nsteps; ++j) {
          float4 x;
          float w;
          issue(j, x, w);
          consume(x, w, j);
        }
      }
      DGNN_PHASE_MARK(0, 2);
      return;
    }
  }

  // ---------------------------------------------------------------- phase C: pipelined gather
  const int span = DGNN_WAVE * NS;
  const int nfb = (P.rowv + span - 1) / span;
  // With a dropout mask a dropped (edge, head) has weight exactly 0: its lanes skip the load,
  // so the dropped share of the feature traffic never leaves HBM (the reference loads the row
  // and multiplies by 0, fused_gatconv_kernel.cu:95-117; identical for finite features).
  const bool skip0 = (MODE == MODE_GAT) && (P.edge_mask != nullptr || P.rng.on);
  for (int fb = 0; fb < nfb; ++fb) {
    bool active[NS];
    size_t lane_off[NS];
    const float *wbase[NS];
    vec_t acc[NS], accn[NS];
    float an[NS];
    bool an_leader[NS];
    int an_head[NS];
    vec_t ring[DGNN_FAST_U][NS];
    float wring[DGNN_FAST_U][NS];
    int hd0 = 0;                           // head of the lane's first slot (summed store)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int vslot = fb * span + s * DGNN_WAVE + lane;
      active[s] = vslot < P.rowv;
      int head;
      bool first_of_head;
      fwd_slot_map<VEC>(P, active[s] ? vslot : 0, head, lane_off[s], first_of_head);
      wbase[s] = sc + head;
      acc[s] = V::zero();
      accn[s] = V::zero();
      an[s] = 0.f;
      an_head[s] = head;
      an_leader[s] = active[s] && first_of_head;
      if (s == 0) hd0 = head;
    }
    int q = 0;
    bool took_steady = false;
    if constexpr (STEADY) {
    // One edge of the stream: accumulate (w, x) and flush at a row end.
    auto consume = [&](const vec_t (&x)[NS], const float (&w)[NS], unsigned ci) {
#pragma unroll
      for (int s = 0; s < NS; ++s) acc_edge<NEG, VEC>(acc[s], accn[s], an[s], w[s], x[s]);
      if (ci & DGNN_ROWEND) {
        const int d = uniform_i32(dq[q]);
        ++q;
        if constexpr (MODE != MODE_GAT && NS == 1 && VEC == 4) {
          if (P.sum_heads && d >= 0) {
            store_row_summed(P, d, active[0], hd0, lane_off[0], acc[0]);
            acc[0] = V::zero();
            return;
          }
        }
        float *dst = (d >= 0) ? P.out + (size_t)d * P.HF : P.slots + (size_t)(-(d + 1)) * P.slot_stride;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (active[s]) V::store(dst + lane_off[s], acc[s]);
          if constexpr (NEG) {
            if (P.out2 != nullptr && d >= 0 && active[s]) V::store(P.out2 + (size_t)d * P.HF + lane_off[s], acc[s]);
          }
          acc[s] = V::zero();
        }
        if constexpr (NEG) {
          float *dn = (d >= 0) ? P.out_neg + (size_t)d * P.HF
                               : P.slots + (size_t)(-(d + 1)) * P.slot_stride + P.slot_neg;
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            if (active[s]) V::store(dn + lane_off[s], accn[s]);
            accn[s] = V::zero();
            if (!an_in_b && an_leader[s]) {
              if (d >= 0) P.a_neg[(size_t)d * H + an_head[s]] = an[s];
              else P.slots[(size_t)(-(d + 1)) * P.slot_stride + P.HF + 2 * H + an_head[s]] = an[s];
            }
            an[s] = 0.f;
          }
        }
      }
    };
    // Guarded step of the pipeline (pipeline fill, tails, dropout): loads sit under conditions.
    auto guarded_step = [&](int p, vec_t (&slot)[NS], float (&wslot)[NS]) {
      vec_t x[NS];
      float w[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        x[s] = slot[s];
        w[s] = wslot[s];
      }
      const unsigned ci = (unsigned)uniform_i32((int)cidx[p]);
      if (p + DGNN_FAST_U < cnt) {
        const unsigned c = (unsigned)uniform_i32((int)(cidx[p + DGNN_FAST_U] & ~DGNN_ROWEND));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          wslot[s] = wbase[s][(size_t)(p + DGNN_FAST_U) * H];
          if (active[s] && !(skip0 && wslot[s] == 0.f))
            slot[s] = V::load(P.in_feat + (size_t)c * P.HF + lane_off[s]);
          else
            slot[s] = V::zero();
        }
      }
      consume(x, w, ci);
    };
    if (P.steady && !skip0 && cnt >= 2 * DGNN_FAST_U) {
      took_steady = true;
      // Steady path for NARROW rows (every full group without dropout), where the loop is bound
      // by its instruction count, not by bandwidth: fill and loop body issue their gathers
      // unconditionally (idle lanes re-read slot 0 of the row) and a slot is consumed before it is
      // refilled, which removes the guards and the register copies of the refilled slots (and
      // lets the compiler count the loads in flight instead of draining them before every use).
#pragma unroll
      for (int k = 0; k < DGNN_FAST_U; ++k) {
        const unsigned c = uniform_i32((int)(cidx[k] & ~DGNN_ROWEND));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          wring[k][s] = wbase[s][(size_t)k * H];
          ring[k][s] = V::load(P.in_feat + (size_t)c * P.HF + lane_off[s]);
        }
        __builtin_amdgcn_sched_barrier(0);   // issue order = slot order (the waits count loads)
      }
      int j = 0;
      for (; j + 2 * DGNN_FAST_U <= cnt; j += DGNN_FAST_U) {
#pragma unroll
        for (int k = 0; k < DGNN_FAST_U; ++k) {
          const int p = j + k;
          const unsigned ci = (unsigned)uniform_i32((int)cidx[p]);
          const unsigned c = (unsigned)uniform_i32((int)(cidx[p + DGNN_FAST_U] & ~DGNN_ROWEND));
          float wnext[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) wnext[s] = wbase[s][(size_t)(p + DGNN_FAST_U) * H];
          consume(ring[k], wring[k], ci);
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            wring[k][s] = wnext[s];
            ring[k][s] = V::load(P.in_feat + (size_t)c * P.HF + lane_off[s]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      for (; j < cnt; j += DGNN_FAST_U) {
#pragma unroll
        for (int k = 0; k < DGNN_FAST_U; ++k)
          if (j + k < cnt) guarded_step(j + k, ring[k], wring[k]);
      }
    }
    }
    if (!took_steady) {
      // Guarded pipeline exactly as tuned for wide rows (bandwidth-bound: 4 x 64 and up).
#pragma unroll
      for (int k = 0; k < DGNN_FAST_U; ++k) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          ring[k][s] = V::zero();
          wring[k][s] = 0.f;
        }
        if (k < cnt) {
          const unsigned c = uniform_i32((int)(cidx[k] & ~DGNN_ROWEND));
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            wring[k][s] = wbase[s][(size_t)k * H];
            if (active[s] && !(skip0 && wring[k][s] == 0.f))
              ring[k][s] = V::load(P.in_feat + (size_t)c * P.HF + lane_off[s]);
          }
        }
      }
      for (int j = 0; j < cnt; j += DGNN_FAST_U) {
#pragma unroll
        for (int k = 0; k < DGNN_FAST_U; ++k) {
          const int p = j + k;
          if (p < cnt) {
            vec_t x[NS];
            float w[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              x[s] = ring[k][s];
              w[s] = wring[k][s];
            }
            const unsigned ci = (unsigned)uniform_i32((int)cidx[p]);
            if (p + DGNN_FAST_U < cnt) {
              const unsigned c = (unsigned)uniform_i32((int)(cidx[p + DGNN_FAST_U] & ~DGNN_ROWEND));
#pragma unroll
              for (int s = 0; s < NS; ++s) {
                wring[k][s] = wbase[s][(size_t)(p + DGNN_FAST_U) * H];
                if (active[s] && !(skip0 && wring[k][s] == 0.f))
                  ring[k][s] = V::load(P.in_feat + (size_t)c * P.HF + lane_off[s]);
                else
                  ring[k][s] = V::zero();
              }
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) acc_edge<NEG, VEC>(acc[s], accn[s], an[s], w[s], x[s]);
            if (ci & DGNN_ROWEND) {
              const int d = uniform_i32(dq[q]);
              ++q;
              bool stored = false;
              if constexpr (MODE != MODE_GAT && NS == 1 && VEC == 4) {
                if (P.sum_heads && d >= 0) {
                  store_row_summed(P, d, active[0], hd0, lane_off[0], acc[0]);
                  acc[0] = V::zero();
                  stored = true;
                }
              }
              if (!stored) {
                float *dst = (d >= 0) ? P.out + (size_t)d * P.HF
                                      : P.slots + (size_t)(-(d + 1)) * P.slot_stride;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                  if (active[s]) V::store(dst + lane_off[s], acc[s]);
                  if constexpr (NEG) {
                    if (P.out2 != nullptr && d >= 0 && active[s]) V::store(P.out2 + (size_t)d * P.HF + lane_off[s], acc[s]);
                  }
                  acc[s] = V::zero();
                }
                if constexpr (NEG) {
                  float *dn = (d >= 0) ? P.out_neg + (size_t)d * P.HF
                                       : P.slots + (size_t)(-(d + 1)) * P.slot_stride + P.slot_neg;
#pragma unroll
                  for (int s = 0; s < NS; ++s) {
                    if (active[s]) V::store(dn + lane_off[s], accn[s]);
                    accn[s] = V::zero();
                    if (!an_in_b && an_leader[s]) {
                      if (d >= 0) P.a_neg[(size_t)d * H + an_head[s]] = an[s];
                      else P.slots[(size_t)(-(d + 1)) * P.slot_stride + P.HF + 2 * H + an_head[s]] = an[s];
                    }
                    an[s] = 0.f;
                  }
                }
              }
            }
          }
        }
      }
    }
  }
  DGNN_PHASE_MARK(0, 3);
}

// Merge the partial slots of rows cut by a group boundary.  One wave per boundary t (1..T-1);
// the wave of the FIRST boundary inside a row owns that row's merge, pieces are combined in
// group order: M = max m_p, S = sum s_p*exp(m_p-M), out = sum acc_p*exp(m_p-M) / S.
//
// A hub row is cut into hundreds of pieces and one wave walking them is a dependent chain of loads
// (0.2 ms for the longest row of the products shape -- and again in every chunk of a partitioned
// block).  Rows of more than DGNN_FIX_C pieces are therefore merged in two launches: in `phase` 0
// the waves of every DGNN_FIX_C-th boundary inside such a row merge their DGNN_FIX_C pieces into
// the first of them, in place and unnormalised (a partial triple (M, S, acc) is itself a slot);
// in phase 1 the owner merges those.  The tree depends on the piece count alone: deterministic.
// (Rows wider than one lane pass keep the one-level walk: their lanes read the piece statistics
// once per pass.)
#ifndef DGNN_FIX_C
#define DGNN_FIX_C 16
#endif
template <int MODE>
__global__ __launch_bounds__(DGNN_BLOCK) void k_fast_fixup(GatherParams P, int phase) {
  const int lane = lane_id();
  const int t = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block()) + 1;
  if (t >= P.T) return;
  const int L = P.L, H = P.H;
  const int r = uniform_i32(P.task_row0[t]);
  if (phase == 0) {
    // cheap necessary condition before the dependent row_ptr loads: a row of more than DGNN_FIX_C
    // pieces also covers the start of the group DGNN_FIX_C - 1 boundaries back (first-level waves
    // k >= 1) or DGNN_FIX_C - 2 ahead (k = 0); nearly every wave of this sweep leaves here
    const int back = uniform_i32(P.task_row0[max(t - (DGNN_FIX_C - 1), 0)]);
    const int ahead = uniform_i32(P.task_row0[min(t + DGNN_FIX_C - 2, P.T - 1)]);
    if (back != r && ahead != r) return;
  }
  const int rs = uniform_i32(P.row_ptr[r]), re = uniform_i32(P.row_ptr[r + 1]);
  const long long edge = (long long)t * L;
  if (rs >= edge) return;                    // the row starts exactly on the boundary: not cut here
  const int t0 = rs / L + 1;                 // the first boundary inside the row: its wave owns the merge
  const int t_last = (re - 1) / L;
  // piece i = 0 is the tail slot of group t0-1, pieces 1.. are the head slots of groups t0..t_last
  const int npieces = t_last - t0 + 2;
  const bool vec4 = P.vec4 != 0;
  const int nslots = P.rowv;
  const bool two_level = npieces > DGNN_FIX_C && nslots <= DGNN_WAVE;
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
    if (cnt <= 1) return;                    // a lone last piece is its own partial
  } else {
    if (t != t0) return;                     // an earlier boundary owns this row
    if (two_level) {
      stride = DGNN_FIX_C;
      cnt = (npieces + DGNN_FIX_C - 1) / DGNN_FIX_C;
    }
  }
  auto slot_of = [&](int i) -> float * {     // i-th piece of this wave's list
    const int pi = first + i * stride;
    const int s = (pi == 0) ? 2 * (t0 - 1) + 1 : 2 * (t0 + pi - 1);
    return P.slots + (size_t)s * P.slot_stride;
  };
  const bool final_pass = phase != 0;
  for (int fb = 0; fb * DGNN_WAVE < nslots; ++fb) {
    const int vs = fb * DGNN_WAVE + lane;
    const bool active = vs < nslots;
    int head;
    size_t slot_off;
    bool first_of_head;
    if (vec4) fwd_slot_map<4>(P, active ? vs : 0, head, slot_off, first_of_head);
    else fwd_slot_map<1>(P, active ? vs : 0, head, slot_off, first_of_head);
    const int fid = (int)slot_off;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // Pieces are loaded in batches whose loads are issued together, the arithmetic then runs over
    // them in piece order (same sums as one piece at a time).  Nearly every cut row has two pieces:
    // those take a batch of two (a batch of eight re-reads the last piece six times, and the
    // address pipe, not the bytes, bounds this sweep); a list that fits one batch finds its
    // maximum in the batch it accumulates from instead of in a walk of its own.
    if (MODE == MODE_GAT) {
      float M = DGNN_NEG_INIT;
      float S = 0.f;
      // the negative-score aggregate and weight sum of the cut row ride the same batches (float4 rows only)
      const bool neg = P.out_neg != nullptr;
      float an = 0.f, nacc[4] = {0.f, 0.f, 0.f, 0.f};
      auto merge = [&](auto fbc) {
        constexpr int FB = decltype(fbc)::value;
        const bool one = cnt <= FB;
        if (!one)
          for (int i0 = 0; i0 < cnt; i0 += FB) {
            float m[FB];
#pragma unroll
            for (int j = 0; j < FB; ++j) m[j] = slot_of(min(i0 + j, cnt - 1))[P.HF + head];
#pragma unroll
            for (int j = 0; j < FB; ++j) M = fmaxf(M, m[j]);
          }
        for (int i0 = 0; i0 < cnt; i0 += FB) {
          float m[FB], sm[FB], sa[FB];
          float4 v[FB], vn[FB];
#pragma unroll
          for (int j = 0; j < FB; ++j) {
            const float *sp = slot_of(min(i0 + j, cnt - 1));
            m[j] = sp[P.HF + head];
            sm[j] = sp[P.HF + H + head];
            if (vec4) v[j] = ld4u(sp + fid);
            else v[j] = make_float4(sp[fid], 0.f, 0.f, 0.f);
            if (neg) {
              sa[j] = sp[P.HF + 2 * H + head];
              vn[j] = ld4u(sp + P.slot_neg + fid);
            }
          }
          if (one) {
#pragma unroll
            for (int j = 0; j < FB; ++j) M = fmaxf(M, m[j]);
          }
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if (i0 + j < cnt) {
              const float fct = expf(m[j] - M);
              S = fmaf(sm[j], fct, S);
              acc[0] = fmaf(v[j].x, fct, acc[0]); acc[1] = fmaf(v[j].y, fct, acc[1]);
              acc[2] = fmaf(v[j].z, fct, acc[2]); acc[3] = fmaf(v[j].w, fct, acc[3]);
              if (neg) {
                an = fmaf(sa[j], fct, an);
                nacc[0] = fmaf(vn[j].x, fct, nacc[0]); nacc[1] = fmaf(vn[j].y, fct, nacc[1]);
                nacc[2] = fmaf(vn[j].z, fct, nacc[2]); nacc[3] = fmaf(vn[j].w, fct, nacc[3]);
              }
            }
        }
      };
      if (cnt <= 2) merge(std::integral_constant<int, 2>{});
      else merge(std::integral_constant<int, 8>{});
      if (!final_pass) {
        // the partial triple replaces the first piece (every lane has consumed its loads: the values
        // stored depend on them)
        float *sp = slot_of(0);
        if (active) {
          if (vec4) st4u(sp + fid, make_float4(acc[0], acc[1], acc[2], acc[3]));
          else sp[fid] = acc[0];
          if (neg) st4u(sp + P.slot_neg + fid, make_float4(nacc[0], nacc[1], nacc[2], nacc[3]));
          if (first_of_head) {
            sp[P.HF + head] = M;
            sp[P.HF + H + head] = S;
            if (neg) sp[P.HF + 2 * H + head] = an;
          }
        }
        continue;
      }
      if (neg && active) {
        st4u(P.out_neg + (size_t)r * P.HF + fid, make_float4(nacc[0] / S, nacc[1] / S, nacc[2] / S, nacc[3] / S));
        if (first_of_head) P.a_neg[(size_t)r * H + head] = an / S;
      }
      if (active) {
        if (vec4)
          st4u(P.out + (size_t)r * P.HF + fid, make_float4(acc[0] / S, acc[1] / S, acc[2] / S, acc[3] / S));
        else
          P.out[(size_t)r * P.HF + fid] = acc[0] / S;
        if (vec4 && P.out2 != nullptr)
          st4u(P.out2 + (size_t)r * P.HF + fid, make_float4(acc[0] / S, acc[1] / S, acc[2] / S, acc[3] / S));
        if (P.edge_max != nullptr && first_of_head) {
          P.edge_max[(size_t)r * H + head] = M;
          P.edge_sum[(size_t)r * H + head] = S;
        }
      }
    } else {
      auto merge = [&](auto fbc) {
        constexpr int FB = decltype(fbc)::value;
        for (int i0 = 0; i0 < cnt; i0 += FB) {
          float4 v[FB];
#pragma unroll
          for (int j = 0; j < FB; ++j) {
            const float *sp = slot_of(min(i0 + j, cnt - 1));
            if (vec4) v[j] = ld4u(sp + fid);
            else v[j] = make_float4(sp[fid], 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int j = 0; j < FB; ++j)
            if (i0 + j < cnt) { acc[0] += v[j].x; acc[1] += v[j].y; acc[2] += v[j].z; acc[3] += v[j].w; }
        }
      };
      if (cnt <= 2) merge(std::integral_constant<int, 2>{});
      else merge(std::integral_constant<int, 8>{});
      if (!final_pass) {
        float *sp = slot_of(0);
        if (active) {
          if (vec4) st4u(sp + fid, make_float4(acc[0], acc[1], acc[2], acc[3]));
          else sp[fid] = acc[0];
        }
        continue;
      }
      if (P.sum_heads) {     // host guarantees float4 lanes and a single lane pass here
        store_row_summed(P, r, active, head, (size_t)(fid - head * P.F), make_float4(acc[0], acc[1], acc[2], acc[3]));
      } else if (active) {
        if (vec4)
          st4u(P.out + (size_t)r * P.HF + fid, make_float4(acc[0], acc[1], acc[2], acc[3]));
        else
          P.out[(size_t)r * P.HF + fid] = acc[0];
      }
    }
  }
}

}  // namespace dgnn

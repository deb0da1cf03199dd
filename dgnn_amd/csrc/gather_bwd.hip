// C-ABI entry points of the row-gather family, BACKWARD side (include/dgnn_hip.h): fused GAT
// backward (one-pass CSC gather, gat_bwd_fast.hpp; wave-per-column fallback, gat_generic.hpp), the
// per-row statistics pass and the fused GMMConv backward.  Forward side: gather.hip.
#include <cstdio>
#include <cstdlib>

#define DGNN_TU_BACKWARD 1
#include "gat_bwd_fast.hpp"
#include "gat_generic.hpp"
#include "launch.hpp"

using namespace dgnn;
using namespace dgnn::host;

namespace {

struct BwdLayout {
  bool fast;
  int L, T, slot_stride;
  size_t off_S, off_dE, off_long, off_stat, off_cols, off_slots, off_out, off_fwd, total;
  int max_long;
};

// Rows longer than this are summed by one block each (k_gat_row_sum4_long).  Tuning override:
// DGNN_ROWSUM_LONG (read once per process: the workspace layout depends on it).
inline int rowsum_long_threshold() {
  static const int v = [] {
    const int e = env_int("DGNN_ROWSUM_LONG", 256);
    return (e >= 32 && e <= (1 << 20)) ? e : 256;
  }();
  return v;
}

// lanes per head of the backward's lane map (gat_bwd_fast.hpp: bwd_lane): power of two covering
// ceil(f / vec) lanes
inline int bwd_lanes_per_head(int f, int vec) {
  const int need = (f + vec - 1) / vec;
  int g = 1;
  while (g < need) g <<= 1;
  return g;
}
inline int bwd_heads_padded(int h) { return h <= 1 ? 1 : h <= 2 ? 2 : h <= 4 ? 4 : 8; }
// Packed stream of the backward (gat_bwd_fast.hpp): rows whose lane map covers at most 32 lanes
// (h heads x fv lanes) -> lanes per row = the power of two covering them, at least 4; 0 = off
// (DGNN_BWD_PACK=0: A/B runs).  Only the STEADY (narrow, float4) instantiation has it.
inline int bwd_pack_lanes(bool steady, int h, int fv) {
  if (!steady || h * fv > env_int("DGNN_PACK_MAX_S", 16) || env_int("DGNN_BWD_PACK", 1) == 0) return 0;
  int s = 4;
  while (s < h * fv) s <<= 1;
  return s;
}
// floats per lane of the backward's lane map: float4 lanes for every width of at least four floats
// (rows need no 16-byte alignment: common.hpp ld4u), scalar lanes below
inline int bwd_vec(int f) { return f >= 4 ? 4 : 1; }
inline bool bwd_fast_shape(int m, int n, int h, int f) {
  if (force_generic() || h > 8 || m >= (1 << 30) || n >= (1 << 30)) return false;
  if (f > 256) return false;
  return bwd_lanes_per_head(f, bwd_vec(f)) <= DGNN_WAVE;   // one head fits one lane pass
}

inline BwdLayout bwd_layout(int m, int n, int nnz, int h, int f, bool need_recompute) {
  BwdLayout w = {};
  w.fast = bwd_fast_shape(m, n, h, f);
  size_t o = 0;
  w.off_S = o;
  o += align_up(sizeof(float) * (size_t)m * h, 256);
  w.off_dE = o;
  o += align_up(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1) * h, 256);
  w.max_long = nnz / rowsum_long_threshold() + 1;   // rows with more edges than the threshold
  w.off_long = o;                                // [0] = counter, [1..] = queue
  o += align_up(sizeof(int) * ((size_t)w.max_long + 1), 256);
  if (w.fast) {
    w.L = bwd_group_size(bwd_heads_padded(h), nnz, narrow_rows(h, f), h * ((f + 3) / 4));
    w.T = (nnz + w.L - 1) / w.L;
    w.slot_stride = (int)align_up((size_t)h * f + (size_t)h, 4);
    w.off_stat = o;
    o += align_up(sizeof(float4) * (size_t)m * h, 256);
    w.off_cols = o;
    o += align_up(sizeof(int) * (size_t)(w.T > 0 ? w.T : 1), 256);
    w.off_slots = o;
    o += align_up(sizeof(float) * 2 * (size_t)w.T * w.slot_stride, 256);
  }
  w.off_out = o;
  if (need_recompute) o += align_up(sizeof(float) * (size_t)m * h * f, 256);
  w.off_fwd = o;
  if (need_recompute) o += align_up(fwd_layout(nnz, h, f).total, 256);
  w.total = o;
  return w;
}

// Which gather stream a narrow (STEADY) launch takes = the PK template argument of k_bwd_col_gather:
// 0 one entry per step; 1 packed with run-time lane widths (also the GMM form with the kernels in
// registers); 2 / 4 / 8 / 16 packed with that many lanes per head as a compile-time constant (rows
// whose lane width is max(4, padded heads x lanes per head), groups of at most 128 entries).
inline int bwd_stream_kind(const BwdParams &P, int hc) {
  if (!P.steady || P.pack_s <= 0) return 0;
  const int want_s = hc * P.fv > 4 ? hc * P.fv : 4;
  const bool konst = P.pack_s == want_s && P.fv_const != 0 && !P.hreg && P.fv >= 2 && hc * P.fv <= 16 && P.L <= 128;
  return konst ? P.fv : 1;
}

template <int BMODE, int HC, int EPL>
void launch_bwd_steady(const BwdParams &P, int pk, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
#define DGNN_BWD_PK(V)                                                                                  \
  if constexpr (V <= 1 || (HC * V <= 16 && EPL <= 2)) {                                                 \
    if (pk == V) {                                                                                      \
      hipLaunchKernelGGL((k_bwd_col_gather<HC, EPL, 4, BMODE, true, V>), grid, block, lds, st, P);      \
      return;                                                                                           \
    }                                                                                                   \
  }
  DGNN_BWD_PK(0) DGNN_BWD_PK(1) DGNN_BWD_PK(2) DGNN_BWD_PK(4) DGNN_BWD_PK(8) DGNN_BWD_PK(16)
#undef DGNN_BWD_PK
}

// One-pass CSC gather kernel of the padded head count hc, group size P.L and lane width vec.
template <int BMODE>
void launch_bwd_gather(const BwdParams &P, int hc, int vec, dim3 grid, dim3 block, size_t lds,
                       hipStream_t st) {
  const int pk = (vec == 4) ? bwd_stream_kind(P, hc) : 0;
#define DGNN_BWD_CASE(HC, EPL)                                                                    \
  if (hc == HC && P.L == 64 * EPL) {                                                              \
    if (timing_on())                                                                              \
      std::snprintf(g_timer.name[1], sizeof(g_timer.name[1]), "k_bwd_col_gather<%d, %d, %d, %d, %s, %d>", HC, EPL, \
                    vec == 4 ? 4 : 1, BMODE, (vec == 4 && P.steady) ? "true" : "false", pk);      \
    if (vec == 4 && P.steady)                                                                     \
      launch_bwd_steady<BMODE, HC, EPL>(P, pk, grid, block, lds, st);                             \
    else if (vec == 4)                                                                            \
      hipLaunchKernelGGL((k_bwd_col_gather<HC, EPL, 4, BMODE, false>), grid, block, lds, st, P);  \
    else                                                                                          \
      hipLaunchKernelGGL((k_bwd_col_gather<HC, EPL, 1, BMODE, false>), grid, block, lds, st, P);  \
  }
  DGNN_BWD_CASE(1, 1) DGNN_BWD_CASE(1, 2) DGNN_BWD_CASE(1, 4)
  DGNN_BWD_CASE(2, 1) DGNN_BWD_CASE(2, 2) DGNN_BWD_CASE(2, 4)
  DGNN_BWD_CASE(4, 1) DGNN_BWD_CASE(4, 2) DGNN_BWD_CASE(4, 4)
  DGNN_BWD_CASE(8, 1) DGNN_BWD_CASE(8, 2)
#undef DGNN_BWD_CASE
}

}  // namespace

extern "C" {

#ifdef DGNN_PHASE_CLOCK
// diagnostic build only: per-phase wave-cycle sums of k_bwd_col_gather (common.hpp; entries [8..11])
int dgnn_debug_phase_cycles_bwd(unsigned long long *out16, int reset) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return (int)e;
  if (out16) {
    e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 16);
    if (e != hipSuccess) return (int)e;
  }
  if (reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z));
    if (e != hipSuccess) return (int)e;
  }
  return DGNN_OK;
}
#endif

int dgnn_gat_row_dot(int m, int h, int f, const float *grad, const float *out_feat,
                     float *row_dot, void *stream) {
  if (m < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if (m == 0) return DGNN_OK;
  if (!grad || !out_feat || !row_dot) return DGNN_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_gat_row_dot, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0,
                     (hipStream_t)stream, m, h, f, grad, out_feat, row_dot);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_gat_row_stats(int m, int h, int f, float negative_slope, const float *attn_row, const float *edge_max,
                       const float *edge_sum, const float *grad, const float *out_feat, const float *out_neg,
                       const float *a_neg, float *row_stat, float *grad_attn_row, void *stream) {
  if (m < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if (m == 0) return DGNN_OK;
  if (!attn_row || !edge_max || !edge_sum || !grad || !out_feat || !row_stat) return DGNN_ERR_BAD_ARG;
  if ((out_neg == nullptr) != (a_neg == nullptr) || (out_neg != nullptr && grad_attn_row == nullptr))
    return DGNN_ERR_BAD_ARG;
  const int vec = bwd_vec(f);
  if (h > 8 || bwd_lanes_per_head(f, vec) > DGNN_WAVE || !aligned16(row_stat)) return DGNN_ERR_UNSUPPORTED;
  if (out_neg != nullptr && vec != 4) return DGNN_ERR_UNSUPPORTED;
  BwdParams P = {};
  P.m = m; P.H = h; P.F = f; P.HF = h * f; P.vec = vec; P.fv = bwd_lanes_per_head(f, vec); P.hpp = DGNN_WAVE / P.fv;
  P.slope = negative_slope;
  P.attn_row = attn_row; P.edge_max = edge_max; P.edge_sum = edge_sum; P.grad = grad; P.out_feat = out_feat;
  P.out_neg = out_neg; P.a_neg = a_neg; P.grad_attn_row = grad_attn_row; P.rowstat = (float4 *)row_stat;
  hipStream_t st = (hipStream_t)stream;
  int rs = 1;
  while (rs < h * P.fv) rs <<= 1;
  if (vec == 4 && rs <= env_int("DGNN_ROWSTAT_PACKED_MAX", 32)) {
    const long long rows_per_wave = (long long)(DGNN_WAVE / rs) * 4;
    hipLaunchKernelGGL(k_bwd_rowstat_packed, dim3(blocks_for_waves((m + rows_per_wave - 1) / rows_per_wave)),
                       dim3(DGNN_BLOCK), 0, st, P, rs);
  } else if (vec == 4) {
    hipLaunchKernelGGL(k_bwd_rowstat<4>, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, P);
  } else {
    hipLaunchKernelGGL(k_bwd_rowstat<1>, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, P);
  }
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

size_t dgnn_gat_backward_workspace_bytes(int m, int n, int nnz, int h, int f, int have_row_term) {
  if (m < 0 || n < 0 || nnz < 0 || h <= 0 || f <= 0) return 0;
  return bwd_layout(m, n, nnz, h, f, !have_row_term).total;
}

}  // extern "C"

// Body of dgnn_gat_backward / dgnn_gat_backward_packed.  `packed` != NULL: the per-row inputs
// arrive as the float4 table the gather kernel reads ({attn_row, edge_max, edge_sum, row_dot} per
// (row, head)), so the packing pass is skipped; only the fast path takes it.
static int gat_backward_impl(int m, int n, int nnz, int h, int f, float negative_slope, float attn_drop,
                             const int *row_ptr, const int *col_ind, const int *col_ptr,
                             const int *row_ind, const int *permute, const float *edge_max,
                             const float *edge_sum, const float *edge_mask, const float *in_feat,
                             const float *attn_row, const float *attn_col, const float *grad,
                             const float *out_feat, const float *row_dot, const float4 *packed,
                             float *grad_feat, float *grad_attn_row, float *grad_attn_col,
                             void *workspace, size_t workspace_bytes, void *stream,
                             DropRng rng = DropRng{0, 0, 0, nullptr}, const float *out_neg = nullptr,
                             const float *a_neg = nullptr) {
  if ((out_neg == nullptr) != (a_neg == nullptr)) return DGNN_ERR_BAD_ARG;
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (n < 0) return DGNN_ERR_BAD_ARG;
  if (!row_ptr || !col_ptr) return DGNN_ERR_BAD_ARG;
  if (m > 0 && !packed && (!edge_max || !edge_sum || !attn_row)) return DGNN_ERR_BAD_ARG;
  if (m > 0 && !grad) return DGNN_ERR_BAD_ARG;
  // grad_attn_row == NULL: the caller computes it itself from the forward's negative-score pair
  // (the row-partitioned runner: a row's owner holds everything that formula needs); then no
  // edge gradients are produced at all.  Only the neighbour-grouped kernels can leave them out.
  const bool no_row_grad = (grad_attn_row == nullptr);
  if (n > 0 && (!in_feat || !attn_col || !grad_feat || !grad_attn_col)) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!col_ind || !row_ind || !permute)) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (attn_drop > 0.f && nnz > 0 && !edge_mask && !rng.on) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop > 0.f)) rng.on = 0;
  const bool have_row_term = (out_feat != nullptr) || (row_dot != nullptr) || (packed != nullptr) || m == 0;
  BwdLayout w = bwd_layout(m, n, nnz, h, f, !have_row_term);
  if (!workspace || workspace_bytes < w.total) return DGNN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const float *mask = (attn_drop > 0.f && !rng.on) ? edge_mask : nullptr;
  char *ws = (char *)workspace;
  float *S = (float *)(ws + w.off_S);
  float *dE = (float *)(ws + w.off_dE);
  const float *out = out_feat;
  if (!have_row_term && m > 0) {
    // Reference-signature path: the saved forward output was not handed over, rebuild it.
    float *tmp = (float *)(ws + w.off_out);
    rc = launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr,
                        col_ind, in_feat, mask, nullptr, nullptr, tmp, ws + w.off_fwd,
                        w.total - w.off_fwd, st, rng);
    if (rc) return rc;
    out = tmp;
  }
  const int vec = bwd_vec(f);
  // H = 4 moves attn_col as float4; the feature rows need no 16-byte alignment (common.hpp: ld4u)
  bool fast = w.fast && (h != 4 || aligned16(attn_col));
  if (packed != nullptr && !(fast && aligned16(packed))) return DGNN_ERR_UNSUPPORTED;
  // grad_attn_row from the forward's negative-score pair (row pass) instead of the [nnz,h] edge
  // gradients: the fast float4 kernels with the unpacked per-row inputs
  const bool neg = fast && vec == 4 && out_neg != nullptr && packed == nullptr && m > 0 && !no_row_grad;
  if (rng.on && !fast) return DGNN_ERR_UNSUPPORTED;   // the wave-per-column kernel reads a materialised mask
  if (no_row_grad && !fast && m > 0) return DGNN_ERR_UNSUPPORTED;
  if (fast) {
    BwdParams P = {};
    P.n = n; P.m = m; P.nnz = nnz; P.H = h; P.F = f; P.HF = h * f;
    P.L = w.L; P.T = w.T; P.vec = vec; P.fv = bwd_lanes_per_head(f, vec); P.slot_stride = w.slot_stride;
    P.hpp = DGNN_WAVE / P.fv;
    P.steady = (vec == 4 && h * ((f + 3) / 4) < DGNN_WAVE && env_int("DGNN_BWD_STEADY", 1) != 0) ? 1 : 0;
    P.pack_s = bwd_pack_lanes(P.steady != 0, h, P.fv);
    P.fv_const = env_int("DGNN_BWD_PACK_CONST", 1);
    const int hc = bwd_heads_padded(h);
    P.attn_drop = attn_drop; P.slope = negative_slope;
    P.col_ptr = col_ptr; P.row_ind = row_ind; P.permute = permute;
    P.attn_row = attn_row; P.attn_col = attn_col; P.edge_max = edge_max; P.edge_sum = edge_sum;
    P.edge_mask = mask; P.rng = rng; P.in_feat = in_feat; P.grad = grad; P.out_feat = out;
    P.S_in = row_dot;
    P.out_neg = neg ? out_neg : nullptr; P.a_neg = neg ? a_neg : nullptr; P.grad_attn_row = grad_attn_row;
    P.rowstat = packed ? const_cast<float4 *>(packed) : (float4 *)(ws + w.off_stat);
    P.dE = (neg || no_row_grad) ? nullptr : dE; P.grad_feat = grad_feat; P.grad_attn_col = grad_attn_col;
    P.task_col0 = (int *)(ws + w.off_cols);
    P.slots = (float *)(ws + w.off_slots);
    if (m > 0 && !packed) {
      int rs = 1;
      while (rs < h * P.fv) rs <<= 1;
      if (vec == 4 && rs <= env_int("DGNN_ROWSTAT_PACKED_MAX", 32) && env_int("DGNN_ROWSTAT_PACKED", 1) != 0) {
        const long long rows_per_wave = (long long)(DGNN_WAVE / rs) * 4;
        hipLaunchKernelGGL(k_bwd_rowstat_packed, dim3(blocks_for_waves((m + rows_per_wave - 1) / rows_per_wave)),
                           dim3(DGNN_BLOCK), 0, st, P, rs);
      } else if (vec == 4)
        hipLaunchKernelGGL(k_bwd_rowstat<4>, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, P);
      else
        hipLaunchKernelGGL(k_bwd_rowstat<1>, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, P);
      DGNN_LAUNCH_CHECK();
    }
    if (n > 0) {
      long long g = ((long long)n + 255) / 256;
      if (g > 8192) g = 8192;
      hipLaunchKernelGGL(k_bwd_prep_cols, dim3((unsigned)g), dim3(256), 0, st, P);
      DGNN_LAUNCH_CHECK();
    }
    if (P.T > 0) {
      const dim3 grid(blocks_for_waves(P.T)), block(DGNN_BLOCK);
      const size_t lds = bwd_lds_bytes(hc, P.L) + (size_t)env_int("DGNN_BWD_LDS_PAD", 0);
      timer_mark(1, 0, st);
      launch_bwd_gather<BWD_GAT>(P, hc, vec, grid, block, lds, st);
      timer_mark(1, 1, st);
      DGNN_LAUNCH_CHECK();
      if (P.T > 1) {
        if (vec == 4)
          { if (P.T - 1 > DGNN_FIX_C) hipLaunchKernelGGL(k_bwd_fixup<4>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 0);
          hipLaunchKernelGGL(k_bwd_fixup<4>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 1); }
        else
          { if (P.T - 1 > DGNN_FIX_C) hipLaunchKernelGGL(k_bwd_fixup<1>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 0);
          hipLaunchKernelGGL(k_bwd_fixup<1>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 1); }
        DGNN_LAUNCH_CHECK();
      }
    }
  } else {
    const float *Sp = row_dot;
    if (!Sp && m > 0) {
      hipLaunchKernelGGL(k_gat_row_dot, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, m, h, f,
                         grad, out, S);
      DGNN_LAUNCH_CHECK();
      Sp = S;
    }
    if (n > 0) {
      hipLaunchKernelGGL(k_gat_bwd_col, dim3(blocks_for_waves(n)), dim3(DGNN_BLOCK), 0, st, n, h, f,
                         negative_slope, attn_drop, col_ptr, row_ind, permute, edge_max, edge_sum,
                         mask, attn_row, attn_col, in_feat, grad, Sp, dE, grad_feat, grad_attn_col);
      DGNN_LAUNCH_CHECK();
    }
  }
  if (m > 0 && !neg && !no_row_grad) {
    int *long_count = (int *)(ws + w.off_long);
    hipError_t e = hipMemsetAsync(long_count, 0, sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    if (h == 4 && aligned16(dE) && aligned16(grad_attn_row)) {
      const int rows_per_block = 16 * DGNN_ROWSUM_R;
      hipLaunchKernelGGL(k_gat_row_sum4, dim3((m + rows_per_block - 1) / rows_per_block),
                         dim3(DGNN_BLOCK), 0, st, m, row_ptr, (const float4 *)dE,
                         (float4 *)grad_attn_row, long_count, long_count + 1, rowsum_long_threshold());
      DGNN_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_gat_row_sum4_long, dim3(w.max_long), dim3(DGNN_BLOCK), 0, st, row_ptr,
                         (const float4 *)dE, (float4 *)grad_attn_row, (const int *)long_count,
                         (const int *)(long_count + 1));
    } else if (h == 8 && aligned16(dE) && aligned16(grad_attn_row)) {
      hipLaunchKernelGGL(k_gat_row_sum8, dim3((m + 15) / 16), dim3(DGNN_BLOCK), 0, st, m, row_ptr,
                         (const float4 *)dE, (float4 *)grad_attn_row, long_count, long_count + 1,
                         rowsum_long_threshold());
      DGNN_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_gat_row_sum8_long, dim3(w.max_long), dim3(DGNN_BLOCK), 0, st, row_ptr,
                         (const float4 *)dE, (float4 *)grad_attn_row, (const int *)long_count,
                         (const int *)(long_count + 1));
    } else {
      hipLaunchKernelGGL(k_gat_row_sum, dim3((m + 15) / 16), dim3(DGNN_BLOCK), 0, st, m, h, row_ptr,
                         dE, grad_attn_row, long_count, long_count + 1, rowsum_long_threshold());
      DGNN_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_gat_row_sum_long, dim3(w.max_long), dim3(DGNN_BLOCK), 0, st, h, row_ptr,
                         (const float *)dE, grad_attn_row, (const int *)long_count,
                         (const int *)(long_count + 1));
    }
    DGNN_LAUNCH_CHECK();
  }
  return DGNN_OK;
}

extern "C" {

int dgnn_gat_backward(int m, int n, int nnz, int h, int f, float negative_slope, float attn_drop,
                      const int *row_ptr, const int *col_ind, const int *col_ptr,
                      const int *row_ind, const int *permute, const float *edge_max,
                      const float *edge_sum, const float *edge_mask, const float *in_feat,
                      const float *attn_row, const float *attn_col, const float *grad,
                      const float *out_feat, const float *row_dot, const float *out_neg,
                      const float *a_neg, float *grad_feat, float *grad_attn_row, float *grad_attn_col,
                      void *workspace, size_t workspace_bytes, void *stream) {
  return gat_backward_impl(m, n, nnz, h, f, negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind,
                           permute, edge_max, edge_sum, edge_mask, in_feat, attn_row, attn_col, grad, out_feat,
                           row_dot, nullptr, grad_feat, grad_attn_row, grad_attn_col, workspace,
                           workspace_bytes, stream, DropRng{0, 0, 0, nullptr}, out_neg, a_neg);
}

int dgnn_gat_backward_packed(int m, int n, int nnz, int h, int f, float negative_slope, float attn_drop,
                             const int *row_ptr, const int *col_ind, const int *col_ptr,
                             const int *row_ind, const int *permute, const float *row_stat,
                             const float *edge_mask, const float *in_feat, const float *attn_col,
                             const float *grad, float *grad_feat, float *grad_attn_row,
                             float *grad_attn_col, void *workspace, size_t workspace_bytes,
                             void *stream) {
  if (m > 0 && !row_stat) return DGNN_ERR_BAD_ARG;
  if (m == 0) row_stat = nullptr;
  return gat_backward_impl(m, n, nnz, h, f, negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind,
                           permute, nullptr, nullptr, edge_mask, in_feat, nullptr, attn_col, grad, nullptr,
                           nullptr, (const float4 *)row_stat, grad_feat, grad_attn_row, grad_attn_col,
                           workspace, workspace_bytes, stream);
}


int dgnn_gat_backward_seeded_dev(int m, int n, int nnz, int h, int f, float negative_slope,
                                 float attn_drop, const int *row_ptr, const int *col_ind,
                                 const int *col_ptr, const int *row_ind, const int *permute,
                                 const float *edge_max, const float *edge_sum, const uint64_t *rng_state,
                                 const float *in_feat, const float *attn_row, const float *attn_col,
                                 const float *grad, const float *out_feat, const float *row_dot,
                                 const float *out_neg, const float *a_neg,
                                 float *grad_feat, float *grad_attn_row, float *grad_attn_col,
                                 void *workspace, size_t workspace_bytes, void *stream) {
  if (attn_drop > 0.f && (!rng_state || ((uintptr_t)rng_state & 7u))) return DGNN_ERR_BAD_ARG;
  return gat_backward_impl(m, n, nnz, h, f, negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind,
                           permute, edge_max, edge_sum, nullptr, in_feat, attn_row, attn_col, grad, out_feat,
                           row_dot, nullptr, grad_feat, grad_attn_row, grad_attn_col, workspace,
                           workspace_bytes, stream,
                           DropRng{0, 0, 1, (const unsigned long long *)rng_state}, out_neg, a_neg);
}

int dgnn_gat_backward_saved(int m, int n, int nnz, int h, int f, float negative_slope, float attn_drop,
                            const int *row_ptr, const int *col_ind, const int *col_ptr,
                            const int *row_ind, const int *permute, const float *edge_max,
                            const float *edge_sum, const float *saved, const float *in_feat,
                            const float *attn_row, const float *attn_col, const float *grad,
                            float *grad_feat, float *grad_attn_row, float *grad_attn_col, void *workspace,
                            size_t workspace_bytes, void *stream) {
  if (m > 0 && (!saved || !aligned16(saved))) return DGNN_ERR_BAD_ARG;
  if (m <= 0 || h <= 0 || f <= 0)
    return gat_backward_impl(m, n, nnz, h, f, negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute,
                             edge_max, edge_sum, nullptr, in_feat, attn_row, attn_col, grad, nullptr, nullptr, nullptr,
                             grad_feat, grad_attn_row, grad_attn_col, workspace, workspace_bytes, stream);
  const SavedLayout w = saved_layout(m, h, f);
  // the dropout pair the forward wrote in front of the saved rows is read by the kernels themselves
  return gat_backward_impl(m, n, nnz, h, f, negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute,
                           edge_max, edge_sum, nullptr, in_feat, attn_row, attn_col, grad, saved + w.off_out, nullptr,
                           nullptr, grad_feat, grad_attn_row, grad_attn_col, workspace, workspace_bytes, stream,
                           DropRng{0, 0, attn_drop > 0.f ? 1 : 0, (const unsigned long long *)saved}, saved + w.off_neg,
                           saved + w.off_aneg);
}

int dgnn_gat_backward_seeded(int m, int n, int nnz, int h, int f, float negative_slope,
                             float attn_drop, const int *row_ptr, const int *col_ind,
                             const int *col_ptr, const int *row_ind, const int *permute,
                             const float *edge_max, const float *edge_sum, uint64_t seed,
                             uint64_t offset, const float *in_feat, const float *attn_row,
                             const float *attn_col, const float *grad, const float *out_feat,
                             const float *row_dot, const float *out_neg, const float *a_neg,
                             float *grad_feat, float *grad_attn_row, float *grad_attn_col, void *workspace,
                             size_t workspace_bytes, void *stream) {
  return gat_backward_impl(m, n, nnz, h, f, negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind,
                           permute, edge_max, edge_sum, nullptr, in_feat, attn_row, attn_col, grad, out_feat,
                           row_dot, nullptr, grad_feat, grad_attn_row, grad_attn_col, workspace,
                           workspace_bytes, stream, DropRng{seed, offset, 1, nullptr}, out_neg, a_neg);
}


// ----------------------------------------------------------------- fused GMMConv backward
// The one-pass GMM backward pays off when all kernels fit ONE lane pass (K * lanes-per-kernel
// <= 64: 3 x 64, 2 x 100 ...); with several passes (3 x 41 scalar lanes) the SDDMM + CSC-gather
// sequence, whose kernels hold several slices per lane, is faster (measured on the Reddit shape).
static bool gmm_backward_fused_shape(int nodes, int kernels, int embed) {
  if (!bwd_fast_shape(nodes, nodes, kernels, embed)) return false;
  return kernels * bwd_lanes_per_head(embed, bwd_vec(embed)) <= DGNN_WAVE;
}

// The summed-gradient form with the kernels in registers (k_bwd_col_gather, run_hreg): float4 rows of
// 17 .. 64 features; its groups hold at most 128 entries (the longer stream has to fit in LDS).
// Measured, K = 3, forward + backward in ms without / with it (profiles/r2i_gmm_hreg_ab.txt): Reddit shape
// F = 64 24.3 / 20.1, F = 32 19.1 / 13.6, F = 16 11.3 / 11.1; products shape 16.3 / 14.7, 12.5 / 10.4,
// 8.1 / 8.8 -- at 16 features the K extra U steps of every short column cost more than the packing
// saves, and the pass is bound by the edge-order gathers of pseudo / scatters of grad_gauss anyway.
static bool gmm_hreg_shape(int kernels, int embed) {
  return embed % 4 == 0 && embed > 16 && embed <= 64 && kernels * embed / 4 < DGNN_WAVE && env_int("DGNN_GMM_HREG", 1) != 0 &&
         env_int("DGNN_BWD_STEADY", 1) != 0 && env_int("DGNN_BWD_PACK", 1) != 0;
}

size_t dgnn_gmm_backward_workspace_bytes(int nodes, int nnz, int kernels, int embed) {
  if (nodes < 0 || nnz < 0 || kernels <= 0 || embed <= 0) return 0;
  if (!gmm_backward_fused_shape(nodes, kernels, embed)) return 0;
  int L = bwd_group_size(bwd_heads_padded(kernels), nnz, narrow_rows(kernels, embed));
  if (gmm_hreg_shape(kernels, embed) && L > 128) L = 128;      // either form of the gradient fits
  const size_t T = ((size_t)nnz + L - 1) / L;
  const size_t stride = align_up((size_t)kernels * embed + kernels, 4);
  return align_up(sizeof(int) * (T > 0 ? T : 1), 256) + align_up(sizeof(float) * 2 * T * stride, 256);
}

static int gmm_backward_impl(int nodes, int nnz, int kernels, int dim, int embed, const int *col_ptr,
                             const int *row_ind, const int *permute, const float *node_feat,
                             const float *pseudo, const float *mu, const float *inv_sigma,
                             const float *grad, int grad_sum, float *grad_node_feat, float *grad_gauss,
                             void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_dims(nodes, nnz, kernels, embed);
  if (rc) return rc;
  if (dim <= 0) return DGNN_ERR_BAD_ARG;
  if (nodes == 0) return DGNN_OK;
  if (!col_ptr || !mu || !inv_sigma || !grad_node_feat) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!row_ind || !permute || !node_feat || !pseudo || !grad || !grad_gauss))
    return DGNN_ERR_BAD_ARG;
  if (!gmm_backward_fused_shape(nodes, kernels, embed)) return DGNN_ERR_UNSUPPORTED;
  const int vec = bwd_vec(embed);
  const size_t need = dgnn_gmm_backward_workspace_bytes(nodes, nnz, kernels, embed);
  if (!workspace || workspace_bytes < need) return DGNN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  BwdParams P = {};
  P.n = nodes; P.m = nodes; P.nnz = nnz; P.H = kernels; P.F = embed; P.HF = kernels * embed;
  const int hc = bwd_heads_padded(kernels);
  P.L = bwd_group_size(hc, nnz, narrow_rows(kernels, embed));
  P.vec = vec; P.fv = bwd_lanes_per_head(embed, vec); P.hpp = DGNN_WAVE / P.fv;
  P.steady = (vec == 4 && kernels * ((embed + 3) / 4) < DGNN_WAVE && env_int("DGNN_BWD_STEADY", 1) != 0) ? 1 : 0;
  P.pack_s = bwd_pack_lanes(P.steady != 0, kernels, P.fv);
  P.hreg = (grad_sum && vec == 4 && P.steady && gmm_hreg_shape(kernels, embed)) ? 1 : 0;
  if (P.hreg) {
    P.pack_s = P.fv > 4 ? P.fv : 4;              // lanes per entry: the row's float4 slots, a power of two
    if (P.L > 128) P.L = 128;
  }
  P.T = (nnz + P.L - 1) / P.L;
  P.fv_const = env_int("DGNN_BWD_PACK_CONST", 1);
  P.slot_stride = (int)align_up((size_t)kernels * embed + kernels, 4);
  P.col_ptr = col_ptr; P.row_ind = row_ind; P.permute = permute;
  P.in_feat = node_feat; P.grad = grad; P.grad_feat = grad_node_feat; P.grad_attn_col = nullptr;
  P.dE = grad_gauss;
  P.pseudo = pseudo; P.mu = mu; P.inv_sigma = inv_sigma; P.dim = dim;
  P.grad_sum = grad_sum ? 1 : 0;
  P.task_col0 = (int *)workspace;
  P.slots = (float *)((char *)workspace + align_up(sizeof(int) * (size_t)(P.T > 0 ? P.T : 1), 256));
  {
    long long g = ((long long)nodes + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_bwd_prep_cols, dim3((unsigned)g), dim3(256), 0, st, P);
    DGNN_LAUNCH_CHECK();
  }
  if (P.T > 0) {
    const dim3 grid(blocks_for_waves(P.T)), block(DGNN_BLOCK);
    timer_mark(1, 0, st);
    launch_bwd_gather<BWD_GMM>(P, hc, vec, grid, block, bwd_lds_bytes(hc, P.L, P.hreg != 0), st);
    timer_mark(1, 1, st);
    DGNN_LAUNCH_CHECK();
    if (P.T > 1) {
      if (vec == 4)
        { if (P.T - 1 > DGNN_FIX_C) hipLaunchKernelGGL(k_bwd_fixup<4>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 0);
          hipLaunchKernelGGL(k_bwd_fixup<4>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 1); }
      else
        { if (P.T - 1 > DGNN_FIX_C) hipLaunchKernelGGL(k_bwd_fixup<1>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 0);
          hipLaunchKernelGGL(k_bwd_fixup<1>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 1); }
      DGNN_LAUNCH_CHECK();
    }
  }
  return DGNN_OK;
}

int dgnn_gmm_backward(int nodes, int nnz, int kernels, int dim, int embed, const int *col_ptr,
                      const int *row_ind, const int *permute, const float *node_feat,
                      const float *pseudo, const float *mu, const float *inv_sigma,
                      const float *grad, float *grad_node_feat, float *grad_gauss,
                      void *workspace, size_t workspace_bytes, void *stream) {
  return gmm_backward_impl(nodes, nnz, kernels, dim, embed, col_ptr, row_ind, permute, node_feat, pseudo, mu,
                           inv_sigma, grad, 0, grad_node_feat, grad_gauss, workspace, workspace_bytes, stream);
}

int dgnn_gmm_backward_sum(int nodes, int nnz, int kernels, int dim, int embed, const int *col_ptr,
                          const int *row_ind, const int *permute, const float *node_feat,
                          const float *pseudo, const float *mu, const float *inv_sigma,
                          const float *grad_sum, float *grad_node_feat, float *grad_gauss,
                          void *workspace, size_t workspace_bytes, void *stream) {
  return gmm_backward_impl(nodes, nnz, kernels, dim, embed, col_ptr, row_ind, permute, node_feat, pseudo, mu,
                           inv_sigma, grad_sum, 1, grad_node_feat, grad_gauss, workspace, workspace_bytes, stream);
}


}  // extern "C"

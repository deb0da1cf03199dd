// Host-side helpers shared by the translation units of the row-gather family (gather.hip: forward
// side; gather_bwd.hip: backward side): alignment tests, the optional kernel timer, the forward's
// workspace layout and the forward launcher the backward calls when it has to rebuild out_feat.
#pragma once
#include <cstdio>
#include <cstdlib>

#include "gather_fast.hpp"

namespace dgnn {
namespace host {


inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }
inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// DGNN_FORCE_GENERIC=1 routes everything through the wave-per-row kernels (A/B testing).
inline bool force_generic() {
  const char *e = std::getenv("DGNN_FORCE_GENERIC");
  return e != nullptr && e[0] == '1';
}

// ----------------------------------------------------------------------------- forward side
struct FwdLayout {
  int L, T, slot_stride, slot_neg;
  size_t off_rows, off_slots, total;
};

// neg: partial slots also hold the negative-score aggregate and weight sum (GatherParams::out_neg)
// rule: which narrow shapes take the longer groups of the round-3 sweep (fast_group_size): 1 = the fused GAT
// forward, all of them; 2 = SpMM, rows of 12 or more float4 slots (1 x 64 -6 %, 2 x 32 -10 %, 4 x 16 -7 %, 2 x 64
// -4 %; 1 x 16 measured 4 % slower and keeps 256); 0 = GMM, none (Reddit shape: 2.5 % slower with them)
inline FwdLayout fwd_layout(int nnz, int h, int f, bool neg = false, int rule = 1) {
  FwdLayout w;
  const int slots = h * ((f + 3) / 4);
  w.L = fast_group_size(h, nnz, narrow_rows(h, f), (rule == 1 || (rule == 2 && slots >= 12)) ? slots : 0);
  w.T = (nnz + w.L - 1) / w.L;
  w.slot_neg = (int)align_up((size_t)h * f + 3 * (size_t)h, 4);
  w.slot_stride = neg ? (int)align_up((size_t)w.slot_neg + (size_t)h * f, 4)
                      : (int)align_up((size_t)h * f + 2 * (size_t)h, 4);
  size_t o = 0;
  w.off_rows = o;
  o += align_up(sizeof(int) * (size_t)(w.T > 0 ? w.T : 1), 256);
  w.off_slots = o;
  o += align_up(sizeof(float) * 2 * (size_t)w.T * w.slot_stride, 256);
  w.total = o;
  return w;
}

inline bool fwd_fast_ok(int m, int h) { return h <= DGNN_FAST_HMAX && m < (1 << 30) && !force_generic(); }

inline int check_dims(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if ((long long)h * f > (1 << 20)) return DGNN_ERR_UNSUPPORTED;
  return DGNN_OK;
}


// gather.hip: the fused GAT forward on the fast or the generic kernels (the backward uses it to
// rebuild out_feat for callers with the reference signature)
int launch_forward(int m, int nnz, int h, int f, float attn_drop, float slope,
                   const float *attn_row, const float *attn_col, const int *row_ptr,
                   const int *col_ind, const float *in_feat, const float *edge_mask,
                   float *edge_max, float *edge_sum, float *out_feat, void *workspace,
                   size_t workspace_bytes, hipStream_t st, DropRng rng = DropRng{0, 0, 0, nullptr},
                   float *out_neg = nullptr, float *a_neg = nullptr, float *out2 = nullptr);

// Layout (in floats) of the buffer the reference-signature path carries from forward to backward in
// the slot of the reference's edge_mask (include/dgnn_hip.h: dgnn_gat_forward_saved)
struct SavedLayout {
  size_t off_out, off_neg, off_aneg, total;
};
inline SavedLayout saved_layout(int m, int h, int f) {
  SavedLayout w;
  const size_t mhf = align_up((size_t)m * h * f, 4), mh = align_up((size_t)m * h, 4);
  w.off_out = 4;                      // [0, 4): the dropout {seed, offset} pair, two uint64
  w.off_neg = w.off_out + mhf;
  w.off_aneg = w.off_neg + mhf;
  w.total = w.off_aneg + mh;
  return w;
}

}  // namespace host
}  // namespace dgnn

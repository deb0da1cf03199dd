// C-ABI entry points of the fused GAT path (include/dgnn_hip.h) and kernel selection.
#include "gat_generic.hpp"

using namespace dgnn;

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct BwdWorkspace {
  size_t off_S, off_dE, off_out, off_fwd, total;
};

BwdWorkspace bwd_layout(int m, int nnz, int h, int f, int have_out) {
  BwdWorkspace w;
  size_t o = 0;
  w.off_S = o;
  o += align_up(sizeof(float) * (size_t)m * h, 256);
  w.off_dE = o;
  o += align_up(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1) * h, 256);
  w.off_out = o;
  if (!have_out) o += align_up(sizeof(float) * (size_t)m * h * f, 256);
  w.off_fwd = o;
  if (!have_out) o += align_up(dgnn_gat_forward_workspace_bytes(m, nnz, h, f), 256);
  w.total = o;
  return w;
}

int check_dims(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  return DGNN_OK;
}

int launch_forward(int m, int nnz, int h, int f, float attn_drop, float slope,
                   const float *attn_row, const float *attn_col, const int *row_ptr,
                   const int *col_ind, const float *in_feat, const float *edge_mask,
                   float *edge_max, float *edge_sum, float *out_feat, void *workspace,
                   size_t workspace_bytes, hipStream_t st) {
  (void)workspace;
  (void)workspace_bytes;
  (void)nnz;
  if (m == 0) return DGNN_OK;
  hipLaunchKernelGGL(k_gat_fwd_row, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, m, h, f,
                     attn_drop, slope, attn_row, attn_col, row_ptr, col_ind, in_feat, edge_mask,
                     edge_max, edge_sum, out_feat);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

}  // namespace

extern "C" {

size_t dgnn_gat_forward_workspace_bytes(int m, int nnz, int h, int f) {
  (void)m;
  (void)nnz;
  (void)h;
  (void)f;
  return 0;
}

int dgnn_gat_forward(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                     const float *attn_row, const float *attn_col, const int *row_ptr,
                     const int *col_ind, const float *in_feat, const float *edge_mask,
                     float *edge_max, float *edge_sum, float *out_feat, void *workspace,
                     size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (!attn_row || !attn_col || !row_ptr || !in_feat || !out_feat || !edge_max || !edge_sum)
    return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && !col_ind) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (attn_drop > 0.f && nnz > 0 && !edge_mask) return DGNN_ERR_BAD_ARG;
  if (workspace_bytes < dgnn_gat_forward_workspace_bytes(m, nnz, h, f)) return DGNN_ERR_WORKSPACE;
  return launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr,
                        col_ind, in_feat, attn_drop > 0.f ? edge_mask : nullptr, edge_max, edge_sum,
                        out_feat, workspace, workspace_bytes, (hipStream_t)stream);
}

int dgnn_gat_inference(int m, int nnz, int h, int f, float negative_slope, const float *attn_row,
                       const float *attn_col, const int *row_ptr, const int *col_ind,
                       const float *in_feat, float *out_feat, void *workspace,
                       size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (!attn_row || !attn_col || !row_ptr || !in_feat || !out_feat) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && !col_ind) return DGNN_ERR_BAD_ARG;
  if (workspace_bytes < dgnn_gat_forward_workspace_bytes(m, nnz, h, f)) return DGNN_ERR_WORKSPACE;
  return launch_forward(m, nnz, h, f, 0.f, negative_slope, attn_row, attn_col, row_ptr, col_ind,
                        in_feat, nullptr, nullptr, nullptr, out_feat, workspace, workspace_bytes,
                        (hipStream_t)stream);
}

int dgnn_gat_forward_tiles(int m, int nnz, int h, int f, float negative_slope,
                           const float *attn_row, const float *attn_col, const int *row_ptr,
                           const int *col_ind, const float *in_feat, const int *tile_scheduler,
                           int n_tiles, float *edge_max, float *edge_sum, float *out_feat,
                           void *workspace, size_t workspace_bytes, void *stream) {
  if (n_tiles < 0 || (n_tiles > 0 && !tile_scheduler)) return DGNN_ERR_BAD_ARG;
  return dgnn_gat_forward(m, nnz, h, f, 0.f, negative_slope, attn_row, attn_col, row_ptr, col_ind,
                          in_feat, nullptr, edge_max, edge_sum, out_feat, workspace,
                          workspace_bytes, stream);
}

size_t dgnn_gat_backward_workspace_bytes(int m, int nnz, int h, int f, int have_out_feat) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return 0;
  return bwd_layout(m, nnz, h, f, have_out_feat).total;
}

int dgnn_gat_backward(int m, int nnz, int h, int f, float negative_slope, float attn_drop,
                      const int *row_ptr, const int *col_ind, const int *col_ptr,
                      const int *row_ind, const int *permute, const float *edge_max,
                      const float *edge_sum, const float *edge_mask, const float *in_feat,
                      const float *attn_row, const float *attn_col, const float *grad,
                      const float *out_feat, float *grad_feat, float *grad_attn_row,
                      float *grad_attn_col, void *workspace, size_t workspace_bytes,
                      void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (!row_ptr || !col_ptr || !edge_max || !edge_sum || !in_feat || !attn_row || !attn_col ||
      !grad || !grad_feat || !grad_attn_row || !grad_attn_col)
    return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!col_ind || !row_ind || !permute)) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (attn_drop > 0.f && nnz > 0 && !edge_mask) return DGNN_ERR_BAD_ARG;
  const BwdWorkspace w = bwd_layout(m, nnz, h, f, out_feat != nullptr);
  if (!workspace || workspace_bytes < w.total) return DGNN_ERR_WORKSPACE;
  if (m == 0) return DGNN_OK;
  hipStream_t st = (hipStream_t)stream;
  const float *mask = attn_drop > 0.f ? edge_mask : nullptr;
  char *ws = (char *)workspace;
  float *S = (float *)(ws + w.off_S);
  float *dE = (float *)(ws + w.off_dE);
  const float *out = out_feat;
  if (!out) {
    // Reference-signature path: the saved forward output was not handed over, rebuild it.
    float *tmp = (float *)(ws + w.off_out);
    rc = launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr,
                        col_ind, in_feat, mask, nullptr, nullptr, tmp, ws + w.off_fwd,
                        w.total - w.off_fwd, st);
    if (rc) return rc;
    out = tmp;
  }
  hipLaunchKernelGGL(k_gat_row_dot, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, m, h, f,
                     grad, out, S);
  DGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gat_bwd_col, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, m, h, f,
                     negative_slope, attn_drop, col_ptr, row_ind, permute, edge_max, edge_sum, mask,
                     attn_row, attn_col, in_feat, grad, S, dE, grad_feat, grad_attn_col);
  DGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gat_row_sum, dim3((m + 15) / 16), dim3(DGNN_BLOCK), 0, st, m, h, row_ptr,
                     dE, grad_attn_row);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

}  // extern "C"

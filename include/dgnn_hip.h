/*
 * dgnn_hip.h -- C ABI of libdgnn_hip.so: dgNN's fused GNN operators as hand-written
 * HIP kernels for AMD Instinct MI355X (gfx950, wave64).
 *
 * This is the drop-in boundary.  Each entry point replaces one host launcher of the
 * reference (dgSPARSE/dgNN, CUDA), cited per function as `file:line` relative to the
 * reference tree.  Plain pointers and sizes only: no torch types, no C++ types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless stated;
 *   - float tensors are fp32, index tensors int32, all contiguous row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *   - nothing here allocates or frees device memory: scratch comes in through the
 *     `workspace` argument, sized by the matching *_workspace_bytes() query
 *     (workspace may be NULL when the query returns 0);
 *   - return value: 0 = success, < 0 = argument error (DGNN_ERR_*), > 0 = hipError_t
 *     reported by a launch.  Functions never throw and never exit().
 *   - calls are re-entrant; there is no global mutable state.
 *
 * Index convention (reference: dgNN/script/train/train_gatconv.py:78-94): CSR rows are
 * DESTINATION nodes, col_ind holds SOURCE nodes ascending within a row; (col_ptr,
 * row_ind) is the CSC of the same matrix; permute[i] = CSR position of CSC entry i.
 */
#ifndef DGNN_HIP_H
#define DGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGNN_OK 0
#define DGNN_ERR_BAD_ARG (-1)       /* negative size, NULL where a tensor is required     */
#define DGNN_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels implement           */
#define DGNN_ERR_WORKSPACE (-3)     /* workspace NULL or smaller than *_workspace_bytes() */
#define DGNN_ERR_ALIGNMENT (-4)     /* a pointer is not 4-byte aligned                    */
#define DGNN_ERR_TOO_LARGE (-5)     /* beyond the static scratch of dgnn_gat_forward_noalloc */

int dgnn_abi_version(void);
const char *dgnn_error_string(int code);

/* Measurement aid, no counterpart in the reference.  With DGNN_TIMING=1 in the environment
 * the library records HIP events on the launch stream right around its two dominant kernels
 * (which = 0: forward row-gather kernel of dgnn_gat_forward / dgnn_gat_inference; which = 1:
 * backward column-gather kernel of dgnn_gat_backward).  This returns the duration of the most
 * recent such launch in milliseconds (it waits for that launch).  DGNN_ERR_UNSUPPORTED when
 * timing is off or nothing was launched yet. */
int dgnn_debug_last_kernel_ms(int which, float *ms);
/* The template instantiation of that launch as a profiler prints it, e.g.
 * "k_bwd_col_gather<4, 4, 4, 0, false>" (so a counter profile of another instantiation is not
 * mistaken for this kernel's).  NULL when timing is off or nothing was launched yet. */
const char *dgnn_debug_last_kernel_name(int which);
/* How many times that kernel has been launched since the library was loaded (-1 when timing is
 * off): lets a test assert that a backward did NOT rebuild the forward. */
long long dgnn_debug_kernel_launches(int which);

/* ------------------------------------------------------------------ fused GAT ---- */

/* Scratch for dgnn_gat_forward / dgnn_gat_inference (neighbour-group partials). */
size_t dgnn_gat_forward_workspace_bytes(int m, int nnz, int h, int f);

/* Fused GAT forward (training): scores -> LeakyReLU -> edge softmax -> (dropout) -> SpMM.
 * Replaces gat_forward(), dgNN/src/fused_gatconv/fused_gatconv_kernel.cu:322-361 and its
 * kernel fused_forward_kernel :24-118.
 *   attn_row [m,h]  destination-side score term      attn_col [n,h] source-side term
 *   in_feat  [n,h,f]                                  out_feat [m,h,f]
 *   edge_max/edge_sum [m,h]  row max / sum(exp(e-max)) kept for backward
 *   edge_mask [nnz,h] uniform floats, edge kept iff mask > attn_drop (:95); survivors are
 *             scaled by 1/(1-attn_drop) (:104).  May be NULL iff attn_drop == 0 (the
 *             reference always draws a mask with cuRAND, :333-341; with attn_drop == 0 it
 *             keeps every edge, so no mask is needed and none is read).
 * Empty row: edge_max = -1e38f, edge_sum = 0, out_feat = 0 (:45,:66,:90).
 *   out_neg [m,h,f], a_neg [m,h]  OPTIONAL pair (both or neither; no counterpart in the reference):
 *             the same aggregate and the sum of the (unmasked) softmax weights restricted to the
 *             edges whose LeakyReLU score is negative.  Handed to dgnn_gat_backward they replace
 *             its [nnz,h] edge-gradient scatter and per-row sums: with sl_j = (e_j < 0 ? slope : 1),
 *               grad_attn_row = sum_j a_j (g_j - S) sl_j = (1 - slope) (S * a_neg - <grad, out_neg>)
 *             (fused_backward_kernel :952-967 restated per row).  float4 rows (f % 4 == 0, 16-byte
 *             aligned) on the neighbour-grouped kernels; otherwise DGNN_ERR_UNSUPPORTED: call again
 *             with NULLs. */
int dgnn_gat_forward(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                     const float *attn_row, const float *attn_col, const int *row_ptr,
                     const int *col_ind, const float *in_feat, const float *edge_mask,
                     float *edge_max, float *edge_sum, float *out_feat, float *out_neg, float *a_neg,
                     void *workspace, size_t workspace_bytes, void *stream);

/* dgnn_gat_forward WITHOUT a workspace argument and without any allocation.  No counterpart in
 * the reference (its launcher allocates four tensors per call, fused_gatconv_kernel.cu:371-381);
 * it exists for callers that cannot allocate or size a workspace: code under HIP-graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  The neighbour-group partials live in a
 * static device array of DGNN_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds until
 * the process ends; its address is looked up once per device on the first call there.  Same
 * arguments, kernels and bits as dgnn_gat_forward.  Limit: dgnn_gat_forward_workspace_bytes(m,
 * nnz, h, f) <= DGNN_NOALLOC_SCRATCH_BYTES (about 15 M edges at 4 x 64), else DGNN_ERR_TOO_LARGE and
 * nothing is launched.  All calls on one GPU share that GPU's array: calls that overlap on one
 * GPU (two streams, two threads) race; serialise them.  Calls no hipMalloc / hipFree.
 * The FIRST call on a device looks the array's address up (hipGetSymbolAddress, which may load the
 * code object lazily): make that one call eagerly, outside any HIP-graph capture (any size, e.g.
 * the usual warm-up step); every later call on that device only launches kernels and captures. */
#define DGNN_NOALLOC_SCRATCH_BYTES (64u << 20)
int dgnn_gat_forward_noalloc(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                             const float *attn_row, const float *attn_col, const int *row_ptr,
                             const int *col_ind, const float *in_feat, const float *edge_mask,
                             float *edge_max, float *edge_sum, float *out_feat, void *stream);

/* Fused GAT inference: forward without mask and without saved statistics.
 * Replaces gat_inference(), fused_gatconv_kernel.cu:671-691 (kernels :391-478, :565-669). */
int dgnn_gat_inference(int m, int nnz, int h, int f, float negative_slope,
                       const float *attn_row, const float *attn_col, const int *row_ptr,
                       const int *col_ind, const float *in_feat, float *out_feat,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Tile-scheduled forward.  Replaces gat_forward_tb(), fused_gatconv_kernel.cu:1239-1272
 * (kernel fused_forward_kernel_tb_sr :1153-1237, whose inter-block hand-off through
 * atomics + __threadfence is not a valid grid sync).  `tile_scheduler` [n_tiles,2] is the
 * reference's (row, tile_in_row) list (script/test/test_gat_tb.py:16-26) for tiles of `tile_nnz`
 * edges (32 in the reference): every NON-EMPTY row, rows ascending, tile_in_row = 0 ..
 * ceil(degree / tile_nnz) - 1.  The list is checked on the device against row_ptr BEFORE anything
 * is computed (rows in range and in order, tiles consecutive, no non-empty row missing, no tile
 * starting past its row's end, the last listed tile of a row reaching that end); a malformed
 * list returns DGNN_ERR_BAD_ARG and leaves the outputs untouched (the reference would silently
 * skip or double-count those edges).  To hand that verdict back as a return code this entry
 * point -- alone in the library -- waits for the stream, so it cannot be captured into a HIP
 * graph.  The kernels balance rows with fixed-size neighbour groups whatever the tile size, so
 * for a well-formed list the result equals dgnn_gat_forward with attn_drop = 0. */
size_t dgnn_gat_forward_tiles_workspace_bytes(int m, int nnz, int h, int f);
int dgnn_gat_forward_tiles(int m, int nnz, int h, int f, float negative_slope,
                           const float *attn_row, const float *attn_col, const int *row_ptr,
                           const int *col_ind, const float *in_feat,
                           const int *tile_scheduler, int n_tiles, int tile_nnz,
                           float *edge_max, float *edge_sum, float *out_feat,
                           void *workspace, size_t workspace_bytes, void *stream);

/* S[r,h] = <grad[r,h,:], out_feat[r,h,:]>: the per-row term sum_j a_j g_j of
 * fused_backward_kernel's first sweep (fused_gatconv_kernel.cu:906-931) as one dense dot
 * product.  dgnn_gat_backward computes it itself when given out_feat; it is exported for
 * callers that hold rows and columns in different places (row-partitioned multi-GPU). */
int dgnn_gat_row_dot(int m, int h, int f, const float *grad, const float *out_feat,
                     float *row_dot, void *stream);

/* The whole per-row pass of the fused backward in one launch, for the same callers: row_stat
 * [m,h,4] = {attn_row, edge_max, edge_sum, S} per (row, head) -- the 16-byte records
 * dgnn_gat_backward_packed reads -- and, given the forward's negative-score pair, grad_attn_row
 * [m,h] = (1 - slope) (S a_neg - <grad, out_neg>) (fused_backward_kernel :952-967 restated per
 * row, see dgnn_gat_forward).  One read of grad, out_feat and out_neg.  out_neg / a_neg /
 * grad_attn_row may be NULL together.  DGNN_ERR_UNSUPPORTED outside h <= 8, f <= 256 (64 when f is
 * not a multiple of four; the pair needs float4 rows): use dgnn_gat_row_dot then. */
int dgnn_gat_row_stats(int m, int h, int f, float negative_slope, const float *attn_row,
                       const float *edge_max, const float *edge_sum, const float *grad,
                       const float *out_feat, const float *out_neg, const float *a_neg,
                       float *row_stat, float *grad_attn_row, void *stream);

/* Scratch for dgnn_gat_backward.  have_row_term != 0 when the caller will pass out_feat or
 * row_dot (then no recompute buffer is needed). */
size_t dgnn_gat_backward_workspace_bytes(int m, int n, int nnz, int h, int f, int have_row_term);

/* Fused GAT backward.  Replaces gat_backward(), fused_gatconv_kernel.cu:998-1038, i.e. the
 * three kernels mhspmm_backward_kernel :712-763, mhsddmm :814-890 and
 * fused_backward_kernel :892-968, and the [nnz,h] grad_edge_csr temporary (:1054).
 * Arguments follow the reference binding gat_backward (fused_gatconv.cpp:104-166).
 *   m rows (destinations) x n columns (sources).  The reference assumes m == n (:1013 launches
 *   m blocks for the columns); rectangular blocks are what a row/column partition produces.
 *   row_ptr [m+1], col_ind [nnz]; col_ptr [n+1], row_ind [nnz], permute [nnz];
 *   edge_max, edge_sum, attn_row [m,h]; attn_col [n,h]; in_feat [n,h,f]; grad [m,h,f].
 *   out_feat [m,h,f]  OPTIONAL saved forward output.  With it the row term
 *                     S[r,h] = sum_j a_j g_j of :906-931 is computed as <grad[r,h,:],
 *                     out_feat[r,h,:]> (identical algebraically) and the whole backward
 *                     needs ONE gather pass over the edges.
 *   row_dot [m,h]     OPTIONAL precomputed S (dgnn_gat_row_dot); takes precedence over out_feat.
 *                     Both NULL = rebuild out_feat with one extra forward pass (the reference
 *                     signature carries neither).
 *   out_neg [m,h,f], a_neg [m,h]  OPTIONAL pair saved by the forward (see dgnn_gat_forward): with
 *                     it grad_attn_row comes out of the per-row pass and the column pass writes no
 *                     edge gradients (no [nnz,h] scatter through permute, no per-row sums).  Used
 *                     on the float4 neighbour-grouped kernels, silently ignored elsewhere (the
 *                     edge-gradient path computes the same values).
 *   grad_feat [n,h,f], grad_attn_row [m,h], grad_attn_col [n,h]: outputs, fully written
 *   (no pre-zeroing by the caller; the reference zeroes grad_attn_col itself, :1057). */
int dgnn_gat_backward(int m, int n, int nnz, int h, int f, float negative_slope, float attn_drop,
                      const int *row_ptr, const int *col_ind, const int *col_ptr,
                      const int *row_ind, const int *permute, const float *edge_max,
                      const float *edge_sum, const float *edge_mask, const float *in_feat,
                      const float *attn_row, const float *attn_col, const float *grad,
                      const float *out_feat, const float *row_dot, const float *out_neg,
                      const float *a_neg, float *grad_feat, float *grad_attn_row,
                      float *grad_attn_col, void *workspace, size_t workspace_bytes, void *stream);

/* dgnn_gat_backward with the per-row inputs handed over PACKED: row_stat [m,h,4] =
 * {attn_row, edge_max, edge_sum, row_dot} per (row, head), the 16-byte records the column-gather
 * kernel reads per (edge, head).  dgnn_gat_backward builds this table from its four [m,h] inputs
 * on every call; a caller that already holds it in this layout -- the row-partitioned multi-GPU
 * runner receives exactly these records from its peers, and runs several column chunks against
 * one table -- skips that pass.  No counterpart in the reference (its fused_backward_kernel,
 * fused_gatconv_kernel.cu:892-968, re-gathers the four arrays per edge).  Same outputs, bit for
 * bit, and same workspace (query with have_row_term = 1).  DGNN_ERR_UNSUPPORTED where the
 * neighbour-grouped kernels do not apply (h > 8, wide or misaligned rows, row_stat not 16-byte
 * aligned): unpack and call dgnn_gat_backward.
 * grad_attn_row may be NULL (here and in dgnn_gat_backward, neighbour-grouped kernels only): the
 * caller derives it from the forward's negative-score pair, grad_attn_row = (1 - slope)(row_dot *
 * a_neg - <grad, out_neg>) -- all per-row data the row's owner holds -- and no edge gradients are
 * produced at all. */
int dgnn_gat_backward_packed(int m, int n, int nnz, int h, int f, float negative_slope,
                             float attn_drop, const int *row_ptr, const int *col_ind,
                             const int *col_ptr, const int *row_ind, const int *permute,
                             const float *row_stat, const float *edge_mask, const float *in_feat,
                             const float *attn_col, const float *grad, float *grad_feat,
                             float *grad_attn_row, float *grad_attn_col, void *workspace,
                             size_t workspace_bytes, void *stream);

/* Dropout without a stored mask.  The reference draws edge_mask [nnz,h] ~ U(0,1] with cuRAND in
 * every forward and keeps it for the backward (fused_gatconv_kernel.cu:333-341: nnz*h*4 bytes
 * written, read three times, 1 GB per layer at the products shape).  The *_seeded entry points
 * take (seed, offset) instead: element i of the virtual row-major [nnz,h] mask is component i % 4
 * of Philox4x32-10 at counter (i / 4, offset) under key seed, mapped to (0,1]; the forward (by CSR
 * position) and the backward (through permute) regenerate it.  Same arguments, outputs and
 * workspace as dgnn_gat_forward / dgnn_gat_backward otherwise, and bit-identical results to those
 * given the mask dgnn_dropout_mask materialises for the same (seed, offset).  DGNN_ERR_UNSUPPORTED
 * where the neighbour-grouped kernels do not apply: materialise the mask and use the mask entry
 * points (same values). */
int dgnn_dropout_mask(int nnz, int h, uint64_t seed, uint64_t offset, float *mask, void *stream);
int dgnn_gat_forward_seeded(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                            const float *attn_row, const float *attn_col, const int *row_ptr,
                            const int *col_ind, const float *in_feat, uint64_t seed,
                            uint64_t offset, float *edge_max, float *edge_sum, float *out_feat,
                            float *out_neg, float *a_neg, void *workspace, size_t workspace_bytes,
                            void *stream);
int dgnn_gat_backward_seeded(int m, int n, int nnz, int h, int f, float negative_slope,
                             float attn_drop, const int *row_ptr, const int *col_ind,
                             const int *col_ptr, const int *row_ind, const int *permute,
                             const float *edge_max, const float *edge_sum, uint64_t seed,
                             uint64_t offset, const float *in_feat, const float *attn_row,
                             const float *attn_col, const float *grad, const float *out_feat,
                             const float *row_dot, const float *out_neg, const float *a_neg,
                             float *grad_feat, float *grad_attn_row, float *grad_attn_col,
                             void *workspace, size_t workspace_bytes, void *stream);

/* The seeded entry points with (seed, offset) read FROM DEVICE MEMORY at launch: rng_state points
 * to two uint64 {seed, offset} (8-byte aligned).  A launch captured into a HIP graph bakes its
 * scalar arguments in, so the by-value variants above would replay ONE mask forever; with the pair
 * in device memory the caller (or a captured kernel of its own) advances it between replays and
 * every replay draws a fresh mask, the backward reading the pair its forward used.  Same bits as
 * the by-value variants for the same pair.  No counterpart in the reference (cuRAND on the host,
 * fused_gatconv_kernel.cu:333-341, cannot be captured at all). */
int dgnn_gat_forward_seeded_dev(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                                const float *attn_row, const float *attn_col, const int *row_ptr,
                                const int *col_ind, const float *in_feat, const uint64_t *rng_state,
                                float *edge_max, float *edge_sum, float *out_feat, float *out_neg,
                                float *a_neg, void *workspace, size_t workspace_bytes, void *stream);
int dgnn_gat_backward_seeded_dev(int m, int n, int nnz, int h, int f, float negative_slope,
                                 float attn_drop, const int *row_ptr, const int *col_ind,
                                 const int *col_ptr, const int *row_ind, const int *permute,
                                 const float *edge_max, const float *edge_sum, const uint64_t *rng_state,
                                 const float *in_feat, const float *attn_row, const float *attn_col,
                                 const float *grad, const float *out_feat, const float *row_dot,
                                 const float *out_neg, const float *a_neg,
                                 float *grad_feat, float *grad_attn_row, float *grad_attn_col,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* The forward -> backward hand-over of a caller that keeps the REFERENCE'S OWN signatures.  The
 * reference's wrapper saves what gat_forward returns in its fourth slot -- the [nnz,h] edge_mask -- and
 * hands it back to gat_backward untouched (dgNN/operators/fused_gatconv.py:16-29; launchers
 * fused_gatconv_kernel.cu:322-361 and :998-1038); nobody else reads it.  Here that slot carries
 * `saved`, dgnn_gat_saved_floats(m, h, f) floats (16-byte aligned):
 *   [0, 4)                 the dropout {seed, offset} pair as two uint64, written by the forward
 *   [4, 4 + mhf')          a second copy of out_feat, written by the lanes that write out_feat itself
 *   [.., + mhf')           out_neg          [.., + mh')   a_neg          (x' = x rounded up to 4)
 * so that dgnn_gat_backward_saved runs the one-pass backward of dgnn_gat_backward(out_feat, out_neg,
 * a_neg) -- no rebuilt forward, no [nnz,h] edge gradients, and at attn_drop > 0 no 4 * nnz * h byte
 * mask: both passes regenerate it from the pair (as dgnn_gat_*_seeded_dev).  Same bits as the
 * dgnn_gat_forward_seeded / dgnn_gat_backward_seeded pair with (out_feat, out_neg, a_neg).
 * DGNN_ERR_UNSUPPORTED (nothing launched) for shapes outside the neighbour-grouped float4 kernels
 * (more than 8 heads, fewer than 4 features): such callers keep a materialised mask. */
size_t dgnn_gat_saved_floats(int m, int h, int f);
int dgnn_gat_forward_saved(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                           const float *attn_row, const float *attn_col, const int *row_ptr,
                           const int *col_ind, const float *in_feat, uint64_t seed, uint64_t offset,
                           float *edge_max, float *edge_sum, float *out_feat, float *saved,
                           void *workspace, size_t workspace_bytes, void *stream);
/* workspace: dgnn_gat_backward_workspace_bytes(m, n, nnz, h, f, 1) */
int dgnn_gat_backward_saved(int m, int n, int nnz, int h, int f, float negative_slope, float attn_drop,
                            const int *row_ptr, const int *col_ind, const int *col_ptr,
                            const int *row_ind, const int *permute, const float *edge_max,
                            const float *edge_sum, const float *saved, const float *in_feat,
                            const float *attn_row, const float *attn_col, const float *grad,
                            float *grad_feat, float *grad_attn_row, float *grad_attn_col,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------- GE-SpMM / SDDMM / permute ---- */

size_t dgnn_spmm_csr_workspace_bytes(int m, int nnz, int h, int f);

/* Multi-head CSR SpMM (GE-SpMM style): out[r,h,:] = sum_j edge_val[j,h] * in_feat[col_ind[j],h,:].
 * The reference has no standalone SpMM entry point; this is phase 3 of fused_forward_kernel
 * (fused_gatconv_kernel.cu:87-117) with caller-supplied edge values. */
int dgnn_spmm_csr(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                  const float *edge_val, const float *in_feat, float *out_feat,
                  void *workspace, size_t workspace_bytes, void *stream);

/* Multi-head SDDMM: out[e,h] = <grad[row(e),h,:], feature[col_ind[e],h,:]>.
 * Replaces mhsddmm_cuda(), dgNN/src/sddmm/mhsddmm_kernel.cu:83-99 (kernel :5-81). */
int dgnn_mhsddmm(int v, int f, int h, int nnz, const int *row_ptr, const int *col_ind,
                 const float *grad, const float *feature, float *out, void *stream);

/* out[i,:] = in[permute[i],:] for [nnz,h] edge data.
 * Replaces mhtranspose_cuda(), dgNN/src/csr2csc/mhtranspose_kernel.cu:28-47. */
int dgnn_mhtranspose(int nnz, int h, const int *permute, const float *in, float *out,
                     void *stream);

/* ------------------------------------------------------------------ fused GMMConv ---- */

size_t dgnn_gmm_forward_workspace_bytes(int nodes, int nnz, int kernels, int embed);

/* out[r,k,:] = sum_j exp(-0.5 * sum_d ((pseudo[j,d]-mu[k,d])*inv_sigma[k,d])^2) * node_feat[col_ind[j],k,:]
 * Replaces gmmconv_cuda(), dgNN/src/fused_gmmconv/fused_gmmconv_kernel.cu:104-122 (fuseGmm :24-55).
 * pseudo is [nnz,dim] in the nnz order of (csr_ptr, col_ind). */
int dgnn_gmm_forward(int nodes, int nnz, int kernels, int dim, int embed, const int *csr_ptr,
                     const int *col_ind, const float *node_feat, const float *pseudo,
                     const float *mu, const float *inv_sigma, float *out,
                     void *workspace, size_t workspace_bytes, void *stream);

/* gauss[e,k].  Replaces gaussian_cuda(), fused_gmmconv_kernel.cu:89-102 (kernel :11-22). */
int dgnn_gaussian(int edges, int kernels, int dim, const float *pseudo, const float *mu,
                  const float *inv_sigma, float *gauss, void *stream);

size_t dgnn_gaussian_backward_workspace_bytes(int edges, int kernels, int dim);

/* Gradients of gauss[e,k] w.r.t. pseudo [edges,dim], mu [k,dim], inv_sigma [k,dim] given
 * grad_gauss [edges,k].  Replaces gaussian_bp_cuda(), fused_gmmconv_kernel.cu:124-144, whose
 * kernel (:57-87) accumulates into uninitialised memory with a no-op warp reduction; this
 * computes the analytic gradient the kernel intends (see oracle/dgnn_oracle.c).  Outputs are
 * fully written and deterministic (two-stage reduction, no float atomics). */
int dgnn_gaussian_backward(int edges, int kernels, int dim, const float *pseudo,
                           const float *mu, const float *inv_sigma, const float *grad_gauss,
                           float *pseudo_out, float *mu_out, float *sigma_out,
                           void *workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------- fused EdgeConv ---- */

/* out[r,c] = max_j (h_dst[r,c] + h_src[src_ind[r*k+j],c]); max_idx[r,c] = arg-max source id
 * with the reference's tie rule (32-lane xor butterfly taking the partner on >=, later
 * 32-tiles win ties), bit-exact.  Replaces edgeconv_forward_cuda(),
 * dgNN/src/fused_edgeconv/fused_edgeconv_kernel.cu:51-74 (kernel :7-49).
 * max_idx_f32 [m,f] is the float copy the reference API returns (:46,:61); max_idx_i32 [m,f]
 * (optional, may be NULL) is the exact int32 copy used by backward. */
int dgnn_edgeconv_forward(int m, int f, int k, const int *src_ind, const float *h_src,
                          const float *h_dst, float *out_feat, float *max_idx_f32,
                          int *max_idx_i32, void *stream);

/* grad_src = 0; grad_src[max_idx[r,c],c] += grad_out[r,c].  Replaces edgeconv_backward_cuda(),
 * fused_edgeconv_kernel.cu:88-100 (kernel :76-86, which zero-fills and accumulates in one
 * launch without a grid sync; here the zero fill is a separate stream-ordered memset). */
int dgnn_edgeconv_backward(int m, int f, const int *max_idx, const float *grad_out,
                           float *grad_src, void *stream);

/* The same gradient WITHOUT float atomics, summed in a fixed order (run-to-run identical bits):
 * the k-NN lists src_ind [m*k] are transposed on the device (dgnn_csr2csc) and every source s
 * gathers, over the points r that list it in ascending r, grad_out[r,c] where max_idx[r,c] == s.
 * Needs the lists the forward was given (ids in [0, m)); the reference's backward signature
 * (fused_edgeconv.cpp:32-45) carries only grad_out and max_idx, hence a separate entry point. */
size_t dgnn_edgeconv_backward_det_workspace_bytes(int m, int k);
int dgnn_edgeconv_backward_det(int m, int f, int k, const int *src_ind, const int *max_idx,
                               const float *grad_out, float *grad_src, void *workspace,
                               size_t workspace_bytes, void *stream);

/* out[i,:] = in[idx[i],:], i < n_out, rows of `width` floats, idx in [0, n_in) (not checked on
 * the device).  No counterpart in the reference (single GPU): packs the rows a rank sends in the
 * halo exchange of the row-partitioned multi-GPU path (dgnn_amd/dist.py) into one contiguous
 * message. */
int dgnn_pack_rows(int n_out, int n_in, int width, const int *idx, const float *in, float *out,
                   void *stream);

/* -------------------------------------------------------------------- index work ---- */

size_t dgnn_csr2csc_workspace_bytes(int m, int n, int nnz);

/* CSR -> CSC with the permutation (bit-exact stable transpose).  Replaces csr2csc_cuda(),
 * dgNN/src/convert/convert_kernel.cu:25-42 (cuSPARSE csr2cscEx2 with csr_data = arange) and
 * the host scipy path of script/train/train_gatconv.py:87-94.  permute is int32 (the
 * reference returns it as float, exact only below 2^24). */
int dgnn_csr2csc(int m, int n, int nnz, const int *row_ptr, const int *col_ind, int *col_ptr,
                 int *row_ind, int *permute, void *workspace, size_t workspace_bytes,
                 void *stream);

/* Sorted COO row indices -> row_ptr [m+1].  Replaces coo2csr_cuda(), convert_kernel.cu:51-59. */
int dgnn_coo2csr(int m, int nnz, const int *coo_row, int *row_ptr, void *stream);

/* Fused GMMConv backward in ONE gather pass over the CSC (square graph): grad_node_feat[c,k,:] =
 * sum_i w_ik grad[r_i,k,:] with w_ik the Gaussian weight of CSC entry i (pseudo is in CSR nnz
 * order and is read through permute), and grad_gauss[permute[i],k] = <grad[r_i,k,:],
 * node_feat[c,k,:]> ([nnz,K], CSR order: the input of dgnn_gaussian_backward).  Replaces the
 * mhsddmm + mhtranspose + gmmconv-on-CSC sequence of dgNN/operators/fused_gmmconv.py:21-24 (two
 * feature gathers and a binary search per edge) by one gather.  Taken when all kernels fit one
 * lane pass: K * P <= 64 with P = the power of two covering ceil(embed/4) lanes (embed lanes when
 * embed % 4 != 0); otherwise DGNN_ERR_UNSUPPORTED / workspace size 0 and the caller keeps the
 * three-call sequence, which is the faster one there. */
size_t dgnn_gmm_backward_workspace_bytes(int nodes, int nnz, int kernels, int embed);
int dgnn_gmm_backward(int nodes, int nnz, int kernels, int dim, int embed, const int *col_ptr,
                      const int *row_ind, const int *permute, const float *node_feat,
                      const float *pseudo, const float *mu, const float *inv_sigma,
                      const float *grad, float *grad_node_feat, float *grad_gauss,
                      void *workspace, size_t workspace_bytes, void *stream);

/* The GMMConv LAYER sums the operator's output over the kernels right away
 * (dgNN/layers/gmmconv_layer.py:58: `GmmConvFuse(...).sum(1)`), i.e. it writes [nodes,K,F], reads it
 * back and writes [nodes,F], and autograd expands the [nodes,F] gradient to [nodes,K,F] again.
 * These two fold that sum into the kernels: the forward stores out_sum [nodes,F] = sum_k out[:,k,:]
 * directly (the K partial rows of a node sit in one wave and are added across lanes before the
 * store), and the one-pass backward reads grad_sum [nodes,F] for every kernel's lanes.  Same
 * arithmetic per (row, kernel) as dgnn_gmm_forward / dgnn_gmm_backward; the sum over kernels is
 * taken in ascending k.  Shapes: embed % 4 == 0, kernels * embed / 4 <= 64, 16-byte aligned rows
 * (and the conditions of dgnn_gmm_backward for the backward); otherwise DGNN_ERR_UNSUPPORTED and
 * the caller keeps the unsummed entry points.  Workspaces: the queries of the unsummed twins. */
int dgnn_gmm_forward_sum(int nodes, int nnz, int kernels, int dim, int embed, const int *csr_ptr,
                         const int *col_ind, const float *node_feat, const float *pseudo,
                         const float *mu, const float *inv_sigma, float *out_sum,
                         void *workspace, size_t workspace_bytes, void *stream);
int dgnn_gmm_backward_sum(int nodes, int nnz, int kernels, int dim, int embed, const int *col_ptr,
                          const int *row_ind, const int *permute, const float *node_feat,
                          const float *pseudo, const float *mu, const float *inv_sigma,
                          const float *grad_sum, float *grad_node_feat, float *grad_gauss,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Exact k-nearest-neighbour graph per point cloud: x [batch, points, dim] -> src_ind
 * [batch*points*k], src_ind[r*k + t] = global id (cloud offset included) of the t-th nearest
 * point of r within r's cloud, r itself included, ascending distance, ties -> lower id: the
 * layout edgeconv_forward consumes (fused_edgeconv_kernel.cu:21-27).  Replaces the DGL call in
 * front of the fused EdgeConv, KNNGraph(k)(h) + g.edges(), script/train/train_edgeconv.py:17,
 * 49-52 (third-party: dgl >= 0.7, README.md:29; not under the reference tree).  Distances are
 * the float32 fma chain sum_f (x_i[f]-x_j[f])^2, f ascending, so the selection is reproducible
 * bit for bit.  k <= min(points, 64). */
int dgnn_knn_graph(int batch, int points, int dim, int k, const float *x, int *src_ind,
                   void *stream);

/* The same graph, same ids bit for bit, with a workspace: for 17 .. 128 dims and enough queries to
 * fill the chip a coarse candidate filter runs on the matrix cores first (split-bf16 products
 * accumulated in float32 give, per query, a list of k + slack candidates by a LOWER BOUND of the
 * exact distance), the exact float32 chain is evaluated for those candidates only, and a query whose
 * k-th exact distance is not strictly below the bound of everything outside its list (ties,
 * duplicate points, non-finite input) is recomputed by the exhaustive kernel of dgnn_knn_graph.
 * Same reference call site as dgnn_knn_graph (script/train/train_edgeconv.py:17,49-52).
 * dgnn_knn_graph_workspace_bytes returns 0 when the filter does not apply (then the call is
 * dgnn_knn_graph); workspace == NULL always selects the exhaustive kernel.  256-byte aligned. */
size_t dgnn_knn_graph_workspace_bytes(int batch, int points, int dim, int k);
int dgnn_knn_graph_ws(int batch, int points, int dim, int k, const float *x, int *src_ind,
                      void *workspace, size_t workspace_bytes, void *stream);

/* ---- GAT layer steps on either side of the fused operator, one pass each -------------------
 * attn_row[n,h] = <attn_l[h,:], feat[n,h,:]>, attn_col[n,h] = <attn_r[h,:], feat[n,h,:]>.
 * Replaces the two multiply + sum chains of dgNN/layers/gatconv_layer.py:74-75 (attn_l is the
 * destination/row term there).  Supported: f % 4 == 0 with f <= 256 and 16-byte aligned
 * tensors, or f <= 64; otherwise DGNN_ERR_UNSUPPORTED (the caller keeps the torch expression). */
int dgnn_gat_attn_scores(int n, int h, int f, const float *feat, const float *attn_l,
                         const float *attn_r, float *attn_row, float *attn_col, void *stream);

size_t dgnn_gat_attn_scores_backward_workspace_bytes(int n, int h, int f);

/* Backward of the above, fused with the accumulation autograd would do afterwards:
 * grad_feat[n,h,:] += g_row[n,h]*attn_l[h,:] + g_col[n,h]*attn_r[h,:]   (IN PLACE on grad_feat,
 * normally the grad_feat dgnn_gat_backward just produced), grad_attn_l[h,:] = sum_n g_row[n,h]*
 * feat[n,h,:], grad_attn_r likewise with g_col.  Fixed-order sums (deterministic). */
int dgnn_gat_attn_scores_backward(int n, int h, int f, const float *feat, const float *attn_l,
                                  const float *attn_r, const float *g_row, const float *g_col,
                                  float *grad_feat, float *grad_attn_l, float *grad_attn_r,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* out = act(rst + residual + bias), residual [n,hf] and bias [hf] optional (NULL).  Replaces the
 * elementwise tail of dgNN/layers/gatconv_layer.py:82-90.  activation: 0 none, 1 ReLU,
 * 2 ELU(alpha = act_param), 3 LeakyReLU(slope = act_param). */
#define DGNN_ACT_NONE 0
#define DGNN_ACT_RELU 1
#define DGNN_ACT_ELU 2
#define DGNN_ACT_LEAKY_RELU 3
int dgnn_gat_epilogue(int n, int hf, int activation, float act_param, const float *rst,
                      const float *residual, const float *bias, float *out, void *stream);

size_t dgnn_gat_epilogue_backward_workspace_bytes(int n, int hf);

/* grad_pre = grad * act'(out) (the gradient of rst and of residual), grad_bias[hf] = column sums
 * of grad_pre (optional, NULL to skip; fixed-order sums).  `out` is the forward OUTPUT. */
int dgnn_gat_epilogue_backward(int n, int hf, int activation, float act_param, const float *out,
                               const float *grad, float *grad_pre, float *grad_bias,
                               void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DGNN_HIP_H */

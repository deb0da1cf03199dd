// C-ABI entry points of the row-gather family, FORWARD side (include/dgnn_hip.h): fused GAT forward /
// inference, GE-SpMM and fused GMMConv forward, plus kernel selection between the neighbour-grouped
// fast path (gather_fast.hpp) and the shape-agnostic wave-per-row path (gat_generic.hpp).  The
// backward side lives in gather_bwd.hip (two translation units: they build in parallel).
#include <cstdio>
#include <cstdlib>
#include <mutex>

#define DGNN_TU_FORWARD 1
#include "gat_generic.hpp"
#include "gather_fast.hpp"
#include "launch.hpp"

using namespace dgnn;
using namespace dgnn::host;

namespace dgnn {
namespace host {
KernelTimer g_timer;
}  // namespace host
}  // namespace dgnn

namespace {

// One instantiation of the forward kernel; narrow float4 rows (STEADY) pick the one whose packed
// phase C has the row's lane width as a compile-time constant (PS == P.pack_s: 4, 8 or 16; 0 = the
// one-edge-per-step streams).
template <int MODE, int VECV, int NSV, int HTV, bool STV, bool NEGV>
void launch_fwd_kernel(const GatherParams &P, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (STV && NSV == 1 && VECV == 4) {
    if (P.pack_s == 4) {
      hipLaunchKernelGGL((k_fast_row_gather<MODE, VECV, NSV, HTV, STV, NEGV, 4>), grid, dim3(DGNN_BLOCK), lds, st, P);
      return;
    }
    if (P.pack_s == 8) {
      hipLaunchKernelGGL((k_fast_row_gather<MODE, VECV, NSV, HTV, STV, NEGV, 8>), grid, dim3(DGNN_BLOCK), lds, st, P);
      return;
    }
    if (P.pack_s == 16) {
      hipLaunchKernelGGL((k_fast_row_gather<MODE, VECV, NSV, HTV, STV, NEGV, 16>), grid, dim3(DGNN_BLOCK), lds, st, P);
      return;
    }
  }
  hipLaunchKernelGGL((k_fast_row_gather<MODE, VECV, NSV, HTV, STV, NEGV, 0>), grid, dim3(DGNN_BLOCK), lds, st, P);
}

template <int MODE>
int launch_fast_gather(GatherParams P, void *workspace, size_t workspace_bytes, hipStream_t st) {
  const bool neg = MODE == MODE_GAT && P.out_neg != nullptr;
  const FwdLayout w = fwd_layout(P.nnz, P.H, P.F, neg, MODE == MODE_GAT ? 1 : (MODE == MODE_VAL ? 2 : 0));
  if (!workspace || workspace_bytes < w.total) return DGNN_ERR_WORKSPACE;
  P.HF = P.H * P.F;
  P.L = w.L;
  P.T = w.T;
  P.slot_stride = w.slot_stride;
  P.slot_neg = w.slot_neg;
  P.task_row0 = (int *)((char *)workspace + w.off_rows);
  P.slots = (float *)((char *)workspace + w.off_slots);
  // float4 lanes for every width of at least four floats: rows need no 16-byte alignment (common.hpp:
  // ld4u), so odd widths (1 x 47, 4 x 41 ...) and misaligned tensors take the same kernels.  The
  // summed store (GMM layer) keeps its aligned float4 lanes: gmm_sum_shape() admits F % 4 == 0 only.
  const bool vec4 = P.F >= 4 && (!P.sum_heads || (P.F % 4 == 0 && aligned16(P.out)));
  const bool attn_ok = (MODE != MODE_GAT) || P.H != 4 || aligned16(P.attn_col);
  if (!attn_ok) return DGNN_ERR_ALIGNMENT;
  // the negative-score pair exists for float4 rows only (the caller falls back to the edge-gradient path)
  if (neg && !(vec4 && aligned16(P.a_neg))) return DGNN_ERR_UNSUPPORTED;
  P.vec4 = vec4 ? 1 : 0;
  P.spf = vec4 ? (P.F + 3) / 4 : P.F;
  P.rowv = P.H * P.spf;
  // rows narrower than one wave of float4 slots are bound by instructions per edge, not bandwidth
  // (the STEADY instantiation of the kernel; DGNN_FWD_STEADY=0 switches it off for A/B runs)
  P.steady = (vec4 && P.rowv < DGNN_WAVE && env_int("DGNN_FWD_STEADY", 1) != 0) ? 1 : 0;
  // rows of at most 32 float4 slots: several edges per load step (DGNN_FWD_PACK=0: A/B runs)
  P.pack_s = (P.steady && env_int("DGNN_FWD_PACK", 1) != 0) ? fast_pack_lanes(P.rowv, vec4) : 0;
  P.batch_rows = env_int("DGNN_FWD_BATCH4", 1);
  {
    long long g = ((long long)P.m + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(k_fast_prep_rows, dim3((unsigned)g), dim3(256), 0, st, P);
    DGNN_LAUNCH_CHECK();
  }
  if (P.T == 0) return DGNN_OK;
  // DGNN_FWD_LDS_PAD: extra dynamic LDS per workgroup = an occupancy limiter for tuning runs
  const size_t lds = fast_lds_bytes(P.H, P.L, P.pack_s > 0) + (size_t)env_int("DGNN_FWD_LDS_PAD", 0);
  const dim3 grid(blocks_for_waves(P.T));
  timer_mark(0, 0, st);
  // vector slots per lane: one walk over the edges for rows up to 64 / 128 / 256 slots
  const int ns = P.rowv <= DGNN_WAVE ? 1 : P.rowv <= 2 * DGNN_WAVE ? 2 : 4;
  const bool steady = P.steady != 0;
  // all-heads-at-once softmax phase for 1 / 2 / 8 heads (per-head vectors need aligned tensors)
  int ht = 0;
  if (MODE == MODE_GAT && (P.H == 1 || P.H == 2 || P.H == 8) && aligned16(P.attn_row) &&
      aligned16(P.attn_col) && aligned16(P.edge_mask) && aligned16(P.edge_max) && aligned16(P.edge_sum))
    ht = P.H;
  // four heads: narrow rows (steady) take the per-head-vector phase B too -- it is the one that puts
  // up to four short rows into a pass; wide rows keep the float4 branch with its row cache
  // (DGNN_FWD_HT4=1 / 0 forces either, A/B runs)
  if (MODE == MODE_GAT && P.H == 4 && vec4 && ns == 1 && aligned16(P.attn_row) && aligned16(P.attn_col) &&
      aligned16(P.edge_mask) && aligned16(P.edge_max) && aligned16(P.edge_sum) &&
      env_int("DGNN_FWD_HT4", steady ? 1 : 0) != 0)
    ht = 4;
  // 3, 5, 6, 7 heads, rows of one lane pass: the same phase with scalar per-head loads (no alignment needed); until
  // round 3 they took the run-time-H loops (6 x 8 forward slower than 8 x 8).  DGNN_FWD_HT_ODD=0: A/B runs.
  if (MODE == MODE_GAT && (P.H == 3 || P.H == 5 || P.H == 6 || P.H == 7) && vec4 && ns == 1 &&
      env_int("DGNN_FWD_HT_ODD", 1) != 0)
    ht = P.H;
#define DGNN_FWD_CASE(VECV, NSV, HTV, STV)                                                     \
  if ((vec4 ? 4 : 1) == VECV && ns == NSV && ht == HTV && steady == STV) {                     \
    if (timing_on())                                                                           \
      std::snprintf(g_timer.name[0], sizeof(g_timer.name[0]), "k_fast_row_gather<%d, %d, %d, %d, %s, %s, %d>", MODE, \
                    VECV, NSV, HTV, STV ? "true" : "false", (MODE == MODE_GAT && VECV == 4 && neg) ? "true" : "false", \
                    (STV && NSV == 1 && VECV == 4) ? P.pack_s : 0);                            \
    if (MODE == MODE_GAT && VECV == 4 && neg)                                                  \
      launch_fwd_kernel<MODE, VECV, NSV, HTV, STV, (MODE == MODE_GAT && VECV == 4)>(P, grid, lds, st); \
    else                                                                                       \
      launch_fwd_kernel<MODE, VECV, NSV, HTV, STV, false>(P, grid, lds, st);                   \
  }
#define DGNN_FWD_CASES(HTV)                                                                    \
  DGNN_FWD_CASE(4, 1, HTV, true)                                                               \
  DGNN_FWD_CASE(4, 1, HTV, false) DGNN_FWD_CASE(4, 2, HTV, false) DGNN_FWD_CASE(4, 4, HTV, false) \
  DGNN_FWD_CASE(1, 1, HTV, false) DGNN_FWD_CASE(1, 2, HTV, false) DGNN_FWD_CASE(1, 4, HTV, false)
  DGNN_FWD_CASES(0)
  if constexpr (MODE == MODE_GAT) {        // (constexpr: SpMM / GMM never instantiate the per-head-vector kernels)
    DGNN_FWD_CASES(1) DGNN_FWD_CASES(2) DGNN_FWD_CASES(8)
    DGNN_FWD_CASE(4, 1, 4, true) DGNN_FWD_CASE(4, 1, 4, false)
    DGNN_FWD_CASE(4, 1, 3, true) DGNN_FWD_CASE(4, 1, 3, false) DGNN_FWD_CASE(4, 1, 5, true) DGNN_FWD_CASE(4, 1, 5, false)
    DGNN_FWD_CASE(4, 1, 6, true) DGNN_FWD_CASE(4, 1, 6, false) DGNN_FWD_CASE(4, 1, 7, true) DGNN_FWD_CASE(4, 1, 7, false)
  }
#undef DGNN_FWD_CASES
#undef DGNN_FWD_CASE
  timer_mark(0, 1, st);
  DGNN_LAUNCH_CHECK();
  if (P.T > 1) {
    // two launches: long rows first merge groups of DGNN_FIX_C pieces (phase 0), then everything (phase 1)
    if (P.T - 1 > DGNN_FIX_C) {
      hipLaunchKernelGGL(k_fast_fixup<MODE>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 0);
      DGNN_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_fast_fixup<MODE>, dim3(blocks_for_waves(P.T - 1)), dim3(DGNN_BLOCK), 0, st, P, 1);
    DGNN_LAUNCH_CHECK();
  }
  return DGNN_OK;
}

}  // namespace

namespace dgnn {
namespace host {
int launch_forward(int m, int nnz, int h, int f, float attn_drop, float slope,
                   const float *attn_row, const float *attn_col, const int *row_ptr,
                   const int *col_ind, const float *in_feat, const float *edge_mask,
                   float *edge_max, float *edge_sum, float *out_feat, void *workspace,
                   size_t workspace_bytes, hipStream_t st, DropRng rng, float *out_neg, float *a_neg, float *out2) {
  if (m == 0) return DGNN_OK;
  if ((out_neg == nullptr) != (a_neg == nullptr)) return DGNN_ERR_BAD_ARG;
  if (out_neg != nullptr && (!edge_max || !edge_sum)) return DGNN_ERR_BAD_ARG;
  // the H == 4 kernels move the per-row / per-edge scalars as float4
  const bool h4_aligned = aligned16(attn_col) && aligned16(attn_row) && aligned16(edge_mask) &&
                          aligned16(edge_max) && aligned16(edge_sum);
  if (fwd_fast_ok(m, h) && (h != 4 || h4_aligned)) {
    GatherParams P = {};
    P.m = m; P.nnz = nnz; P.H = h; P.F = f;
    P.attn_drop = attn_drop; P.slope = slope;
    P.row_ptr = row_ptr; P.col_ind = col_ind;
    P.attn_row = attn_row; P.attn_col = attn_col; P.edge_mask = edge_mask; P.rng = rng;
    P.in_feat = in_feat; P.edge_max = edge_max; P.edge_sum = edge_sum; P.out = out_feat;
    P.out_neg = out_neg; P.a_neg = a_neg; P.out2 = out_neg != nullptr ? out2 : nullptr;
    return launch_fast_gather<MODE_GAT>(P, workspace, workspace_bytes, st);
  }
  if (out_neg != nullptr) return DGNN_ERR_UNSUPPORTED;   // the wave-per-row kernels have no negative-score pair
  if (rng.on) return DGNN_ERR_UNSUPPORTED;   // the wave-per-row kernels read a materialised mask
  hipLaunchKernelGGL(k_gat_fwd_row, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0, st, m, h, f,
                     attn_drop, slope, attn_row, attn_col, row_ptr, col_ind, in_feat, edge_mask,
                     edge_max, edge_sum, out_feat);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

}  // namespace host
}  // namespace dgnn


extern "C" {

int dgnn_debug_last_kernel_ms(int which, float *ms) {
  if (which < 0 || which > 1 || !ms) return DGNN_ERR_BAD_ARG;
  if (!timing_on() || !g_timer.valid[which]) return DGNN_ERR_UNSUPPORTED;
  hipError_t e = hipEventSynchronize(g_timer.ev[which][1]);
  if (e != hipSuccess) return (int)e;
  e = hipEventElapsedTime(ms, g_timer.ev[which][0], g_timer.ev[which][1]);
  return e == hipSuccess ? DGNN_OK : (int)e;
}

#ifdef DGNN_PHASE_CLOCK
// diagnostic build only: the per-phase wave-cycle sums of the two gather kernels (common.hpp)
int dgnn_debug_phase_cycles(unsigned long long *out16, int reset) {
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

long long dgnn_debug_kernel_launches(int which) {
  if (which < 0 || which > 1 || !timing_on()) return -1;
  return g_timer.launches[which];
}

const char *dgnn_debug_last_kernel_name(int which) {
  if (which < 0 || which > 1 || !timing_on() || !g_timer.valid[which]) return nullptr;
  return g_timer.name[which];
}

size_t dgnn_gat_forward_workspace_bytes(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return 0;
  return fwd_layout(nnz, h, f, true).total;      // enough with or without the out_neg / a_neg pair
}

int dgnn_gat_forward(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                     const float *attn_row, const float *attn_col, const int *row_ptr,
                     const int *col_ind, const float *in_feat, const float *edge_mask,
                     float *edge_max, float *edge_sum, float *out_feat, float *out_neg, float *a_neg,
                     void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (m == 0) return DGNN_OK;                          // nothing to write
  if (!attn_row || !row_ptr || !out_feat || !edge_max || !edge_sum) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!attn_col || !in_feat)) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && !col_ind) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (attn_drop > 0.f && nnz > 0 && !edge_mask) return DGNN_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < fwd_layout(nnz, h, f, out_neg != nullptr).total)
    return DGNN_ERR_WORKSPACE;
  return launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr,
                        col_ind, in_feat, attn_drop > 0.f ? edge_mask : nullptr, edge_max, edge_sum,
                        out_feat, workspace, workspace_bytes, (hipStream_t)stream, DropRng{0, 0, 0, nullptr},
                        out_neg, a_neg);
}

// ------------------------------------------------------------ entry point that allocates nothing
// dgnn_gat_forward for callers that cannot bring a workspace (HIP-graph capture, serving loops that
// forbid allocation): the neighbour-group partial slots live in a file-scope device array.  Every
// GPU the library runs on holds its own copy of the array at its own address, so the address is
// looked up once per device (hipGetSymbolAddress with that device current) and kept; host code only
// ever uses that pointer, never the array's name.
}  // extern "C"
static __device__ __align__(256) unsigned char scratch[DGNN_NOALLOC_SCRATCH_BYTES];
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[kMaxDevices] = {};
int scratch_of_current_device(void **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return DGNN_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    void *ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (ptr == nullptr) return DGNN_ERR_WORKSPACE;
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return DGNN_OK;
}
}  // namespace
extern "C" {

int dgnn_gat_forward_noalloc(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                             const float *attn_row, const float *attn_col, const int *row_ptr,
                             const int *col_ind, const float *in_feat, const float *edge_mask,
                             float *edge_max, float *edge_sum, float *out_feat, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  // sizes beyond the static array are refused before anything is launched
  if (fwd_layout(nnz, h, f).total > (size_t)DGNN_NOALLOC_SCRATCH_BYTES) return DGNN_ERR_TOO_LARGE;
  void *ptr = nullptr;
  rc = scratch_of_current_device(&ptr);
  if (rc) return rc;
  // The kernels write every slot and every task_row0 entry they later read (each group holds at
  // least one edge, hence one row), so nothing of the previous call's contents needs clearing.
  return dgnn_gat_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr,
                          col_ind, in_feat, edge_mask, edge_max, edge_sum, out_feat, nullptr, nullptr, ptr,
                          (size_t)DGNN_NOALLOC_SCRATCH_BYTES, stream);
}

int dgnn_gat_inference(int m, int nnz, int h, int f, float negative_slope, const float *attn_row,
                       const float *attn_col, const int *row_ptr, const int *col_ind,
                       const float *in_feat, float *out_feat, void *workspace,
                       size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (m == 0) return DGNN_OK;
  if (!attn_row || !row_ptr || !out_feat) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!attn_col || !in_feat)) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && !col_ind) return DGNN_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < fwd_layout(nnz, h, f).total) return DGNN_ERR_WORKSPACE;
  return launch_forward(m, nnz, h, f, 0.f, negative_slope, attn_row, attn_col, row_ptr, col_ind,
                        in_feat, nullptr, nullptr, nullptr, out_feat, workspace, workspace_bytes,
                        (hipStream_t)stream);
}

size_t dgnn_gat_forward_tiles_workspace_bytes(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return 0;
  return align_up(fwd_layout(nnz, h, f, true).total, 256) + 256;   // + the validation flag
}

}  // extern "C"

// The list the reference's scheduler loop builds (script/test/test_gat_tb.py:16-26) names every
// non-empty row once, rows ascending, with tile_in_row = 0, 1, 2 ... per row.  One thread per
// entry checks it against its predecessor and row_ptr; any violation raises the flag.
__global__ void k_check_tile_list(int m, int n_tiles, int tile_nnz, const int *__restrict__ row_ptr,
                                  const int *__restrict__ tiles, int *__restrict__ flag) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i > n_tiles) return;
  // entry n_tiles is a virtual terminator (row m): rows after the last entry must be empty
  const int rid = i < n_tiles ? tiles[2 * i] : m;
  const int tin = i < n_tiles ? tiles[2 * i + 1] : 0;
  bool bad = false;
  if (i < n_tiles) {
    if (rid < 0 || rid >= m || tin < 0) bad = true;
    else if ((long long)tin * tile_nnz >= row_ptr[rid + 1] - row_ptr[rid]) bad = true;   // the tile starts past its row's end
  }
  if (!bad) {
    const int prid = i > 0 ? tiles[2 * (i - 1)] : -1;
    const int ptin = i > 0 ? tiles[2 * (i - 1) + 1] : 0;
    if (i > 0 && (prid < 0 || prid >= m)) {
      bad = true;                                        // reported by the predecessor's own thread too
    } else if (rid == prid) {
      bad = tin != ptin + 1;                             // tiles of a row: consecutive
    } else if (rid < prid) {
      bad = true;                                        // rows ascending
    } else {
      // a new row starts with its tile 0, every row skipped in between is empty, and the
      // previous row's last listed tile reaches that row's end
      bad = tin != 0 || row_ptr[rid] != row_ptr[prid + 1];
      if (i > 0 && ((long long)ptin + 1) * tile_nnz < row_ptr[prid + 1] - row_ptr[prid]) bad = true;
    }
  }
  if (bad) atomicOr(flag, 1);
}

extern "C" {

int dgnn_gat_forward_tiles(int m, int nnz, int h, int f, float negative_slope,
                           const float *attn_row, const float *attn_col, const int *row_ptr,
                           const int *col_ind, const float *in_feat, const int *tile_scheduler,
                           int n_tiles, int tile_nnz, float *edge_max, float *edge_sum, float *out_feat,
                           void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (tile_nnz <= 0) return DGNN_ERR_BAD_ARG;
  if (n_tiles < 0 || (n_tiles > 0 && !tile_scheduler) || (m > 0 && !row_ptr)) return DGNN_ERR_BAD_ARG;
  // no rows: nothing to check against (row_ptr may be NULL) and nothing to compute; a list that
  // names tiles of a graph without rows is malformed
  if (m == 0) return n_tiles > 0 ? DGNN_ERR_BAD_ARG : DGNN_OK;
  const size_t fwd_bytes = align_up(fwd_layout(nnz, h, f, true).total, 256);
  if (!workspace || workspace_bytes < fwd_bytes + 256) return DGNN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int *flag = (int *)((char *)workspace + fwd_bytes);
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_check_tile_list, dim3((unsigned)(((long long)n_tiles + 1 + 255) / 256)), dim3(256), 0, st,
                     m, n_tiles, tile_nnz, row_ptr, tile_scheduler, flag);
  DGNN_LAUNCH_CHECK();
  int host_flag = 0;
  e = hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e != hipSuccess) return (int)e;
  e = hipStreamSynchronize(st);      // the one entry point that waits: a malformed list is an ERROR CODE
  if (e != hipSuccess) return (int)e;
  if (host_flag != 0) return DGNN_ERR_BAD_ARG;
  return dgnn_gat_forward(m, nnz, h, f, 0.f, negative_slope, attn_row, attn_col, row_ptr, col_ind,
                          in_feat, nullptr, edge_max, edge_sum, out_feat, nullptr, nullptr, workspace, fwd_bytes,
                          stream);
}


// Materialise the mask the seeded entry points regenerate (one thread per Philox block of four).
__global__ void k_dropout_mask(long long total, DropRng rng, float *__restrict__ mask) {
  drop_resolve(rng);
  for (long long b = blockIdx.x * (long long)blockDim.x + threadIdx.x; b * 4 < total;
       b += (long long)gridDim.x * blockDim.x) {
    const float4 v = drop_uniform4(rng, (unsigned long long)b);
    const float a[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < 4 && b * 4 + i < total; ++i) mask[b * 4 + i] = a[i];
  }
}

int dgnn_dropout_mask(int nnz, int h, uint64_t seed, uint64_t offset, float *mask, void *stream) {
  if (nnz < 0 || h <= 0) return DGNN_ERR_BAD_ARG;
  const long long total = (long long)nnz * h;
  if (total == 0) return DGNN_OK;
  if (!mask) return DGNN_ERR_BAD_ARG;
  long long g = ((total + 3) / 4 + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, total,
                     DropRng{seed, offset, 1, nullptr}, mask);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_gat_forward_seeded(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                            const float *attn_row, const float *attn_col, const int *row_ptr,
                            const int *col_ind, const float *in_feat, uint64_t seed, uint64_t offset,
                            float *edge_max, float *edge_sum, float *out_feat, float *out_neg,
                            float *a_neg, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (m == 0) return DGNN_OK;
  if (!attn_row || !row_ptr || !out_feat || !edge_max || !edge_sum) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!attn_col || !in_feat || !col_ind)) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < fwd_layout(nnz, h, f, out_neg != nullptr).total)
    return DGNN_ERR_WORKSPACE;
  return launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr, col_ind,
                        in_feat, nullptr, edge_max, edge_sum, out_feat, workspace, workspace_bytes,
                        (hipStream_t)stream, DropRng{seed, offset, attn_drop > 0.f ? 1 : 0, nullptr}, out_neg,
                        a_neg);
}

int dgnn_gat_forward_seeded_dev(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                                const float *attn_row, const float *attn_col, const int *row_ptr,
                                const int *col_ind, const float *in_feat, const uint64_t *rng_state,
                                float *edge_max, float *edge_sum, float *out_feat, float *out_neg,
                                float *a_neg, void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (m == 0) return DGNN_OK;
  if (!attn_row || !row_ptr || !out_feat || !edge_max || !edge_sum) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!attn_col || !in_feat || !col_ind)) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (attn_drop > 0.f && (!rng_state || ((uintptr_t)rng_state & 7u))) return DGNN_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < fwd_layout(nnz, h, f, out_neg != nullptr).total)
    return DGNN_ERR_WORKSPACE;
  return launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr, col_ind,
                        in_feat, nullptr, edge_max, edge_sum, out_feat, workspace, workspace_bytes,
                        (hipStream_t)stream,
                        DropRng{0, 0, attn_drop > 0.f ? 1 : 0, (const unsigned long long *)rng_state}, out_neg,
                        a_neg);
}


// ------------------------------------------------- forward state carried in the edge_mask slot
__global__ void k_write_rng_pair(unsigned long long seed, unsigned long long offset, unsigned long long *dst) {
  dst[0] = seed;
  dst[1] = offset;
}

size_t dgnn_gat_saved_floats(int m, int h, int f) {
  if (m < 0 || h <= 0 || f <= 0) return 0;
  return saved_layout(m, h, f).total;
}

int dgnn_gat_forward_saved(int m, int nnz, int h, int f, float attn_drop, float negative_slope,
                           const float *attn_row, const float *attn_col, const int *row_ptr,
                           const int *col_ind, const float *in_feat, uint64_t seed, uint64_t offset,
                           float *edge_max, float *edge_sum, float *out_feat, float *saved, void *workspace,
                           size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (m == 0) return DGNN_OK;
  if (!attn_row || !row_ptr || !out_feat || !edge_max || !edge_sum || !saved) return DGNN_ERR_BAD_ARG;
  if (nnz > 0 && (!attn_col || !in_feat || !col_ind)) return DGNN_ERR_BAD_ARG;
  if (!(attn_drop >= 0.f && attn_drop < 1.f)) return DGNN_ERR_BAD_ARG;
  if (!aligned16(saved)) return DGNN_ERR_ALIGNMENT;
  // the pair exists on the neighbour-grouped float4 kernels only; elsewhere the caller keeps a mask
  if (!fwd_fast_ok(m, h) || f < 4) return DGNN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < fwd_layout(nnz, h, f, true).total) return DGNN_ERR_WORKSPACE;
  const SavedLayout w = saved_layout(m, h, f);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_write_rng_pair, dim3(1), dim3(1), 0, st, (unsigned long long)seed, (unsigned long long)offset,
                     (unsigned long long *)saved);
  DGNN_LAUNCH_CHECK();
  return launch_forward(m, nnz, h, f, attn_drop, negative_slope, attn_row, attn_col, row_ptr, col_ind, in_feat,
                        nullptr, edge_max, edge_sum, out_feat, workspace, workspace_bytes, st,
                        DropRng{seed, offset, attn_drop > 0.f ? 1 : 0, nullptr}, saved + w.off_neg, saved + w.off_aneg,
                        saved + w.off_out);
}

// ------------------------------------------------------------------------- GE-SpMM / GMM
size_t dgnn_spmm_csr_workspace_bytes(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return 0;
  return fwd_layout(nnz, h, f, false, 2).total;
}

int dgnn_spmm_csr(int m, int nnz, int h, int f, const int *row_ptr, const int *col_ind,
                  const float *edge_val, const float *in_feat, float *out_feat, void *workspace,
                  size_t workspace_bytes, void *stream) {
  int rc = check_dims(m, nnz, h, f);
  if (rc) return rc;
  if (m == 0) return DGNN_OK;
  if (!row_ptr || !out_feat || (nnz > 0 && (!col_ind || !edge_val || !in_feat))) return DGNN_ERR_BAD_ARG;
  if (fwd_fast_ok(m, h)) {
    GatherParams P = {};
    P.m = m; P.nnz = nnz; P.H = h; P.F = f;
    P.row_ptr = row_ptr; P.col_ind = col_ind; P.edge_val = edge_val;
    P.in_feat = in_feat; P.out = out_feat;
    return launch_fast_gather<MODE_VAL>(P, workspace, workspace_bytes, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(k_spmm_row<0>, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0,
                     (hipStream_t)stream, m, h, f, 0, row_ptr, col_ind, edge_val,
                     (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, in_feat,
                     out_feat);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}


size_t dgnn_gmm_forward_workspace_bytes(int nodes, int nnz, int kernels, int embed) {
  if (nodes < 0 || nnz < 0 || kernels <= 0 || embed <= 0) return 0;
  return fwd_layout(nnz, kernels, embed, false, 0).total;
}

// float4 rows of at most 64 slots, one lane pass: where the sum over kernels can be folded into the store
static bool gmm_sum_shape(int nodes, int kernels, int embed) {
  return fwd_fast_ok(nodes, kernels) && embed % 4 == 0 && kernels * embed / 4 <= DGNN_WAVE;
}

static int gmm_forward_impl(int nodes, int nnz, int kernels, int dim, int embed, const int *csr_ptr,
                            const int *col_ind, const float *node_feat, const float *pseudo,
                            const float *mu, const float *inv_sigma, float *out, int sum_kernels,
                            void *workspace, size_t workspace_bytes, void *stream) {
  int rc = check_dims(nodes, nnz, kernels, embed);
  if (rc) return rc;
  if (dim <= 0) return DGNN_ERR_BAD_ARG;
  if (nodes == 0) return DGNN_OK;
  if (!csr_ptr || !mu || !inv_sigma || !out || (nnz > 0 && (!col_ind || !pseudo || !node_feat)))
    return DGNN_ERR_BAD_ARG;
  if (sum_kernels && !(gmm_sum_shape(nodes, kernels, embed) && aligned16(node_feat) && aligned16(out)))
    return DGNN_ERR_UNSUPPORTED;
  if (fwd_fast_ok(nodes, kernels)) {
    GatherParams P = {};
    P.m = nodes; P.nnz = nnz; P.H = kernels; P.F = embed; P.dim = dim;
    P.row_ptr = csr_ptr; P.col_ind = col_ind; P.pseudo = pseudo; P.mu = mu; P.inv_sigma = inv_sigma;
    P.in_feat = node_feat; P.out = out; P.sum_heads = sum_kernels ? 1 : 0;
    return launch_fast_gather<MODE_GMM>(P, workspace, workspace_bytes, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(k_spmm_row<1>, dim3(blocks_for_waves(nodes)), dim3(DGNN_BLOCK), 0,
                     (hipStream_t)stream, nodes, kernels, embed, dim, csr_ptr, col_ind,
                     (const float *)nullptr, pseudo, mu, inv_sigma, node_feat, out);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_gmm_forward(int nodes, int nnz, int kernels, int dim, int embed, const int *csr_ptr,
                     const int *col_ind, const float *node_feat, const float *pseudo,
                     const float *mu, const float *inv_sigma, float *out, void *workspace,
                     size_t workspace_bytes, void *stream) {
  return gmm_forward_impl(nodes, nnz, kernels, dim, embed, csr_ptr, col_ind, node_feat, pseudo, mu, inv_sigma, out,
                          0, workspace, workspace_bytes, stream);
}

int dgnn_gmm_forward_sum(int nodes, int nnz, int kernels, int dim, int embed, const int *csr_ptr,
                         const int *col_ind, const float *node_feat, const float *pseudo,
                         const float *mu, const float *inv_sigma, float *out_sum, void *workspace,
                         size_t workspace_bytes, void *stream) {
  return gmm_forward_impl(nodes, nnz, kernels, dim, embed, csr_ptr, col_ind, node_feat, pseudo, mu, inv_sigma,
                          out_sum, 1, workspace, workspace_bytes, stream);
}

}  // extern "C"


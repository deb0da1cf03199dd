// C-ABI entry points of the row-gather family (include/dgnn_hip.h): fused GAT forward /
// inference / backward, GE-SpMM and fused GMMConv forward, plus kernel selection between the
// neighbour-grouped fast path (gather_fast.hpp, gat_bwd_fast.hpp) and the shape-agnostic
// wave-per-row path (gat_generic.hpp).
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "gat_bwd_fast.hpp"
#include "gat_generic.hpp"
#include "gather_fast.hpp"

using namespace dgnn;

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }
inline bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

// DGNN_FORCE_GENERIC=1 routes everything through the wave-per-row kernels (A/B testing).
inline bool force_generic() {
  const char *e = std::getenv("DGNN_FORCE_GENERIC");
  return e != nullptr && e[0] == '1';
}

// Optional timing of the two dominant kernels (DGNN_TIMING=1): HIP events recorded on the launch
// stream right around the kernel, read back with dgnn_debug_last_kernel_ms().  Off by default;
// this is the only process-global state in the library and it never changes results.
struct KernelTimer {
  bool on = false, init = false;
  hipEvent_t ev[2][2];
  bool valid[2] = {false, false};
  char name[2][96] = {{0}, {0}};   // template instantiation of the kernel last timed, as rocprofv3 prints it
};
KernelTimer g_timer;
inline bool timing_on() {
  if (!g_timer.init) {
    g_timer.init = true;
    g_timer.on = env_int("DGNN_TIMING", 0) != 0;
    if (g_timer.on)
      for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 2; ++i)
          if (hipEventCreate(&g_timer.ev[k][i]) != hipSuccess) g_timer.on = false;
  }
  return g_timer.on;
}
inline void timer_mark(int which, int edge, hipStream_t st) {
  if (timing_on()) {
    (void)hipEventRecord(g_timer.ev[which][edge], st);
    if (edge == 1) g_timer.valid[which] = true;
  }
}

// ----------------------------------------------------------------------------- forward side
struct FwdLayout {
  int L, T, slot_stride, slot_neg;
  size_t off_rows, off_slots, total;
};

// neg: partial slots also hold the negative-score aggregate and weight sum (GatherParams::out_neg)
FwdLayout fwd_layout(int nnz, int h, int f, bool neg = false) {
  FwdLayout w;
  w.L = fast_group_size(h, nnz, narrow_rows(h, f));
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

bool fwd_fast_ok(int m, int h) { return h <= DGNN_FAST_HMAX && m < (1 << 30) && !force_generic(); }

template <int MODE>
int launch_fast_gather(GatherParams P, void *workspace, size_t workspace_bytes, hipStream_t st) {
  const bool neg = MODE == MODE_GAT && P.out_neg != nullptr;
  const FwdLayout w = fwd_layout(P.nnz, P.H, P.F, neg);
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
  if (MODE == MODE_GAT) timer_mark(0, 0, st);
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
#define DGNN_FWD_CASE(VECV, NSV, HTV, STV)                                                     \
  if ((vec4 ? 4 : 1) == VECV && ns == NSV && ht == HTV && steady == STV) {                     \
    if (MODE == MODE_GAT && timing_on())                                                       \
      std::snprintf(g_timer.name[0], sizeof(g_timer.name[0]), "k_fast_row_gather<%d, %d, %d, %d, %s>", MODE, VECV, \
                    NSV, HTV, STV ? "true" : "false");                                         \
    if (MODE == MODE_GAT && VECV == 4 && neg)                                                  \
      hipLaunchKernelGGL((k_fast_row_gather<MODE, VECV, NSV, HTV, STV, (MODE == MODE_GAT && VECV == 4)>), grid, \
                         dim3(DGNN_BLOCK), lds, st, P);                                        \
    else                                                                                       \
      hipLaunchKernelGGL((k_fast_row_gather<MODE, VECV, NSV, HTV, STV>), grid, dim3(DGNN_BLOCK), lds, st, P); \
  }
#define DGNN_FWD_CASES(HTV)                                                                    \
  DGNN_FWD_CASE(4, 1, HTV, true)                                                               \
  DGNN_FWD_CASE(4, 1, HTV, false) DGNN_FWD_CASE(4, 2, HTV, false) DGNN_FWD_CASE(4, 4, HTV, false) \
  DGNN_FWD_CASE(1, 1, HTV, false) DGNN_FWD_CASE(1, 2, HTV, false) DGNN_FWD_CASE(1, 4, HTV, false)
  DGNN_FWD_CASES(0)
  if (MODE == MODE_GAT) {
    DGNN_FWD_CASES(1) DGNN_FWD_CASES(2) DGNN_FWD_CASES(8)
    DGNN_FWD_CASE(4, 1, 4, true) DGNN_FWD_CASE(4, 1, 4, false)
  }
#undef DGNN_FWD_CASES
#undef DGNN_FWD_CASE
  if (MODE == MODE_GAT) timer_mark(0, 1, st);
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

int launch_forward(int m, int nnz, int h, int f, float attn_drop, float slope,
                   const float *attn_row, const float *attn_col, const int *row_ptr,
                   const int *col_ind, const float *in_feat, const float *edge_mask,
                   float *edge_max, float *edge_sum, float *out_feat, void *workspace,
                   size_t workspace_bytes, hipStream_t st, DropRng rng = DropRng{0, 0, 0, nullptr},
                   float *out_neg = nullptr, float *a_neg = nullptr) {
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
    P.out_neg = out_neg; P.a_neg = a_neg;
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

// ----------------------------------------------------------------------------- backward side
struct BwdLayout {
  bool fast;
  int L, T, slot_stride;
  size_t off_S, off_dE, off_long, off_stat, off_cols, off_slots, off_out, off_fwd, total;
  int max_long;
};

// Rows longer than this are summed by one block each (k_gat_row_sum4_long).  Tuning override:
// DGNN_ROWSUM_LONG (read once per process: the workspace layout depends on it).
int rowsum_long_threshold() {
  static const int v = [] {
    const int e = env_int("DGNN_ROWSUM_LONG", 256);
    return (e >= 32 && e <= (1 << 20)) ? e : 256;
  }();
  return v;
}

// lanes per head of the backward's lane map (gat_bwd_fast.hpp: bwd_lane): power of two covering
// ceil(f / vec) lanes
int bwd_lanes_per_head(int f, int vec) {
  const int need = (f + vec - 1) / vec;
  int g = 1;
  while (g < need) g <<= 1;
  return g;
}
int bwd_heads_padded(int h) { return h <= 1 ? 1 : h <= 2 ? 2 : h <= 4 ? 4 : 8; }
// Packed stream of the backward (gat_bwd_fast.hpp): rows whose lane map covers at most 32 lanes
// (h heads x fv lanes) -> lanes per row = the power of two covering them, at least 4; 0 = off
// (DGNN_BWD_PACK=0: A/B runs).  Only the STEADY (narrow, float4) instantiation has it.
int bwd_pack_lanes(bool steady, int h, int fv) {
  if (!steady || h * fv > env_int("DGNN_PACK_MAX_S", 16) || env_int("DGNN_BWD_PACK", 1) == 0) return 0;
  int s = 4;
  while (s < h * fv) s <<= 1;
  return s;
}
// floats per lane of the backward's lane map: float4 lanes for every width of at least four floats
// (rows need no 16-byte alignment: common.hpp ld4u), scalar lanes below
int bwd_vec(int f) { return f >= 4 ? 4 : 1; }
bool bwd_fast_shape(int m, int n, int h, int f) {
  if (force_generic() || h > 8 || m >= (1 << 30) || n >= (1 << 30)) return false;
  if (f > 256) return false;
  return bwd_lanes_per_head(f, bwd_vec(f)) <= DGNN_WAVE;   // one head fits one lane pass
}

BwdLayout bwd_layout(int m, int n, int nnz, int h, int f, bool need_recompute) {
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
    w.L = bwd_group_size(bwd_heads_padded(h), nnz, narrow_rows(h, f));
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

// One-pass CSC gather kernel of the padded head count hc, group size P.L and lane width vec.
template <int BMODE>
void launch_bwd_gather(const BwdParams &P, int hc, int vec, dim3 grid, dim3 block, size_t lds,
                       hipStream_t st) {
#define DGNN_BWD_CASE(HC, EPL)                                                                    \
  if (hc == HC && P.L == 64 * EPL) {                                                              \
    if (BMODE == BWD_GAT && timing_on())                                                          \
      std::snprintf(g_timer.name[1], sizeof(g_timer.name[1]), "k_bwd_col_gather<%d, %d, %d, %d, %s>", HC, EPL,  \
                    vec == 4 ? 4 : 1, BMODE, (vec == 4 && P.steady) ? "true" : "false");          \
    if (vec == 4 && P.steady)                                                                     \
      hipLaunchKernelGGL((k_bwd_col_gather<HC, EPL, 4, BMODE, true>), grid, block, lds, st, P);   \
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

int check_dims(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if ((long long)h * f > (1 << 20)) return DGNN_ERR_UNSUPPORTED;
  return DGNN_OK;
}

}  // namespace

extern "C" {

int dgnn_debug_last_kernel_ms(int which, float *ms) {
  if (which < 0 || which > 1 || !ms) return DGNN_ERR_BAD_ARG;
  if (!timing_on() || !g_timer.valid[which]) return DGNN_ERR_UNSUPPORTED;
  hipError_t e = hipEventSynchronize(g_timer.ev[which][1]);
  if (e != hipSuccess) return (int)e;
  e = hipEventElapsedTime(ms, g_timer.ev[which][0], g_timer.ev[which][1]);
  return e == hipSuccess ? DGNN_OK : (int)e;
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

// ------------------------------------------------------------------------- GE-SpMM / GMM
size_t dgnn_spmm_csr_workspace_bytes(int m, int nnz, int h, int f) {
  if (m < 0 || nnz < 0 || h <= 0 || f <= 0) return 0;
  return fwd_layout(nnz, h, f).total;
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
    launch_bwd_gather<BWD_GMM>(P, hc, vec, grid, block, bwd_lds_bytes(hc, P.L, P.hreg != 0), st);
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

size_t dgnn_gmm_forward_workspace_bytes(int nodes, int nnz, int kernels, int embed) {
  if (nodes < 0 || nnz < 0 || kernels <= 0 || embed <= 0) return 0;
  return fwd_layout(nnz, kernels, embed).total;
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

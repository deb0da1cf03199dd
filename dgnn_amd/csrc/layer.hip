// The GAT layer's steps on either side of the fused operator, each as ONE pass over the
// [N,H,F] tensor instead of the chain of elementwise torch kernels the reference layer runs
// (SURVEY.md 8(f) rank 2):
//   * attention score terms  attn_row[n,h] = <attn_l[h,:], feat[n,h,:]>, attn_col likewise with
//     attn_r (dgNN/layers/gatconv_layer.py:74-75: two multiplies and two reductions, four
//     passes plus two [N,H,F] temporaries) and their backward;
//   * the epilogue  act(rst + residual + bias) (gatconv_layer.py:82-90) and its backward.
// All reductions are fixed-order: results do not vary from run to run.
#include "../../include/dgnn_hip.h"
#include "common.hpp"

namespace dgnn {
namespace {

constexpr int LAYER_WAVES = 8192;     // waves of the score kernels (partials are per wave)
constexpr int EPI_BLOCKS = 2048;      // blocks of the epilogue backward (partials per thread)

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
  using T = float4;
  static __device__ __forceinline__ T load(const float *p) { return *(const float4 *)p; }
  static __device__ __forceinline__ void store(float *p, const T &v) { *(float4 *)p = v; }
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ float dot(const T &a, const T &b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
  }
  static __device__ __forceinline__ void axpy(T &y, float a, const T &x) {
    y.x = fmaf(a, x.x, y.x); y.y = fmaf(a, x.y, y.y); y.z = fmaf(a, x.z, y.z); y.w = fmaf(a, x.w, y.w);
  }
};
template <>
struct Vec<1> {
  using T = float;
  static __device__ __forceinline__ T load(const float *p) { return *p; }
  static __device__ __forceinline__ void store(float *p, const T &v) { *p = v; }
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ float dot(const T &a, const T &b) { return a * b; }
  static __device__ __forceinline__ void axpy(T &y, float a, const T &x) { y = fmaf(a, x, y); }
};

// Segment = one (node, head) pair = F contiguous floats; G lanes (power of two) cover it with
// one VEC-wide element each (F <= VEC*G), 64/G segments per wave pass.  The number of waves
// times 64/G is a multiple of H, so a lane meets the same head in every pass and keeps its
// slice of attn_l / attn_r in registers.
struct ScoreShape {
  int vec, G, spw, nw;
};
static int score_waves(long long segs, int spw, int h) {
  // at least ~4 passes per wave; a multiple of h so that a lane keeps its head; never more than
  // the partial buffer holds (LAYER_WAVES + h waves)
  long long want = (segs + (long long)spw * 4 - 1) / ((long long)spw * 4);
  if (want > LAYER_WAVES) want = LAYER_WAVES;
  if (want < 1) want = 1;
  return (int)((want + h - 1) / h) * h;
}
static bool score_shape(long long segs, int h, int f, const void *a, const void *b, const void *c,
                        ScoreShape *s) {
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  s->vec = (f % 4 == 0 && al16(a) && al16(b) && al16(c)) ? 4 : 1;
  const int slots = f / s->vec;
  if (slots > 64) return false;
  s->G = 1;
  while (s->G < slots) s->G <<= 1;
  s->spw = 64 / s->G;
  s->nw = score_waves(segs, s->spw, h);
  return true;
}

template <int VEC>
__global__ __launch_bounds__(DGNN_BLOCK) void k_attn_scores(long long segs, int H, int F, int G, int nw,
                                                            const float *__restrict__ feat,
                                                            const float *__restrict__ attn_l,
                                                            const float *__restrict__ attn_r,
                                                            float *__restrict__ attn_row,
                                                            float *__restrict__ attn_col) {
  using V = Vec<VEC>;
  const int w = blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block();
  if (w >= nw) return;
  const int lane = lane_id(), spw = 64 / G, sub = lane / G, q = lane % G;
  const bool live = q * VEC < F;
  const int hh = (int)(((long long)w * spw + sub) % H);
  typename V::T al = V::zero(), ar = V::zero();
  if (live) {
    al = V::load(attn_l + (size_t)hh * F + q * VEC);
    ar = V::load(attn_r + (size_t)hh * F + q * VEC);
  }
  for (long long s0 = (long long)w * spw; s0 < segs; s0 += (long long)nw * spw) {
    const long long seg = s0 + sub;
    float pl = 0.f, pr = 0.f;
    if (live && seg < segs) {
      const typename V::T x = V::load(feat + (size_t)seg * F + q * VEC);
      pl = V::dot(al, x);
      pr = V::dot(ar, x);
    }
    pl = group_reduce(pl, G, OpSum());
    pr = group_reduce(pr, G, OpSum());
    if (q == 0 && seg < segs) {
      attn_row[seg] = pl;
      attn_col[seg] = pr;
    }
  }
}

// grad_feat[n,h,:] += g_row[n,h]*attn_l[h,:] + g_col[n,h]*attn_r[h,:]   (in place)
// partial[w][lane] = this lane's share of d attn_l / d attn_r
template <int VEC>
__global__ __launch_bounds__(DGNN_BLOCK) void k_attn_scores_bwd(long long segs, int H, int F, int G, int nw,
                                                                const float *__restrict__ feat,
                                                                const float *__restrict__ attn_l,
                                                                const float *__restrict__ attn_r,
                                                                const float *__restrict__ g_row,
                                                                const float *__restrict__ g_col,
                                                                float *__restrict__ grad_feat,
                                                                float *__restrict__ partial) {
  using V = Vec<VEC>;
  const int w = blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block();
  if (w >= nw) return;
  const int lane = lane_id(), spw = 64 / G, sub = lane / G, q = lane % G;
  const bool live = q * VEC < F;
  const int hh = (int)(((long long)w * spw + sub) % H);
  typename V::T al = V::zero(), ar = V::zero(), dl = V::zero(), dr = V::zero();
  if (live) {
    al = V::load(attn_l + (size_t)hh * F + q * VEC);
    ar = V::load(attn_r + (size_t)hh * F + q * VEC);
  }
  for (long long s0 = (long long)w * spw; s0 < segs; s0 += (long long)nw * spw) {
    const long long seg = s0 + sub;
    if (live && seg < segs) {
      const float gr = g_row[seg], gc = g_col[seg];
      const size_t off = (size_t)seg * F + q * VEC;
      const typename V::T x = V::load(feat + off);
      typename V::T gf = V::load(grad_feat + off);
      V::axpy(gf, gr, al);
      V::axpy(gf, gc, ar);
      V::store(grad_feat + off, gf);
      V::axpy(dl, gr, x);
      V::axpy(dr, gc, x);
    }
  }
  float *p = partial + ((size_t)w * 64 + lane) * 2 * VEC;
  V::store(p, dl);
  V::store(p + VEC, dr);
}

// d attn_l[h,f], d attn_r[h,f]: sum of the lane partials that belong to (h, f).  One wave per
// output element: lanes stride over the (wave, sub-segment) slots in a fixed pattern, then a
// fixed butterfly -- the same bits every run.
__global__ __launch_bounds__(DGNN_BLOCK) void k_attn_scores_bwd_reduce(int H, int F, int G, int nw, int vec,
                                                                       const float *__restrict__ partial,
                                                                       float *__restrict__ grad_attn_l,
                                                                       float *__restrict__ grad_attn_r) {
  const int o = blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block();
  if (o >= 2 * H * F) return;                        // wave-uniform exit
  const int which = o / (H * F), hf = o % (H * F), h = hf / F, f = hf % F;
  const int q = f / vec, comp = f % vec, spw = 64 / G;
  const int slots = nw * spw;
  float acc = 0.f;
  for (int idx = lane_id(); idx < slots; idx += DGNN_WAVE) {
    if (idx % H != h) continue;                      // slot idx = w*spw + sub serves head idx % H
    const int w = idx / spw, sub = idx - w * spw;
    acc += partial[(((size_t)w * 64 + sub * G + q) * 2 + which) * vec + comp];
  }
  acc = wave_sum(acc);
  if (lane_id() == 0) (which == 0 ? grad_attn_l : grad_attn_r)[hf] = acc;
}

// ------------------------------------------------------------------------------- epilogue
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_LEAKY = 3 };

__device__ __forceinline__ float act_fwd(float x, int act, float a) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_ELU: return x > 0.f ? x : a * expm1f(x);
    case ACT_LEAKY: return x > 0.f ? x : a * x;
    default: return x;
  }
}
// derivative written in terms of the OUTPUT y (what the backward has at hand)
__device__ __forceinline__ float act_bwd(float y, int act, float a) {
  switch (act) {
    case ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case ACT_ELU: return y > 0.f ? 1.f : y + a;          // a*exp(x) = y + a
    case ACT_LEAKY: return y > 0.f ? 1.f : a;
    default: return 1.f;
  }
}

__global__ __launch_bounds__(DGNN_BLOCK) void k_epilogue(long long total, int hf, int act, float a,
                                                         const float *__restrict__ x,
                                                         const float *__restrict__ res,
                                                         const float *__restrict__ bias,
                                                         float *__restrict__ y) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (res) v += res[i];
    if (bias) v += bias[i % hf];
    y[i] = act_fwd(v, act, a);
  }
}
__global__ __launch_bounds__(DGNN_BLOCK) void k_epilogue4(long long total4, int hf4, int act, float a,
                                                          const float4 *__restrict__ x,
                                                          const float4 *__restrict__ res,
                                                          const float4 *__restrict__ bias,
                                                          float4 *__restrict__ y) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    float4 v = x[i];
    if (res) { const float4 r = res[i]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
    if (bias) { const float4 b = bias[i % hf4]; v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    y[i] = make_float4(act_fwd(v.x, act, a), act_fwd(v.y, act, a), act_fwd(v.z, act, a), act_fwd(v.w, act, a));
  }
}

// grad_pre = grad * act'(y); every live thread walks ONE column slot of VEC columns (its stride
// is a multiple of hf/VEC) and leaves its column sums in partial[tid*VEC ...].
template <int VEC>
__global__ __launch_bounds__(DGNN_BLOCK) void k_epilogue_bwd(long long totalv, long long tt, int act, float a,
                                                             const float *__restrict__ y,
                                                             const float *__restrict__ grad,
                                                             float *__restrict__ grad_pre,
                                                             float *__restrict__ partial) {
  using V = Vec<VEC>;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (tid >= tt) return;
  float acc[VEC];
#pragma unroll
  for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
  for (long long i = tid; i < totalv; i += tt) {
    float g[VEC], o[VEC];
    *(typename V::T *)g = V::load(grad + i * VEC);
    *(typename V::T *)o = V::load(y + i * VEC);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      g[c] *= act_bwd(o[c], act, a);
      acc[c] += g[c];
    }
    V::store(grad_pre + i * VEC, *(typename V::T *)g);
  }
  if (partial) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) partial[tid * VEC + c] = acc[c];
  }
}
// one wave per bias column: the partials of the threads that walked the column's slot, in a
// fixed lane pattern, then a fixed butterfly
__global__ __launch_bounds__(DGNN_BLOCK) void k_epilogue_bwd_reduce(int hf, int vec, long long tt,
                                                                    const float *__restrict__ partial,
                                                                    float *__restrict__ grad_bias) {
  const int c = blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block();
  if (c >= hf) return;                               // wave-uniform exit
  const int hfv = hf / vec, comp = c % vec;
  float acc = 0.f;
  for (long long t = c / vec + (long long)lane_id() * hfv; t < tt; t += (long long)DGNN_WAVE * hfv)
    acc += partial[t * vec + comp];
  acc = wave_sum(acc);
  if (lane_id() == 0) grad_bias[c] = acc;
}

// threads of k_epilogue_bwd: a multiple of the column slots hfv = hf/vec (a thread keeps its
// slot), ~8 rows per thread, at most EPI_BLOCKS blocks worth
static long long epi_threads(int n, int hfv) {
  const long long cap_raw = (long long)EPI_BLOCKS * DGNN_BLOCK;
  const long long cap = hfv > cap_raw ? hfv : cap_raw / hfv * hfv;
  long long rows = ((long long)n + 7) / 8;
  if (rows < 1) rows = 1;
  const long long tt = rows * hfv;
  return tt < cap ? tt : cap;
}

}  // namespace
}  // namespace dgnn

using namespace dgnn;

extern "C" {

int dgnn_gat_attn_scores(int n, int h, int f, const float *feat, const float *attn_l,
                         const float *attn_r, float *attn_row, float *attn_col, void *stream) {
  if (n < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if (n == 0) return DGNN_OK;
  if (!feat || !attn_l || !attn_r || !attn_row || !attn_col) return DGNN_ERR_BAD_ARG;
  ScoreShape s;
  const long long segs = (long long)n * h;
  if (!score_shape(segs, h, f, feat, attn_l, attn_r, &s)) return DGNN_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks_for_waves(s.nw)), block(DGNN_BLOCK);
  if (s.vec == 4)
    hipLaunchKernelGGL(k_attn_scores<4>, grid, block, 0, st, segs, h, f, s.G, s.nw, feat, attn_l, attn_r,
                       attn_row, attn_col);
  else
    hipLaunchKernelGGL(k_attn_scores<1>, grid, block, 0, st, segs, h, f, s.G, s.nw, feat, attn_l, attn_r,
                       attn_row, attn_col);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

size_t dgnn_gat_attn_scores_backward_workspace_bytes(int n, int h, int f) {
  if (n < 0 || h <= 0 || f <= 0) return 0;
  return sizeof(float) * ((size_t)LAYER_WAVES + h) * 64 * 2 * 4;    // upper bound of score_waves()
}

int dgnn_gat_attn_scores_backward(int n, int h, int f, const float *feat, const float *attn_l,
                                  const float *attn_r, const float *g_row, const float *g_col,
                                  float *grad_feat, float *grad_attn_l, float *grad_attn_r,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  if (n < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if (!attn_l || !attn_r || !grad_attn_l || !grad_attn_r) return DGNN_ERR_BAD_ARG;
  if (n > 0 && (!feat || !g_row || !g_col || !grad_feat)) return DGNN_ERR_BAD_ARG;
  ScoreShape s;
  const long long segs = (long long)n * h;
  // a misaligned grad_feat takes the scalar layout (pass it as the alignment witness)
  if (!score_shape(segs, h, f, ((uintptr_t)grad_feat & 15) ? (const void *)1 : feat, attn_l, attn_r, &s))
    return DGNN_ERR_UNSUPPORTED;
  if (workspace_bytes < dgnn_gat_attn_scores_backward_workspace_bytes(n, h, f) || !workspace)
    return DGNN_ERR_WORKSPACE;
  if ((uintptr_t)workspace & 15) return DGNN_ERR_ALIGNMENT;
  hipStream_t st = (hipStream_t)stream;
  float *partial = (float *)workspace;
  const dim3 grid((unsigned)blocks_for_waves(s.nw)), block(DGNN_BLOCK);
  if (s.vec == 4)
    hipLaunchKernelGGL(k_attn_scores_bwd<4>, grid, block, 0, st, segs, h, f, s.G, s.nw, feat, attn_l, attn_r,
                       g_row, g_col, grad_feat, partial);
  else
    hipLaunchKernelGGL(k_attn_scores_bwd<1>, grid, block, 0, st, segs, h, f, s.G, s.nw, feat, attn_l, attn_r,
                       g_row, g_col, grad_feat, partial);
  DGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attn_scores_bwd_reduce, dim3((unsigned)blocks_for_waves(2 * h * f)), dim3(DGNN_BLOCK), 0, st, h, f, s.G,
                     s.nw, s.vec, (const float *)partial, grad_attn_l, grad_attn_r);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_gat_epilogue(int n, int hf, int activation, float act_param, const float *rst,
                      const float *residual, const float *bias, float *out, void *stream) {
  if (n < 0 || hf <= 0 || activation < 0 || activation > 3) return DGNN_ERR_BAD_ARG;
  if (n == 0) return DGNN_OK;
  if (!rst || !out) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)n * hf;
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  if (hf % 4 == 0 && al16(rst) && al16(out) && al16(residual) && al16(bias)) {
    const long long t4 = total / 4;
    const unsigned blocks = (unsigned)((t4 + DGNN_BLOCK - 1) / DGNN_BLOCK < 16384 ? (t4 + DGNN_BLOCK - 1) / DGNN_BLOCK : 16384);
    hipLaunchKernelGGL(k_epilogue4, dim3(blocks), dim3(DGNN_BLOCK), 0, st, t4, hf / 4, activation, act_param,
                       (const float4 *)rst, (const float4 *)residual, (const float4 *)bias, (float4 *)out);
  } else {
    const unsigned blocks = (unsigned)((total + DGNN_BLOCK - 1) / DGNN_BLOCK < 16384 ? (total + DGNN_BLOCK - 1) / DGNN_BLOCK : 16384);
    hipLaunchKernelGGL(k_epilogue, dim3(blocks), dim3(DGNN_BLOCK), 0, st, total, hf, activation, act_param,
                       rst, residual, bias, out);
  }
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

size_t dgnn_gat_epilogue_backward_workspace_bytes(int n, int hf) {
  if (n < 0 || hf <= 0) return 0;
  const size_t cap = (size_t)EPI_BLOCKS * DGNN_BLOCK;      // thread slots, up to 4 floats each
  return sizeof(float) * 4 * (cap > (size_t)hf ? cap : (size_t)hf);
}

int dgnn_gat_epilogue_backward(int n, int hf, int activation, float act_param, const float *out,
                               const float *grad, float *grad_pre, float *grad_bias,
                               void *workspace, size_t workspace_bytes, void *stream) {
  if (n < 0 || hf <= 0 || activation < 0 || activation > 3) return DGNN_ERR_BAD_ARG;
  if (n > 0 && (!grad || !grad_pre || (activation != 0 && !out))) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (grad_bias) {
      hipError_t e = hipMemsetAsync(grad_bias, 0, sizeof(float) * (size_t)hf, st);
      if (e != hipSuccess) return (int)e;
    }
    return DGNN_OK;
  }
  auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
  const int vec = (hf % 4 == 0 && al16(out) && al16(grad) && al16(grad_pre)) ? 4 : 1;
  const long long tt = epi_threads(n, hf / vec);
  if (grad_bias && (!workspace || workspace_bytes < dgnn_gat_epilogue_backward_workspace_bytes(n, hf)))
    return DGNN_ERR_WORKSPACE;
  const long long totalv = (long long)n * hf / vec;
  const float *y = activation != 0 ? out : grad;      // unused by act_bwd when activation == 0
  const dim3 grid((unsigned)((tt + DGNN_BLOCK - 1) / DGNN_BLOCK)), block(DGNN_BLOCK);
  float *partial = grad_bias ? (float *)workspace : nullptr;
  if (vec == 4)
    hipLaunchKernelGGL(k_epilogue_bwd<4>, grid, block, 0, st, totalv, tt, activation, act_param, y, grad,
                       grad_pre, partial);
  else
    hipLaunchKernelGGL(k_epilogue_bwd<1>, grid, block, 0, st, totalv, tt, activation, act_param, y, grad,
                       grad_pre, partial);
  DGNN_LAUNCH_CHECK();
  if (grad_bias) {
    hipLaunchKernelGGL(k_epilogue_bwd_reduce, dim3((unsigned)blocks_for_waves(hf)), dim3(DGNN_BLOCK), 0, st, hf,
                       vec, tt, (const float *)workspace, grad_bias);
    DGNN_LAUNCH_CHECK();
  }
  return DGNN_OK;
}

}  // extern "C"

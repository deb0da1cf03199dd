// Multi-head SDDMM, edge permute, Gaussian kernels and fused EdgeConv:
// kernels + C-ABI entry points (include/dgnn_hip.h).
#include <cstdlib>

#include "common.hpp"

using namespace dgnn;

namespace dgnn {

// --------------------------------------------------------------------------- SDDMM
// out[e,h] = <A[row(e),h,:], B[col(e),h,:]>  (mhsddmm_kernel.cu:5-81).  One wave per row, so
// the row operand is read once and no binary search (findRow, computeUtil.h:38-54) is needed.
__global__ __launch_bounds__(DGNN_BLOCK) void k_sddmm_row(int v, int H, int F,
                                                          const int *__restrict__ row_ptr,
                                                          const int *__restrict__ col_ind,
                                                          const float *__restrict__ A,
                                                          const float *__restrict__ B,
                                                          float *__restrict__ out) {
  const int r = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (r >= v) return;
  const int lane = lane_id();
  const int lb = row_ptr[r], hb = row_ptr[r + 1];
  const size_t HF = (size_t)H * F;
  for (int p = lb; p < hb; ++p) {
    const int c = uniform_i32(col_ind[p]);
    for (int h = 0; h < H; ++h) {
      const float *a = A + (size_t)r * HF + (size_t)h * F;
      const float *b = B + (size_t)c * HF + (size_t)h * F;
      float s = 0.f;
      for (int q = lane; q < F; q += DGNN_WAVE) s = fmaf(a[q], b[q], s);
      s = wave_sum(s);
      if (lane == 0) out[(size_t)p * H + h] = s;
    }
  }
}

// Neighbour-grouped SDDMM (fast path): one wave64 per group of DGNN_SDDMM_L consecutive edges,
// whatever rows it spans.  The row operand A[r] rides the same software-pipelined load stream
// as the gathered B rows (a "U item" in front of the row's edges), each a fully coalesced
// load; the dot product reduces inside the power-of-two lane group of a head with DPP row ops
// (lane map and head passes as in the fused backward, gat_bwd_fast.hpp: any H <= 8 and F).
// No binary search per edge (findRow, computeUtil.h:38-54): the group finds its first row once
// with a 64-ary cooperative search.  Per-edge outputs are staged in LDS and stored coalesced.
#define DGNN_SDDMM_L 256
#define DGNN_SDDMM_U 4
#define DGNN_SDDMM_UFLAG 0x80000000u
template <int VEC>
struct SdVec;
template <>
struct SdVec<4> {
  using T = float4;
  static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  static __device__ __forceinline__ float dot(const T &x, const T &u) {
    return fmaf(x.x, u.x, fmaf(x.y, u.y, fmaf(x.z, u.z, x.w * u.w)));
  }
  // rows of any width >= 4 floats at any dword alignment; a head's last slot may repeat floats of
  // its predecessor, which the ROW operand drops before the dot (common.hpp: ld4u, slot4_*)
  static __device__ __forceinline__ T load(const float *p) { return ld4u(p); }
  static __device__ __forceinline__ int off(int q, int F) { return slot4_off(q, F); }
  static __device__ __forceinline__ T row_operand(const T &v, int q, int F) { return slot4_mask(v, slot4_dup(q, F)); }
};
template <>
struct SdVec<1> {
  using T = float;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ float dot(const T &x, const T &u) { return x * u; }
  static __device__ __forceinline__ T load(const float *p) { return *p; }
  static __device__ __forceinline__ int off(int q, int) { return q; }
  static __device__ __forceinline__ T row_operand(const T &v, int, int) { return v; }
};

// NS = head slices per lane: one walk of the stream serves hpp*NS heads.
template <int VEC, int NS>
__global__ __launch_bounds__(DGNN_BLOCK) void k_sddmm_fast(int v, int nnz, int H, int F, int HF,
                                                           int hpp, int fv,
                                                           const int *__restrict__ row_ptr,
                                                           const int *__restrict__ col_ind,
                                                           const float *__restrict__ A,
                                                           const float *__restrict__ B,
                                                           float *__restrict__ out) {
  extern __shared__ __align__(16) float lds_raw[];
  constexpr int L = DGNN_SDDMM_L;
  const int wib = wave_in_block();
  const int lane = lane_id();
  const int T = (nnz + L - 1) / L;
  const int t = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wib);
  if (t >= T) return;
  const size_t per_wave = (size_t)L * 3 + (size_t)L * H;
  unsigned *stream = (unsigned *)(lds_raw + (size_t)wib * per_wave);   // [2L]
  int *kof = (int *)(stream + 2 * L);                                   // [L] row ordinal of edge p
  float *dbuf = (float *)(kof + L);                                     // [L][H]
  const int e0 = t * L;
  const int e1 = min(e0 + L, nnz);
  const int cnt = e1 - e0;

  // first row of the group: largest r with row_ptr[r] <= e0 (64-ary search, ~4 rounds)
  int lo = 0, hi = v;
  while (hi - lo > 1) {
    const int step = (hi - lo + DGNN_WAVE - 1) / DGNN_WAVE;
    const long long idx = (long long)lo + (long long)(lane + 1) * step;
    const bool ok = (idx < hi) && (row_ptr[idx] <= e0);
    const int c = __popcll(__ballot(ok));
    const int nlo = lo + c * step;
    const long long nhi = (long long)lo + (long long)(c + 1) * step;
    lo = nlo;
    hi = (int)(nhi < hi ? nhi : hi);
  }
  // rows of the group, one lane per row
  int K = 0;
  {
    int rbase = lo;
    while (true) {
      const int r = rbase + lane;
      int rs = 0x7fffffff, re = 0x7fffffff;
      if (r < v) {
        rs = row_ptr[r];
        re = row_ptr[r + 1];
      }
      const int a = max(rs, e0) - e0;
      const int b = (re < e1 ? re : e1) - e0;
      const bool nonempty = (r < v) && (rs < e1) && (b > a);
      const unsigned long long bal = __ballot(nonempty);
      const int k = K + __popcll(bal & ((1ull << lane) - 1ull));
      if (nonempty) {
        stream[a + k] = (unsigned)r | DGNN_SDDMM_UFLAG;
        for (int p = a; p < b; ++p) kof[p] = k;
      }
      K += __popcll(bal);
      const bool reached_end = (r >= v - 1) || (re >= e1);
      if (__ballot(reached_end) != 0ull) break;
      rbase += DGNN_WAVE;
    }
  }
  wave_lds_sync();
  for (int p = lane; p < cnt; p += DGNN_WAVE) stream[p + kof[p] + 1] = (unsigned)col_ind[e0 + p];
  wave_lds_sync();

  using V = SdVec<VEC>;
  using vec_t = typename V::T;
  const int items = cnt + K;
  for (int hb = 0; hb < H; hb += hpp * NS) {      // one stream pass per block of heads
    bool active[NS], head_leader[NS];
    int head[NS];
    size_t lane_off[NS];
    vec_t u[NS];
    const int q = lane % fv;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int hd = hb + s * hpp + lane / fv;
      active[s] = hd < H && q * VEC < F;
      head[s] = active[s] ? hd : 0;
      lane_off[s] = active[s] ? (size_t)hd * F + (size_t)V::off(q, F) : 0;
      head_leader[s] = active[s] && q == 0;
      u[s] = V::zero();
    }
    int k = -1;
    vec_t ring[DGNN_SDDMM_U][NS];
#pragma unroll
    for (int i = 0; i < DGNN_SDDMM_U; ++i) {
#pragma unroll
      for (int s = 0; s < NS; ++s) ring[i][s] = V::zero();
      if (i < items) {
        const unsigned sv = (unsigned)uniform_i32((int)stream[i]);
        const float *base = ((sv & DGNN_SDDMM_UFLAG) ? A : B) + (size_t)(sv & 0x7fffffffu) * HF;
#pragma unroll
        for (int s = 0; s < NS; ++s)
          if (active[s]) ring[i][s] = V::load(base + lane_off[s]);
      }
    }
    for (int j = 0; j < items; j += DGNN_SDDMM_U) {
#pragma unroll
      for (int i = 0; i < DGNN_SDDMM_U; ++i) {
        const int it = j + i;
        if (it < items) {
          vec_t x[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) x[s] = ring[i][s];
          const unsigned sv = (unsigned)uniform_i32((int)stream[it]);
          if (it + DGNN_SDDMM_U < items) {
            const unsigned nv = (unsigned)uniform_i32((int)stream[it + DGNN_SDDMM_U]);
            const float *base = ((nv & DGNN_SDDMM_UFLAG) ? A : B) + (size_t)(nv & 0x7fffffffu) * HF;
#pragma unroll
            for (int s = 0; s < NS; ++s)
              if (active[s]) ring[i][s] = V::load(base + lane_off[s]);
          }
          if (sv & DGNN_SDDMM_UFLAG) {
#pragma unroll
            for (int s = 0; s < NS; ++s) u[s] = V::row_operand(x[s], q, F);
            ++k;
          } else {
            const int p = it - (k + 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
              float dot = active[s] ? V::dot(x[s], u[s]) : 0.f;
              dot = group_reduce(dot, fv, OpSum());
              if (head_leader[s]) dbuf[(size_t)p * H + head[s]] = dot;
            }
          }
        }
      }
    }
  }
  wave_lds_sync();
  for (int i = lane; i < cnt * H; i += DGNN_WAVE) out[(size_t)e0 * H + i] = dbuf[i];
}

// Narrow rows (a padded row of at most 16 float4 lanes: 1 x 16, 1 x 64, 4 x 16, 8 x 8, 1 x 47 ...): the walk above
// handles ONE edge per load step, so 48 .. 60 of the 64 lanes idle and the kernel is bound by instructions per
// edge (1 x 16 on the products shape: 3.4 ms against 0.95 ms for the SpMM of the same rows).  Here the wave's
// lanes form G = 64 / S groups of S lanes (S = padded heads x lanes per head, at least 4); group g takes edge
// p0 + g of a step and loads its own slice of BOTH operand rows (consecutive edges share the A row: those
// loads hit the same lines), U = 4 steps in flight.  Same per-lane FMA chain and the same DPP reduction inside
// a head's lanes as k_sddmm_fast: identical bits.
template <int S>
__global__ __launch_bounds__(DGNN_BLOCK) void k_sddmm_packed(int v, int nnz, int H, int F, int HF, int fv,
                                                             const int *__restrict__ row_ptr,
                                                             const int *__restrict__ col_ind,
                                                             const float *__restrict__ A,
                                                             const float *__restrict__ B,
                                                             float *__restrict__ out) {
  extern __shared__ __align__(16) float lds_raw[];
  constexpr int L = DGNN_SDDMM_L;
  constexpr int G = DGNN_WAVE / S;
  constexpr int U = DGNN_SDDMM_U;
  const int wib = wave_in_block();
  const int lane = lane_id();
  const int T = (nnz + L - 1) / L;
  const int t = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wib);
  if (t >= T) return;
  const size_t per_wave = (size_t)L * 2 + (size_t)L * H;
  int *rowE = (int *)(lds_raw + (size_t)wib * per_wave);   // [L] row of edge p
  int *colE = rowE + L;                                    // [L] column of edge p
  float *dbuf = (float *)(colE + L);                       // [L][H]
  const int e0 = t * L;
  const int e1 = min(e0 + L, nnz);
  const int cnt = e1 - e0;
  for (int p = lane; p < cnt; p += DGNN_WAVE) colE[p] = col_ind[e0 + p];
  // first row of the group: largest r with row_ptr[r] <= e0 (64-ary search, ~4 rounds)
  int lo = 0, hi = v;
  while (hi - lo > 1) {
    const int step = (hi - lo + DGNN_WAVE - 1) / DGNN_WAVE;
    const long long idx = (long long)lo + (long long)(lane + 1) * step;
    const bool ok = (idx < hi) && (row_ptr[idx] <= e0);
    const int c = __popcll(__ballot(ok));
    const int nlo = lo + c * step;
    const long long nhi = (long long)lo + (long long)(c + 1) * step;
    lo = nlo;
    hi = (int)(nhi < hi ? nhi : hi);
  }
  // rows of the group, one lane per row; a row longer than 64 edges is filled by the whole wave
  {
    int rbase = lo;
    while (true) {
      const int r = rbase + lane;
      int rs = 0x7fffffff, re = 0x7fffffff;
      if (r < v) {
        rs = row_ptr[r];
        re = row_ptr[r + 1];
      }
      const int a = max(rs, e0) - e0;
      const int b = (re < e1 ? re : e1) - e0;
      const bool nonempty = (r < v) && (rs < e1) && (b > a);
      const bool is_long = nonempty && (b - a) > DGNN_WAVE;
      if (nonempty && !is_long)
        for (int p = a; p < b; ++p) rowE[p] = r;
      unsigned long long lg = __ballot(is_long);
      while (lg != 0ull) {                                  // hub rows: every lane writes a share
        const int src = uniform_i32(__ffsll((long long)lg) - 1);
        lg &= lg - 1ull;
        const int ra = bcast_i32(a, src), rb = bcast_i32(b, src), rr = bcast_i32(r, src);
        for (int p = ra + lane; p < rb; p += DGNN_WAVE) rowE[p] = rr;
      }
      const bool reached_end = (r >= v - 1) || (re >= e1);
      if (__ballot(reached_end) != 0ull) break;
      rbase += DGNN_WAVE;
    }
  }
  wave_lds_sync();

  const int ls = lane % S, g = lane / S;
  const int hd = ls / fv, q = ls % fv;
  const bool active = hd < H && q * 4 < F;
  const size_t lane_off = active ? (size_t)hd * F + (size_t)slot4_off(q, F) : 0;
  const int dup = slot4_dup(q, F);
  const bool leader = active && q == 0;
  float4 ra_[U], rb_[U];
  auto issue = [&](int p0, float4 &xa, float4 &xb) {
    const int p = min(p0 + g, cnt - 1);
    const int r = rowE[p], c = colE[p];
    xa = ld4u(A + (size_t)r * HF + lane_off);
    xb = ld4u(B + (size_t)c * HF + lane_off);
  };
  auto consume = [&](int p0, const float4 &xa, const float4 &xb) {
    const int p = p0 + g;
    const float4 u = slot4_mask(xa, dup);
    float dot = active ? fmaf(xb.x, u.x, fmaf(xb.y, u.y, fmaf(xb.z, u.z, xb.w * u.w))) : 0.f;
    dot = group_reduce(dot, fv, OpSum());
    if (leader && p < cnt) dbuf[(size_t)p * H + hd] = dot;
  };
#pragma unroll
  for (int i = 0; i < U; ++i)
    if (i * G < cnt) issue(i * G, ra_[i], rb_[i]);
  for (int p0 = 0; p0 < cnt; p0 += U * G) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int pi = p0 + i * G;
      if (pi < cnt) {
        const float4 xa = ra_[i], xb = rb_[i];
        if (pi + U * G < cnt) issue(pi + U * G, ra_[i], rb_[i]);
        consume(pi, xa, xb);
      }
    }
  }
  wave_lds_sync();
  for (int i = lane; i < cnt * H; i += DGNN_WAVE) out[(size_t)e0 * H + i] = dbuf[i];
}

// out[i,:] = in[permute[i],:]   (mhtranspose_kernel.cu:6-26)
__global__ void k_mhtranspose(long long total, int H, const int *__restrict__ permute,
                              const float *__restrict__ in, float *__restrict__ out) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long i = t / H;
    const int h = (int)(t - i * H);
    out[t] = in[(size_t)permute[i] * H + h];
  }
}

// --------------------------------------------------------------------------- Gaussian
__global__ void k_gaussian(long long total, int K, int dim, const float *__restrict__ pseudo,
                           const float *__restrict__ mu, const float *__restrict__ inv_sigma,
                           float *__restrict__ gauss) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long e = t / K;
    const int k = (int)(t - e * K);
    gauss[t] = gmm_weight(dim, pseudo + (size_t)e * dim, mu + (size_t)k * dim,
                            inv_sigma + (size_t)k * dim);
  }
}

// Stage 1 of the Gaussian backward: each thread owns one edge; pseudo_out is written directly,
// the [K,dim] parameter gradients are reduced per block into partial[block][2][K*dim].
#define DGNN_GBP_BLOCK 256
__global__ __launch_bounds__(DGNN_GBP_BLOCK) void k_gaussian_bwd_stage1(
    int edges, int K, int dim, const float *__restrict__ pseudo, const float *__restrict__ mu,
    const float *__restrict__ inv_sigma, const float *__restrict__ grad_gauss,
    float *__restrict__ pseudo_out, float *__restrict__ partial) {
  __shared__ float red[DGNN_GBP_BLOCK / DGNN_WAVE];
  const int e = blockIdx.x * DGNN_GBP_BLOCK + threadIdx.x;
  const bool live = e < edges;
  const int KD = K * dim;
  for (int d = 0; d < dim; ++d) {
    if (live) pseudo_out[(size_t)e * dim + d] = 0.f;
  }
  for (int k = 0; k < K; ++k) {
    float g = 0.f;
    if (live)
      g = gmm_weight(dim, pseudo + (size_t)e * dim, mu + (size_t)k * dim,
                       inv_sigma + (size_t)k * dim) *
          grad_gauss[(size_t)e * K + k];
    for (int d = 0; d < dim; ++d) {
      float cm = 0.f, cs = 0.f;
      if (live) {
        const float sig = inv_sigma[k * dim + d];
        const float diff = pseudo[(size_t)e * dim + d] - mu[k * dim + d];
        const float t = g * diff * sig;
        pseudo_out[(size_t)e * dim + d] += -t * sig;
        cm = t * sig;          // d gauss / d mu
        cs = -g * diff * diff * sig;  // d gauss / d inv_sigma
      }
      for (int which = 0; which < 2; ++which) {
        float v = wave_sum(which == 0 ? cm : cs);
        if (lane_id() == 0) red[wave_in_block()] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
          float s = 0.f;
          for (int w = 0; w < DGNN_GBP_BLOCK / DGNN_WAVE; ++w) s += red[w];
          partial[((size_t)blockIdx.x * 2 + which) * KD + k * dim + d] = s;
        }
        __syncthreads();
      }
    }
  }
}

// The same stage for K <= MAXK kernels and dim <= MAXD (MoNet: K = 2..3, dim = 2), as a
// grid-stride kernel: parameters and the 2*K*dim running sums live in registers, every thread
// walks edges e = tid, tid + stride, ..., and a block reduces ONCE at the end -- the kernel above
// reduces 2*K*dim times per 256 edges and leaves one partial per 256 edges to stage 2 (Reddit
// shape, K = 3: 2.2 + 0.76 ms; this one streams the 3.2 GB once).  Per-edge arithmetic and
// pseudo_out are the same bits; the parameter sums are fixed-order too (fixed grid), only
// grouped differently.
#define DGNN_GBP_MAX_BLOCKS 2048
template <int MAXK, int MAXD>
__global__ __launch_bounds__(DGNN_GBP_BLOCK) void k_gaussian_bwd_stream(
    int edges, int K, int dim, const float *__restrict__ pseudo, const float *__restrict__ mu,
    const float *__restrict__ inv_sigma, const float *__restrict__ grad_gauss,
    float *__restrict__ pseudo_out, float *__restrict__ partial) {
  __shared__ float red[DGNN_GBP_BLOCK / DGNN_WAVE][2 * MAXK * MAXD];
  float pm[MAXK][MAXD], ps[MAXK][MAXD], am[MAXK][MAXD], as[MAXK][MAXD];
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      const bool in = k < K && d < dim;
      pm[k][d] = in ? mu[k * dim + d] : 0.f;
      ps[k][d] = in ? inv_sigma[k * dim + d] : 0.f;
      am[k][d] = 0.f;
      as[k][d] = 0.f;
    }
  const long long stride = (long long)gridDim.x * DGNN_GBP_BLOCK;
  for (long long e = blockIdx.x * (long long)DGNN_GBP_BLOCK + threadIdx.x; e < edges; e += stride) {
    float x[MAXD], po[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      x[d] = d < dim ? pseudo[(size_t)e * dim + d] : 0.f;
      po[d] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      if (k < K) {
        float acc = 0.f;                       // gmm_weight(), operand for operand
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
          if (d < dim) {
            const float tmp = x[d] - pm[k][d];
            acc = fmaf(tmp * tmp * ps[k][d], ps[k][d], acc);
          }
        const float g = (float)exp(-0.5 * (double)acc) * grad_gauss[(size_t)e * K + k];
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
          if (d < dim) {
            const float sig = ps[k][d];
            const float diff = x[d] - pm[k][d];
            const float t = g * diff * sig;
            po[d] += -t * sig;
            am[k][d] += t * sig;
            as[k][d] += -g * diff * diff * sig;
          }
      }
    }
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
      if (d < dim) pseudo_out[(size_t)e * dim + d] = po[d];
  }
#pragma unroll
  for (int k = 0; k < MAXK; ++k)
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
      const float vm = wave_sum(am[k][d]), vs = wave_sum(as[k][d]);
      if (lane_id() == 0) {
        red[wave_in_block()][k * MAXD + d] = vm;
        red[wave_in_block()][MAXK * MAXD + k * MAXD + d] = vs;
      }
    }
  __syncthreads();
  const int KD = K * dim;
  if ((int)threadIdx.x < 2 * KD) {
    const int which = threadIdx.x / KD, idx = threadIdx.x % KD;
    const int k = idx / dim, d = idx % dim;
    float sum = 0.f;
    for (int w = 0; w < DGNN_GBP_BLOCK / DGNN_WAVE; ++w) sum += red[w][which * MAXK * MAXD + k * MAXD + d];
    partial[((size_t)blockIdx.x * 2 + which) * KD + idx] = sum;
  }
}

// Stage 2: one block per (which, k*dim+d) sums the per-block partials in a fixed order.
__global__ __launch_bounds__(DGNN_GBP_BLOCK) void k_gaussian_bwd_stage2(
    int nblocks, int KD, const float *__restrict__ partial, float *__restrict__ mu_out,
    float *__restrict__ sigma_out) {
  __shared__ float red[DGNN_GBP_BLOCK];
  const int which = blockIdx.x / KD, idx = blockIdx.x % KD;
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += DGNN_GBP_BLOCK)
    s += partial[((size_t)b * 2 + which) * KD + idx];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = DGNN_GBP_BLOCK / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) (which == 0 ? mu_out : sigma_out)[idx] = red[0];
}

// --------------------------------------------------------------------------- EdgeConv
// One wave per point, one lane per feature column (coalesced h_src row reads).  The arg-max
// reproduces the reference's 32-lane xor butterfly (`tmp >= feat` takes the partner,
// fused_edgeconv_kernel.cu:31-38) in closed form: inside a 32-neighbour tile the tied
// candidate with the LARGEST BIT-REVERSED 5-bit tile position wins; padding lanes of the last
// tile carry (-1e38, -1) and, holding position 31 (bit-reversal 31), win every tie; a later
// tile replaces the running best on >= (:39).  max_val starts at -1e39 == -inf (:19).
static inline int edgeconv_xcd() {
  const char *e = getenv("DGNN_EDGECONV_XCD");
  return e ? atoi(e) : 1;
}
static inline int edgeconv_vec4() {         // DGNN_EDGECONV_VEC4=0: the one-lane-per-column kernel (A/B runs)
  const char *e = getenv("DGNN_EDGECONV_VEC4");
  return e ? atoi(e) : 1;
}
__device__ __forceinline__ int bitrev5(int x) { return (int)(__brev((unsigned)x) >> 27); }

__global__ __launch_bounds__(DGNN_BLOCK) void k_edgeconv_fwd(
    int m, int F, int k, const int *__restrict__ src_ind, const float *__restrict__ h_src,
    const float *__restrict__ h_dst, float *__restrict__ out_feat, float *__restrict__ idx_f32,
    int *__restrict__ idx_i32, int xcd) {
  // points in id order are clouds in id order, and a point's neighbours lie in its own cloud:
  // contiguous point ranges per XCD keep each cloud's rows in ONE L2 (DGNN_EDGECONV_XCD=0: launch order)
  const unsigned lb_id = xcd ? xcd_contiguous_id(blockIdx.x, gridDim.x) : blockIdx.x;
  const int r = uniform_i32((int)lb_id * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (r >= m) return;
  const int lane = lane_id();
  const size_t lb = (size_t)r * k;
  for (int fb = 0; fb < F; fb += DGNN_WAVE) {
    const int fid = fb + lane;
    const bool live = fid < F;
    const float dst_val = live ? h_dst[(size_t)r * F + fid] : 0.f;
    float max_val = -INFINITY;
    int max_id = -1;
    for (int t0 = 0; t0 < k; t0 += 32) {
      const int cnt = min(32, k - t0);
      // lanes 0..cnt-1 fetch the tile's neighbour ids once, then broadcast them
      int my_src = 0;
      if (lane < cnt) my_src = src_ind[lb + t0 + lane];
      float bv = 0.f;
      int bkey = -1, bid = -1;
      for (int q = 0; q < cnt; ++q) {
        const int s = bcast_i32(my_src, q);
        const float v = live ? dst_val + h_src[(size_t)s * F + fid] : 0.f;
        const int key = bitrev5(q);
        if (bkey < 0 || v > bv || (v == bv && key > bkey)) {
          bv = v;
          bkey = key;
          bid = s;
        }
      }
      if (cnt < 32 && -1e38f >= bv) {  // padding lanes (:24-25) sit at tile position 31
        bv = -1e38f;
        bid = -1;
      }
      if (bv >= max_val) {
        max_val = bv;
        max_id = bid;
      }
    }
    if (live) {
      out_feat[(size_t)r * F + fid] = max_val;
      idx_f32[(size_t)r * F + fid] = (float)max_id;
      if (idx_i32 != nullptr) idx_i32[(size_t)r * F + fid] = max_id;
    }
  }
}

// The same for rows of float4 slots (F % 4 == 0, 16-byte aligned tensors): a point takes LP lanes
// (LP = slots per row rounded up to a power of two, at most 64), a wave 64 / LP points, and the
// neighbour rows are loaded four at a time before the first compare.  The one-lane-per-column form
// above keeps ONE 256-byte load in flight per wave and is bound by its latency (F = 64: 20
// dependent round trips per point, 4.9 TB/s out of L2); here a wave issues 4 KiB per batch.
// Candidates are visited in the same order with the same comparisons: identical results.
struct EcBest { float bv; int bkey, bid; };
__device__ __forceinline__ void ec_take(EcBest &b, float v, int key, int s) {
  if (b.bkey < 0 || v > b.bv || (v == b.bv && key > b.bkey)) { b.bv = v; b.bkey = key; b.bid = s; }
}
__global__ __launch_bounds__(DGNN_BLOCK) void k_edgeconv_fwd4(
    int m, int F, int k, int LP, const int *__restrict__ src_ind, const float *__restrict__ h_src,
    const float *__restrict__ h_dst, float *__restrict__ out_feat, float *__restrict__ idx_f32,
    int *__restrict__ idx_i32, int xcd) {
  const unsigned lb_id = xcd ? xcd_contiguous_id(blockIdx.x, gridDim.x) : blockIdx.x;
  const int lane = lane_id();
  const int G = DGNN_WAVE / LP;
  const int gl = lane & (LP - 1), gbase = lane & ~(LP - 1);
  const long long r = ((long long)lb_id * DGNN_WAVES_PER_BLOCK + wave_in_block()) * G + lane / LP;
  const bool row_ok = r < m;
  const size_t lb = (size_t)(row_ok ? r : 0) * k;
  for (int fb = 0; fb < F; fb += 4 * LP) {
    const int fid = fb + 4 * gl;
    const bool live = row_ok && fid < F;
    const float4 dv = live ? *(const float4 *)(h_dst + (size_t)r * F + fid) : make_float4(0.f, 0.f, 0.f, 0.f);
    float mv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int mi[4] = {-1, -1, -1, -1};
    for (int t0 = 0; t0 < k; t0 += 32) {
      const int cnt = min(32, k - t0);
      // the group's lanes hold the tile's neighbour ids: position q in lane q % LP of register q / LP
      int ids0 = 0, ids1 = 0;
      if (row_ok && gl < cnt) ids0 = src_ind[lb + t0 + gl];
      if (row_ok && LP < 32 && gl + LP < cnt) ids1 = src_ind[lb + t0 + gl + LP];
      EcBest best[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { best[c].bv = 0.f; best[c].bkey = -1; best[c].bid = -1; }
      constexpr int U = 4;
      for (int q0 = 0; q0 < cnt; q0 += U) {
        int sid[U];
        float4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int q = min(q0 + u, cnt - 1);
          sid[u] = __shfl(q < LP ? ids0 : ids1, gbase | (q & (LP - 1)), DGNN_WAVE);
          x[u] = live ? *(const float4 *)(h_src + (size_t)sid[u] * F + fid) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (q0 + u < cnt) {
            const int key = bitrev5(q0 + u);
            ec_take(best[0], dv.x + x[u].x, key, sid[u]);
            ec_take(best[1], dv.y + x[u].y, key, sid[u]);
            ec_take(best[2], dv.z + x[u].z, key, sid[u]);
            ec_take(best[3], dv.w + x[u].w, key, sid[u]);
          }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (cnt < 32 && -1e38f >= best[c].bv) { best[c].bv = -1e38f; best[c].bid = -1; }
        if (best[c].bv >= mv[c]) { mv[c] = best[c].bv; mi[c] = best[c].bid; }
      }
    }
    if (live) {
      const size_t o = (size_t)r * F + fid;
      *(float4 *)(out_feat + o) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      *(float4 *)(idx_f32 + o) = make_float4((float)mi[0], (float)mi[1], (float)mi[2], (float)mi[3]);
      if (idx_i32 != nullptr) *(int4 *)(idx_i32 + o) = make_int4(mi[0], mi[1], mi[2], mi[3]);
    }
  }
}

__global__ void k_edgeconv_bwd(long long total, int F, const int *__restrict__ max_idx,
                               const float *__restrict__ grad_out, float *__restrict__ grad_src) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int s = max_idx[t];
    if (s >= 0) atomicAdd(&grad_src[(size_t)s * F + (t % F)], grad_out[t]);
  }
}

static inline int grid_1d(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 32;  // grid-stride beyond ~32 blocks per CU
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace dgnn

// ---- deterministic EdgeConv backward: a gather over the REVERSED k-NN lists -----------------
// out[i,:] = in[idx[i],:] for rows of `wv` vector slots: the pack of the rows a rank sends in the
// halo exchange (dist.py).  One thread per slot, four independent loads in flight per thread.
template <class vec_t>
__global__ __launch_bounds__(256) void k_pack_rows(long long total, int wv, const int *__restrict__ idx,
                                                   const float *__restrict__ in, float *__restrict__ out) {
  constexpr int VEC = sizeof(vec_t) / sizeof(float);
  const long long stride = (long long)gridDim.x * blockDim.x;
  constexpr int U = 4;
  for (long long e0 = blockIdx.x * (long long)blockDim.x + threadIdx.x; e0 < total; e0 += U * stride) {
    vec_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = e0 + u * stride;
      if (e < total) {
        const long long row = e / wv;
        const int q = (int)(e - row * wv);
        v[u] = *(const vec_t *)(in + ((size_t)idx[row] * wv + q) * VEC);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long e = e0 + u * stride;
      if (e < total) *(vec_t *)(out + (size_t)e * VEC) = v[u];
    }
  }
}

// grad_src[s,c] = sum over the points r that list s among their neighbours, ascending r, of
// grad_out[r,c] where max_idx[r,c] == s.  Same values as the scatter of k_edgeconv_bwd, summed in
// a fixed order and without float atomics (whose 64 lanes land in up to k different rows).
template <int VEC>
__global__ __launch_bounds__(256) void k_edgeconv_bwd_gather(long long total, int slots, int F,
                                                             const int *__restrict__ col_ptr,
                                                             const int *__restrict__ row_ind,
                                                             const int *__restrict__ max_idx,
                                                             const float *__restrict__ grad_out,
                                                             float *__restrict__ grad_src, int xcd) {
  // the rows that list s are points of s's own cloud: contiguous ranges of s per XCD (see k_edgeconv_fwd)
  const unsigned lb_id = xcd ? xcd_contiguous_id(blockIdx.x, gridDim.x) : blockIdx.x;
  for (long long t = lb_id * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
  const int s = (int)(t / slots), q = (int)(t - (long long)s * slots);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  int prev = -1;
  // four listing rows per trip: their ids first, then their eight loads, then the sums in list order
  // (one row per trip was a chain of two dependent round trips per neighbour)
  constexpr int U = 4;
  const int e1 = col_ptr[s + 1];
  for (int e = col_ptr[s]; e < e1; e += U) {
    int rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) rr[u] = row_ind[min(e + u, e1 - 1)];
    int4 id[U];
    float4 g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t at = (size_t)rr[u] * F + (size_t)q * VEC;
      if (VEC == 4) {
        id[u] = *(const int4 *)(max_idx + at);
        g[u] = *(const float4 *)(grad_out + at);
      } else {
        id[u] = make_int4(max_idx[at], -1, -1, -1);
        g[u] = make_float4(grad_out[at], 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (e + u >= e1) break;
      const bool again = rr[u] == prev;      // s listed twice by r: one contribution, as the scatter
      prev = rr[u];
      if (again) continue;
      acc[0] += id[u].x == s ? g[u].x : 0.f;
      if (VEC == 4) {
        acc[1 % VEC] += id[u].y == s ? g[u].y : 0.f;
        acc[2 % VEC] += id[u].z == s ? g[u].z : 0.f; acc[3 % VEC] += id[u].w == s ? g[u].w : 0.f;
      }
    }
  }
  const size_t o = (size_t)s * F + (size_t)q * VEC;
  if (VEC == 4)
    *(float4 *)(grad_src + o) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
  else
    grad_src[o] = acc[0];
  }
}

extern "C" {

int dgnn_abi_version(void) { return 2; }   // 2: out_neg / a_neg in the fused GAT entry points

const char *dgnn_error_string(int code) {
  switch (code) {
    case DGNN_OK: return "ok";
    case DGNN_ERR_BAD_ARG: return "bad argument (negative size or NULL tensor)";
    case DGNN_ERR_UNSUPPORTED: return "shape not supported";
    case DGNN_ERR_WORKSPACE: return "workspace missing or too small";
    case DGNN_ERR_ALIGNMENT: return "pointer not aligned";
    case DGNN_ERR_TOO_LARGE: return "problem exceeds the static scratch of the no-allocation entry point";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown dgnn error";
  }
}

int dgnn_mhsddmm(int v, int f, int h, int nnz, const int *row_ptr, const int *col_ind,
                 const float *grad, const float *feature, float *out, void *stream) {
  if (v < 0 || nnz < 0 || h <= 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if (!row_ptr || (nnz > 0 && (!col_ind || !grad || !feature || !out))) return DGNN_ERR_BAD_ARG;
  if (v == 0 || nnz == 0) return DGNN_OK;
  const int hf = h * f;
  const char *fg = getenv("DGNN_FORCE_GENERIC");
  // lanes per head: power of two covering ceil(f / vec) lanes; a head must fit one lane pass
  // float4 lanes for every width of at least four floats (rows need no 16-byte alignment: common.hpp ld4u)
  const int vec = f >= 4 ? 4 : 1;
  int fv = 1;
  while (fv < (f + vec - 1) / vec) fv <<= 1;
  const bool fast = fv <= DGNN_WAVE && h <= 8 && v < (1 << 30) && !(fg && fg[0] == '1');
  // narrow float4 rows: several edges per load step (k_sddmm_packed; DGNN_SDDMM_PACK=0: A/B runs)
  int hp = 1;
  while (hp < h) hp <<= 1;
  const int pack_s = (vec == 4 && hp * fv <= 16) ? (hp * fv < 4 ? 4 : hp * fv) : 0;
  const char *pk = getenv("DGNN_SDDMM_PACK");
  if (fast && pack_s != 0 && !(pk && pk[0] == '0')) {
    const int T = (nnz + DGNN_SDDMM_L - 1) / DGNN_SDDMM_L;
    const size_t lds = (size_t)DGNN_WAVES_PER_BLOCK * ((size_t)DGNN_SDDMM_L * 2 + (size_t)DGNN_SDDMM_L * h) * sizeof(float);
#define DGNN_SDDMM_PACKED(SV)                                                                                     \
  if (pack_s == SV)                                                                                               \
    hipLaunchKernelGGL((k_sddmm_packed<SV>), dim3(blocks_for_waves(T)), dim3(DGNN_BLOCK), lds, (hipStream_t)stream, \
                       v, nnz, h, f, hf, fv, row_ptr, col_ind, grad, feature, out);
    DGNN_SDDMM_PACKED(4) DGNN_SDDMM_PACKED(8) DGNN_SDDMM_PACKED(16)
#undef DGNN_SDDMM_PACKED
  } else if (fast) {
    const int T = (nnz + DGNN_SDDMM_L - 1) / DGNN_SDDMM_L;
    const size_t lds = (size_t)DGNN_WAVES_PER_BLOCK * ((size_t)DGNN_SDDMM_L * 3 + (size_t)DGNN_SDDMM_L * h) * sizeof(float);
    const int hpp = DGNN_WAVE / fv;
    const int passes = (h + hpp - 1) / hpp;              // lane passes needed for all heads
    const int ns = passes <= 1 ? 1 : passes <= 2 ? 2 : 4;
#define DGNN_SDDMM_CASE(VECV, NSV)                                                                   \
  if (vec == VECV && ns == NSV)                                                                      \
    hipLaunchKernelGGL((k_sddmm_fast<VECV, NSV>), dim3(blocks_for_waves(T)), dim3(DGNN_BLOCK), lds,  \
                       (hipStream_t)stream, v, nnz, h, f, hf, hpp, fv, row_ptr, col_ind, grad, feature, out);
    DGNN_SDDMM_CASE(4, 1) DGNN_SDDMM_CASE(4, 2) DGNN_SDDMM_CASE(4, 4)
    DGNN_SDDMM_CASE(1, 1) DGNN_SDDMM_CASE(1, 2) DGNN_SDDMM_CASE(1, 4)
#undef DGNN_SDDMM_CASE
  } else {
    hipLaunchKernelGGL(k_sddmm_row, dim3(blocks_for_waves(v)), dim3(DGNN_BLOCK), 0,
                       (hipStream_t)stream, v, h, f, row_ptr, col_ind, grad, feature, out);
  }
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_mhtranspose(int nnz, int h, const int *permute, const float *in, float *out,
                     void *stream) {
  if (nnz < 0 || h <= 0) return DGNN_ERR_BAD_ARG;
  if (nnz == 0) return DGNN_OK;
  if (!permute || !in || !out) return DGNN_ERR_BAD_ARG;
  const long long total = (long long)nnz * h;
  hipLaunchKernelGGL(k_mhtranspose, dim3(grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     total, h, permute, in, out);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_gaussian(int edges, int kernels, int dim, const float *pseudo, const float *mu,
                  const float *inv_sigma, float *gauss, void *stream) {
  if (edges < 0 || kernels <= 0 || dim <= 0) return DGNN_ERR_BAD_ARG;
  if (edges == 0) return DGNN_OK;
  if (!pseudo || !mu || !inv_sigma || !gauss) return DGNN_ERR_BAD_ARG;
  const long long total = (long long)edges * kernels;
  hipLaunchKernelGGL(k_gaussian, dim3(grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     total, kernels, dim, pseudo, mu, inv_sigma, gauss);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

size_t dgnn_gaussian_backward_workspace_bytes(int edges, int kernels, int dim) {
  if (edges <= 0 || kernels <= 0 || dim <= 0) return 0;
  const size_t nblocks = ((size_t)edges + DGNN_GBP_BLOCK - 1) / DGNN_GBP_BLOCK;
  return nblocks * 2 * (size_t)kernels * dim * sizeof(float);
}

int dgnn_gaussian_backward(int edges, int kernels, int dim, const float *pseudo, const float *mu,
                           const float *inv_sigma, const float *grad_gauss, float *pseudo_out,
                           float *mu_out, float *sigma_out, void *workspace,
                           size_t workspace_bytes, void *stream) {
  if (edges < 0 || kernels <= 0 || dim <= 0) return DGNN_ERR_BAD_ARG;
  if (!mu || !inv_sigma || !mu_out || !sigma_out) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int KD = kernels * dim;
  if (edges == 0) {
    hipError_t e = hipMemsetAsync(mu_out, 0, sizeof(float) * KD, st);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(sigma_out, 0, sizeof(float) * KD, st);
    return e == hipSuccess ? DGNN_OK : (int)e;
  }
  if (!pseudo || !grad_gauss || !pseudo_out) return DGNN_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < dgnn_gaussian_backward_workspace_bytes(edges, kernels, dim))
    return DGNN_ERR_WORKSPACE;
  int nblocks = (edges + DGNN_GBP_BLOCK - 1) / DGNN_GBP_BLOCK;
  if (kernels <= 4 && dim <= 2) {
    nblocks = nblocks < DGNN_GBP_MAX_BLOCKS ? nblocks : DGNN_GBP_MAX_BLOCKS;
    hipLaunchKernelGGL((k_gaussian_bwd_stream<4, 2>), dim3(nblocks), dim3(DGNN_GBP_BLOCK), 0, st, edges, kernels,
                       dim, pseudo, mu, inv_sigma, grad_gauss, pseudo_out, (float *)workspace);
  } else if (kernels <= 8 && dim <= 4) {
    nblocks = nblocks < DGNN_GBP_MAX_BLOCKS ? nblocks : DGNN_GBP_MAX_BLOCKS;
    hipLaunchKernelGGL((k_gaussian_bwd_stream<8, 4>), dim3(nblocks), dim3(DGNN_GBP_BLOCK), 0, st, edges, kernels,
                       dim, pseudo, mu, inv_sigma, grad_gauss, pseudo_out, (float *)workspace);
  } else {
    hipLaunchKernelGGL(k_gaussian_bwd_stage1, dim3(nblocks), dim3(DGNN_GBP_BLOCK), 0, st, edges,
                       kernels, dim, pseudo, mu, inv_sigma, grad_gauss, pseudo_out,
                       (float *)workspace);
  }
  DGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gaussian_bwd_stage2, dim3(2 * KD), dim3(DGNN_GBP_BLOCK), 0, st, nblocks, KD,
                     (const float *)workspace, mu_out, sigma_out);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_edgeconv_forward(int m, int f, int k, const int *src_ind, const float *h_src,
                          const float *h_dst, float *out_feat, float *max_idx_f32,
                          int *max_idx_i32, void *stream) {
  if (m < 0 || f <= 0 || k < 0) return DGNN_ERR_BAD_ARG;
  if (m == 0) return DGNN_OK;
  if (!h_src || !h_dst || !out_feat || !max_idx_f32 || (k > 0 && !src_ind)) return DGNN_ERR_BAD_ARG;
  const bool al16 = ((((uintptr_t)h_src) | ((uintptr_t)h_dst) | ((uintptr_t)out_feat) | ((uintptr_t)max_idx_f32) |
                      ((uintptr_t)max_idx_i32)) & 15u) == 0;
  if (f % 4 == 0 && al16 && edgeconv_vec4()) {
    int lp = 16;                 // at least 16 lanes per point: two registers then hold a 32-neighbour tile's ids
    while (lp < f / 4 && lp < DGNN_WAVE) lp <<= 1;
    const long long waves = ((long long)m + DGNN_WAVE / lp - 1) / (DGNN_WAVE / lp);
    host::timer_name(0, "k_edgeconv_fwd4");
    host::timer_mark(0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(k_edgeconv_fwd4, dim3(blocks_for_waves(waves)), dim3(DGNN_BLOCK), 0,
                       (hipStream_t)stream, m, f, k, lp, src_ind, h_src, h_dst, out_feat, max_idx_f32,
                       max_idx_i32, edgeconv_xcd());
  } else {
    host::timer_name(0, "k_edgeconv_fwd");
    host::timer_mark(0, 0, (hipStream_t)stream);
    hipLaunchKernelGGL(k_edgeconv_fwd, dim3(blocks_for_waves(m)), dim3(DGNN_BLOCK), 0,
                       (hipStream_t)stream, m, f, k, src_ind, h_src, h_dst, out_feat, max_idx_f32,
                       max_idx_i32, edgeconv_xcd());
  }
  host::timer_mark(0, 1, (hipStream_t)stream);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_edgeconv_backward(int m, int f, const int *max_idx, const float *grad_out,
                           float *grad_src, void *stream) {
  if (m < 0 || f <= 0) return DGNN_ERR_BAD_ARG;
  if (m == 0) return DGNN_OK;
  if (!max_idx || !grad_out || !grad_src) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)m * f;
  hipError_t e = hipMemsetAsync(grad_src, 0, sizeof(float) * (size_t)total, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_edgeconv_bwd, dim3(grid_1d(total, 256)), dim3(256), 0, st, total, f, max_idx,
                     grad_out, grad_src);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

static size_t edgeconv_det_align(size_t x) { return (x + 255) / 256 * 256; }

size_t dgnn_edgeconv_backward_det_workspace_bytes(int m, int k) {
  if (m < 0 || k < 0 || (long long)m * k > 0x7fffffffLL) return 0;
  const size_t nnz = (size_t)m * k;
  return 2 * edgeconv_det_align(sizeof(int) * ((size_t)m + 1)) + 2 * edgeconv_det_align(sizeof(int) * (nnz > 0 ? nnz : 1)) +
         edgeconv_det_align(dgnn_csr2csc_workspace_bytes(m, m, (int)nnz));
}

int dgnn_edgeconv_backward_det(int m, int f, int k, const int *src_ind, const int *max_idx,
                               const float *grad_out, float *grad_src, void *workspace,
                               size_t workspace_bytes, void *stream) {
  if (m < 0 || f <= 0 || k < 0 || (long long)m * k > 0x7fffffffLL) return DGNN_ERR_BAD_ARG;
  if (m == 0) return DGNN_OK;
  if (!max_idx || !grad_out || !grad_src || (k > 0 && !src_ind)) return DGNN_ERR_BAD_ARG;
  if (!workspace || workspace_bytes < dgnn_edgeconv_backward_det_workspace_bytes(m, k)) return DGNN_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nnz = m * k;
  char *w = (char *)workspace;
  w += edgeconv_det_align(sizeof(int) * ((size_t)m + 1));      // (the size function's layout: a row_ptr array no longer made)
  int *col_ptr = (int *)w;                w += edgeconv_det_align(sizeof(int) * ((size_t)m + 1));
  int *row_ind = (int *)w;                w += edgeconv_det_align(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  // the k-NN lists are the CSR of an m x m matrix with exactly k entries per row (row = point, columns
  // = its neighbours): transposed without a row_ptr array or expanded rows (convert.hip); the rest of the
  // workspace (laid out for dgnn_csr2csc by the size function above) is the sort's
  const size_t used = (size_t)(w - (char *)workspace);
  int rc = dgnn::transpose_uniform_lists(m, k, src_ind, col_ptr, row_ind, w, workspace_bytes - used, st);
  if (rc) return rc;
  const bool vec4 = (f % 4 == 0) && ((((uintptr_t)max_idx | (uintptr_t)grad_out | (uintptr_t)grad_src) & 15u) == 0);
  const int slots = vec4 ? f / 4 : f;
  const long long total = (long long)m * slots;
  host::timer_name(1, vec4 ? "k_edgeconv_bwd_gather<4>" : "k_edgeconv_bwd_gather<1>");
  host::timer_mark(1, 0, st);
  if (vec4)
    hipLaunchKernelGGL(k_edgeconv_bwd_gather<4>, dim3(grid_1d(total, 256)), dim3(256), 0, st, total, slots, f,
                       col_ptr, row_ind, max_idx, grad_out, grad_src, edgeconv_xcd());
  else
    hipLaunchKernelGGL(k_edgeconv_bwd_gather<1>, dim3(grid_1d(total, 256)), dim3(256), 0, st, total, slots, f,
                       col_ptr, row_ind, max_idx, grad_out, grad_src, edgeconv_xcd());
  host::timer_mark(1, 1, st);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_pack_rows(int n_out, int n_in, int width, const int *idx, const float *in, float *out,
                   void *stream) {
  if (n_out < 0 || n_in < 0 || width <= 0) return DGNN_ERR_BAD_ARG;
  if (n_out == 0) return DGNN_OK;
  if (!idx || !in || !out || n_in == 0) return DGNN_ERR_BAD_ARG;
  const bool vec4 = (width % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out) & 15u) == 0);
  const int wv = vec4 ? width / 4 : width;
  const long long total = (long long)n_out * wv;
  // four slots per thread per trip
  const int grid = grid_1d((total + 3) / 4, 256);
  if (vec4)
    hipLaunchKernelGGL(k_pack_rows<float4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, wv, idx, in, out);
  else
    hipLaunchKernelGGL(k_pack_rows<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, total, wv, idx, in, out);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

}  // extern "C"

// Generic (any H, any F) fused-GAT kernels: one wave64 per CSR row / CSC column.
// These are the shape-agnostic fallback; the neighbour-grouped fast path lives in
// gat_fast.hpp and is used whenever its layout constraints hold (see gat.hip).
#pragma once
#include "common.hpp"
#include "gather_fast.hpp"

namespace dgnn {

// (non-template kernels: each is compiled by the one translation unit that launches it)
#ifdef DGNN_TU_FORWARD
// Forward.  Semantics of fused_forward_kernel (fused_gatconv_kernel.cu:24-118): row max of
// LeakyReLU(attn_row[r,h]+attn_col[c,h]) from -1e38, sum of exp(e-max), then
// out[r,h,:] = sum_j [mask_j > p] * exp(e_j-max)/sum / (1.0-p) * in_feat[c_j,h,:].
// edge_max/edge_sum may be NULL (inference).  edge_mask may be NULL (keep all).
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_fwd_row(
    int m, int H, int F, float attn_drop, float slope, const float *__restrict__ attn_row,
    const float *__restrict__ attn_col, const int *__restrict__ row_ptr,
    const int *__restrict__ col_ind, const float *__restrict__ in_feat,
    const float *__restrict__ edge_mask, float *__restrict__ edge_max,
    float *__restrict__ edge_sum, float *__restrict__ out) {
  const int r = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (r >= m) return;
  const int lane = lane_id();
  const int lb = row_ptr[r], hb = row_ptr[r + 1];
  const size_t HF = (size_t)H * F;
  for (int h = 0; h < H; ++h) {
    const float ar = attn_row[(size_t)r * H + h];
    float mx = DGNN_NEG_INIT;
    for (int p = lb + lane; p < hb; p += DGNN_WAVE)
      mx = fmaxf(mx, leaky_relu(ar + attn_col[(size_t)col_ind[p] * H + h], slope));
    mx = wave_max(mx);
    float sm = 0.f;
    for (int p = lb + lane; p < hb; p += DGNN_WAVE)
      sm += expf(leaky_relu(ar + attn_col[(size_t)col_ind[p] * H + h], slope) - mx);
    sm = wave_sum(sm);
    if (edge_max != nullptr && lane == 0) {
      edge_max[(size_t)r * H + h] = mx;
      edge_sum[(size_t)r * H + h] = sm;
    }
    for (int fb = 0; fb < F; fb += DGNN_WAVE) {
      const int fid = fb + lane;
      float acc = 0.f;
      for (int base = lb; base < hb; base += DGNN_WAVE) {
        const int p = base + lane;
        float w = 0.f;
        int c = 0;
        if (p < hb) {
          c = col_ind[p];
          const bool keep = (edge_mask == nullptr) || (edge_mask[(size_t)p * H + h] > attn_drop);
          if (keep) w = expf(leaky_relu(ar + attn_col[(size_t)c * H + h], slope) - mx) / sm;
          if (attn_drop > 0.f) w = drop_scale(w, attn_drop);
        }
        const int n = min(DGNN_WAVE, hb - base);
        for (int k = 0; k < n; ++k) {
          const int ck = bcast_i32(c, k);
          const float wk = bcast_f32(w, k);
          if (fid < F) acc = fmaf(wk, in_feat[(size_t)ck * HF + (size_t)h * F + fid], acc);
        }
      }
      if (fid < F) out[(size_t)r * HF + (size_t)h * F + fid] = acc;
    }
  }
}

#endif
#ifdef DGNN_TU_BACKWARD
// S[r,h] = <grad[r,h,:], out[r,h,:]>.  Equals the row term sum_j a_j g_j that
// fused_backward_kernel builds in its first sweep (fused_gatconv_kernel.cu:906-931), because
// out[r,h,:] = sum_j a_j*mask_j/(1-p) * in_feat[c_j,h,:] and g_j = mask_j/(1-p) * <grad[r],in_feat[c_j]>.
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_dot(int m, int H, int F,
                                                            const float *__restrict__ grad,
                                                            const float *__restrict__ out,
                                                            float *__restrict__ S) {
  const int r = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (r >= m) return;
  const int lane = lane_id();
  const size_t HF = (size_t)H * F;
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    for (int c = lane; c < F; c += DGNN_WAVE)
      s = fmaf(grad[(size_t)r * HF + (size_t)h * F + c], out[(size_t)r * HF + (size_t)h * F + c], s);
    s = wave_sum(s);
    if (lane == 0) S[(size_t)r * H + h] = s;
  }
}

// Transposed (CSC) pass, one wave per source column c.  Fuses mhspmm_backward_kernel
// (:712-763, grad_feat), the SDDMM of mhsddmm (:814-890, edge dot products, here against the
// column's own in_feat row) and the per-edge part of fused_backward_kernel (:934-958):
//   a   = exp(LeakyReLU(attn_row[r]+attn_col[c]) - edge_max[r]) / edge_sum[r]
//   g   = [mask>p]/(1.0-p) * <grad[r,h,:], in_feat[c,h,:]>
//   d   = a * (g - S[r,h]) * (e < 0 ? slope : 1)
//   grad_feat[c,h,:]   = sum_i [mask>p] a_i/(1-p) * grad[r_i,h,:]
//   grad_attn_col[c,h] = sum_i d_i            (a plain sum: no atomics, deterministic)
//   dE[permute[i],h]   = d_i                  (CSR-ordered, summed per row by k_gat_row_sum)
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_bwd_col(
    int n, int H, int F, float slope, float attn_drop, const int *__restrict__ col_ptr,
    const int *__restrict__ row_ind, const int *__restrict__ permute,
    const float *__restrict__ edge_max, const float *__restrict__ edge_sum,
    const float *__restrict__ edge_mask, const float *__restrict__ attn_row,
    const float *__restrict__ attn_col, const float *__restrict__ in_feat,
    const float *__restrict__ grad, const float *__restrict__ S, float *__restrict__ dE,
    float *__restrict__ grad_feat, float *__restrict__ grad_attn_col) {
  const int c = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (c >= n) return;
  const int lane = lane_id();
  const int lb = col_ptr[c], hb = col_ptr[c + 1];
  const size_t HF = (size_t)H * F;
  for (int h = 0; h < H; ++h) {
    const float ac = attn_col[(size_t)c * H + h];
    const float *u = in_feat + (size_t)c * HF + (size_t)h * F;
    float dsum = 0.f;
    for (int base = lb; base < hb; base += DGNN_WAVE) {
      const int p = base + lane;
      int rid = 0, pos = 0;
      float a = 0.f, e = 0.f, Sv = 0.f;
      bool keep = false;
      if (p < hb) {
        rid = row_ind[p];
        pos = permute[p];
        e = leaky_relu(attn_row[(size_t)rid * H + h] + ac, slope);
        a = expf(e - edge_max[(size_t)rid * H + h]) / edge_sum[(size_t)rid * H + h];
        keep = (edge_mask == nullptr) || (edge_mask[(size_t)pos * H + h] > attn_drop);
        Sv = S[(size_t)rid * H + h];
      }
      const int cnt = min(DGNN_WAVE, hb - base);
      float mydot = 0.f;
      for (int k = 0; k < cnt; ++k) {
        const int rk = bcast_i32(rid, k);
        const float *g = grad + (size_t)rk * HF + (size_t)h * F;
        float part = 0.f;
        for (int q = lane; q < F; q += DGNN_WAVE) part = fmaf(g[q], u[q], part);
        part = wave_sum(part);
        if (lane == k) mydot = part;
      }
      if (p < hb) {
        float ge = 0.f;
        if (keep) ge = (attn_drop > 0.f) ? drop_scale(mydot, attn_drop) : mydot;
        float d = a * (ge - Sv);
        if (e < 0.f) d *= slope;
        dsum += d;
        dE[(size_t)pos * H + h] = d;
      }
    }
    dsum = wave_sum(dsum);
    if (lane == 0) grad_attn_col[(size_t)c * H + h] = dsum;

    for (int fb = 0; fb < F; fb += DGNN_WAVE) {
      const int fid = fb + lane;
      float acc = 0.f;
      for (int base = lb; base < hb; base += DGNN_WAVE) {
        const int p = base + lane;
        int rid = 0;
        float w = 0.f;
        if (p < hb) {
          rid = row_ind[p];
          const bool keep =
              (edge_mask == nullptr) || (edge_mask[(size_t)permute[p] * H + h] > attn_drop);
          if (keep) {
            const float e = leaky_relu(attn_row[(size_t)rid * H + h] + ac, slope);
            w = expf(e - edge_max[(size_t)rid * H + h]) / edge_sum[(size_t)rid * H + h];
          }
          if (attn_drop > 0.f) w = w / (1 - attn_drop);  // float divide, :749
        }
        const int cnt = min(DGNN_WAVE, hb - base);
        for (int k = 0; k < cnt; ++k) {
          const int rk = bcast_i32(rid, k);
          const float wk = bcast_f32(w, k);
          if (fid < F) acc = fmaf(wk, grad[(size_t)rk * HF + (size_t)h * F + fid], acc);
        }
      }
      if (fid < F) grad_feat[(size_t)c * HF + (size_t)h * F + fid] = acc;
    }
  }
}

// grad_attn_row[r,h] = sum over the row's edges of dE[e,h]  (fused_backward_kernel :961-967).
// H == 4 fast path: a 16-lane group owns DGNN_ROWSUM_R consecutive rows and issues the first
// two 256-byte steps of ALL of them before reducing (8 independent float4 loads in flight per
// lane: the per-row work is tiny, so the kernel lives on memory-level parallelism); longer rows
// continue in a loop.  Rows beyond `long_thr` edges (rowsum_long_threshold(), a few hundred: on a
// power-law graph the heavy rows sit next to each other, and a block lives as long as its
// slowest 16-lane group) are queued and finished by one whole block each with a fixed reduction
// tree.  Everything is a fixed-order sum: deterministic.
#define DGNN_ROWSUM_R 4
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_sum4(int m, const int *__restrict__ row_ptr,
                                                             const float4 *__restrict__ d4,
                                                             float4 *__restrict__ grad_attn_row,
                                                             int *__restrict__ long_count,
                                                             int *__restrict__ long_list, int long_thr) {
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int rows_per_block = 16 * DGNN_ROWSUM_R;
  const int rg = blockIdx.x * rows_per_block + grp * DGNN_ROWSUM_R;
  // lanes 0..R of the group fetch the R+1 row pointers, then everybody reads them by shuffle
  int myptr = 0;
  if (q <= DGNN_ROWSUM_R && rg + q <= m) myptr = row_ptr[rg + q];
  int lb[DGNN_ROWSUM_R], hb[DGNN_ROWSUM_R];
  bool skip[DGNN_ROWSUM_R];
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 v0[DGNN_ROWSUM_R], v1[DGNN_ROWSUM_R];
#pragma unroll
  for (int i = 0; i < DGNN_ROWSUM_R; ++i) {
    lb[i] = __shfl(myptr, i, 16);
    hb[i] = __shfl(myptr, i + 1, 16);
    skip[i] = rg + i >= m;
    if (skip[i]) { lb[i] = 0; hb[i] = 0; }
    if (hb[i] - lb[i] > long_thr) {
      // hub row: queued for k_gat_row_sum4_long (one whole block per hub row).  The queue order
      // varies from run to run, the value written for each row does not.
      if (q == 0) long_list[atomicAdd(long_count, 1)] = rg + i;
      skip[i] = true;
      hb[i] = lb[i];
    }
    v0[i] = (lb[i] + q < hb[i]) ? d4[lb[i] + q] : z;
    v1[i] = (lb[i] + q + 16 < hb[i]) ? d4[lb[i] + q + 16] : z;
  }
#pragma unroll
  for (int i = 0; i < DGNN_ROWSUM_R; ++i) {
    float4 s = make_float4(v0[i].x + v1[i].x, v0[i].y + v1[i].y, v0[i].z + v1[i].z, v0[i].w + v1[i].w);
    for (int p = lb[i] + q + 32; p < hb[i]; p += 32) {
      const float4 a = d4[p];
      const float4 c = (p + 16 < hb[i]) ? d4[p + 16] : z;
      s.x += a.x + c.x; s.y += a.y + c.y; s.z += a.z + c.z; s.w += a.w + c.w;
    }
    s.x = row16_reduce(s.x, OpSum());
    s.y = row16_reduce(s.y, OpSum());
    s.z = row16_reduce(s.z, OpSum());
    s.w = row16_reduce(s.w, OpSum());
    if (q == 0 && !skip[i]) grad_attn_row[rg + i] = s;
  }
}

// One block per queued hub row: 256 threads, 4 independent float4 loads per step, fixed LDS
// tree.  The grid is sized for the worst case (nnz / threshold rows); spare blocks exit.
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_sum4_long(const int *__restrict__ row_ptr,
                                                                  const float4 *__restrict__ d4,
                                                                  float4 *__restrict__ grad_attn_row,
                                                                  const int *__restrict__ long_count,
                                                                  const int *__restrict__ long_list) {
  __shared__ float4 red[DGNN_BLOCK];
  if ((int)blockIdx.x >= *long_count) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int rr = long_list[blockIdx.x];
  const int l = row_ptr[rr], h2 = row_ptr[rr + 1];
  float4 s0 = z, s1 = z, s2 = z, s3 = z;
  for (int p = l + (int)threadIdx.x; p < h2; p += 4 * DGNN_BLOCK) {
    const float4 a0 = d4[p];
    const float4 a1 = (p + DGNN_BLOCK < h2) ? d4[p + DGNN_BLOCK] : z;
    const float4 a2 = (p + 2 * DGNN_BLOCK < h2) ? d4[p + 2 * DGNN_BLOCK] : z;
    const float4 a3 = (p + 3 * DGNN_BLOCK < h2) ? d4[p + 3 * DGNN_BLOCK] : z;
    s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
    s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
    s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
    s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
  }
  red[threadIdx.x] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                                 (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  for (int st = DGNN_BLOCK / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      const float4 o = red[threadIdx.x + st];
      float4 mine = red[threadIdx.x];
      mine.x += o.x; mine.y += o.y; mine.z += o.z; mine.w += o.w;
      red[threadIdx.x] = mine;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) grad_attn_row[rr] = red[0];
}

// H == 8: a row's [deg][8] slab is 2*deg float4s; the 16 lanes of a group stride over it, so a lane
// always holds the same half of the heads (q % 2) and adds whole float4s (the scalar kernel below
// moves one float per lane: 8 x 8 on the products shape 1.5 + 1.1 ms for the two row-sum kernels).
// Hub rows are queued for k_gat_row_sum8_long as in the H == 4 pair.  Fixed-order sums.
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_sum8(int m, const int *__restrict__ row_ptr,
                                                             const float4 *__restrict__ d4,
                                                             float4 *__restrict__ grad_attn_row,
                                                             int *__restrict__ long_count,
                                                             int *__restrict__ long_list, int long_thr) {
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int r = blockIdx.x * 16 + grp;
  if (r >= m) return;
  const int lb = row_ptr[r], hb = row_ptr[r + 1];
  if (hb - lb > long_thr) {
    if (q == 0) long_list[atomicAdd(long_count, 1)] = r;
    return;
  }
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s = z;
  const long long beg = 2ll * lb, end = 2ll * hb;
  for (long long i = beg + q; i < end; i += 32) {
    const float4 a = d4[i];
    const float4 c = (i + 16 < end) ? d4[i + 16] : z;
    s.x += a.x + c.x; s.y += a.y + c.y; s.z += a.z + c.z; s.w += a.w + c.w;
  }
#pragma unroll
  for (int st = 2; st < 16; st <<= 1) {
    s.x += __shfl_xor(s.x, st, 16); s.y += __shfl_xor(s.y, st, 16);
    s.z += __shfl_xor(s.z, st, 16); s.w += __shfl_xor(s.w, st, 16);
  }
  if (q < 2) grad_attn_row[2ll * r + q] = s;
}

__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_sum8_long(const int *__restrict__ row_ptr,
                                                                  const float4 *__restrict__ d4,
                                                                  float4 *__restrict__ grad_attn_row,
                                                                  const int *__restrict__ long_count,
                                                                  const int *__restrict__ long_list) {
  __shared__ float4 red[DGNN_BLOCK];
  if ((int)blockIdx.x >= *long_count) return;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  const int rr = long_list[blockIdx.x];
  const long long beg = 2ll * row_ptr[rr], end = 2ll * row_ptr[rr + 1];
  float4 s0 = z, s1 = z;
  for (long long i = beg + (int)threadIdx.x; i < end; i += 2 * DGNN_BLOCK) {
    const float4 a0 = d4[i];
    const float4 a1 = (i + DGNN_BLOCK < end) ? d4[i + DGNN_BLOCK] : z;
    s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
    s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
  }
  red[threadIdx.x] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
  __syncthreads();
  for (int st = DGNN_BLOCK / 2; st >= 2; st >>= 1) {        // down to stride 2: the two head halves stay apart
    if ((int)threadIdx.x < st) {
      const float4 o = red[threadIdx.x + st];
      float4 mine = red[threadIdx.x];
      mine.x += o.x; mine.y += o.y; mine.z += o.z; mine.w += o.w;
      red[threadIdx.x] = mine;
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) grad_attn_row[2ll * rr + threadIdx.x] = red[threadIdx.x];
}

// Any H: one 16-lane group per row; rows above `long_thr` edges are queued for
// k_gat_row_sum_long (one block per row), as in the H == 4 pair above.
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_sum(int m, int H,
                                                            const int *__restrict__ row_ptr,
                                                            const float *__restrict__ dE,
                                                            float *__restrict__ grad_attn_row,
                                                            int *__restrict__ long_count,
                                                            int *__restrict__ long_list, int long_thr) {
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int r = blockIdx.x * 16 + grp;
  if (r >= m) return;
  const int lb = row_ptr[r], hb = row_ptr[r + 1];
  if (hb - lb > long_thr) {
    if (q == 0) long_list[atomicAdd(long_count, 1)] = r;
    return;
  }
  if (H <= 16 && (16 % H) == 0) {
    // the 16 lanes walk the row's contiguous [deg*H] slab; since 16 % H == 0 a lane always
    // lands on head q % H
    float s = 0.f;
    const size_t beg = (size_t)lb * H, end = (size_t)hb * H;
    for (size_t i = beg + q; i < end; i += 16) s += dE[i];
    for (int st = H; st < 16; st <<= 1) s += __shfl_xor(s, st, 16);
    if (q < H) grad_attn_row[(size_t)r * H + q] = s;
  } else {
    for (int h = q; h < H; h += 16) {
      float s = 0.f;
      for (int p = lb; p < hb; ++p) s += dE[(size_t)p * H + h];
      grad_attn_row[(size_t)r * H + h] = s;
    }
  }
}

// One block per queued long row, any H.  256 % H == 0: a thread keeps its head while striding
// over the row's slab and the tree keeps heads apart down to stride H; other H: one tree per head.
__global__ __launch_bounds__(DGNN_BLOCK) void k_gat_row_sum_long(int H, const int *__restrict__ row_ptr,
                                                                 const float *__restrict__ dE,
                                                                 float *__restrict__ grad_attn_row,
                                                                 const int *__restrict__ long_count,
                                                                 const int *__restrict__ long_list) {
  __shared__ float red[DGNN_BLOCK];
  if ((int)blockIdx.x >= *long_count) return;
  const int rr = long_list[blockIdx.x];
  const int lb = row_ptr[rr], hb = row_ptr[rr + 1];
  if (H <= DGNN_BLOCK && (DGNN_BLOCK % H) == 0) {
    float s = 0.f;
    const size_t beg = (size_t)lb * H, end = (size_t)hb * H;
    for (size_t i = beg + threadIdx.x; i < end; i += DGNN_BLOCK) s += dE[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = DGNN_BLOCK / 2; st >= H; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    if ((int)threadIdx.x < H) grad_attn_row[(size_t)rr * H + threadIdx.x] = red[threadIdx.x];
  } else {
    for (int h = 0; h < H; ++h) {
      float s = 0.f;
      for (int p = lb + (int)threadIdx.x; p < hb; p += DGNN_BLOCK) s += dE[(size_t)p * H + h];
      red[threadIdx.x] = s;
      __syncthreads();
      for (int st = DGNN_BLOCK / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
      }
      if (threadIdx.x == 0) grad_attn_row[(size_t)rr * H + h] = red[0];
      __syncthreads();
    }
  }
}

#endif
#ifdef DGNN_TU_FORWARD
// --------------------------------------------------------------------------- SpMM / GMM
// One wave per CSR row.  WEIGHT = 0: caller-supplied edge values (GE-SpMM, phase 3 of
// fused_gatconv_kernel.cu:87-117).  WEIGHT = 1: Gaussian kernel weight computed on the fly
// from pseudo coordinates (fuseGmm, fused_gmmconv_kernel.cu:24-55): float accumulation of
// ((p-mu)^2 * sig^2), exponent in DOUBLE as the reference's `exp(-0.5 * gauss)` (:44).
template <int WEIGHT>
__global__ __launch_bounds__(DGNN_BLOCK) void k_spmm_row(
    int m, int H, int F, int dim, const int *__restrict__ row_ptr,
    const int *__restrict__ col_ind, const float *__restrict__ edge_val,
    const float *__restrict__ pseudo, const float *__restrict__ mu,
    const float *__restrict__ inv_sigma, const float *__restrict__ in_feat,
    float *__restrict__ out) {
  const int r = uniform_i32(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block());
  if (r >= m) return;
  const int lane = lane_id();
  const int lb = row_ptr[r], hb = row_ptr[r + 1];
  const size_t HF = (size_t)H * F;
  for (int h = 0; h < H; ++h) {
    for (int fb = 0; fb < F; fb += DGNN_WAVE) {
      const int fid = fb + lane;
      float acc = 0.f;
      for (int base = lb; base < hb; base += DGNN_WAVE) {
        const int p = base + lane;
        float w = 0.f;
        int c = 0;
        if (p < hb) {
          c = col_ind[p];
          if (WEIGHT == 0)
            w = edge_val[(size_t)p * H + h];
          else
            w = gmm_weight(dim, pseudo + (size_t)p * dim, mu + (size_t)h * dim,
                             inv_sigma + (size_t)h * dim);
        }
        const int n = min(DGNN_WAVE, hb - base);
        for (int k = 0; k < n; ++k) {
          const int ck = bcast_i32(c, k);
          const float wk = bcast_f32(w, k);
          if (fid < F) acc = fmaf(wk, in_feat[(size_t)ck * HF + (size_t)h * F + fid], acc);
        }
      }
      if (fid < F) out[(size_t)r * HF + (size_t)h * F + fid] = acc;
    }
  }
}


#endif
}  // namespace dgnn

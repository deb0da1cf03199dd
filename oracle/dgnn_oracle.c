/*
 * dgnn_oracle.c -- CPU restatement of dgNN's fused GNN operators.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under dgnn_amd/ may import, link or call
 * this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" against reference OUTPUTS.  The reference
 * (/root/reference, dgSPARSE/dgNN) is CUDA-only (nvcc, cuRAND, cuSPARSE), ships
 * no golden vectors, no fixtures and no CPU path, so it can be neither built
 * nor run here.  What pins this file instead is (a) a line-by-line restatement
 * of the reference kernels cited on every function below, including the
 * 32-lane tile order of their reductions, and (b) formula-equivalence tests
 * (tests/test_oracle.py) against the dense DGL/PyG formulation that the
 * reference's own functional tests compare with (test/func_test/func_gat_dgl.py:46-68,
 * func_gmm_pyg.py:74-89, func_edgeconv_pyg.py:24-41) and against torch autograd
 * for every gradient.
 *
 * Two flavours per operator:
 *   *_f32 : float arithmetic in the reference's order (32-lane butterfly
 *           reductions, sequential FMA accumulation as nvcc contracts
 *           `acc += a*b`), one double divide where the reference has one.
 *   *_f64 : the same formula in double with plain left-to-right sums; used as
 *           the tolerance anchor (both the HIP path and *_f32 must sit within
 *           rel 1e-5 / abs 1e-6 of it).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LANES 32 /* the reference's CUDA warp width; its tile size */

static int g_threads = 1;
void oracle_set_threads(int n) {
  g_threads = n > 0 ? n : 1;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}
int oracle_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_num_procs();
#else
  return 1;
#endif
}

/* fused_gatconv_kernel.cu:13  #define LeakyRelu(x, s) ((x > 0) ? (x) : ((x)*s)) */
static inline float lrelu_f(float x, float s) { return (x > 0) ? x : x * s; }
static inline double lrelu_d(double x, double s) { return (x > 0) ? x : x * s; }

/* xor-butterfly all-reduce over 32 lanes, as `for (stride=16;stride>0;stride>>=1)
 * v += __shfl_xor_sync(.., v, stride, 32)` (fused_gatconv_kernel.cu:78-81). */
static inline void butterfly_sum(float v[LANES]) {
  float t[LANES];
  for (int stride = 16; stride > 0; stride >>= 1) {
    for (int l = 0; l < LANES; ++l) t[l] = v[l ^ stride];
    for (int l = 0; l < LANES; ++l) v[l] += t[l];
  }
}

/* ------------------------------------------------------------------------- *
 * GAT forward (training).  fused_gatconv_kernel.cu:24-118 (kernel), :322-361
 * (launcher).  edge_mask may be NULL: then every edge is kept (the reference
 * always draws a mask in (0,1] and keeps `mask > attn_drop`, :95; with
 * attn_drop == 0 that keeps everything).
 * ------------------------------------------------------------------------- */
void oracle_gat_forward_f32(int m, int nnz, int h, int f, float attn_drop,
                            float negative_slope, const float *attn_row,
                            const float *attn_col, const int *row_ptr,
                            const int *col_ind, const float *in_feat,
                            const float *edge_mask, float *edge_max,
                            float *edge_sum, float *out_feat) {
  (void)nnz;
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < m; ++rid) {
    for (int hid = 0; hid < h; ++hid) {
      int lb = row_ptr[rid], hb = row_ptr[rid + 1];
      int loop = (hb - lb + 31) / 32;
      float attn_row_val = attn_row[rid * h + hid];
      float weightMax = -1e38; /* :45 */
      for (int j = 0; j < loop; ++j) {
        float tmax = -1e38; /* :49, max is order independent */
        for (int tx = 0; tx < LANES; ++tx) {
          int pid = lb + tx + (j << 5);
          if (pid < hb) {
            float w = attn_row_val + attn_col[(size_t)col_ind[pid] * h + hid];
            w = lrelu_f(w, negative_slope);
            tmax = tmax < w ? w : tmax;
          }
        }
        weightMax = tmax < weightMax ? weightMax : tmax; /* MAX(weight, weightMax) :61 */
      }
      edge_max[rid * h + hid] = weightMax; /* :64 */

      float expAll[LANES];
      for (int l = 0; l < LANES; ++l) expAll[l] = 0;
      for (int j = 0; j < loop; ++j) {
        float e[LANES];
        for (int tx = 0; tx < LANES; ++tx) {
          int pid = lb + tx + (j << 5);
          e[tx] = 0;
          if (pid < hb) {
            float w = attn_row_val + attn_col[(size_t)col_ind[pid] * h + hid];
            w = lrelu_f(w, negative_slope);
            e[tx] = expf(w - weightMax); /* :75 exp(float) */
          }
        }
        butterfly_sum(e);
        for (int l = 0; l < LANES; ++l) expAll[l] += e[l]; /* :82 */
      }
      edge_sum[rid * h + hid] = expAll[0]; /* :85 lane 0 */

      for (int fid = 0; fid < f; ++fid) {
        float acc = 0;
        for (int j = 0; j < loop; ++j) {
          int jj = lb + (j << 5);
          for (int kk = 0; kk < 32 && jj + kk < hb; ++kk) {
            int pid = jj + kk;
            float weight = 0;
            int cid = 0;
            if (edge_mask == NULL ||
                edge_mask[(size_t)pid * h + hid] > attn_drop) { /* :95 */
              cid = col_ind[pid];
              float w = attn_row_val + attn_col[(size_t)cid * h + hid];
              w = lrelu_f(w, negative_slope);
              weight = expf(w - weightMax) / expAll[kk]; /* :102, lane kk's own sum */
            }
            float val = (float)((double)weight / (1.0 - (double)attn_drop)); /* :104 */
            acc = fmaf(val, in_feat[(size_t)cid * h * f + hid * f + fid], acc); /* :111 */
          }
        }
        out_feat[(size_t)rid * h * f + hid * f + fid] = acc;
      }
    }
  }
}

void oracle_gat_forward_f64(int m, int nnz, int h, int f, double attn_drop,
                            double negative_slope, const float *attn_row,
                            const float *attn_col, const int *row_ptr,
                            const int *col_ind, const float *in_feat,
                            const float *edge_mask, double *edge_max,
                            double *edge_sum, double *out_feat) {
  (void)nnz;
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < m; ++rid) {
    for (int hid = 0; hid < h; ++hid) {
      int lb = row_ptr[rid], hb = row_ptr[rid + 1];
      double ar = attn_row[rid * h + hid];
      double mx = (double)(float)-1e38;
      for (int p = lb; p < hb; ++p) {
        double w = lrelu_d(ar + (double)attn_col[(size_t)col_ind[p] * h + hid],
                           negative_slope);
        if (w > mx) mx = w;
      }
      double s = 0;
      for (int p = lb; p < hb; ++p) {
        double w = lrelu_d(ar + (double)attn_col[(size_t)col_ind[p] * h + hid],
                           negative_slope);
        s += exp(w - mx);
      }
      edge_max[rid * h + hid] = mx;
      edge_sum[rid * h + hid] = s;
      double *o = out_feat + (size_t)rid * h * f + hid * f;
      for (int fid = 0; fid < f; ++fid) o[fid] = 0;
      for (int p = lb; p < hb; ++p) {
        if (edge_mask != NULL &&
            !(edge_mask[(size_t)p * h + hid] > (float)attn_drop))
          continue;
        int cid = col_ind[p];
        double w = lrelu_d(ar + (double)attn_col[(size_t)cid * h + hid],
                           negative_slope);
        double a = exp(w - mx) / s / (1.0 - attn_drop);
        const float *x = in_feat + (size_t)cid * h * f + hid * f;
        for (int fid = 0; fid < f; ++fid) o[fid] += a * (double)x[fid];
      }
    }
  }
}

/* GAT inference.  fused_inference_kernel, fused_gatconv_kernel.cu:391-478 (the
 * F<=64 variant :565-669 computes the same formula).  No mask, no 1/(1-p). */
void oracle_gat_inference_f32(int m, int nnz, int h, int f,
                              float negative_slope, const float *attn_row,
                              const float *attn_col, const int *row_ptr,
                              const int *col_ind, const float *in_feat,
                              float *out_feat) {
  float *mx = (float *)malloc(sizeof(float) * (size_t)m * h);
  float *sm = (float *)malloc(sizeof(float) * (size_t)m * h);
  oracle_gat_forward_f32(m, nnz, h, f, 0.0f, negative_slope, attn_row, attn_col,
                         row_ptr, col_ind, in_feat, NULL, mx, sm, out_feat);
  free(mx);
  free(sm);
}

/* ------------------------------------------------------------------------- *
 * GAT backward.  Launcher fused_gatconv_kernel.cu:998-1038 runs three kernels:
 *   mhspmm_backward_kernel :712-763  (CSC pass -> grad_feat)
 *   mhsddmm                :814-890  (grad_edge_csr[e,h] = <grad[row(e)],in_feat[col(e)]>)
 *   fused_backward_kernel  :892-968  (softmax/LeakyReLU backward)
 * edge_mask may be NULL (all kept).
 * ------------------------------------------------------------------------- */
static float sddmm_edge_f32(int f, const float *a, const float *b) {
  /* mhsddmm: lane l accumulates f = l, l+32, ... with FMA (:838-852), then the
   * xor butterfly of AllReduce4 (computeUtil.h:143-151); lane 0 stores. */
  float lane[LANES];
  for (int l = 0; l < LANES; ++l) lane[l] = 0;
  for (int l = 0; l < LANES; ++l)
    for (int c = l; c < f; c += 32) lane[l] = fmaf(a[c], b[c], lane[l]);
  butterfly_sum(lane);
  return lane[0];
}

void oracle_mhsddmm_f32(int v, int f, int h, int nnz, const int *rowptr,
                        const int *colind, const float *grad,
                        const float *feature, float *out) {
  (void)nnz;
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < v; ++r)
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e)
      for (int hid = 0; hid < h; ++hid)
        out[(size_t)e * h + hid] =
            sddmm_edge_f32(f, grad + (size_t)r * h * f + hid * f,
                           feature + (size_t)colind[e] * h * f + hid * f);
}

void oracle_mhsddmm_f64(int v, int f, int h, int nnz, const int *rowptr,
                        const int *colind, const float *grad,
                        const float *feature, double *out) {
  (void)nnz;
#pragma omp parallel for schedule(dynamic, 64)
  for (int r = 0; r < v; ++r)
    for (int e = rowptr[r]; e < rowptr[r + 1]; ++e)
      for (int hid = 0; hid < h; ++hid) {
        const float *a = grad + (size_t)r * h * f + hid * f;
        const float *b = feature + (size_t)colind[e] * h * f + hid * f;
        double s = 0;
        for (int c = 0; c < f; ++c) s += (double)a[c] * (double)b[c];
        out[(size_t)e * h + hid] = s;
      }
}

void oracle_gat_backward_f32(int m, int nnz, int h, int f, float negative_slope,
                             float attn_drop, const int *row_ptr,
                             const int *col_ind, const int *col_ptr,
                             const int *row_ind, const int *permute,
                             const float *edge_max, const float *edge_sum,
                             const float *edge_mask, const float *in_feat,
                             const float *attn_row, const float *attn_col,
                             const float *grad, float *grad_feat,
                             float *grad_attn_row, float *grad_attn_col) {
  /* --- mhspmm_backward_kernel :712-763 --- */
#pragma omp parallel for schedule(dynamic, 64)
  for (int cid = 0; cid < m; ++cid) {
    for (int hid = 0; hid < h; ++hid) {
      int lb = col_ptr[cid], hb = col_ptr[cid + 1];
      float attn_col_val = attn_col[cid * h + hid];
      for (int fid = 0; fid < f; ++fid) {
        float acc = 0;
        for (int pid = lb; pid < hb; ++pid) {
          float weight = 0;
          int rid = 0;
          if (edge_mask == NULL ||
              edge_mask[(size_t)permute[pid] * h + hid] > attn_drop) { /* :738 */
            rid = row_ind[pid];
            float w = attn_row[(size_t)rid * h + hid] + attn_col_val;
            w = lrelu_f(w, negative_slope);
            weight = expf(w - edge_max[(size_t)rid * h + hid]) /
                     edge_sum[(size_t)rid * h + hid]; /* :747 */
          }
          float val = weight / (1 - attn_drop); /* :749 float divide */
          acc = fmaf(val, grad[(size_t)rid * h * f + hid * f + fid], acc);
        }
        grad_feat[(size_t)cid * h * f + hid * f + fid] = acc;
      }
    }
  }
  /* --- mhsddmm :814-890 --- */
  float *grad_edge_csr = (float *)malloc(sizeof(float) * (size_t)(nnz > 0 ? nnz : 1) * h);
  oracle_mhsddmm_f32(m, f, h, nnz, row_ptr, col_ind, grad, in_feat, grad_edge_csr);
  /* --- fused_backward_kernel :892-968 ; grad_attn_col zeroed by host :1057 --- */
  memset(grad_attn_col, 0, sizeof(float) * (size_t)m * h);
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < m; ++rid) {
    for (int hid = 0; hid < h; ++hid) {
      int lb = row_ptr[rid], hb = row_ptr[rid + 1];
      int loop = (hb - lb + 31) / 32;
      float attn_row_val = attn_row[rid * h + hid];
      float emax = edge_max[rid * h + hid], esum = edge_sum[rid * h + hid];
      float weightSum[LANES];
      for (int l = 0; l < LANES; ++l) weightSum[l] = 0;
      for (int j = 0; j < loop; ++j) {
        float w[LANES];
        for (int tx = 0; tx < LANES; ++tx) {
          int pid = lb + tx + (j << 5);
          w[tx] = 0;
          if (pid < hb) {
            float vl = lrelu_f(attn_row_val + attn_col[(size_t)col_ind[pid] * h + hid],
                               negative_slope);
            float vs = expf(vl - emax) / esum;
            float g_edge = 0;
            if (edge_mask == NULL || edge_mask[(size_t)pid * h + hid] > attn_drop)
              g_edge = (float)((double)grad_edge_csr[(size_t)pid * h + hid] /
                               (1.0 - (double)attn_drop)); /* :923 */
            w[tx] = vs * g_edge;
          }
        }
        butterfly_sum(w);
        for (int l = 0; l < LANES; ++l) weightSum[l] = w[l] + weightSum[l]; /* :930 */
      }
      float grad_row_sum = 0;
      for (int j = 0; j < loop; ++j) {
        float go[LANES];
        for (int tx = 0; tx < LANES; ++tx) {
          int pid = lb + tx + (j << 5);
          go[tx] = 0;
          if (pid < hb) {
            int cid = col_ind[pid];
            float vl = lrelu_f(attn_row_val + attn_col[(size_t)cid * h + hid],
                               negative_slope);
            float vs = expf(vl - emax) / esum;
            float g_edge = 0;
            if (edge_mask == NULL || edge_mask[(size_t)pid * h + hid] > attn_drop)
              g_edge = (float)((double)grad_edge_csr[(size_t)pid * h + hid] /
                               (1.0 - (double)attn_drop));
            float g = vs * (g_edge - weightSum[tx]); /* :952 */
            if (vl < 0) g = g * negative_slope;      /* :953 */
            go[tx] = g;
#pragma omp atomic
            grad_attn_col[(size_t)cid * h + hid] += g; /* :957 atomicAdd */
          }
        }
        butterfly_sum(go);
        grad_row_sum += go[0]; /* :964, lane 0 stores :966 */
      }
      grad_attn_row[rid * h + hid] = grad_row_sum;
    }
  }
  free(grad_edge_csr);
}

void oracle_gat_backward_f64(int m, int nnz, int h, int f, double negative_slope,
                             double attn_drop, const int *row_ptr,
                             const int *col_ind, const float *edge_mask,
                             const float *in_feat, const float *attn_row,
                             const float *attn_col, const float *grad,
                             double *grad_feat, double *grad_attn_row,
                             double *grad_attn_col) {
  /* Same formula, CSR order only, softmax statistics recomputed in double. */
  (void)nnz;
  memset(grad_feat, 0, sizeof(double) * (size_t)m * h * f);
  memset(grad_attn_col, 0, sizeof(double) * (size_t)m * h);
  for (int rid = 0; rid < m; ++rid) {
    for (int hid = 0; hid < h; ++hid) {
      int lb = row_ptr[rid], hb = row_ptr[rid + 1];
      double ar = attn_row[rid * h + hid];
      double mx = (double)(float)-1e38, s = 0;
      for (int p = lb; p < hb; ++p) {
        double w = lrelu_d(ar + (double)attn_col[(size_t)col_ind[p] * h + hid], negative_slope);
        if (w > mx) mx = w;
      }
      for (int p = lb; p < hb; ++p) {
        double w = lrelu_d(ar + (double)attn_col[(size_t)col_ind[p] * h + hid], negative_slope);
        s += exp(w - mx);
      }
      const float *g = grad + (size_t)rid * h * f + hid * f;
      double S = 0;
      for (int p = lb; p < hb; ++p) {
        int cid = col_ind[p];
        double w = lrelu_d(ar + (double)attn_col[(size_t)cid * h + hid], negative_slope);
        double a = exp(w - mx) / s;
        int keep = (edge_mask == NULL) || (edge_mask[(size_t)p * h + hid] > (float)attn_drop);
        if (!keep) continue;
        const float *x = in_feat + (size_t)cid * h * f + hid * f;
        double dot = 0;
        double *gf = grad_feat + (size_t)cid * h * f + hid * f;
        double coef = a / (1.0 - attn_drop);
        for (int c = 0; c < f; ++c) {
          dot += (double)g[c] * (double)x[c];
          gf[c] += coef * (double)g[c];
        }
        S += a * dot / (1.0 - attn_drop);
      }
      double rowsum = 0;
      for (int p = lb; p < hb; ++p) {
        int cid = col_ind[p];
        double w = lrelu_d(ar + (double)attn_col[(size_t)cid * h + hid], negative_slope);
        double a = exp(w - mx) / s;
        int keep = (edge_mask == NULL) || (edge_mask[(size_t)p * h + hid] > (float)attn_drop);
        double ge = 0;
        if (keep) {
          const float *x = in_feat + (size_t)cid * h * f + hid * f;
          double dot = 0;
          for (int c = 0; c < f; ++c) dot += (double)g[c] * (double)x[c];
          ge = dot / (1.0 - attn_drop);
        }
        double d = a * (ge - S);
        if (w < 0) d *= negative_slope;
        rowsum += d;
        grad_attn_col[(size_t)cid * h + hid] += d;
      }
      grad_attn_row[rid * h + hid] = rowsum;
    }
  }
}

/* ------------------------------------------------------------------------- *
 * GE-SpMM style CSR aggregation with caller-supplied edge values.  No
 * reference function of its own: semantics =
 * phase 3 of fused_forward_kernel (fused_gatconv_kernel.cu:87-117) with
 * edge_val[e,h] in place of the softmax weight.
 * ------------------------------------------------------------------------- */
void oracle_spmm_csr_f32(int m, int nnz, int h, int f, const int *row_ptr,
                         const int *col_ind, const float *edge_val,
                         const float *in_feat, float *out_feat) {
  (void)nnz;
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < m; ++rid)
    for (int hid = 0; hid < h; ++hid)
      for (int fid = 0; fid < f; ++fid) {
        float acc = 0;
        for (int p = row_ptr[rid]; p < row_ptr[rid + 1]; ++p)
          acc = fmaf(edge_val[(size_t)p * h + hid],
                     in_feat[(size_t)col_ind[p] * h * f + hid * f + fid], acc);
        out_feat[(size_t)rid * h * f + hid * f + fid] = acc;
      }
}

void oracle_spmm_csr_f64(int m, int nnz, int h, int f, const int *row_ptr,
                         const int *col_ind, const float *edge_val,
                         const float *in_feat, double *out_feat) {
  (void)nnz;
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < m; ++rid)
    for (int hid = 0; hid < h; ++hid)
      for (int fid = 0; fid < f; ++fid) {
        double acc = 0;
        for (int p = row_ptr[rid]; p < row_ptr[rid + 1]; ++p)
          acc += (double)edge_val[(size_t)p * h + hid] *
                 (double)in_feat[(size_t)col_ind[p] * h * f + hid * f + fid];
        out_feat[(size_t)rid * h * f + hid * f + fid] = acc;
      }
}

/* mhtranspose: out[i,:] = in[permute[i],:]   csr2csc/mhtranspose_kernel.cu:6-26 */
void oracle_mhtranspose(int nnz, int h, const int *permute, const float *in,
                        float *out) {
  for (int i = 0; i < nnz; ++i)
    memcpy(out + (size_t)i * h, in + (size_t)permute[i] * h, sizeof(float) * h);
}

/* ------------------------------------------------------------------------- *
 * GMMConv.  fused_gmmconv_kernel.cu: gaussian :11-22, fuseGmm :24-55.
 * Gaussian weight is accumulated in float and exponentiated in DOUBLE
 * (`exp(-0.5 * gauss)` with a double literal, :21/:44) then narrowed.
 * ------------------------------------------------------------------------- */
static inline float gauss_f32(int dim, const float *ps, const float *mu,
                              const float *isg) {
  float acc = 0;
  for (int d = 0; d < dim; ++d) {
    float tmp = ps[d] - mu[d];
    float sig = isg[d];
    acc = fmaf(tmp * tmp * sig, sig, acc); /* :42, as nvcc contracts `gauss += tmp*tmp*sig*sig` */
  }
  return (float)exp(-0.5 * (double)acc);
}
static inline double gauss_f64(int dim, const float *ps, const float *mu,
                               const float *isg) {
  double acc = 0;
  for (int d = 0; d < dim; ++d) {
    double tmp = (double)ps[d] - (double)mu[d];
    double sig = isg[d];
    acc += tmp * tmp * sig * sig;
  }
  return exp(-0.5 * acc);
}

void oracle_gaussian_f32(int edges, int kernels, int dim, const float *pseudo,
                         const float *mu, const float *inv_sigma, float *gauss) {
  for (int e = 0; e < edges; ++e)
    for (int k = 0; k < kernels; ++k)
      gauss[(size_t)e * kernels + k] =
          gauss_f32(dim, pseudo + (size_t)e * dim, mu + k * dim, inv_sigma + k * dim);
}

void oracle_gmm_forward_f32(int nodes, int kernels, int dim, int embed,
                            const int *csrptr, const int *colind,
                            const float *node_feat, const float *pseudo,
                            const float *mu, const float *inv_sigma, float *out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < nodes; ++rid)
    for (int kid = 0; kid < kernels; ++kid)
      for (int fid = 0; fid < embed; ++fid) {
        float acc = 0;
        for (int p = csrptr[rid]; p < csrptr[rid + 1]; ++p) {
          float g = gauss_f32(dim, pseudo + (size_t)p * dim, mu + kid * dim,
                              inv_sigma + kid * dim);
          /* :50  acc += node_feat[...] * shmem[cnt] */
          acc = fmaf(node_feat[(size_t)colind[p] * embed * kernels + kid * embed + fid], g, acc);
        }
        out[(size_t)rid * embed * kernels + kid * embed + fid] = acc;
      }
}

void oracle_gmm_forward_f64(int nodes, int kernels, int dim, int embed,
                            const int *csrptr, const int *colind,
                            const float *node_feat, const float *pseudo,
                            const float *mu, const float *inv_sigma, double *out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < nodes; ++rid)
    for (int kid = 0; kid < kernels; ++kid)
      for (int fid = 0; fid < embed; ++fid) {
        double acc = 0;
        for (int p = csrptr[rid]; p < csrptr[rid + 1]; ++p) {
          double g = gauss_f64(dim, pseudo + (size_t)p * dim, mu + kid * dim,
                               inv_sigma + kid * dim);
          acc += (double)node_feat[(size_t)colind[p] * embed * kernels + kid * embed + fid] * g;
        }
        out[(size_t)rid * embed * kernels + kid * embed + fid] = acc;
      }
}

/* Gaussian backward.  The reference kernel (fused_gmmconv_kernel.cu:57-87) is
 * NOT a usable oracle: its outputs are torch::empty then atomicAdd-ed
 * (:134-136,:80-84), AllReduce takes its argument by value so the warp
 * reduction is a no-op (computeUtil.h:154), the d-summed scalar is added to
 * every d (:82-84) and tmp_pout accumulates cumulatively (:70).  This function
 * restates what the kernel INTENDS = the analytic gradient of
 *   gauss[e,k] = exp(-0.5 * sum_d ((pseudo[e,d]-mu[k,d]) * inv_sigma[k,d])^2)
 * (what DGL's GMMConv autograd computes); pinned by torch autograd in
 * tests/test_oracle.py.  Return order follows :143 {pseudo_out, mu_out, sigma_out}. */
void oracle_gaussian_bp_f64(int edges, int kernels, int dim, const float *pseudo,
                            const float *mu, const float *inv_sigma,
                            const float *grad_gauss, double *pseudo_out,
                            double *mu_out, double *sigma_out) {
  memset(pseudo_out, 0, sizeof(double) * (size_t)edges * dim);
  memset(mu_out, 0, sizeof(double) * (size_t)kernels * dim);
  memset(sigma_out, 0, sizeof(double) * (size_t)kernels * dim);
  for (int e = 0; e < edges; ++e)
    for (int k = 0; k < kernels; ++k) {
      double g = gauss_f64(dim, pseudo + (size_t)e * dim, mu + k * dim, inv_sigma + k * dim) *
                 (double)grad_gauss[(size_t)e * kernels + k];
      for (int d = 0; d < dim; ++d) {
        double sig = inv_sigma[k * dim + d];
        double diff = (double)pseudo[(size_t)e * dim + d] - (double)mu[k * dim + d];
        pseudo_out[(size_t)e * dim + d] += -g * diff * sig * sig;
        mu_out[k * dim + d] += g * diff * sig * sig;
        sigma_out[k * dim + d] += -g * diff * diff * sig;
      }
    }
}

void oracle_gaussian_bp_f32(int edges, int kernels, int dim, const float *pseudo,
                            const float *mu, const float *inv_sigma,
                            const float *grad_gauss, float *pseudo_out,
                            float *mu_out, float *sigma_out) {
  memset(pseudo_out, 0, sizeof(float) * (size_t)edges * dim);
  memset(mu_out, 0, sizeof(float) * (size_t)kernels * dim);
  memset(sigma_out, 0, sizeof(float) * (size_t)kernels * dim);
  for (int e = 0; e < edges; ++e)
    for (int k = 0; k < kernels; ++k) {
      float g = gauss_f32(dim, pseudo + (size_t)e * dim, mu + k * dim, inv_sigma + k * dim) *
                grad_gauss[(size_t)e * kernels + k];
      for (int d = 0; d < dim; ++d) {
        float sig = inv_sigma[k * dim + d];
        float diff = pseudo[(size_t)e * dim + d] - mu[k * dim + d];
        pseudo_out[(size_t)e * dim + d] += -g * diff * sig * sig;
        mu_out[k * dim + d] += g * diff * sig * sig;
        sigma_out[k * dim + d] += -g * diff * diff * sig;
      }
    }
}

/* ------------------------------------------------------------------------- *
 * EdgeConv.  fused_edgeconv_kernel.cu: forward :7-49, backward :76-86.
 * The arg-max tie rule is the literal 32-lane xor butterfly with `tmp >= feat`
 * taking the partner (:31-38); later 32-tiles win ties (`feat >= max_val`,
 * :39); padding lanes carry (-1e38, -1) (:24-25); max_val starts at -1e39
 * which is -inf in float (:19).  max_idx is returned as float (:46,:61).
 * ------------------------------------------------------------------------- */
void oracle_edgeconv_forward(int m, int f, int k, const int *src_ind,
                             const float *h_src, const float *h_dst,
                             float *out_feat, float *max_idx) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int rid = 0; rid < m; ++rid) {
    int lb = rid * k, hb = (rid + 1) * k;
    int loop = (k + 31) / 32;
    for (int fid = 0; fid < f; ++fid) {
      float max_val = -INFINITY; /* (float)-1e39 */
      float max_id = -1;
      float dst_val = h_dst[(size_t)rid * f + fid];
      for (int j = 0; j < loop; ++j) {
        float feat[LANES], tf[LANES];
        int sid[LANES], ts[LANES];
        for (int tx = 0; tx < LANES; ++tx) {
          int pid = lb + tx + (j << 5);
          feat[tx] = -1e38;
          sid[tx] = -1;
          if (pid < hb) {
            sid[tx] = src_ind[pid];
            feat[tx] = dst_val + h_src[(size_t)sid[tx] * f + fid];
          }
        }
        for (int stride = 16; stride > 0; stride >>= 1) {
          for (int l = 0; l < LANES; ++l) {
            tf[l] = feat[l ^ stride];
            ts[l] = (int)(float)sid[l ^ stride]; /* float tmp_id = shfl(int) :33 */
          }
          for (int l = 0; l < LANES; ++l)
            if (tf[l] >= feat[l]) {
              feat[l] = tf[l];
              sid[l] = ts[l];
            }
        }
        if (feat[0] >= max_val) { /* every lane holds its own copy; lane 0 stores */
          max_val = feat[0];
          max_id = (float)sid[0];
        }
      }
      out_feat[(size_t)rid * f + fid] = max_val;
      max_idx[(size_t)rid * f + fid] = max_id;
    }
  }
}

/* Backward as the reference INTENDS it (zero, then scatter-add); its kernel
 * zero-fills and accumulates in one launch without a grid sync (:83-84), a race. */
void oracle_edgeconv_backward_f32(int m, int f, const int *max_idx,
                                  const float *grad_out, float *grad_src) {
  memset(grad_src, 0, sizeof(float) * (size_t)m * f);
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < f; ++c) {
      int s = max_idx[(size_t)r * f + c];
      if (s >= 0) grad_src[(size_t)s * f + c] += grad_out[(size_t)r * f + c];
    }
}
void oracle_edgeconv_backward_f64(int m, int f, const int *max_idx,
                                  const float *grad_out, double *grad_src) {
  memset(grad_src, 0, sizeof(double) * (size_t)m * f);
  for (int r = 0; r < m; ++r)
    for (int c = 0; c < f; ++c) {
      int s = max_idx[(size_t)r * f + c];
      if (s >= 0) grad_src[(size_t)s * f + c] += (double)grad_out[(size_t)r * f + c];
    }
}

/* ------------------------------------------------------------------------- *
 * Index work (bit-exact).  convert_kernel.cu:6-59 calls cuSPARSE csr2csc /
 * coo2csr; the training scripts do the same on the host with scipy
 * (script/train/train_gatconv.py:78-94).  Both are a STABLE counting sort by
 * column: CSC entries of a column keep ascending row order; `permute[i]` is
 * the CSR position of the i-th CSC entry (csr_data = arange).
 * ------------------------------------------------------------------------- */
void oracle_csr2csc(int m, int n, int nnz, const int *row_ptr, const int *col_ind,
                    int *col_ptr, int *row_ind, int *permute) {
  memset(col_ptr, 0, sizeof(int) * (size_t)(n + 1));
  for (int e = 0; e < nnz; ++e) col_ptr[col_ind[e] + 1]++;
  for (int c = 0; c < n; ++c) col_ptr[c + 1] += col_ptr[c];
  int *cur = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  memcpy(cur, col_ptr, sizeof(int) * (size_t)n);
  for (int r = 0; r < m; ++r)
    for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
      int pos = cur[col_ind[e]]++;
      row_ind[pos] = r;
      permute[pos] = e;
    }
  free(cur);
}

/* cusparseXcoo2csr on a sorted COO row array: row_ptr[r] = #entries with row < r */
void oracle_coo2csr(int m, int nnz, const int *coo_row, int *row_ptr) {
  memset(row_ptr, 0, sizeof(int) * (size_t)(m + 1));
  for (int e = 0; e < nnz; ++e) row_ptr[coo_row[e] + 1]++;
  for (int r = 0; r < m; ++r) row_ptr[r + 1] += row_ptr[r];
}

/* ---------------------------------------------------------------------------
 * k-nearest-neighbour graph in front of EdgeConv.  The reference takes it from DGL
 * (KNNGraph(k)(h); src, dst = g.edges(): script/train/train_edgeconv.py:17,49-52; dgl >= 0.7,
 * README.md:29 -- third-party, not under the reference tree).  DGL's published algorithm
 * (bruteforce): pairwise squared distances inside each cloud, the k smallest per point
 * (the point itself included), src = neighbour, dst = point repeated k times.  Restated with
 * a pinned float32 distance chain (f ascending, fmaf) and a total order (distance, then id),
 * which fixes what DGL leaves to topk's tie handling.  Index parity with DGL itself: unpinned.
 * ------------------------------------------------------------------------- */
void oracle_knn(int batch, int points, int dim, int k, const float *x, int *src_ind) {
#pragma omp parallel
  {
    float *d = (float *)malloc(sizeof(float) * (size_t)(points > 0 ? points : 1));
#pragma omp for schedule(static)
    for (long long q = 0; q < (long long)batch * points; ++q) {
      const int b = (int)(q / points);
      const float *xc = x + (size_t)b * points * dim;
      const float *xi = x + (size_t)q * dim;
      for (int j = 0; j < points; ++j) {
        float acc = 0.f;
        for (int f = 0; f < dim; ++f) {
          const float t = xi[f] - xc[(size_t)j * dim + f];
          acc = fmaf(t, t, acc);
        }
        d[j] = acc;
      }
      /* k rounds of selection: smallest (distance, id) not yet taken */
      float last_d = -1.f;
      int last_j = -1;
      for (int t = 0; t < k; ++t) {
        int bj = -1;
        float bd = 0.f;
        for (int j = 0; j < points; ++j) {
          const float dj = d[j];
          if (!(dj < INFINITY)) continue; /* NaN or overflowed distance: never a neighbour */
          if (dj < last_d || (dj == last_d && j <= last_j)) continue; /* already taken */
          if (bj < 0 || dj < bd) { bj = j; bd = dj; }
        }
        /* fewer than k finite distances (a diverged model: NaN / inf features): the remaining
         * slots name the query itself, so downstream gathers stay in range and the NaNs travel
         * on as VALUES */
        src_ind[(size_t)q * k + t] = bj < 0 ? (int)q : b * points + bj;
        if (bj < 0) continue;
        last_d = bd;
        last_j = bj;
      }
    }
    free(d);
  }
}

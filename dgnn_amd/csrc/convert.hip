// Device-side index work: CSR -> CSC with permutation (stable), sorted COO -> CSR.
// Replaces the cuSPARSE calls of dgNN/src/convert/convert_kernel.cu:6-59 and the host scipy
// path of script/train/train_gatconv.py:87-94.  The stable sort by column is rocPRIM's LSD
// radix sort (index plumbing, not a hot kernel); everything else is hand-written.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "common.hpp"

using namespace dgnn;

namespace dgnn {

__device__ __forceinline__ int lower_bound_i32(const int *__restrict__ a, int n, int key) {
  int lo = 0, hi = n;  // first index with a[idx] >= key
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ptr[c] = number of sorted keys < c, for c in [0, n]
__global__ void k_ptr_from_sorted(int n, int nnz, const int *__restrict__ sorted_keys,
                                  int *__restrict__ ptr) {
  for (long long c = blockIdx.x * (long long)blockDim.x + threadIdx.x; c <= n;
       c += (long long)gridDim.x * blockDim.x)
    ptr[c] = lower_bound_i32(sorted_keys, nnz, (int)c);
}

// rows_csr[p] = row that owns CSR position p.  One wave per 64 consecutive rows: the lanes load the
// 65 row bounds once and then fill row after row with coalesced stores (a hub row of 10^5 entries
// is 1600 store instructions of one wave; short rows cost one instruction each).
__global__ __launch_bounds__(DGNN_BLOCK) void k_expand_rows(int m, const int *__restrict__ row_ptr,
                                                            int *__restrict__ rows_csr) {
  const int lane = lane_id();
  const long long r0 = (long long)(blockIdx.x * DGNN_WAVES_PER_BLOCK + wave_in_block()) * DGNN_WAVE;
  if (r0 >= m) return;
  const int lo = row_ptr[min((long long)m, r0 + lane)];
  const int hi = row_ptr[min((long long)m, r0 + lane + 1)];
  const int nrows = (int)min((long long)DGNN_WAVE, m - r0);
  for (int rr = 0; rr < nrows; ++rr) {
    const int s = __shfl(lo, rr, DGNN_WAVE), e = __shfl(hi, rr, DGNN_WAVE);
    for (int p = s + lane; p < e; p += DGNN_WAVE) rows_csr[p] = (int)(r0 + rr);
  }
}

// row_ind[i] = rows_csr[permute[i]]  (replaces a 22-step binary search over row_ptr per entry:
// 3.9 ms and 14.7 GB of L2 traffic for 62 M entries, profiles/r2d_*_products.csv)
__global__ void k_gather_rows(int nnz, const int *__restrict__ permute, const int *__restrict__ rows_csr,
                              int *__restrict__ row_ind) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz;
       i += (long long)gridDim.x * blockDim.x)
    row_ind[i] = rows_csr[permute[i]];
}

// row_ind[i] = permute[i] / k: the row of a CSR position when every row holds exactly k entries
// (k-NN lists): no row_ptr, no expanded row array.
__global__ void k_rows_uniform(int nnz, int k, const int *__restrict__ permute, int *__restrict__ row_ind) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nnz;
       i += (long long)gridDim.x * blockDim.x)
    row_ind[i] = permute[i] / k;
}

static inline int grid_1d(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 32;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

static inline int key_bits(int n) {
  int b = 1;
  while (b < 31 && (1LL << b) < (long long)n) ++b;
  return b;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static hipError_t sort_temp_bytes(int nnz, int bits, size_t *bytes) {
  *bytes = 0;
  return rocprim::radix_sort_pairs(nullptr, *bytes, (const int *)nullptr, (int *)nullptr,
                                   rocprim::counting_iterator<int>(0), (int *)nullptr,
                                   (size_t)nnz, 0, bits, (hipStream_t)0);
}

// Transpose of m lists of exactly k column ids each (the k-NN lists in front of EdgeConv): col_ptr /
// row_ind of the CSC form, rows ascending inside a column.  dgnn_csr2csc without the row_ptr
// array, the expanded rows and the permutation as an output: sort, pointer search, one divide
// per entry.  Workspace: dgnn_csr2csc_workspace_bytes(m, m, m * k) + 4 * m * k bytes.
int transpose_uniform_lists(int m, int k, const int *col_ind, int *col_ptr, int *row_ind, void *workspace,
                            size_t workspace_bytes, hipStream_t st) {
  const long long nnz64 = (long long)m * k;
  if (m < 0 || k < 0 || nnz64 > 0x7fffffffLL || !col_ptr) return DGNN_ERR_BAD_ARG;
  const int nnz = (int)nnz64;
  if (nnz == 0) {
    hipError_t e = hipMemsetAsync(col_ptr, 0, sizeof(int) * ((size_t)m + 1), st);
    return e == hipSuccess ? DGNN_OK : (int)e;
  }
  if (!col_ind || !row_ind) return DGNN_ERR_BAD_ARG;
  const int bits = key_bits(m);
  size_t temp = 0;
  hipError_t e = sort_temp_bytes(nnz, bits, &temp);
  if (e != hipSuccess) return (int)e;
  const size_t keys_bytes = align_up(sizeof(int) * (size_t)nnz, 256);
  if (!workspace || workspace_bytes < 2 * keys_bytes + align_up(temp, 256)) return DGNN_ERR_WORKSPACE;
  int *sorted_keys = (int *)workspace;
  int *permute = (int *)((char *)workspace + keys_bytes);
  void *sort_tmp = (char *)workspace + 2 * keys_bytes;
  e = rocprim::radix_sort_pairs(sort_tmp, temp, col_ind, sorted_keys, rocprim::counting_iterator<int>(0), permute,
                                (size_t)nnz, 0, bits, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_ptr_from_sorted, dim3(grid_1d((long long)m + 1, 256)), dim3(256), 0, st, m, nnz,
                     (const int *)sorted_keys, col_ptr);
  DGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_rows_uniform, dim3(grid_1d(nnz, 256)), dim3(256), 0, st, nnz, k, (const int *)permute, row_ind);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}
size_t transpose_uniform_lists_workspace_bytes(int m, int k) {
  const long long nnz64 = (long long)m * k;
  if (m <= 0 || k <= 0 || nnz64 > 0x7fffffffLL) return 0;
  size_t temp = 0;
  if (sort_temp_bytes((int)nnz64, key_bits(m), &temp) != hipSuccess) return 0;
  return 2 * align_up(sizeof(int) * (size_t)nnz64, 256) + align_up(temp, 256);
}

}  // namespace dgnn

extern "C" {

size_t dgnn_csr2csc_workspace_bytes(int m, int n, int nnz) {
  (void)m;
  if (nnz <= 0 || n <= 0) return 0;
  size_t temp = 0;
  if (sort_temp_bytes(nnz, key_bits(n), &temp) != hipSuccess) return 0;
  // sorted keys; then the sort's scratch and, after the sort, the row of every CSR position share a region
  const size_t after = align_up(temp, 256) > align_up(sizeof(int) * (size_t)nnz, 256) ? align_up(temp, 256)
                                                                                       : align_up(sizeof(int) * (size_t)nnz, 256);
  return align_up(sizeof(int) * (size_t)nnz, 256) + after;
}

int dgnn_csr2csc(int m, int n, int nnz, const int *row_ptr, const int *col_ind, int *col_ptr,
                 int *row_ind, int *permute, void *workspace, size_t workspace_bytes,
                 void *stream) {
  if (m < 0 || n < 0 || nnz < 0) return DGNN_ERR_BAD_ARG;
  if (!row_ptr || !col_ptr) return DGNN_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (nnz == 0) {
    hipError_t e = hipMemsetAsync(col_ptr, 0, sizeof(int) * ((size_t)n + 1), st);
    return e == hipSuccess ? DGNN_OK : (int)e;
  }
  if (!col_ind || !row_ind || !permute) return DGNN_ERR_BAD_ARG;
  const int bits = key_bits(n);
  size_t temp = 0;
  hipError_t e = sort_temp_bytes(nnz, bits, &temp);
  if (e != hipSuccess) return (int)e;
  const size_t keys_bytes = align_up(sizeof(int) * (size_t)nnz, 256);
  if (!workspace || workspace_bytes < dgnn_csr2csc_workspace_bytes(m, n, nnz)) return DGNN_ERR_WORKSPACE;
  int *sorted_keys = (int *)workspace;
  void *sort_tmp = (char *)workspace + keys_bytes;
  // stable sort of (column, CSR position): positions of one column stay ascending, i.e. rows ascending
  e = rocprim::radix_sort_pairs(sort_tmp, temp, col_ind, sorted_keys,
                                rocprim::counting_iterator<int>(0), permute, (size_t)nnz, 0, bits, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_ptr_from_sorted, dim3(grid_1d((long long)n + 1, 256)), dim3(256), 0, st, n,
                     nnz, (const int *)sorted_keys, col_ptr);
  DGNN_LAUNCH_CHECK();
  int *rows_csr = (int *)sort_tmp;          // the sort is done (stream order): its scratch is free
  hipLaunchKernelGGL(k_expand_rows, dim3(blocks_for_waves(((long long)m + DGNN_WAVE - 1) / DGNN_WAVE)),
                     dim3(DGNN_BLOCK), 0, st, m, row_ptr, rows_csr);
  DGNN_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_gather_rows, dim3(grid_1d(nnz, 256)), dim3(256), 0, st, nnz, (const int *)permute,
                     (const int *)rows_csr, row_ind);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

int dgnn_coo2csr(int m, int nnz, const int *coo_row, int *row_ptr, void *stream) {
  if (m < 0 || nnz < 0 || !row_ptr || (nnz > 0 && !coo_row)) return DGNN_ERR_BAD_ARG;
  hipLaunchKernelGGL(k_ptr_from_sorted, dim3(grid_1d((long long)m + 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, m, nnz, coo_row, row_ptr);
  DGNN_LAUNCH_CHECK();
  return DGNN_OK;
}

}  // extern "C"

// Probe (round 3): are 16-byte global loads / stores at 4-byte-aligned addresses safe and fast on
// gfx950?  Random whole-row gathers of a [n, hf] float table, one wave per destination row, each lane
// one float4 slot: (a) hf = 164 floats (rows 4-byte aligned only, unaligned dwordx4), (b) hf = 176
// (the same rows padded to 16 bytes, aligned dwordx4), (c) hf = 164 with scalar lanes.
// Prints ms and GB/s per form and checks (a) against (c).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>   // 0: unaligned float4, 1: aligned float4, 2: scalar
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, const int* __restrict__ idx, float* __restrict__ out,
                                                 int rows, int deg, int hf, int hf_used) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  if (MODE == 2) {
    for (int f = lane; f < hf_used; f += 64) {
      float acc = 0.f;
      for (int j = 0; j < deg; ++j) acc += tab[(size_t)idx[(size_t)r * deg + j] * hf + f];
      out[(size_t)r * hf + f] = acc;
    }
    return;
  }
  const int nslots = (hf_used + 3) / 4;
  if (lane >= nslots) return;
  int off = lane * 4;
  if (off + 4 > hf_used) off = hf_used - 4;      // tail slot: overlaps its predecessor, same values
  f4u acc = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < deg; j += 4) {
    f4u x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = *(const f4u*)(tab + (size_t)idx[(size_t)r * deg + j + u] * hf + off);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += x[u];
  }
  *(f4u*)(out + (size_t)r * hf + off) = acc;
}

int main() {
  const int n = 2449029, rows = 1 << 20, deg = 16, used = 163;
  for (int pass = 0; pass < 2; ++pass) {
    const int hf_used = pass == 0 ? 164 : used;      // 164: slots aligned inside a row but rows only 4-byte aligned when hf is odd... (163: odd rows)
    const int hfs[3] = {hf_used, (hf_used + 3) / 4 * 4 + (pass == 0 ? 12 : 0), hf_used};
    std::vector<int> hidx((size_t)rows * deg);
    unsigned s = 12345u;
    for (auto& v : hidx) { s = s * 1664525u + 1013904223u; v = (int)((s >> 4) % (unsigned)n); }
    int* idx; CK(hipMalloc(&idx, hidx.size() * 4)); CK(hipMemcpy(idx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
    float *tab[3], *out[3];
    for (int m = 0; m < 3; ++m) {
      CK(hipMalloc(&tab[m], (size_t)n * hfs[m] * 4 + 64)); CK(hipMalloc(&out[m], (size_t)rows * hfs[m] * 4 + 64));
      std::vector<float> h((size_t)n * hfs[m]);
      for (size_t i = 0; i < h.size(); ++i) { const size_t row = i / hfs[m], f = i % hfs[m]; h[i] = f < (size_t)hf_used ? (float)((row * 31 + f * 7) % 1000) * 0.001f : 0.f; }
      CK(hipMemcpy(tab[m], h.data(), h.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemset(out[m], 0, (size_t)rows * hfs[m] * 4));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"unaligned float4", "aligned float4 (padded rows)", "scalar lanes"};
    for (int m = 0; m < 3; ++m) {
      float best = 1e9f;
      for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        if (m == 0) hipLaunchKernelGGL(k_gather<0>, dim3(rows / 4), dim3(256), 0, 0, tab[m], idx, out[m], rows, deg, hfs[m], hf_used);
        if (m == 1) hipLaunchKernelGGL(k_gather<1>, dim3(rows / 4), dim3(256), 0, 0, tab[m], idx, out[m], rows, deg, hfs[m], hf_used);
        if (m == 2) hipLaunchKernelGGL(k_gather<2>, dim3(rows / 4), dim3(256), 0, 0, tab[m], idx, out[m], rows, deg, hfs[m], hf_used);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("hf_used=%d hf=%d %-32s %.3f ms  %.1f GB/s (used bytes)\n", hf_used, hfs[m], names[m], best,
             (double)rows * deg * hf_used * 4 / best / 1e6);
    }
    std::vector<float> a((size_t)rows * hfs[0]), c((size_t)rows * hfs[2]);
    CK(hipMemcpy(a.data(), out[0], a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c.data(), out[2], c.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) if (a[i] != c[i]) ++bad;
    printf("hf_used=%d unaligned float4 vs scalar: %zu mismatches of %zu\n", hf_used, bad, a.size());
    for (int m = 0; m < 3; ++m) { CK(hipFree(tab[m])); CK(hipFree(out[m])); }
    CK(hipFree(idx));
  }
  return 0;
}

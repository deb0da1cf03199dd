// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libdgnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/dgnn_hip.h"

#define DGNN_WAVE 64
#define DGNN_BLOCK 256                       // 4 waves per workgroup
#define DGNN_WAVES_PER_BLOCK (DGNN_BLOCK / DGNN_WAVE)
#define DGNN_NEG_INIT (-1e38f)               // fused_gatconv_kernel.cu:45  `float weightMax = -1e38`

#define DGNN_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t e__ = hipGetLastError();             \
    if (e__ != hipSuccess) return (int)e__;         \
  } while (0)

namespace dgnn {

__device__ __forceinline__ int lane_id() { return threadIdx.x & (DGNN_WAVE - 1); }
__device__ __forceinline__ int wave_in_block() { return threadIdx.x >> 6; }

// 16-byte global accesses at 4-BYTE-aligned addresses.  Feature rows of H*F floats are 16-byte
// aligned only when H*F is a multiple of four; gfx950 serves global_load/store_dwordx4 at any
// dword-aligned address at the rate of aligned ones (tools/probe_unaligned.hip: random 656-byte
// rows 4.90 TB/s unaligned against 4.74 TB/s for the same rows padded to 704 bytes, scalar lanes
// 1.93 TB/s), so every width of at least four floats takes the float4 lanes: a head's last slot, when
// F is not a multiple of four, is the four floats that END at the head's end and overlaps its
// predecessor (both lanes then hold identical values for the shared floats).  The type only
// drops the compiler's alignment assumption; the instruction is the same dwordx4.
typedef float dgnn_f4u __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld4u(const float *p) {
  const dgnn_f4u v = *(const dgnn_f4u *)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4u(float *p, const float4 &v) {
  dgnn_f4u t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  *(dgnn_f4u *)p = t;
}
// float offset of vector slot q (of ceil(F/4)) inside a head of F >= 4 floats, and the number of
// its leading components that repeat the previous slot (0 except for the last slot of an F that
// is not a multiple of four): dot products skip those, sums and stores may keep them
__device__ __forceinline__ int slot4_off(int q, int F) { return min(q * 4, F - 4); }
__device__ __forceinline__ int slot4_dup(int q, int F) { return max(q * 4 - (F - 4), 0); }
__device__ __forceinline__ float4 slot4_mask(float4 v, int dup) {
  if (dup > 0) v.x = 0.f;
  if (dup > 1) v.y = 0.f;
  if (dup > 2) v.z = 0.f;
  return v;
}

// fused_gatconv_kernel.cu:13  LeakyRelu(x, s) = x > 0 ? x : x*s
__device__ __forceinline__ float leaky_relu(float x, float s) { return (x > 0.f) ? x : x * s; }

// ---- exp and divide of the softmax weights ------------------------------------------------
// The weights are exp(e - max) / sum with e - max <= 0 (fused_gatconv_kernel.cu:75,102).  libm's
// expf spends half of its ~15 instructions on overflow and denormal-input handling that cannot
// occur here, and an IEEE divide is ~10; per (edge, head) of a narrow row that is a large share of
// the kernel's vector instructions (8 x 8 forward: the softmax phase is ~60 % of them).  These forms
// keep the accuracy class of the reference's own expf / division (each within ~1.5 ulp; CUDA's
// expf is specified to 2 ulp): exp2 of the exactly reduced argument through v_exp_f32 + v_ldexp_f32,
// and a multiply by v_rcp_f32 (1 ulp).  -DDGNN_EXACT_EXPDIV=1 builds with expf and '/' (A/B runs).
__device__ __forceinline__ float exp_nonpos(float x) {
#if defined(DGNN_EXACT_EXPDIV) && DGNN_EXACT_EXPDIV
  return expf(x);
#else
  const float l2e_hi = 1.44269502162933349609375f;      // float(log2 e)
  const float l2e_lo = 1.925963033500011e-08f;          // log2 e - float(log2 e)
  const float n = rintf(x * l2e_hi);
  float r = fmaf(x, l2e_hi, -n);                         // exact product, so |r| <= 0.5 + 2^-24 |n|
  r = fmaf(x, l2e_lo, r);
  return ldexpf(__builtin_amdgcn_exp2f(r), (int)n);      // x far below -150: the int conversion saturates, ldexp gives 0
#endif
}
__device__ __forceinline__ float div_pos(float a, float b) {   // a / b for finite b > 0
#if defined(DGNN_EXACT_EXPDIV) && DGNN_EXACT_EXPDIV
  return a / b;
#else
  return a * __builtin_amdgcn_rcpf(b);
#endif
}

// ---- cross-lane reductions over the full 64-lane wave -----------------------------------
// DPP row operations do the 16-lane part in the VALU (no LDS traffic); the four row results
// are then combined through v_readlane.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define DGNN_DPP_QUAD_XOR1 0xB1   // quad_perm [1,0,3,2]
#define DGNN_DPP_QUAD_XOR2 0x4E   // quad_perm [2,3,0,1]
#define DGNN_DPP_ROW_HALF_MIRROR 0x141
#define DGNN_DPP_ROW_MIRROR 0x140

struct OpSum {
  __device__ __forceinline__ float operator()(float a, float b) const { return a + b; }
};
struct OpMax {
  __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); }
};

// max(v, v permuted inside its DPP row) as ONE v_max_f32 with a DPP source operand.  Written through
// fmaxf the compiler cannot prove the permuted value is no signalling NaN and emits the move plus two
// canonicalising v_max before the real one (four instructions per step; the 16-lane max of every head
// of every softmax pass is four such steps).  The s_nop covers the two wait states a DPP read needs
// after the VALU write of its source (the hazard recogniser does not look inside inline assembly).
#define DGNN_DPP_MAX_STEP(name, ctrl)                                                                \
  __device__ __forceinline__ float name(float v) {                                                   \
    float r;                                                                                         \
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); \
    return r;                                                                                        \
  }
DGNN_DPP_MAX_STEP(dpp_max_xor1, "quad_perm:[1,0,3,2]")
DGNN_DPP_MAX_STEP(dpp_max_xor2, "quad_perm:[2,3,0,1]")
DGNN_DPP_MAX_STEP(dpp_max_half_mirror, "row_half_mirror")
DGNN_DPP_MAX_STEP(dpp_max_mirror, "row_mirror")
#undef DGNN_DPP_MAX_STEP
__device__ __forceinline__ float row16_reduce(float v, OpMax) {
  return dpp_max_mirror(dpp_max_half_mirror(dpp_max_xor2(dpp_max_xor1(v))));
}

// Reduce inside each aligned group of 16 lanes (one DPP row); every lane of the row gets the result.
template <class Op>
__device__ __forceinline__ float row16_reduce(float v, Op op) {
  v = op(v, dpp_f32<DGNN_DPP_QUAD_XOR1>(v));
  v = op(v, dpp_f32<DGNN_DPP_QUAD_XOR2>(v));
  v = op(v, dpp_f32<DGNN_DPP_ROW_HALF_MIRROR>(v));
  v = op(v, dpp_f32<DGNN_DPP_ROW_MIRROR>(v));
  return v;
}

// Full-wave reduction; the result is wave-uniform.
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
  v = row16_reduce(v, op);
  float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce(v, OpSum()); }
__device__ __forceinline__ float wave_max(float v) { return wave_reduce(v, OpMax()); }

// Reduce over aligned groups of W lanes (W power of two, 1..64); every lane gets its group's result.
template <class Op>
__device__ __forceinline__ float group_reduce(float v, int w, Op op) {
  if (w >= 2) v = op(v, dpp_f32<DGNN_DPP_QUAD_XOR1>(v));
  if (w >= 4) v = op(v, dpp_f32<DGNN_DPP_QUAD_XOR2>(v));
  if (w >= 8) v = op(v, dpp_f32<DGNN_DPP_ROW_HALF_MIRROR>(v));
  if (w >= 16) v = op(v, dpp_f32<DGNN_DPP_ROW_MIRROR>(v));
  if (w >= 32) v = op(v, __shfl_xor(v, 16, DGNN_WAVE));
  if (w >= 64) v = op(v, __shfl_xor(v, 32, DGNN_WAVE));
  return v;
}

// Broadcast lane `src` (wave-uniform index) of v to every lane.
__device__ __forceinline__ float bcast_f32(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ int bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// Make a value the compiler cannot prove uniform into an SGPR value (it IS uniform by construction).
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Logical workgroup id for L2 affinity.  Consecutive hardware workgroup ids are observed to land
// on consecutive XCDs (MI355X_MICROARCH.md, workgroup dispatch), each with its own 4 MiB L2, so
// neighbours in id order share nothing.  This bijection hands every residue class of ids mod 8
// one contiguous range of logical ids: workgroups that read the same rows (one point cloud's
// queries, one graph band) then share an L2.  A speed choice only -- any placement is correct.
__device__ __forceinline__ unsigned xcd_contiguous_id(unsigned orig, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, x = orig & 7u;
  const unsigned base = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (orig >> 3);
}
__device__ __forceinline__ float uniform_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Order LDS traffic between lanes of ONE wave (no workgroup barrier is needed: each wave owns
// its LDS slice).  Emits at most an s_waitcnt.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Dropout scaling exactly as the reference writes it: weight / (1.0 - attn_drop) in double
// (fused_gatconv_kernel.cu:104).
__device__ __forceinline__ float drop_scale(float w, float attn_drop) {
  return (float)((double)w / (1.0 - (double)attn_drop));
}

// ---- counter-based dropout mask ----------------------------------------------------------
// The reference draws edge_mask [nnz,h] ~ U(0,1] with cuRAND on every forward and keeps it for
// the backward (fused_gatconv_kernel.cu:333-341; 1 GB at the products shape).  Here element i of
// the (virtual) row-major mask is component i % 4 of Philox4x32-10 at counter (i / 4, offset)
// under key `seed`: the forward (CSR position), the CSC backward (through permute) and the
// materialising kernel of dgnn_dropout_mask regenerate identical values, nothing is stored.
struct DropRng {
  unsigned long long seed, offset;
  int on;
  // optional DEVICE pair {seed, offset} read by the kernel at launch instead of the two values
  // above: a launch captured into a HIP graph then draws from whatever the pair holds at each
  // replay (the *_seeded_dev entry points), not from values frozen into the kernel arguments
  const unsigned long long *dev;
};
__device__ __forceinline__ void drop_resolve(DropRng &r) {
  if (r.dev != nullptr) {
    r.seed = r.dev[0];
    r.offset = r.dev[1];
  }
}
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}
// 24 random bits -> (0, 1], exactly representable: never 0, so `mask > attn_drop` keeps
// everything at attn_drop = 0 like the reference's (0,1] floats.
__device__ __forceinline__ float bits_to_unit(unsigned x) { return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f); }
// mask elements 4*block .. 4*block+3
__device__ __forceinline__ float4 drop_uniform4(const DropRng &r, unsigned long long block) {
  const uint4 v = philox4x32_10(make_uint4((unsigned)block, (unsigned)(block >> 32), (unsigned)r.offset,
                                           (unsigned)(r.offset >> 32)),
                                make_uint2((unsigned)r.seed, (unsigned)(r.seed >> 32)));
  return make_float4(bits_to_unit(v.x), bits_to_unit(v.y), bits_to_unit(v.z), bits_to_unit(v.w));
}
// mask element i
__device__ __forceinline__ float drop_uniform(const DropRng &r, unsigned long long i) {
  const float4 v = drop_uniform4(r, i >> 2);
  const unsigned c = (unsigned)i & 3u;
  return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}

// Gaussian mixture weight of one (edge, kernel) pair.
__device__ __forceinline__ float gmm_weight(int dim, const float *__restrict__ ps,
                                            const float *__restrict__ mu,
                                            const float *__restrict__ isg) {
  // fuseGmm, fused_gmmconv_kernel.cu:38-44: float accumulation, exponent in double
  float acc = 0.f;
  for (int d = 0; d < dim; ++d) {
    const float tmp = ps[d] - mu[d];
    const float sig = isg[d];
    acc = fmaf(tmp * tmp * sig, sig, acc);
  }
  return (float)exp(-0.5 * (double)acc);
}

// ---- diagnostic build only (-DDGNN_PHASE_CLOCK, tools/phase_clock.sh): where do the waves of the
// two gather kernels spend their time?  Each wave adds the shader-clock cycles between phase
// boundaries to a device table read back with dgnn_debug_phase_cycles().  No stamp exists in the
// product build.
#ifdef DGNN_PHASE_CLOCK
static __device__ unsigned long long g_phase_cycles[2][8];
#define DGNN_PHASE_DECL() unsigned long long pc_t = __builtin_readcyclecounter()
#define DGNN_PHASE_MARK(kernel, phase)                                                     \
  do {                                                                                     \
    const unsigned long long pc_n = __builtin_readcyclecounter();                          \
    if (lane_id() == 0) atomicAdd(&g_phase_cycles[kernel][phase], pc_n - pc_t);            \
    pc_t = pc_n;                                                                           \
  } while (0)
#else
#define DGNN_PHASE_DECL() do { } while (0)
#define DGNN_PHASE_MARK(kernel, phase) do { } while (0)
#endif

namespace host {
// Optional timing of the dominant kernels (DGNN_TIMING=1): HIP events recorded on the launch stream
// right around the kernel, read back with dgnn_debug_last_kernel_ms().  Slot 0 = the forward-side
// kernel launched last (row gather of the fused GAT / SpMM / GMMConv forward, fused EdgeConv forward),
// slot 1 = the backward-side one (column gather of the GAT / GMMConv backward, EdgeConv backward
// gather).  Off by default; this is the only process-global state in the library and it never
// changes results.
struct KernelTimer {
  bool on = false, init = false;
  hipEvent_t ev[2][2];
  bool valid[2] = {false, false};
  long long launches[2] = {0, 0};  // launches of the timed kernel since the library was loaded
  char name[2][96] = {{0}, {0}};   // template instantiation of the kernel last timed, as rocprofv3 prints it
};
extern KernelTimer g_timer;      // defined in gather.hip
inline bool timing_on() {
  if (!g_timer.init) {
    g_timer.init = true;
    { const char *e = getenv("DGNN_TIMING"); g_timer.on = e != nullptr && e[0] != 0 && atoi(e) != 0; }
    if (g_timer.on)
      for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 2; ++i)
          if (hipEventCreate(&g_timer.ev[k][i]) != hipSuccess) g_timer.on = false;
  }
  return g_timer.on;
}
inline void timer_name(int which, const char *name) {
  if (timing_on()) snprintf(g_timer.name[which], sizeof(g_timer.name[which]), "%s", name);
}
inline void timer_mark(int which, int edge, hipStream_t st) {
  if (timing_on()) {
    (void)hipEventRecord(g_timer.ev[which][edge], st);
    if (edge == 1) {
      g_timer.valid[which] = true;
      ++g_timer.launches[which];
    }
  }
}

}  // namespace host

static inline int blocks_for_waves(long long n_waves) {
  return (int)((n_waves + DGNN_WAVES_PER_BLOCK - 1) / DGNN_WAVES_PER_BLOCK);
}

// convert.hip: CSC form (col_ptr [m+1], row_ind [m*k], rows ascending inside a column) of m lists of
// exactly k column ids each
int transpose_uniform_lists(int m, int k, const int *col_ind, int *col_ptr, int *row_ind, void *workspace,
                            size_t workspace_bytes, hipStream_t st);
size_t transpose_uniform_lists_workspace_bytes(int m, int k);

}  // namespace dgnn

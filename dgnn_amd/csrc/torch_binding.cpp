// PyTorch-ROCm C++ extension in front of libdgnn_hip.so: the pybind11 modules the reference
// builds from dgNN/src/*/*.cpp|.cc (setup.py:29-34) -- fused_gatconv, fused_gmmconv,
// fused_edgeconv, mhsddmm, mhtranspose, format_conversion -- with the same function names,
// argument order and return lists, plus spmm, knn and gat_layer_ops (no reference module).
//
// This layer is plumbing only: it checks tensors, takes outputs and workspaces from torch's
// caching allocator, sets the device from the input tensor, passes torch's CURRENT HIP stream,
// and calls the C ABI of include/dgnn_hip.h.  No kernel lives here and no HIP call is made
// here.  There is no CPU path: a tensor that is not on the GPU is an error.
//
// One translation unit defines every PyInit_<module>; the build links it once per module name
// (dgnn_amd/native/<module>.cpython-*.so).  The C-ABI library is opened with dlopen so that
// DGNN_HIP_LIB can select another build of the same ABI (kernel A/B runs), exactly as
// dgnn_amd/_lib.py does for the ctypes users (tests, bench timing hook).
#include <dlfcn.h>

#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <torch/extension.h>

#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/dgnn_hip.h"

namespace py = pybind11;
using at::Tensor;
using OptTensor = c10::optional<Tensor>;

namespace {

struct DgnnError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

template <class... A>
[[noreturn]] void fail(const A &...a) {
  throw DgnnError(c10::str(a...));
}

// ------------------------------------------------------------------ the C-ABI library
std::string self_dir() {
  Dl_info info;
  if (dladdr((void *)&self_dir, &info) == 0 || info.dli_fname == nullptr) return ".";
  std::string p(info.dli_fname);
  const size_t k = p.rfind('/');
  return k == std::string::npos ? "." : p.substr(0, k);
}

void *dgnn_handle() {
  static void *handle = nullptr;
  static std::once_flag once;
  static std::string err;
  std::call_once(once, [] {
    const char *env = getenv("DGNN_HIP_LIB");
    const std::string path = (env != nullptr && env[0] != 0) ? std::string(env) : self_dir() + "/../csrc/libdgnn_hip.so";
    handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (handle == nullptr) {
      const char *why = dlerror();
      err = path + " could not be loaded (" + (why ? why : "?") +
            "): build it first (python -c 'import __graft_entry__ as g; g.build()' or make -C "
            "dgnn_amd/csrc).  dgnn_amd has no CPU/eager fallback.";
    }
  });
  if (handle == nullptr) fail(err);
  return handle;
}

void *dgnn_symbol(const char *name) {
  void *s = dlsym(dgnn_handle(), name);
  if (s == nullptr) fail(name, " is declared in include/dgnn_hip.h but missing from libdgnn_hip.so");
  return s;
}
// API(fn)(args...) calls the C-ABI entry point `fn` with the prototype the header declares.
#define API(fn)                                                                  \
  ([]() -> decltype(&fn) {                                                       \
    static auto p = reinterpret_cast<decltype(&fn)>(dgnn_symbol(#fn));           \
    return p;                                                                    \
  }())

void check_rc(int rc, const char *what) {
  if (rc == 0) return;
  const char *msg = API(dgnn_error_string)(rc);
  fail(what, " failed with code ", rc, ": ", msg ? msg : "?");
}

// ------------------------------------------------------------------ tensor checks / plumbing
// The reference bindings assert CUDA + dtype + contiguity (fused_gatconv.cpp:15-29); asserts
// vanish under NDEBUG, these do not.
const Tensor &on_gpu(const Tensor &t, const char *name, at::ScalarType dtype, const char *dtype_name, int ndim) {
  if (!t.defined() || !t.is_cuda()) fail(name, " must be a GPU tensor (dgnn_amd has no CPU path)");
  if (t.scalar_type() != dtype) fail(name, " must be ", dtype_name, ", got ", t.scalar_type());
  if (!t.is_contiguous()) fail(name, " must be contiguous");
  if (ndim >= 0 && t.dim() != ndim) fail(name, " must have ", ndim, " dims, got shape ", t.sizes());
  return t;
}
const Tensor &f32(const Tensor &t, const char *name, int ndim = -1) {
  return on_gpu(t, name, at::kFloat, "float32", ndim);
}
const Tensor &i32(const Tensor &t, const char *name, int ndim = 1) {
  return on_gpu(t, name, at::kInt, "int32", ndim);
}
bool given(const OptTensor &t) { return t.has_value() && t->defined(); }

template <class T>
T *ptr(const Tensor &t) {
  return (t.defined() && t.numel() > 0) ? static_cast<T *>(t.data_ptr()) : nullptr;
}
template <class T>
T *ptr(const OptTensor &t) {
  return given(t) ? ptr<T>(*t) : nullptr;
}
const float *fp(const Tensor &t) { return ptr<const float>(t); }
const float *fp(const OptTensor &t) { return ptr<const float>(t); }
const int *ip(const Tensor &t) { return ptr<const int>(t); }
float *fo(const Tensor &t) { return ptr<float>(t); }
int *io(const Tensor &t) { return ptr<int>(t); }

void *stream_of(const Tensor &t) {
  return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}
Tensor new_f32(at::IntArrayRef sizes, const Tensor &like) { return at::empty(sizes, like.options().dtype(at::kFloat)); }
Tensor new_i32(at::IntArrayRef sizes, const Tensor &like) { return at::empty(sizes, like.options().dtype(at::kInt)); }
// Scratch from torch's caching allocator: stream-ordered reuse on the current stream.
Tensor workspace(size_t nbytes, const Tensor &like) {
  return nbytes == 0 ? Tensor() : at::empty({(int64_t)nbytes}, like.options().dtype(at::kByte));
}
// Caller-provided outputs (keyword `out`: a sequence, entries may be None): entry i when it is a
// tensor -- float32, contiguous, on the inputs' device, exactly `sizes` -- else a fresh tensor.
using OutList = c10::optional<std::vector<OptTensor>>;
Tensor out_f32(const OutList &out, size_t i, at::IntArrayRef sizes, const Tensor &like, const char *name) {
  if (!out.has_value() || i >= out->size() || !given((*out)[i])) return new_f32(sizes, like);
  const Tensor &t = *(*out)[i];
  f32(t, name);
  if (t.sizes() != sizes || t.device() != like.device())
    fail("out: ", name, " must be a float32 tensor of shape ", sizes, " on ", like.device(), ", got ", t.sizes());
  return t;
}
int as_int(int64_t v, const char *what) {
  if (v < 0 || v > INT32_MAX) fail(what, " = ", v, " does not fit the int32 sizes of the C ABI");
  return (int)v;
}

// ================================================================== fused_gatconv
struct GatDims {
  int m, nnz, h, f;
};
// shapes are inferred exactly as gat_forward_cuda does (fused_gatconv_kernel.cu:367-370)
GatDims check_gat_fwd(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr, const Tensor &col_ind,
                      const Tensor &in_feat) {
  f32(attn_row, "attn_row", 2), f32(attn_col, "attn_col", 2), f32(in_feat, "in_feat", 3);
  i32(row_ptr, "row_ptr"), i32(col_ind, "col_ind");
  if (in_feat.size(1) != attn_row.size(1) || attn_col.size(1) != attn_row.size(1))
    fail("attn_row [m,h], attn_col [n,h] and in_feat [n,h,f] disagree on h");
  if (attn_col.size(0) != in_feat.size(0)) fail("attn_col and in_feat disagree on the number of source nodes");
  if (attn_row.size(0) != row_ptr.size(0) - 1) fail("attn_row rows != len(row_ptr) - 1");
  return {as_int(row_ptr.size(0) - 1, "rows"), as_int(col_ind.size(0), "nnz"), as_int(attn_row.size(1), "heads"),
          as_int(in_feat.size(2), "features")};
}

// -> [out_feat, edge_max, edge_sum, edge_mask]   (fused_gatconv.cpp:11-32)
// DGNN_REF_SAVED=0: the reference-signature calls keep a materialised mask (round-2 behaviour)
bool ref_saved_enabled() {
  const char *e = std::getenv("DGNN_REF_SAVED");
  return !(e != nullptr && e[0] == '0');
}

std::vector<Tensor> gat_forward(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr,
                                const Tensor &col_ind, double negative_slope, const Tensor &in_feat,
                                double attn_drop, const OptTensor &edge_mask_in, const OutList &outs, bool neg,
                                bool saved) {
  const GatDims d = check_gat_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat);
  c10::DeviceGuard guard(in_feat.device());
  Tensor out = out_f32(outs, 0, {d.m, d.h, d.f}, in_feat, "out_feat");
  Tensor edge_max = out_f32(outs, 1, {d.m, d.h}, in_feat, "edge_max");
  Tensor edge_sum = out_f32(outs, 2, {d.m, d.h}, in_feat, "edge_sum");
  // The reference's own wrapper (dgNN/operators/fused_gatconv.py:16-29) keeps the fourth result and hands
  // it back to gat_backward untouched; nobody else reads it.  A plain call with the reference's seven
  // arguments therefore gets in that slot, instead of an [nnz,h] mask, the 1-D tensor dgnn_gat_forward_saved
  // fills (a copy of out_feat, the negative-score pair and the dropout seed): gat_backward recognises it
  // by its shape and runs the one-pass backward -- no rebuilt forward, no mask.  saved=False, an explicit
  // edge_mask, or DGNN_REF_SAVED=0 give the mask of round 2.
  if (saved && !neg && !given(edge_mask_in) && !outs.has_value() && d.m > 0 && d.f >= 4 && d.h <= 8 &&
      ref_saved_enabled()) {
    uint64_t seed = 0;
    if (attn_drop > 0.0)      // torch's CPU generator: reproducible under torch.manual_seed, no device sync
      seed = (uint64_t)at::randint(0, (int64_t)1 << 62, {1}, at::TensorOptions().dtype(at::kLong)).item<int64_t>();
    Tensor carrier = new_f32({(int64_t)API(dgnn_gat_saved_floats)(d.m, d.h, d.f)}, in_feat);
    const size_t nb = API(dgnn_gat_forward_workspace_bytes)(d.m, d.nnz, d.h, d.f);
    Tensor ws = workspace(nb, in_feat);
    const int rc = API(dgnn_gat_forward_saved)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope,
                                               fp(attn_row), fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), seed,
                                               (uint64_t)0, fo(edge_max), fo(edge_sum), fo(out), fo(carrier),
                                               ptr<void>(ws), nb, stream_of(in_feat));
    if (rc != DGNN_ERR_UNSUPPORTED) {
      check_rc(rc, "dgnn_gat_forward_saved");
      return {out, edge_max, edge_sum, carrier};
    }
  }
  Tensor edge_mask;
  if (attn_drop > 0.0) {
    // the reference draws the mask with cuRAND on every call (fused_gatconv_kernel.cu:333-341)
    edge_mask = given(edge_mask_in) ? *edge_mask_in : at::rand({d.nnz, d.h}, in_feat.options());
    f32(edge_mask, "edge_mask", 2);
    if (edge_mask.size(0) != d.nnz || edge_mask.size(1) != d.h) fail("edge_mask must be [nnz, h]");
  } else {
    edge_mask = new_f32({0, d.h}, in_feat);   // every edge is kept: no mask is materialised
  }
  const size_t nb = API(dgnn_gat_forward_workspace_bytes)(d.m, d.nnz, d.h, d.f);
  Tensor ws = workspace(nb, in_feat);
  // neg: also the negative-score pair [out_neg, a_neg] the backward turns into grad_attn_row
  // (appended to the list; two undefined tensors = None where the kernels have no such pair)
  Tensor out_neg, a_neg;
  if (neg && d.m > 0 && d.f >= 4) {
    out_neg = out_f32(outs, 3, {d.m, d.h, d.f}, in_feat, "out_neg");
    a_neg = out_f32(outs, 4, {d.m, d.h}, in_feat, "a_neg");
  }
  int rc = API(dgnn_gat_forward)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope, fp(attn_row),
                                 fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), fp(edge_mask), fo(edge_max),
                                 fo(edge_sum), fo(out), fo(out_neg), fo(a_neg), ptr<void>(ws), nb, stream_of(in_feat));
  if (rc == DGNN_ERR_UNSUPPORTED && out_neg.defined()) {
    out_neg = Tensor();
    a_neg = Tensor();
    rc = API(dgnn_gat_forward)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope, fp(attn_row),
                               fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), fp(edge_mask), fo(edge_max),
                               fo(edge_sum), fo(out), nullptr, nullptr, ptr<void>(ws), nb, stream_of(in_feat));
  }
  check_rc(rc, "dgnn_gat_forward");
  if (neg) return {out, edge_max, edge_sum, edge_mask, out_neg, a_neg};
  return {out, edge_max, edge_sum, edge_mask};
}

// -> [out_feat, edge_max, edge_sum]   (fused_gatconv.cpp:68-95)
std::vector<Tensor> gat_forward_tb(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr,
                                   const Tensor &col_ind, double negative_slope, const Tensor &in_feat,
                                   const Tensor &tile_scheduler, int64_t tile) {
  const GatDims d = check_gat_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat);
  i32(tile_scheduler, "tile_scheduler", 2);
  c10::DeviceGuard guard(in_feat.device());
  Tensor out = new_f32({d.m, d.h, d.f}, in_feat);
  Tensor edge_max = new_f32({d.m, d.h}, in_feat), edge_sum = new_f32({d.m, d.h}, in_feat);
  if (tile_scheduler.size(1) != 2) fail("tile_scheduler must be [n_tiles, 2] (row, tile_in_row)");
  const size_t nb = API(dgnn_gat_forward_tiles_workspace_bytes)(d.m, d.nnz, d.h, d.f);
  Tensor ws = workspace(nb, in_feat);
  check_rc(API(dgnn_gat_forward_tiles)(d.m, d.nnz, d.h, d.f, (float)negative_slope, fp(attn_row), fp(attn_col),
                                       ip(row_ptr), ip(col_ind), fp(in_feat), ip(tile_scheduler),
                                       as_int(tile_scheduler.size(0), "tiles"), as_int(tile, "tile"), fo(edge_max), fo(edge_sum), fo(out),
                                       ptr<void>(ws), nb, stream_of(in_feat)),
           "dgnn_gat_forward_tiles");
  return {out, edge_max, edge_sum};
}

// The (row, tile_in_row) list gat_forward_tb takes -- one pair per `tile` nonzeros of a row --
// built with index arithmetic on row_ptr's device instead of the reference's host loop over
// all rows (script/test/test_gat_tb.py:16-26).  -> int32 [n_tiles, 2].
Tensor make_tile_scheduler(const Tensor &row_ptr, int64_t tile) {
  if (tile <= 0) fail("tile must be positive");
  const Tensor rp = row_ptr.to(at::kLong);
  const int64_t n = rp.size(0) - 1;
  const Tensor tiles = at::floor_divide(rp.slice(0, 1, n + 1) - rp.slice(0, 0, n) + (tile - 1), tile);
  const Tensor rid = at::repeat_interleave(at::arange(n, rp.options()), tiles);
  const Tensor first = at::cumsum(tiles, 0) - tiles;
  const Tensor tin = at::arange(rid.size(0), rp.options()) - first.index_select(0, rid);
  return at::stack({rid, tin}, 1).to(at::kInt).contiguous();
}

// -> out_feat   (fused_gatconv.cpp:38-60)
Tensor gat_inference(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr, const Tensor &col_ind,
                     double negative_slope, const Tensor &in_feat) {
  const GatDims d = check_gat_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat);
  c10::DeviceGuard guard(in_feat.device());
  Tensor out = new_f32({d.m, d.h, d.f}, in_feat);
  const size_t nb = API(dgnn_gat_forward_workspace_bytes)(d.m, d.nnz, d.h, d.f);
  Tensor ws = workspace(nb, in_feat);
  check_rc(API(dgnn_gat_inference)(d.m, d.nnz, d.h, d.f, (float)negative_slope, fp(attn_row), fp(attn_col),
                                   ip(row_ptr), ip(col_ind), fp(in_feat), fo(out), ptr<void>(ws), nb,
                                   stream_of(in_feat)),
           "dgnn_gat_inference");
  return out;
}

// -> S [m,h] = <grad[r,h,:], out_feat[r,h,:]> (the row term of the backward)
// -> (row_stat [m,h,4], grad_attn_row [m,h] or None), or None where the kernels have no such pass
py::object gat_row_stats(double negative_slope, const Tensor &attn_row, const Tensor &edge_max, const Tensor &edge_sum,
                         const Tensor &grad, const Tensor &out_feat, const c10::optional<Tensor> &out_neg,
                         const c10::optional<Tensor> &a_neg) {
  f32(grad, "grad", 3), f32(out_feat, "out_feat", 3), f32(attn_row, "attn_row", 2), f32(edge_max, "edge_max", 2),
      f32(edge_sum, "edge_sum", 2);
  if (grad.sizes() != out_feat.sizes()) fail("gat_row_stats: grad and out_feat must have the same shape");
  const int m = as_int(grad.size(0), "rows"), h = as_int(grad.size(1), "heads"), f = as_int(grad.size(2), "features");
  for (const Tensor *t : {&attn_row, &edge_max, &edge_sum})
    if (t->size(0) != m || t->size(1) != h) fail("gat_row_stats: attn_row / edge_max / edge_sum must be [m,h]");
  if (given(out_neg) != given(a_neg)) fail("gat_row_stats: out_neg and a_neg come together");
  if (given(out_neg)) {
    f32(*out_neg, "out_neg", 3), f32(*a_neg, "a_neg", 2);
    if (out_neg->sizes() != grad.sizes() || a_neg->size(0) != m || a_neg->size(1) != h)
      fail("gat_row_stats: out_neg must be [m,h,f] and a_neg [m,h]");
  }
  c10::DeviceGuard guard(grad.device());
  Tensor stat = new_f32({m, h, 4}, grad);
  Tensor gr = given(out_neg) ? new_f32({m, h}, grad) : Tensor();
  const int rc = API(dgnn_gat_row_stats)(m, h, f, (float)negative_slope, fp(attn_row), fp(edge_max), fp(edge_sum),
                                         fp(grad), fp(out_feat), fp(out_neg), fp(a_neg), fo(stat),
                                         given(out_neg) ? fo(gr) : nullptr, stream_of(grad));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_row_stats");
  return py::make_tuple(stat, given(out_neg) ? py::cast(gr) : py::none());
}

Tensor gat_row_dot(const Tensor &grad, const Tensor &out_feat) {
  f32(grad, "grad", 3), f32(out_feat, "out_feat", 3);
  if (grad.sizes() != out_feat.sizes()) fail("gat_row_dot: grad and out_feat must have the same shape");
  c10::DeviceGuard guard(grad.device());
  Tensor S = new_f32({grad.size(0), grad.size(1)}, grad);
  check_rc(API(dgnn_gat_row_dot)(as_int(grad.size(0), "rows"), as_int(grad.size(1), "heads"),
                                 as_int(grad.size(2), "features"), fp(grad), fp(out_feat), fo(S), stream_of(grad)),
           "dgnn_gat_row_dot");
  return S;
}

// the optional (out_neg [m,h,f], a_neg [m,h]) pair of the backward entry points: both or neither
void check_neg_pair(const OptTensor &out_neg, const OptTensor &a_neg, const Tensor &grad) {
  if (given(out_neg) != given(a_neg)) fail("out_neg and a_neg come as a pair");
  if (!given(out_neg)) return;
  f32(*out_neg, "out_neg", 3), f32(*a_neg, "a_neg", 2);
  if (out_neg->sizes() != grad.sizes() || a_neg->size(0) != grad.size(0) || a_neg->size(1) != grad.size(1))
    fail("out_neg must be [m,h,f] like grad, a_neg [m,h]");
}

// -> [grad_feat, grad_attn_row, grad_attn_col]   (fused_gatconv.cpp:104-166).
// out_feat / row_dot (keywords, not in the reference signature): the saved forward output, or
// the precomputed row term; either removes one whole gather pass (include/dgnn_hip.h).  Rows
// (len(row_ptr)-1) and columns (len(col_ptr)-1) may differ (row / column partitions).
std::vector<Tensor> gat_backward(double negative_slope, double attn_drop, const Tensor &row_ptr, const Tensor &col_ind,
                                 const Tensor &col_ptr, const Tensor &row_ind, const Tensor &permute,
                                 const Tensor &edge_max, const Tensor &edge_sum, const OptTensor &edge_mask,
                                 const Tensor &in_feat, const Tensor &attn_row, const Tensor &attn_col,
                                 const Tensor &grad, const OptTensor &out_feat, const OptTensor &row_dot,
                                 const OutList &outs, const OptTensor &out_neg, const OptTensor &a_neg) {
  i32(row_ptr, "row_ptr"), i32(col_ind, "col_ind"), i32(col_ptr, "col_ptr"), i32(row_ind, "row_ind");
  i32(permute, "permute");
  f32(edge_max, "edge_max", 2), f32(edge_sum, "edge_sum", 2), f32(in_feat, "in_feat", 3);
  f32(attn_row, "attn_row", 2), f32(attn_col, "attn_col", 2), f32(grad, "grad", 3);
  const int m = as_int(row_ptr.size(0) - 1, "rows"), nnz = as_int(col_ind.size(0), "nnz");
  const int n = as_int(col_ptr.size(0) - 1, "columns");
  const int h = as_int(in_feat.size(1), "heads"), f = as_int(in_feat.size(2), "features");   // :1049-1050
  check_neg_pair(out_neg, a_neg, grad);
  // the tensor gat_forward returned in the edge_mask slot of the reference signature (see there): 1-D,
  // exactly dgnn_gat_saved_floats(m, h, f) long -- a mask is [nnz, h]
  if (given(edge_mask) && edge_mask->dim() == 1 && m > 0 &&
      edge_mask->numel() == (int64_t)API(dgnn_gat_saved_floats)(m, h, f) && !given(out_neg)) {
    f32(*edge_mask, "edge_mask (saved forward state)", 1);
    if (in_feat.size(0) != n || attn_col.size(0) != n) fail("gat_backward: in_feat/attn_col rows must equal len(col_ptr)-1");
    if (grad.size(0) != m || grad.size(1) != h || grad.size(2) != f) fail("gat_backward: grad must be [m,h,f]");
    c10::DeviceGuard guard(in_feat.device());
    Tensor grad_feat = out_f32(outs, 0, {n, h, f}, in_feat, "grad_feat");
    Tensor grad_attn_row = out_f32(outs, 1, {m, h}, in_feat, "grad_attn_row");
    Tensor grad_attn_col = out_f32(outs, 2, {n, h}, in_feat, "grad_attn_col");
    const size_t nb = API(dgnn_gat_backward_workspace_bytes)(m, n, nnz, h, f, 1);
    Tensor ws = workspace(nb, in_feat);
    check_rc(API(dgnn_gat_backward_saved)(m, n, nnz, h, f, (float)negative_slope, (float)attn_drop, ip(row_ptr),
                                          ip(col_ind), ip(col_ptr), ip(row_ind), ip(permute), fp(edge_max), fp(edge_sum),
                                          fp(*edge_mask), fp(in_feat), fp(attn_row), fp(attn_col), fp(grad), fo(grad_feat),
                                          fo(grad_attn_row), fo(grad_attn_col), ptr<void>(ws), nb, stream_of(in_feat)),
             "dgnn_gat_backward_saved");
    return {grad_feat, grad_attn_row, grad_attn_col};
  }
  if (in_feat.size(0) != n || attn_col.size(0) != n)
    fail("gat_backward: in_feat/attn_col rows must equal len(col_ptr)-1");
  if (grad.size(0) != m || attn_row.size(0) != m || edge_max.size(0) != m || edge_sum.size(0) != m)
    fail("gat_backward: grad/attn_row/edge_max/edge_sum rows must equal len(row_ptr)-1");
  if (grad.size(1) != h || grad.size(2) != f) fail("gat_backward: grad must be [m,h,f]");
  if (row_ind.size(0) != nnz || permute.size(0) != nnz) fail("gat_backward: row_ind / permute must hold nnz entries");
  if (attn_drop > 0.0) {
    if (!given(edge_mask)) fail("gat_backward: edge_mask is required when attn_drop > 0");
    f32(*edge_mask, "edge_mask", 2);
  }
  if (given(out_feat)) f32(*out_feat, "out_feat", 3);
  if (given(row_dot)) f32(*row_dot, "row_dot", 2);
  const int have = (given(out_feat) || given(row_dot)) ? 1 : 0;
  c10::DeviceGuard guard(in_feat.device());
  Tensor grad_feat = out_f32(outs, 0, {n, h, f}, in_feat, "grad_feat");
  Tensor grad_attn_row = out_f32(outs, 1, {m, h}, in_feat, "grad_attn_row");
  Tensor grad_attn_col = out_f32(outs, 2, {n, h}, in_feat, "grad_attn_col");
  const size_t nb = API(dgnn_gat_backward_workspace_bytes)(m, n, nnz, h, f, have);
  Tensor ws = workspace(nb, in_feat);
  check_rc(API(dgnn_gat_backward)(m, n, nnz, h, f, (float)negative_slope, (float)attn_drop, ip(row_ptr), ip(col_ind),
                                  ip(col_ptr), ip(row_ind), ip(permute), fp(edge_max), fp(edge_sum),
                                  attn_drop > 0.0 ? fp(edge_mask) : nullptr, fp(in_feat), fp(attn_row), fp(attn_col),
                                  fp(grad), fp(out_feat), fp(row_dot), fp(out_neg), fp(a_neg), fo(grad_feat),
                                  fo(grad_attn_row), fo(grad_attn_col), ptr<void>(ws), nb, stream_of(in_feat)),
           "dgnn_gat_backward");
  return {grad_feat, grad_attn_row, grad_attn_col};
}

// ---- dropout without a stored mask (include/dgnn_hip.h: dgnn_*_seeded) ----------------------
// -> the [nnz,h] mask the seeded entry points regenerate from (seed, offset)
Tensor dropout_mask(int64_t nnz, int64_t h, uint64_t seed, uint64_t offset, const Tensor &like) {
  if (!like.defined() || !like.is_cuda()) fail("like must be a GPU tensor (dgnn_amd has no CPU path)");
  c10::DeviceGuard guard(like.device());
  Tensor mask = new_f32({nnz, h}, like);
  check_rc(API(dgnn_dropout_mask)(as_int(nnz, "nnz"), as_int(h, "heads"), seed, offset, fo(mask), stream_of(like)),
           "dgnn_dropout_mask");
  return mask;
}

// gat_forward with the dropout mask regenerated from (seed, offset) instead of drawn and stored.
// -> [out_feat, edge_max, edge_sum], or None where the seeded kernels do not apply (the caller
// materialises the same mask with dropout_mask and takes gat_forward).
py::object gat_forward_seeded(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr,
                              const Tensor &col_ind, double negative_slope, const Tensor &in_feat, double attn_drop,
                              uint64_t seed, uint64_t offset, const OutList &outs, bool neg) {
  const GatDims d = check_gat_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat);
  c10::DeviceGuard guard(in_feat.device());
  Tensor out = out_f32(outs, 0, {d.m, d.h, d.f}, in_feat, "out_feat");
  Tensor edge_max = out_f32(outs, 1, {d.m, d.h}, in_feat, "edge_max");
  Tensor edge_sum = out_f32(outs, 2, {d.m, d.h}, in_feat, "edge_sum");
  const size_t nb = API(dgnn_gat_forward_workspace_bytes)(d.m, d.nnz, d.h, d.f);
  Tensor ws = workspace(nb, in_feat);
  Tensor out_neg, a_neg;
  if (neg && d.m > 0 && d.f >= 4) {
    out_neg = new_f32({d.m, d.h, d.f}, in_feat);
    a_neg = new_f32({d.m, d.h}, in_feat);
  }
  int rc = API(dgnn_gat_forward_seeded)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope,
                                        fp(attn_row), fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), seed,
                                        offset, fo(edge_max), fo(edge_sum), fo(out), fo(out_neg), fo(a_neg),
                                        ptr<void>(ws), nb, stream_of(in_feat));
  if (rc == DGNN_ERR_UNSUPPORTED && out_neg.defined()) {     // no negative-score pair here: plain seeded forward
    out_neg = Tensor();
    a_neg = Tensor();
    rc = API(dgnn_gat_forward_seeded)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope, fp(attn_row),
                                      fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), seed, offset, fo(edge_max),
                                      fo(edge_sum), fo(out), nullptr, nullptr, ptr<void>(ws), nb, stream_of(in_feat));
  }
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_forward_seeded");
  if (neg) return py::cast(std::vector<Tensor>{out, edge_max, edge_sum, out_neg, a_neg});
  return py::cast(std::vector<Tensor>{out, edge_max, edge_sum});
}

// gat_backward with (seed, offset) in place of edge_mask.  -> [grad_feat, grad_attn_row,
// grad_attn_col] or None (see gat_forward_seeded).
py::object gat_backward_seeded(double negative_slope, double attn_drop, const Tensor &row_ptr, const Tensor &col_ind,
                               const Tensor &col_ptr, const Tensor &row_ind, const Tensor &permute,
                               const Tensor &edge_max, const Tensor &edge_sum, uint64_t seed, uint64_t offset,
                               const Tensor &in_feat, const Tensor &attn_row, const Tensor &attn_col,
                               const Tensor &grad, const OptTensor &out_feat, const OptTensor &row_dot,
                               const OutList &outs, const OptTensor &out_neg, const OptTensor &a_neg) {
  i32(row_ptr, "row_ptr"), i32(col_ind, "col_ind"), i32(col_ptr, "col_ptr"), i32(row_ind, "row_ind");
  i32(permute, "permute");
  f32(edge_max, "edge_max", 2), f32(edge_sum, "edge_sum", 2), f32(in_feat, "in_feat", 3);
  f32(attn_row, "attn_row", 2), f32(attn_col, "attn_col", 2), f32(grad, "grad", 3);
  check_neg_pair(out_neg, a_neg, grad);
  const int m = as_int(row_ptr.size(0) - 1, "rows"), nnz = as_int(col_ind.size(0), "nnz");
  const int n = as_int(col_ptr.size(0) - 1, "columns");
  const int h = as_int(in_feat.size(1), "heads"), f = as_int(in_feat.size(2), "features");
  if (in_feat.size(0) != n || attn_col.size(0) != n)
    fail("gat_backward_seeded: in_feat/attn_col rows must equal len(col_ptr)-1");
  if (grad.size(0) != m || attn_row.size(0) != m || edge_max.size(0) != m || edge_sum.size(0) != m)
    fail("gat_backward_seeded: grad/attn_row/edge_max/edge_sum rows must equal len(row_ptr)-1");
  if (grad.size(1) != h || grad.size(2) != f) fail("gat_backward_seeded: grad must be [m,h,f]");
  if (row_ind.size(0) != nnz || permute.size(0) != nnz) fail("gat_backward_seeded: row_ind / permute must hold nnz entries");
  if (given(out_feat)) f32(*out_feat, "out_feat", 3);
  if (given(row_dot)) f32(*row_dot, "row_dot", 2);
  const int have = (given(out_feat) || given(row_dot)) ? 1 : 0;
  c10::DeviceGuard guard(in_feat.device());
  Tensor grad_feat = out_f32(outs, 0, {n, h, f}, in_feat, "grad_feat");
  Tensor grad_attn_row = out_f32(outs, 1, {m, h}, in_feat, "grad_attn_row");
  Tensor grad_attn_col = out_f32(outs, 2, {n, h}, in_feat, "grad_attn_col");
  const size_t nb = API(dgnn_gat_backward_workspace_bytes)(m, n, nnz, h, f, have);
  Tensor ws = workspace(nb, in_feat);
  const int rc = API(dgnn_gat_backward_seeded)(
      m, n, nnz, h, f, (float)negative_slope, (float)attn_drop, ip(row_ptr), ip(col_ind), ip(col_ptr), ip(row_ind),
      ip(permute), fp(edge_max), fp(edge_sum), seed, offset, fp(in_feat), fp(attn_row), fp(attn_col), fp(grad),
      fp(out_feat), fp(row_dot), fp(out_neg), fp(a_neg), fo(grad_feat), fo(grad_attn_row), fo(grad_attn_col),
      ptr<void>(ws), nb, stream_of(in_feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_backward_seeded");
  return py::cast(std::vector<Tensor>{grad_feat, grad_attn_row, grad_attn_col});
}

// ---- the seeded pair with (seed, offset) in DEVICE memory (include/dgnn_hip.h: dgnn_*_seeded_dev):
// rng_state = int64 tensor [2] on the inputs' device.  What a HIP-graph capture needs: scalar
// arguments are frozen into a captured launch, a device pair is read anew at every replay.
const uint64_t *rng_state_ptr(const Tensor &rng_state, const Tensor &like) {
  if (!rng_state.defined() || !rng_state.is_cuda() || rng_state.scalar_type() != at::kLong || rng_state.numel() != 2 ||
      !rng_state.is_contiguous() || rng_state.device() != like.device())
    fail("rng_state must be a contiguous int64 tensor [2] = {seed, offset} on the inputs' device");
  return reinterpret_cast<const uint64_t *>(rng_state.data_ptr());
}

py::object gat_forward_seeded_dev(const Tensor &attn_row, const Tensor &attn_col, const Tensor &row_ptr,
                                  const Tensor &col_ind, double negative_slope, const Tensor &in_feat, double attn_drop,
                                  const Tensor &rng_state, const OutList &outs, bool neg) {
  const GatDims d = check_gat_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat);
  const uint64_t *state = rng_state_ptr(rng_state, in_feat);
  c10::DeviceGuard guard(in_feat.device());
  Tensor out = out_f32(outs, 0, {d.m, d.h, d.f}, in_feat, "out_feat");
  Tensor edge_max = out_f32(outs, 1, {d.m, d.h}, in_feat, "edge_max");
  Tensor edge_sum = out_f32(outs, 2, {d.m, d.h}, in_feat, "edge_sum");
  const size_t nb = API(dgnn_gat_forward_workspace_bytes)(d.m, d.nnz, d.h, d.f);
  Tensor ws = workspace(nb, in_feat);
  Tensor out_neg, a_neg;
  if (neg && d.m > 0 && d.f >= 4) {
    out_neg = new_f32({d.m, d.h, d.f}, in_feat);
    a_neg = new_f32({d.m, d.h}, in_feat);
  }
  int rc = API(dgnn_gat_forward_seeded_dev)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope,
                                            fp(attn_row), fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), state,
                                            fo(edge_max), fo(edge_sum), fo(out), fo(out_neg), fo(a_neg), ptr<void>(ws),
                                            nb, stream_of(in_feat));
  if (rc == DGNN_ERR_UNSUPPORTED && out_neg.defined()) {
    out_neg = Tensor();
    a_neg = Tensor();
    rc = API(dgnn_gat_forward_seeded_dev)(d.m, d.nnz, d.h, d.f, (float)attn_drop, (float)negative_slope, fp(attn_row),
                                          fp(attn_col), ip(row_ptr), ip(col_ind), fp(in_feat), state, fo(edge_max),
                                          fo(edge_sum), fo(out), nullptr, nullptr, ptr<void>(ws), nb,
                                          stream_of(in_feat));
  }
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_forward_seeded_dev");
  if (neg) return py::cast(std::vector<Tensor>{out, edge_max, edge_sum, out_neg, a_neg});
  return py::cast(std::vector<Tensor>{out, edge_max, edge_sum});
}

py::object gat_backward_seeded_dev(double negative_slope, double attn_drop, const Tensor &row_ptr,
                                   const Tensor &col_ind, const Tensor &col_ptr, const Tensor &row_ind,
                                   const Tensor &permute, const Tensor &edge_max, const Tensor &edge_sum,
                                   const Tensor &rng_state, const Tensor &in_feat, const Tensor &attn_row,
                                   const Tensor &attn_col, const Tensor &grad, const OptTensor &out_feat,
                                   const OptTensor &row_dot, const OutList &outs, const OptTensor &out_neg,
                                   const OptTensor &a_neg) {
  check_neg_pair(out_neg, a_neg, grad);
  i32(row_ptr, "row_ptr"), i32(col_ind, "col_ind"), i32(col_ptr, "col_ptr"), i32(row_ind, "row_ind");
  i32(permute, "permute");
  f32(edge_max, "edge_max", 2), f32(edge_sum, "edge_sum", 2), f32(in_feat, "in_feat", 3);
  f32(attn_row, "attn_row", 2), f32(attn_col, "attn_col", 2), f32(grad, "grad", 3);
  const uint64_t *state = rng_state_ptr(rng_state, in_feat);
  const int m = as_int(row_ptr.size(0) - 1, "rows"), nnz = as_int(col_ind.size(0), "nnz");
  const int n = as_int(col_ptr.size(0) - 1, "columns");
  const int h = as_int(in_feat.size(1), "heads"), f = as_int(in_feat.size(2), "features");
  if (in_feat.size(0) != n || attn_col.size(0) != n)
    fail("gat_backward_seeded_dev: in_feat/attn_col rows must equal len(col_ptr)-1");
  if (grad.size(0) != m || attn_row.size(0) != m || edge_max.size(0) != m || edge_sum.size(0) != m)
    fail("gat_backward_seeded_dev: grad/attn_row/edge_max/edge_sum rows must equal len(row_ptr)-1");
  if (grad.size(1) != h || grad.size(2) != f) fail("gat_backward_seeded_dev: grad must be [m,h,f]");
  if (row_ind.size(0) != nnz || permute.size(0) != nnz) fail("gat_backward_seeded_dev: row_ind / permute must hold nnz entries");
  if (given(out_feat)) f32(*out_feat, "out_feat", 3);
  if (given(row_dot)) f32(*row_dot, "row_dot", 2);
  const int have = (given(out_feat) || given(row_dot)) ? 1 : 0;
  c10::DeviceGuard guard(in_feat.device());
  Tensor grad_feat = out_f32(outs, 0, {n, h, f}, in_feat, "grad_feat");
  Tensor grad_attn_row = out_f32(outs, 1, {m, h}, in_feat, "grad_attn_row");
  Tensor grad_attn_col = out_f32(outs, 2, {n, h}, in_feat, "grad_attn_col");
  const size_t nb = API(dgnn_gat_backward_workspace_bytes)(m, n, nnz, h, f, have);
  Tensor ws = workspace(nb, in_feat);
  const int rc = API(dgnn_gat_backward_seeded_dev)(
      m, n, nnz, h, f, (float)negative_slope, (float)attn_drop, ip(row_ptr), ip(col_ind), ip(col_ptr), ip(row_ind),
      ip(permute), fp(edge_max), fp(edge_sum), state, fp(in_feat), fp(attn_row), fp(attn_col), fp(grad),
      fp(out_feat), fp(row_dot), fp(out_neg), fp(a_neg), fo(grad_feat), fo(grad_attn_row), fo(grad_attn_col),
      ptr<void>(ws), nb, stream_of(in_feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_backward_seeded_dev");
  return py::cast(std::vector<Tensor>{grad_feat, grad_attn_row, grad_attn_col});
}

// gat_backward with the per-row inputs packed: row_stat [m,h,4] = {attn_row, edge_max, edge_sum,
// row_dot} (include/dgnn_hip.h, dgnn_gat_backward_packed).  -> [grad_feat, grad_attn_row,
// grad_attn_col], or None where the packed entry point does not apply (caller unpacks).
py::object gat_backward_packed(double negative_slope, double attn_drop, const Tensor &row_ptr, const Tensor &col_ind,
                               const Tensor &col_ptr, const Tensor &row_ind, const Tensor &permute,
                               const Tensor &row_stat, const OptTensor &edge_mask, const Tensor &in_feat,
                               const Tensor &attn_col, const Tensor &grad, const OutList &outs,
                               bool need_grad_attn_row) {
  i32(row_ptr, "row_ptr"), i32(col_ind, "col_ind"), i32(col_ptr, "col_ptr"), i32(row_ind, "row_ind");
  i32(permute, "permute");
  f32(row_stat, "row_stat", 3), f32(in_feat, "in_feat", 3), f32(attn_col, "attn_col", 2), f32(grad, "grad", 3);
  const int m = as_int(row_ptr.size(0) - 1, "rows"), nnz = as_int(col_ind.size(0), "nnz");
  const int n = as_int(col_ptr.size(0) - 1, "columns");
  const int h = as_int(in_feat.size(1), "heads"), f = as_int(in_feat.size(2), "features");
  if (in_feat.size(0) != n || attn_col.size(0) != n || attn_col.size(1) != h)
    fail("gat_backward_packed: in_feat/attn_col rows must equal len(col_ptr)-1");
  if (row_stat.size(0) != m || row_stat.size(1) != h || row_stat.size(2) != 4)
    fail("gat_backward_packed: row_stat must be [len(row_ptr)-1, h, 4]");
  if (grad.size(0) != m || grad.size(1) != h || grad.size(2) != f) fail("gat_backward_packed: grad must be [m,h,f]");
  if (row_ind.size(0) != nnz || permute.size(0) != nnz) fail("gat_backward_packed: row_ind / permute must hold nnz entries");
  if (attn_drop > 0.0) {
    if (!given(edge_mask)) fail("gat_backward_packed: edge_mask is required when attn_drop > 0");
    f32(*edge_mask, "edge_mask", 2);
  }
  c10::DeviceGuard guard(in_feat.device());
  Tensor grad_feat = out_f32(outs, 0, {n, h, f}, in_feat, "grad_feat");
  Tensor grad_attn_row = need_grad_attn_row ? out_f32(outs, 1, {m, h}, in_feat, "grad_attn_row") : Tensor();
  Tensor grad_attn_col = out_f32(outs, 2, {n, h}, in_feat, "grad_attn_col");
  const size_t nb = API(dgnn_gat_backward_workspace_bytes)(m, n, nnz, h, f, 1);
  Tensor ws = workspace(nb, in_feat);
  const int rc = API(dgnn_gat_backward_packed)(
      m, n, nnz, h, f, (float)negative_slope, (float)attn_drop, ip(row_ptr), ip(col_ind), ip(col_ptr), ip(row_ind),
      ip(permute), fp(row_stat), attn_drop > 0.0 ? fp(edge_mask) : nullptr, fp(in_feat), fp(attn_col), fp(grad),
      fo(grad_feat), fo(grad_attn_row), fo(grad_attn_col), ptr<void>(ws), nb, stream_of(in_feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_backward_packed");
  return py::cast(std::vector<Tensor>{grad_feat, grad_attn_row, grad_attn_col});
}

// ================================================================== spmm / mhsddmm / mhtranspose
// -> out[r,h,:] = sum_j edge_val[j,h] * in_feat[colind[j],h,:]   (GE-SpMM)
Tensor spmm_csr(const Tensor &rowptr, const Tensor &colind, const Tensor &edge_val, const Tensor &in_feat) {
  i32(rowptr, "rowptr"), i32(colind, "colind"), f32(edge_val, "edge_val", 2), f32(in_feat, "in_feat", 3);
  const int m = as_int(rowptr.size(0) - 1, "rows"), nnz = as_int(colind.size(0), "nnz");
  const int h = as_int(in_feat.size(1), "heads"), f = as_int(in_feat.size(2), "features");
  if (edge_val.size(0) != nnz || edge_val.size(1) != h) fail("spmm_csr: edge_val must be [nnz, H]");
  c10::DeviceGuard guard(in_feat.device());
  Tensor out = new_f32({m, h, f}, in_feat);
  const size_t nb = API(dgnn_spmm_csr_workspace_bytes)(m, nnz, h, f);
  Tensor ws = workspace(nb, in_feat);
  check_rc(API(dgnn_spmm_csr)(m, nnz, h, f, ip(rowptr), ip(colind), fp(edge_val), fp(in_feat), fo(out),
                              ptr<void>(ws), nb, stream_of(in_feat)),
           "dgnn_spmm_csr");
  return out;
}

// -> out [E,H] = <grad[row(e),h,:], feature[col(e),h,:]>   (mhsddmm.cc:11-28)
Tensor mhsddmm(const Tensor &rowptr, const Tensor &colind, const Tensor &grad, const Tensor &feature) {
  i32(rowptr, "rowptr"), i32(colind, "colind"), f32(grad, "grad", 3), f32(feature, "feature", 3);
  const int v = as_int(rowptr.size(0) - 1, "rows"), nnz = as_int(colind.size(0), "nnz");
  const int h = as_int(feature.size(1), "heads"), f = as_int(feature.size(2), "features");
  if (grad.size(1) != h || grad.size(2) != f) fail("mhsddmm: grad and feature must both be [*, H, F]");
  c10::DeviceGuard guard(grad.device());
  Tensor out = new_f32({nnz, h}, grad);
  check_rc(API(dgnn_mhsddmm)(v, f, h, nnz, ip(rowptr), ip(colind), fp(grad), fp(feature), fo(out), stream_of(grad)),
           "dgnn_mhsddmm");
  return out;
}

// -> out[i,:] = attention[permute[i],:]   (mhtranspose.cc:10-18)
Tensor mhtranspose(const Tensor &permute, const Tensor &attention) {
  i32(permute, "permute"), f32(attention, "attention", 2);
  const int nnz = as_int(permute.size(0), "nnz"), h = as_int(attention.size(1), "heads");
  c10::DeviceGuard guard(attention.device());
  Tensor out = new_f32({nnz, h}, attention);
  check_rc(API(dgnn_mhtranspose)(nnz, h, ip(permute), fp(attention), fo(out), stream_of(attention)),
           "dgnn_mhtranspose");
  return out;
}

// ================================================================== fused_gmmconv
// -> out [nodes, K, embed]   (fused_gmmconv.cpp:37-47)
Tensor gmmconv(const Tensor &csrptr, const Tensor &colind, const Tensor &node_feat, const Tensor &pseudo,
               const Tensor &mu, const Tensor &inv_sigma) {
  i32(csrptr, "csrptr"), i32(colind, "colind"), f32(node_feat, "node_feat", 3), f32(pseudo, "pseudo", 2);
  f32(mu, "mu", 2), f32(inv_sigma, "inv_sigma", 2);
  const int nodes = as_int(csrptr.size(0) - 1, "nodes"), nnz = as_int(colind.size(0), "nnz");
  const int k = as_int(mu.size(0), "kernels"), dim = as_int(mu.size(1), "dim");   // fused_gmmconv_kernel.cu:107-111
  const int embed = as_int(node_feat.size(2), "features");
  if (node_feat.size(1) != k || pseudo.size(0) != nnz || pseudo.size(1) != dim)
    fail("gmmconv: node_feat [n,K,F], pseudo [nnz,dim], mu [K,dim] disagree");
  c10::DeviceGuard guard(node_feat.device());
  Tensor out = new_f32({nodes, k, embed}, node_feat);
  const size_t nb = API(dgnn_gmm_forward_workspace_bytes)(nodes, nnz, k, embed);
  Tensor ws = workspace(nb, node_feat);
  check_rc(API(dgnn_gmm_forward)(nodes, nnz, k, dim, embed, ip(csrptr), ip(colind), fp(node_feat), fp(pseudo), fp(mu),
                                 fp(inv_sigma), fo(out), ptr<void>(ws), nb, stream_of(node_feat)),
           "dgnn_gmm_forward");
  return out;
}

// -> gauss [E,K]   (fused_gmmconv.cpp:29-35)
Tensor Gaussian(const Tensor &pseudo, const Tensor &mu, const Tensor &inv_sigma) {
  f32(pseudo, "pseudo", 2), f32(mu, "mu", 2), f32(inv_sigma, "inv_sigma", 2);
  const int e = as_int(pseudo.size(0), "edges"), k = as_int(mu.size(0), "kernels"), dim = as_int(mu.size(1), "dim");
  c10::DeviceGuard guard(pseudo.device());
  Tensor out = new_f32({e, k}, pseudo);
  check_rc(API(dgnn_gaussian)(e, k, dim, fp(pseudo), fp(mu), fp(inv_sigma), fo(out), stream_of(pseudo)),
           "dgnn_gaussian");
  return out;
}

// -> [pseudo_out, mu_out, sigma_out]   (fused_gmmconv.cpp:49-57, order of
// fused_gmmconv_kernel.cu:143).  Analytic gradient; see include/dgnn_hip.h.
std::vector<Tensor> GaussianBp(const Tensor &pseudo, const Tensor &mu, const Tensor &inv_sigma, const Tensor &grad_out) {
  f32(pseudo, "pseudo", 2), f32(mu, "mu", 2), f32(inv_sigma, "inv_sigma", 2), f32(grad_out, "grad_out", 2);
  const int e = as_int(pseudo.size(0), "edges"), k = as_int(mu.size(0), "kernels"), dim = as_int(mu.size(1), "dim");
  c10::DeviceGuard guard(pseudo.device());
  Tensor po = new_f32({e, dim}, pseudo), mo = new_f32({k, dim}, pseudo), so = new_f32({k, dim}, pseudo);
  const size_t nb = API(dgnn_gaussian_backward_workspace_bytes)(e, k, dim);
  Tensor ws = workspace(nb, pseudo);
  check_rc(API(dgnn_gaussian_backward)(e, k, dim, fp(pseudo), fp(mu), fp(inv_sigma), fp(grad_out), fo(po), fo(mo),
                                       fo(so), ptr<void>(ws), nb, stream_of(pseudo)),
           "dgnn_gaussian_backward");
  return {po, mo, so};
}

// -> (grad_node_feat [nodes,K,embed], grad_gauss [E,K] in CSR order), or None when the shape is
// outside the one-pass kernel's range.  Not a reference export: it stands for the mhsddmm +
// mhtranspose + gmmconv(CSC) sequence of dgNN/operators/fused_gmmconv.py:21-24.
py::object gmm_backward_fused(const Tensor &colptr, const Tensor &rowind, const Tensor &permute,
                              const Tensor &node_feat, const Tensor &pseudo, const Tensor &mu,
                              const Tensor &inv_sigma, const Tensor &grad_out) {
  i32(colptr, "colptr"), i32(rowind, "rowind"), i32(permute, "permute");
  f32(node_feat, "node_feat", 3), f32(pseudo, "pseudo", 2), f32(grad_out, "grad_out", 3);
  f32(mu, "mu", 2), f32(inv_sigma, "inv_sigma", 2);
  const int nodes = as_int(colptr.size(0) - 1, "nodes"), nnz = as_int(rowind.size(0), "nnz");
  const int kernels = as_int(node_feat.size(1), "kernels"), embed = as_int(node_feat.size(2), "features");
  const int dim = as_int(pseudo.size(1), "dim");
  if (grad_out.sizes() != node_feat.sizes() || pseudo.size(0) != nnz || node_feat.size(0) != nodes)
    fail("gmm_backward_fused: grad_out / node_feat / pseudo shapes disagree with the graph");
  const size_t nb = API(dgnn_gmm_backward_workspace_bytes)(nodes, nnz, kernels, embed);
  if (nb == 0) return py::none();
  c10::DeviceGuard guard(node_feat.device());
  Tensor grad_feat = at::empty_like(node_feat);
  Tensor grad_gauss = new_f32({nnz, kernels}, node_feat);
  Tensor ws = workspace(nb, node_feat);
  const int rc = API(dgnn_gmm_backward)(nodes, nnz, kernels, dim, embed, ip(colptr), ip(rowind), ip(permute),
                                        fp(node_feat), fp(pseudo), fp(mu), fp(inv_sigma), fp(grad_out), fo(grad_feat),
                                        fo(grad_gauss), ptr<void>(ws), nb, stream_of(node_feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gmm_backward");
  return py::cast(std::make_tuple(grad_feat, grad_gauss));
}

// ---- GMMConv with the layer's sum over the kernels folded in (include/dgnn_hip.h: dgnn_gmm_*_sum)
// -> out_sum [nodes, embed] = gmmconv(...).sum(1), or None where the summed kernels do not apply
py::object gmmconv_sum(const Tensor &csrptr, const Tensor &colind, const Tensor &node_feat, const Tensor &pseudo,
                       const Tensor &mu, const Tensor &inv_sigma) {
  i32(csrptr, "csrptr"), i32(colind, "colind"), f32(node_feat, "node_feat", 3), f32(pseudo, "pseudo", 2);
  f32(mu, "mu", 2), f32(inv_sigma, "inv_sigma", 2);
  const int nodes = as_int(csrptr.size(0) - 1, "nodes"), nnz = as_int(colind.size(0), "nnz");
  const int k = as_int(mu.size(0), "kernels"), dim = as_int(mu.size(1), "dim");
  const int embed = as_int(node_feat.size(2), "features");
  if (node_feat.size(1) != k || pseudo.size(0) != nnz || pseudo.size(1) != dim)
    fail("gmmconv_sum: node_feat [n,K,F], pseudo [nnz,dim], mu [K,dim] disagree");
  c10::DeviceGuard guard(node_feat.device());
  Tensor out = new_f32({nodes, embed}, node_feat);
  const size_t nb = API(dgnn_gmm_forward_workspace_bytes)(nodes, nnz, k, embed);
  Tensor ws = workspace(nb, node_feat);
  const int rc = API(dgnn_gmm_forward_sum)(nodes, nnz, k, dim, embed, ip(csrptr), ip(colind), fp(node_feat), fp(pseudo),
                                           fp(mu), fp(inv_sigma), fo(out), ptr<void>(ws), nb, stream_of(node_feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gmm_forward_sum");
  return py::cast(out);
}

// -> (grad_node_feat [nodes,K,embed], grad_gauss [E,K]) from grad_sum [nodes, embed], or None
py::object gmm_backward_fused_sum(const Tensor &colptr, const Tensor &rowind, const Tensor &permute,
                                  const Tensor &node_feat, const Tensor &pseudo, const Tensor &mu,
                                  const Tensor &inv_sigma, const Tensor &grad_sum) {
  i32(colptr, "colptr"), i32(rowind, "rowind"), i32(permute, "permute");
  f32(node_feat, "node_feat", 3), f32(pseudo, "pseudo", 2), f32(grad_sum, "grad_sum", 2);
  f32(mu, "mu", 2), f32(inv_sigma, "inv_sigma", 2);
  const int nodes = as_int(colptr.size(0) - 1, "nodes"), nnz = as_int(rowind.size(0), "nnz");
  const int kernels = as_int(node_feat.size(1), "kernels"), embed = as_int(node_feat.size(2), "features");
  const int dim = as_int(pseudo.size(1), "dim");
  if (grad_sum.size(0) != nodes || grad_sum.size(1) != embed || pseudo.size(0) != nnz || node_feat.size(0) != nodes)
    fail("gmm_backward_fused_sum: grad_sum [n,F] / node_feat [n,K,F] / pseudo shapes disagree with the graph");
  const size_t nb = API(dgnn_gmm_backward_workspace_bytes)(nodes, nnz, kernels, embed);
  if (nb == 0) return py::none();
  c10::DeviceGuard guard(node_feat.device());
  Tensor grad_feat = at::empty_like(node_feat);
  Tensor grad_gauss = new_f32({nnz, kernels}, node_feat);
  Tensor ws = workspace(nb, node_feat);
  const int rc = API(dgnn_gmm_backward_sum)(nodes, nnz, kernels, dim, embed, ip(colptr), ip(rowind), ip(permute),
                                            fp(node_feat), fp(pseudo), fp(mu), fp(inv_sigma), fp(grad_sum),
                                            fo(grad_feat), fo(grad_gauss), ptr<void>(ws), nb, stream_of(node_feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gmm_backward_sum");
  return py::cast(std::make_tuple(grad_feat, grad_gauss));
}

// ================================================================== fused_edgeconv / knn
// -> [out_feat, max_idx]; max_idx is a float32 tensor holding ids, as in the reference
// (fused_edgeconv_kernel.cu:46,61).  return_int_idx additionally returns the exact int32 copy
// (ids >= 2^24 are not representable in float32).
std::vector<Tensor> edgeconv_forward(int64_t k, const Tensor &src_ind, const Tensor &h_src, const Tensor &h_dst,
                                     bool return_int_idx) {
  i32(src_ind, "src_ind"), f32(h_src, "h_src", 2), f32(h_dst, "h_dst", 2);
  const int m = as_int(h_src.size(0), "points"), f = as_int(h_src.size(1), "features");   // :55-56
  if (h_dst.size(0) != m || h_dst.size(1) != f || src_ind.numel() != (int64_t)m * k)
    fail("edgeconv_forward: h_src/h_dst must be [m,f] and src_ind [m*k]");
  c10::DeviceGuard guard(h_src.device());
  Tensor out = new_f32({m, f}, h_src), idx_f = new_f32({m, f}, h_src);
  Tensor idx_i = return_int_idx ? new_i32({m, f}, h_src) : Tensor();
  check_rc(API(dgnn_edgeconv_forward)(m, f, as_int(k, "k"), ip(src_ind), fp(h_src), fp(h_dst), fo(out), fo(idx_f),
                                      io(idx_i), stream_of(h_src)),
           "dgnn_edgeconv_forward");
  if (return_int_idx) return {out, idx_f, idx_i};
  return {out, idx_f};
}

// -> grad_src   (fused_edgeconv.cpp:32-45); max_idx int32.  src_ind (keyword, not in the reference
// signature): the k-NN lists the forward was given -- with them the gradient is gathered over the
// transposed lists in a fixed order, without float atomics (dgnn_edgeconv_backward_det).
Tensor edgeconv_backward(const Tensor &grad_out, const Tensor &max_idx, const OptTensor &src_ind) {
  f32(grad_out, "grad_out", 2), i32(max_idx, "max_idx", 2);
  if (max_idx.sizes() != grad_out.sizes()) fail("edgeconv_backward: max_idx and grad_out must both be [m,f]");
  const int m = as_int(grad_out.size(0), "points"), f = as_int(grad_out.size(1), "features");
  c10::DeviceGuard guard(grad_out.device());
  Tensor grad_src = new_f32({m, f}, grad_out);
  if (given(src_ind) && m > 0) {
    i32(*src_ind, "src_ind");
    if (src_ind->numel() % m != 0) fail("edgeconv_backward: src_ind must hold m*k ids");
    const int k = as_int(src_ind->numel() / m, "k");
    const size_t nb = API(dgnn_edgeconv_backward_det_workspace_bytes)(m, k);
    Tensor ws = workspace(nb, grad_out);
    check_rc(API(dgnn_edgeconv_backward_det)(m, f, k, ip(*src_ind), ip(max_idx), fp(grad_out), fo(grad_src),
                                             ptr<void>(ws), nb, stream_of(grad_out)),
             "dgnn_edgeconv_backward_det");
    return grad_src;
  }
  check_rc(API(dgnn_edgeconv_backward)(m, f, ip(max_idx), fp(grad_out), fo(grad_src), stream_of(grad_out)),
           "dgnn_edgeconv_backward");
  return grad_src;
}

// x [batch, points, dim] (or [points, dim] = one cloud) -> int32 src_ind [batch*points*k]: what
// the reference takes from DGL's KNNGraph (script/train/train_edgeconv.py:17,49-52).
Tensor knn_graph(const Tensor &x_in, int64_t k) {
  const Tensor x = (x_in.defined() && x_in.dim() == 2) ? x_in.unsqueeze(0) : x_in;
  f32(x, "x", 3);
  const int batch = as_int(x.size(0), "batch"), points = as_int(x.size(1), "points"), dim = as_int(x.size(2), "dim");
  c10::DeviceGuard guard(x.device());
  Tensor src_ind = new_i32({(int64_t)batch * points * k}, x);
  const size_t nb = API(dgnn_knn_graph_workspace_bytes)(batch, points, dim, as_int(k, "k"));
  Tensor ws = workspace(nb, x);
  check_rc(API(dgnn_knn_graph_ws)(batch, points, dim, as_int(k, "k"), fp(x), io(src_ind), nb ? ws.data_ptr() : nullptr,
                                  nb, stream_of(x)),
           "dgnn_knn_graph_ws");
  return src_ind;
}

// ================================================================== format_conversion
// -> (colptr, rowind, permute), all int32, bit-exact stable transpose on the device
std::tuple<Tensor, Tensor, Tensor> csr2csc_int(const Tensor &rowptr, const Tensor &colind,
                                               const c10::optional<int64_t> &n_cols) {
  i32(rowptr, "rowptr"), i32(colind, "colind");
  const int m = as_int(rowptr.size(0) - 1, "rows"), nnz = as_int(colind.size(0), "nnz");
  const int n = n_cols.has_value() ? as_int(*n_cols, "n_cols") : m;   // reference assumes square, convert.cpp:38
  c10::DeviceGuard guard(rowptr.device());
  Tensor colptr = new_i32({n + 1}, rowptr), rowind = new_i32({nnz}, rowptr), permute = new_i32({nnz}, rowptr);
  const size_t nb = API(dgnn_csr2csc_workspace_bytes)(m, n, nnz);
  Tensor ws = workspace(nb, rowptr);
  check_rc(API(dgnn_csr2csc)(m, n, nnz, ip(rowptr), ip(colind), io(colptr), io(rowind), io(permute), ptr<void>(ws), nb,
                             stream_of(rowptr)),
           "dgnn_csr2csc");
  return std::make_tuple(colptr, rowind, permute);
}

// -> [colptr, rowind, cscval]   (convert.cpp:16-25): csr_data (float32 [nnz]) is carried to CSC
// order, exactly what cusparseCsr2cscEx2 does with the values.
std::vector<Tensor> csr2csc(const Tensor &rowptr, const Tensor &colind, const Tensor &csr_data) {
  f32(csr_data, "csr_data", 1);
  auto t = csr2csc_int(rowptr, colind, c10::nullopt);
  return {std::get<0>(t), std::get<1>(t), csr_data.index_select(0, std::get<2>(t).to(at::kLong))};
}

// -> rowptr [m+1]   (convert.cpp:27-35); cooRowInd sorted ascending
Tensor coo2csr(const Tensor &cooRowInd, int64_t m) {
  i32(cooRowInd, "cooRowInd");
  c10::DeviceGuard guard(cooRowInd.device());
  Tensor rowptr = new_i32({m + 1}, cooRowInd);
  check_rc(API(dgnn_coo2csr)(as_int(m, "m"), as_int(cooRowInd.size(0), "nnz"), ip(cooRowInd), io(rowptr),
                             stream_of(cooRowInd)),
           "dgnn_coo2csr");
  return rowptr;
}

// ================================================================== gat_layer_ops
// feat [n,h,f], attn_l/attn_r [h,f] (any leading 1) -> (attn_row, attn_col) [n,h], or None when
// the shape is outside the kernel's range (the caller keeps the torch expression).
py::object attn_scores(const Tensor &feat, const Tensor &attn_l, const Tensor &attn_r) {
  f32(feat, "feat", 3), f32(attn_l, "attn_l"), f32(attn_r, "attn_r");
  const int n = as_int(feat.size(0), "nodes"), h = as_int(feat.size(1), "heads"), f = as_int(feat.size(2), "features");
  if (attn_l.numel() != (int64_t)h * f || attn_r.numel() != (int64_t)h * f) fail("attn_l / attn_r must hold h*f values");
  c10::DeviceGuard guard(feat.device());
  Tensor attn_row = new_f32({n, h}, feat), attn_col = new_f32({n, h}, feat);
  const int rc = API(dgnn_gat_attn_scores)(n, h, f, fp(feat), fp(attn_l), fp(attn_r), fo(attn_row), fo(attn_col),
                                           stream_of(feat));
  if (rc == DGNN_ERR_UNSUPPORTED) return py::none();
  check_rc(rc, "dgnn_gat_attn_scores");
  return py::cast(std::make_tuple(attn_row, attn_col));
}

// Adds the score terms' gradient into grad_feat IN PLACE and returns (grad_attn_l, grad_attn_r)
// shaped like attn_l / attn_r.
std::tuple<Tensor, Tensor> attn_scores_backward(const Tensor &feat, const Tensor &attn_l, const Tensor &attn_r,
                                                const Tensor &g_row, const Tensor &g_col, Tensor grad_feat) {
  f32(feat, "feat", 3), f32(attn_l, "attn_l"), f32(attn_r, "attn_r");
  f32(g_row, "g_row", 2), f32(g_col, "g_col", 2), f32(grad_feat, "grad_feat", 3);
  const int n = as_int(feat.size(0), "nodes"), h = as_int(feat.size(1), "heads"), f = as_int(feat.size(2), "features");
  if (grad_feat.sizes() != feat.sizes() || g_row.size(0) != n || g_col.size(0) != n || g_row.size(1) != h ||
      g_col.size(1) != h)
    fail("attn_scores_backward: grad_feat must be [n,h,f], g_row / g_col [n,h]");
  c10::DeviceGuard guard(feat.device());
  Tensor gl = at::empty_like(attn_l), gr = at::empty_like(attn_r);
  const size_t nb = API(dgnn_gat_attn_scores_backward_workspace_bytes)(n, h, f);
  Tensor ws = workspace(nb, feat);
  check_rc(API(dgnn_gat_attn_scores_backward)(n, h, f, fp(feat), fp(attn_l), fp(attn_r), fp(g_row), fp(g_col),
                                              fo(grad_feat), fo(gl), fo(gr), ptr<void>(ws), nb, stream_of(feat)),
           "dgnn_gat_attn_scores_backward");
  return std::make_tuple(gl, gr);
}

Tensor epilogue(const Tensor &rst, const OptTensor &residual, const OptTensor &bias, int64_t activation,
                double act_param) {
  f32(rst, "rst");
  const int64_t n = rst.dim() ? rst.size(0) : 0;
  const int64_t hf = n ? rst.numel() / n : std::max<int64_t>(rst.numel(), 1);
  if (given(residual)) {
    f32(*residual, "residual");
    if (residual->numel() != rst.numel()) fail("residual must have rst's size");
  }
  if (given(bias)) {
    f32(*bias, "bias");
    if (bias->numel() != hf) fail("bias must hold one value per (head, feature)");
  }
  c10::DeviceGuard guard(rst.device());
  Tensor out = at::empty_like(rst);
  check_rc(API(dgnn_gat_epilogue)(as_int(n, "rows"), as_int(hf, "row width"), (int)activation, (float)act_param, fp(rst),
                                  fp(residual), fp(bias), fo(out), stream_of(rst)),
           "dgnn_gat_epilogue");
  return out;
}

// -> (grad_pre, grad_bias or None)
std::tuple<Tensor, py::object> epilogue_backward(const Tensor &out, const Tensor &grad, int64_t activation,
                                                 double act_param, bool want_bias) {
  f32(out, "out"), f32(grad, "grad");
  if (grad.numel() != out.numel()) fail("epilogue_backward: grad must have out's size");
  const int64_t n = out.dim() ? out.size(0) : 0;
  const int64_t hf = n ? out.numel() / n : 1;
  c10::DeviceGuard guard(out.device());
  Tensor grad_pre = at::empty_like(grad);
  Tensor grad_bias = want_bias ? new_f32({hf}, out) : Tensor();
  const size_t nb = want_bias ? API(dgnn_gat_epilogue_backward_workspace_bytes)(as_int(n, "rows"), as_int(hf, "row width")) : 0;
  Tensor ws = workspace(nb, out);
  check_rc(API(dgnn_gat_epilogue_backward)(as_int(n, "rows"), as_int(hf, "row width"), (int)activation, (float)act_param,
                                           fp(out), fp(grad), fo(grad_pre), fo(grad_bias), ptr<void>(ws), nb,
                                           stream_of(out)),
           "dgnn_gat_epilogue_backward");
  return std::make_tuple(grad_pre, want_bias ? py::cast(grad_bias) : py::object(py::none()));
}

// ------------------------------------------------------------------ module plumbing
// DgnnError surfaces as dgnn_amd._lib.DgnnError (a RuntimeError), the class the ctypes users raise.
void translate_errors() {
  py::register_local_exception_translator([](std::exception_ptr p) {
    try {
      if (p) std::rethrow_exception(p);
    } catch (const DgnnError &e) {
      py::object cls;
      try {
        cls = py::module_::import("dgnn_amd._lib").attr("DgnnError");
      } catch (py::error_already_set &) {
        PyErr_Clear();
      }
      PyErr_SetString(cls ? cls.ptr() : PyExc_RuntimeError, e.what());
    }
  });
}

// def() with a text signature inspect.signature() understands ("name(a, b, c=None)\n--\n\n").
template <class F, class... Extra>
void def(py::module_ &m, const char *name, F &&f, const char *params, const char *doc, const Extra &...extra) {
  const std::string text = std::string(name) + "(" + params + ")\n--\n\n" + doc;
  m.def(name, std::forward<F>(f), text.c_str(), extra...);
}

void init(py::module_ &m, const char *what) {
  m.doc() = what;
  translate_errors();
}

}  // namespace

using py::arg;

// -> src[idx]: the rows a rank sends in the halo exchange, packed into one message (dist.py).
// `out` (keyword): write into the first len(idx) rows of a caller-held buffer instead.
Tensor pack_rows(const Tensor &src, const Tensor &idx, const OptTensor &out_in) {
  f32(src, "src"), i32(idx, "idx");
  if (src.dim() < 1) fail("src must have at least one dim");
  if (src.device() != idx.device()) fail("src and idx must be on the same device");
  const int64_t width = src.size(0) > 0 ? src.numel() / src.size(0) : 0;
  std::vector<int64_t> sizes = src.sizes().vec();
  sizes[0] = idx.size(0);
  c10::DeviceGuard guard(src.device());
  Tensor out = given(out_in) ? *out_in : new_f32(sizes, src);
  if (given(out_in)) {
    f32(out, "out");
    if (out.sizes() != at::IntArrayRef(sizes) || out.device() != src.device())
      fail("out must be a float32 tensor of shape ", at::IntArrayRef(sizes), " on ", src.device());
  }
  if (idx.size(0) == 0 || width == 0) return out;
  check_rc(API(dgnn_pack_rows)(as_int(idx.size(0), "rows out"), as_int(src.size(0), "rows in"), as_int(width, "width"),
                               ip(idx), fp(src), fo(out), stream_of(src)),
           "dgnn_pack_rows");
  return out;
}

PYBIND11_MODULE(fused_gatconv, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "Drop-in for the reference's pybind module fused_gatconv (fused_gatconv.cpp:168-174).");
  def(m, "gat_forward", &gat_forward,
      "attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop, edge_mask=None, out=None, neg=False, saved=True",
      "-> [out_feat, edge_max, edge_sum, edge_mask]   (fused_gatconv.cpp:11-32).  Called with the reference's seven\n"
      "arguments the fourth result is the forward state gat_backward wants back in its edge_mask argument (a 1-D\n"
      "tensor: copy of out_feat, negative-score pair, dropout seed -- no [nnz,h] mask is drawn or kept).  saved=False or\n"
      "an explicit edge_mask: the mask itself, drawn with torch's generator only when attn_drop > 0.\n"
      "out = (out_feat, edge_max, edge_sum), entries may be None: write into these tensors instead of new ones.",
      arg("attn_row"), arg("attn_col"), arg("row_ptr"), arg("col_ind"), arg("negative_slope"), arg("in_feat"),
      arg("attn_drop"), arg("edge_mask") = py::none(), arg("out") = py::none(), arg("neg") = false,
      arg("saved") = true);
  def(m, "gat_forward_tb", &gat_forward_tb,
      "attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, tile_scheduler, tile=32",
      "-> [out_feat, edge_max, edge_sum]   (fused_gatconv.cpp:68-95).  tile_scheduler must be the list the\n"
      "reference's scheduler builds for tiles of `tile` edges (make_tile_scheduler); checked on the device.",
      arg("attn_row"), arg("attn_col"), arg("row_ptr"), arg("col_ind"), arg("negative_slope"), arg("in_feat"),
      arg("tile_scheduler"), arg("tile") = 32);
  def(m, "make_tile_scheduler", &make_tile_scheduler, "row_ptr, tile=32",
      "-> int32 [n_tiles, 2]: the (row, tile_in_row) list of script/test/test_gat_tb.py:16-26", arg("row_ptr"),
      arg("tile") = 32);
  def(m, "gat_inference", &gat_inference, "attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat",
      "-> out_feat   (fused_gatconv.cpp:38-60)", arg("attn_row"), arg("attn_col"), arg("row_ptr"), arg("col_ind"),
      arg("negative_slope"), arg("in_feat"));
  def(m, "pack_rows", &pack_rows, "src, idx, out=None", "-> src[idx] (int32 idx, rows of any shape)", arg("src"),
      arg("idx"), arg("out") = py::none());
  def(m, "gat_row_dot", &gat_row_dot, "grad, out_feat", "-> S [m,h] = <grad[r,h,:], out_feat[r,h,:]>", arg("grad"),
      arg("out_feat"));
  def(m, "gat_row_stats", &gat_row_stats, "negative_slope, attn_row, edge_max, edge_sum, grad, out_feat, out_neg=None, a_neg=None",
      "-> (row_stat [m,h,4], grad_attn_row or None) in one pass over the rows; None where unsupported", arg("negative_slope"),
      arg("attn_row"), arg("edge_max"), arg("edge_sum"), arg("grad"), arg("out_feat"), arg("out_neg") = py::none(),
      arg("a_neg") = py::none());
  def(m, "gat_backward", &gat_backward,
      "negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, "
      "in_feat, attn_row, attn_col, grad, out_feat=None, row_dot=None, out=None, out_neg=None, a_neg=None",
      "-> [grad_feat, grad_attn_row, grad_attn_col]   (fused_gatconv.cpp:104-166).  out_feat / row_dot (keywords,\n"
      "not in the reference signature) remove one whole gather pass from the backward.  out = (grad_feat,\n"
      "grad_attn_row, grad_attn_col), entries may be None: write into these tensors instead of new ones.",
      arg("negative_slope"), arg("attn_drop"), arg("row_ptr"), arg("col_ind"), arg("col_ptr"), arg("row_ind"),
      arg("permute"), arg("edge_max"), arg("edge_sum"), arg("edge_mask"), arg("in_feat"), arg("attn_row"),
      arg("attn_col"), arg("grad"), arg("out_feat") = py::none(), arg("row_dot") = py::none(),
      arg("out") = py::none(), arg("out_neg") = py::none(), arg("a_neg") = py::none());
  def(m, "dropout_mask", &dropout_mask, "nnz, h, seed, offset, like",
      "-> the [nnz,h] mask (on like's device) the *_seeded functions regenerate from (seed, offset)", arg("nnz"),
      arg("h"), arg("seed"), arg("offset"), arg("like"));
  def(m, "gat_forward_seeded", &gat_forward_seeded,
      "attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop, seed, offset=0, out=None",
      "gat_forward with the dropout mask regenerated from (seed, offset): nothing is stored for the backward.\n"
      "-> [out_feat, edge_max, edge_sum], or None where the seeded kernels do not apply.",
      arg("attn_row"), arg("attn_col"), arg("row_ptr"), arg("col_ind"), arg("negative_slope"), arg("in_feat"),
      arg("attn_drop"), arg("seed"), arg("offset") = 0, arg("out") = py::none(), arg("neg") = false);
  def(m, "gat_forward_seeded_dev", &gat_forward_seeded_dev,
      "attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop, rng_state, out=None",
      "gat_forward_seeded with {seed, offset} read from the int64 device tensor rng_state [2] at launch\n"
      "(capturable into a HIP graph: every replay draws from what the tensor then holds).",
      arg("attn_row"), arg("attn_col"), arg("row_ptr"), arg("col_ind"), arg("negative_slope"), arg("in_feat"),
      arg("attn_drop"), arg("rng_state"), arg("out") = py::none(), arg("neg") = false);
  def(m, "gat_backward_seeded_dev", &gat_backward_seeded_dev,
      "negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, rng_state, "
      "in_feat, attn_row, attn_col, grad, out_feat=None, row_dot=None, out=None",
      "gat_backward_seeded with the device pair rng_state its forward used.",
      arg("negative_slope"), arg("attn_drop"), arg("row_ptr"), arg("col_ind"), arg("col_ptr"), arg("row_ind"),
      arg("permute"), arg("edge_max"), arg("edge_sum"), arg("rng_state"), arg("in_feat"), arg("attn_row"),
      arg("attn_col"), arg("grad"), arg("out_feat") = py::none(), arg("row_dot") = py::none(),
      arg("out") = py::none(), arg("out_neg") = py::none(), arg("a_neg") = py::none());
  def(m, "gat_backward_seeded", &gat_backward_seeded,
      "negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, seed, offset, "
      "in_feat, attn_row, attn_col, grad, out_feat=None, row_dot=None, out=None",
      "gat_backward with (seed, offset) in place of edge_mask.  -> [grad_feat, grad_attn_row, grad_attn_col] or None.",
      arg("negative_slope"), arg("attn_drop"), arg("row_ptr"), arg("col_ind"), arg("col_ptr"), arg("row_ind"),
      arg("permute"), arg("edge_max"), arg("edge_sum"), arg("seed"), arg("offset"), arg("in_feat"), arg("attn_row"),
      arg("attn_col"), arg("grad"), arg("out_feat") = py::none(), arg("row_dot") = py::none(),
      arg("out") = py::none(), arg("out_neg") = py::none(), arg("a_neg") = py::none());
  def(m, "gat_backward_packed", &gat_backward_packed,
      "negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, row_stat, edge_mask, in_feat, "
      "attn_col, grad, out=None",
      "gat_backward with row_stat [m,h,4] = {attn_row, edge_max, edge_sum, row_dot} packed per (row, head).\n"
      "-> [grad_feat, grad_attn_row, grad_attn_col], or None where the packed kernels do not apply.",
      arg("negative_slope"), arg("attn_drop"), arg("row_ptr"), arg("col_ind"), arg("col_ptr"), arg("row_ind"),
      arg("permute"), arg("row_stat"), arg("edge_mask"), arg("in_feat"), arg("attn_col"), arg("grad"),
      arg("out") = py::none(), arg("need_grad_attn_row") = true);
}

PYBIND11_MODULE(fused_gmmconv, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "Drop-in for the reference's pybind module fused_gmmconv (fused_gmmconv.cpp:59-64).");
  def(m, "gmmconv", &gmmconv, "csrptr, colind, node_feat, pseudo, mu, inv_sigma",
      "-> out [nodes, K, embed]   (fused_gmmconv.cpp:37-47)", arg("csrptr"), arg("colind"), arg("node_feat"),
      arg("pseudo"), arg("mu"), arg("inv_sigma"));
  def(m, "Gaussian", &Gaussian, "pseudo, mu, inv_sigma", "-> gauss [E, K]   (fused_gmmconv.cpp:29-35)", arg("pseudo"),
      arg("mu"), arg("inv_sigma"));
  def(m, "GaussianBp", &GaussianBp, "pseudo, mu, inv_sigma, grad_out",
      "-> [pseudo_out, mu_out, sigma_out]   (fused_gmmconv.cpp:49-57); analytic gradient", arg("pseudo"), arg("mu"),
      arg("inv_sigma"), arg("grad_out"));
  def(m, "gmmconv_sum", &gmmconv_sum, "csrptr, colind, node_feat, pseudo, mu, inv_sigma",
      "-> out_sum [nodes, embed] = gmmconv(...).sum(1) with the sum folded into the store (gmmconv_layer.py:58), or None",
      arg("csrptr"), arg("colind"), arg("node_feat"), arg("pseudo"), arg("mu"), arg("inv_sigma"));
  def(m, "gmm_backward_fused_sum", &gmm_backward_fused_sum,
      "colptr, rowind, permute, node_feat, pseudo, mu, inv_sigma, grad_sum",
      "gmm_backward_fused for grad_sum [nodes, embed], the gradient of the summed output; or None", arg("colptr"),
      arg("rowind"), arg("permute"), arg("node_feat"), arg("pseudo"), arg("mu"), arg("inv_sigma"), arg("grad_sum"));
  def(m, "gmm_backward_fused", &gmm_backward_fused,
      "colptr, rowind, permute, node_feat, pseudo, mu, inv_sigma, grad_out",
      "-> (grad_node_feat, grad_gauss [E,K] in CSR order) or None (shape outside the one-pass kernel)", arg("colptr"),
      arg("rowind"), arg("permute"), arg("node_feat"), arg("pseudo"), arg("mu"), arg("inv_sigma"), arg("grad_out"));
}

PYBIND11_MODULE(fused_edgeconv, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "Drop-in for the reference's pybind module fused_edgeconv (fused_edgeconv.cpp:47-51).");
  def(m, "edgeconv_forward", &edgeconv_forward, "k, src_ind, h_src, h_dst, return_int_idx=False",
      "-> [out_feat, max_idx(float32)] (+ the exact int32 ids with return_int_idx)   (fused_edgeconv.cpp:11-27)",
      arg("k"), arg("src_ind"), arg("h_src"), arg("h_dst"), arg("return_int_idx") = false);
  def(m, "edgeconv_backward", &edgeconv_backward, "grad_out, max_idx, src_ind=None",
      "-> grad_src   (fused_edgeconv.cpp:32-45).  With src_ind (the forward's k-NN lists): fixed summation\n"
      "order, no float atomics.",
      arg("grad_out"), arg("max_idx"), arg("src_ind") = py::none());
}

PYBIND11_MODULE(mhsddmm, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "Drop-in for the reference's pybind module mhsddmm (mhsddmm.cc:30-33).");
  def(m, "mhsddmm", &mhsddmm, "rowptr, colind, grad, feature",
      "-> out [E, H] = <grad[row(e),h,:], feature[col(e),h,:]>   (mhsddmm.cc:11-28)", arg("rowptr"), arg("colind"),
      arg("grad"), arg("feature"));
}

PYBIND11_MODULE(mhtranspose, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "Drop-in for the reference's pybind module mhtranspose (mhtranspose.cc:20-23).");
  def(m, "mhtranspose", &mhtranspose, "permute, attention",
      "-> out[i,:] = attention[permute[i],:]   (mhtranspose.cc:10-18)", arg("permute"), arg("attention"));
}

PYBIND11_MODULE(format_conversion, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "Drop-in for the reference's pybind module format_conversion (convert.cpp:37-41).");
  def(m, "csr2csc_int", &csr2csc_int, "rowptr, colind, n_cols=None",
      "-> (colptr, rowind, permute), all int32, bit-exact stable transpose on the device", arg("rowptr"),
      arg("colind"), arg("n_cols") = py::none());
  def(m, "csr2csc", &csr2csc, "rowptr, colind, csr_data", "-> [colptr, rowind, cscval]   (convert.cpp:16-25)",
      arg("rowptr"), arg("colind"), arg("csr_data"));
  def(m, "coo2csr", &coo2csr, "cooRowInd, m", "-> rowptr [m+1]   (convert.cpp:27-35)", arg("cooRowInd"), arg("m"));
}

PYBIND11_MODULE(spmm, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "GE-SpMM entry point (new API surface: the reference has no standalone SpMM module).");
  def(m, "spmm_csr", &spmm_csr, "rowptr, colind, edge_val, in_feat",
      "-> out[r,h,:] = sum_j edge_val[j,h] * in_feat[colind[j],h,:]", arg("rowptr"), arg("colind"), arg("edge_val"),
      arg("in_feat"));
}

PYBIND11_MODULE(knn, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "k-nearest-neighbour graph in front of the fused EdgeConv (replaces DGL's KNNGraph).");
  def(m, "knn_graph", &knn_graph, "x, k",
      "x [batch, points, dim] (or [points, dim]) -> int32 src_ind [batch*points*k], itself first, ascending\n"
      "distance, ties -> lower id", arg("x"), arg("k"));
}

PYBIND11_MODULE(gat_layer_ops, m) {
  py::options options;
  options.disable_function_signatures();
  init(m, "The GAT layer's single-pass steps around the fused operator (gatconv_layer.py:74-75, :82-90).");
  m.attr("ACT_NONE") = DGNN_ACT_NONE;
  m.attr("ACT_RELU") = DGNN_ACT_RELU;
  m.attr("ACT_ELU") = DGNN_ACT_ELU;
  m.attr("ACT_LEAKY_RELU") = DGNN_ACT_LEAKY_RELU;
  def(m, "attn_scores", &attn_scores, "feat, attn_l, attn_r",
      "-> (attn_row, attn_col) [n,h], or None when the shape is outside the kernel's range", arg("feat"),
      arg("attn_l"), arg("attn_r"));
  def(m, "attn_scores_backward", &attn_scores_backward, "feat, attn_l, attn_r, g_row, g_col, grad_feat",
      "adds the score terms' gradient into grad_feat IN PLACE; -> (grad_attn_l, grad_attn_r)", arg("feat"),
      arg("attn_l"), arg("attn_r"), arg("g_row"), arg("g_col"), arg("grad_feat"));
  def(m, "epilogue", &epilogue, "rst, residual, bias, activation, act_param", "-> act(rst + residual + bias)",
      arg("rst"), arg("residual"), arg("bias"), arg("activation"), arg("act_param"));
  def(m, "epilogue_backward", &epilogue_backward, "out, grad, activation, act_param, want_bias",
      "-> (grad_pre, grad_bias or None)", arg("out"), arg("grad"), arg("activation"), arg("act_param"),
      arg("want_bias"));
}

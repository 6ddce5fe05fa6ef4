// torch_bind.cpp -- PyTorch-ROCm dispatcher ops over the C ABI ("extern "C" / TORCH_LIBRARY ops (namespace
// mi355x_graph)"; north_star: "PyTorch-ROCm custom ops").
//
// The seam DGL v0.6 crosses at _CAPI_DGLKernelSpMM / _CAPI_DGLKernelSDDMM (UPSTREAM python/dgl/sparse.py::_gspmm/_gsddmm, reached
// from kernel/dgl-new.py:20,39) as torch ops over plain tensors:
//
//   mi355x_graph::gspmm(indptr, indices, eids?, num_cols, op, reduce, ufeat?, efeat?, plan=0, flags=0)  -> (out, arg_u, arg_e)
//       (flags: MGX_SPMM_SHORT_ROWS = 2, the caller's hint that the work items are short and even)
//   mi355x_graph::gsddmm(indptr, indices, eids?, num_cols, op, lhs?, rhs?, lhs_target, rhs_target, plan=0) -> out
//   mi355x_graph::gsddmm_coo(src, dst, num_src, num_dst, op, lhs?, rhs?, lhs_target, rhs_target)            -> out
//   mi355x_graph::edge_softmax_fwd / _bwd(indptr, indices, eids?, num_cols, ..., plan=0)
//   mi355x_graph::segment_reduce(offsets, x, reduce) / coo_to_csr / csr_transpose / in_degrees
//
// This translation unit is HOST code only: argument checks, output allocation with PyTorch's caching allocator, the current
// HIP stream, one call into libmi355x_graph.so (include/mi355x_graph.h stays free of torch types).  `plan`: address of a
// live mgx_spmm_plan (the execution schedule of that CSR, owned by the caller -- mi355x_graph.sparse.CsrView.plan()), 0 =
// natural row order.  Fake (meta) implementations and autograd formulas are registered from Python (mi355x_graph/torch_ops.py)
// on these same ops.  Built by csrc/Makefile into libmi355x_graph_torch.so, loaded with torch.ops.load_library.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string>
#include <tuple>

#include "../../include/mi355x_graph.h"

namespace {

using at::Tensor;
using c10::optional;

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

// Every op makes the device of its graph / feature tensors current for the duration of the call: allocations, the stream looked up by
// stream_of() and the kernel launches inside the C ABI must all refer to THAT device, not to whichever one the calling thread last used
// (several GPUs in one process; the ctypes route wraps its calls in `with torch.cuda.device(dev)` for the same reason).
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
#define MGX_DEVICE_GUARD(t, what)                                                                     \
  TORCH_CHECK((t).is_cuda(), what, ": runs on MI355X (HIP) tensors only -- there is no CPU path");    \
  const DeviceGuard device_guard_((t).device())

void check_status(int32_t st, const char* what) {
  TORCH_CHECK(st == MGX_OK, what, ": ", mgx_last_error(), " (mgx status ", st, ")");
}

int idx_bits(const Tensor& t, const char* what) {
  TORCH_CHECK(t.scalar_type() == at::kInt || t.scalar_type() == at::kLong, what, ": index tensors must be int32 or int64");
  return t.scalar_type() == at::kInt ? 32 : 64;
}

const void* ptr(const optional<Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }

mgx_csr make_csr(const Tensor& indptr, const Tensor& indices, const optional<Tensor>& eids, int64_t num_cols, const char* what) {
  TORCH_CHECK(indptr.is_cuda() && indices.is_cuda(), what, ": runs on MI355X (HIP) tensors only -- there is no CPU path");
  TORCH_CHECK(indptr.dim() == 1 && indptr.numel() >= 1 && indices.dim() == 1, what, ": indptr [rows + 1], indices [nnz]");
  TORCH_CHECK(indptr.is_contiguous() && indices.is_contiguous(), what, ": CSR arrays must be contiguous");
  TORCH_CHECK(indptr.scalar_type() == indices.scalar_type(), what, ": indptr and indices differ in index width");
  if (eids.has_value() && eids->defined())
    TORCH_CHECK(eids->scalar_type() == indices.scalar_type() && eids->is_contiguous() && eids->numel() == indices.numel() &&
                    eids->device() == indices.device(),
                what, ": eids must match indices");
  mgx_csr c;
  c.num_rows = indptr.numel() - 1;
  c.num_cols = num_cols;
  c.nnz = indices.numel();
  c.indptr = indptr.data_ptr();
  c.indices = indices.numel() ? indices.data_ptr() : nullptr;
  c.eids = ptr(eids);
  c.idx_bits = idx_bits(indptr, what);
  c.reserved = 0;
  return c;
}

int op_code(const std::string& op, const char* what) {
  if (op == "add") return MGX_OP_ADD;
  if (op == "sub") return MGX_OP_SUB;
  if (op == "mul") return MGX_OP_MUL;
  if (op == "div") return MGX_OP_DIV;
  if (op == "copy_lhs") return MGX_OP_COPY_LHS;
  if (op == "copy_rhs") return MGX_OP_COPY_RHS;
  if (op == "dot") return MGX_OP_DOT;
  TORCH_CHECK(false, what, ": unsupported binary op '", op, "'");
}

int reduce_code(const std::string& r, const char* what) {
  if (r == "sum") return MGX_REDUCE_SUM;
  if (r == "max") return MGX_REDUCE_MAX;
  if (r == "min") return MGX_REDUCE_MIN;
  if (r == "mean") return MGX_REDUCE_MEAN;
  TORCH_CHECK(false, what, ": unsupported reduce op '", r, "'");
}

int target_code(const std::string& t, const char* what) {
  if (t == "u") return MGX_TARGET_U;
  if (t == "e") return MGX_TARGET_E;
  if (t == "v") return MGX_TARGET_V;
  TORCH_CHECK(false, what, ": unsupported target '", t, "'");
}

Tensor as_f32(const Tensor& t, const Tensor& like, const char* what) {
  TORCH_CHECK(t.scalar_type() == at::kFloat, what, ": float32 features only");
  TORCH_CHECK(t.device() == like.device(), what, ": features and graph live on different devices");
  return t.contiguous();
}

int64_t row_len(const Tensor& t) { return t.dim() > 0 && t.size(0) > 0 ? t.numel() / t.size(0) : (t.dim() > 1 ? t.numel() : 1); }

int64_t trailing(const Tensor& t) {
  int64_t n = 1;
  for (int64_t d = 1; d < t.dim(); ++d) n *= t.size(d);
  return n;
}

// Output feature shape of op(U, E): equal shapes, or the head-wise broadcast the ABI takes without offset tables --
// (N, H, F) x (E, H, 1) / (N, D) x (E, 1).  Anything else needs offset tables: that is the Python layer's job (ops.gspmm).
std::vector<int64_t> feature_shape(const Tensor* U, const Tensor* E, int64_t& u_len, int64_t& e_len, int64_t& out_len, const char* what) {
  std::vector<int64_t> shape;
  const Tensor* ref = U ? U : E;
  if (U && E) {
    u_len = trailing(*U);
    e_len = trailing(*E);
    const Tensor* wide = u_len >= e_len ? U : E;
    const Tensor* thin = u_len >= e_len ? E : U;
    bool ok = wide->dim() == thin->dim();
    if (ok && u_len != e_len) {  // thin = wide with its last dimension set to 1 (one weight per head)
      for (int64_t d = 1; d + 1 < wide->dim(); ++d) ok = ok && wide->size(d) == thin->size(d);
      ok = ok && thin->size(thin->dim() - 1) == 1 && u_len > e_len;  // only E may be the per-head operand
    } else if (ok) {
      for (int64_t d = 1; d < wide->dim(); ++d) ok = ok && wide->size(d) == thin->size(d);
    }
    TORCH_CHECK(ok, what, ": operand shapes ", U->sizes(), " and ", E->sizes(),
                " need general broadcasting; call mi355x_graph.ops (it builds the offset tables)");
    ref = wide;
  } else {
    u_len = U ? trailing(*U) : 0;
    e_len = E ? trailing(*E) : 0;
  }
  for (int64_t d = 1; d < ref->dim(); ++d) shape.push_back(ref->size(d));
  out_len = trailing(*ref);
  return shape;
}

const mgx_spmm_plan* plan_of(int64_t handle) { return reinterpret_cast<const mgx_spmm_plan*>(handle); }

// ----------------------------------------------------------------------------- gspmm
std::tuple<Tensor, Tensor, Tensor> gspmm(const Tensor& indptr, const Tensor& indices, const optional<Tensor>& eids, int64_t num_cols,
                                         std::string op, std::string reduce, const optional<Tensor>& ufeat,
                                         const optional<Tensor>& efeat, int64_t plan_handle, int64_t flags) {
  const char* what = "mi355x_graph::gspmm";
  MGX_DEVICE_GUARD(indptr, what);
  mgx_csr csr = make_csr(indptr, indices, eids, num_cols, what);
  int opc = op_code(op, what);
  const int red = reduce_code(reduce, what);
  TORCH_CHECK(opc != MGX_OP_DOT, what, ": 'dot' is a g-SDDMM op");
  Tensor U, E;
  if (opc != MGX_OP_COPY_RHS) {
    TORCH_CHECK(ufeat.has_value(), what, ": op '", op, "' needs ufeat");
    U = as_f32(*ufeat, indptr, what);
    TORCH_CHECK(U.dim() >= 1 && U.size(0) == num_cols, what, ": ufeat must have num_cols = ", num_cols, " rows");
  }
  if (opc != MGX_OP_COPY_LHS) {
    TORCH_CHECK(efeat.has_value(), what, ": op '", op, "' needs efeat");
    E = as_f32(*efeat, indptr, what);
    TORCH_CHECK(E.dim() >= 1 && E.size(0) == csr.nnz, what, ": efeat must have nnz = ", csr.nnz, " rows");
    if (opc == MGX_OP_SUB) { E = E.neg(); opc = MGX_OP_ADD; }          // the rewrites DGL's ops/spmm.py applies
    else if (opc == MGX_OP_DIV) { E = E.reciprocal(); opc = MGX_OP_MUL; }
  }
  int64_t u_len = 0, e_len = 0, out_len = 0;
  std::vector<int64_t> shape = feature_shape(U.defined() ? &U : nullptr, E.defined() ? &E : nullptr, u_len, e_len, out_len, what);
  shape.insert(shape.begin(), csr.num_rows);
  const Tensor& ref = U.defined() ? U : E;
  Tensor out = at::empty(shape, ref.options());
  const bool want_arg = red == MGX_REDUCE_MAX || red == MGX_REDUCE_MIN;
  Tensor arg_u = at::empty((want_arg && opc != MGX_OP_COPY_RHS) ? at::IntArrayRef(shape) : at::IntArrayRef({0}), indptr.options());
  Tensor arg_e = at::empty((want_arg && opc != MGX_OP_COPY_LHS) ? at::IntArrayRef(shape) : at::IntArrayRef({0}), indptr.options());
  const mgx_spmm_plan* plan = want_arg ? nullptr : plan_of(plan_handle);
  Tensor partial;
  const int64_t slots = plan ? plan->num_slots + (plan->rest ? plan->rest->num_slots : 0) : 0;  // a two-part plan keeps its split rows in `rest`
  if (slots > 0) partial = at::empty({slots, out_len}, ref.options());
  check_status(mgx_spmm_csr(&csr, plan, opc, red, U.defined() ? U.data_ptr<float>() : nullptr, E.defined() ? E.data_ptr<float>() : nullptr,
                            u_len, e_len, out_len, nullptr, nullptr, nullptr, nullptr, out.data_ptr<float>(),
                            arg_u.numel() ? arg_u.data_ptr() : nullptr, arg_e.numel() ? arg_e.data_ptr() : nullptr,
                            partial.defined() ? partial.data_ptr<float>() : nullptr, (int32_t)(flags & MGX_SPMM_SHORT_ROWS), stream_of(indptr)),
               what);
  return std::make_tuple(out, arg_u, arg_e);
}

// ----------------------------------------------------------------------------- gsddmm (CSR walk, output by edge id)
Tensor gsddmm(const Tensor& indptr, const Tensor& indices, const optional<Tensor>& eids, int64_t num_cols, std::string op,
              const optional<Tensor>& lhs, const optional<Tensor>& rhs, std::string lhs_target, std::string rhs_target,
              int64_t plan_handle) {
  const char* what = "mi355x_graph::gsddmm";
  MGX_DEVICE_GUARD(indptr, what);
  mgx_csr csr = make_csr(indptr, indices, eids, num_cols, what);
  const int opc = op_code(op, what);
  const int lt = target_code(lhs_target, what), rt = target_code(rhs_target, what);
  Tensor L, R;
  auto rows_of = [&](int t) { return t == MGX_TARGET_U ? num_cols : (t == MGX_TARGET_V ? csr.num_rows : csr.nnz); };
  if (opc != MGX_OP_COPY_RHS) {
    TORCH_CHECK(lhs.has_value(), what, ": op '", op, "' needs lhs");
    L = as_f32(*lhs, indptr, what);
    TORCH_CHECK(L.dim() >= 1 && L.size(0) == rows_of(lt), what, ": lhs has ", L.size(0), " rows, its target has ", rows_of(lt));
  }
  if (opc != MGX_OP_COPY_LHS) {
    TORCH_CHECK(rhs.has_value(), what, ": op '", op, "' needs rhs");
    R = as_f32(*rhs, indptr, what);
    TORCH_CHECK(R.dim() >= 1 && R.size(0) == rows_of(rt), what, ": rhs has ", R.size(0), " rows, its target has ", rows_of(rt));
  }
  const Tensor& ref = L.defined() ? L : R;
  std::vector<int64_t> shape{csr.nnz};
  int64_t l_len = L.defined() ? trailing(L) : 0, r_len = R.defined() ? trailing(R) : 0, out_len = 0, reduce_size = 1;
  if (opc == MGX_OP_DOT) {
    TORCH_CHECK(L.dim() >= 2 && L.sizes().slice(1) == R.sizes().slice(1), what, ": dot needs operands of equal feature shape");
    reduce_size = L.size(L.dim() - 1);
    for (int64_t d = 1; d + 1 < L.dim(); ++d) shape.push_back(L.size(d));
    shape.push_back(1);
    out_len = reduce_size ? l_len / reduce_size : 0;
  } else {
    if (L.defined() && R.defined())
      TORCH_CHECK(L.sizes().slice(1) == R.sizes().slice(1), what, ": operand shapes ", L.sizes(), " and ", R.sizes(),
                  " need broadcasting; call mi355x_graph.ops.gsddmm (it builds the offset tables)");
    for (int64_t d = 1; d < ref.dim(); ++d) shape.push_back(ref.size(d));
    out_len = trailing(ref);
  }
  Tensor out = at::empty(shape, ref.options());
  check_status(mgx_sddmm_csr(&csr, plan_of(plan_handle), opc, L.defined() ? L.data_ptr<float>() : nullptr,
                             R.defined() ? R.data_ptr<float>() : nullptr, lt, rt, l_len, r_len, out_len, reduce_size, nullptr, nullptr,
                             out.data_ptr<float>(), stream_of(indptr)),
               what);
  return out;
}

// COO walk (graphs that keep their edge list: src / dst in edge-id order) -- the form batched small graphs take
// (main_dgl_molhiv_gcn.py:50-52: the gathers behind edges.src[...] of the UDF message)
Tensor gsddmm_coo(const Tensor& src, const Tensor& dst, int64_t num_src, int64_t num_dst, std::string op, const optional<Tensor>& lhs,
                  const optional<Tensor>& rhs, std::string lhs_target, std::string rhs_target) {
  const char* what = "mi355x_graph::gsddmm_coo";
  MGX_DEVICE_GUARD(src, what);
  TORCH_CHECK(src.is_cuda() && dst.is_cuda() && src.device() == dst.device(), what, ": runs on MI355X (HIP) tensors only");
  TORCH_CHECK(src.dim() == 1 && src.sizes() == dst.sizes() && src.scalar_type() == dst.scalar_type() && src.is_contiguous() &&
                  dst.is_contiguous(), what, ": src / dst must be matching contiguous 1-D index tensors");
  const int bits = idx_bits(src, what);
  const int64_t nnz = src.numel();
  const int opc = op_code(op, what);
  const int lt = target_code(lhs_target, what), rt = target_code(rhs_target, what);
  auto rows_of = [&](int t) { return t == MGX_TARGET_U ? num_src : (t == MGX_TARGET_V ? num_dst : nnz); };
  Tensor L, R;
  if (opc != MGX_OP_COPY_RHS) {
    TORCH_CHECK(lhs.has_value(), what, ": op '", op, "' needs lhs");
    L = as_f32(*lhs, src, what);
    TORCH_CHECK(L.dim() >= 1 && L.size(0) == rows_of(lt), what, ": lhs has ", L.size(0), " rows, its target has ", rows_of(lt));
  }
  if (opc != MGX_OP_COPY_LHS) {
    TORCH_CHECK(rhs.has_value(), what, ": op '", op, "' needs rhs");
    R = as_f32(*rhs, src, what);
    TORCH_CHECK(R.dim() >= 1 && R.size(0) == rows_of(rt), what, ": rhs has ", R.size(0), " rows, its target has ", rows_of(rt));
  }
  const Tensor& ref = L.defined() ? L : R;
  std::vector<int64_t> shape{nnz};
  int64_t l_len = L.defined() ? trailing(L) : 0, r_len = R.defined() ? trailing(R) : 0, out_len = 0, reduce_size = 1;
  if (opc == MGX_OP_DOT) {
    TORCH_CHECK(L.dim() >= 2 && L.sizes().slice(1) == R.sizes().slice(1), what, ": dot needs operands of equal feature shape");
    reduce_size = L.size(L.dim() - 1);
    for (int64_t d = 1; d + 1 < L.dim(); ++d) shape.push_back(L.size(d));
    shape.push_back(1);
    out_len = reduce_size ? l_len / reduce_size : 0;
  } else {
    if (L.defined() && R.defined())
      TORCH_CHECK(L.sizes().slice(1) == R.sizes().slice(1), what, ": operand shapes ", L.sizes(), " and ", R.sizes(),
                  " need broadcasting; call mi355x_graph.ops.gsddmm (it builds the offset tables)");
    for (int64_t d = 1; d < ref.dim(); ++d) shape.push_back(ref.size(d));
    out_len = trailing(ref);
  }
  Tensor out = at::empty(shape, ref.options());
  check_status(mgx_sddmm_coo(num_src, num_dst, nnz, src.data_ptr(), dst.data_ptr(), bits, opc, L.defined() ? L.data_ptr<float>() : nullptr,
                             R.defined() ? R.data_ptr<float>() : nullptr, lt, rt, l_len, r_len, out_len, reduce_size, nullptr, nullptr,
                             out.data_ptr<float>(), stream_of(src)),
               what);
  return out;
}

// ----------------------------------------------------------------------------- edge softmax
Tensor softmax_ws(const mgx_spmm_plan* plan, int64_t H, const Tensor& like) {
  if (!plan || plan->num_slots + plan->num_hubs == 0) return Tensor();
  return at::empty({(plan->num_slots + plan->num_hubs) * 2 * H}, like.options());
}

Tensor edge_softmax_fwd(const Tensor& indptr, const Tensor& indices, const optional<Tensor>& eids, int64_t num_cols, const Tensor& z,
                        int64_t plan_handle) {
  const char* what = "mi355x_graph::edge_softmax_fwd";
  MGX_DEVICE_GUARD(indptr, what);
  mgx_csr csr = make_csr(indptr, indices, eids, num_cols, what);
  Tensor zc = as_f32(z, indptr, what);
  TORCH_CHECK(zc.dim() >= 1 && zc.size(0) == csr.nnz, what, ": logits must have nnz = ", csr.nnz, " rows");
  const int64_t H = trailing(zc);
  Tensor a = at::empty_like(zc);
  Tensor ws = softmax_ws(plan_of(plan_handle), H, zc);
  check_status(mgx_edge_softmax_fwd(&csr, plan_of(plan_handle), H, zc.data_ptr<float>(), a.data_ptr<float>(),
                                    ws.defined() ? ws.data_ptr<float>() : nullptr, stream_of(indptr)),
               what);
  return a;
}

Tensor edge_softmax_bwd(const Tensor& indptr, const Tensor& indices, const optional<Tensor>& eids, int64_t num_cols, const Tensor& a,
                        const Tensor& da, int64_t plan_handle) {
  const char* what = "mi355x_graph::edge_softmax_bwd";
  MGX_DEVICE_GUARD(indptr, what);
  mgx_csr csr = make_csr(indptr, indices, eids, num_cols, what);
  Tensor ac = as_f32(a, indptr, what), dac = as_f32(da, indptr, what);
  TORCH_CHECK(ac.sizes() == dac.sizes() && ac.dim() >= 1 && ac.size(0) == csr.nnz, what, ": a and da must both be [nnz, ...]");
  const int64_t H = trailing(ac);
  Tensor dz = at::empty_like(ac);
  Tensor ws = softmax_ws(plan_of(plan_handle), H, ac);
  check_status(mgx_edge_softmax_bwd(&csr, plan_of(plan_handle), H, ac.data_ptr<float>(), dac.data_ptr<float>(), dz.data_ptr<float>(),
                                    ws.defined() ? ws.data_ptr<float>() : nullptr, stream_of(indptr)),
               what);
  return dz;
}

// ----------------------------------------------------------------------------- segment reduce
Tensor segment_reduce(const Tensor& offsets, const Tensor& x, std::string reduce) {
  const char* what = "mi355x_graph::segment_reduce";
  MGX_DEVICE_GUARD(x, what);
  TORCH_CHECK(x.is_cuda() && offsets.is_cuda() && offsets.device() == x.device(), what, ": runs on MI355X (HIP) tensors only");
  TORCH_CHECK(offsets.scalar_type() == at::kLong && offsets.dim() == 1 && offsets.numel() >= 1, what, ": offsets must be int64 [segments + 1]");
  Tensor xc = as_f32(x, offsets, what);
  Tensor off = offsets.contiguous();
  const int64_t nseg = off.numel() - 1, D = trailing(xc);
  std::vector<int64_t> shape{nseg};
  for (int64_t d = 1; d < xc.dim(); ++d) shape.push_back(xc.size(d));
  Tensor out = at::empty(shape, xc.options());
  check_status(mgx_segment_reduce(nseg, off.data_ptr<int64_t>(), D, reduce_code(reduce, what), xc.data_ptr<float>(), out.data_ptr<float>(),
                                  nullptr, stream_of(x)),
               what);
  return out;
}

// ----------------------------------------------------------------------------- formats (integer work, bit-exact)
std::tuple<Tensor, Tensor, Tensor> coo_to_csr(const Tensor& row, const Tensor& col, int64_t num_rows, int64_t num_cols) {
  const char* what = "mi355x_graph::coo_to_csr";
  MGX_DEVICE_GUARD(row, what);
  TORCH_CHECK(row.is_cuda() && col.is_cuda() && row.device() == col.device(), what, ": runs on MI355X (HIP) tensors only");
  TORCH_CHECK(row.dim() == 1 && row.sizes() == col.sizes() && row.scalar_type() == col.scalar_type(), what, ": row / col must match");
  const int bits = idx_bits(row, what);
  (void)num_cols;
  Tensor r = row.contiguous(), c = col.contiguous();
  const int64_t nnz = r.numel();
  Tensor indptr = at::empty({num_rows + 1}, r.options()), indices = at::empty({nnz}, r.options()), eids = at::empty({nnz}, r.options());
  const int64_t ws_bytes = mgx_coo_to_csr_workspace(num_rows, nnz, bits);
  TORCH_CHECK(ws_bytes >= 0, what, ": ", mgx_last_error());
  Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, r.options().dtype(at::kByte));
  check_status(mgx_coo_to_csr(num_rows, nnz, r.data_ptr(), c.data_ptr(), bits, indptr.data_ptr(), indices.data_ptr(), eids.data_ptr(),
                              ws.data_ptr(), ws_bytes, stream_of(row)),
               what);
  return std::make_tuple(indptr, indices, eids);
}

std::tuple<Tensor, Tensor, Tensor> csr_transpose(const Tensor& indptr, const Tensor& indices, const optional<Tensor>& eids, int64_t num_cols) {
  const char* what = "mi355x_graph::csr_transpose";
  MGX_DEVICE_GUARD(indptr, what);
  mgx_csr csr = make_csr(indptr, indices, eids, num_cols, what);
  Tensor ip = at::empty({num_cols + 1}, indptr.options()), ix = at::empty({csr.nnz}, indptr.options()), ei = at::empty({csr.nnz}, indptr.options());
  const int64_t ws_bytes = mgx_csr_transpose_workspace(num_cols, csr.nnz, csr.idx_bits);
  TORCH_CHECK(ws_bytes >= 0, what, ": ", mgx_last_error());
  Tensor ws = at::empty({std::max<int64_t>(ws_bytes, 1)}, indptr.options().dtype(at::kByte));
  check_status(mgx_csr_transpose(&csr, ip.data_ptr(), ix.data_ptr(), ei.data_ptr(), ws.data_ptr(), ws_bytes, stream_of(indptr)), what);
  return std::make_tuple(ip, ix, ei);
}

Tensor in_degrees(const Tensor& indptr) {
  const char* what = "mi355x_graph::in_degrees";
  MGX_DEVICE_GUARD(indptr, what);
  TORCH_CHECK(indptr.is_cuda(), what, " runs on MI355X (HIP) tensors");
  TORCH_CHECK(indptr.dim() == 1 && indptr.numel() >= 1 && indptr.is_contiguous(), what, ": indptr [rows + 1]");
  const int bits = idx_bits(indptr, what);
  Tensor deg = at::empty({indptr.numel() - 1}, indptr.options());
  check_status(mgx_csr_degrees(indptr.numel() - 1, indptr.data_ptr(), bits, deg.data_ptr(), stream_of(indptr)), what);
  return deg;
}

}  // namespace

TORCH_LIBRARY(mi355x_graph, m) {
  m.def("gspmm(Tensor indptr, Tensor indices, Tensor? eids, int num_cols, str op, str reduce, Tensor? ufeat, Tensor? efeat, "
        "int plan=0, int flags=0) -> (Tensor, Tensor, Tensor)");
  m.def("gsddmm(Tensor indptr, Tensor indices, Tensor? eids, int num_cols, str op, Tensor? lhs, Tensor? rhs, str lhs_target, "
        "str rhs_target, int plan=0) -> Tensor");
  m.def("gsddmm_coo(Tensor src, Tensor dst, int num_src, int num_dst, str op, Tensor? lhs, Tensor? rhs, str lhs_target, "
        "str rhs_target) -> Tensor");
  m.def("edge_softmax_fwd(Tensor indptr, Tensor indices, Tensor? eids, int num_cols, Tensor z, int plan=0) -> Tensor");
  m.def("edge_softmax_bwd(Tensor indptr, Tensor indices, Tensor? eids, int num_cols, Tensor a, Tensor da, int plan=0) -> Tensor");
  m.def("segment_reduce(Tensor offsets, Tensor x, str reduce) -> Tensor");
  m.def("coo_to_csr(Tensor row, Tensor col, int num_rows, int num_cols) -> (Tensor, Tensor, Tensor)");
  m.def("csr_transpose(Tensor indptr, Tensor indices, Tensor? eids, int num_cols) -> (Tensor, Tensor, Tensor)");
  m.def("in_degrees(Tensor indptr) -> Tensor");
}

// ROCm PyTorch dispatches HIP tensors under the CUDA key
TORCH_LIBRARY_IMPL(mi355x_graph, CUDA, m) {
  m.impl("gspmm", &gspmm);
  m.impl("gsddmm", &gsddmm);
  m.impl("gsddmm_coo", &gsddmm_coo);
  m.impl("edge_softmax_fwd", &edge_softmax_fwd);
  m.impl("edge_softmax_bwd", &edge_softmax_bwd);
  m.impl("segment_reduce", &segment_reduce);
  m.impl("coo_to_csr", &coo_to_csr);
  m.impl("csr_transpose", &csr_transpose);
  m.impl("in_degrees", &in_degrees);
}

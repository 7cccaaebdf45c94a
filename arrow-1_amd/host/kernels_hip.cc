// kernels_hip.cc -- the Register* hooks of the hot-path kernel files, each kernel's exec being a
// call into the C ABI (include/amdarrow.h).  Mirrors what the reference registers:
//   RegisterVectorSelection   compute/kernels/vector_selection.cc:2130-2187
//   RegisterVectorSort        compute/kernels/vector_sort.cc:1739-1765
//   RegisterScalarComparison  compute/kernels/scalar_compare.cc:100-220
//   RegisterScalarArithmetic  compute/kernels/scalar_arithmetic.cc:414-606
//   RegisterHashAggregateBasic + Grouper + GroupBy   compute/kernels/hash_aggregate.cc:313-437, 770-966, 1039-1062
#include <cstring>

#include "amdarrow_compute.h"

#include <cmath>
#include <cstdlib>

namespace amdarrow {
namespace compute {
namespace internal {

namespace {

const Type::type kNumeric[] = {Type::UINT8, Type::INT8, Type::UINT16, Type::INT16, Type::UINT32,
                               Type::INT32, Type::UINT64, Type::INT64, Type::FLOAT, Type::DOUBLE};

int64_t ValueBytes(const DataType& t, int64_t n) { return t.bit_width() == 1 ? (n + 7) / 8 : n * t.bit_width() / 8; }

std::function<Result<std::shared_ptr<DataType>>(const std::vector<ValueDescr>&)> FixedOut(std::shared_ptr<DataType> t) {
  return [t](const std::vector<ValueDescr>&) -> Result<std::shared_ptr<DataType>> { return t; };
}
std::function<Result<std::shared_ptr<DataType>>(const std::vector<ValueDescr>&)> FirstArgOut() {
  return [](const std::vector<ValueDescr>& d) -> Result<std::shared_ptr<DataType>> { return d[0].type; };
}

// ---- array_filter: PrimitiveFilter (vector_selection.cc:786-835) ------------------------------------
Status ExecFilter(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const FilterOptions*>(ctx->options());
  const int null_sel = opts ? static_cast<int>(opts->null_selection_behavior) : AMDARROW_DROP;
  const ArrayData& values = *batch[0].array();
  const ArrayData& filter = *batch[1].array();
  amdarrow_array v = values.abi(), f = filter.abi();
  int64_t out_len = 0;  // GetFilterOutputSize (:792), then PreallocateData (:244-259)
  int rc = amdarrow_filter_output_size(ctx->abi(), &f, null_sel, &out_len);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  const bool allocate_validity = (values.buffers[0] ? values.null_count : 0) != 0 || (filter.buffers[0] ? filter.null_count : 0) != 0;
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*values.type, out_len)));
  std::shared_ptr<Buffer> nbuf;
  if (allocate_validity) { AMD_ASSIGN_OR_RAISE(nbuf, ctx->AllocateBitmap(out_len)); }
  amdarrow_out o;
  rc = amdarrow_filter(ctx->abi(), &v, &f, null_sel, vbuf->mutable_data(), nbuf ? nbuf->mutable_data() : nullptr, out_len, &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(values.type, o.length, {nbuf, vbuf}, o.null_count));
  return Status::OK();
}

// ---- array_take: PrimitiveTake (vector_selection.cc:493-530) ---------------------------------------
Status ExecTake(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const TakeOptions*>(ctx->options());
  const ArrayData& values = *batch[0].array();
  const ArrayData& indices = *batch[1].array();
  amdarrow_array v = values.abi(), i = indices.abi();
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*values.type, indices.length)));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(indices.length));  // always allocated (:505-510)
  amdarrow_out o;
  const int rc = amdarrow_take(ctx->abi(), &v, &i, !opts || opts->boundscheck, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(values.type, o.length, {nbuf, vbuf}, o.null_count));
  return Status::OK();
}

// ---- array_take / array_filter for BINARY / STRING: VarBinaryImpl :1493-1573, BinaryFilter :840-1124 -------
Result<std::shared_ptr<ArrayData>> TakeBinary(KernelContext* ctx, const ArrayData& values, const ArrayData& indices, bool boundscheck) {
  amdarrow_binary_array v = values.binary_abi();
  amdarrow_array i = indices.abi();
  AMD_ASSIGN_OR_RAISE(auto obuf, ctx->Allocate(4 * (indices.length + 1)));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(indices.length));
  amdarrow_out o;
  int64_t nbytes = 0;
  int rc = amdarrow_binary_take_offsets(ctx->abi(), &v, &i, boundscheck, reinterpret_cast<int32_t*>(obuf->mutable_data()),
                                        nbuf->mutable_data(), &nbytes, &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  AMD_ASSIGN_OR_RAISE(auto dbuf, ctx->Allocate(nbytes));
  rc = amdarrow_binary_take_data(ctx->abi(), &v, &i, reinterpret_cast<const int32_t*>(obuf->data()), dbuf->mutable_data());
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  auto out = ArrayData::Make(values.type, o.length, {nbuf, obuf}, o.null_count);
  out->buffers.push_back(dbuf);
  return out;
}
Status ExecTakeBinary(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const TakeOptions*>(ctx->options());
  AMD_ASSIGN_OR_RAISE(auto r, TakeBinary(ctx, *batch[0].array(), *batch[1].array(), !opts || opts->boundscheck));
  *out = Datum(r);
  return Status::OK();
}
Status ExecFilterBinary(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  // Filter == Take(GetTakeIndices(filter)) (the identity ValidateFilter checks, vector_selection_test.cc:74-100), with
  // BinaryFilter's conventions for the validity buffer and the null count (:1090-1112)
  const auto* opts = static_cast<const FilterOptions*>(ctx->options());
  const auto null_sel = opts ? opts->null_selection_behavior : FilterOptions::DROP;
  const ArrayData& values = *batch[0].array();
  const ArrayData& filter = *batch[1].array();
  AMD_ASSIGN_OR_RAISE(auto indices, GetTakeIndices(filter, null_sel, ctx->exec_context()));
  AMD_ASSIGN_OR_RAISE(auto taken, TakeBinary(ctx, values, *indices, /*boundscheck=*/false));
  const int64_t vnc = values.buffers[0] ? values.null_count : 0, fnc = filter.buffers[0] ? filter.null_count : 0;
  if (vnc == 0 && fnc == 0) taken->buffers[0] = nullptr;
  taken->null_count = (vnc == 0 && (null_sel == FilterOptions::DROP || fnc == 0)) ? 0 : kUnknownNullCount;
  *out = Datum(taken);
  return Status::OK();
}

// ---- the other array_take / array_filter kernels (vector_selection.cc:1127-1781, registered :2133-2176) -------------
// FSBImpl, VarBinaryImpl<LargeBinaryType>, ListImpl<List / LargeList / Map>, FSLImpl, StructImpl over Selection::VisitTake,
// NullTake / NullFilter, DictionaryTake / DictionaryFilter, ExtensionTake / ExtensionFilter.  One recursive function per
// direction: the device resolves every output row to its source row once (amdarrow_take_rows = VisitTake + FinishCommon),
// the per-layout calls consume the resolved rows, child arrays are taken with the generated child indices exactly as
// the reference's Finish() does.
Result<std::shared_ptr<ArrayData>> TakeAny(KernelContext* ctx, const ArrayData& values, const ArrayData& indices, bool boundscheck);
Result<std::shared_ptr<ArrayData>> FilterAny(KernelContext* ctx, const ArrayData& values, const ArrayData& filter,
                                             FilterOptions::NullSelectionBehavior null_sel);

std::shared_ptr<DataType> IndexTypeOf(const DataType& dict_type) { return dict_type.index_type() ? dict_type.index_type() : int32(); }
// the indices of a dictionary array as a primitive array of the index type (DictionaryArray::indices)
std::shared_ptr<ArrayData> DictIndices(const ArrayData& d) {
  return ArrayData::Make(IndexTypeOf(*d.type), d.length, {d.buffers[0], d.buffers[1]}, d.null_count, d.offset);
}
Result<std::shared_ptr<ArrayData>> PrimitiveTake(KernelContext* ctx, const ArrayData& values, const ArrayData& indices, bool boundscheck) {
  amdarrow_array v = values.abi(), i = indices.abi();
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*values.type, indices.length)));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(indices.length));
  amdarrow_out o;
  const int rc = amdarrow_take(ctx->abi(), &v, &i, boundscheck, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return ArrayData::Make(values.type, o.length, {nbuf, vbuf}, o.null_count);
}

Result<std::shared_ptr<ArrayData>> TakeAny(KernelContext* ctx, const ArrayData& values, const ArrayData& indices, bool boundscheck) {
  const Type::type id = values.type->id();
  const int64_t n = indices.length;
  if (values.type->bit_width() > 0) return PrimitiveTake(ctx, values, indices, boundscheck);
  if (is_binary_like(id)) return TakeBinary(ctx, values, indices, boundscheck);
  if (id == Type::DICTIONARY) {  // DictionaryTake :1151-1160
    AMD_ASSIGN_OR_RAISE(auto t, PrimitiveTake(ctx, *DictIndices(values), indices, boundscheck));
    auto out = ArrayData::Make(values.type, t->length, t->buffers, t->null_count);
    out->dictionary = values.dictionary;
    return out;
  }
  if (id == Type::EXTENSION) {  // ExtensionTake :1177-1187: the storage array, re-wrapped
    auto storage = std::make_shared<ArrayData>(values);
    storage->type = values.type->storage_type();
    AMD_ASSIGN_OR_RAISE(auto t, TakeAny(ctx, *storage, indices, boundscheck));
    t->type = values.type;
    return t;
  }
  // Selection::VisitTake: bounds check, validity, null count, and the source row of every output row
  amdarrow_array i = indices.abi();
  AMD_ASSIGN_OR_RAISE(auto rows, ctx->Allocate(8 * n));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(n));
  amdarrow_out o;
  const uint8_t* vvalid = !values.buffers.empty() && values.buffers[0] ? values.buffers[0]->data() : nullptr;
  int rc = amdarrow_take_rows(ctx->abi(), vvalid, values.length, values.offset, vvalid ? values.null_count : 0, &i, boundscheck,
                              reinterpret_cast<int64_t*>(rows->mutable_data()), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  const auto* src = reinterpret_cast<const int64_t*>(rows->data());
  if (id == Type::NA) {  // NullTake :1132-1140
    auto out = ArrayData::Make(values.type, n, {nullptr}, n);
    return out;
  }
  if (id == Type::FIXED_SIZE_BINARY || id == Type::DECIMAL128 || id == Type::DECIMAL256) {  // FSBImpl :1575-1606
    const int w = values.type->byte_width();
    AMD_ASSIGN_OR_RAISE(auto dbuf, ctx->Allocate(n * w));
    rc = amdarrow_gather_fixed_width(ctx->abi(), values.buffers[1] ? values.buffers[1]->data() : nullptr, values.offset, w, src, n,
                                     dbuf->mutable_data());
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    return ArrayData::Make(values.type, n, {nbuf, dbuf}, o.null_count);
  }
  if (is_large_binary_like(id) || id == Type::LIST || id == Type::LARGE_LIST || id == Type::MAP) {
    amdarrow_var_array v = values.var_abi();
    const int ow = v.offset_bits / 8;
    AMD_ASSIGN_OR_RAISE(auto obuf, ctx->Allocate(ow * (n + 1)));
    int64_t total = 0;
    rc = amdarrow_var_take_offsets(ctx->abi(), &v, src, n, obuf->mutable_data(), &total);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    if (is_large_binary_like(id)) {  // VarBinaryImpl<LargeBinaryType> :1493-1573
      AMD_ASSIGN_OR_RAISE(auto dbuf, ctx->Allocate(total));
      rc = amdarrow_var_take_data(ctx->abi(), &v, src, n, obuf->data(), dbuf->mutable_data());
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
      auto out = ArrayData::Make(values.type, n, {nbuf, obuf}, o.null_count);
      out->buffers.push_back(dbuf);
      return out;
    }
    // ListImpl :1608-1680: child = Take(values.values(), child_indices, NoBoundsCheck)
    AMD_ASSIGN_OR_RAISE(auto cbuf, ctx->Allocate(ow * total));
    rc = amdarrow_list_take_child_indices(ctx->abi(), &v, src, n, obuf->data(), cbuf->mutable_data());
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    auto child_idx = ArrayData::Make(ow == 4 ? int32() : int64(), total, {nullptr, cbuf}, 0);
    AMD_ASSIGN_OR_RAISE(auto child, TakeAny(ctx, *values.child_data[0], *child_idx, false));
    auto out = ArrayData::Make(values.type, n, {nbuf, obuf}, o.null_count);
    out->child_data = {child};
    return out;
  }
  if (id == Type::FIXED_SIZE_LIST) {  // FSLImpl :1682-1724
    const int ls = values.type->list_size();
    AMD_ASSIGN_OR_RAISE(auto cbuf, ctx->Allocate(8 * n * ls));
    AMD_ASSIGN_OR_RAISE(auto cbits, ctx->AllocateBitmap(n * ls));
    rc = amdarrow_fixed_size_list_child_indices(ctx->abi(), src, n, ls, values.offset, reinterpret_cast<int64_t*>(cbuf->mutable_data()),
                                                cbits->mutable_data());
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    auto child_idx = ArrayData::Make(int64(), n * ls, {o.null_count ? cbits : nullptr, cbuf}, o.null_count * ls);
    AMD_ASSIGN_OR_RAISE(auto child, TakeAny(ctx, *values.child_data[0], *child_idx, false));
    auto out = ArrayData::Make(values.type, n, {nbuf}, o.null_count);
    out->child_data = {child};
    return out;
  }
  if (id == Type::STRUCT) {  // StructImpl :1736-1763: the fields are taken with the same selection, no bounds check
    auto out = ArrayData::Make(values.type, n, {nbuf}, o.null_count);
    out->child_names = values.child_names;
    for (const auto& field : values.child_data) {
      AMD_ASSIGN_OR_RAISE(auto taken, TakeAny(ctx, *field->Slice(values.offset, values.length), indices, false));
      out->child_data.push_back(taken);
    }
    return out;
  }
  return Status::NotImplemented("take: values of type " + values.type->ToString());
}

Result<std::shared_ptr<ArrayData>> FilterAny(KernelContext* ctx, const ArrayData& values, const ArrayData& filter,
                                             FilterOptions::NullSelectionBehavior null_sel) {
  const Type::type id = values.type->id();
  if (id == Type::DICTIONARY) {  // DictionaryFilter :1162-1172
    AMD_ASSIGN_OR_RAISE(auto t, FilterAny(ctx, *DictIndices(values), filter, null_sel));
    auto out = ArrayData::Make(values.type, t->length, t->buffers, t->null_count);
    out->dictionary = values.dictionary;
    return out;
  }
  if (id == Type::EXTENSION) {  // ExtensionFilter :1189-1199
    auto storage = std::make_shared<ArrayData>(values);
    storage->type = values.type->storage_type();
    AMD_ASSIGN_OR_RAISE(auto t, FilterAny(ctx, *storage, filter, null_sel));
    t->type = values.type;
    return t;
  }
  if (values.type->bit_width() > 0) {  // PrimitiveFilter through its own kernel
    Datum out;
    FilterOptions opts(null_sel);
    KernelContext kctx(ctx->exec_context());
    kctx.SetOptions(&opts);
    ExecBatch batch;
    batch.values = {Datum(std::make_shared<ArrayData>(values)), Datum(std::make_shared<ArrayData>(filter))};
    batch.length = values.length;
    AMD_RETURN_NOT_OK(ExecFilter(&kctx, batch, &out));
    return out.array();
  }
  if (id == Type::NA) {  // NullFilter :1142-1147
    amdarrow_array f = filter.abi();
    int64_t out_len = 0;
    const int rc = amdarrow_filter_output_size(ctx->abi(), &f, static_cast<int>(null_sel), &out_len);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    return ArrayData::Make(values.type, out_len, {nullptr}, out_len);
  }
  // FilterExec<Impl> visits the rows GetTakeIndices lists, in that order (StructFilter :1765-1781 says so itself)
  AMD_ASSIGN_OR_RAISE(auto indices, GetTakeIndices(filter, null_sel, ctx->exec_context()));
  AMD_ASSIGN_OR_RAISE(auto taken, TakeAny(ctx, values, *indices, /*boundscheck=*/false));
  if (is_binary_like(id) || is_large_binary_like(id)) {  // BinaryFilter's conventions, :1090-1112
    const int64_t vnc = values.buffers[0] ? values.null_count : 0, fnc = filter.buffers[0] ? filter.null_count : 0;
    if (vnc == 0 && fnc == 0) taken->buffers[0] = nullptr;
    taken->null_count = (vnc == 0 && (null_sel == FilterOptions::DROP || fnc == 0)) ? 0 : kUnknownNullCount;
  }
  return taken;
}
Status ExecTakeAny(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const TakeOptions*>(ctx->options());
  AMD_ASSIGN_OR_RAISE(auto r, TakeAny(ctx, *batch[0].array(), *batch[1].array(), !opts || opts->boundscheck));
  *out = Datum(r);
  return Status::OK();
}
Status ExecFilterAny(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const FilterOptions*>(ctx->options());
  AMD_ASSIGN_OR_RAISE(auto r, FilterAny(ctx, *batch[0].array(), *batch[1].array(), opts ? opts->null_selection_behavior : FilterOptions::DROP));
  *out = Datum(r);
  return Status::OK();
}

// ---- array_sort_indices (vector_sort.cc:558-575) ---------------------------------------------------
Status SortChunks(KernelContext* ctx, const std::vector<std::shared_ptr<ArrayData>>& chunks, SortOrder order, Datum* out) {
  std::vector<amdarrow_array> abi;
  int64_t total = 0;
  for (const auto& c : chunks) { abi.push_back(c->abi()); total += c->length; }
  AMD_ASSIGN_OR_RAISE(auto buf, ctx->Allocate(8 * total));
  const int rc = amdarrow_sort_indices(ctx->abi(), abi.data(), static_cast<int>(abi.size()),
                                       order == SortOrder::Descending ? AMDARROW_DESCENDING : AMDARROW_ASCENDING,
                                       reinterpret_cast<uint64_t*>(buf->mutable_data()));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(uint64(), total, {nullptr, buf}, 0));  // OUTPUT_NOT_NULL (vector_sort.cc:1745-1747)
  return Status::OK();
}
Status ExecArraySortIndices(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const ArraySortOptions*>(ctx->options());
  return SortChunks(ctx, {batch[0].array()}, opts ? opts->order : SortOrder::Ascending, out);
}

// ---- partition_nth_indices (PartitionNthToIndices, vector_sort.cc:321-358) ---------------------------
// The contract (and what the reference's tests validate, vector_sort_test.cc:132-150): a permutation
// with the pivot-th smallest non-null, non-NaN value at `pivot`, nothing larger before it, nothing
// smaller after it, NaNs and nulls at the end; std::nth_element leaves the order inside the two sides
// unspecified.  The fully sorted order is one such answer, and on this device the radix sort IS the
// selection algorithm of choice (a quickselect's data-dependent rounds each cost a pass over HBM), so
// the kernel is sort_indices ascending.  pivot == length: the identity (:339-343).
Status ExecPartitionNth(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto* opts = static_cast<const PartitionNthOptions*>(ctx->options());
  if (!opts) return Status::Invalid("NthToIndices requires PartitionNthOptions");  // :328-330
  const ArrayData& arr = *batch[0].array();
  if (opts->pivot > arr.length) return Status::IndexError("NthToIndices index out of bound");  // :335-337
  if (opts->pivot == arr.length) {
    std::vector<uint64_t> iota(static_cast<size_t>(arr.length));
    for (int64_t i = 0; i < arr.length; ++i) iota[i] = static_cast<uint64_t>(i);
    AMD_ASSIGN_OR_RAISE(auto buf, ctx->Allocate(8 * arr.length));
    if (arr.length) {
      const int rc = amdarrow_memcpy_h2d(ctx->abi(), buf->mutable_data(), iota.data(), 8 * arr.length);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    *out = Datum(ArrayData::Make(uint64(), arr.length, {nullptr, buf}, 0));
    return Status::OK();
  }
  return SortChunks(ctx, {batch[0].array()}, SortOrder::Ascending, out);
}

// ---- compare / arithmetic (ScalarBinary applicators, codegen_internal.h:758-932) -------------------
struct BinaryOperands {
  amdarrow_array l, r;
  const amdarrow_array* lp = nullptr;
  const amdarrow_array* rp = nullptr;
  const void* ls = nullptr;
  const void* rs = nullptr;
  int64_t length = 0;
  bool null_scalar = false;
};
BinaryOperands Unpack(const ExecBatch& batch) {
  BinaryOperands b;
  if (batch[0].is_array()) { b.l = batch[0].array()->abi(); b.lp = &b.l; b.length = b.l.length; }
  else { b.ls = &batch[0].scalar()->bits; b.null_scalar |= !batch[0].scalar()->is_valid; }
  if (batch[1].is_array()) { b.r = batch[1].array()->abi(); b.rp = &b.r; b.length = b.r.length; }
  else { b.rs = &batch[1].scalar()->bits; b.null_scalar |= !batch[1].scalar()->is_valid; }
  return b;
}
// A null scalar operand makes every slot null (AllNullShortCircuit, exec.cc:326-348); the value
// buffer content is unspecified in the reference, here it is the op applied to a zero scalar.
Status FinishAllNull(KernelContext* ctx, int64_t length, std::shared_ptr<Buffer> nbuf, amdarrow_out* o) {
  if (length > 0) {
    const int rc = amdarrow_memset(ctx->abi(), nbuf->mutable_data(), 0, (length + 7) / 8);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  }
  o->null_count = length;
  return Status::OK();
}

template <int kOp>
Status ExecCompare(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  BinaryOperands b = Unpack(batch);
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->AllocateBitmap(b.length));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(b.length));
  amdarrow_out o;
  const int rc = amdarrow_compare(ctx->abi(), kOp, b.lp, b.ls, b.rp, b.rs, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  if (b.null_scalar) AMD_RETURN_NOT_OK(FinishAllNull(ctx, b.length, nbuf, &o));
  *out = Datum(ArrayData::Make(boolean(), o.length, {nbuf, vbuf}, o.null_count));
  return Status::OK();
}
template <int kOp>
Status ExecArith(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  BinaryOperands b = Unpack(batch);
  auto type = batch[0].type();
  if (b.null_scalar && kOp >= AMDARROW_DIV) {
    // ScalarBinaryNotNull with a null scalar never calls the op (codegen_internal.h:853-881): no error can be raised
    AMD_ASSIGN_OR_RAISE(auto vbuf0, ctx->Allocate(ValueBytes(*type, b.length)));
    AMD_ASSIGN_OR_RAISE(auto nbuf0, ctx->AllocateBitmap(b.length));
    if (b.length > 0) {
      const int rc0 = amdarrow_memset(ctx->abi(), vbuf0->mutable_data(), 0, ValueBytes(*type, b.length));
      if (rc0 != AMDARROW_OK) return Status::FromAbi(rc0);
    }
    *out = Datum(ArrayData::Make(type, b.length, {nbuf0, vbuf0}, b.length));
    return Status::OK();
  }
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*type, b.length)));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(b.length));
  amdarrow_out o;
  const int rc = amdarrow_arith(ctx->abi(), kOp, b.lp, b.ls, b.rp, b.rs, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  if (b.null_scalar) AMD_RETURN_NOT_OK(FinishAllNull(ctx, b.length, nbuf, &o));
  *out = Datum(ArrayData::Make(type, o.length, {nbuf, vbuf}, o.null_count));
  return Status::OK();
}

// negate / negate_checked (scalar_arithmetic.cc:253-294): ScalarUnary / ScalarUnaryNotNull over one array
template <int kOp>
Status ExecNegate(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  if (!batch[0].is_array()) return Status::NotImplemented("negate of a scalar is left to the CPU caller");
  const amdarrow_array a = batch[0].array()->abi();
  auto type = batch[0].type();
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*type, a.length)));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(a.length));
  amdarrow_out o;
  const int rc = amdarrow_arith(ctx->abi(), kOp, &a, nullptr, nullptr, nullptr, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(type, o.length, {nbuf, vbuf}, o.null_count));
  return Status::OK();
}

// ---- mask-building scalars: scalar_boolean.cc / scalar_validity.cc / scalar_fill_null.cc ----
int ScalarState(const Scalar& s) {
  return !s.is_valid ? AMDARROW_SCALAR_NULL : (s.bits & 0xFF) ? AMDARROW_SCALAR_TRUE : AMDARROW_SCALAR_FALSE;
}
template <int kOp>
Status ExecBoolean(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  amdarrow_array l{}, r{};
  const amdarrow_array* lp = nullptr;
  const amdarrow_array* rp = nullptr;
  int ls = AMDARROW_SCALAR_NONE, rs = AMDARROW_SCALAR_NONE;
  int64_t length = 0;
  if (batch[0].is_array()) { l = batch[0].array()->abi(); lp = &l; length = l.length; }
  else ls = ScalarState(*batch[0].scalar());
  if (kOp != AMDARROW_INVERT) {
    if (batch[1].is_array()) { r = batch[1].array()->abi(); rp = &r; length = r.length; }
    else rs = ScalarState(*batch[1].scalar());
  }
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->AllocateBitmap(length));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(length));
  amdarrow_out o;
  const int rc = amdarrow_boolean(ctx->abi(), kOp, lp, ls, rp, rs, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  // the Kleene kernels drop the validity buffer where no slot can be null (scalar_boolean.cc:205-209 etc.)
  *out = Datum(ArrayData::Make(boolean(), o.length, {o.has_validity ? nbuf : nullptr, vbuf}, o.null_count));
  return Status::OK();
}
template <int kFunc>
Status ExecValidityTest(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  amdarrow_array a = batch[0].array()->abi();
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->AllocateBitmap(a.length));
  amdarrow_out o;
  const int rc = amdarrow_validity_test(ctx->abi(), kFunc, &a, vbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(boolean(), o.length, {nullptr, vbuf}, 0));  // OUTPUT_NOT_NULL (scalar_validity.cc:100)
  return Status::OK();
}
Status ExecIsNan(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  amdarrow_array a = batch[0].array()->abi();
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->AllocateBitmap(a.length));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(a.length));
  amdarrow_out o;
  const int rc = amdarrow_is_nan(ctx->abi(), &a, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(boolean(), o.length, {nbuf, vbuf}, o.null_count));
  return Status::OK();
}
Status ExecFillNull(KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const ArrayData& values = *batch[0].array();
  const Scalar& fill = *batch[1].scalar();
  amdarrow_array a = values.abi();
  AMD_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*values.type, a.length)));
  AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(a.length));
  amdarrow_out o;
  const int rc = amdarrow_fill_null(ctx->abi(), &a, fill.is_valid ? &fill.bits : nullptr, vbuf->mutable_data(),
                                    nbuf->mutable_data(), &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  *out = Datum(ArrayData::Make(values.type, o.length, {o.has_validity ? nbuf : nullptr, vbuf}, o.null_count));
  return Status::OK();
}

// ---- scalar aggregates: aggregate_basic.cc:58-130, aggregate_basic_internal.h:54-330 ----
// One amdarrow_aggregate call per batch yields everything a Consume of count / sum / mean / min_max needs.
struct AggState : KernelState {
  Type::type in_type = Type::NA;
  int64_t length = 0, valid = 0;  // rows seen, non-null rows
  uint64_t sum = 0, min = 0, max = 0;
  bool any = false;               // a batch has been consumed (min / max hold real identities)
};
enum AggFamily { kFamUnsigned, kFamSigned, kFamFloat };
AggFamily FamilyOf(Type::type t) { return is_floating(t) ? kFamFloat : is_signed_integer(t) ? kFamSigned : kFamUnsigned; }
double AsDouble(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
uint64_t FromDouble(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }

Result<std::unique_ptr<KernelState>> AggInit(KernelContext*, const std::shared_ptr<DataType>& type, const FunctionOptions*) {
  auto st = std::unique_ptr<AggState>(new AggState());
  st->in_type = type->id();
  return std::unique_ptr<KernelState>(std::move(st));
}
Status AggConsume(KernelContext* ctx, KernelState* state, const ExecBatch& batch) {
  auto* st = static_cast<AggState*>(state);
  amdarrow_array a = batch[0].array()->abi();
  amdarrow_agg r;
  const int rc = amdarrow_aggregate(ctx->abi(), &a, &r);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  st->length = a.length; st->valid = r.valid_count; st->sum = r.sum; st->min = r.min; st->max = r.max; st->any = true;
  return Status::OK();
}
// MergeFrom: SumImpl :73-78 (count, sum +=), MinMaxState::operator+= (:150-157, :173-180, :196-203), CountImpl :69-74
Status AggMerge(KernelContext*, KernelState&& src, KernelState* dst) {
  auto& o = static_cast<AggState&>(src);
  auto* st = static_cast<AggState*>(dst);
  if (!o.any) return Status::OK();
  if (!st->any) { const Type::type t = st->in_type; *st = o; st->in_type = t; return Status::OK(); }
  st->length += o.length;
  st->valid += o.valid;
  switch (FamilyOf(st->in_type)) {
    case kFamFloat:
      st->sum = FromDouble(AsDouble(st->sum) + AsDouble(o.sum));
      st->min = FromDouble(std::fmin(AsDouble(st->min), AsDouble(o.min)));
      st->max = FromDouble(std::fmax(AsDouble(st->max), AsDouble(o.max)));
      break;
    case kFamSigned:
      st->sum += o.sum;
      st->min = static_cast<uint64_t>(std::min(static_cast<int64_t>(st->min), static_cast<int64_t>(o.min)));
      st->max = static_cast<uint64_t>(std::max(static_cast<int64_t>(st->max), static_cast<int64_t>(o.max)));
      break;
    default:
      st->sum += o.sum;
      st->min = std::min(st->min, o.min);
      st->max = std::max(st->max, o.max);
      break;
  }
  return Status::OK();
}
std::shared_ptr<DataType> AccumulatorType(Type::type t) {  // FindAccumulatorType, aggregate_internal.h:29-50
  return is_floating(t) ? float64() : is_signed_integer(t) ? int64() : uint64();
}
Status SumFinalize(KernelContext*, KernelState* state, Datum* out) {  // SumImpl::Finalize :80-87
  auto* st = static_cast<AggState*>(state);
  auto type = AccumulatorType(st->in_type);
  *out = st->valid == 0 ? Datum(Scalar::MakeNull(type)) : Datum(Scalar::Make(type, st->sum));
  return Status::OK();
}
Status MeanFinalize(KernelContext*, KernelState* state, Datum* out) {  // MeanImpl::Finalize :94-102
  auto* st = static_cast<AggState*>(state);
  if (st->valid == 0) { *out = Datum(Scalar::MakeNull(float64())); return Status::OK(); }
  double sum;
  switch (FamilyOf(st->in_type)) {
    case kFamFloat: sum = AsDouble(st->sum); break;
    case kFamSigned: sum = static_cast<double>(static_cast<int64_t>(st->sum)); break;
    default: sum = static_cast<double>(st->sum); break;
  }
  *out = Datum(Scalar::Make(float64(), sum / static_cast<double>(st->valid)));
  return Status::OK();
}
Status CountFinalize(KernelContext* ctx, KernelState* state, Datum* out) {  // CountImpl::Finalize :76-89
  auto* st = static_cast<AggState*>(state);
  const auto* opts = static_cast<const CountOptions*>(ctx->options());
  const bool nulls = opts && opts->count_mode == CountOptions::COUNT_NULL;
  *out = Datum(Scalar::Make(int64(), nulls ? st->length - st->valid : st->valid));
  return Status::OK();
}
// value bits in the accumulator family -> a scalar of the INPUT type (MinMaxImpl::Finalize :254-267)
std::shared_ptr<Scalar> NarrowScalar(Type::type t, uint64_t bits) {
  auto type = TypeFromId(t);
  if (t == Type::FLOAT) return Scalar::Make(type, static_cast<float>(AsDouble(bits)));
  if (t == Type::DOUBLE) return Scalar::Make(type, AsDouble(bits));
  const int bw = type->bit_width();
  return Scalar::Make(type, bw >= 64 ? bits : (bits & ((uint64_t(1) << (bw == 1 ? 8 : bw)) - 1)));
}
Status MinMaxFinalize(KernelContext* ctx, KernelState* state, Datum* out) {
  auto* st = static_cast<AggState*>(state);
  const auto* opts = static_cast<const MinMaxOptions*>(ctx->options());
  const bool emit_null = opts && opts->null_handling == MinMaxOptions::EMIT_NULL;
  auto s = std::make_shared<Scalar>();
  s->type = struct_();
  s->field_names = {"min", "max"};
  const bool has_nulls = st->length - st->valid > 0;
  if (st->valid == 0 || (has_nulls && emit_null)) {
    s->fields = {Scalar::MakeNull(TypeFromId(st->in_type)), Scalar::MakeNull(TypeFromId(st->in_type))};
  } else {
    s->fields = {NarrowScalar(st->in_type, st->min), NarrowScalar(st->in_type, st->max)};
  }
  *out = Datum(s);
  return Status::OK();
}

// ---- meta functions ------------------------------------------------------------------------------------
// TakeAA .. TakeTC (vector_selection.cc:1914-2011).  Chunked VALUES are concatenated first, as the
// reference does (:1938-1940) -- once per call here, not once per index chunk (:1953-1956 notes the
// repeated copy as an inefficiency; the result is the same array).
Result<std::shared_ptr<ArrayData>> TakeAA(const std::shared_ptr<ArrayData>& values, const std::shared_ptr<ArrayData>& indices,
                                          const TakeOptions& options, ExecContext* ctx) {
  AMD_ASSIGN_OR_RAISE(Datum out, CallFunction("array_take", {Datum(values), Datum(indices)}, &options, ctx));
  return out.array();
}
Result<std::shared_ptr<ArrayData>> OneChunk(const ChunkedArray& values, ExecContext* ctx) {
  if (values.num_chunks() == 1) return values.chunk(0);  // :1928-1930
  return Concatenate(values.chunks(), ctx->memory_pool());
}
Result<std::shared_ptr<ChunkedArray>> TakeCA(const ChunkedArray& values, const std::shared_ptr<ArrayData>& indices,
                                             const TakeOptions& options, ExecContext* ctx) {
  AMD_ASSIGN_OR_RAISE(auto chunk, OneChunk(values, ctx));
  AMD_ASSIGN_OR_RAISE(auto taken, TakeAA(chunk, indices, options, ctx));
  return std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{taken}, values.type());
}
Result<std::shared_ptr<ChunkedArray>> TakeCC(const ChunkedArray& values, const ChunkedArray& indices, const TakeOptions& options,
                                             ExecContext* ctx) {
  std::vector<std::shared_ptr<ArrayData>> chunks(indices.num_chunks());
  if (indices.num_chunks() > 0) {
    AMD_ASSIGN_OR_RAISE(auto chunk, OneChunk(values, ctx));
    for (int i = 0; i < indices.num_chunks(); ++i) { AMD_ASSIGN_OR_RAISE(chunks[i], TakeAA(chunk, indices.chunk(i), options, ctx)); }
  }
  return std::make_shared<ChunkedArray>(std::move(chunks), values.type());
}
Result<std::shared_ptr<ChunkedArray>> TakeAC(const std::shared_ptr<ArrayData>& values, const ChunkedArray& indices,
                                             const TakeOptions& options, ExecContext* ctx) {
  std::vector<std::shared_ptr<ArrayData>> chunks(indices.num_chunks());
  for (int i = 0; i < indices.num_chunks(); ++i) { AMD_ASSIGN_OR_RAISE(chunks[i], TakeAA(values, indices.chunk(i), options, ctx)); }
  return std::make_shared<ChunkedArray>(std::move(chunks), values->type);
}
Result<std::shared_ptr<RecordBatch>> TakeRA(const RecordBatch& batch, const std::shared_ptr<ArrayData>& indices,
                                            const TakeOptions& options, ExecContext* ctx) {
  std::vector<std::shared_ptr<ArrayData>> columns(batch.num_columns());
  for (int j = 0; j < batch.num_columns(); ++j) { AMD_ASSIGN_OR_RAISE(columns[j], TakeAA(batch.column(j), indices, options, ctx)); }
  return RecordBatch::Make(batch.schema(), indices->length, std::move(columns));
}
Result<std::shared_ptr<Table>> TakeTA(const Table& table, const std::shared_ptr<ArrayData>& indices, const TakeOptions& options,
                                      ExecContext* ctx) {
  std::vector<std::shared_ptr<ChunkedArray>> columns(table.num_columns());
  for (int j = 0; j < table.num_columns(); ++j) { AMD_ASSIGN_OR_RAISE(columns[j], TakeCA(*table.column(j), indices, options, ctx)); }
  return Table::Make(table.schema(), std::move(columns));
}
Result<std::shared_ptr<Table>> TakeTC(const Table& table, const ChunkedArray& indices, const TakeOptions& options,
                                      ExecContext* ctx) {
  std::vector<std::shared_ptr<ChunkedArray>> columns(table.num_columns());
  for (int j = 0; j < table.num_columns(); ++j) { AMD_ASSIGN_OR_RAISE(columns[j], TakeCC(*table.column(j), indices, options, ctx)); }
  return Table::Make(table.schema(), std::move(columns));
}

// FilterRecordBatch (vector_selection.cc:1776-1797): the filter becomes take indices once, every
// column is gathered with them.
Result<std::shared_ptr<RecordBatch>> FilterRecordBatch(const RecordBatch& batch, const Datum& filter, const FilterOptions& options,
                                                       ExecContext* ctx) {
  if (batch.num_rows() != filter.length()) return Status::Invalid("Filter inputs must all be the same length");
  if (!filter.is_array()) return Status::NotImplemented("Filter should be array-like");
  AMD_ASSIGN_OR_RAISE(auto indices, GetTakeIndices(*filter.array(), options.null_selection_behavior, ctx));
  std::vector<std::shared_ptr<ArrayData>> columns(batch.num_columns());
  for (int i = 0; i < batch.num_columns(); ++i) {
    AMD_ASSIGN_OR_RAISE(columns[i], TakeAA(batch.column(i), indices, TakeOptions::NoBoundsCheck(), ctx));
  }
  return RecordBatch::Make(batch.schema(), indices->length, std::move(columns));
}
// FilterTable (vector_selection.cc:1799-1866): columns and filter are cut at the union of their
// chunk boundaries; each filter piece becomes take indices applied to that piece of every column;
// pieces that select nothing leave no chunk.
Result<std::shared_ptr<Table>> FilterTable(const Table& table, const Datum& filter, const FilterOptions& options, ExecContext* ctx) {
  if (table.num_rows() != filter.length()) return Status::Invalid("Filter inputs must all be the same length");
  if (table.num_rows() == 0) return Table::Make(table.schema(), table.columns(), 0);
  const int ncols = table.num_columns();
  std::vector<std::vector<std::shared_ptr<ArrayData>>> inputs(ncols + 1);
  for (int i = 0; i < ncols; ++i) inputs[i] = table.column(i)->chunks();
  if (filter.is_array()) inputs.back().push_back(filter.array());
  else if (filter.kind() == Datum::CHUNKED_ARRAY) inputs.back() = filter.chunked_array()->chunks();
  else return Status::NotImplemented("Filter should be array-like");
  inputs = RechunkArraysConsistently(inputs);
  std::vector<std::vector<std::shared_ptr<ArrayData>>> out_columns(ncols);
  int64_t out_rows = 0;
  for (size_t i = 0; i < inputs.back().size(); ++i) {
    AMD_ASSIGN_OR_RAISE(auto indices, GetTakeIndices(*inputs.back()[i], options.null_selection_behavior, ctx));
    if (indices->length == 0) continue;
    for (int c = 0; c < ncols; ++c) {
      AMD_ASSIGN_OR_RAISE(auto piece, TakeAA(inputs[c][i], indices, TakeOptions::NoBoundsCheck(), ctx));
      out_columns[c].push_back(std::move(piece));
    }
    out_rows += indices->length;
  }
  std::vector<std::shared_ptr<ChunkedArray>> out(ncols);
  for (int c = 0; c < ncols; ++c) out[c] = std::make_shared<ChunkedArray>(std::move(out_columns[c]), table.column(c)->type());
  return Table::Make(table.schema(), std::move(out), out_rows);
}

// FilterMetaFunction (vector_selection.cc:1877-1903)
class FilterMetaFunction : public MetaFunction {
 public:
  FilterMetaFunction() : MetaFunction("filter", Arity::Binary(), {}, &kDefaults) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override {
    if (!args[1].type() || args[1].type()->id() != Type::BOOL)
      return Status::NotImplemented("Filter argument must be boolean type");  // :1885-1887
    const auto& opts = *static_cast<const FilterOptions*>(options);
    if (args[0].kind() == Datum::RECORD_BATCH) {
      AMD_ASSIGN_OR_RAISE(auto out, FilterRecordBatch(*args[0].record_batch(), args[1], opts, ctx));
      return Datum(out);
    }
    if (args[0].kind() == Datum::TABLE) {
      AMD_ASSIGN_OR_RAISE(auto out, FilterTable(*args[0].table(), args[1], opts, ctx));
      return Datum(out);
    }
    return CallFunction("array_filter", args, options, ctx);
  }
  static FilterOptions kDefaults;
};
FilterOptions FilterMetaFunction::kDefaults = FilterOptions::Defaults();

// TakeMetaFunction (vector_selection.cc:2026-2072)
class TakeMetaFunction : public MetaFunction {
 public:
  TakeMetaFunction() : MetaFunction("take", Arity::Binary(), {}, &kDefaults) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override {
    const auto& opts = *static_cast<const TakeOptions*>(options);
    const Datum::Kind index_kind = args[1].kind();
    switch (args[0].kind()) {
      case Datum::ARRAY:
        if (index_kind == Datum::ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeAA(args[0].array(), args[1].array(), opts, ctx)); return Datum(r); }
        if (index_kind == Datum::CHUNKED_ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeAC(args[0].array(), *args[1].chunked_array(), opts, ctx)); return Datum(r); }
        break;
      case Datum::CHUNKED_ARRAY:
        if (index_kind == Datum::ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeCA(*args[0].chunked_array(), args[1].array(), opts, ctx)); return Datum(r); }
        if (index_kind == Datum::CHUNKED_ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeCC(*args[0].chunked_array(), *args[1].chunked_array(), opts, ctx)); return Datum(r); }
        break;
      case Datum::RECORD_BATCH:
        if (index_kind == Datum::ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeRA(*args[0].record_batch(), args[1].array(), opts, ctx)); return Datum(r); }
        break;
      case Datum::TABLE:
        if (index_kind == Datum::ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeTA(*args[0].table(), args[1].array(), opts, ctx)); return Datum(r); }
        if (index_kind == Datum::CHUNKED_ARRAY) { AMD_ASSIGN_OR_RAISE(auto r, TakeTC(*args[0].table(), *args[1].chunked_array(), opts, ctx)); return Datum(r); }
        break;
      default:
        break;
    }
    return Status::NotImplemented("Unsupported types for take operation");  // :2067-2070
  }
  static TakeOptions kDefaults;
};
TakeOptions TakeMetaFunction::kDefaults = TakeOptions::Defaults();

// SortIndicesMetaFunction (vector_sort.cc:1565-1711): arrays and chunked arrays by their one key,
// batches and tables by SortOptions::sort_keys (column names).
class SortIndicesMetaFunction : public MetaFunction {
 public:
  SortIndicesMetaFunction() : MetaFunction("sort_indices", Arity::Unary(), {}, &kDefaults) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override {
    const auto& so = *static_cast<const SortOptions*>(options);
    switch (args[0].kind()) {
      case Datum::ARRAY: {
        ArraySortOptions ao(FirstOrder(so));
        return CallFunction("array_sort_indices", args, &ao, ctx);  // :1598-1606
      }
      case Datum::CHUNKED_ARRAY:
        return SortChunked(*args[0].chunked_array(), FirstOrder(so), ctx);  // :1608-1630
      case Datum::RECORD_BATCH: {  // :1632-1668
        const RecordBatch& batch = *args[0].record_batch();
        if (so.sort_keys.empty()) return Status::Invalid("Must specify one or more sort keys");
        std::vector<std::vector<std::shared_ptr<ArrayData>>> keys;
        for (const auto& key : so.sort_keys) {
          auto column = batch.GetColumnByName(key.name);
          if (!column) return Status::Invalid("Nonexistent sort key column: " + key.name);
          keys.push_back({column});
        }
        if (keys.size() == 1) {
          ArraySortOptions ao(so.sort_keys[0].order);
          return CallFunction("array_sort_indices", {Datum(keys[0][0])}, &ao, ctx);
        }
        return SortMulti(keys, so, batch.num_rows(), ctx);
      }
      case Datum::TABLE: {  // :1670-1707
        const Table& table = *args[0].table();
        if (so.sort_keys.empty()) return Status::Invalid("Must specify one or more sort keys");
        std::vector<std::vector<std::shared_ptr<ArrayData>>> keys;
        for (const auto& key : so.sort_keys) {
          auto column = table.GetColumnByName(key.name);
          if (!column) return Status::Invalid("Nonexistent sort key column: " + key.name);
          keys.push_back(column->chunks());
        }
        if (keys.size() == 1) return SortChunked(*table.GetColumnByName(so.sort_keys[0].name), so.sort_keys[0].order, ctx);
        return SortMulti(keys, so, table.num_rows(), ctx);
      }
      default:
        break;
    }
    return Status::NotImplemented("Unsupported types for sort_indices operation");
  }
  static SortOptions kDefaults;

 private:
  static SortOrder FirstOrder(const SortOptions& so) { return so.sort_keys.empty() ? SortOrder::Ascending : so.sort_keys[0].order; }
  static Status CheckSortable(const std::shared_ptr<DataType>& t) {
    if (!t || t->bit_width() < 8 || t->id() == Type::HALF_FLOAT) return Status::NotImplemented("sort_indices: numeric types only");
    return Status::OK();
  }
  static Result<Datum> SortChunked(const ChunkedArray& values, SortOrder order, ExecContext* ctx) {
    AMD_RETURN_NOT_OK(CheckSortable(values.type()));
    KernelContext kctx(ctx);
    Datum out;
    AMD_RETURN_NOT_OK(SortChunks(&kctx, values.chunks(), order, &out));
    return out;
  }
  // RadixRecordBatchSorter / MultipleKey{RecordBatch,Table}Sorter (vector_sort.cc:955-1549)
  static Result<Datum> SortMulti(const std::vector<std::vector<std::shared_ptr<ArrayData>>>& keys, const SortOptions& so,
                                 int64_t num_rows, ExecContext* ctx) {
    std::vector<amdarrow_array> abi;
    std::vector<int> nchunks, orders;
    for (size_t k = 0; k < keys.size(); ++k) {
      for (const auto& c : keys[k]) { AMD_RETURN_NOT_OK(CheckSortable(c->type)); abi.push_back(c->abi()); }
      if (keys[k].empty()) return Status::Invalid("sort_indices: a sort key column without chunks");
      nchunks.push_back(static_cast<int>(keys[k].size()));
      orders.push_back(so.sort_keys[k].order == SortOrder::Descending ? AMDARROW_DESCENDING : AMDARROW_ASCENDING);
    }
    AMD_ASSIGN_OR_RAISE(auto buf, AllocateBuffer(8 * num_rows, ctx->memory_pool()));
    const int rc = amdarrow_sort_indices_multi(ctx->memory_pool()->ctx(), abi.data(), nchunks.data(), orders.data(),
                                               static_cast<int>(keys.size()), reinterpret_cast<uint64_t*>(buf->mutable_data()));
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    return Datum(ArrayData::Make(uint64(), num_rows, {nullptr, buf}, 0));
  }
};
SortOptions SortIndicesMetaFunction::kDefaults = SortOptions::Defaults();

// ---- hash aggregates -------------------------------------------------------------------------------------
// KernelState of hash_sum / hash_count / hash_mean: device sums + counts that grow with the groups
// (GroupedSumImpl / GroupedCountImpl, hash_aggregate.cc:466-622).
struct GroupedState : KernelState {
  int func = AMDARROW_HASH_SUM;
  int count_mode = AMDARROW_COUNT_NON_NULL;
  std::shared_ptr<DataType> in_type;
  std::shared_ptr<Buffer> sums, counts;
  int64_t num_groups = 0, capacity = 0;
};
Status Reserve(KernelContext* ctx, GroupedState* st, int64_t new_groups) {  // MaybeReserve :450-459
  if (new_groups <= st->num_groups) return Status::OK();
  if (new_groups > st->capacity) {
    const int64_t cap = std::max<int64_t>(1024, 2 * new_groups);
    AMD_ASSIGN_OR_RAISE(auto s, ctx->Allocate(8 * cap));
    AMD_ASSIGN_OR_RAISE(auto c, ctx->Allocate(8 * cap));
    int rc = amdarrow_memset(ctx->abi(), s->mutable_data(), 0, 8 * cap);
    if (rc == AMDARROW_OK) rc = amdarrow_memset(ctx->abi(), c->mutable_data(), 0, 8 * cap);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    if (st->num_groups > 0) {
      // carry the existing groups over (stream-ordered device copies)
      rc = amdarrow_memcpy_d2d(ctx->abi(), s->mutable_data(), st->sums->data(), 8 * st->num_groups);
      if (rc == AMDARROW_OK) rc = amdarrow_memcpy_d2d(ctx->abi(), c->mutable_data(), st->counts->data(), 8 * st->num_groups);
      if (rc == AMDARROW_OK) rc = amdarrow_ctx_synchronize(ctx->abi());  // the old buffers are released below
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    st->sums = s; st->counts = c; st->capacity = cap;
  }
  st->num_groups = new_groups;
  return Status::OK();
}
HashAggregateKernel MakeHashKernel(int func) {  // MakeKernel :770-807
  HashAggregateKernel k;
  k.signature.in_types = {InputType::Primitive(), InputType::Exact(Type::UINT32, ValueDescr::ARRAY),
                          InputType::Exact(Type::UINT32, ValueDescr::SCALAR)};
  k.signature.out_type = [func](const std::vector<ValueDescr>& d) -> Result<std::shared_ptr<DataType>> {
    if (func == AMDARROW_HASH_COUNT) return int64();
    const Type::type t = d[0].type->id();
    if (func == AMDARROW_HASH_MEAN || is_floating(t)) return float64();
    return is_signed_integer(t) ? int64() : uint64();  // FindAccumulatorType, aggregate_internal.h:29-50
  };
  k.init = [func](KernelContext*, const std::shared_ptr<DataType>& in, const FunctionOptions* opts) -> Result<std::unique_ptr<KernelState>> {
    auto st = std::unique_ptr<GroupedState>(new GroupedState());
    st->func = func;
    st->in_type = in;
    if (func == AMDARROW_HASH_COUNT && opts) st->count_mode = static_cast<const CountOptions*>(opts)->count_mode;
    return std::unique_ptr<KernelState>(std::move(st));
  };
  k.consume = [](KernelContext* ctx, KernelState* state, const ExecBatch& batch) -> Status {
    auto* st = static_cast<GroupedState*>(state);
    uint32_t ng;
    memcpy(&ng, &batch[2].scalar()->bits, 4);
    AMD_RETURN_NOT_OK(Reserve(ctx, st, ng));
    amdarrow_array v = batch[0].array()->abi();
    const ArrayData& ids = *batch[1].array();
    const int rc = amdarrow_hash_accumulate(ctx->abi(), st->func, &v,
                                            reinterpret_cast<const uint32_t*>(ids.buffers[1]->data()) + ids.offset,
                                            st->count_mode, st->sums->mutable_data(), reinterpret_cast<int64_t*>(st->counts->mutable_data()));
    return Status::FromAbi(rc);
  };
  k.merge = [](KernelContext*, KernelState&&, KernelState*) -> Status {
    return Status::NotImplemented("Merge hashed aggregations");  // :795-798
  };
  k.finalize = [](KernelContext* ctx, KernelState* state, Datum* out) -> Status {
    auto* st = static_cast<GroupedState*>(state);
    std::shared_ptr<DataType> ot = st->func == AMDARROW_HASH_COUNT ? int64()
                                 : (st->func == AMDARROW_HASH_MEAN || is_floating(st->in_type->id())) ? float64()
                                 : is_signed_integer(st->in_type->id()) ? int64() : uint64();
    if (st->num_groups == 0) { *out = Datum(ArrayData::Make(ot, 0, {nullptr, nullptr}, 0)); return Status::OK(); }
    AMD_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(st->num_groups));
    amdarrow_out o;
    const int rc = amdarrow_hash_finalize(ctx->abi(), st->func, static_cast<int>(st->in_type->id()), st->num_groups,
                                          st->sums->mutable_data(), reinterpret_cast<const int64_t*>(st->counts->data()),
                                          nbuf->mutable_data(), &o);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    *out = Datum(ArrayData::Make(ot, st->num_groups, {o.has_validity ? nbuf : nullptr, st->sums}, o.null_count));
    return Status::OK();
  };
  return k;
}

// hash_min_max: GroupedMinMaxImpl (hash_aggregate.cc:627-768).  State = the reference's four builders
// on the device; the output is the StructArray{min, max} whose children share one validity bitmap.
struct GroupedMinMaxState : KernelState {
  int null_handling = AMDARROW_MINMAX_SKIP;
  std::shared_ptr<DataType> in_type;
  std::shared_ptr<Buffer> mins, maxes, has_values, has_nulls;
  int64_t num_groups = 0, capacity = 0;
};
Status ReserveMinMax(KernelContext* ctx, GroupedMinMaxState* st, int64_t new_groups) {  // MaybeReserve + resize lambdas :713-721
  if (new_groups <= st->num_groups) return Status::OK();
  const int w = st->in_type->bit_width() / 8;
  if (new_groups > st->capacity) {
    const int64_t cap = std::max<int64_t>(1024, 2 * new_groups);
    AMD_ASSIGN_OR_RAISE(auto mn, ctx->Allocate(cap * w));
    AMD_ASSIGN_OR_RAISE(auto mx, ctx->Allocate(cap * w));
    AMD_ASSIGN_OR_RAISE(auto hv, ctx->AllocateBitmap(cap));
    AMD_ASSIGN_OR_RAISE(auto hn, ctx->AllocateBitmap(cap));
    if (st->num_groups > 0) {
      int rc = amdarrow_memcpy_d2d(ctx->abi(), mn->mutable_data(), st->mins->data(), st->num_groups * w);
      if (rc == AMDARROW_OK) rc = amdarrow_memcpy_d2d(ctx->abi(), mx->mutable_data(), st->maxes->data(), st->num_groups * w);
      if (rc == AMDARROW_OK) rc = amdarrow_memcpy_d2d(ctx->abi(), hv->mutable_data(), st->has_values->data(), (st->num_groups + 7) / 8);
      if (rc == AMDARROW_OK) rc = amdarrow_memcpy_d2d(ctx->abi(), hn->mutable_data(), st->has_nulls->data(), (st->num_groups + 7) / 8);
      if (rc == AMDARROW_OK) rc = amdarrow_ctx_synchronize(ctx->abi());  // the old buffers are released below
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    st->mins = mn; st->maxes = mx; st->has_values = hv; st->has_nulls = hn; st->capacity = cap;
  }
  const int rc = amdarrow_hash_minmax_init(ctx->abi(), static_cast<int>(st->in_type->id()), st->mins->mutable_data(),
                                           st->maxes->mutable_data(), st->num_groups, new_groups - st->num_groups);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  st->num_groups = new_groups;
  return Status::OK();
}
HashAggregateKernel MakeHashMinMaxKernel() {
  HashAggregateKernel k;
  k.signature.in_types = {InputType::Primitive(), InputType::Exact(Type::UINT32, ValueDescr::ARRAY),
                          InputType::Exact(Type::UINT32, ValueDescr::SCALAR)};
  k.signature.out_type = FixedOut(struct_());  // struct_({field("min", type), field("max", type)}) :750-752
  k.init = [](KernelContext*, const std::shared_ptr<DataType>& in, const FunctionOptions* opts) -> Result<std::unique_ptr<KernelState>> {
    if (in->bit_width() < 8 || in->id() == Type::HALF_FLOAT)
      return Status::NotImplemented("Grouped MinMax data of type " + in->ToString());  // :674-684
    auto st = std::unique_ptr<GroupedMinMaxState>(new GroupedMinMaxState());
    st->in_type = in;
    if (opts) st->null_handling = static_cast<const MinMaxOptions*>(opts)->null_handling == MinMaxOptions::EMIT_NULL
                                      ? AMDARROW_MINMAX_EMIT_NULL : AMDARROW_MINMAX_SKIP;
    return std::unique_ptr<KernelState>(std::move(st));
  };
  k.consume = [](KernelContext* ctx, KernelState* state, const ExecBatch& batch) -> Status {
    auto* st = static_cast<GroupedMinMaxState*>(state);
    uint32_t ng;
    memcpy(&ng, &batch[2].scalar()->bits, 4);
    AMD_RETURN_NOT_OK(ReserveMinMax(ctx, st, ng));
    amdarrow_array v = batch[0].array()->abi();
    const ArrayData& ids = *batch[1].array();
    return Status::FromAbi(amdarrow_hash_minmax_accumulate(ctx->abi(), &v, reinterpret_cast<const uint32_t*>(ids.buffers[1]->data()) + ids.offset,
                                                           st->mins->mutable_data(), st->maxes->mutable_data(),
                                                           st->has_values->mutable_data(), st->has_nulls->mutable_data()));
  };
  k.merge = [](KernelContext*, KernelState&&, KernelState*) -> Status { return Status::NotImplemented("Merge hashed aggregations"); };
  k.finalize = [](KernelContext* ctx, KernelState* state, Datum* out) -> Status {  // :729-748
    auto* st = static_cast<GroupedMinMaxState*>(state);
    auto result = ArrayData::Make(struct_(), st->num_groups, {nullptr, nullptr}, 0);
    result->child_names = {"min", "max"};
    std::shared_ptr<Buffer> nbuf;
    amdarrow_out o;
    o.length = 0; o.null_count = 0; o.has_validity = 1;
    if (st->num_groups > 0) {
      AMD_ASSIGN_OR_RAISE(nbuf, ctx->AllocateBitmap(st->num_groups));
      const int rc = amdarrow_hash_minmax_finalize(ctx->abi(), st->num_groups, st->null_handling, st->has_values->data(),
                                                   st->has_nulls->data(), nbuf->mutable_data(), &o);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    result->child_data = {ArrayData::Make(st->in_type, st->num_groups, {nbuf, st->mins}, o.null_count),
                          ArrayData::Make(st->in_type, st->num_groups, {nbuf, st->maxes}, o.null_count)};
    *out = Datum(result);
    return Status::OK();
  };
  return k;
}

// CompareFunction::DispatchBest (scalar_compare.cc:75-98) and ArithmeticFunction::DispatchBest
// (scalar_arithmetic.cc:389-412): exact match first; else dictionaries decoded, a null operand takes the other's
// type, numbers meet at CommonNumeric.  (The timestamp / binary promotions of the comparison concern types that are
// not on this path.)  Function::Execute then casts the arguments.
struct PromotingFunction : ScalarFunction {
  using ScalarFunction::ScalarFunction;
  Result<const ScalarKernel*> DispatchBest(std::vector<ValueDescr>* values) const override {
    AMD_RETURN_NOT_OK(CheckArity(static_cast<int>(values->size())));
    auto exact = DispatchExact(*values);
    if (exact.ok()) return exact;
    EnsureDictionaryDecoded(values);
    if (values->size() == 2) {  // "Only promote types for binary functions"
      ReplaceNullWithOtherType(values);
      if (auto type = CommonNumeric(*values)) ReplaceTypes(type, values);
    }
    return DispatchExact(*values);
  }
};
using CompareFunction = PromotingFunction;
using ArithmeticFunction = PromotingFunction;

template <typename Fn>
void AddBinaryFunctions(FunctionRegistry* registry, const char* const* names, int n, bool compare, Fn make_exec) {
  for (int op = 0; op < n; ++op) {
    auto func = std::make_shared<PromotingFunction>(names[op], Arity::Binary());
    for (Type::type t : kNumeric) {
      ScalarKernel k;
      k.signature.in_types = {InputType::Exact(t), InputType::Exact(t)};
      k.signature.out_type = compare ? FixedOut(boolean()) : FirstArgOut();
      k.exec = make_exec(op);
      k.can_write_into_slices = false;
      (void)func->AddKernel(std::move(k));
    }
    if (compare) {  // boolean comparisons too (scalar_compare.cc:106-110)
      ScalarKernel k;
      k.signature.in_types = {InputType::Exact(Type::BOOL), InputType::Exact(Type::BOOL)};
      k.signature.out_type = FixedOut(boolean());
      k.exec = make_exec(op);
      k.can_write_into_slices = false;
      (void)func->AddKernel(std::move(k));
    }
    (void)registry->AddFunction(std::move(func));
  }
}

}  // namespace

Result<std::shared_ptr<ArrayData>> GetTakeIndices(const ArrayData& filter, FilterOptions::NullSelectionBehavior null_selection,
                                                  ExecContext* ctx) {
  const auto& pool = ctx->memory_pool();
  amdarrow_array f = filter.abi();
  int64_t out_len = 0;
  int rc = amdarrow_filter_output_size(pool->ctx(), &f, static_cast<int>(null_selection), &out_len);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  AMD_ASSIGN_OR_RAISE(auto vbuf, AllocateBuffer(4 * out_len, pool));  // uint16 when length <= 65535, else uint32 (:229-236)
  AMD_ASSIGN_OR_RAISE(auto nbuf, AllocateBitmap(out_len, pool));
  amdarrow_out o;
  int width = 0;
  rc = amdarrow_take_indices(pool->ctx(), &f, static_cast<int>(null_selection), vbuf->mutable_data(), nbuf->mutable_data(), out_len,
                             &width, &o);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return ArrayData::Make(width == 16 ? uint16() : uint32(), o.length, {o.has_validity ? nbuf : nullptr, vbuf}, o.null_count);
}

void RegisterVectorSelection(FunctionRegistry* registry) {
  {
    static FilterOptions defaults = FilterOptions::Defaults();
    auto f = std::make_shared<VectorFunction>("array_filter", Arity::Binary(), FunctionDoc{}, &defaults);
    VectorKernel k;  // {match::Primitive(), boolean()} -> PrimitiveFilter (:2133)
    k.signature.in_types = {InputType::Primitive(), InputType::Exact(Type::BOOL)};
    k.signature.out_type = FirstArgOut();
    k.exec = ExecFilter;
    (void)f->AddKernel(std::move(k));
    for (Type::type t : {Type::BINARY, Type::STRING}) {  // {match::BinaryLike(), boolean()} -> BinaryFilter (:2134)
      VectorKernel kb;
      kb.signature.in_types = {InputType::Exact(t), InputType::Exact(Type::BOOL)};
      kb.signature.out_type = FirstArgOut();
      kb.exec = ExecFilterBinary;
      (void)f->AddKernel(std::move(kb));
    }
    // LargeBinaryLike, FIXED_SIZE_BINARY, null, DECIMAL (= DECIMAL128: there is no decimal256 filter kernel), DICTIONARY,
    // EXTENSION, LIST, LARGE_LIST, FIXED_SIZE_LIST, STRUCT, MAP (:2135-2147)
    for (Type::type t : {Type::LARGE_BINARY, Type::LARGE_STRING, Type::FIXED_SIZE_BINARY, Type::NA, Type::DECIMAL128, Type::DICTIONARY,
                         Type::EXTENSION, Type::LIST, Type::LARGE_LIST, Type::FIXED_SIZE_LIST, Type::STRUCT, Type::MAP}) {
      VectorKernel ka;
      ka.signature.in_types = {InputType::Exact(t), InputType::Exact(Type::BOOL)};
      ka.signature.out_type = FirstArgOut();
      ka.exec = ExecFilterAny;
      (void)f->AddKernel(std::move(ka));
    }
    (void)registry->AddFunction(std::move(f));
    (void)registry->AddFunction(std::make_shared<FilterMetaFunction>());
  }
  {
    static TakeOptions defaults = TakeOptions::Defaults();
    auto f = std::make_shared<VectorFunction>("array_take", Arity::Binary(), FunctionDoc{}, &defaults);
    VectorKernel k;  // {match::Primitive(), match::Integer()} -> PrimitiveTake (:2160-2163)
    k.signature.in_types = {InputType::Primitive(), InputType::Integer()};
    k.signature.out_type = FirstArgOut();
    k.exec = ExecTake;
    k.can_execute_chunkwise = false;  // :2156
    (void)f->AddKernel(std::move(k));
    for (Type::type t : {Type::BINARY, Type::STRING}) {  // {match::BinaryLike(), match::Integer()} -> VarBinaryImpl (:2164)
      VectorKernel kb;
      kb.signature.in_types = {InputType::Exact(t), InputType::Integer()};
      kb.signature.out_type = FirstArgOut();
      kb.exec = ExecTakeBinary;
      kb.can_execute_chunkwise = false;
      (void)f->AddKernel(std::move(kb));
    }
    // LargeBinaryLike, FIXED_SIZE_BINARY, null, DECIMAL128, DECIMAL256, DICTIONARY, EXTENSION, LIST, LARGE_LIST,
    // FIXED_SIZE_LIST, STRUCT, MAP (:2166-2176)
    for (Type::type t : {Type::LARGE_BINARY, Type::LARGE_STRING, Type::FIXED_SIZE_BINARY, Type::NA, Type::DECIMAL128, Type::DECIMAL256,
                         Type::DICTIONARY, Type::EXTENSION, Type::LIST, Type::LARGE_LIST, Type::FIXED_SIZE_LIST, Type::STRUCT, Type::MAP}) {
      VectorKernel ka;
      ka.signature.in_types = {InputType::Exact(t), InputType::Integer()};
      ka.signature.out_type = FirstArgOut();
      ka.exec = ExecTakeAny;
      ka.can_execute_chunkwise = false;
      (void)f->AddKernel(std::move(ka));
    }
    (void)registry->AddFunction(std::move(f));
    (void)registry->AddFunction(std::make_shared<TakeMetaFunction>());
  }
}

void RegisterVectorSort(FunctionRegistry* registry) {
  static ArraySortOptions defaults = ArraySortOptions::Defaults();
  auto f = std::make_shared<VectorFunction>("array_sort_indices", Arity::Unary(), FunctionDoc{}, &defaults);
  for (Type::type t : kNumeric) {  // AddSortingKernels over NumericTypes() (:581-587)
    VectorKernel k;
    k.signature.in_types = {InputType::Exact(t, ValueDescr::ARRAY)};
    k.signature.out_type = FixedOut(uint64());
    k.exec = ExecArraySortIndices;
    k.mem_allocation = MemAllocation::PREALLOCATE;
    k.null_handling = NullHandling::OUTPUT_NOT_NULL;
    k.can_execute_chunkwise = false;
    (void)f->AddKernel(std::move(k));
  }
  (void)registry->AddFunction(std::move(f));
  (void)registry->AddFunction(std::make_shared<SortIndicesMetaFunction>());
  // partition_nth_indices: no default options (:1758-1764; CallFunction without them is Invalid)
  auto nth = std::make_shared<VectorFunction>("partition_nth_indices", Arity::Unary());
  for (Type::type t : kNumeric) {
    VectorKernel k;
    k.signature.in_types = {InputType::Exact(t, ValueDescr::ARRAY)};
    k.signature.out_type = FixedOut(uint64());
    k.exec = ExecPartitionNth;
    k.mem_allocation = MemAllocation::PREALLOCATE;
    k.null_handling = NullHandling::OUTPUT_NOT_NULL;
    k.can_execute_chunkwise = false;
    (void)nth->AddKernel(std::move(k));
  }
  (void)registry->AddFunction(std::move(nth));
}

template <int... kOps>
ArrayKernelExec PickCompare(int op) {
  static const ArrayKernelExec table[] = {ExecCompare<kOps>...};
  return table[op];
}
template <int... kOps>
ArrayKernelExec PickArith(int op) {
  static const ArrayKernelExec table[] = {ExecArith<kOps>...};
  return table[op];
}

void RegisterScalarComparison(FunctionRegistry* registry) {
  static const char* names[] = {"equal", "not_equal", "greater", "greater_equal", "less", "less_equal"};
  AddBinaryFunctions(registry, names, 6, true, [](int op) { return PickCompare<0, 1, 2, 3, 4, 5>(op); });
}
void RegisterScalarArithmetic(FunctionRegistry* registry) {
  static const char* names[] = {"add", "subtract", "multiply", "divide",
                                "add_checked", "subtract_checked", "multiply_checked", "divide_checked",
                                "power", "power_checked"};
  AddBinaryFunctions(registry, names, 10, false, [](int op) { return PickArith<0, 1, 2, 3, 4, 5, 6, 7, 8, 9>(op); });
  // MakeUnaryArithmeticFunction<Negate> over every number type, MakeUnarySignedArithmeticFunctionNotNull<NegateChecked>
  // over the signed and floating ones (:437-466, 589-596)
  for (int checked = 0; checked < 2; ++checked) {
    auto func = std::make_shared<PromotingFunction>(checked ? "negate_checked" : "negate", Arity::Unary());
    for (Type::type t : kNumeric) {
      if (checked && is_integer(t) && !is_signed_integer(t)) continue;
      ScalarKernel k;
      k.signature.in_types = {InputType::Exact(t)};
      k.signature.out_type = FirstArgOut();
      k.exec = checked ? ExecNegate<AMDARROW_NEGATE_CHECKED> : ExecNegate<AMDARROW_NEGATE>;
      k.can_write_into_slices = false;
      (void)func->AddKernel(std::move(k));
    }
    (void)registry->AddFunction(std::move(func));
  }
}

template <int... kOps>
ArrayKernelExec PickBoolean(int op) {
  static const ArrayKernelExec table[] = {ExecBoolean<kOps>...};
  return table[op];
}
// scalar_boolean.cc:521-542: invert, and, and_not, or, xor (INTERSECTION) and the Kleene variants
// (COMPUTED_PREALLOCATE, cannot write into slices)
void RegisterScalarBoolean(FunctionRegistry* registry) {
  static const char* names[] = {"invert", "and", "and_not", "or", "xor", "and_kleene", "and_not_kleene", "or_kleene"};
  for (int op = 0; op < 8; ++op) {
    const int arity = op == AMDARROW_INVERT ? 1 : 2;
    auto func = std::make_shared<ScalarFunction>(names[op], arity == 1 ? Arity::Unary() : Arity::Binary());
    ScalarKernel k;
    k.signature.in_types = std::vector<InputType>(arity, InputType::Exact(Type::BOOL));
    k.signature.out_type = FixedOut(boolean());
    k.exec = PickBoolean<0, 1, 2, 3, 4, 5, 6, 7>(op);
    k.can_write_into_slices = false;
    if (op >= AMDARROW_AND_KLEENE) k.null_handling = NullHandling::COMPUTED_PREALLOCATE;
    (void)func->AddKernel(std::move(k));
    (void)registry->AddFunction(std::move(func));
  }
}
// scalar_validity.cc:171-182: is_valid / is_null over ANY type, is_nan over float32 / float64
void RegisterScalarValidity(FunctionRegistry* registry) {
  struct Def { const char* name; ArrayKernelExec exec; };
  const Def defs[] = {{"is_valid", ExecValidityTest<0>}, {"is_null", ExecValidityTest<1>}};
  for (const auto& d : defs) {
    auto func = std::make_shared<ScalarFunction>(d.name, Arity::Unary());
    ScalarKernel k;
    k.signature.in_types = {InputType::Any()};
    k.signature.out_type = FixedOut(boolean());
    k.exec = d.exec;
    k.null_handling = NullHandling::OUTPUT_NOT_NULL;
    k.can_write_into_slices = false;
    (void)func->AddKernel(std::move(k));
    (void)registry->AddFunction(std::move(func));
  }
  auto func = std::make_shared<ScalarFunction>("is_nan", Arity::Unary());
  for (Type::type t : {Type::FLOAT, Type::DOUBLE}) {
    ScalarKernel k;
    k.signature.in_types = {InputType::Exact(t)};
    k.signature.out_type = FixedOut(boolean());
    k.exec = ExecIsNan;
    k.can_write_into_slices = false;
    (void)func->AddKernel(std::move(k));
  }
  (void)registry->AddFunction(std::move(func));
}
// scalar_fill_null.cc:229-240: {Array(ty), Scalar(ty)} -> ty for numeric and boolean types
void RegisterScalarFillNull(FunctionRegistry* registry) {
  auto func = std::make_shared<ScalarFunction>("fill_null", Arity::Binary());
  auto add = [&](Type::type t) {
    ScalarKernel k;
    k.signature.in_types = {InputType::Exact(t, ValueDescr::ARRAY), InputType::Exact(t, ValueDescr::SCALAR)};
    k.signature.out_type = FirstArgOut();
    k.exec = ExecFillNull;
    k.null_handling = NullHandling::COMPUTED_NO_PREALLOCATE;
    k.mem_allocation = MemAllocation::NO_PREALLOCATE;
    k.can_write_into_slices = false;
    (void)func->AddKernel(std::move(k));
  };
  for (Type::type t : kNumeric) add(t);
  add(Type::BOOL);
  (void)registry->AddFunction(std::move(func));
}

// aggregate_basic.cc:143-330: count over any type; sum / mean / min_max over boolean and numeric types
void RegisterScalarAggregateBasic(FunctionRegistry* registry) {
  static CountOptions count_defaults = CountOptions::Defaults();
  static MinMaxOptions minmax_defaults = MinMaxOptions::Defaults();
  struct Def { const char* name; const FunctionOptions* opts; bool any_type; std::function<Status(KernelContext*, KernelState*, Datum*)> finalize; };
  const Def defs[] = {{"count", &count_defaults, true, CountFinalize}, {"sum", nullptr, false, SumFinalize},
                      {"mean", nullptr, false, MeanFinalize}, {"min_max", &minmax_defaults, false, MinMaxFinalize}};
  for (const auto& d : defs) {
    auto f = std::make_shared<ScalarAggregateFunction>(d.name, Arity::Unary(), FunctionDoc{}, d.opts);
    auto add = [&](InputType in) {
      ScalarAggregateKernel k;
      k.signature.in_types = {std::move(in)};
      k.init = AggInit;
      k.consume = AggConsume;
      k.merge = AggMerge;
      k.finalize = d.finalize;
      (void)f->AddKernel(std::move(k));
    };
    if (d.any_type) {
      add(InputType::Primitive());  // the fixed-width types of this backend
    } else {
      add(InputType::Exact(Type::BOOL));
      for (Type::type t : kNumeric) add(InputType::Exact(t));
    }
    (void)registry->AddFunction(std::move(f));
  }
}

void RegisterHashAggregateBasic(FunctionRegistry* registry) {
  static CountOptions count_defaults = CountOptions::Defaults();
  struct Def { const char* name; int func; const FunctionOptions* opts; };
  const Def defs[] = {{"hash_count", AMDARROW_HASH_COUNT, &count_defaults}, {"hash_sum", AMDARROW_HASH_SUM, nullptr},
                      {"hash_mean", AMDARROW_HASH_MEAN, nullptr}};  // hash_mean: SURVEY 8a row c6
  for (const auto& d : defs) {
    auto f = std::make_shared<HashAggregateFunction>(d.name, Arity::Ternary(), FunctionDoc{}, d.opts);
    (void)f->AddKernel(MakeHashKernel(d.func));
    (void)registry->AddFunction(std::move(f));
  }
  static MinMaxOptions minmax_defaults = MinMaxOptions::Defaults();  // :1052-1057
  auto f = std::make_shared<HashAggregateFunction>("hash_min_max", Arity::Ternary(), FunctionDoc{}, &minmax_defaults);
  (void)f->AddKernel(MakeHashMinMaxKernel());
  (void)registry->AddFunction(std::move(f));
}

// ---- Grouper (api_aggregate.h:313-340, hash_aggregate.cc:313-437) ---------------------------------------
Grouper::~Grouper() {
  if (impl_) amdarrow_grouper_destroy(impl_);
  for (auto* v : var_) if (v) amdarrow_binary_grouper_destroy(v);
}
Result<std::unique_ptr<Grouper>> Grouper::Make(const std::vector<ValueDescr>& descrs, ExecContext* ctx) {
  auto g = std::unique_ptr<Grouper>(new Grouper());
  g->ctx_ = ctx;
  g->descrs_ = descrs;
  g->var_.assign(descrs.size(), nullptr);
  g->dictionaries_.assign(descrs.size(), nullptr);
  std::vector<int> types;
  for (size_t k = 0; k < descrs.size(); ++k) {  // GrouperImpl::Make, hash_aggregate.cc:313-356
    const auto& t = *descrs[k].type;
    if (is_binary_like(t.id()) || is_large_binary_like(t.id())) {  // VarLengthKeyEncoder<BinaryType / LargeBinaryType>
      const int rc = amdarrow_binary_grouper_create(ctx->memory_pool()->ctx(), is_binary_like(t.id()) ? 32 : 64, &g->var_[k]);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
      types.push_back(static_cast<int>(Type::UINT32));
    } else if (t.id() == Type::DICTIONARY) {  // DictionaryKeyEncoder: the index, fixed width
      types.push_back(static_cast<int>(IndexTypeOf(t)->id()));
    } else if (t.bit_width() > 0) {
      types.push_back(static_cast<int>(t.id()));
    } else {
      return Status::NotImplemented("Keys of type " + t.ToString());  // :353
    }
  }
  const int rc = amdarrow_grouper_create(ctx->memory_pool()->ctx(), static_cast<int>(types.size()), types.data(), &g->impl_);
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return g;
}
namespace {
// Array::Equals for two (small) dictionaries: same type and length, same validity, same values where valid
Result<bool> SameDictionary(const ArrayData& a, const ArrayData& b, ExecContext* ctx) {
  if (a.type->id() != b.type->id() || a.length != b.length) return false;
  if (a.buffers == b.buffers && a.offset == b.offset) return true;
  const auto& pool = ctx->memory_pool();
  if (is_binary_like(a.type->id())) {
    AMD_ASSIGN_OR_RAISE(auto x, BinaryToHost(a, pool));
    AMD_ASSIGN_OR_RAISE(auto y, BinaryToHost(b, pool));
    return x == y;
  }
  if (a.type->bit_width() <= 0) return Status::NotImplemented("comparing dictionaries of type " + a.type->ToString());
  const int64_t nb = ValueBytes(*a.type, a.length);
  std::vector<uint8_t> va(nb + 8), vb(nb + 8), ba((a.length + 7) / 8 + 8), bb((a.length + 7) / 8 + 8);
  AMD_RETURN_NOT_OK(CopyValuesToHost(a, pool, va.data(), ba.data()));
  AMD_RETURN_NOT_OK(CopyValuesToHost(b, pool, vb.data(), bb.data()));
  const int w = a.type->bit_width();
  for (int64_t i = 0; i < a.length; ++i) {
    const bool xa = (ba[i >> 3] >> (i & 7)) & 1, xb = (bb[i >> 3] >> (i & 7)) & 1;
    if (xa != xb) return false;
    if (!xa) continue;
    if (w == 1) { if (((va[i >> 3] >> (i & 7)) & 1) != ((vb[i >> 3] >> (i & 7)) & 1)) return false; }
    else if (memcmp(&va[i * (w / 8)], &vb[i * (w / 8)], w / 8) != 0) return false;
  }
  return true;
}
}  // namespace
Result<std::vector<std::shared_ptr<ArrayData>>> Grouper::FixedColumns(const ExecBatch& batch, bool insert) {
  std::vector<std::shared_ptr<ArrayData>> cols;
  for (size_t k = 0; k < batch.values.size(); ++k) {
    if (!batch.values[k].is_array()) return Status::NotImplemented("Grouper: array keys only");
    const auto& a = batch.values[k].array();
    if (k < var_.size() && var_[k]) {
      if (!insert) return Status::NotImplemented("Grouper::Lookup over variable-width keys");
      amdarrow_var_array v = a->var_abi();
      AMD_ASSIGN_OR_RAISE(auto ids, AllocateBuffer(4 * std::max<int64_t>(a->length, 1), ctx_->memory_pool()));
      const int rc = amdarrow_binary_grouper_consume(var_[k], &v, reinterpret_cast<uint32_t*>(ids->mutable_data()));
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
      cols.push_back(ArrayData::Make(uint32(), a->length, {nullptr, ids}, 0));
    } else if (a->type->id() == Type::DICTIONARY) {
      if (!dictionaries_[k]) dictionaries_[k] = a->dictionary;
      else if (a->dictionary) {
        AMD_ASSIGN_OR_RAISE(const bool same, SameDictionary(*dictionaries_[k], *a->dictionary, ctx_));
        if (!same) return Status::NotImplemented("Unifying differing dictionaries");  // :207-211
      }
      cols.push_back(DictIndices(*a));
    } else {
      cols.push_back(a);
    }
  }
  return cols;
}
Result<Datum> Grouper::Consume(const ExecBatch& batch) {
  AMD_ASSIGN_OR_RAISE(auto cols, FixedColumns(batch, true));
  std::vector<amdarrow_array> keys;
  for (const auto& c : cols) keys.push_back(c->abi());
  AMD_ASSIGN_OR_RAISE(auto buf, AllocateBuffer(4 * batch.length, ctx_->memory_pool()));
  const int rc = amdarrow_grouper_consume(impl_, keys.data(), reinterpret_cast<uint32_t*>(buf->mutable_data()));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return Datum(ArrayData::Make(uint32(), batch.length, {nullptr, buf}, 0));
}
Result<Datum> Grouper::Lookup(const ExecBatch& batch) {
  AMD_ASSIGN_OR_RAISE(auto cols, FixedColumns(batch, false));
  std::vector<amdarrow_array> keys;
  for (const auto& c : cols) keys.push_back(c->abi());
  AMD_ASSIGN_OR_RAISE(auto buf, AllocateBuffer(4 * batch.length, ctx_->memory_pool()));
  const int rc = amdarrow_grouper_lookup(impl_, keys.data(), reinterpret_cast<uint32_t*>(buf->mutable_data()));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  return Datum(ArrayData::Make(uint32(), batch.length, {nullptr, buf}, 0));
}
Result<std::shared_ptr<ArrayData>> Grouper::MakeGroupings(const ArrayData& ids, uint32_t num_groups, ExecContext* ctx) {
  if (ids.type->id() != Type::UINT32) return Status::Invalid("MakeGroupings: ids must be uint32");
  if (ids.null_count != 0 && ids.buffers[0]) {
    AMD_ASSIGN_OR_RAISE(Datum valid_count, CallFunction("count", {Datum(std::make_shared<ArrayData>(ids))}, ctx));
    int64_t valid = 0;
    memcpy(&valid, &valid_count.scalar()->bits, 8);
    if (valid != ids.length) return Status::Invalid("MakeGroupings with null ids");  // :982-984
  }
  const auto& pool = ctx->memory_pool();
  AMD_ASSIGN_OR_RAISE(auto offsets, AllocateBuffer(4 * (static_cast<int64_t>(num_groups) + 1), pool));
  AMD_ASSIGN_OR_RAISE(auto indices, AllocateBuffer(4 * std::max<int64_t>(ids.length, 1), pool));
  const int rc = amdarrow_make_groupings(pool->ctx(), reinterpret_cast<const uint32_t*>(ids.buffers[1]->data()) + ids.offset, ids.length,
                                         num_groups, reinterpret_cast<int32_t*>(offsets->mutable_data()),
                                         reinterpret_cast<int32_t*>(indices->mutable_data()));
  if (rc != AMDARROW_OK) return Status::FromAbi(rc);
  auto out = ArrayData::Make(list_(), num_groups, {nullptr, offsets}, 0);
  out->child_data = {ArrayData::Make(int32(), ids.length, {nullptr, indices}, 0)};
  return out;
}
Result<std::shared_ptr<ArrayData>> Grouper::ApplyGroupings(const ArrayData& groupings, const ArrayData& array, ExecContext* ctx) {
  if (groupings.type->id() != Type::LIST || groupings.child_data.size() != 1) return Status::Invalid("ApplyGroupings: groupings must be list<int32>");
  AMD_ASSIGN_OR_RAISE(Datum sorted, Take(Datum(std::make_shared<ArrayData>(array)), Datum(groupings.child_data[0]),
                                         TakeOptions::NoBoundsCheck(), ctx));
  auto out = std::make_shared<ArrayData>(groupings);
  out->child_data = {sorted.array()};
  return out;
}
uint32_t Grouper::num_groups() const {
  uint32_t n = 0;
  amdarrow_grouper_num_groups(impl_, &n);
  return n;
}
Result<ExecBatch> Grouper::GetUniques() {
  const uint32_t ng = num_groups();
  ExecBatch out({}, ng);
  for (size_t k = 0; k < descrs_.size(); ++k) {
    auto t = descrs_[k].type;
    const bool is_var = var_[k] != nullptr, is_dict = t->id() == Type::DICTIONARY;
    if (is_var) t = uint32();            // the unique tuples name the column's keys by the column grouper's ids
    else if (is_dict) t = IndexTypeOf(*t);
    AMD_ASSIGN_OR_RAISE(auto vbuf, AllocateBuffer(ValueBytes(*t, ng), ctx_->memory_pool()));
    AMD_ASSIGN_OR_RAISE(auto nbuf, AllocateBitmap(ng, ctx_->memory_pool()));
    amdarrow_out o;
    const int rc = amdarrow_grouper_uniques(impl_, static_cast<int>(k), vbuf->mutable_data(), nbuf->mutable_data(), &o);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    auto col = ArrayData::Make(t, ng, {o.has_validity ? nbuf : nullptr, vbuf}, o.null_count);
    if (is_var) {  // VarLengthKeyEncoder::Decode :270-307: the keys themselves, a validity bitmap only when one is null
      const auto& pool = ctx_->memory_pool();
      const bool large = is_large_binary_like(descrs_[k].type->id());
      int64_t rows = 0, nbytes = 0;
      int rc2 = amdarrow_binary_grouper_uniques_size(var_[k], &rows, &nbytes);
      if (rc2 != AMDARROW_OK) return Status::FromAbi(rc2);
      AMD_ASSIGN_OR_RAISE(auto obuf, AllocateBuffer((large ? 8 : 4) * (rows + 1), pool));
      AMD_ASSIGN_OR_RAISE(auto dbuf, AllocateBuffer(nbytes, pool));
      AMD_ASSIGN_OR_RAISE(auto vb, AllocateBitmap(rows, pool));
      amdarrow_out so;
      rc2 = amdarrow_binary_grouper_uniques(var_[k], obuf->mutable_data(), dbuf->mutable_data(), vb->mutable_data(), &so);
      if (rc2 != AMDARROW_OK) return Status::FromAbi(rc2);
      auto store = ArrayData::Make(descrs_[k].type, so.length, {so.has_validity ? vb : nullptr, obuf}, so.null_count);
      store->buffers.push_back(dbuf);
      if (ng > 0) {
        KernelContext kctx(ctx_);
        AMD_ASSIGN_OR_RAISE(store, TakeAny(&kctx, *store, *col, false));
        if (store->null_count == 0) store->buffers[0] = nullptr;
      }
      col = store;
    } else if (is_dict) {  // DictionaryKeyEncoder::Decode :217-231
      col->type = descrs_[k].type;
      col->dictionary = dictionaries_[k];
    }
    out.values.push_back(Datum(col));
  }
  return out;
}

// ---- unique / value_counts / dictionary_encode (vector_hash.cc:55-225, 617-760) ------------------------
// The reference's three hash actions over one memo table; here over one Grouper (the same
// first-appearance numbering, a null being one more entry), fed every chunk of the input in order.
// They are registered as meta functions: the finalize hook that lets the reference's vector kernels
// share state across chunks (kernel.h:611-631) is not part of this mirror's VectorKernel.
namespace {
struct Hashed {
  std::vector<std::shared_ptr<ArrayData>> ids;  // per chunk
  std::shared_ptr<ArrayData> uniques;
  uint32_t num_groups = 0;
};
Result<Hashed> HashChunks(const Datum& input, ExecContext* ctx, std::vector<std::shared_ptr<ArrayData>>* chunks) {
  if (input.is_array()) *chunks = {input.array()};
  else if (input.kind() == Datum::CHUNKED_ARRAY) *chunks = input.chunked_array()->chunks();
  else return Status::NotImplemented("hash kernels take arrays and chunked arrays");
  auto type = input.type();
  if (!type || type->bit_width() < 0 || type->id() == Type::HALF_FLOAT) return Status::NotImplemented("hash kernels: fixed-width types only");
  AMD_ASSIGN_OR_RAISE(auto grouper, Grouper::Make({ValueDescr::Array(type)}, ctx));
  Hashed h;
  for (const auto& c : *chunks) {
    AMD_ASSIGN_OR_RAISE(auto batch, ExecBatch::Make({Datum(c)}));
    AMD_ASSIGN_OR_RAISE(Datum ids, grouper->Consume(batch));
    h.ids.push_back(ids.array());
  }
  AMD_ASSIGN_OR_RAISE(auto uniques, grouper->GetUniques());
  h.uniques = uniques.values[0].array();
  h.num_groups = grouper->num_groups();
  return h;
}
class UniqueFunction : public MetaFunction {
 public:
  UniqueFunction() : MetaFunction("unique", Arity::Unary()) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions*, ExecContext* ctx) const override {
    std::vector<std::shared_ptr<ArrayData>> chunks;
    AMD_ASSIGN_OR_RAISE(auto h, HashChunks(args[0], ctx, &chunks));
    return Datum(h.uniques);  // UniqueFinalize :617-623
  }
};
class ValueCountsFunction : public MetaFunction {
 public:
  ValueCountsFunction() : MetaFunction("value_counts", Arity::Unary()) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions*, ExecContext* ctx) const override {
    std::vector<std::shared_ptr<ArrayData>> chunks;
    AMD_ASSIGN_OR_RAISE(auto h, HashChunks(args[0], ctx, &chunks));
    // ValueCountsAction :91-160: rows per memo index, nulls counted -- hash_count over the ids themselves
    AMD_ASSIGN_OR_RAISE(auto counts, AllocateBuffer(8 * std::max<int64_t>(h.num_groups, 1), ctx->memory_pool()));
    AMD_ASSIGN_OR_RAISE(auto scratch_sums, AllocateBuffer(8 * std::max<int64_t>(h.num_groups, 1), ctx->memory_pool()));
    amdarrow_ctx* abi = ctx->memory_pool()->ctx();
    int rc = amdarrow_memset(abi, counts->mutable_data(), 0, 8 * std::max<int64_t>(h.num_groups, 1));
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    for (const auto& ids : h.ids) {
      amdarrow_array v = ids->abi();
      rc = amdarrow_hash_accumulate(abi, AMDARROW_HASH_COUNT, &v, reinterpret_cast<const uint32_t*>(ids->buffers[1]->data()) + ids->offset,
                                    AMDARROW_COUNT_NON_NULL, scratch_sums->mutable_data(), reinterpret_cast<int64_t*>(counts->mutable_data()));
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    rc = amdarrow_ctx_synchronize(abi);
    if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    auto out = ArrayData::Make(struct_(), h.num_groups, {nullptr, nullptr}, 0);  // BoxValueCounts :634-640
    out->child_names = {"values", "counts"};
    out->child_data = {h.uniques, ArrayData::Make(int64(), h.num_groups, {nullptr, counts}, 0)};
    return Datum(out);
  }
};
class DictionaryEncodeFunction : public MetaFunction {
 public:
  DictionaryEncodeFunction() : MetaFunction("dictionary_encode", Arity::Unary(), {}, &kDefaults) {}
  Result<Datum> ExecuteImpl(const std::vector<Datum>& args, const FunctionOptions* options, ExecContext* ctx) const override {
    const auto* opts = static_cast<const DictionaryEncodeOptions*>(options);
    const bool mask = !opts || opts->null_encoding_behavior == DictionaryEncodeOptions::MASK;
    std::vector<std::shared_ptr<ArrayData>> chunks;
    AMD_ASSIGN_OR_RAISE(auto h, HashChunks(args[0], ctx, &chunks));
    amdarrow_ctx* abi = ctx->memory_pool()->ctx();
    std::shared_ptr<ArrayData> dict = h.uniques;
    if (mask && dict->buffers[0]) {  // the dictionary holds no null: Filter(uniques, is_valid(uniques))
      AMD_ASSIGN_OR_RAISE(Datum valid, CallFunction("is_valid", {Datum(dict)}, nullptr, ctx));
      FilterOptions drop;
      AMD_ASSIGN_OR_RAISE(Datum kept, CallFunction("array_filter", {Datum(dict), valid}, &drop, ctx));
      dict = ArrayData::Make(dict->type, kept.array()->length, {nullptr, kept.array()->buffers[1]}, 0);
    }
    std::vector<std::shared_ptr<ArrayData>> out_chunks;
    for (const auto& ids : h.ids) {
      std::shared_ptr<ArrayData> indices;
      if (!mask) {  // ENCODE: the ids are the indices (uint32 < 2^31 reads as int32)
        indices = ArrayData::Make(dictionary_(), ids->length, {nullptr, ids->buffers[1]}, 0, ids->offset);
      } else {
        AMD_ASSIGN_OR_RAISE(auto ibuf, AllocateBuffer(4 * ids->length, ctx->memory_pool()));
        AMD_ASSIGN_OR_RAISE(auto nbuf, AllocateBitmap(ids->length, ctx->memory_pool()));
        amdarrow_out o;
        const int rc = amdarrow_dictionary_indices(abi, reinterpret_cast<const uint32_t*>(ids->buffers[1]->data()) + ids->offset, ids->length,
                                                   h.uniques->buffers[0] ? h.uniques->buffers[0]->data() : nullptr, h.num_groups,
                                                   reinterpret_cast<int32_t*>(ibuf->mutable_data()), nbuf->mutable_data(), &o);
        if (rc != AMDARROW_OK) return Status::FromAbi(rc);
        indices = ArrayData::Make(dictionary_(), ids->length, {o.has_validity ? nbuf : nullptr, ibuf}, o.null_count);
      }
      indices->dictionary = dict;  // DictEncodeFinalize :625-636: every chunk shares the final dictionary
      out_chunks.push_back(std::move(indices));
    }
    if (args[0].is_array()) return Datum(out_chunks[0]);
    return Datum(std::make_shared<ChunkedArray>(std::move(out_chunks), dictionary_()));
  }
  static DictionaryEncodeOptions kDefaults;
};
DictionaryEncodeOptions DictionaryEncodeFunction::kDefaults = DictionaryEncodeOptions::Defaults();
}  // namespace

void RegisterVectorHash(FunctionRegistry* registry) {
  (void)registry->AddFunction(std::make_shared<UniqueFunction>());
  (void)registry->AddFunction(std::make_shared<ValueCountsFunction>());
  (void)registry->AddFunction(std::make_shared<DictionaryEncodeFunction>());
}

// ---- GroupBy (hash_aggregate.cc:890-966) -----------------------------------------------------------------
namespace {
// The fused device path (amdarrow_filter_groupby without a mask: partitioned LDS tables, 1 pass per
// argument column) serves the common shape of a GroupBy call: ONE key column, every argument an
// array of one batch, aggregates among hash_sum / hash_count(non-null) / hash_mean.  It returns the
// rows the batch-wise path below returns -- same first-appearance group order, same output types
// and validity -- 13.8 ms -> ~1.2 ms for 100 M rows / 1 M groups.  Anything else (several keys,
// chunked inputs split into batches, other aggregates) takes the Grouper + per-kernel consume path.
struct FusedRun {
  const ArrayData* column;
  std::shared_ptr<Buffer> sums, counts, first_rows;
  std::vector<std::shared_ptr<Buffer>> keys, key_validity;  // one per key column
  std::vector<int64_t> key_nulls;
};
int HashFuncOf(const std::string& name) {
  return name == "hash_sum" ? AMDARROW_HASH_SUM : name == "hash_count" ? AMDARROW_HASH_COUNT : name == "hash_mean" ? AMDARROW_HASH_MEAN : -1;
}
bool IsFusedAggregate(const Aggregate& a) {
  const int func = HashFuncOf(a.function);
  if (func < 0) return false;
  if (func == AMDARROW_HASH_COUNT) {
    const auto* co = static_cast<const CountOptions*>(a.options);
    if (co && co->count_mode != CountOptions::COUNT_NON_NULL) return false;
  }
  return true;
}
bool CanFuseGroupBy(const std::vector<Datum>& arguments, const std::vector<Datum>& keys, const std::vector<Aggregate>& aggregates,
                    ExecContext* ctx) {
  if (keys.empty() || keys.size() > 4 || !keys[0].is_array() || arguments.empty()) return false;
  const ArrayData& key = *keys[0].array();
  if (key.length == 0 || key.length > ctx->exec_chunksize() || key.type->bit_width() < 0) return false;
  for (const auto& k : keys) {  // several keys: packed into one key word on the device (integer / boolean columns)
    if (!k.is_array() || k.array()->length != key.length || k.array()->type->bit_width() < 0) return false;
    if (keys.size() > 1 && is_floating(k.array()->type->id())) return false;
  }
  for (size_t i = 0; i < arguments.size(); ++i) {
    if (!arguments[i].is_array() || arguments[i].array()->length != key.length) return false;
    if (arguments[i].array()->type->bit_width() < 0) return false;
    // hash_sum / hash_count(non-null) / hash_mean ride the fused pass; any other hash aggregate (hash_min_max,
    // hash_count of nulls) is served afterwards through per-row group ids (FusedGroupBy) -- it only has to exist
    if (!IsFusedAggregate(aggregates[i])) {
      auto f = ctx->func_registry()->GetFunction(aggregates[i].function);
      if (!f.ok() || (*f)->kind() != Function::HASH_AGGREGATE) return false;
      // one key column the Grouper groups by batch (at most 32 bits, or 64 bits without a validity buffer; >= 2^18 rows):
      // its Consume hands out the row ids at the cost of a look-up, so the batch-wise path -- Grouper, then every
      // kernel's own consume -- beats the fused pass plus a second, hashed look-up of every row (4.0 against 4.9 ms
      // for {hash_sum, hash_min_max} over 50 M rows)
      const int bw = key.type->bit_width();
      if (keys.size() == 1 && key.length >= (int64_t(1) << 18) && (bw <= 32 || (bw == 64 && key.buffers[0] == nullptr))) return false;
    }
  }
  return true;
}
Result<GroupByResult> FusedGroupBy(const std::vector<Datum>& arguments, const std::vector<Datum>& keys,
                                   const std::vector<Aggregate>& aggregates, ExecContext* ctx) {
  KernelContext kctx(ctx);
  const ArrayData& key = *keys[0].array();
  const int64_t n = key.length;
  const int nkeys = static_cast<int>(keys.size());
  std::vector<amdarrow_array> k;
  for (const auto& kd : keys) k.push_back(kd.array()->abi());
  std::vector<FusedRun> runs;
  std::vector<int> run_of(arguments.size());
  int64_t num_groups = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {
    // group capacity: most inputs have far fewer groups than rows; retry with n if they do not
    const int64_t cap = attempt == 0 ? std::min<int64_t>(n, int64_t(1) << 22) : n;
    runs.clear();
    bool too_small = false;
    bool any_fused = false;
    for (const auto& a : aggregates) any_fused = any_fused || IsFusedAggregate(a);
    for (size_t i = 0; i < arguments.size() && !too_small; ++i) {
      // a column needs its own fused pass only for an aggregate that rides it; without any, the first column's pass does the grouping
      if (!IsFusedAggregate(aggregates[i]) && (any_fused || i > 0)) { run_of[i] = -1; continue; }
      const ArrayData* col = arguments[i].array().get();
      int found = -1;
      for (size_t r = 0; r < runs.size(); ++r) if (runs[r].column == col) found = static_cast<int>(r);
      if (found >= 0) { run_of[i] = found; continue; }
      FusedRun run;
      run.column = col;
      AMD_ASSIGN_OR_RAISE(run.sums, kctx.Allocate(8 * cap));
      AMD_ASSIGN_OR_RAISE(run.counts, kctx.Allocate(8 * cap));
      std::vector<void*> kp;
      std::vector<uint8_t*> vp;
      for (int c = 0; c < nkeys; ++c) {
        AMD_ASSIGN_OR_RAISE(auto kb, kctx.Allocate(ValueBytes(*keys[c].array()->type, cap) + 8));
        AMD_ASSIGN_OR_RAISE(auto vb, kctx.AllocateBitmap(cap));
        kp.push_back(kb->mutable_data());
        vp.push_back(vb->mutable_data());
        run.keys.push_back(kb);
        run.key_validity.push_back(vb);
      }
      run.key_nulls.assign(nkeys, -1);
      AMD_ASSIGN_OR_RAISE(run.first_rows, kctx.Allocate(4 * cap));
      amdarrow_array v = col->abi();
      int64_t g = 0;
      const int rc = nkeys == 1
          ? amdarrow_filter_groupby(kctx.abi(), &v, &k[0], nullptr, cap, run.sums->mutable_data(),
                                    reinterpret_cast<int64_t*>(run.counts->mutable_data()), kp[0], vp[0],
                                    reinterpret_cast<uint32_t*>(run.first_rows->mutable_data()), &g)
          : amdarrow_filter_groupby_multi(kctx.abi(), &v, k.data(), nkeys, nullptr, cap, run.sums->mutable_data(),
                                          reinterpret_cast<int64_t*>(run.counts->mutable_data()), kp.data(), vp.data(),
                                          run.key_nulls.data(), reinterpret_cast<uint32_t*>(run.first_rows->mutable_data()), &g);
      if (rc == AMDARROW_CAPACITY_ERROR && cap < n) { too_small = true; break; }
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
      num_groups = g;  // identical for every column: the grouping only looks at the keys
      run_of[i] = static_cast<int>(runs.size());
      runs.push_back(std::move(run));
    }
    if (!too_small) break;
  }
  GroupByResult result;
  result.length = num_groups;
  // the key columns: a bitmap exists only when a null key exists (FixedWidthKeyEncoder::Decode, hash_aggregate.cc:64-92)
  std::vector<std::shared_ptr<ArrayData>> key_columns;
  {
    const FusedRun& r0 = runs[0];
    for (int c = 0; c < nkeys; ++c) {
      int64_t nulls = r0.key_nulls[c];
      if (nulls < 0) {  // the single-key entry point does not report it
        int64_t valid_keys = num_groups;
        if (num_groups > 0) {
          const int rc = amdarrow_count_set_bits(kctx.abi(), r0.key_validity[c]->data(), 0, num_groups, &valid_keys);
          if (rc != AMDARROW_OK) return Status::FromAbi(rc);
        }
        nulls = num_groups - valid_keys;
      }
      key_columns.push_back(ArrayData::Make(keys[c].array()->type, num_groups, {nulls > 0 ? r0.key_validity[c] : nullptr, r0.keys[c]}, nulls));
    }
  }
  Datum row_ids;  // per-row group ids, built on first need: a Grouper fed the unique keys (ids 0..G-1 in this order), then every row
  for (size_t i = 0; i < arguments.size(); ++i) {
    if (!IsFusedAggregate(aggregates[i])) {
      if (row_ids.kind() == Datum::NONE) {
        std::vector<ValueDescr> descrs;
        std::vector<Datum> uniq, all;
        for (int c = 0; c < nkeys; ++c) {
          descrs.push_back(ValueDescr::Array(keys[c].array()->type));
          uniq.push_back(Datum(key_columns[c]));
          all.push_back(keys[c]);
        }
        AMD_ASSIGN_OR_RAISE(auto grouper, Grouper::Make(descrs, ctx));
        if (num_groups > 0) {
          AMD_ASSIGN_OR_RAISE(auto ub, ExecBatch::Make(uniq));
          AMD_ASSIGN_OR_RAISE(Datum ignored, grouper->Consume(ub));
          (void)ignored;
        }
        AMD_ASSIGN_OR_RAISE(auto ab, ExecBatch::Make(all));
        if (num_groups > 0) { AMD_ASSIGN_OR_RAISE(row_ids, grouper->Lookup(ab)); }
        else { AMD_ASSIGN_OR_RAISE(row_ids, grouper->Consume(ab)); }
        if (static_cast<int64_t>(grouper->num_groups()) != num_groups) return Status::Invalid("GroupBy: group ids disagree with the fused grouping");
      }
      AMD_ASSIGN_OR_RAISE(auto function, ctx->func_registry()->GetFunction(aggregates[i].function));
      auto* hf = static_cast<const HashAggregateFunction*>(function.get());
      AMD_ASSIGN_OR_RAISE(const HashAggregateKernel* kernel,
                          hf->DispatchExact({ValueDescr::Array(arguments[i].type()), ValueDescr::Array(uint32()), ValueDescr::Scalar(uint32())}));
      const FunctionOptions* o = aggregates[i].options ? aggregates[i].options : function->default_options();
      AMD_ASSIGN_OR_RAISE(auto state, kernel->init(&kctx, arguments[i].type(), o));
      ExecBatch batch({arguments[i], row_ids, Datum(Scalar::Make(uint32(), static_cast<uint32_t>(num_groups)))}, n);
      AMD_RETURN_NOT_OK(kernel->consume(&kctx, state.get(), batch));
      Datum out;
      AMD_RETURN_NOT_OK(kernel->finalize(&kctx, state.get(), &out));
      result.field_names.push_back(aggregates[i].function);
      result.columns.push_back(out.array());
      continue;
    }
    const FusedRun& run = runs[run_of[i]];
    const int func = HashFuncOf(aggregates[i].function);
    const Type::type in = arguments[i].array()->type->id();
    std::shared_ptr<DataType> ot = func == AMDARROW_HASH_COUNT ? int64()
                                 : (func == AMDARROW_HASH_MEAN || is_floating(in)) ? float64()
                                 : is_signed_integer(in) ? int64() : uint64();
    // the finalize kernel works in place: give every aggregate its own copy of the column's sums
    AMD_ASSIGN_OR_RAISE(auto vbuf, kctx.Allocate(8 * std::max<int64_t>(num_groups, 1)));
    AMD_ASSIGN_OR_RAISE(auto nbuf, kctx.AllocateBitmap(std::max<int64_t>(num_groups, 1)));
    amdarrow_out o{};
    if (num_groups > 0) {
      int rc = amdarrow_memcpy_d2d(kctx.abi(), vbuf->mutable_data(), run.sums->data(), 8 * num_groups);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
      rc = amdarrow_hash_finalize(kctx.abi(), func, static_cast<int>(in), num_groups, vbuf->mutable_data(),
                                  reinterpret_cast<const int64_t*>(run.counts->data()), nbuf->mutable_data(), &o);
      if (rc != AMDARROW_OK) return Status::FromAbi(rc);
    }
    result.field_names.push_back(aggregates[i].function);
    result.columns.push_back(ArrayData::Make(ot, num_groups, {o.has_validity ? nbuf : nullptr, vbuf}, o.null_count));
  }
  for (int c = 0; c < nkeys; ++c) {
    result.field_names.push_back("key_" + std::to_string(c));
    result.columns.push_back(key_columns[c]);
  }
  return result;
}
}  // namespace

Result<GroupByResult> GroupBy(const std::vector<Datum>& arguments, const std::vector<Datum>& keys,
                              const std::vector<Aggregate>& aggregates, ExecContext* ctx) {
  if (aggregates.size() != arguments.size())  // GetKernels :811-815
    return Status::Invalid(std::to_string(aggregates.size()) + " aggregate functions were specified but " +
                           std::to_string(arguments.size()) + " arguments were provided.");
  if (CanFuseGroupBy(arguments, keys, aggregates, ctx) && !getenv("AMDARROW_GROUPBY_BATCHWISE")) {
    auto fused = FusedGroupBy(arguments, keys, aggregates, ctx);
    // several keys: the packed path declines shapes it does not take (too wide together, too few rows, too many groups)
    const bool declined = keys.size() > 1 && !fused.ok() && (fused.status().IsNotImplemented() || fused.status().IsCapacityError());
    if (!declined) return fused;
  }
  std::vector<const HashAggregateKernel*> kernels;
  std::vector<std::unique_ptr<KernelState>> states;
  std::vector<const FunctionOptions*> opts;
  for (size_t i = 0; i < aggregates.size(); ++i) {
    AMD_ASSIGN_OR_RAISE(auto function, ctx->func_registry()->GetFunction(aggregates[i].function));
    if (function->kind() != Function::HASH_AGGREGATE) return Status::Invalid(aggregates[i].function + " is not a hash aggregate function");
    auto* hf = static_cast<const HashAggregateFunction*>(function.get());
    AMD_ASSIGN_OR_RAISE(const HashAggregateKernel* kernel,
                        hf->DispatchExact({ValueDescr::Array(arguments[i].type()), ValueDescr::Array(uint32()), ValueDescr::Scalar(uint32())}));
    const FunctionOptions* o = aggregates[i].options ? aggregates[i].options : function->default_options();  // InitKernels :840-846
    KernelContext kctx(ctx);
    AMD_ASSIGN_OR_RAISE(auto state, kernel->init(&kctx, arguments[i].type(), o));
    kernels.push_back(kernel);
    states.push_back(std::move(state));
    opts.push_back(o);
  }
  std::vector<ValueDescr> key_descrs;
  for (const auto& k : keys) key_descrs.push_back(ValueDescr::Array(k.type()));
  AMD_ASSIGN_OR_RAISE(auto grouper, Grouper::Make(key_descrs, ctx));
  AMD_ASSIGN_OR_RAISE(auto arg_it, detail::ExecBatchIterator::Make(arguments, ctx->exec_chunksize()));
  AMD_ASSIGN_OR_RAISE(auto key_it, detail::ExecBatchIterator::Make(keys, ctx->exec_chunksize()));
  ExecBatch key_batch, arg_batch;
  while (arg_it->Next(&arg_batch) && key_it->Next(&key_batch)) {  // :928-943
    if (key_batch.length == 0) continue;
    if (arg_batch.length != key_batch.length)
      return Status::NotImplemented("GroupBy: arguments and keys must share their chunk layout");
    AMD_ASSIGN_OR_RAISE(Datum id_batch, grouper->Consume(key_batch));
    for (size_t i = 0; i < kernels.size(); ++i) {
      KernelContext kctx(ctx);
      ExecBatch batch({arg_batch[static_cast<int>(i)], id_batch, Datum(Scalar::Make(uint32(), grouper->num_groups()))}, key_batch.length);
      AMD_RETURN_NOT_OK(kernels[i]->consume(&kctx, states[i].get(), batch));
    }
  }
  GroupByResult result;
  result.length = grouper->num_groups();
  for (size_t i = 0; i < kernels.size(); ++i) {
    KernelContext kctx(ctx);
    Datum out;
    AMD_RETURN_NOT_OK(kernels[i]->finalize(&kctx, states[i].get(), &out));
    result.field_names.push_back(aggregates[i].function);  // field(aggregates[i].function, ...) :874
    result.columns.push_back(out.array());
  }
  AMD_ASSIGN_OR_RAISE(ExecBatch uniques, grouper->GetUniques());
  for (size_t k = 0; k < uniques.values.size(); ++k) {
    result.field_names.push_back("key_" + std::to_string(k));  // :917-919
    result.columns.push_back(uniques.values[k].array());
  }
  return result;
}

}  // namespace internal
}  // namespace compute
}  // namespace amdarrow

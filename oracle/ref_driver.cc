// ref_driver.cc -- OUR C-ABI over the REAL reference (arrow::compute built from
// /root/reference by oracle/ref_build.mk).  TEST INFRASTRUCTURE ONLY.
//
// Exports ref_* with the signatures of orc_* in oracle.h, so that a test can run the
// same inputs through the restatement and through the reference and compare bytes.
// Everything goes through the reference's public entry points: CallFunction with an
// ExecContext bound to a registry built by FunctionRegistry::Make() + the hot-path
// Register* hooks (the reference's own extension mechanism, compute/exec_test.cc:651-717),
// internal::Grouper and internal::GroupBy (compute/api_aggregate.h:313-403).
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "arrow/array.h"
#include "arrow/array/concatenate.h"
#include "arrow/array/data.h"
#include "arrow/buffer.h"
#include "arrow/chunked_array.h"
#include "arrow/compute/api_aggregate.h"
#include "arrow/compute/api_scalar.h"
#include "arrow/compute/api_vector.h"
#include "arrow/compute/cast.h"
#include "arrow/compute/cast_internal.h"
#include "arrow/compute/exec.h"
#include "arrow/compute/exec_internal.h"
#include "arrow/compute/function.h"
#include "arrow/compute/kernel.h"
#include "arrow/compute/registry.h"
#include "arrow/compute/registry_internal.h"
#include "arrow/datum.h"
#include "arrow/record_batch.h"
#include "arrow/scalar.h"
#include "arrow/table.h"
#include "arrow/type.h"
#include "arrow/util/bit_util.h"
#include "arrow/util/bitmap_ops.h"

#include "oracle.h"

using namespace arrow;           // NOLINT
using namespace arrow::compute;  // NOLINT
using arrow::internal::checked_cast;

namespace {

thread_local std::string g_err;
int Fail(const Status& st) {
  g_err = st.message();
  return static_cast<int>(st.code());
}

FunctionRegistry* HotPathRegistry() {
  static std::unique_ptr<FunctionRegistry> reg = [] {
    auto r = FunctionRegistry::Make();
    compute::internal::RegisterScalarArithmetic(r.get());
    compute::internal::RegisterScalarComparison(r.get());
    compute::internal::RegisterScalarBoolean(r.get());
    compute::internal::RegisterScalarValidity(r.get());
    compute::internal::RegisterScalarFillNull(r.get());
    compute::internal::RegisterVectorSelection(r.get());
    compute::internal::RegisterVectorSort(r.get());
    compute::internal::RegisterVectorHash(r.get());
    compute::internal::RegisterScalarAggregateBasic(r.get());
    compute::internal::RegisterHashAggregateBasic(r.get());
    return r;
  }();
  return reg.get();
}
ExecContext* Ctx() {
  static ExecContext ctx(default_memory_pool(), HotPathRegistry());
  return &ctx;
}

std::shared_ptr<DataType> TypeOf(int id) {
  switch (id) {
    case ORC_BOOL: return boolean();
    case ORC_UINT8: return uint8();
    case ORC_INT8: return int8();
    case ORC_UINT16: return uint16();
    case ORC_INT16: return int16();
    case ORC_UINT32: return uint32();
    case ORC_INT32: return int32();
    case ORC_UINT64: return uint64();
    case ORC_INT64: return int64();
    case ORC_FLOAT: return float32();
    case ORC_DOUBLE: return float64();
  }
  return nullptr;
}
int BitWidth(int id) {
  auto t = TypeOf(id);
  return t ? checked_cast<const FixedWidthType&>(*t).bit_width() : -1;
}
int64_t BufBytes(int bw, int64_t n) { return bw == 1 ? BitUtil::BytesForBits(n) : n * bw / 8; }

// Wrap caller memory (no copy, no ownership) as ArrayData.
std::shared_ptr<ArrayData> Wrap(const orc_array* a) {
  const int bw = BitWidth(a->type_id);
  const int64_t n = a->offset + a->length;
  std::shared_ptr<Buffer> validity;
  if (a->validity) validity = std::make_shared<Buffer>(a->validity, BitUtil::BytesForBits(n));
  auto values = std::make_shared<Buffer>(reinterpret_cast<const uint8_t*>(a->values), BufBytes(bw, n));
  return ArrayData::Make(TypeOf(a->type_id), a->length, {validity, values},
                         a->validity ? a->null_count : 0, a->offset);
}

std::shared_ptr<Scalar> WrapScalar(int type_id, const void* p) {
  switch (type_id) {
    case ORC_BOOL: return std::make_shared<BooleanScalar>(*reinterpret_cast<const uint8_t*>(p) != 0);
    case ORC_UINT8: return std::make_shared<UInt8Scalar>(*reinterpret_cast<const uint8_t*>(p));
    case ORC_INT8: return std::make_shared<Int8Scalar>(*reinterpret_cast<const int8_t*>(p));
    case ORC_UINT16: return std::make_shared<UInt16Scalar>(*reinterpret_cast<const uint16_t*>(p));
    case ORC_INT16: return std::make_shared<Int16Scalar>(*reinterpret_cast<const int16_t*>(p));
    case ORC_UINT32: return std::make_shared<UInt32Scalar>(*reinterpret_cast<const uint32_t*>(p));
    case ORC_INT32: return std::make_shared<Int32Scalar>(*reinterpret_cast<const int32_t*>(p));
    case ORC_UINT64: return std::make_shared<UInt64Scalar>(*reinterpret_cast<const uint64_t*>(p));
    case ORC_INT64: return std::make_shared<Int64Scalar>(*reinterpret_cast<const int64_t*>(p));
    case ORC_FLOAT: return std::make_shared<FloatScalar>(*reinterpret_cast<const float*>(p));
    case ORC_DOUBLE: return std::make_shared<DoubleScalar>(*reinterpret_cast<const double*>(p));
  }
  return nullptr;
}

// Copy a produced primitive array out, normalised to bit offset 0.
void Export(const ArrayData& d, void* out_values, uint8_t* out_validity, orc_out* out) {
  const int bw = checked_cast<const FixedWidthType&>(*d.type).bit_width();
  out->length = d.length;
  out->null_count = d.null_count.load();
  out->has_validity = d.buffers[0] != nullptr;
  if (out_validity) {
    std::memset(out_validity, 0, BitUtil::BytesForBits(d.length));
    if (d.buffers[0] && d.length > 0)
      arrow::internal::CopyBitmap(d.buffers[0]->data(), d.offset, d.length, out_validity, 0);
  }
  if (out_values && d.length > 0) {
    if (bw == 1) {
      std::memset(out_values, 0, BitUtil::BytesForBits(d.length));
      arrow::internal::CopyBitmap(d.buffers[1]->data(), d.offset, d.length,
                                  reinterpret_cast<uint8_t*>(out_values), 0);
    } else {
      std::memcpy(out_values, d.buffers[1]->data() + d.offset * bw / 8, d.length * bw / 8);
    }
  }
}

}  // namespace

extern "C" {

const char* ref_last_error(void) { return g_err.c_str(); }

int ref_num_functions(void) { return HotPathRegistry()->num_functions(); }

int64_t ref_filter_output_size(const orc_array* filter, int null_selection) {
  return compute::internal::GetFilterOutputSize(
      *Wrap(filter), static_cast<FilterOptions::NullSelectionBehavior>(null_selection));
}

int ref_filter(const orc_array* values, const orc_array* filter, int null_selection,
               void* out_values, uint8_t* out_validity, orc_out* out) {
  FilterOptions opts(static_cast<FilterOptions::NullSelectionBehavior>(null_selection));
  auto res = CallFunction("filter", {Datum(Wrap(values)), Datum(Wrap(filter))}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_values, out_validity, out);
  return 0;
}

int ref_take_indices(const orc_array* filter, int null_selection, void* out_indices,
                     uint8_t* out_validity, int* out_index_width, orc_out* out) {
  auto res = compute::internal::GetTakeIndices(
      *Wrap(filter), static_cast<FilterOptions::NullSelectionBehavior>(null_selection));
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = **res;
  *out_index_width = checked_cast<const FixedWidthType&>(*d.type).bit_width();
  Export(d, out_indices, out_validity, out);
  return 0;
}

int ref_take(const orc_array* values, const orc_array* indices, int boundscheck,
             void* out_values, uint8_t* out_validity, orc_out* out) {
  TakeOptions opts(boundscheck != 0);
  auto res = CallFunction("take", {Datum(Wrap(values)), Datum(Wrap(indices))}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_values, out_validity, out);
  return 0;
}

int ref_sort_indices(const orc_array* chunks, int nchunks, int order, uint64_t* out) {
  SortOrder so = order == ORC_DESCENDING ? SortOrder::Descending : SortOrder::Ascending;
  Datum input;
  if (nchunks == 1) {
    input = Datum(Wrap(&chunks[0]));
  } else {
    ArrayVector arrs;
    std::shared_ptr<DataType> type = nchunks ? TypeOf(chunks[0].type_id) : int64();
    for (int c = 0; c < nchunks; ++c) arrs.push_back(MakeArray(Wrap(&chunks[c])));
    input = Datum(std::make_shared<ChunkedArray>(arrs, type));
  }
  SortOptions opts({SortKey("", so)});
  auto res = CallFunction("sort_indices", {input}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  std::memcpy(out, d.buffers[1]->data() + d.offset * 8, d.length * 8);
  return 0;
}

// Multi-key sort_indices through the reference's meta function (vector_sort.cc:1632-1707):
// a RecordBatch when every key is one chunk and !force_table (RadixRecordBatchSorter up to 8 keys,
// MultipleKeyRecordBatchSorter above), else a Table (MultipleKeyTableSorter).
static int SortMulti(const orc_array* chunks, const int* nchunks, const int* orders, int nkeys, bool force_table,
                     uint64_t* out) {
  std::vector<std::shared_ptr<Field>> fields;
  std::vector<std::shared_ptr<ChunkedArray>> columns;
  std::vector<SortKey> keys;
  bool single = true;
  const orc_array* p = chunks;
  for (int k = 0; k < nkeys; ++k) {
    ArrayVector arrs;
    for (int c = 0; c < nchunks[k]; ++c) arrs.push_back(MakeArray(Wrap(&p[c])));
    auto type = TypeOf(p[0].type_id);
    const std::string name = "k" + std::to_string(k);
    fields.push_back(field(name, type));
    columns.push_back(std::make_shared<ChunkedArray>(arrs, type));
    keys.emplace_back(name, orders[k] == ORC_DESCENDING ? SortOrder::Descending : SortOrder::Ascending);
    single = single && nchunks[k] == 1;
    p += nchunks[k];
  }
  SortOptions opts(keys);
  Datum input;
  if (single && !force_table) {
    ArrayVector cols;
    for (auto& c : columns) cols.push_back(c->chunk(0));
    input = Datum(RecordBatch::Make(schema(fields), cols.empty() ? 0 : cols[0]->length(), cols));
  } else {
    input = Datum(Table::Make(schema(fields), columns));
  }
  auto res = CallFunction("sort_indices", {input}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  std::memcpy(out, d.buffers[1]->data() + d.offset * 8, d.length * 8);
  return 0;
}
int ref_sort_indices_multi(const orc_array* chunks, const int* nchunks, const int* orders, int nkeys, uint64_t* out) {
  return SortMulti(chunks, nchunks, orders, nkeys, false, out);
}
int ref_sort_indices_multi_table(const orc_array* chunks, const int* nchunks, const int* orders, int nkeys, uint64_t* out) {
  return SortMulti(chunks, nchunks, orders, nkeys, true, out);
}

int ref_concatenate(const orc_array* chunks, int nchunks, void* out_values, uint8_t* out_validity, orc_out* out) {
  ArrayVector arrs;
  for (int c = 0; c < nchunks; ++c) arrs.push_back(MakeArray(Wrap(&chunks[c])));
  auto res = Concatenate(arrs, default_memory_pool());
  if (!res.ok()) return Fail(res.status());
  Export(*(*res)->data(), out_values, out_validity, out);
  return 0;
}

struct ref_grouper {
  std::unique_ptr<compute::internal::Grouper> impl;
  int nkeys;
};

ref_grouper* ref_grouper_new(int nkeys, const int* key_type_ids) {
  std::vector<ValueDescr> descrs;
  for (int k = 0; k < nkeys; ++k) descrs.push_back(ValueDescr::Array(TypeOf(key_type_ids[k])));
  auto res = compute::internal::Grouper::Make(descrs, Ctx());
  if (!res.ok()) { Fail(res.status()); return nullptr; }
  auto* g = new ref_grouper;
  g->impl = std::move(*res);
  g->nkeys = nkeys;
  return g;
}
void ref_grouper_free(ref_grouper* g) { delete g; }
int ref_grouper_consume(ref_grouper* g, const orc_array* keys, uint32_t* out_ids) {
  std::vector<Datum> cols;
  for (int k = 0; k < g->nkeys; ++k) cols.emplace_back(Wrap(&keys[k]));
  auto batch = ExecBatch::Make(cols);
  if (!batch.ok()) return Fail(batch.status());
  auto res = g->impl->Consume(*batch);
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  std::memcpy(out_ids, d.buffers[1]->data() + d.offset * 4, d.length * 4);
  return 0;
}
uint32_t ref_grouper_num_groups(const ref_grouper* g) { return g->impl->num_groups(); }
int ref_grouper_uniques(const ref_grouper* g, int k, void* out_values, uint8_t* out_validity,
                        int64_t* out_null_count) {
  auto res = g->impl->GetUniques();
  if (!res.ok()) return Fail(res.status());
  orc_out o;
  Export(*res->values[k].array(), out_values, out_validity, &o);
  if (!o.has_validity) {  // the restatement always fills the bitmap; absent == all valid
    for (int64_t i = 0; i < o.length; ++i) BitUtil::SetBit(out_validity, i);
  }
  *out_null_count = o.null_count;
  return 0;
}

// Drive one HashAggregateKernel the way GroupBy does (hash_aggregate.cc:809-859, 928-960).
int ref_hash_aggregate(int func, const orc_array* values, const uint32_t* ids,
                       int64_t num_groups, int count_mode, void* out_values,
                       uint8_t* out_validity, orc_out* out) {
  const char* name = func == 0 ? "hash_sum" : func == 1 ? "hash_count" : nullptr;
  if (!name) { g_err = "hash_mean does not exist in the reference"; return ORC_KEY_ERROR; }
  auto fres = Ctx()->func_registry()->GetFunction(name);
  if (!fres.ok()) return Fail(fres.status());
  auto vals = Wrap(values);
  auto kres = (*fres)->DispatchExact(
      {ValueDescr::Array(vals->type), ValueDescr::Array(uint32()), ValueDescr::Scalar(uint32())});
  if (!kres.ok()) return Fail(kres.status());
  auto kernel = static_cast<const HashAggregateKernel*>(*kres);
  CountOptions copts(static_cast<CountOptions::Mode>(count_mode));
  const FunctionOptions* options = func == 1 ? static_cast<const FunctionOptions*>(&copts) : nullptr;
  KernelContext kctx(Ctx());
  auto sres = kernel->init(&kctx, KernelInitArgs{kernel, {vals->type, uint32(), uint32()}, options});
  if (!sres.ok()) return Fail(sres.status());
  std::unique_ptr<KernelState> state = std::move(*sres);
  kctx.SetState(state.get());
  auto idbuf = std::make_shared<Buffer>(reinterpret_cast<const uint8_t*>(ids), values->length * 4);
  auto id_arr = ArrayData::Make(uint32(), values->length, {nullptr, idbuf}, 0);
  auto batch = ExecBatch::Make({Datum(vals), Datum(id_arr), Datum(static_cast<uint32_t>(num_groups))});
  if (!batch.ok()) return Fail(batch.status());
  Status st = kernel->consume(&kctx, *batch);
  if (!st.ok()) return Fail(st);
  Datum result;
  st = kernel->finalize(&kctx, &result);
  if (!st.ok()) return Fail(st);
  Export(*result.array(), out_values, out_validity, out);
  if (!out->has_validity)
    for (int64_t i = 0; i < out->length; ++i) BitUtil::SetBit(out_validity, i);
  return 0;
}

// hash_min_max through the reference's HashAggregateKernel (hash_aggregate.cc:627-768): the struct's
// two children come back as out_mins / out_maxes, their shared bitmap as out_validity.
int ref_hash_min_max(const orc_array* values, const uint32_t* ids, int64_t num_groups, int null_handling,
                     void* out_mins, void* out_maxes, uint8_t* out_validity, orc_out* out) {
  auto fres = Ctx()->func_registry()->GetFunction("hash_min_max");
  if (!fres.ok()) return Fail(fres.status());
  auto vals = Wrap(values);
  auto kres = (*fres)->DispatchExact(
      {ValueDescr::Array(vals->type), ValueDescr::Array(uint32()), ValueDescr::Scalar(uint32())});
  if (!kres.ok()) return Fail(kres.status());
  auto kernel = static_cast<const HashAggregateKernel*>(*kres);
  MinMaxOptions opts(static_cast<MinMaxOptions::Mode>(null_handling));
  KernelContext kctx(Ctx());
  auto sres = kernel->init(&kctx, KernelInitArgs{kernel, {vals->type, uint32(), uint32()}, &opts});
  if (!sres.ok()) return Fail(sres.status());
  std::unique_ptr<KernelState> state = std::move(*sres);
  kctx.SetState(state.get());
  auto idbuf = std::make_shared<Buffer>(reinterpret_cast<const uint8_t*>(ids), values->length * 4);
  auto id_arr = ArrayData::Make(uint32(), values->length, {nullptr, idbuf}, 0);
  auto batch = ExecBatch::Make({Datum(vals), Datum(id_arr), Datum(static_cast<uint32_t>(num_groups))});
  if (!batch.ok()) return Fail(batch.status());
  Status st = kernel->consume(&kctx, *batch);
  if (!st.ok()) return Fail(st);
  Datum result;
  st = kernel->finalize(&kctx, &result);
  if (!st.ok()) return Fail(st);
  const ArrayData& s = *result.array();
  orc_out ignored;
  Export(*s.child_data[0], out_mins, out_validity, out);
  Export(*s.child_data[1], out_maxes, nullptr, &ignored);
  out->null_count = num_groups - arrow::internal::CountSetBits(out_validity, 0, num_groups);
  return 0;
}

// ---- unique / value_counts / dictionary_encode (vector_hash.cc) ----------------------
int ref_unique(const orc_array* a, void* out_values, uint8_t* out_validity, orc_out* out) {
  auto res = CallFunction("unique", {Datum(Wrap(a))}, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_values, out_validity, out);
  return 0;
}
int ref_value_counts(const orc_array* a, void* out_values, uint8_t* out_validity, int64_t* out_counts, orc_out* out) {
  auto res = CallFunction("value_counts", {Datum(Wrap(a))}, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  const ArrayData& st = *res->array();
  orc_out ignored;
  Export(*st.child_data[0], out_values, out_validity, out);
  Export(*st.child_data[1], out_counts, nullptr, &ignored);
  return 0;
}
int ref_dictionary_encode(const orc_array* a, int null_encoding, int32_t* out_indices, uint8_t* out_index_validity,
                          orc_out* om, void* out_dict, uint8_t* out_dict_validity, orc_out* dm) {
  DictionaryEncodeOptions opts(null_encoding == 0 ? DictionaryEncodeOptions::ENCODE : DictionaryEncodeOptions::MASK);
  auto res = CallFunction("dictionary_encode", {Datum(Wrap(a))}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  auto indices = d.Copy();
  indices->type = int32();
  indices->dictionary = nullptr;
  Export(*indices, out_indices, out_index_validity, om);
  if (!om->has_validity) std::memset(out_index_validity, 0xFF, BitUtil::BytesForBits(d.length));
  Export(*d.dictionary, out_dict, out_dict_validity, dm);
  return 0;
}

// ---- variable-width (binary) Take / Filter through the reference's "take" / "filter" ----
static std::shared_ptr<ArrayData> WrapBinary(const orc_binary* a) {
  const int64_t n = a->offset + a->length;
  std::shared_ptr<Buffer> validity;
  if (a->validity) validity = std::make_shared<Buffer>(a->validity, BitUtil::BytesForBits(n));
  auto offsets = std::make_shared<Buffer>(reinterpret_cast<const uint8_t*>(a->offsets), (n + 1) * 4);
  auto data = std::make_shared<Buffer>(a->data, n ? a->offsets[n] : 0);
  return ArrayData::Make(binary(), a->length, {validity, offsets, data}, a->validity ? a->null_count : 0, a->offset);
}
static void ExportBinary(const ArrayData& d, int32_t* out_offsets, uint8_t* out_data, uint8_t* out_validity, orc_out* out) {
  out->length = d.length;
  out->null_count = d.null_count.load();
  out->has_validity = d.buffers[0] != nullptr;
  std::memset(out_validity, 0, BitUtil::BytesForBits(d.length));
  if (d.buffers[0] && d.length > 0) arrow::internal::CopyBitmap(d.buffers[0]->data(), d.offset, d.length, out_validity, 0);
  else for (int64_t i = 0; i < d.length; ++i) BitUtil::SetBit(out_validity, i);
  const int32_t* off = d.GetValues<int32_t>(1);
  const int32_t base = d.length || d.buffers[1] ? off[0] : 0;
  for (int64_t i = 0; i <= d.length; ++i) out_offsets[i] = off[i] - base;
  if (d.length > 0 && d.buffers[2]) std::memcpy(out_data, d.buffers[2]->data() + base, off[d.length] - base);
}
int ref_binary_concatenate(const orc_binary* chunks, int nchunks, int32_t* out_offsets, uint8_t* out_data,
                           uint8_t* out_validity, orc_out* out) {
  ArrayVector arrs;
  for (int c = 0; c < nchunks; ++c) arrs.push_back(MakeArray(WrapBinary(&chunks[c])));
  auto res = Concatenate(arrs, default_memory_pool());
  if (!res.ok()) return Fail(res.status());
  ExportBinary(*(*res)->data(), out_offsets, out_data, out_validity, out);
  return 0;
}
int ref_binary_take(const orc_binary* values, const orc_array* indices, int boundscheck, int32_t* out_offsets,
                    uint8_t* out_data, uint8_t* out_validity, orc_out* out) {
  TakeOptions opts(boundscheck != 0);
  auto res = CallFunction("take", {Datum(WrapBinary(values)), Datum(Wrap(indices))}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  ExportBinary(*res->array(), out_offsets, out_data, out_validity, out);
  return 0;
}
int ref_binary_filter(const orc_binary* values, const orc_array* filter, int null_selection, int32_t* out_offsets,
                      uint8_t* out_data, uint8_t* out_validity, orc_out* out) {
  FilterOptions opts(static_cast<FilterOptions::NullSelectionBehavior>(null_selection));
  auto res = CallFunction("filter", {Datum(WrapBinary(values)), Datum(Wrap(filter))}, &opts, Ctx());
  if (!res.ok()) return Fail(res.status());
  ExportBinary(*res->array(), out_offsets, out_data, out_validity, out);
  return 0;
}

// ---- e3: implicit promotion and the casts behind it -----------------------------------------
int ref_common_numeric(int a, int b) {  // through ArithmeticFunction::DispatchBest of "add"
  std::vector<ValueDescr> d{ValueDescr::Array(TypeOf(a)), ValueDescr::Array(TypeOf(b))};
  auto f = Ctx()->func_registry()->GetFunction("add");
  if (!f.ok()) return -1;
  auto k = (*f)->DispatchBest(&d);
  if (!k.ok() || !d[0].type->Equals(*d[1].type)) return -1;
  for (int id = ORC_UINT8; id <= ORC_DOUBLE; ++id)
    if (id != 10 && TypeOf(id)->Equals(*d[0].type)) return id;
  return -1;
}
// The reference's number -> number CastFunction, taken from GetNumericCasts() directly: Cast() /
// CallFunction("cast") would initialise the whole cast table (cast.cc:47-60), whose string / nested /
// temporal / dictionary families are not part of this build (ref_build.mk).
static Result<Datum> NumericCast(const Datum& in, int to, bool safe) {
  static const auto casts = arrow::compute::internal::GetNumericCasts();
  auto type = TypeOf(to);
  CastOptions o = safe ? CastOptions::Safe(type) : CastOptions::Unsafe(type);
  for (const auto& f : casts)
    if (f->out_type_id() == type->id()) return f->Execute({in}, &o, Ctx());
  return Status::NotImplemented("cast to ", type->ToString());
}
int ref_cast_numeric(const orc_array* a, int to, int safe, void* out_values) {
  auto res = NumericCast(Datum(Wrap(a)), to, safe != 0);
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  const int64_t w = bit_width(d.type->id()) / 8;
  if (d.length > 0) memcpy(out_values, d.buffers[1]->data() + d.offset * w, static_cast<size_t>(d.length * w));
  return 0;
}
// Function::Execute (function.cc:157-201) for two arrays of different types, step by step with the
// reference's own pieces: DispatchBest rewrites the descriptors, each argument whose type changed is
// cast safely (cast.cc:228-243), the call then matches a kernel exactly.
static Status ImplicitlyCast(const char* name, Datum* l, Datum* r) {
  if (!l->is_array() || !r->is_array() || l->type()->Equals(*r->type())) return Status::OK();
  std::vector<ValueDescr> d{l->descr(), r->descr()};
  ARROW_ASSIGN_OR_RAISE(auto f, Ctx()->func_registry()->GetFunction(name));
  ARROW_RETURN_NOT_OK(f->DispatchBest(&d).status());
  Datum* args[2] = {l, r};
  for (int k = 0; k < 2; ++k) {
    if (d[k].type->Equals(*args[k]->type())) continue;
    int to = -1;
    for (int id = ORC_UINT8; id <= ORC_DOUBLE; ++id)
      if (id != 10 && TypeOf(id)->Equals(*d[k].type)) to = id;
    if (to < 0) return Status::NotImplemented("implicit cast to ", d[k].type->ToString());
    ARROW_ASSIGN_OR_RAISE(*args[k], NumericCast(*args[k], to, true));
  }
  return Status::OK();
}

static const char* kCmpNames[] = {"equal", "not_equal", "greater", "greater_equal", "less", "less_equal"};
static const char* kArithNames[] = {"add", "subtract", "multiply", "divide",
                                    "add_checked", "subtract_checked", "multiply_checked", "divide_checked",
                                    "power", "power_checked", "negate", "negate_checked"};

static int Binary(const char* name, const orc_array* lhs, const void* ls, const orc_array* rhs,
                  const void* rs, void* out_values, uint8_t* out_validity, orc_out* out) {
  const int t = lhs ? lhs->type_id : rhs->type_id;
  Datum l = lhs ? Datum(Wrap(lhs)) : Datum(WrapScalar(t, ls));
  Datum r = rhs ? Datum(Wrap(rhs)) : Datum(WrapScalar(t, rs));
  auto st = ImplicitlyCast(name, &l, &r);
  if (!st.ok()) return Fail(st);
  auto res = CallFunction(name, {l, r}, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_values, out_validity, out);
  return 0;
}

int ref_compare(int op, const orc_array* lhs, const void* ls, const orc_array* rhs, const void* rs,
                uint8_t* out_bits, uint8_t* out_validity, orc_out* out) {
  return Binary(kCmpNames[op], lhs, ls, rhs, rs, out_bits, out_validity, out);
}
int ref_arith(int op, const orc_array* lhs, const void* ls, const orc_array* rhs, const void* rs,
              void* out_values, uint8_t* out_validity, orc_out* out) {
  if (op >= ORC_NEGATE) {  // unary
    auto res = CallFunction(kArithNames[op], {Datum(Wrap(lhs))}, nullptr, Ctx());
    if (!res.ok()) return Fail(res.status());
    Export(*res->array(), out_values, out_validity, out);
    return 0;
  }
  return Binary(kArithNames[op], lhs, ls, rhs, rs, out_values, out_validity, out);
}

int ref_make_groupings(const uint32_t* ids, int64_t n, uint32_t num_groups, int32_t* out_offsets, int32_t* out_indices) {
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] >= num_groups) { g_err = "MakeGroupings: an id is not below num_groups"; return ORC_INVALID; }  // DCHECK in the reference
  UInt32Array arr(n, Buffer::Wrap(ids, static_cast<size_t>(n)));
  auto res = compute::internal::Grouper::MakeGroupings(arr, num_groups, Ctx());
  if (!res.ok()) return Fail(res.status());
  const ListArray& l = **res;
  memcpy(out_offsets, l.raw_value_offsets(), sizeof(int32_t) * (static_cast<size_t>(num_groups) + 1));
  const auto& v = static_cast<const Int32Array&>(*l.values());
  if (n > 0) memcpy(out_indices, v.raw_values(), sizeof(int32_t) * static_cast<size_t>(n));
  return 0;
}

// ---- mask-building scalars -------------------------------------------------
static const char* kBoolNames[] = {"invert", "and", "and_not", "or", "xor", "and_kleene", "and_not_kleene", "or_kleene"};

static Datum BoolOperand(const orc_array* a, int state) {
  if (a) return Datum(Wrap(a));
  if (state == ORC_SCALAR_NULL) return Datum(std::make_shared<BooleanScalar>());
  return Datum(std::make_shared<BooleanScalar>(state == ORC_SCALAR_TRUE));
}

int ref_boolean(int op, const orc_array* lhs, int lsc, const orc_array* rhs, int rsc, uint8_t* out_bits,
                uint8_t* out_validity, orc_out* out) {
  std::vector<Datum> args{BoolOperand(lhs, lsc)};
  if (op != ORC_INVERT) args.push_back(BoolOperand(rhs, rsc));
  auto res = CallFunction(kBoolNames[op], args, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  Export(d, out_bits, out_validity, out);
  if (op >= ORC_AND_KLEENE) {  // the Kleene kernels leave the count unknown: report it exactly
    out->null_count = d.GetNullCount();
    if (!out->has_validity)
      for (int64_t i = 0; i < out->length; ++i) BitUtil::SetBit(out_validity, i);
  }
  return 0;
}

int ref_validity_test(int func, const orc_array* a, uint8_t* out_bits, orc_out* out) {
  auto res = CallFunction(func == 0 ? "is_valid" : "is_null", {Datum(Wrap(a))}, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_bits, nullptr, out);
  return 0;
}

int ref_is_nan(const orc_array* a, uint8_t* out_bits, uint8_t* out_validity, orc_out* out) {
  auto res = CallFunction("is_nan", {Datum(Wrap(a))}, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_bits, out_validity, out);
  return 0;
}

int ref_fill_null(const orc_array* a, const void* fill, void* out_values, uint8_t* out_validity, orc_out* out) {
  Datum scalar = fill ? Datum(WrapScalar(a->type_id, fill)) : Datum(MakeNullScalar(TypeOf(a->type_id)));
  auto res = CallFunction("fill_null", {Datum(Wrap(a)), scalar}, nullptr, Ctx());
  if (!res.ok()) return Fail(res.status());
  Export(*res->array(), out_values, out_validity, out);
  if (!out->has_validity)
    for (int64_t i = 0; i < out->length; ++i) BitUtil::SetBit(out_validity, i);
  return 0;
}

// ---- scalar aggregates: count / sum / min_max through CallFunction ----------------
static uint64_t ScalarBits64(const Scalar& s) {  // widened to the accumulator family
  if (!s.is_valid) return 0;
  switch (s.type->id()) {
    case Type::BOOL: return checked_cast<const BooleanScalar&>(s).value ? 1 : 0;
    case Type::UINT8: return checked_cast<const UInt8Scalar&>(s).value;
    case Type::UINT16: return checked_cast<const UInt16Scalar&>(s).value;
    case Type::UINT32: return checked_cast<const UInt32Scalar&>(s).value;
    case Type::UINT64: return checked_cast<const UInt64Scalar&>(s).value;
    case Type::INT8: return static_cast<uint64_t>(static_cast<int64_t>(checked_cast<const Int8Scalar&>(s).value));
    case Type::INT16: return static_cast<uint64_t>(static_cast<int64_t>(checked_cast<const Int16Scalar&>(s).value));
    case Type::INT32: return static_cast<uint64_t>(static_cast<int64_t>(checked_cast<const Int32Scalar&>(s).value));
    case Type::INT64: return static_cast<uint64_t>(checked_cast<const Int64Scalar&>(s).value);
    case Type::FLOAT: { double d = checked_cast<const FloatScalar&>(s).value; uint64_t b; std::memcpy(&b, &d, 8); return b; }
    case Type::DOUBLE: { double d = checked_cast<const DoubleScalar&>(s).value; uint64_t b; std::memcpy(&b, &d, 8); return b; }
    default: return 0;
  }
}
// Outputs follow orc_aggregate; where the reference returns a null scalar (no valid slot) the
// field is left at the identity orc_aggregate documents, so both sides compare equal.
int ref_aggregate(const orc_array* a, orc_agg* out) {
  Datum arr(Wrap(a));
  CountOptions copts(CountOptions::COUNT_NON_NULL);
  auto c = CallFunction("count", {arr}, &copts, Ctx());
  if (!c.ok()) return Fail(c.status());
  out->valid_count = checked_cast<const Int64Scalar&>(*c->scalar()).value;
  auto s = CallFunction("sum", {arr}, nullptr, Ctx());
  if (!s.ok()) return Fail(s.status());
  out->sum = ScalarBits64(*s->scalar());
  MinMaxOptions mopts(MinMaxOptions::SKIP);
  auto m = CallFunction("min_max", {arr}, &mopts, Ctx());
  if (!m.ok()) return Fail(m.status());
  const auto& st = checked_cast<const StructScalar&>(*m->scalar());
  if (st.value[0]->is_valid) {
    out->min = ScalarBits64(*st.value[0]);
    out->max = ScalarBits64(*st.value[1]);
  } else {  // identities of MinMaxState (:140-219)
    const int bw = BitWidth(a->type_id);
    switch (a->type_id) {
      case ORC_BOOL: out->min = 1; out->max = 0; break;
      case ORC_FLOAT: case ORC_DOUBLE: {
        const double pinf = std::numeric_limits<double>::infinity(), ninf = -pinf;
        std::memcpy(&out->min, &pinf, 8); std::memcpy(&out->max, &ninf, 8);
        break;
      }
      case ORC_INT8: case ORC_INT16: case ORC_INT32: case ORC_INT64:
        out->min = static_cast<uint64_t>(bw == 64 ? INT64_MAX : (int64_t(1) << (bw - 1)) - 1);
        out->max = static_cast<uint64_t>(bw == 64 ? INT64_MIN : -(int64_t(1) << (bw - 1)));
        break;
      default:
        out->min = bw == 64 ? UINT64_MAX : (uint64_t(1) << bw) - 1;
        out->max = 0;
        break;
    }
  }
  return 0;
}

// Filter(values, mask), Filter(keys, mask), GroupBy({values}, {keys}, {hash_sum})
int64_t ref_filter_groupby_sum(const orc_array* values, const orc_array* keys, const orc_array* mask,
                               void* out_sums, int64_t* out_counts, void* out_keys,
                               uint8_t* out_key_validity) {
  FilterOptions fopts(FilterOptions::DROP);
  auto fv = CallFunction("filter", {Datum(Wrap(values)), Datum(Wrap(mask))}, &fopts, Ctx());
  if (!fv.ok()) return -Fail(fv.status());
  auto fk = CallFunction("filter", {Datum(Wrap(keys)), Datum(Wrap(mask))}, &fopts, Ctx());
  if (!fk.ok()) return -Fail(fk.status());
  CountOptions copts(CountOptions::COUNT_NON_NULL);
  auto res = compute::internal::GroupBy({*fv, *fv}, {*fk}, {{"hash_sum", nullptr}, {"hash_count", &copts}}, Ctx());
  if (!res.ok()) return -Fail(res.status());
  const ArrayData& st = *res->array();
  orc_out o;
  uint8_t* tmp = static_cast<uint8_t*>(std::malloc(BitUtil::BytesForBits(st.length) + 8));
  Export(*st.child_data[0], out_sums, tmp, &o);
  Export(*st.child_data[1], out_counts, tmp, &o);
  Export(*st.child_data[2], out_keys, out_key_validity, &o);
  if (!o.has_validity)
    for (int64_t i = 0; i < o.length; ++i) BitUtil::SetBit(out_key_validity, i);
  std::free(tmp);
  return st.length;
}

}  // extern "C"

// ---- round 3: arrays of any of the array_take / array_filter value types as node lists -----------------------------
namespace {
std::shared_ptr<DataType> NodeType(const orc_node* nodes, int i);
std::shared_ptr<DataType> ChildType(const orc_node* nodes, const orc_node& n, int c) { return NodeType(nodes, n.first_child + c); }
std::shared_ptr<DataType> NodeType(const orc_node* nodes, int i) {
  const orc_node& n = nodes[i];
  switch (n.kind) {
    case ORC_K_PRIM: return TypeOf(n.type_id);
    case ORC_K_BINARY: return binary();
    case ORC_K_UTF8: return utf8();
    case ORC_K_LARGE_BINARY: return large_binary();
    case ORC_K_LARGE_UTF8: return large_utf8();
    case ORC_K_FSB: return fixed_size_binary(n.byte_width);
    case ORC_K_DECIMAL128: return decimal128(20, 3);
    case ORC_K_DECIMAL256: return decimal256(40, 3);
    case ORC_K_LIST: return list(ChildType(nodes, n, 0));
    case ORC_K_LARGE_LIST: return large_list(ChildType(nodes, n, 0));
    case ORC_K_MAP: {
      const orc_node& st = nodes[n.first_child];
      return map(NodeType(nodes, st.first_child), NodeType(nodes, st.first_child + 1));
    }
    case ORC_K_FSL: return fixed_size_list(ChildType(nodes, n, 0), n.list_size);
    case ORC_K_STRUCT: {
      std::vector<std::shared_ptr<Field>> f;
      for (int c = 0; c < n.nchildren; ++c) f.push_back(field("f" + std::to_string(c), ChildType(nodes, n, c)));
      return struct_(f);
    }
    case ORC_K_DICT: return dictionary(TypeOf(n.type_id), ChildType(nodes, n, 0));
    case ORC_K_NULL: return null();
  }
  return nullptr;
}
std::shared_ptr<Buffer> Bits(const uint8_t* p, int64_t n) { return p ? std::make_shared<Buffer>(p, BitUtil::BytesForBits(n)) : nullptr; }
std::shared_ptr<ArrayData> NodeData(const orc_node* nodes, int i, std::shared_ptr<DataType> type = nullptr) {
  const orc_node& n = nodes[i];
  if (!type) type = NodeType(nodes, i);
  const int64_t end = n.offset + n.length;
  const int64_t nc = n.validity ? n.null_count : 0;
  auto validity = Bits(n.validity, end);
  auto raw = [](const void* p, int64_t bytes) { return std::make_shared<Buffer>(reinterpret_cast<const uint8_t*>(p), bytes); };
  switch (n.kind) {
    case ORC_K_PRIM: return ArrayData::Make(type, n.length, {validity, raw(n.buf1, BufBytes(BitWidth(n.type_id), end))}, nc, n.offset);
    case ORC_K_BINARY: case ORC_K_UTF8: {
      const int32_t* off = static_cast<const int32_t*>(n.buf1);
      return ArrayData::Make(type, n.length, {validity, raw(off, (end + 1) * 4), raw(n.buf2, end ? off[end] : 0)}, nc, n.offset);
    }
    case ORC_K_LARGE_BINARY: case ORC_K_LARGE_UTF8: {
      const int64_t* off = static_cast<const int64_t*>(n.buf1);
      return ArrayData::Make(type, n.length, {validity, raw(off, (end + 1) * 8), raw(n.buf2, end ? off[end] : 0)}, nc, n.offset);
    }
    case ORC_K_FSB: return ArrayData::Make(type, n.length, {validity, raw(n.buf1, end * n.byte_width)}, nc, n.offset);
    case ORC_K_DECIMAL128: return ArrayData::Make(type, n.length, {validity, raw(n.buf1, end * 16)}, nc, n.offset);
    case ORC_K_DECIMAL256: return ArrayData::Make(type, n.length, {validity, raw(n.buf1, end * 32)}, nc, n.offset);
    case ORC_K_LIST: case ORC_K_MAP:
      return ArrayData::Make(type, n.length, {validity, raw(n.buf1, (end + 1) * 4)}, {NodeData(nodes, n.first_child, type->field(0)->type())}, nc, n.offset);
    case ORC_K_LARGE_LIST:
      return ArrayData::Make(type, n.length, {validity, raw(n.buf1, (end + 1) * 8)}, {NodeData(nodes, n.first_child)}, nc, n.offset);
    case ORC_K_FSL: return ArrayData::Make(type, n.length, {validity}, {NodeData(nodes, n.first_child)}, nc, n.offset);
    case ORC_K_STRUCT: {
      std::vector<std::shared_ptr<ArrayData>> kids;
      for (int c = 0; c < n.nchildren; ++c) kids.push_back(NodeData(nodes, n.first_child + c, type->field(c)->type()));
      return ArrayData::Make(type, n.length, {validity}, kids, nc, n.offset);
    }
    case ORC_K_DICT: {
      auto d = ArrayData::Make(type, n.length, {validity, raw(n.buf1, BufBytes(BitWidth(n.type_id), end))}, nc, n.offset);
      d->dictionary = NodeData(nodes, n.first_child);
      return d;
    }
    case ORC_K_NULL: return ArrayData::Make(type, n.length, {nullptr}, n.length, n.offset);
  }
  return nullptr;
}

struct Held {
  std::shared_ptr<ArrayData> root;
  std::vector<orc_node> nodes;
  std::vector<std::shared_ptr<ArrayData>> keep;
} g_held;
void FillAt(size_t at, const std::shared_ptr<ArrayData>& d);
int KindOf(const DataType& t) {
  switch (t.id()) {
    case Type::BINARY: return ORC_K_BINARY;
    case Type::STRING: return ORC_K_UTF8;
    case Type::LARGE_BINARY: return ORC_K_LARGE_BINARY;
    case Type::LARGE_STRING: return ORC_K_LARGE_UTF8;
    case Type::FIXED_SIZE_BINARY: return ORC_K_FSB;
    case Type::DECIMAL128: return ORC_K_DECIMAL128;
    case Type::DECIMAL256: return ORC_K_DECIMAL256;
    case Type::LIST: return ORC_K_LIST;
    case Type::LARGE_LIST: return ORC_K_LARGE_LIST;
    case Type::MAP: return ORC_K_MAP;
    case Type::FIXED_SIZE_LIST: return ORC_K_FSL;
    case Type::STRUCT: return ORC_K_STRUCT;
    case Type::DICTIONARY: return ORC_K_DICT;
    case Type::NA: return ORC_K_NULL;
    default: return ORC_K_PRIM;
  }
}
int IdOf(const DataType& t) {
  for (int id = 1; id <= 12; ++id) { auto c = TypeOf(id); if (c && c->Equals(t)) return id; }
  return 0;
}
void FillAt(size_t at, const std::shared_ptr<ArrayData>& d) {
  g_held.keep.push_back(d);
  orc_node n{};
  n.kind = KindOf(*d->type);
  n.length = d->length; n.offset = d->offset; n.null_count = d->null_count.load();
  n.validity = d->buffers.size() > 0 && d->buffers[0] ? d->buffers[0]->data() : nullptr;
  if (n.kind == ORC_K_PRIM) n.type_id = IdOf(*d->type);
  if (n.kind == ORC_K_DICT) n.type_id = IdOf(*checked_cast<const DictionaryType&>(*d->type).index_type());
  if (n.kind == ORC_K_FSB) n.byte_width = checked_cast<const FixedSizeBinaryType&>(*d->type).byte_width();
  if (n.kind == ORC_K_DECIMAL128) n.byte_width = 16;
  if (n.kind == ORC_K_DECIMAL256) n.byte_width = 32;
  if (n.kind == ORC_K_FSL) n.list_size = checked_cast<const FixedSizeListType&>(*d->type).list_size();
  if (d->buffers.size() > 1 && d->buffers[1]) { n.buf1 = d->buffers[1]->data(); n.buf1_bytes = d->buffers[1]->size(); }
  if (d->buffers.size() > 2 && d->buffers[2]) { n.buf2 = d->buffers[2]->data(); n.buf2_bytes = d->buffers[2]->size(); }
  std::vector<std::shared_ptr<ArrayData>> kids = d->child_data;
  if (n.kind == ORC_K_DICT) kids = {d->dictionary};
  n.nchildren = static_cast<int>(kids.size());
  n.first_child = static_cast<int>(g_held.nodes.size());
  g_held.nodes[at] = n;
  const size_t first = g_held.nodes.size();
  g_held.nodes.resize(first + kids.size());
  for (size_t c = 0; c < kids.size(); ++c) FillAt(first + c, kids[c]);
}
void Hold(const std::shared_ptr<ArrayData>& d) {
  g_held.root = d;
  g_held.nodes.clear();
  g_held.keep.clear();
  g_held.nodes.resize(1);
  FillAt(0, d);
}
}  // namespace

extern "C" {
int ref_nested_select(const orc_node* nodes, int nnodes, const orc_array* selection, int is_filter, int option) {
  (void)nnodes;
  auto values = NodeData(nodes, 0);
  Result<Datum> res = Status::Invalid("unset");
  if (is_filter) {
    FilterOptions opts(static_cast<FilterOptions::NullSelectionBehavior>(option));
    res = CallFunction("filter", {Datum(values), Datum(Wrap(selection))}, &opts, Ctx());
  } else {
    TakeOptions opts(option != 0);
    res = CallFunction("take", {Datum(values), Datum(Wrap(selection))}, &opts, Ctx());
  }
  if (!res.ok()) return Fail(res.status());
  Hold(res->array());
  return 0;
}
int ref_nested_result_count(void) { return static_cast<int>(g_held.nodes.size()); }
int ref_nested_result_node(int i, orc_node* out) {
  if (i < 0 || i >= static_cast<int>(g_held.nodes.size())) return ORC_INVALID;
  *out = g_held.nodes[i];
  return 0;
}

struct ref_ngrouper {
  std::unique_ptr<compute::internal::Grouper> impl;
  int nkeys;
};
ref_ngrouper* ref_nested_grouper_new(const orc_node* nodes, int nnodes, const int* roots, int nkeys) {
  (void)nnodes;
  std::vector<ValueDescr> descrs;
  for (int k = 0; k < nkeys; ++k) descrs.push_back(ValueDescr::Array(NodeType(nodes, roots[k])));
  auto res = compute::internal::Grouper::Make(descrs, Ctx());
  if (!res.ok()) { Fail(res.status()); return nullptr; }
  auto* g = new ref_ngrouper;
  g->impl = std::move(*res);
  g->nkeys = nkeys;
  return g;
}
void ref_nested_grouper_free(ref_ngrouper* g) { delete g; }
int ref_nested_grouper_consume(ref_ngrouper* g, const orc_node* nodes, int nnodes, const int* roots, uint32_t* out_ids) {
  (void)nnodes;
  std::vector<Datum> cols;
  for (int k = 0; k < g->nkeys; ++k) cols.emplace_back(NodeData(nodes, roots[k]));
  auto batch = ExecBatch::Make(cols);
  if (!batch.ok()) return Fail(batch.status());
  auto res = g->impl->Consume(*batch);
  if (!res.ok()) return Fail(res.status());
  const ArrayData& d = *res->array();
  std::memcpy(out_ids, d.buffers[1]->data() + d.offset * 4, d.length * 4);
  return 0;
}
uint32_t ref_nested_grouper_num_groups(const ref_ngrouper* g) { return g->impl->num_groups(); }
int ref_nested_grouper_uniques(ref_ngrouper* g, int k) {
  auto res = g->impl->GetUniques();
  if (!res.ok()) return Fail(res.status());
  Hold(res->values[k].array());
  return 0;
}
}  // extern "C"

// host_test.cc -- tests of the C++ host mirror, driven by pytest (tests/test_host_cpp.py).
//
//   host_test cpu   plumbing with toy kernels over HOST buffers, no device: registry semantics,
//                   arity / dispatch errors, ExecBatchIterator chunking, executor result wrapping --
//                   the pattern of the reference's compute/exec_test.cc:462-515, 635-800 and
//                   registry_test.cc.
//   host_test gpu   CallFunction / GroupBy on the device against the reference's known-answer
//                   vectors (vector_selection_test.cc:296-321, 1038-1051; vector_sort_test.cc:423-464,
//                   629-647; hash_aggregate_test.cc:251-292, 480-605; scalar_compare_test.cc:248-296;
//                   scalar_arithmetic_test.cc:337-362) and its error codes.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <map>
#include <tuple>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <thread>

#include "amdarrow_compute.h"
#include "expression.h"
#include "sharded_groupby.h"

using namespace amdarrow;           // NOLINT
using namespace amdarrow::compute;  // NOLINT

static int g_failures = 0, g_checks = 0;
#define CHECK(cond)                                                              \
  do {                                                                           \
    ++g_checks;                                                                  \
    if (!(cond)) { ++g_failures; fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
  } while (0)
#define CHECK_OK(expr)                                                           \
  do {                                                                           \
    ++g_checks;                                                                  \
    auto _s = (expr);                                                            \
    if (!_s.ok()) { ++g_failures; fprintf(stderr, "FAIL %s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, _s.ToString().c_str()); } \
  } while (0)

// ------------------------------------------------------------------------------------------------- cpu
static std::shared_ptr<ArrayData> HostInt32(const std::vector<int32_t>& v) {
  auto* copy = new std::vector<int32_t>(v);  // leaked on purpose: test-lifetime storage
  return ArrayData::Make(int32(), static_cast<int64_t>(v.size()), {nullptr, Buffer::WrapHost(copy->data(), 4 * copy->size())}, 0);
}

// CheckDispatchBest (scalar_arithmetic_test.cc:863-895, 920-922; scalar_compare_test.cc:454-481, the numeric rows):
// the descriptors DispatchBest settles on.  No device involved.
static void DispatchBestTests() {
  auto* registry = GetFunctionRegistry();
  struct Row { std::shared_ptr<DataType> a, b, want; };
  const std::vector<Row> rows = {
      {int32(), int32(), int32()},   {int32(), null(), int32()},    {null(), int32(), int32()},    {int32(), int8(), int32()},
      {int32(), int16(), int32()},   {int32(), int64(), int64()},   {int32(), uint8(), int32()},   {int32(), uint16(), int32()},
      {int32(), uint32(), int64()},  {int32(), uint64(), int64()},  {uint8(), uint8(), uint8()},   {uint8(), uint16(), uint16()},
      {int32(), float32(), float32()}, {float32(), int64(), float32()}, {float64(), int32(), float64()},
      {dictionary(float64()), float64(), float64()}, {dictionary(float64()), int16(), float64()}, {int8(), uint64(), int64()}};
  for (const char* name : {"add", "subtract", "multiply", "divide", "add_checked", "subtract_checked", "multiply_checked",
                           "divide_checked", "equal", "not_equal", "less", "less_equal", "greater", "greater_equal"}) {
    auto fr = registry->GetFunction(name);
    CHECK_OK(fr.status());
    if (!fr.ok()) continue;
    auto* f = static_cast<const ScalarFunction*>(fr->get());
    for (const auto& r : rows) {
      std::vector<ValueDescr> d{ValueDescr::Array(r.a), ValueDescr::Array(r.b)};
      auto k = f->DispatchBest(&d);
      CHECK_OK(k.status());
      CHECK(d[0].type->id() == r.want->id() && d[1].type->id() == r.want->id());
    }
    std::vector<ValueDescr> d{ValueDescr::Array(boolean()), ValueDescr::Array(int8())};  // nothing to meet at
    CHECK(f->DispatchBest(&d).status().IsNotImplemented());
  }
}

static int RunCpu() {
  DispatchBestTests();
  // ---- registry semantics (registry.cc:35-110; registry_test.cc) ----
  auto registry = FunctionRegistry::Make();
  CHECK(registry->num_functions() == 0);
  auto f1 = std::make_shared<ScalarFunction>("f1", Arity::Unary());
  CHECK_OK(registry->AddFunction(f1));
  CHECK(registry->AddFunction(std::make_shared<ScalarFunction>("f1", Arity::Unary())).IsKeyError());   // no overwrite by default
  CHECK_OK(registry->AddFunction(std::make_shared<ScalarFunction>("f1", Arity::Binary()), /*allow_overwrite=*/true));
  CHECK_OK(registry->AddAlias("f1_alias", "f1"));
  CHECK(registry->AddAlias("x", "no_such").IsKeyError());
  CHECK(registry->num_functions() == 2);
  CHECK(registry->GetFunction("nope").status().IsKeyError());
  CHECK(registry->GetFunction("nope").status().message() == "No function registered with name: nope");
  auto names = registry->GetFunctionNames();
  CHECK(names.size() == 2 && names[0] == "f1" && names[1] == "f1_alias");

  // ---- toy kernels on host memory (exec_test.cc "test_copy" style) ----
  ExecContext ctx(nullptr, registry.get());
  int exec_calls = 0;
  std::vector<int64_t> seen_lengths;
  auto copy_fn = std::make_shared<VectorFunction>("test_copy", Arity::Unary());
  VectorKernel k;
  k.signature.in_types = {InputType::Exact(Type::INT32)};
  k.signature.out_type = [](const std::vector<ValueDescr>& d) -> Result<std::shared_ptr<DataType>> { return d[0].type; };
  k.exec = [&](KernelContext*, const ExecBatch& batch, Datum* out) -> Status {
    ++exec_calls;
    seen_lengths.push_back(batch.length);
    *out = batch[0];  // zero-copy "copy"
    return Status::OK();
  };
  CHECK_OK(copy_fn->AddKernel(k));
  CHECK_OK(registry->AddFunction(copy_fn));
  auto arr = HostInt32({1, 2, 3, 4, 5, 6, 7, 8, 9, 10});
  {
    auto r = CallFunction("test_copy", {Datum(arr)}, &ctx);
    CHECK_OK(r.status());
    CHECK(r->is_array() && r->array()->length == 10 && exec_calls == 1);
  }
  {  // exec_chunksize splits the array into batches -> ChunkedArray out (exec.cc:762-776)
    ctx.set_exec_chunksize(4);
    exec_calls = 0; seen_lengths.clear();
    auto r = CallFunction("test_copy", {Datum(arr)}, &ctx);
    CHECK_OK(r.status());
    CHECK(exec_calls == 3 && seen_lengths == std::vector<int64_t>({4, 4, 2}));
    CHECK(r->kind() == Datum::CHUNKED_ARRAY && r->chunked_array()->num_chunks() == 3 && r->chunked_array()->length() == 10);
    CHECK(r->chunked_array()->chunk(1)->offset == 4);
    ctx.set_exec_chunksize(kDefaultExecChunksize);
  }
  {  // chunked input -> one exec per chunk, empty chunks dropped (exec.cc:468-471)
    auto ca = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{HostInt32({1, 2, 3}), HostInt32({}), HostInt32({4, 5})}, int32());
    exec_calls = 0; seen_lengths.clear();
    auto r = CallFunction("test_copy", {Datum(ca)}, &ctx);
    CHECK_OK(r.status());
    CHECK(seen_lengths == std::vector<int64_t>({3, 2}));
    CHECK(r->kind() == Datum::CHUNKED_ARRAY && r->chunked_array()->num_chunks() == 2);
  }
  // arity -> Invalid (function.cc:41-56); no matching kernel -> NotImplemented; unknown -> KeyError
  CHECK(CallFunction("test_copy", {Datum(arr), Datum(arr)}, &ctx).status().IsInvalid());
  auto f64arr = ArrayData::Make(float64(), 0, {nullptr, Buffer::WrapHost(nullptr, 0)}, 0);
  CHECK(CallFunction("test_copy", {Datum(f64arr)}, &ctx).status().IsNotImplemented());
  CHECK(CallFunction("no_such_function", {Datum(arr)}, &ctx).status().IsKeyError());

  // ---- ExecBatchIterator (exec_test.cc:462-515) ----
  {
    auto a = HostInt32({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
    auto ca = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{HostInt32({0, 1, 2, 3}), HostInt32({4, 5, 6, 7, 8, 9})}, int32());
    auto s = Scalar::Make(int32(), int32_t(7));
    auto it = detail::ExecBatchIterator::Make({Datum(a), Datum(ca), Datum(s)}, 3);
    CHECK_OK(it.status());
    std::vector<int64_t> lens;
    ExecBatch b;
    while ((*it)->Next(&b)) {
      lens.push_back(b.length);
      CHECK(b[0].array()->length == b.length && b[1].array()->length == b.length && b[2].is_scalar());
    }
    CHECK(lens == std::vector<int64_t>({3, 1, 3, 3}));  // chunk boundary at 4 respected
    CHECK((*it)->position() == 10);
    // length mismatch -> Invalid "Array arguments must all be the same length" (exec.cc:190-192)
    auto bad = detail::ExecBatchIterator::Make({Datum(a), Datum(HostInt32({1, 2}))}, 100);
    CHECK(bad.status().IsInvalid() && bad.status().message() == "Array arguments must all be the same length");
  }
  {  // ExecBatch::Make
    auto ok = ExecBatch::Make({Datum(HostInt32({1, 2})), Datum(Scalar::Make(int32(), int32_t(1)))});
    CHECK_OK(ok.status());
    CHECK(ok->length == 2);
    CHECK(ExecBatch::Make({Datum(HostInt32({1, 2})), Datum(HostInt32({1}))}).status().IsInvalid());
  }
  // ---- the default (device) registry carries the hot-path functions under the reference's names ----
  {
    FunctionRegistry* dev = GetFunctionRegistry();
    for (const char* n : {"filter", "array_filter", "take", "array_take", "sort_indices", "array_sort_indices", "partition_nth_indices", "unique", "value_counts", "dictionary_encode", "equal", "not_equal",
                          "greater", "greater_equal", "less", "less_equal", "add", "subtract", "multiply", "divide", "add_checked",
                          "subtract_checked", "multiply_checked", "divide_checked", "hash_sum", "hash_count", "hash_mean", "hash_min_max",
                          "invert", "and", "and_not", "or", "xor", "and_kleene", "and_not_kleene", "or_kleene", "is_valid",
                          "is_null", "is_nan", "fill_null", "count", "sum", "mean", "min_max"})
      CHECK(dev->GetFunction(n).ok());
    auto f = dev->GetFunction("hash_sum");
    CHECK((*f)->kind() == Function::HASH_AGGREGATE && (*f)->arity().num_args == 3);
    ExecContext dctx(nullptr);  // no pool: anything that must allocate fails loudly, never falls back
    auto direct = CallFunction("hash_sum", {Datum(arr), Datum(arr), Datum(arr)}, &dctx);
    CHECK(direct.status().IsNotImplemented() && direct.status().message() == "Direct execution of HASH_AGGREGATE functions");
    CHECK((*dev->GetFunction("filter"))->kind() == Function::META);
    CHECK((*dev->GetFunction("array_take"))->num_kernels() == 16);  // primitive, binary, utf8 + the 13 other value types
    CHECK((*dev->GetFunction("array_sort_indices"))->num_kernels() == 10);
    // a product call without a device pool must raise, not compute on the CPU
    auto nofallback = CallFunction("add", {Datum(arr), Datum(arr)}, &dctx);
    CHECK(!nofallback.ok());
  }
  printf("host_test cpu: %d checks, %d failures\n", g_checks, g_failures);
  return g_failures ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------- gpu
template <typename T> struct Opt { bool valid; T v; };
template <typename T>
static std::shared_ptr<ArrayData> Dev(const std::shared_ptr<MemoryPool>& pool, std::shared_ptr<DataType> type, const std::vector<Opt<T>>& items) {
  std::vector<T> vals;
  std::vector<uint8_t> bits((items.size() + 7) / 8 + 1, 0);
  bool any_null = false;
  int64_t nulls = 0;
  for (size_t i = 0; i < items.size(); ++i) {
    vals.push_back(items[i].valid ? items[i].v : T{});
    if (items[i].valid) bits[i / 8] |= uint8_t(1u << (i % 8)); else { any_null = true; ++nulls; }
  }
  if (vals.empty()) vals.push_back(T{});
  auto r = ArrayFromHost(pool, type, vals.data(), static_cast<int64_t>(items.size()), any_null ? bits.data() : nullptr, nulls);
  return r.ok() ? *r : nullptr;
}
static std::shared_ptr<ArrayData> DevBool(const std::shared_ptr<MemoryPool>& pool, const std::vector<int>& items /* -1 = null */) {
  std::vector<uint8_t> vals((items.size() + 7) / 8 + 1, 0), bits((items.size() + 7) / 8 + 1, 0);
  bool any_null = false;
  int64_t nulls = 0;
  for (size_t i = 0; i < items.size(); ++i) {
    if (items[i] > 0) vals[i / 8] |= uint8_t(1u << (i % 8));
    if (items[i] >= 0) bits[i / 8] |= uint8_t(1u << (i % 8)); else { any_null = true; ++nulls; }
  }
  auto r = ArrayFromHost(pool, boolean(), vals.data(), static_cast<int64_t>(items.size()), any_null ? bits.data() : nullptr, nulls);
  return r.ok() ? *r : nullptr;
}
template <typename T>
static std::vector<Opt<T>> Host(const std::shared_ptr<MemoryPool>& pool, const ArrayData& a) {
  std::vector<T> vals(a.length + 1);
  std::vector<uint8_t> bits((a.length + 7) / 8 + 1, 0);
  auto st = CopyValuesToHost(a, pool, vals.data(), bits.data());
  std::vector<Opt<T>> out;
  if (!st.ok()) { ++g_failures; fprintf(stderr, "copy failed: %s\n", st.ToString().c_str()); return out; }
  for (int64_t i = 0; i < a.length; ++i) out.push_back({bool((bits[i / 8] >> (i % 8)) & 1), vals[i]});
  return out;
}
static std::vector<int> HostBool(const std::shared_ptr<MemoryPool>& pool, const ArrayData& a) {
  std::vector<uint8_t> vals((a.length + 7) / 8 + 1, 0), bits((a.length + 7) / 8 + 1, 0);
  auto st = CopyValuesToHost(a, pool, vals.data(), bits.data());
  std::vector<int> out;
  if (!st.ok()) { ++g_failures; return out; }
  for (int64_t i = 0; i < a.length; ++i) out.push_back(((bits[i / 8] >> (i % 8)) & 1) ? int((vals[i / 8] >> (i % 8)) & 1) : -1);
  return out;
}
template <typename T>
static bool Same(const std::vector<Opt<T>>& got, const std::vector<Opt<T>>& want) {
  if (got.size() != want.size()) return false;
  for (size_t i = 0; i < got.size(); ++i) {
    if (got[i].valid != want[i].valid) return false;
    if (got[i].valid && !(got[i].v == want[i].v || (got[i].v != got[i].v && want[i].v != want[i].v))) return false;
  }
  return true;
}
#define N(T) Opt<T>{false, T{}}
#define V(T, x) Opt<T>{true, T(x)}

template <typename T>
static void TypedTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx, std::shared_ptr<DataType> type) {
  using O = Opt<T>;
  // FilterNumeric (vector_selection_test.cc:296-321)
  struct FC { std::vector<O> values; std::vector<int> filter; std::vector<O> emit_null, drop; };
  std::vector<FC> fcs = {
      {{V(T, 9)}, {0}, {}, {}}, {{V(T, 9)}, {1}, {V(T, 9)}, {V(T, 9)}}, {{V(T, 9)}, {-1}, {N(T)}, {}},
      {{N(T)}, {1}, {N(T)}, {N(T)}}, {{N(T)}, {-1}, {N(T)}, {}},
      {{V(T, 7), V(T, 8), V(T, 9)}, {0, 1, 0}, {V(T, 8)}, {V(T, 8)}},
      {{V(T, 7), V(T, 8), V(T, 9)}, {1, 0, 1}, {V(T, 7), V(T, 9)}, {V(T, 7), V(T, 9)}},
      {{N(T), V(T, 8), V(T, 9)}, {0, 1, 0}, {V(T, 8)}, {V(T, 8)}},
      {{V(T, 7), V(T, 8), V(T, 9)}, {-1, 1, 0}, {N(T), V(T, 8)}, {V(T, 8)}},
      {{V(T, 7), V(T, 8), V(T, 9)}, {1, -1, 1}, {V(T, 7), N(T), V(T, 9)}, {V(T, 7), V(T, 9)}},
  };
  for (const auto& c : fcs) {
    for (int mode = 0; mode < 2; ++mode) {
      FilterOptions opts(mode ? FilterOptions::EMIT_NULL : FilterOptions::DROP);
      auto r = Filter(Datum(Dev<T>(pool, type, c.values)), Datum(DevBool(pool, c.filter)), opts, ctx);
      CHECK_OK(r.status());
      if (r.ok()) CHECK(Same(Host<T>(pool, *r->array()), mode ? c.emit_null : c.drop));
    }
  }
  {  // sliced filter: [0,1,1,1,0,1].Slice(3,3) over [7,8,9] -> [7,9]
    auto f = DevBool(pool, {0, 1, 1, 1, 0, 1})->Slice(3, 3);
    auto r = Filter(Datum(Dev<T>(pool, type, {V(T, 7), V(T, 8), V(T, 9)})), Datum(f), FilterOptions::Defaults(), ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(Same(Host<T>(pool, *r->array()), {V(T, 7), V(T, 9)}));
    auto bad = Filter(Datum(Dev<T>(pool, type, {V(T, 7), V(T, 8), V(T, 9)})), Datum(DevBool(pool, {})), FilterOptions::Defaults(), ctx);
    CHECK(bad.status().IsInvalid());  // :317-320
  }
  // TakeNumeric (:1038-1051) with int8 and uint32 indices
  {
    auto vals = Dev<T>(pool, type, {N(T), V(T, 8), V(T, 9)});
    auto r = Take(Datum(vals), Datum(Dev<int8_t>(pool, int8(), {V(int8_t, 0), V(int8_t, 1), V(int8_t, 0)})), TakeOptions::Defaults(), ctx);
    CHECK_OK(r.status());
    if (r.ok()) { CHECK(Same(Host<T>(pool, *r->array()), {N(T), V(T, 8), N(T)})); CHECK(r->array()->null_count == 2); }
    auto r2 = Take(Datum(Dev<T>(pool, type, {V(T, 7), V(T, 8), V(T, 9)})), Datum(Dev<uint32_t>(pool, uint32(), {N(uint32_t), V(uint32_t, 1), V(uint32_t, 0)})),
                   TakeOptions::Defaults(), ctx);
    CHECK_OK(r2.status());
    if (r2.ok()) CHECK(Same(Host<T>(pool, *r2->array()), {N(T), V(T, 8), V(T, 7)}));
    auto oob = Take(Datum(vals), Datum(Dev<int8_t>(pool, int8(), {V(int8_t, 0), V(int8_t, 9), V(int8_t, 0)})), TakeOptions::Defaults(), ctx);
    CHECK(oob.status().IsIndexError() && oob.status().message() == "Index 9 out of bounds");
    auto neg = Take(Datum(vals), Datum(Dev<int8_t>(pool, int8(), {V(int8_t, 0), V(int8_t, -1), V(int8_t, 0)})), TakeOptions::Defaults(), ctx);
    CHECK(neg.status().IsIndexError());
  }
  // sort_indices (vector_sort_test.cc:448-464)
  {
    auto a = Dev<T>(pool, type, {V(T, 10), V(T, 12), V(T, 4), V(T, 50), V(T, 50), V(T, 32), V(T, 11)});
    auto asc = SortIndices(Datum(a), SortOrder::Ascending, ctx);
    auto desc = SortIndices(Datum(a), SortOrder::Descending, ctx);
    CHECK_OK(asc.status()); CHECK_OK(desc.status());
    if (asc.ok()) CHECK(Same(Host<uint64_t>(pool, *asc->array()), {V(uint64_t, 2), V(uint64_t, 0), V(uint64_t, 6), V(uint64_t, 1), V(uint64_t, 5), V(uint64_t, 3), V(uint64_t, 4)}));
    if (desc.ok()) CHECK(Same(Host<uint64_t>(pool, *desc->array()), {V(uint64_t, 3), V(uint64_t, 4), V(uint64_t, 5), V(uint64_t, 1), V(uint64_t, 6), V(uint64_t, 0), V(uint64_t, 2)}));
    auto n = Dev<T>(pool, type, {N(T), V(T, 1), V(T, 3), N(T), V(T, 2), V(T, 5)});
    auto r = SortIndices(Datum(n), SortOrder::Descending, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(Same(Host<uint64_t>(pool, *r->array()), {V(uint64_t, 5), V(uint64_t, 2), V(uint64_t, 4), V(uint64_t, 1), V(uint64_t, 0), V(uint64_t, 3)}));
  }
  // compare with a scalar (scalar_compare_test.cc:248-296) and arithmetic (scalar_arithmetic_test.cc:337-362)
  {
    auto one = Datum(Scalar::Make(type, T(1)));
    auto a = Dev<T>(pool, type, {N(T), V(T, 0), V(T, 1), V(T, 1)});
    auto r = Compare(Datum(a), one, GREATER_EQUAL, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(HostBool(pool, *r->array()) == std::vector<int>({-1, 0, 1, 1}));
    auto r2 = Compare(Datum(Dev<T>(pool, type, {V(T, 0), V(T, 1), V(T, 2), V(T, 3), V(T, 4), V(T, 5)})), one, LESS_EQUAL, ctx);
    CHECK_OK(r2.status());
    if (r2.ok()) CHECK(HostBool(pool, *r2->array()) == std::vector<int>({1, 1, 0, 0, 0, 0}));
    for (bool checked : {false, true}) {
      auto s = Add(Datum(Dev<T>(pool, type, {N(T), V(T, 1), V(T, 2)})), Datum(Dev<T>(pool, type, {V(T, 3), V(T, 4), N(T)})), ArithmeticOptions(checked), ctx);
      CHECK_OK(s.status());
      if (s.ok()) CHECK(Same(Host<T>(pool, *s->array()), {N(T), V(T, 5), N(T)}));
      auto s2 = Subtract(Datum(Scalar::Make(type, T(3))), Datum(Dev<T>(pool, type, {N(T), V(T, 2)})), ArithmeticOptions(checked), ctx);
      CHECK_OK(s2.status());
      if (s2.ok()) CHECK(Same(Host<T>(pool, *s2->array()), {N(T), V(T, 1)}));
      auto s3 = Add(Datum(Dev<T>(pool, type, {V(T, 1), V(T, 2)})), Datum(Scalar::MakeNull(type)), ArithmeticOptions(checked), ctx);
      CHECK_OK(s3.status());
      if (s3.ok()) CHECK(Same(Host<T>(pool, *s3->array()), {N(T), N(T)}));
    }
  }
  // Grouper.NumericKey (hash_aggregate_test.cc:251-278)
  {
    auto g = internal::Grouper::Make({ValueDescr::Array(type)}, ctx);
    CHECK_OK(g.status());
    if (g.ok()) {
      auto consume = [&](const std::vector<O>& keys, const std::vector<Opt<uint32_t>>& want) {
        auto b = ExecBatch::Make({Datum(Dev<T>(pool, type, keys))});
        auto ids = (*g)->Consume(*b);
        CHECK_OK(ids.status());
        if (ids.ok()) CHECK(Same(Host<uint32_t>(pool, *ids->array()), want));
      };
      consume({V(T, 3), V(T, 3)}, {V(uint32_t, 0), V(uint32_t, 0)});
      consume({V(T, 3), V(T, 3)}, {V(uint32_t, 0), V(uint32_t, 0)});
      consume({V(T, 27), V(T, 81)}, {V(uint32_t, 1), V(uint32_t, 2)});
      consume({V(T, 3), V(T, 27), V(T, 3), V(T, 27), N(T), V(T, 81), V(T, 27), V(T, 81)},
              {V(uint32_t, 0), V(uint32_t, 1), V(uint32_t, 0), V(uint32_t, 1), V(uint32_t, 3), V(uint32_t, 2), V(uint32_t, 1), V(uint32_t, 2)});
      CHECK((*g)->num_groups() == 4);
      auto u = (*g)->GetUniques();
      CHECK_OK(u.status());
      if (u.ok()) CHECK(Same(Host<T>(pool, *u->values[0].array()), {V(T, 3), V(T, 27), V(T, 81), N(T)}));
    }
  }
}

// ---- batches and tables through the meta functions ---------------------------
// Known answers of vector_selection_test.cc:660-689 (FilterRecordBatch), :737-796 (FilterTable),
// :1460-1493 (TakeRecordBatch), :1536-1550 (TakeChunkedArray), :1596-1609 (TakeTable) and
// vector_sort_test.cc:763-980 (batch / table sort_indices).  The reference's second column is utf8,
// which is outside this path: a float64 column numbering the rows (10, 11, ...) stands in for it.
static std::vector<uint64_t> HostU64(const std::shared_ptr<MemoryPool>& pool, const ArrayData& a) {
  std::vector<uint64_t> out;
  for (const auto& o : Host<uint64_t>(pool, a)) out.push_back(o.v);
  return out;
}
static void TabularTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  using I = int32_t;
  using D = double;
  auto schm = schema({field("a", int32()), field("b", float64())});
  auto a = Dev<I>(pool, int32(), {N(I), V(I, 1), V(I, 2), V(I, 4)});
  auto b = Dev<D>(pool, float64(), {V(D, 10), V(D, 11), V(D, 12), V(D, 13)});
  auto batch = RecordBatch::Make(schm, 4, {a, b});
  FilterOptions drop(FilterOptions::DROP), emit(FilterOptions::EMIT_NULL);
  auto check_batch = [&](const Result<Datum>& r, const std::vector<Opt<I>>& wa, const std::vector<Opt<D>>& wb) {
    CHECK_OK(r.status());
    if (!r.ok()) return;
    CHECK(r->kind() == Datum::RECORD_BATCH);
    const auto& out = *r->record_batch();
    CHECK(out.num_rows() == static_cast<int64_t>(wa.size()) && out.num_columns() == 2 && out.schema() == schm);
    CHECK(Same(Host<I>(pool, *out.column(0)), wa));
    CHECK(Same(Host<D>(pool, *out.column(1)), wb));
  };
  for (const auto* o : {&emit, &drop}) {
    check_batch(Filter(Datum(batch), Datum(DevBool(pool, {0, 0, 0, 0})), *o, ctx), {}, {});
    check_batch(Filter(Datum(batch), Datum(DevBool(pool, {1, 1, 1, 1})), *o, ctx), {N(I), V(I, 1), V(I, 2), V(I, 4)},
                {V(D, 10), V(D, 11), V(D, 12), V(D, 13)});
    check_batch(Filter(Datum(batch), Datum(DevBool(pool, {1, 0, 1, 0})), *o, ctx), {N(I), V(I, 2)}, {V(D, 10), V(D, 12)});
  }
  check_batch(Filter(Datum(batch), Datum(DevBool(pool, {0, 1, 1, -1})), drop, ctx), {V(I, 1), V(I, 2)}, {V(D, 11), V(D, 12)});
  check_batch(Filter(Datum(batch), Datum(DevBool(pool, {0, 1, 1, -1})), emit, ctx), {V(I, 1), V(I, 2), N(I)}, {V(D, 11), V(D, 12), N(D)});
  CHECK(Filter(Datum(batch), Datum(DevBool(pool, {0, 1, 1})), drop, ctx).status().IsInvalid());  // :1780-1782
  CHECK(Filter(Datum(batch), Datum(a), drop, ctx).status().IsNotImplemented());                  // :1885-1887

  // TakeRecordBatch with int8 and uint32 indices
  for (int pass = 0; pass < 2; ++pass) {
    auto idx = [&](const std::vector<int>& v) -> Datum {
      if (pass == 0) { std::vector<Opt<int8_t>> o; for (int x : v) o.push_back(V(int8_t, x)); return Datum(Dev<int8_t>(pool, int8(), o)); }
      std::vector<Opt<uint32_t>> o; for (int x : v) o.push_back(V(uint32_t, x)); return Datum(Dev<uint32_t>(pool, uint32(), o));
    };
    TakeOptions to;
    check_batch(Take(Datum(batch), idx({}), to, ctx), {}, {});
    check_batch(Take(Datum(batch), idx({3, 1, 3, 1, 3}), to, ctx), {V(I, 4), V(I, 1), V(I, 4), V(I, 1), V(I, 4)},
                {V(D, 13), V(D, 11), V(D, 13), V(D, 11), V(D, 13)});
    check_batch(Take(Datum(batch), idx({3, 1, 0}), to, ctx), {V(I, 4), V(I, 1), N(I)}, {V(D, 13), V(D, 11), V(D, 10)});
    check_batch(Take(Datum(batch), idx({0, 2, 2, 2, 2, 2, 2}), to, ctx), {N(I), V(I, 2), V(I, 2), V(I, 2), V(I, 2), V(I, 2), V(I, 2)},
                {V(D, 10), V(D, 12), V(D, 12), V(D, 12), V(D, 12), V(D, 12), V(D, 12)});
  }

  // the same vectors with the reference's own utf8 column "b" (vector_selection_test.cc:660-689, 1460-1493)
  {
    using SV = std::vector<std::pair<std::string, bool>>;
    auto sschm = schema({field("a", int32()), field("b", utf8())});
    auto sb = BinaryFromHost(pool, utf8(), {"yo", "", "hello", "eh"});
    CHECK_OK(sb.status());
    if (sb.ok()) {
      auto sbatch = RecordBatch::Make(sschm, 4, {a, *sb});
      auto strs = [&](const Result<Datum>& r) -> SV {
        CHECK_OK(r.status());
        if (!r.ok()) return {};
        auto h = BinaryToHost(*r->record_batch()->column(1), pool);
        CHECK_OK(h.status());
        return h.ok() ? *h : SV{};
      };
      CHECK((strs(Filter(Datum(sbatch), Datum(DevBool(pool, {1, 0, 1, 0})), drop, ctx)) == SV{{"yo", true}, {"hello", true}}));
      CHECK((strs(Filter(Datum(sbatch), Datum(DevBool(pool, {0, 1, 1, -1})), drop, ctx)) == SV{{"", true}, {"hello", true}}));
      CHECK((strs(Filter(Datum(sbatch), Datum(DevBool(pool, {0, 1, 1, -1})), emit, ctx)) == SV{{"", true}, {"hello", true}, {"", false}}));
      CHECK((strs(Filter(Datum(sbatch), Datum(DevBool(pool, {0, 0, 0, 0})), drop, ctx)) == SV{}));
      TakeOptions to;
      auto idx = Dev<int8_t>(pool, int8(), {V(int8_t, 3), V(int8_t, 1), V(int8_t, 0)});
      auto taken = Take(Datum(sbatch), Datum(idx), to, ctx);
      CHECK((strs(taken) == SV{{"eh", true}, {"", true}, {"yo", true}}));
      if (taken.ok()) CHECK(Same(Host<I>(pool, *taken->record_batch()->column(0)), {V(I, 4), V(I, 1), N(I)}));
      // arrays: TestFilterKernelWithString :445-452, TestTakeKernelWithString :1183-1190, null values and null indices
      auto abc = BinaryFromHost(pool, utf8(), {"a", "b", "c"}), nbc = BinaryFromHost(pool, utf8(), {"", "b", "c"}, {false, true, true});
      CHECK_OK(abc.status());
      CHECK_OK(nbc.status());
      if (abc.ok() && nbc.ok()) {
        auto one = [&](const Result<Datum>& r) -> SV { CHECK_OK(r.status()); if (!r.ok()) return {}; auto h = BinaryToHost(*r->array(), pool); return h.ok() ? *h : SV{}; };
        CHECK((one(Filter(Datum(*abc), Datum(DevBool(pool, {0, 1, 0})), drop, ctx)) == SV{{"b", true}}));
        CHECK((one(Filter(Datum(*nbc), Datum(DevBool(pool, {1, 1, 0})), drop, ctx)) == SV{{"", false}, {"b", true}}));
        auto i8n = Dev<int8_t>(pool, int8(), {V(int8_t, 0), N(int8_t), V(int8_t, 0)});
        CHECK((one(Take(Datum(*abc), Datum(i8n), to, ctx)) == SV{{"a", true}, {"", false}, {"a", true}}));
        CHECK((one(Take(Datum(*nbc), Datum(i8n), to, ctx)) == SV{{"", false}, {"", false}, {"", false}}));
        auto bad = Dev<int8_t>(pool, int8(), {V(int8_t, 0), V(int8_t, 5)});
        CHECK(Take(Datum(*abc), Datum(bad), to, ctx).status().IsIndexError());
        // chunked utf8 values are concatenated first (TakeCA :1923-1943)
        auto chunked_s = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{*abc, *nbc}, utf8());
        auto i6 = Dev<int8_t>(pool, int8(), {V(int8_t, 5), V(int8_t, 0), V(int8_t, 3), V(int8_t, 4)});
        auto tc = Take(Datum(chunked_s), Datum(i6), to, ctx);
        CHECK_OK(tc.status());
        if (tc.ok()) {
          auto h = BinaryToHost(*tc->chunked_array()->chunk(0), pool);
          CHECK(h.ok() && (*h == SV{{"c", true}, {"a", true}, {"", false}, {"b", true}}));
        }
        auto f = Filter(Datum(*abc), Datum(DevBool(pool, {0, 1, 1})), drop, ctx);
        if (f.ok()) CHECK(f->array()->buffers[0] == nullptr && f->array()->null_count == 0);  // BinaryFilterNonNullImpl: no bitmap
      }
    }
  }

  // TakeChunkedArray (:1536-1550)
  using B = int8_t;
  auto i8 = [&](const std::vector<int>& v) { std::vector<Opt<B>> o; for (int x : v) o.push_back(V(B, x)); return Dev<B>(pool, int8(), o); };
  auto chunked = [&](std::vector<std::shared_ptr<ArrayData>> c, std::shared_ptr<DataType> t) { return std::make_shared<ChunkedArray>(std::move(c), t); };
  auto flat = [&](const ChunkedArray& c) { std::vector<std::vector<int>> out; for (const auto& ch : c.chunks()) { std::vector<int> v; for (const auto& o : Host<B>(pool, *ch)) v.push_back(o.valid ? o.v : -99); out.push_back(v); } return out; };
  {
    TakeOptions to;
    auto vals = chunked({i8({7}), i8({8, 9})}, int8());
    auto r = Take(Datum(vals), Datum(i8({0, 1, 0, 2})), to, ctx);
    CHECK_OK(r.status());
    if (r.ok()) { CHECK(r->kind() == Datum::CHUNKED_ARRAY); CHECK((flat(*r->chunked_array()) == std::vector<std::vector<int>>{{7, 8, 7, 9}})); }
    r = Take(Datum(vals), Datum(chunked({i8({0, 1, 0}), i8({}), i8({2})}, int8())), to, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK((flat(*r->chunked_array()) == std::vector<std::vector<int>>{{7, 8, 7}, {}, {9}}));
    r = Take(Datum(vals), Datum(i8({2, 1})), to, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK((flat(*r->chunked_array()) == std::vector<std::vector<int>>{{9, 8}}));
    CHECK(Take(Datum(vals), Datum(i8({0, 5})), to, ctx).status().IsIndexError());
    CHECK(Take(Datum(vals), Datum(chunked({i8({0, 1, 0}), i8({5, 1})}, int8())), to, ctx).status().IsIndexError());
    r = Take(Datum(chunked({i8({})}, int8())), Datum(i8({})), to, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(r->chunked_array()->length() == 0);
  }

  // tables: columns chunked as {[null, 1], [2, 4]} (:1596-1609, :737-796)
  auto tab_cols = [&]() {
    return std::vector<std::shared_ptr<ChunkedArray>>{
        chunked({Dev<I>(pool, int32(), {N(I), V(I, 1)}), Dev<I>(pool, int32(), {V(I, 2), V(I, 4)})}, int32()),
        chunked({Dev<D>(pool, float64(), {V(D, 10), V(D, 11)}), Dev<D>(pool, float64(), {V(D, 12), V(D, 13)})}, float64())};
  };
  auto table = Table::Make(schm, tab_cols());
  CHECK(table->num_rows() == 4);
  auto check_table = [&](const Result<Datum>& r, const std::vector<std::vector<Opt<I>>>& wa, const std::vector<std::vector<Opt<D>>>& wb) {
    CHECK_OK(r.status());
    if (!r.ok()) return;
    CHECK(r->kind() == Datum::TABLE);
    const auto& out = *r->table();
    int64_t rows = 0;
    for (const auto& c : wa) rows += static_cast<int64_t>(c.size());
    CHECK(out.num_rows() == rows && out.num_columns() == 2);
    CHECK(out.column(0)->num_chunks() == static_cast<int>(wa.size()) && out.column(1)->num_chunks() == static_cast<int>(wb.size()));
    if (out.column(0)->num_chunks() != static_cast<int>(wa.size())) return;
    for (size_t c = 0; c < wa.size(); ++c) {
      CHECK(Same(Host<I>(pool, *out.column(0)->chunk(static_cast<int>(c))), wa[c]));
      CHECK(Same(Host<D>(pool, *out.column(1)->chunk(static_cast<int>(c))), wb[c]));
    }
  };
  {
    TakeOptions to;
    check_table(Take(Datum(table), Datum(i8({})), to, ctx), {{}}, {{}});
    check_table(Take(Datum(table), Datum(i8({3, 1, 0})), to, ctx), {{V(I, 4), V(I, 1), N(I)}}, {{V(D, 13), V(D, 11), V(D, 10)}});
    check_table(Take(Datum(table), Datum(chunked({i8({0, 1}), i8({2, 3})}, int8())), to, ctx), {{N(I), V(I, 1)}, {V(I, 2), V(I, 4)}},
                {{V(D, 10), V(D, 11)}, {V(D, 12), V(D, 13)}});
    // FilterTable: filter as one array and chunked differently from the table; chunks that select nothing vanish
    check_table(Filter(Datum(table), Datum(DevBool(pool, {0, 1, 1, -1})), drop, ctx), {{V(I, 1)}, {V(I, 2)}}, {{V(D, 11)}, {V(D, 12)}});
    check_table(Filter(Datum(table), Datum(DevBool(pool, {0, 1, 1, -1})), emit, ctx), {{V(I, 1)}, {V(I, 2), N(I)}}, {{V(D, 11)}, {V(D, 12), N(D)}});
    check_table(Filter(Datum(table), Datum(chunked({DevBool(pool, {0, 1, 1}), DevBool(pool, {-1})}, boolean())), emit, ctx),
                {{V(I, 1)}, {V(I, 2)}, {N(I)}}, {{V(D, 11)}, {V(D, 12)}, {N(D)}});
    check_table(Filter(Datum(table), Datum(DevBool(pool, {0, 0, 1, 1})), drop, ctx), {{V(I, 2), V(I, 4)}}, {{V(D, 12), V(D, 13)}});
    check_table(Filter(Datum(table), Datum(DevBool(pool, {0, 0, 0, 0})), drop, ctx), {}, {});
    CHECK(Filter(Datum(table), Datum(DevBool(pool, {0, 0, 0})), drop, ctx).status().IsInvalid());
  }

  // RechunkArraysConsistently (array_test.cc:2549-2580): {3,2,4} against {2,3,2,2} -> {2,1,2,2,2}
  {
    auto mk = [&](int n, int from) { std::vector<Opt<I>> o; for (int k = 0; k < n; ++k) o.push_back(V(I, from + k)); return Dev<I>(pool, int32(), o); };
    auto out = internal::RechunkArraysConsistently({{mk(3, 1), mk(2, 4), mk(4, 6)}, {mk(2, 41), mk(3, 43), mk(2, 46), mk(2, 48)}});
    CHECK(out.size() == 2 && out[0].size() == 5 && out[1].size() == 5);
    const int want[] = {2, 1, 2, 2, 2};
    for (size_t c = 0; c < 5 && out[0].size() == 5; ++c) CHECK(out[0][c]->length == want[c] && out[1][c]->length == want[c]);
    if (out[0].size() == 5) CHECK(Same(Host<I>(pool, *out[0][1]), {V(I, 3)}) && Same(Host<I>(pool, *out[1][4]), {V(I, 48), V(I, 49)}));
    auto empty = internal::RechunkArraysConsistently({{mk(0, 0), mk(0, 0)}, {}, {mk(0, 0)}});  // :2527-2547
    CHECK(empty.size() == 3);
  }

  // Concatenate (concatenate_test.cc): validity only when some input has nulls
  {
    auto c = Concatenate({Dev<I>(pool, int32(), {V(I, 1), N(I)}), Dev<I>(pool, int32(), {}), Dev<I>(pool, int32(), {V(I, 3)})}, pool);
    CHECK_OK(c.status());
    if (c.ok()) { CHECK(Same(Host<I>(pool, **c), {V(I, 1), N(I), V(I, 3)})); CHECK((*c)->null_count == 1 && (*c)->buffers[0]); }
    c = Concatenate({Dev<I>(pool, int32(), {V(I, 1)}), Dev<I>(pool, int32(), {V(I, 3)})}, pool);
    CHECK_OK(c.status());
    if (c.ok()) CHECK((*c)->null_count == 0 && !(*c)->buffers[0] && (*c)->length == 2);
    auto cb = Concatenate({DevBool(pool, {1, 0, -1}), DevBool(pool, {1, 1, 1, 0, 1, 0, 0, 1, 1})}, pool);
    CHECK_OK(cb.status());
    if (cb.ok()) CHECK((HostBool(pool, **cb) == std::vector<int>{1, 0, -1, 1, 1, 1, 0, 1, 0, 0, 1, 1}));
    CHECK(Concatenate({}, pool).status().IsInvalid());
    CHECK(Concatenate({Dev<I>(pool, int32(), {V(I, 1)}), DevBool(pool, {1})}, pool).status().IsInvalid());
  }

  // unique / value_counts / dictionary_encode (vector_hash_test.cc:162-212)
  {
    auto vals = Dev<I>(pool, int32(), {V(I, 2), N(I), V(I, 2), V(I, 1), V(I, 2), V(I, 3), N(I)});
    auto u = Unique(Datum(vals), ctx);
    CHECK_OK(u.status());
    if (u.ok()) CHECK(Same(Host<I>(pool, *u->array()), {V(I, 2), N(I), V(I, 1), V(I, 3)}));
    auto vc = ValueCounts(Datum(vals), ctx);
    CHECK_OK(vc.status());
    if (vc.ok()) {
      const ArrayData& st = *vc->array();
      CHECK(st.type->id() == Type::STRUCT && st.child_names == std::vector<std::string>({"values", "counts"}) && st.length == 4);
      CHECK(Same(Host<I>(pool, *st.child_data[0]), {V(I, 2), N(I), V(I, 1), V(I, 3)}));
      CHECK(Same(Host<int64_t>(pool, *st.child_data[1]), {V(int64_t, 3), V(int64_t, 2), V(int64_t, 1), V(int64_t, 1)}));
    }
    auto six = Dev<I>(pool, int32(), {V(I, 2), N(I), V(I, 2), V(I, 1), V(I, 2), V(I, 3)});
    auto de = DictionaryEncode(Datum(six), DictionaryEncodeOptions::Defaults(), ctx);
    CHECK_OK(de.status());
    if (de.ok()) {
      const ArrayData& d = *de->array();
      CHECK(d.type->id() == Type::DICTIONARY && d.dictionary && d.dictionary->type->id() == Type::INT32);
      CHECK(Same(Host<I>(pool, *d.dictionary), {V(I, 2), V(I, 1), V(I, 3)}) && d.dictionary->buffers[0] == nullptr);
      auto idx = ArrayData::Make(int32(), d.length, d.buffers, d.null_count, d.offset);
      CHECK(Same(Host<I>(pool, *idx), {V(I, 0), N(I), V(I, 0), V(I, 1), V(I, 0), V(I, 2)}));
    }
    de = DictionaryEncode(Datum(six), DictionaryEncodeOptions(DictionaryEncodeOptions::ENCODE), ctx);  // api_vector.h:315
    CHECK_OK(de.status());
    if (de.ok()) {
      const ArrayData& d = *de->array();
      CHECK(Same(Host<I>(pool, *d.dictionary), {V(I, 2), N(I), V(I, 1), V(I, 3)}));
      auto idx = ArrayData::Make(int32(), d.length, d.buffers, d.null_count, d.offset);
      CHECK(Same(Host<I>(pool, *idx), {V(I, 0), V(I, 1), V(I, 0), V(I, 2), V(I, 0), V(I, 3)}));
    }
    // sliced input (:170-172, :197-200) and chunks sharing one memo table
    auto seven = Dev<I>(pool, int32(), {V(I, 2), V(I, 1), N(I), V(I, 4), V(I, 3), V(I, 1), V(I, 42)});
    de = DictionaryEncode(Datum(seven->Slice(1, 5)), DictionaryEncodeOptions::Defaults(), ctx);
    CHECK_OK(de.status());
    if (de.ok()) {
      const ArrayData& d = *de->array();
      CHECK(Same(Host<I>(pool, *d.dictionary), {V(I, 1), V(I, 4), V(I, 3)}));
      CHECK(Same(Host<I>(pool, *ArrayData::Make(int32(), d.length, d.buffers, d.null_count, d.offset)), {V(I, 0), N(I), V(I, 1), V(I, 2), V(I, 0)}));
    }
    auto two = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{Dev<I>(pool, int32(), {V(I, 5), V(I, 7)}),
                                                                                      Dev<I>(pool, int32(), {V(I, 7), N(I), V(I, 9), V(I, 5)})}, int32());
    u = Unique(Datum(two), ctx);
    CHECK_OK(u.status());
    if (u.ok()) CHECK(Same(Host<I>(pool, *u->array()), {V(I, 5), V(I, 7), N(I), V(I, 9)}));
    de = DictionaryEncode(Datum(two), DictionaryEncodeOptions::Defaults(), ctx);
    CHECK_OK(de.status());
    if (de.ok()) {
      CHECK(de->kind() == Datum::CHUNKED_ARRAY && de->chunked_array()->num_chunks() == 2);
      const ArrayData& c1 = *de->chunked_array()->chunk(1);
      CHECK(Same(Host<I>(pool, *c1.dictionary), {V(I, 5), V(I, 7), V(I, 9)}));
      CHECK(Same(Host<I>(pool, *ArrayData::Make(int32(), c1.length, c1.buffers, c1.null_count, c1.offset)), {V(I, 1), N(I), V(I, 2), V(I, 0)}));
    }
    auto zero = DictionaryEncode(Datum(std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{}, int32())),
                                 DictionaryEncodeOptions::Defaults(), ctx);  // ZeroChunks :203-212
    CHECK_OK(zero.status());
    if (zero.ok()) CHECK(zero->kind() == Datum::CHUNKED_ARRAY && zero->chunked_array()->num_chunks() == 0);
  }

  // sort_indices over batches and tables (vector_sort_test.cc:763-980)
  {
    using U8 = uint8_t; using U32 = uint32_t; using F = float;
    const F nanf = std::numeric_limits<F>::quiet_NaN();
    const D nan = std::numeric_limits<D>::quiet_NaN();
    SortOptions ad({SortKey("a", SortOrder::Ascending), SortKey("b", SortOrder::Descending)});
    auto s8 = schema({field("a", uint8()), field("b", uint32())});
    auto sf = schema({field("a", float32()), field("b", float64())});
    auto sorted = [&](const Datum& in, const SortOptions& o) { auto r = SortIndices(in, o, ctx); CHECK_OK(r.status()); return r.ok() ? HostU64(pool, *r->array()) : std::vector<uint64_t>{}; };
    auto b1 = RecordBatch::Make(s8, 7, {Dev<U8>(pool, uint8(), {V(U8, 3), V(U8, 1), V(U8, 3), V(U8, 0), V(U8, 2), V(U8, 1), V(U8, 1)}),
                                        Dev<U32>(pool, uint32(), {V(U32, 5), V(U32, 3), V(U32, 4), V(U32, 6), V(U32, 5), V(U32, 5), V(U32, 3)})});
    CHECK((sorted(Datum(b1), ad) == std::vector<uint64_t>{3, 5, 1, 6, 4, 0, 2}));  // NoNull :763-781
    auto b2 = RecordBatch::Make(s8, 6, {Dev<U8>(pool, uint8(), {N(U8), V(U8, 1), V(U8, 3), N(U8), V(U8, 2), V(U8, 1)}),
                                        Dev<U32>(pool, uint32(), {V(U32, 5), V(U32, 3), N(U32), N(U32), V(U32, 5), V(U32, 5)})});
    CHECK((sorted(Datum(b2), ad) == std::vector<uint64_t>{5, 1, 4, 2, 0, 3}));  // Null :783-800
    auto t2 = Table::Make(s8, {chunked({Dev<U8>(pool, uint8(), {N(U8), V(U8, 1), V(U8, 3)}), Dev<U8>(pool, uint8(), {N(U8), V(U8, 2), V(U8, 1)})}, uint8()),
                               chunked({Dev<U32>(pool, uint32(), {V(U32, 5), V(U32, 3), N(U32)}), Dev<U32>(pool, uint32(), {N(U32), V(U32, 5), V(U32, 5)})}, uint32())});
    CHECK((sorted(Datum(t2), ad) == std::vector<uint64_t>{5, 1, 4, 2, 0, 3}));  // table, two chunks :904-914
    auto b3 = RecordBatch::Make(sf, 8, {Dev<F>(pool, float32(), {V(F, 3), V(F, 1), V(F, 3), V(F, 0), V(F, nanf), V(F, nanf), V(F, nanf), V(F, 1)}),
                                        Dev<D>(pool, float64(), {V(D, 5), V(D, nan), V(D, 4), V(D, 6), V(D, 5), V(D, nan), V(D, 5), V(D, 5)})});
    CHECK((sorted(Datum(b3), ad) == std::vector<uint64_t>{3, 7, 1, 0, 2, 4, 6, 5}));  // NaN :802-821
    auto t4 = Table::Make(sf, {chunked({Dev<F>(pool, float32(), {N(F), V(F, 1), V(F, 3), N(F)}), Dev<F>(pool, float32(), {V(F, nanf), V(F, nanf), V(F, nanf), V(F, 1)})}, float32()),
                               chunked({Dev<D>(pool, float64(), {V(D, 5), V(D, 3), N(D), N(D)}), Dev<D>(pool, float64(), {N(D), V(D, nan), V(D, 5), V(D, 5)})}, float64())});
    CHECK((sorted(Datum(t4), ad) == std::vector<uint64_t>{7, 1, 2, 6, 5, 4, 0, 3}));  // NaNAndNull, two chunks :967-980
    // one key: delegates to the array / chunked-array sorters (:1638-1645, :1676-1683)
    CHECK((sorted(Datum(b1), SortOptions({SortKey("b", SortOrder::Descending)})) == std::vector<uint64_t>{3, 0, 4, 5, 2, 1, 6}));
    CHECK((sorted(Datum(t2), SortOptions({SortKey("a", SortOrder::Ascending)})) == std::vector<uint64_t>{1, 5, 4, 2, 0, 3}));
    // partition_nth_indices: the property the reference validates (vector_sort_test.cc:132-150, :198-260)
    {
      const std::vector<Opt<D>> vals = {N(D), V(D, 1), V(D, 3.3), N(D), V(D, 2), V(D, 5.3), V(D, nan), V(D, -7), V(D, 2)};
      auto arr = Dev<D>(pool, float64(), vals);
      for (int64_t n = 0; n <= static_cast<int64_t>(vals.size()); ++n) {
        auto r = NthToIndices(Datum(arr), n, ctx);
        CHECK_OK(r.status());
        if (!r.ok()) continue;
        CHECK(r->array()->null_count == 0);
        const auto idx = HostU64(pool, *r->array());
        std::vector<int> seen(vals.size(), 0);
        for (uint64_t i : idx) if (i < vals.size()) ++seen[i];
        CHECK(idx.size() == vals.size() && std::count(seen.begin(), seen.end(), 1) == static_cast<long>(vals.size()));
        if (idx.size() != vals.size()) continue;
        if (n == static_cast<int64_t>(vals.size())) { for (size_t i = 0; i < idx.size(); ++i) CHECK(idx[i] == i); continue; }
        // NthComparator: nulls / NaNs compare as the largest
        auto less_eq = [&](uint64_t l, uint64_t r) {
          const bool ln = !vals[l].valid || vals[l].v != vals[l].v, rn = !vals[r].valid || vals[r].v != vals[r].v;
          if (rn) return true;
          if (ln) return false;
          return vals[l].v <= vals[r].v;
        };
        for (int64_t i = 0; i < n; ++i) CHECK(less_eq(idx[i], idx[n]));
        for (size_t i = n + 1; i < idx.size(); ++i) CHECK(less_eq(idx[n], idx[i]));
      }
      CHECK(NthToIndices(Datum(arr), 10, ctx).status().IsIndexError());
      CHECK(CallFunction("partition_nth_indices", {Datum(arr)}, ctx).status().IsInvalid());  // no default options (:198-201)
    }
    CHECK(SortIndices(Datum(b1), SortOptions(), ctx).status().IsInvalid());                                      // :1634-1636
    CHECK(SortIndices(Datum(b1), SortOptions({SortKey("nope")}), ctx).status().IsInvalid());                     // :1639-1642
    CHECK(SortIndices(Datum(t2), SortOptions({SortKey("a"), SortKey("zz")}), ctx).status().IsInvalid());
    CHECK(SortIndices(Datum(t2), SortOptions({SortKey("a"), SortKey("zz")}), ctx).status().message() == "Nonexistent sort key column: zz");
  }
}

// AddWithImplicitCasts / ...Uint64EdgeCase (scalar_arithmetic_test.cc:897-933), GreaterWithImplicitCasts
// (scalar_compare_test.cc:494-520) and explicit casts (cast_test.cc: safe / unsafe integer and floating cases).
static void PromotionTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  auto i32 = Dev<int32_t>(pool, int32(), {V(int32_t, 0), V(int32_t, 1), V(int32_t, 2), N(int32_t)});
  auto f64 = Dev<double>(pool, float64(), {V(double, 0.25), V(double, 0.5), V(double, 0.75), V(double, 1.0)});
  auto r = CallFunction("add", {Datum(i32), Datum(f64)}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(r->type()->id() == Type::DOUBLE && Same(Host<double>(pool, *r->array()), {V(double, 0.25), V(double, 1.5), V(double, 2.75), N(double)}));
  auto i8 = Dev<int8_t>(pool, int8(), {V(int8_t, -16), V(int8_t, 0), V(int8_t, 16), N(int8_t)});
  auto u32 = Dev<uint32_t>(pool, uint32(), {V(uint32_t, 3), V(uint32_t, 4), V(uint32_t, 5), V(uint32_t, 7)});
  r = CallFunction("add", {Datum(i8), Datum(u32)}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(r->type()->id() == Type::INT64 && Same(Host<int64_t>(pool, *r->array()), {V(int64_t, -13), V(int64_t, 4), V(int64_t, 21), N(int64_t)}));
  {  // dictionary(int32, int32) [8, 6, 3, null, 2] + uint32 -> int64 [11, 10, 8, null, 2]
    auto values = Dev<int32_t>(pool, int32(), {V(int32_t, 8), V(int32_t, 6), V(int32_t, 3), N(int32_t), V(int32_t, 2)});
    auto enc = DictionaryEncode(Datum(values), DictionaryEncodeOptions(), ctx);
    CHECK_OK(enc.status());
    auto rhs = Dev<uint32_t>(pool, uint32(), {V(uint32_t, 3), V(uint32_t, 4), V(uint32_t, 5), V(uint32_t, 7), V(uint32_t, 0)});
    if (enc.ok()) {
      r = CallFunction("add", {*enc, Datum(rhs)}, ctx);
      CHECK_OK(r.status());
      if (r.ok()) CHECK(r->type()->id() == Type::INT64 && Same(Host<int64_t>(pool, *r->array()), {V(int64_t, 11), V(int64_t, 10), V(int64_t, 8), N(int64_t), V(int64_t, 2)}));
      r = CallFunction("greater", {*enc, Datum(rhs)}, ctx);  // :507-510 shape
      CHECK_OK(r.status());
      if (r.ok()) CHECK(HostBool(pool, *r->array()) == (std::vector<int>{1, 1, 0, -1, 1}));
    }
  }
  {  // int32 + NullArray(4) -> int32 [null, null, null, null]
    auto nulls = ArrayData::Make(null(), 4, {nullptr}, 4);
    r = CallFunction("add", {Datum(i32), Datum(nulls)}, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(r->type()->id() == Type::INT32 && Same(Host<int32_t>(pool, *r->array()), {N(int32_t), N(int32_t), N(int32_t), N(int32_t)}));
    r = CallFunction("greater", {Datum(i32), Datum(nulls)}, ctx);  // :512-514
    CHECK_OK(r.status());
    if (r.ok()) CHECK(HostBool(pool, *r->array()) == (std::vector<int>{-1, -1, -1, -1}));
  }
  {  // the uint64 edge: int64 is as wide as the promotion goes
    auto m1 = Dev<int8_t>(pool, int8(), {V(int8_t, -1)});
    auto z = Dev<uint64_t>(pool, uint64(), {V(uint64_t, 0)});
    r = CallFunction("add", {Datum(m1), Datum(z)}, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(r->type()->id() == Type::INT64 && Same(Host<int64_t>(pool, *r->array()), {V(int64_t, -1)}));
    auto m64 = Dev<int64_t>(pool, int64(), {V(int64_t, -1)});
    auto big = Dev<uint64_t>(pool, uint64(), {V(uint64_t, 18446744073709551615ull)});
    r = CallFunction("add", {Datum(m64), Datum(big)}, ctx);
    CHECK(r.status().IsInvalid());
    CHECK(r.status().message() == "Integer value 18446744073709551615 not in range: 0 to 9223372036854775807");
  }
  // greater with implicit casts
  r = CallFunction("greater", {Datum(i32), Datum(Dev<double>(pool, float64(), {V(double, 0.5), V(double, 1.0), V(double, 1.5), V(double, 2.0)}))}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(HostBool(pool, *r->array()) == (std::vector<int>{0, 0, 1, -1}));
  r = CallFunction("greater", {Datum(i8), Datum(u32)}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(HostBool(pool, *r->array()) == (std::vector<int>{0, 0, 1, -1}));
  r = CallFunction("greater", {Datum(i8), Datum(Dev<uint8_t>(pool, uint8(), {V(uint8_t, 255), V(uint8_t, 254), V(uint8_t, 1), V(uint8_t, 0)}))}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(HostBool(pool, *r->array()) == (std::vector<int>{0, 0, 1, -1}));
  // an array against a scalar of another type: both are cast (the scalar on the host)
  r = CallFunction("add", {Datum(i8), Datum(Scalar::Make<int32_t>(int32(), 1000))}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(r->type()->id() == Type::INT32 && Same(Host<int32_t>(pool, *r->array()), {V(int32_t, 984), V(int32_t, 1000), V(int32_t, 1016), N(int32_t)}));
  // sliced operands keep their validity
  r = CallFunction("multiply", {Datum(i8->Slice(1, 3)), Datum(f64->Slice(0, 3))}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<double>(pool, *r->array()), {V(double, 0.0), V(double, 8.0), N(double)}));
  // negate / power (scalar_arithmetic_test.cc:479-592, 708-861 literals)
  r = CallFunction("negate", {Datum(i8)}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<int8_t>(pool, *r->array()), {V(int8_t, 16), V(int8_t, 0), V(int8_t, -16), N(int8_t)}));
  r = CallFunction("negate", {Datum(u32)}, ctx);  // unsigned: ~x + 1
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<uint32_t>(pool, *r->array()), {V(uint32_t, 4294967293u), V(uint32_t, 4294967292u), V(uint32_t, 4294967291u), V(uint32_t, 4294967289u)}));
  auto mins = Dev<int8_t>(pool, int8(), {V(int8_t, -128), V(int8_t, 127)});
  r = CallFunction("negate", {Datum(mins)}, ctx);  // wraps
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<int8_t>(pool, *r->array()), {V(int8_t, -128), V(int8_t, -127)}));
  r = CallFunction("negate_checked", {Datum(mins)}, ctx);
  CHECK(r.status().IsInvalid() && r.status().message() == "overflow");
  CHECK(CallFunction("negate_checked", {Datum(u32)}, ctx).status().IsNotImplemented());
  auto base = Dev<int32_t>(pool, int32(), {V(int32_t, 2), V(int32_t, -3), N(int32_t), V(int32_t, 7)});
  auto expo = Dev<int32_t>(pool, int32(), {V(int32_t, 10), V(int32_t, 3), V(int32_t, 2), V(int32_t, 0)});
  r = CallFunction("power", {Datum(base), Datum(expo)}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<int32_t>(pool, *r->array()), {V(int32_t, 1024), V(int32_t, -27), N(int32_t), V(int32_t, 1)}));
  r = CallFunction("power", {Datum(base), Datum(Scalar::Make<int32_t>(int32(), -1))}, ctx);
  CHECK(r.status().IsInvalid() && r.status().message() == "integers to negative integer powers are not allowed");
  r = CallFunction("power_checked", {Datum(base), Datum(Scalar::Make<int32_t>(int32(), 40))}, ctx);
  CHECK(r.status().IsInvalid() && r.status().message() == "overflow");
  r = CallFunction("power", {Datum(base), Datum(Scalar::Make<int32_t>(int32(), 40))}, ctx);  // wraps in uint64
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<int32_t>(pool, *r->array()), {V(int32_t, 0), V(int32_t, int32_t(12157665459056928801ull)), N(int32_t), V(int32_t, int32_t(12233820880559894977ull))}));
  r = CallFunction("power", {Datum(f64), Datum(Scalar::Make<double>(float64(), 2.0))}, ctx);
  CHECK_OK(r.status());
  if (r.ok()) {  // pow(): within 2 ulp of the host libm (include/amdarrow.h)
    auto got = Host<double>(pool, *r->array());
    const double want[4] = {0.0625, 0.25, 0.5625, 1.0};
    for (int i = 0; i < 4; ++i) CHECK(got[i].valid && std::fabs(got[i].v - want[i]) <= want[i] * 4.5e-16);
  }
  r = CallFunction("power", {Datum(i8), Datum(f64)}, ctx);  // promoted to float64 first
  CHECK(r.ok() && r->type()->id() == Type::DOUBLE);
  {  // scalar (x) scalar -> scalar (ScalarBinary's fourth shape, codegen_internal.h:758-824; exec.cc:553-716)
    auto three = Datum(Scalar::Make<int32_t>(int32(), 3)), four = Datum(Scalar::Make<int32_t>(int32(), 4));
    r = CallFunction("add", {three, four}, ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(r->is_scalar() && r->scalar()->is_valid && r->type()->id() == Type::INT32 && static_cast<int32_t>(r->scalar()->bits) == 7);
    r = CallFunction("greater", {four, three}, ctx);
    CHECK(r.ok() && r->is_scalar() && r->type()->id() == Type::BOOL && r->scalar()->is_valid && r->scalar()->bits == 1);
    r = CallFunction("multiply", {three, Datum(Scalar::MakeNull(int32()))}, ctx);
    CHECK(r.ok() && r->is_scalar() && !r->scalar()->is_valid);
    r = CallFunction("add", {Datum(Scalar::Make<int8_t>(int8(), -5)), four}, ctx);  // promoted to int32
    CHECK(r.ok() && r->is_scalar() && r->type()->id() == Type::INT32 && static_cast<int32_t>(r->scalar()->bits) == -1);
    r = CallFunction("divide", {three, Datum(Scalar::Make<int32_t>(int32(), 0))}, ctx);
    CHECK(r.status().IsInvalid() && r.status().message() == "divide by zero");
    r = CallFunction("is_null", {Datum(Scalar::MakeNull(float64()))}, ctx);
    CHECK(r.ok() && r->is_scalar() && r->scalar()->is_valid && r->scalar()->bits == 1);
    r = CallFunction("negate", {three}, ctx);
    CHECK(r.ok() && r->is_scalar() && static_cast<int32_t>(r->scalar()->bits) == -3);
  }
  // explicit casts
  auto wide = Dev<int64_t>(pool, int64(), {V(int64_t, 1), V(int64_t, -2), N(int64_t), V(int64_t, 1099511627776ll)});
  r = Cast(Datum(wide), int32(), CastOptions::Safe(), ctx);
  CHECK(r.status().IsInvalid() && r.status().message() == "Integer value 1099511627776 not in range: -2147483648 to 2147483647");
  r = Cast(Datum(wide), int32(), CastOptions::Unsafe(), ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<int32_t>(pool, *r->array()), {V(int32_t, 1), V(int32_t, -2), N(int32_t), V(int32_t, 0)}));
  r = Cast(Datum(wide->Slice(0, 3)), int8(), CastOptions::Safe(), ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<int8_t>(pool, *r->array()), {V(int8_t, 1), V(int8_t, -2), N(int8_t)}));
  CastOptions to_f32 = CastOptions::Safe(float32());
  r = CallFunction("cast", {Datum(wide)}, &to_f32, ctx);
  CHECK(r.status().IsInvalid() && r.status().message() == "Integer value 1099511627776 not in range: -16777216 to 16777216");
  r = Cast(Datum(wide), float64(), CastOptions::Safe(), ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(Same(Host<double>(pool, *r->array()), {V(double, 1), V(double, -2), N(double), V(double, 1099511627776.0)}));
  CHECK(Cast(Datum(f64), int32(), CastOptions::Safe(), ctx).status().IsNotImplemented());
  CHECK(CallFunction("cast", {Datum(wide)}, ctx).status().IsInvalid());  // no to_type
  auto chunked = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{i8, i8->Slice(2, 2)}, int8());
  r = Cast(Datum(chunked), int64(), CastOptions::Safe(), ctx);
  CHECK_OK(r.status());
  if (r.ok()) CHECK(r->kind() == Datum::CHUNKED_ARRAY && r->chunked_array()->num_chunks() == 2 &&
                    Same(Host<int64_t>(pool, *r->chunked_array()->chunk(1)), {V(int64_t, 16), N(int64_t)}));
}

// Grouper.MakeGroupings (hash_aggregate_test.cc:423-452): the literals, ApplyGroupings on the ids, null ids.
static void GroupingsTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  struct Case { std::vector<uint32_t> ids; std::vector<std::vector<int32_t>> want; };
  const std::vector<Case> cases = {{{}, {{}}},
                                   {{0, 0, 0}, {{0, 1, 2}}},
                                   {{0, 0, 0, 1, 1, 2}, {{0, 1, 2}, {3, 4}, {5}, {}}},
                                   {{2, 1, 2, 1, 1, 2}, {{}, {1, 3, 4}, {0, 2, 5}, {}, {}}},
                                   {{2, 2, 5, 5, 2, 3}, {{}, {}, {0, 1, 4}, {5}, {}, {2, 3}, {}, {}}}};
  for (const auto& c : cases) {
    std::vector<Opt<uint32_t>> items;
    for (uint32_t v : c.ids) items.push_back(V(uint32_t, v));
    auto ids = Dev<uint32_t>(pool, uint32(), items);
    const uint32_t ng = static_cast<uint32_t>(c.want.size());
    auto g = compute::internal::Grouper::MakeGroupings(*ids, ng, ctx);
    CHECK_OK(g.status());
    if (!g.ok()) continue;
    CHECK((*g)->type->id() == Type::LIST && (*g)->length == ng && (*g)->child_data.size() == 1);
    auto off = Host<int32_t>(pool, *ArrayData::Make(int32(), ng + 1, {nullptr, (*g)->buffers[1]}, 0));
    auto idx = Host<int32_t>(pool, *(*g)->child_data[0]);
    auto grouped = compute::internal::Grouper::ApplyGroupings(**g, *ids, ctx);
    CHECK_OK(grouped.status());
    std::vector<Opt<uint32_t>> gid;
    if (grouped.ok()) gid = Host<uint32_t>(pool, *(*grouped)->child_data[0]);
    for (uint32_t k = 0; k < ng; ++k) {
      std::vector<int32_t> got;
      for (int32_t i = off[k].v; i < off[k + 1].v; ++i) {
        got.push_back(idx[i].v);
        if (grouped.ok()) CHECK(gid[i].v == k);
      }
      CHECK(got == c.want[k]);
    }
  }
  auto with_null = Dev<uint32_t>(pool, uint32(), {V(uint32_t, 0), N(uint32_t), V(uint32_t, 1)});
  auto st = compute::internal::Grouper::MakeGroupings(*with_null, 5, ctx).status();
  CHECK(st.IsInvalid() && st.message() == "MakeGroupings with null ids");
}

// GroupBy over key rows wider than 64 bits ({int64, float64, int64}: the composed Grouper), against a host map.
static void WideKeyGroupByTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  const int64_t n = 50000;
  std::vector<Opt<int64_t>> v, ka, kc;
  std::vector<Opt<double>> kb;
  uint64_t st = 99;
  auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return st >> 33; };
  for (int64_t i = 0; i < n; ++i) {
    v.push_back(V(int64_t, int64_t(rnd() % 1000)));
    ka.push_back(rnd() % 30 == 0 ? N(int64_t) : V(int64_t, (int64_t(rnd() % 7) - 3) * 4000000000000ll));
    kb.push_back(V(double, double(rnd() % 5) * 0.5));
    kc.push_back(V(int64_t, int64_t(rnd() % 3) << 40));
  }
  auto r = internal::GroupBy({Datum(Dev<int64_t>(pool, int64(), v))},
                             {Datum(Dev<int64_t>(pool, int64(), ka)), Datum(Dev<double>(pool, float64(), kb)), Datum(Dev<int64_t>(pool, int64(), kc))},
                             {{"hash_sum", nullptr}}, ctx);
  CHECK_OK(r.status());
  if (!r.ok()) return;
  struct Key { bool a_valid; int64_t a; double b; int64_t c; bool operator<(const Key& o) const { return std::tie(a_valid, a, b, c) < std::tie(o.a_valid, o.a, o.b, o.c); } };
  std::map<Key, std::pair<int64_t, int64_t>> want;  // key -> (first row, sum)
  for (int64_t i = 0; i < n; ++i) {
    Key k{ka[i].valid, ka[i].valid ? ka[i].v : 0, kb[i].v, kc[i].v};
    auto it = want.find(k);
    if (it == want.end()) want[k] = {i, v[i].v};
    else it->second.second += v[i].v;
  }
  CHECK(r->length == static_cast<int64_t>(want.size()) && r->columns.size() == 4);
  auto sums = Host<int64_t>(pool, *r->columns[0]);
  auto ga = Host<int64_t>(pool, *r->columns[1]);
  auto gb = Host<double>(pool, *r->columns[2]);
  auto gc = Host<int64_t>(pool, *r->columns[3]);
  int64_t last_first = -1;
  bool ok = true;
  for (int64_t g = 0; g < r->length && ok; ++g) {
    Key k{ga[g].valid, ga[g].valid ? ga[g].v : 0, gb[g].v, gc[g].v};
    auto it = want.find(k);
    ok = it != want.end() && it->second.second == sums[g].v && it->second.first > last_first;  // first-appearance order
    if (ok) last_first = it->second.first;
  }
  CHECK(ok);
}

// Expression + the scanner's per-batch step (expression_test.cc: Bind / ExecuteScalarExpression cases; scanner.cc:456-500)
// The other array_take / array_filter value types and variable-width / dictionary Grouper keys, on the reference's own
// literals: vector_selection_test.cc:496-500, 507-515, 587-615, 1100-1104, 1135-1148, 1182-1188, 1190-1208, 1246-1275;
// hash_aggregate_test.cc:295-336 (Grouper over utf8 / dictionary keys), 338-376 (StringInt64Key), 606-646 (GroupBy).
static std::shared_ptr<ArrayData> DevOffsets32(const std::shared_ptr<MemoryPool>& pool, const std::vector<int32_t>& v) {
  return *ArrayFromHost(pool, int32(), v.data(), static_cast<int64_t>(v.size()));
}
static std::shared_ptr<ArrayData> DevList(const std::shared_ptr<MemoryPool>& pool, std::shared_ptr<DataType> type, const std::vector<int64_t>& offsets,
                                          const std::vector<bool>& valid, std::shared_ptr<ArrayData> child) {
  const bool large = type->id() == Type::LARGE_LIST;
  std::shared_ptr<ArrayData> off;
  if (large) off = *ArrayFromHost(pool, int64(), offsets.data(), static_cast<int64_t>(offsets.size()));
  else { std::vector<int32_t> o32(offsets.begin(), offsets.end()); off = DevOffsets32(pool, o32); }
  const int64_t n = static_cast<int64_t>(offsets.size()) - 1;
  std::vector<int> vb;
  bool any_null = false;
  for (int64_t i = 0; i < n; ++i) { const bool v = valid.empty() || valid[i]; vb.push_back(v ? 1 : -1); any_null |= !v; }
  auto bits = DevBool(pool, vb);
  auto out = ArrayData::Make(type, n, {any_null ? bits->buffers[0] : nullptr, off->buffers[1]}, any_null ? kUnknownNullCount : 0);
  out->child_data = {std::move(child)};
  return out;
}
// list<int32> as (null | vector) per slot, for comparisons
using IntList = std::pair<bool, std::vector<Opt<int32_t>>>;
static std::vector<IntList> HostList(const std::shared_ptr<MemoryPool>& pool, const ArrayData& a) {
  std::vector<IntList> out;
  const bool large = a.type->id() == Type::LARGE_LIST;
  auto offs = ArrayData::Make(large ? int64() : int32(), a.length + 1, {nullptr, a.buffers[1]}, 0, a.offset);
  std::vector<int64_t> o;
  if (large) for (auto& x : Host<int64_t>(pool, *offs)) o.push_back(x.v);
  else for (auto& x : Host<int32_t>(pool, *offs)) o.push_back(x.v);
  auto child = Host<int32_t>(pool, *a.child_data[0]);
  auto validity = ArrayData::Make(boolean(), a.length, {nullptr, a.buffers[0]}, 0, a.offset);
  std::vector<int> vb = a.buffers[0] ? HostBool(pool, *validity) : std::vector<int>(a.length, 1);
  for (int64_t i = 0; i < a.length; ++i) {
    IntList e{vb[i] > 0, {}};
    if (e.first) for (int64_t j = o[i]; j < o[i + 1]; ++j) e.second.push_back(child[j]);
    out.push_back(e);
  }
  return out;
}
static bool SameLists(const std::vector<IntList>& a, const std::vector<IntList>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); ++i) {
    if (a[i].first != b[i].first) return false;
    if (a[i].first && !Same(a[i].second, b[i].second)) return false;
  }
  return true;
}
static std::vector<std::pair<std::string, bool>> Strs(std::initializer_list<const char*> items) {
  std::vector<std::pair<std::string, bool>> out;
  for (const char* x : items) out.push_back(x ? std::make_pair(std::string(x), true) : std::make_pair(std::string(), false));
  return out;
}
static void NestedSelectionTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  using I = int32_t;
  const IntList LN{false, {}};
  auto L = [](std::initializer_list<Opt<I>> v) { return IntList{true, std::vector<Opt<I>>(v)}; };
  {  // the registry carries every kernel of vector_selection.cc:2133-2176
    auto* reg = ctx->func_registry();
    CHECK((*reg->GetFunction("array_take"))->num_kernels() == 16);    // primitive + 2 binary-like + 13 (large binary-like counts twice)
    CHECK((*reg->GetFunction("array_filter"))->num_kernels() == 15);  // no decimal256 (:2139)
  }
  for (auto type : {list(int32()), large_list(int32())}) {  // TakeListInt32 :1135-1148, TakeLargeListInt32 :1182-1188
    auto child = Dev<I>(pool, int32(), {V(I, 1), V(I, 2), V(I, 3)});
    auto values = DevList(pool, type, {0, 0, 2, 2, 3}, {true, true, false, true}, child);  // [[], [1,2], null, [3]]
    auto r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {V(I, 3), V(I, 2), V(I, 1)})), TakeOptions::Defaults(), ctx);
    CHECK(r.ok());
    if (r.ok()) CHECK(SameLists(HostList(pool, *r->array()), {L({V(I, 3)}), LN, L({V(I, 1), V(I, 2)})}));
    r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {N(I), V(I, 3), V(I, 0)})), TakeOptions::Defaults(), ctx);
    if (r.ok()) { CHECK(SameLists(HostList(pool, *r->array()), {LN, L({V(I, 3)}), L({})})); CHECK(r->array()->null_count == 1); }
    r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {V(I, 0), V(I, 0), V(I, 0), V(I, 0), V(I, 0), V(I, 0), V(I, 1)})), TakeOptions::Defaults(), ctx);
    if (r.ok()) CHECK(SameLists(HostList(pool, *r->array()), {L({}), L({}), L({}), L({}), L({}), L({}), L({V(I, 1), V(I, 2)})}));
    CHECK(Take(Datum(values), Datum(Dev<I>(pool, int32(), {V(I, 4)})), TakeOptions::Defaults(), ctx).status().IsIndexError());
    // FilterListInt32 :507-515
    auto f = Filter(Datum(values), Datum(DevBool(pool, {0, 1, 1, -1})), FilterOptions(FilterOptions::DROP), ctx);
    CHECK(f.ok());
    if (f.ok()) CHECK(SameLists(HostList(pool, *f->array()), {L({V(I, 1), V(I, 2)}), LN}));
    f = Filter(Datum(values), Datum(DevBool(pool, {0, 1, 1, -1})), FilterOptions(FilterOptions::EMIT_NULL), ctx);
    if (f.ok()) CHECK(SameLists(HostList(pool, *f->array()), {L({V(I, 1), V(I, 2)}), LN, LN}));
  }
  {  // TakeFixedSizeListInt32 :1190-1208: [null, [1, null, 3], [4, 5, 6], [7, 8, null]]
    auto child = Dev<I>(pool, int32(), {V(I, 0), V(I, 0), V(I, 0), V(I, 1), N(I), V(I, 3), V(I, 4), V(I, 5), V(I, 6), V(I, 7), V(I, 8), N(I)});
    auto bits = DevBool(pool, {-1, 1, 1, 1});
    auto values = ArrayData::Make(fixed_size_list(int32(), 3), 4, {bits->buffers[0]}, 1);
    values->child_data = {child};
    auto r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {N(I), V(I, 2), V(I, 0), V(I, 3)})), TakeOptions::Defaults(), ctx);
    CHECK(r.ok());
    if (r.ok()) {
      const auto& a = *r->array();
      CHECK(a.length == 4 && a.null_count == 2 && a.child_data[0]->length == 12);
      auto got = Host<I>(pool, *a.child_data[0]);
      CHECK(Same(std::vector<Opt<I>>(got.begin() + 3, got.begin() + 6), {V(I, 4), V(I, 5), V(I, 6)}));
      CHECK(Same(std::vector<Opt<I>>(got.begin() + 9, got.end()), {V(I, 7), V(I, 8), N(I)}));
      CHECK(!got[0].valid && !got[1].valid && !got[2].valid);  // list_size null child indices per null slot (:1708)
    }
  }
  {  // TakeStruct :1246-1275 / FilterStruct :587-615: struct<a int32, b utf8> = [null, {1, ""}, {2, "hello"}, {4, "eh"}]
    auto bits = DevBool(pool, {-1, 1, 1, 1});
    auto values = ArrayData::Make(struct_({{"a", int32()}, {"b", utf8()}}), 4, {bits->buffers[0]}, 1);
    values->child_data = {Dev<I>(pool, int32(), {V(I, 0), V(I, 1), V(I, 2), V(I, 4)}), *BinaryFromHost(pool, utf8(), {"", "", "hello", "eh"})};
    values->child_names = {"a", "b"};
    auto r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {V(I, 3), V(I, 1), V(I, 0)})), TakeOptions::Defaults(), ctx);
    CHECK(r.ok());
    if (r.ok()) {
      const auto& a = *r->array();
      CHECK(a.length == 3 && a.null_count == 1 && a.child_data.size() == 2);
      CHECK(Same(Host<I>(pool, *a.child_data[0]), {V(I, 4), V(I, 1), V(I, 0)}));
      CHECK(*BinaryToHost(*a.child_data[1], pool) == Strs({"eh", "", ""}));
      auto validity = ArrayData::Make(boolean(), 3, {nullptr, a.buffers[0]}, 0);
      CHECK(HostBool(pool, *validity) == std::vector<int>({1, 1, 0}));
    }
    auto f = Filter(Datum(values), Datum(DevBool(pool, {0, 1, 1, -1})), FilterOptions(FilterOptions::EMIT_NULL), ctx);
    CHECK(f.ok());
    if (f.ok()) {
      CHECK(f->array()->length == 3 && f->array()->null_count == 1);
      CHECK(Same(Host<I>(pool, *f->array()->child_data[0]), {V(I, 1), V(I, 2), N(I)}));
    }
  }
  {  // TakeDictionary :1100-1104 / FilterDictionary :496-500 over dictionary(int8, utf8) ["a", "b", "c", "d", "e"]
    auto dict = *BinaryFromHost(pool, utf8(), {"a", "b", "c", "d", "e"});
    auto idx = Dev<int8_t>(pool, int8(), {N(int8_t), V(int8_t, 4), V(int8_t, 2)});
    auto values = ArrayData::Make(dictionary(int8(), utf8()), 3, idx->buffers, 1);
    values->dictionary = dict;
    auto r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {V(I, 0), V(I, 1), V(I, 0)})), TakeOptions::Defaults(), ctx);
    CHECK(r.ok());
    if (r.ok()) {
      CHECK(r->array()->type->id() == Type::DICTIONARY && r->array()->dictionary == dict);
      CHECK(Same(Host<int8_t>(pool, *ArrayData::Make(int8(), 3, r->array()->buffers, r->array()->null_count)), {N(int8_t), V(int8_t, 4), N(int8_t)}));
    }
    auto f = Filter(Datum(values), Datum(DevBool(pool, {-1, 1, 0})), FilterOptions(FilterOptions::EMIT_NULL), ctx);
    CHECK(f.ok());
    if (f.ok()) CHECK(Same(Host<int8_t>(pool, *ArrayData::Make(int8(), 2, f->array()->buffers, f->array()->null_count)), {N(int8_t), V(int8_t, 4)}));
  }
  {  // fixed_size_binary(3), decimal128 and large_utf8 through the same kernels (:1030-1051 run them over every type)
    const char raw[] = "abcdefghijkl";
    auto bytes = *ArrayFromHost(pool, uint8(), raw, 12);
    auto values = ArrayData::Make(fixed_size_binary(3), 4, {nullptr, bytes->buffers[1]}, 0);
    auto r = Take(Datum(values), Datum(Dev<I>(pool, int32(), {V(I, 3), N(I), V(I, 0)})), TakeOptions::Defaults(), ctx);
    CHECK(r.ok());
    if (r.ok()) {
      auto got = Host<uint8_t>(pool, *ArrayData::Make(uint8(), 9, {nullptr, r->array()->buffers[1]}, 0));
      std::string str;
      for (auto& g : got) str.push_back(static_cast<char>(g.v));
      CHECK(str == std::string("jkl\0\0\0abc", 9) && r->array()->null_count == 1);
    }
    CHECK(Filter(Datum(ArrayData::Make(decimal256(40, 3), 0, {nullptr, bytes->buffers[1]}, 0)), Datum(DevBool(pool, {})),
                 FilterOptions::Defaults(), ctx).status().IsNotImplemented());
    std::vector<int64_t> lo = {0, 2, 2, 7};
    auto loff = *ArrayFromHost(pool, int64(), lo.data(), 4);
    auto ldata = *ArrayFromHost(pool, uint8(), "hihello", 7);
    auto large = ArrayData::Make(large_utf8(), 3, {nullptr, loff->buffers[1]}, 0);
    large->buffers.push_back(ldata->buffers[1]);
    auto t = Take(Datum(large), Datum(Dev<I>(pool, int32(), {V(I, 2), V(I, 0), N(I)})), TakeOptions::Defaults(), ctx);
    CHECK(t.ok());
    if (t.ok()) {
      auto o = Host<int64_t>(pool, *ArrayData::Make(int64(), 4, {nullptr, t->array()->buffers[1]}, 0));
      CHECK(Same(o, {V(int64_t, 0), V(int64_t, 5), V(int64_t, 7), V(int64_t, 7)}));
    }
  }
  {  // Grouper StringInt64Key, hash_aggregate_test.cc:338-376
    auto g = internal::Grouper::Make({ValueDescr::Array(utf8()), ValueDescr::Array(int64())}, ctx);
    CHECK(g.ok());
    if (g.ok()) {
      auto consume = [&](std::initializer_list<const char*> s, std::vector<Opt<int64_t>> v, std::vector<Opt<uint32_t>> want) {
        std::vector<std::string> items; std::vector<bool> valid;
        for (const char* x : s) { items.push_back(x ? x : ""); valid.push_back(x != nullptr); }
        auto ids = (*g)->Consume(ExecBatch({Datum(*BinaryFromHost(pool, utf8(), items, valid)), Datum(Dev<int64_t>(pool, int64(), v))}, static_cast<int64_t>(v.size())));
        CHECK(ids.ok());
        if (ids.ok()) CHECK(Same(Host<uint32_t>(pool, *ids->array()), want));
      };
      using L64 = int64_t; using U = uint32_t;
      consume({"eh", "eh"}, {V(L64, 0), V(L64, 0)}, {V(U, 0), V(U, 0)});
      consume({"eh", "eh"}, {V(L64, 0), N(L64)}, {V(U, 0), V(U, 1)});
      consume({"eh", "bee"}, {V(L64, 1), V(L64, 1)}, {V(U, 2), V(U, 3)});
      consume({"eh", "bee"}, {N(L64), V(L64, 1)}, {V(U, 1), V(U, 3)});
      auto u = (*g)->GetUniques();
      CHECK(u.ok());
      if (u.ok()) {
        CHECK(*BinaryToHost(*u->values[0].array(), pool) == Strs({"eh", "eh", "eh", "bee"}));
        CHECK(Same(Host<L64>(pool, *u->values[1].array()), {V(L64, 0), N(L64), V(L64, 1), V(L64, 1)}));
      }
    }
  }
  for (int dict_keys = 0; dict_keys < 2; ++dict_keys) {  // GroupBy SumOnlyStringAndDictKeys, hash_aggregate_test.cc:606-646
    using D = double;
    auto arg = Dev<D>(pool, float64(), {V(D, 1.0), N(D), V(D, 0.0), N(D), V(D, 4.0), V(D, 3.25), V(D, 0.125), V(D, -0.25), V(D, 0.75), N(D)});
    std::shared_ptr<ArrayData> key;
    auto words = *BinaryFromHost(pool, utf8(), {"alfa", "beta", "gama"});
    if (!dict_keys) {
      key = *BinaryFromHost(pool, utf8(), {"alfa", "alfa", "beta", "gama", "", "alfa", "beta", "beta", "", "gama"},
                            {true, true, true, true, false, true, true, true, false, true});
    } else {
      auto idx = Dev<I>(pool, int32(), {V(I, 0), V(I, 0), V(I, 1), V(I, 2), N(I), V(I, 0), V(I, 1), V(I, 1), N(I), V(I, 2)});
      key = ArrayData::Make(dictionary(int32(), utf8()), 10, idx->buffers, 2);
      key->dictionary = words;
    }
    auto r = internal::GroupBy({Datum(arg)}, {Datum(key)}, {{"hash_sum", nullptr}}, ctx);
    CHECK(r.ok());
    if (r.ok()) {
      CHECK(r->length == 4 && r->field_names == std::vector<std::string>({"hash_sum", "key_0"}));
      CHECK(Same(Host<D>(pool, *r->columns[0]), {V(D, 4.25), V(D, -0.125), N(D), V(D, 4.75)}));
      if (!dict_keys) CHECK(*BinaryToHost(*r->columns[1], pool) == Strs({"alfa", "beta", "gama", nullptr}));
      else {
        CHECK(r->columns[1]->type->id() == Type::DICTIONARY && r->columns[1]->dictionary == words);
        CHECK(Same(Host<I>(pool, *ArrayData::Make(int32(), 4, r->columns[1]->buffers, r->columns[1]->null_count)), {V(I, 0), V(I, 1), V(I, 2), N(I)}));
      }
    }
  }
  {  // differing dictionaries: NotImplemented("Unifying differing dictionaries"), hash_aggregate_test.cc:324-336
    auto g = internal::Grouper::Make({ValueDescr::Array(dictionary(int32(), utf8()))}, ctx);
    auto mk = [&](std::vector<std::string> words) {
      auto idx = Dev<I>(pool, int32(), {V(I, 0), V(I, 1)});
      auto a = ArrayData::Make(dictionary(int32(), utf8()), 2, idx->buffers, 0);
      a->dictionary = *BinaryFromHost(pool, utf8(), words);
      return a;
    };
    CHECK((*g)->Consume(ExecBatch({Datum(mk({"ex", "why"}))}, 2)).ok());
    CHECK((*g)->Consume(ExecBatch({Datum(mk({"ex", "why"}))}, 2)).ok());  // equal content, other buffers
    auto bad = (*g)->Consume(ExecBatch({Datum(mk({"different", "dictionary"}))}, 2));
    CHECK(bad.status().IsNotImplemented() && bad.status().message() == "Unifying differing dictionaries");
  }
}

// ---- BASELINE config 5 in C++: ShardedGroupBy (sharded_groupby.h).  Two RANKS AS THREADS on the one GPU, each with its
// own MemoryPool / ExecContext, the collectives staged through host memory (RCCL refuses two ranks per device); and the
// RCCL transport itself on a communicator of one rank.
struct ThreadWorld {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  std::vector<std::vector<uint8_t>> slots;
  explicit ThreadWorld(int w) : world(w), slots(w) {}
  void Barrier() {
    std::unique_lock<std::mutex> lock(mu);
    const int gen = generation;
    if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
    else cv.wait(lock, [&] { return generation != gen; });
  }
};
class ThreadCollectives : public internal::Collectives {
 public:
  ThreadCollectives(std::shared_ptr<ThreadWorld> w, int rank, amdarrow_ctx* abi) : w_(std::move(w)), rank_(rank), abi_(abi) {}
  int rank() const override { return rank_; }
  int world() const override { return w_->world; }
  Status AllReduce(void* buf, int64_t count, Word word, Op op, void*) override {
    auto& mine = w_->slots[rank_];
    mine.resize(8 * count);
    if (amdarrow_memcpy_d2h(abi_, mine.data(), buf, 8 * count) != 0) return Status::Invalid("d2h");
    w_->Barrier();
    std::vector<uint8_t> out(8 * count);
    for (int64_t i = 0; i < count; ++i) {
      if (word == INT64) {
        int64_t acc; memcpy(&acc, &w_->slots[0][8 * i], 8);
        for (int r = 1; r < w_->world; ++r) { int64_t v; memcpy(&v, &w_->slots[r][8 * i], 8); acc = op == SUM ? acc + v : std::min(acc, v); }
        memcpy(&out[8 * i], &acc, 8);
      } else {
        double acc; memcpy(&acc, &w_->slots[0][8 * i], 8);
        for (int r = 1; r < w_->world; ++r) { double v; memcpy(&v, &w_->slots[r][8 * i], 8); acc = op == SUM ? acc + v : std::min(acc, v); }
        memcpy(&out[8 * i], &acc, 8);
      }
    }
    w_->Barrier();
    return amdarrow_memcpy_h2d(abi_, buf, out.data(), 8 * count) == 0 ? Status::OK() : Status::Invalid("h2d");
  }
  Status AllGather(const void* send, void* recv, int64_t bytes, void*) override {
    auto& mine = w_->slots[rank_];
    mine.resize(bytes);
    if (amdarrow_memcpy_d2h(abi_, mine.data(), send, bytes) != 0) return Status::Invalid("d2h");
    w_->Barrier();
    std::vector<uint8_t> out;
    for (int r = 0; r < w_->world; ++r) out.insert(out.end(), w_->slots[r].begin(), w_->slots[r].begin() + bytes);
    w_->Barrier();
    return amdarrow_memcpy_h2d(abi_, recv, out.data(), bytes * w_->world) == 0 ? Status::OK() : Status::Invalid("h2d");
  }

 private:
  std::shared_ptr<ThreadWorld> w_;
  int rank_;
  amdarrow_ctx* abi_;
};

struct ShardStep { std::vector<int64_t> vals; std::vector<int32_t> keys; std::vector<int> mask; };
static std::vector<ShardStep> MakeShardSteps(int64_t n) {
  std::vector<ShardStep> steps(5);
  uint64_t x = 0x0ff1ce;
  auto next = [&x] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int s = 0; s < 5; ++s) {
    auto& st = steps[s];
    for (int64_t i = 0; i < n; ++i) {
      st.vals.push_back(static_cast<int64_t>(next() % 201) - 100);
      int32_t k = static_cast<int32_t>(next() % 30000);
      if (s == 1 && i >= n / 2) k += 5000000;                         // rank 1's rows leave the dense plan at step 1
      if (s >= 3) k = static_cast<int32_t>(next() % 2000000000);      // a range no dense plan holds: gather for good
      st.keys.push_back(k);
      st.mask.push_back(next() % 2 ? 1 : 0);
    }
  }
  return steps;
}
static void ShardedGroupByTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  using L = int64_t; using K = int32_t;
  const int64_t n = 400000;
  const auto steps = MakeShardSteps(n);
  // what one process computes over all rows: Filter, Filter, GroupBy (the reference's definition of the merged table)
  std::vector<std::vector<Opt<L>>> want_sums(steps.size()), want_counts(steps.size());
  std::vector<std::vector<Opt<K>>> want_keys(steps.size());
  for (size_t s = 0; s < steps.size(); ++s) {
    std::vector<Opt<L>> v; std::vector<Opt<K>> k;
    for (int64_t i = 0; i < n; ++i) if (steps[s].mask[i]) { v.push_back(V(L, steps[s].vals[i])); k.push_back(V(K, steps[s].keys[i])); }
    auto r = internal::GroupBy({Datum(Dev<L>(pool, int64(), v)), Datum(Dev<L>(pool, int64(), v))}, {Datum(Dev<K>(pool, int32(), k))},
                               {{"hash_sum", nullptr}, {"hash_count", nullptr}}, ctx);
    CHECK(r.ok());
    if (!r.ok()) return;
    want_sums[s] = Host<L>(pool, *r->columns[0]);
    want_counts[s] = Host<L>(pool, *r->columns[1]);
    want_keys[s] = Host<K>(pool, *r->columns[2]);
  }
  const int world = 2;
  auto tw = std::make_shared<ThreadWorld>(world);
  std::vector<int> ok(world, 0), replans(world, -1), dense_at_end(world, -1);
  auto rank_main = [&](int rank) {
    auto rpool = MemoryPool::Make(0);
    if (!rpool.ok()) return;
    auto reg = FunctionRegistry::Make();
    internal::RegisterHashAggregateBasic(reg.get());
    ExecContext rctx(*rpool, reg.get());
    auto coll = std::make_shared<ThreadCollectives>(tw, rank, (*rpool)->ctx());
    auto sg = internal::ShardedGroupBy::Make(&rctx, coll, int32(), int64(), 1 << 20);
    if (!sg.ok()) return;
    const int64_t lo = rank * (n / world), hi = (rank + 1) * (n / world);
    bool good = true;
    for (size_t s = 0; s < steps.size(); ++s) {
      std::vector<Opt<L>> v; std::vector<Opt<K>> k; std::vector<int> m;
      for (int64_t i = lo; i < hi; ++i) { v.push_back(V(L, steps[s].vals[i])); k.push_back(V(K, steps[s].keys[i])); m.push_back(steps[s].mask[i]); }
      auto r = (*sg)->Run(Datum(Dev<L>(*rpool, int64(), v)), Datum(Dev<K>(*rpool, int32(), k)), Datum(DevBool(*rpool, m)));
      if (!r.ok()) { fprintf(stderr, "rank %d step %zu: %s\n", rank, s, r.status().ToString().c_str()); good = false; tw->Barrier(); continue; }
      good = good && r->length == static_cast<int64_t>(want_sums[s].size());
      good = good && Same(Host<L>(*rpool, *r->columns[0]), want_sums[s]) && Same(Host<L>(*rpool, *r->columns[1]), want_counts[s]) &&
             Same(Host<K>(*rpool, *r->columns[2]), want_keys[s]);
    }
    ok[rank] = good;
    replans[rank] = (*sg)->replans();
    dense_at_end[rank] = (*sg)->dense();
  };
  std::vector<std::thread> threads;
  for (int r = 0; r < world; ++r) threads.emplace_back(rank_main, r);
  for (auto& t : threads) t.join();
  for (int r = 0; r < world; ++r) {
    CHECK(ok[r] == 1);
    CHECK(replans[r] == 2);        // step 1 (keys jumped) and step 3 (keys spread out): redone on the gather protocol
    CHECK(dense_at_end[r] == 0);   // ... and from step 4 on the plan is the gather protocol
  }
  {  // the RCCL transport: a communicator of one rank, the protocol forced to run (ncclAllReduce x2, then ncclAllGather)
    auto self = internal::MakeRcclSelfCollectives(0);
    CHECK(self.ok());
    if (self.ok()) {
      auto sg = internal::ShardedGroupBy::Make(ctx, *self, int32(), int64(), 1 << 20);
      CHECK(sg.ok());
      (*sg)->set_merge_even_alone(true);
      for (size_t s : {size_t(0), size_t(4)}) {  // step 0: dense plan; step 4: a key range that needs the gather protocol
        std::vector<Opt<L>> v; std::vector<Opt<K>> k; std::vector<int> m;
        for (int64_t i = 0; i < n; ++i) { v.push_back(V(L, steps[s].vals[i])); k.push_back(V(K, steps[s].keys[i])); m.push_back(steps[s].mask[i]); }
        auto r = (*sg)->Run(Datum(Dev<L>(pool, int64(), v)), Datum(Dev<K>(pool, int32(), k)), Datum(DevBool(pool, m)));
        CHECK(r.ok());
        if (r.ok()) CHECK(Same(Host<L>(pool, *r->columns[0]), want_sums[s]) && Same(Host<L>(pool, *r->columns[1]), want_counts[s]) &&
                          Same(Host<K>(pool, *r->columns[2]), want_keys[s]));
      }
    }
  }
}

static void ExpressionTests(const std::shared_ptr<MemoryPool>& pool, ExecContext* ctx) {
  auto a = Dev<int32_t>(pool, int32(), {V(int32_t, 1), V(int32_t, 5), N(int32_t), V(int32_t, 3), V(int32_t, 9), V(int32_t, 4)});
  auto b = Dev<double>(pool, float64(), {V(double, 0.5), V(double, 1.5), V(double, 2.5), N(double), V(double, 4.5), V(double, 5.5)});
  auto c = Dev<int64_t>(pool, int64(), {V(int64_t, 10), V(int64_t, 20), V(int64_t, 30), V(int64_t, 40), V(int64_t, 50), V(int64_t, 60)});
  auto sch = schema({field("a", int32()), field("b", float64()), field("c", int64())});
  auto batch = RecordBatch::Make(sch, 6, {a, b, c});
  // a > 2 and b is valid  (int32 column against an int64 literal: promoted by DispatchBest)
  Expression filter = and_(greater(field_ref("a"), literal<int64_t>(int64(), 2)), is_valid(field_ref("b")));
  Expression projection = project({field_ref("c"), call("add", {field_ref("a"), field_ref("b")})}, {"c", "a_plus_b"});
  CHECK(ExecuteScalarExpression(filter, Datum(batch), ctx).status().message() == "Cannot Execute unbound expression.");
  CHECK(FieldsInExpression(filter) == (std::vector<std::string>{"a", "b"}) && !filter.IsBound());
  auto bf = filter.Bind(*sch, ctx);
  auto bp = projection.Bind(*sch, ctx);
  CHECK_OK(bf.status());
  CHECK_OK(bp.status());
  if (!bf.ok() || !bp.ok()) return;
  CHECK(bf->IsBound() && bf->type()->id() == Type::BOOL && bp->type()->id() == Type::STRUCT);
  CHECK(bp->call()->arguments[1].type()->id() == Type::DOUBLE);  // int32 + float64
  CHECK(field_ref("nope").Bind(*sch, ctx)->type()->id() == Type::NA);
  CHECK(call("no_such_function", {field_ref("a")}).Bind(*sch, ctx).status().IsKeyError());
  auto mask = ExecuteScalarExpression(*bf, Datum(batch), ctx);
  CHECK_OK(mask.status());
  if (mask.ok()) CHECK(HostBool(pool, *mask->array()) == (std::vector<int>{0, 1, -1, 0, 1, 1}));  // Kleene: null and true = null, true and false = false
  auto out = FilterAndProjectRecordBatch(batch, *bf, *bp, ctx);
  CHECK_OK(out.status());
  if (out.ok()) {
    CHECK((*out)->num_rows() == 3 && (*out)->num_columns() == 2 && (*out)->schema()->field(1).name == "a_plus_b");
    CHECK(Same(Host<int64_t>(pool, *(*out)->column(0)), {V(int64_t, 20), V(int64_t, 50), V(int64_t, 60)}));
    CHECK(Same(Host<double>(pool, *(*out)->column(1)), {V(double, 6.5), V(double, 13.5), V(double, 9.5)}));
  }
  // a literal filter keeps everything or nothing (scanner.cc:468-478)
  auto all = FilterAndProjectRecordBatch(batch, literal<uint8_t>(boolean(), 1), *bp, ctx);
  auto none = FilterAndProjectRecordBatch(batch, literal<uint8_t>(boolean(), 0), *bp, ctx);
  CHECK(all.ok() && (*all)->num_rows() == 6);
  CHECK(none.ok() && (*none)->num_rows() == 0);
  // or_ / not_ / equal over several fields
  auto e2 = or_(not_(less_equal(field_ref("c"), literal<int64_t>(int64(), 40))), equal(field_ref("a"), literal<int32_t>(int32(), 1))).Bind(*sch, ctx);
  CHECK_OK(e2.status());
  if (e2.ok()) {
    auto m2 = ExecuteScalarExpression(*e2, Datum(batch), ctx);
    CHECK_OK(m2.status());
    if (m2.ok()) CHECK(HostBool(pool, *m2->array()) == (std::vector<int>{1, 0, -1, 0, 1, 1}));
  }
}

static int RunGpu() {
  auto pool_r = MemoryPool::Make(0);
  if (!pool_r.ok()) { fprintf(stderr, "no device: %s\n", pool_r.status().ToString().c_str()); return 2; }
  auto pool = *pool_r;
  ExecContext ctx(pool);  // default registry = the HIP-backed one
  TypedTests<uint8_t>(pool, &ctx, uint8());
  TypedTests<int8_t>(pool, &ctx, int8());
  TypedTests<uint16_t>(pool, &ctx, uint16());
  TypedTests<int16_t>(pool, &ctx, int16());
  TypedTests<uint32_t>(pool, &ctx, uint32());
  TypedTests<int32_t>(pool, &ctx, int32());
  TypedTests<uint64_t>(pool, &ctx, uint64());
  TypedTests<int64_t>(pool, &ctx, int64());
  TypedTests<float>(pool, &ctx, float32());
  TypedTests<double>(pool, &ctx, float64());

  const double nan = std::numeric_limits<double>::quiet_NaN();
  {  // SortReal with NaN and null (vector_sort_test.cc:436-439), chunked NaN case (:639-647)
    auto a = Dev<double>(pool, float64(), {V(double, 3), V(double, 4), V(double, nan), V(double, 1), V(double, 2), N(double)});
    auto r = SortIndices(Datum(a), SortOrder::Descending, &ctx);
    CHECK_OK(r.status());
    if (r.ok()) CHECK(Same(Host<uint64_t>(pool, *r->array()), {V(uint64_t, 1), V(uint64_t, 0), V(uint64_t, 4), V(uint64_t, 3), V(uint64_t, 2), V(uint64_t, 5)}));
    auto ca = std::make_shared<ChunkedArray>(
        std::vector<std::shared_ptr<ArrayData>>{Dev<float>(pool, float32(), {N(float), V(float, 1)}), Dev<float>(pool, float32(), {V(float, 3), N(float), V(float, nan)}),
                                                Dev<float>(pool, float32(), {V(float, nan), V(float, 1)})}, float32());
    auto rc = SortIndices(Datum(ca), SortOrder::Ascending, &ctx);
    CHECK_OK(rc.status());
    if (rc.ok()) CHECK(Same(Host<uint64_t>(pool, *rc->array()), {V(uint64_t, 1), V(uint64_t, 6), V(uint64_t, 2), V(uint64_t, 4), V(uint64_t, 5), V(uint64_t, 0), V(uint64_t, 3)}));
  }
  {  // chunked filter follows the input chunking (vector_selection_test.cc:733-746)
    auto values = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{Dev<int32_t>(pool, int32(), {V(int32_t, 7)}),
                                                                                         Dev<int32_t>(pool, int32(), {V(int32_t, 8), V(int32_t, 9)})}, int32());
    auto filter = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{DevBool(pool, {0, 1}), DevBool(pool, {1})}, boolean());
    auto r = Filter(Datum(values), Datum(filter), FilterOptions::Defaults(), &ctx);
    CHECK_OK(r.status());
    if (r.ok()) {
      CHECK(r->kind() == Datum::CHUNKED_ARRAY && r->chunked_array()->length() == 2);
      std::vector<int32_t> all;
      for (const auto& c : r->chunked_array()->chunks()) for (auto& o : Host<int32_t>(pool, *c)) all.push_back(o.v);
      CHECK(all == std::vector<int32_t>({8, 9}));
    }
    auto nb = CallFunction("filter", {Datum(Dev<int32_t>(pool, int32(), {V(int32_t, 7)})), Datum(Dev<int32_t>(pool, int32(), {V(int32_t, 1)}))}, &ctx);
    CHECK(nb.status().IsNotImplemented());  // non-boolean filter (:1885-1887)
  }
  {  // checked arithmetic errors and wrap-around (scalar_arithmetic.cc:69-251)
    auto mx = Dev<int8_t>(pool, int8(), {V(int8_t, 127)});
    auto one = Dev<int8_t>(pool, int8(), {V(int8_t, 1)});
    auto w = Add(Datum(mx), Datum(one), ArithmeticOptions(false), &ctx);
    CHECK_OK(w.status());
    if (w.ok()) CHECK(Same(Host<int8_t>(pool, *w->array()), {V(int8_t, -128)}));
    auto e = Add(Datum(mx), Datum(one), ArithmeticOptions(true), &ctx);
    CHECK(e.status().IsInvalid() && e.status().message() == "overflow");
    auto d = Divide(Datum(one), Datum(Dev<int8_t>(pool, int8(), {V(int8_t, 0)})), ArithmeticOptions(false), &ctx);
    CHECK(d.status().IsInvalid() && d.status().message() == "divide by zero");
  }
  {  // GroupBy.CountAndSum (hash_aggregate_test.cc:561-605) + hash_mean
    auto argument = Dev<double>(pool, float64(), {V(double, 1.0), N(double), V(double, 0.0), N(double), V(double, 4.0), V(double, 3.25), V(double, 0.125),
                                                  V(double, -0.25), V(double, 0.75), N(double)});
    auto key = Dev<int64_t>(pool, int64(), {V(int64_t, 1), V(int64_t, 1), V(int64_t, 2), V(int64_t, 3), N(int64_t), V(int64_t, 1), V(int64_t, 2),
                                            V(int64_t, 2), N(int64_t), V(int64_t, 3)});
    CountOptions count_options;
    auto r = internal::GroupBy({Datum(argument), Datum(argument), Datum(key), Datum(argument)}, {Datum(key)},
                               {{"hash_count", &count_options}, {"hash_sum", nullptr}, {"hash_sum", nullptr}, {"hash_mean", nullptr}}, &ctx);
    CHECK_OK(r.status());
    if (r.ok()) {
      CHECK(r->length == 4 && r->field_names == std::vector<std::string>({"hash_count", "hash_sum", "hash_sum", "hash_mean", "key_0"}));
      CHECK(Same(Host<int64_t>(pool, *r->columns[0]), {V(int64_t, 2), V(int64_t, 3), V(int64_t, 0), V(int64_t, 2)}));
      CHECK(Same(Host<double>(pool, *r->columns[1]), {V(double, 4.25), V(double, -0.125), N(double), V(double, 4.75)}));
      CHECK(Same(Host<int64_t>(pool, *r->columns[2]), {V(int64_t, 3), V(int64_t, 6), V(int64_t, 6), N(int64_t)}));
      CHECK(Same(Host<double>(pool, *r->columns[3]), {V(double, 2.125), V(double, -0.125 / 3), N(double), V(double, 2.375)}));
      CHECK(Same(Host<int64_t>(pool, *r->columns[4]), {V(int64_t, 1), V(int64_t, 2), V(int64_t, 3), N(int64_t)}));
      CHECK(r->columns[0]->buffers[0] == nullptr);  // hash_count never carries a bitmap
    }
    // multi-batch: exec_chunksize 3 -> same rows (groups keep first-appearance ids across batches)
    ctx.set_exec_chunksize(3);
    auto r2 = internal::GroupBy({Datum(argument)}, {Datum(key)}, {{"hash_sum", nullptr}}, &ctx);
    ctx.set_exec_chunksize(kDefaultExecChunksize);
    CHECK_OK(r2.status());
    if (r2.ok()) {
      CHECK(Same(Host<double>(pool, *r2->columns[0]), {V(double, 4.25), V(double, -0.125), N(double), V(double, 4.75)}));
      CHECK(Same(Host<int64_t>(pool, *r2->columns[1]), {V(int64_t, 1), V(int64_t, 2), V(int64_t, 3), N(int64_t)}));
    }
    // GroupBy.MinMaxOnly (hash_aggregate_test.cc:516-553): struct{min, max}, null for a group without values;
    // also over three batches, and with EMIT_NULL (a null in the group nulls it)
    for (int pass = 0; pass < 3; ++pass) {
      MinMaxOptions emit(MinMaxOptions::EMIT_NULL);
      if (pass == 1) ctx.set_exec_chunksize(4);
      auto mm = internal::GroupBy({Datum(argument)}, {Datum(key)}, {{"hash_min_max", pass == 2 ? &emit : nullptr}}, &ctx);
      ctx.set_exec_chunksize(kDefaultExecChunksize);
      CHECK_OK(mm.status());
      if (!mm.ok()) continue;
      CHECK(mm->length == 4 && mm->field_names == std::vector<std::string>({"hash_min_max", "key_0"}));
      const ArrayData& st = *mm->columns[0];
      CHECK(st.type->id() == Type::STRUCT && st.child_data.size() == 2 && st.child_names == std::vector<std::string>({"min", "max"}));
      if (st.child_data.size() != 2) continue;
      if (pass < 2) {
        CHECK(Same(Host<double>(pool, *st.child_data[0]), {V(double, 1.0), V(double, -0.25), N(double), V(double, 0.75)}));
        CHECK(Same(Host<double>(pool, *st.child_data[1]), {V(double, 3.25), V(double, 0.125), N(double), V(double, 4.0)}));
      } else {  // group 1 holds a null argument
        CHECK(Same(Host<double>(pool, *st.child_data[0]), {N(double), V(double, -0.25), N(double), V(double, 0.75)}));
        CHECK(Same(Host<double>(pool, *st.child_data[1]), {N(double), V(double, 0.125), N(double), V(double, 4.0)}));
      }
      CHECK(st.child_data[0]->buffers[0] == st.child_data[1]->buffers[0]);  // one shared bitmap (:741-742)
      CHECK(Same(Host<int64_t>(pool, *mm->columns[1]), {V(int64_t, 1), V(int64_t, 2), V(int64_t, 3), N(int64_t)}));
    }
    CHECK(internal::GroupBy({Datum(DevBool(pool, {1, 0}))}, {Datum(Dev<int64_t>(pool, int64(), {V(int64_t, 1), V(int64_t, 2)}))},
                            {{"hash_min_max", nullptr}}, &ctx).status().IsNotImplemented());  // :674-676
    // two key columns: the packed fused path (large input) against the batch-wise Grouper path -- same rows
    {
      const int64_t nn = 400000;
      std::vector<Opt<int64_t>> av; std::vector<Opt<int32_t>> k1; std::vector<Opt<int16_t>> k2;
      uint64_t st = 12345;
      auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return st >> 33; };
      for (int64_t i = 0; i < nn; ++i) {
        av.push_back(rnd() % 50 == 0 ? N(int64_t) : V(int64_t, int64_t(rnd() % 2001) - 1000));
        k1.push_back(rnd() % 40 == 0 ? N(int32_t) : V(int32_t, int32_t(rnd() % 300) + 7000));
        k2.push_back(V(int16_t, int16_t(rnd() % 90) - 45));
      }
      auto a2 = Dev<int64_t>(pool, int64(), av);
      auto ka = Dev<int32_t>(pool, int32(), k1);
      auto kb = Dev<int16_t>(pool, int16(), k2);
      const std::vector<internal::Aggregate> ag = {{"hash_sum", nullptr}, {"hash_count", nullptr}};
      auto fused = internal::GroupBy({Datum(a2), Datum(a2)}, {Datum(ka), Datum(kb)}, ag, &ctx);
      setenv("AMDARROW_GROUPBY_BATCHWISE", "1", 1);
      auto batch = internal::GroupBy({Datum(a2), Datum(a2)}, {Datum(ka), Datum(kb)}, ag, &ctx);
      unsetenv("AMDARROW_GROUPBY_BATCHWISE");
      // aggregates the fused pass does not carry (hash_min_max, hash_count of nulls) ride on group ids behind it
      {
        CountOptions count_nulls(CountOptions::COUNT_NULL);
        const std::vector<internal::Aggregate> mix = {{"hash_min_max", nullptr}, {"hash_sum", nullptr}, {"hash_count", &count_nulls}};
        auto f2 = internal::GroupBy({Datum(a2), Datum(a2), Datum(a2)}, {Datum(ka), Datum(kb)}, mix, &ctx);
        auto only = internal::GroupBy({Datum(a2)}, {Datum(ka)}, {{"hash_min_max", nullptr}}, &ctx);  // no fused aggregate at all
        setenv("AMDARROW_GROUPBY_BATCHWISE", "1", 1);
        auto b2 = internal::GroupBy({Datum(a2), Datum(a2), Datum(a2)}, {Datum(ka), Datum(kb)}, mix, &ctx);
        auto only_b = internal::GroupBy({Datum(a2)}, {Datum(ka)}, {{"hash_min_max", nullptr}}, &ctx);
        unsetenv("AMDARROW_GROUPBY_BATCHWISE");
        CHECK_OK(f2.status());
        CHECK_OK(b2.status());
        CHECK_OK(only.status());
        CHECK_OK(only_b.status());
        if (f2.ok() && b2.ok()) {
          CHECK(f2->length == b2->length && f2->field_names == b2->field_names);
          CHECK(Same(Host<int64_t>(pool, *f2->columns[0]->child_data[0]), Host<int64_t>(pool, *b2->columns[0]->child_data[0])));
          CHECK(Same(Host<int64_t>(pool, *f2->columns[0]->child_data[1]), Host<int64_t>(pool, *b2->columns[0]->child_data[1])));
          CHECK(Same(Host<int64_t>(pool, *f2->columns[1]), Host<int64_t>(pool, *b2->columns[1])));
          CHECK(Same(Host<int64_t>(pool, *f2->columns[2]), Host<int64_t>(pool, *b2->columns[2])));
          CHECK(Same(Host<int32_t>(pool, *f2->columns[3]), Host<int32_t>(pool, *b2->columns[3])));
        }
        if (only.ok() && only_b.ok()) {
          CHECK(only->length == only_b->length && only->length == 301);
          CHECK(Same(Host<int64_t>(pool, *only->columns[0]->child_data[1]), Host<int64_t>(pool, *only_b->columns[0]->child_data[1])));
          CHECK(Same(Host<int32_t>(pool, *only->columns[1]), Host<int32_t>(pool, *only_b->columns[1])));
        }
      }
      CHECK_OK(fused.status());
      CHECK_OK(batch.status());
      if (fused.ok() && batch.ok()) {
        CHECK(fused->length == batch->length && fused->length > 20000);
        CHECK(fused->field_names == batch->field_names && fused->field_names.back() == "key_1");
        CHECK(Same(Host<int64_t>(pool, *fused->columns[0]), Host<int64_t>(pool, *batch->columns[0])));
        CHECK(Same(Host<int64_t>(pool, *fused->columns[1]), Host<int64_t>(pool, *batch->columns[1])));
        CHECK(Same(Host<int32_t>(pool, *fused->columns[2]), Host<int32_t>(pool, *batch->columns[2])));
        CHECK(Same(Host<int16_t>(pool, *fused->columns[3]), Host<int16_t>(pool, *batch->columns[3])));
        CHECK(fused->columns[2]->null_count == batch->columns[2]->null_count && fused->columns[3]->buffers[0] == nullptr);
      }
    }
    auto bad = internal::GroupBy({Datum(argument)}, {Datum(key)}, {{"hash_sum", nullptr}, {"hash_count", nullptr}}, &ctx);
    CHECK(bad.status().IsInvalid());  // aggregates.size() != arguments.size() (:811-815)
    CHECK(CallFunction("hash_sum", {Datum(argument), Datum(key), Datum(key)}, &ctx).status().IsNotImplemented());
  }
  {  // FloatingPointKey (hash_aggregate_test.cc:280-292): -0.0 != 0.0, NaNs with equal bits group
    auto g = internal::Grouper::Make({ValueDescr::Array(float32())}, &ctx);
    CHECK_OK(g.status());
    if (g.ok()) {
      auto b1 = ExecBatch::Make({Datum(Dev<float>(pool, float32(), {V(float, 0.0f), V(float, -0.0f)}))});
      auto i1 = (*g)->Consume(*b1);
      CHECK_OK(i1.status());
      if (i1.ok()) CHECK(Same(Host<uint32_t>(pool, *i1->array()), {V(uint32_t, 0), V(uint32_t, 1)}));
      auto b2 = ExecBatch::Make({Datum(Dev<float>(pool, float32(), {V(float, float(nan)), V(float, float(nan))}))});
      auto i2 = (*g)->Consume(*b2);
      CHECK_OK(i2.status());
      if (i2.ok()) CHECK(Same(Host<uint32_t>(pool, *i2->array()), {V(uint32_t, 2), V(uint32_t, 2)}));
    }
  }
  {  // mask-building scalars: scalar_boolean_test.cc:58-155, scalar_validity_test.cc, scalar_fill_null_test.cc
    auto left = DevBool(pool, {1, 1, 1, 0, 0, -1}), right = DevBool(pool, {1, 0, -1, 0, -1, -1});
    struct BC { const char* name; std::vector<int> want; };
    for (const BC& c : {BC{"and", {1, 0, -1, 0, -1, -1}}, BC{"or", {1, 1, -1, 0, -1, -1}}, BC{"xor", {0, 1, -1, 0, -1, -1}},
                        BC{"and_kleene", {1, 0, -1, 0, 0, -1}}, BC{"or_kleene", {1, 1, 1, 0, -1, -1}}}) {
      auto r = CallFunction(c.name, {Datum(left), Datum(right)}, &ctx);
      CHECK_OK(r.status());
      if (r.ok()) CHECK(HostBool(pool, *r->array()) == c.want);
    }
    auto inv = CallFunction("invert", {Datum(DevBool(pool, {1, 0, 1, -1, 0, 1, 0, -1}))}, &ctx);
    CHECK_OK(inv.status());
    if (inv.ok()) CHECK(HostBool(pool, *inv->array()) == std::vector<int>({0, 1, 0, -1, 1, 0, 1, -1}));
    auto l9 = DevBool(pool, {1, 1, 1, 0, 0, 0, -1, -1, -1}), r9 = DevBool(pool, {1, 0, -1, 1, 0, -1, 1, 0, -1});
    auto an = CallFunction("and_not", {Datum(l9), Datum(r9)}, &ctx);
    CHECK_OK(an.status());
    if (an.ok()) CHECK(HostBool(pool, *an->array()) == std::vector<int>({0, 1, -1, 0, 0, -1, -1, -1, -1}));
    auto ank = CallFunction("and_not_kleene", {Datum(l9), Datum(r9)}, &ctx);
    CHECK_OK(ank.status());
    if (ank.ok()) CHECK(HostBool(pool, *ank->array()) == std::vector<int>({0, 1, -1, 0, 0, 0, 0, -1, -1}));
    // a null scalar: Kleene logic keeps what the array decides alone (:178-190); "and" nulls everything
    auto null_scalar = Datum(Scalar::MakeNull(boolean()));
    auto ks = CallFunction("and_kleene", {Datum(left), null_scalar}, &ctx);
    CHECK_OK(ks.status());
    if (ks.ok()) CHECK(HostBool(pool, *ks->array()) == std::vector<int>({-1, -1, -1, 0, 0, -1}));
    auto ps = CallFunction("and", {Datum(left), null_scalar}, &ctx);
    CHECK_OK(ps.status());
    if (ps.ok()) CHECK(HostBool(pool, *ps->array()) == std::vector<int>({-1, -1, -1, -1, -1, -1}));
    auto no_nulls = CallFunction("or_kleene", {Datum(DevBool(pool, {1, 1, 0, 0})), Datum(DevBool(pool, {1, 0, 0, 1}))}, &ctx);
    CHECK_OK(no_nulls.status());
    if (no_nulls.ok()) CHECK(no_nulls->array()->buffers[0] == nullptr && HostBool(pool, *no_nulls->array()) == std::vector<int>({1, 1, 0, 1}));
    CHECK(CallFunction("and", {Datum(left), Datum(Dev<int32_t>(pool, int32(), {V(int32_t, 1)}))}, &ctx).status().IsNotImplemented());
    // is_valid / is_null / is_nan
    auto v4 = Dev<int32_t>(pool, int32(), {N(int32_t), V(int32_t, 1), V(int32_t, 0), N(int32_t)});
    auto iv = CallFunction("is_valid", {Datum(v4)}, &ctx);
    CHECK_OK(iv.status());
    if (iv.ok()) CHECK(iv->array()->buffers[0] == nullptr && HostBool(pool, *iv->array()) == std::vector<int>({0, 1, 1, 0}));
    auto in = CallFunction("is_null", {Datum(v4)}, &ctx);
    CHECK_OK(in.status());
    if (in.ok()) CHECK(in->array()->null_count == 0 && HostBool(pool, *in->array()) == std::vector<int>({1, 0, 0, 1}));
    const double nan = std::nan("");
    auto nn = CallFunction("is_nan", {Datum(Dev<double>(pool, float64(), {V(double, 0.0), V(double, nan), V(double, 2.0), V(double, nan), V(double, 4.0), N(double)}))}, &ctx);
    CHECK_OK(nn.status());
    if (nn.ok()) CHECK(HostBool(pool, *nn->array()) == std::vector<int>({0, 1, 0, 1, 0, -1}));
    CHECK(CallFunction("is_nan", {Datum(v4)}, &ctx).status().IsNotImplemented());
    // fill_null
    auto f5 = CallFunction("fill_null", {Datum(Dev<int32_t>(pool, int32(), {N(int32_t), V(int32_t, 4), N(int32_t), V(int32_t, 8)})),
                                         Datum(Scalar::Make(int32(), int32_t(5)))}, &ctx);
    CHECK_OK(f5.status());
    if (f5.ok()) {
      CHECK(f5->array()->null_count == 0 && f5->array()->buffers[0] == nullptr);  // FillNullSetsZeroNullCount :175-183
      CHECK(Same(Host<int32_t>(pool, *f5->array()), {V(int32_t, 5), V(int32_t, 4), V(int32_t, 5), V(int32_t, 8)}));
    }
    auto fb = CallFunction("fill_null", {Datum(DevBool(pool, {1, -1, 0, -1})), Datum(Scalar::Make(boolean(), uint8_t(1)))}, &ctx);
    CHECK_OK(fb.status());
    if (fb.ok()) CHECK(HostBool(pool, *fb->array()) == std::vector<int>({1, 1, 0, 1}));
    auto fn = CallFunction("fill_null", {Datum(Dev<int8_t>(pool, int8(), {V(int8_t, 1), N(int8_t), V(int8_t, 3), V(int8_t, 2)})),
                                         Datum(Scalar::MakeNull(int8()))}, &ctx);
    CHECK_OK(fn.status());
    if (fn.ok()) CHECK(Same(Host<int8_t>(pool, *fn->array()), {V(int8_t, 1), N(int8_t), V(int8_t, 3), V(int8_t, 2)}));  // FillNullInvalidScalar :73-78
    // the predicate pipeline these kernels exist for: filter(x, and_kleene(greater(x, 1), is_valid(y)))
    auto x = Dev<int32_t>(pool, int32(), {V(int32_t, 0), V(int32_t, 2), V(int32_t, 3), N(int32_t), V(int32_t, 5)});
    auto y = Dev<int32_t>(pool, int32(), {V(int32_t, 1), N(int32_t), V(int32_t, 1), V(int32_t, 1), V(int32_t, 1)});
    auto gt = Compare(Datum(x), Datum(Scalar::Make(int32(), int32_t(1))), GREATER, &ctx);
    CHECK_OK(gt.status());
    auto yv = CallFunction("is_valid", {Datum(y)}, &ctx);
    CHECK_OK(yv.status());
    if (gt.ok() && yv.ok()) {
      auto m = CallFunction("and_kleene", {Datum(*gt), Datum(*yv)}, &ctx);
      CHECK_OK(m.status());
      if (m.ok()) {
        auto fr = Filter(Datum(x), *m, FilterOptions::Defaults(), &ctx);
        CHECK_OK(fr.status());
        if (fr.ok()) CHECK(Same(Host<int32_t>(pool, *fr->array()), {V(int32_t, 3), V(int32_t, 5)}));
      }
    }
  }
  {  // scalar aggregates: aggregate_test.cc:178-212 SimpleSum, :399-418 SimpleMean, :334-343 SimpleCount, :556-611 MinMax
    auto v7 = Dev<int32_t>(pool, int32(), {V(int32_t, 1), N(int32_t), V(int32_t, 3), N(int32_t), V(int32_t, 3), N(int32_t), V(int32_t, 7)});
    auto sum = CallFunction("sum", {Datum(v7)}, &ctx);
    CHECK_OK(sum.status());
    if (sum.ok()) CHECK(sum->is_scalar() && sum->scalar()->is_valid && sum->scalar()->type->id() == Type::INT64 && static_cast<int64_t>(sum->scalar()->bits) == 14);
    auto chunked = std::make_shared<ChunkedArray>(std::vector<std::shared_ptr<ArrayData>>{Dev<int32_t>(pool, int32(), {V(int32_t, 0), V(int32_t, 1), V(int32_t, 2)}),
                                                                                          Dev<int32_t>(pool, int32(), {}),
                                                                                          Dev<int32_t>(pool, int32(), {V(int32_t, 3), V(int32_t, 4), V(int32_t, 5)})}, int32());
    auto csum = CallFunction("sum", {Datum(chunked)}, &ctx);
    CHECK_OK(csum.status());
    if (csum.ok()) CHECK(static_cast<int64_t>(csum->scalar()->bits) == 15);
    auto nsum = CallFunction("sum", {Datum(Dev<int32_t>(pool, int32(), {N(int32_t)}))}, &ctx);
    CHECK_OK(nsum.status());
    if (nsum.ok()) CHECK(!nsum->scalar()->is_valid);  // no valid value -> null (SumImpl::Finalize :80-87)
    auto fsum = CallFunction("sum", {Datum(Dev<float>(pool, float32(), {V(float, 0.5f), V(float, 1.25f), N(float)}))}, &ctx);
    CHECK_OK(fsum.status());
    if (fsum.ok()) { double d; memcpy(&d, &fsum->scalar()->bits, 8); CHECK(fsum->scalar()->type->id() == Type::DOUBLE && d == 1.75); }
    auto bsum = CallFunction("sum", {Datum(DevBool(pool, {1, 0, 1, 1, -1}))}, &ctx);
    CHECK_OK(bsum.status());
    if (bsum.ok()) CHECK(bsum->scalar()->type->id() == Type::UINT64 && bsum->scalar()->bits == 3);
    auto mean = CallFunction("mean", {Datum(Dev<int8_t>(pool, int8(), {V(int8_t, 1), V(int8_t, 2), V(int8_t, 3), V(int8_t, 4), V(int8_t, 5), V(int8_t, 6), V(int8_t, 7), V(int8_t, 8)}))}, &ctx);
    CHECK_OK(mean.status());
    if (mean.ok()) { double d; memcpy(&d, &mean->scalar()->bits, 8); CHECK(d == 4.5); }
    auto bmean = CallFunction("mean", {Datum(DevBool(pool, {1, -1, 0, 1, 1}))}, &ctx);
    CHECK_OK(bmean.status());
    if (bmean.ok()) { double d; memcpy(&d, &bmean->scalar()->bits, 8); CHECK(d == 0.75); }
    CountOptions count_nulls(CountOptions::COUNT_NULL);
    auto c1 = CallFunction("count", {Datum(v7)}, &ctx);
    auto c2 = CallFunction("count", {Datum(v7)}, &count_nulls, &ctx);
    CHECK_OK(c1.status());
    CHECK_OK(c2.status());
    if (c1.ok() && c2.ok()) CHECK(static_cast<int64_t>(c1->scalar()->bits) == 4 && static_cast<int64_t>(c2->scalar()->bits) == 3);
    // min_max: SKIP / EMIT_NULL, NaN ignored, merged across batches of 2 rows
    auto f5 = Dev<double>(pool, float64(), {V(double, 5), V(double, std::nan("")), V(double, 2), N(double), V(double, 4)});
    ctx.set_exec_chunksize(2);
    auto mm = CallFunction("min_max", {Datum(f5)}, &ctx);
    ctx.set_exec_chunksize(kDefaultExecChunksize);
    CHECK_OK(mm.status());
    if (mm.ok()) {
      const Scalar& st = *mm->scalar();
      CHECK(st.type->id() == Type::STRUCT && st.fields.size() == 2 && st.field_names == std::vector<std::string>({"min", "max"}));
      double mn, mx;
      memcpy(&mn, &st.fields[0]->bits, 8); memcpy(&mx, &st.fields[1]->bits, 8);
      CHECK(st.fields[0]->is_valid && mn == 2 && mx == 5);
    }
    MinMaxOptions emit(MinMaxOptions::EMIT_NULL);
    auto mme = CallFunction("min_max", {Datum(f5)}, &emit, &ctx);
    CHECK_OK(mme.status());
    if (mme.ok()) CHECK(!mme->scalar()->fields[0]->is_valid && !mme->scalar()->fields[1]->is_valid);
    auto mi = CallFunction("min_max", {Datum(Dev<int16_t>(pool, int16(), {V(int16_t, 5), V(int16_t, -1), V(int16_t, 2), V(int16_t, 3), V(int16_t, 4)}))}, &ctx);
    CHECK_OK(mi.status());
    if (mi.ok()) CHECK(static_cast<int16_t>(mi->scalar()->fields[0]->bits) == -1 && static_cast<int16_t>(mi->scalar()->fields[1]->bits) == 5);
    auto mb = CallFunction("min_max", {Datum(DevBool(pool, {1, -1, 0, 1}))}, &ctx);
    CHECK_OK(mb.status());
    if (mb.ok()) CHECK(mb->scalar()->fields[0]->bits == 0 && mb->scalar()->fields[1]->bits == 1);
    auto me = CallFunction("min_max", {Datum(Dev<int16_t>(pool, int16(), {}))}, &ctx);
    CHECK_OK(me.status());
    if (me.ok()) CHECK(!me->scalar()->fields[0]->is_valid);
  }
  {  // GroupBy's fused device path (one key column, one batch) returns the rows of the batch-wise Grouper path
    const int64_t n = 400000;
    std::vector<int64_t> vals(n);
    std::vector<int32_t> keys(n);
    std::vector<uint8_t> vvalid((n + 7) / 8 + 1, 0), kvalid((n + 7) / 8 + 1, 0);
    uint64_t x = 12345;
    int64_t vnulls = 0, knulls = 0;
    for (int64_t i = 0; i < n; ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      vals[i] = static_cast<int64_t>(x % 2001) - 1000;
      keys[i] = static_cast<int32_t>((x >> 24) % 30000) - 15000;
      if ((x >> 60) != 0) vvalid[i / 8] |= uint8_t(1u << (i % 8)); else ++vnulls;
      if (((x >> 50) & 63) != 0) kvalid[i / 8] |= uint8_t(1u << (i % 8)); else ++knulls;
    }
    auto v = ArrayFromHost(pool, int64(), vals.data(), n, vvalid.data(), vnulls);
    auto k = ArrayFromHost(pool, int32(), keys.data(), n, kvalid.data(), knulls);
    CHECK(v.ok() && k.ok());
    if (v.ok() && k.ok()) {
      CountOptions co;
      const std::vector<internal::Aggregate> aggs = {{"hash_sum", nullptr}, {"hash_count", &co}, {"hash_mean", nullptr}, {"hash_count", &co}};
      const std::vector<Datum> args = {Datum(*v), Datum(*v), Datum(*v), Datum(*k)};
      unsetenv("AMDARROW_GROUPBY_BATCHWISE");
      auto fused = internal::GroupBy(args, {Datum(*k)}, aggs, &ctx);
      setenv("AMDARROW_GROUPBY_BATCHWISE", "1", 1);
      auto batch = internal::GroupBy(args, {Datum(*k)}, aggs, &ctx);
      unsetenv("AMDARROW_GROUPBY_BATCHWISE");
      CHECK_OK(fused.status());
      CHECK_OK(batch.status());
      if (fused.ok() && batch.ok()) {
        CHECK(fused->length == batch->length && fused->length == 30001 && fused->field_names == batch->field_names);
        CHECK(Same(Host<int64_t>(pool, *fused->columns[0]), Host<int64_t>(pool, *batch->columns[0])));
        CHECK(Same(Host<int64_t>(pool, *fused->columns[1]), Host<int64_t>(pool, *batch->columns[1])));
        CHECK(Same(Host<double>(pool, *fused->columns[2]), Host<double>(pool, *batch->columns[2])));
        CHECK(Same(Host<int64_t>(pool, *fused->columns[3]), Host<int64_t>(pool, *batch->columns[3])));
        CHECK(Same(Host<int32_t>(pool, *fused->columns[4]), Host<int32_t>(pool, *batch->columns[4])));
        CHECK((fused->columns[4]->buffers[0] != nullptr) == (batch->columns[4]->buffers[0] != nullptr));
        CHECK(fused->columns[0]->null_count == batch->columns[0]->null_count);
      }
    }
  }
  TabularTests(pool, &ctx);
  PromotionTests(pool, &ctx);
  GroupingsTests(pool, &ctx);
  WideKeyGroupByTests(pool, &ctx);
  ExpressionTests(pool, &ctx);
  NestedSelectionTests(pool, &ctx);
  ShardedGroupByTests(pool, &ctx);
  CHECK(pool->bytes_allocated() >= 0 && pool->max_memory() > 0);  // the MemoryPool counters (memory_pool.h:38-105)
  pool->ReleaseUnused();
  CHECK(pool->cached_bytes() == 0);
  printf("host_test gpu: %d checks, %d failures\n", g_checks, g_failures);
  return g_failures ? 1 : 0;
}

// host_test bench [rows]: the drop-in API timed end to end -- Filter + GroupBy through CallFunction /
// internal::GroupBy on `rows` int64 values, int32 keys over 1 M groups, a 50 % mask.
static int RunBench(int64_t n) {
  auto pool_r = MemoryPool::Make(0);
  if (!pool_r.ok()) { fprintf(stderr, "no device: %s\n", pool_r.status().ToString().c_str()); return 2; }
  auto pool = *pool_r;
  ExecContext ctx(pool);
  std::vector<int64_t> vals(n);
  std::vector<int32_t> keys(n);
  std::vector<uint8_t> mask((n + 7) / 8 + 8, 0);
  uint64_t x = 0x0ff1ce;
  auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int64_t i = 0; i < n; ++i) {
    const uint64_t r = next();
    vals[i] = static_cast<int64_t>(r % 201) - 100;
    keys[i] = static_cast<int32_t>((r >> 20) % 1000000);
    if ((r >> 50) & 1) mask[i / 8] |= uint8_t(1u << (i % 8));
  }
  auto v = ArrayFromHost(pool, int64(), vals.data(), n, nullptr, 0);
  auto k = ArrayFromHost(pool, int32(), keys.data(), n, nullptr, 0);
  auto m = ArrayFromHost(pool, boolean(), mask.data(), n, nullptr, 0);
  if (!v.ok() || !k.ok() || !m.ok()) return 2;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int rep = 0; rep < 4; ++rep) {
    const double t0 = now();
    auto fv = Filter(Datum(*v), Datum(*m), FilterOptions::Defaults(), &ctx);
    auto fk = Filter(Datum(*k), Datum(*m), FilterOptions::Defaults(), &ctx);
    if (!fv.ok() || !fk.ok()) { fprintf(stderr, "filter failed\n"); return 1; }
    const double t1 = now();
    CountOptions co;
    auto r = internal::GroupBy({*fv, *fv}, {*fk}, {{"hash_sum", nullptr}, {"hash_count", &co}}, &ctx);
    if (!r.ok()) { fprintf(stderr, "GroupBy failed: %s\n", r.status().ToString().c_str()); return 1; }
    (void)amdarrow_ctx_synchronize(pool->ctx());
    const double t2 = now();
    printf("rep %d: Filter x2 %.3f ms, GroupBy{hash_sum,hash_count} %.3f ms, groups %lld\n", rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
           static_cast<long long>(r->length));
    if (rep >= 2) {  // an aggregate the fused pass does not carry: group ids behind the fused grouping vs the batch-wise path
      for (int mode = 0; mode < 2; ++mode) {
        if (mode) setenv("AMDARROW_GROUPBY_BATCHWISE", "1", 1);
        const double t3 = now();
        auto r2 = internal::GroupBy({*fv, *fv}, {*fk}, {{"hash_sum", nullptr}, {"hash_min_max", nullptr}}, &ctx);
        (void)amdarrow_ctx_synchronize(pool->ctx());
        const double t4 = now();
        if (mode) unsetenv("AMDARROW_GROUPBY_BATCHWISE");
        if (!r2.ok()) { fprintf(stderr, "GroupBy failed: %s\n", r2.status().ToString().c_str()); return 1; }
        printf("        GroupBy{hash_sum,hash_min_max} %s %.3f ms\n", mode ? "batch-wise" : "fused+ids ", (t4 - t3) * 1e3);
      }
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "bench")) return RunBench(argc >= 3 ? atoll(argv[2]) : 100000000);
  if (argc >= 2 && !strcmp(argv[1], "cpu")) return RunCpu();
  if (argc >= 2 && !strcmp(argv[1], "gpu")) return RunGpu();
  fprintf(stderr, "usage: host_test cpu|gpu\n");
  return 64;
}

// arrow_mi355x_plugin.cc -- the reference-side binding of INTEGRATION.md section 2, COMPILED: built against the
// reference's own headers (/root/reference/cpp/src), linked against arrow-1_amd/libamdarrow.so (the C ABI of
// include/amdarrow.h) and the reference library built by oracle/ref_build.mk.  TEST INFRASTRUCTURE: it shows -- and
// tests -- what a maintainer of the reference writes to install the MI355X kernels under the reference's own function
// names in an arrow::compute::FunctionRegistry; nothing of it ships in the product.
//
//   plugin_test cpu   (build container, no GPU): MakeRegistry() registers array_filter / array_take /
//                     array_sort_indices / equal .. less_equal / add .. multiply_checked under the reference's names;
//                     GetFunction + DispatchExact resolve them for every type the reference registers them for.
//   plugin_test gpu   (GPU box): the REFERENCE'S CallFunction("filter" / "take" / "sort_indices" / "less" / "add") runs
//                     once with its own CPU registry and once with an ExecContext bound to this registry -- its
//                     MetaFunctions route unchanged, the array_* kernels are the HIP ones -- on the same buffers
//                     (pinned host memory, which both sides can address), and the two results are compared bit for bit.
#include <cstdio>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "arrow/array.h"
#include "arrow/array/data.h"
#include "arrow/buffer.h"
#include "arrow/compute/api_scalar.h"
#include "arrow/compute/api_vector.h"
#include "arrow/compute/exec.h"
#include "arrow/compute/function.h"
#include "arrow/compute/kernel.h"
#include "arrow/compute/kernels/codegen_internal.h"
#include "arrow/compute/registry.h"
#include "arrow/compute/registry_internal.h"
#include "arrow/datum.h"
#include "arrow/memory_pool.h"
#include "arrow/type.h"
#include "arrow/util/bit_util.h"

#include "amdarrow.h"

namespace arrow {
namespace compute {
namespace mi355x {

static Status FromAbi(int rc) {
  return rc == 0 ? Status::OK() : Status(static_cast<StatusCode>(rc), amdarrow_last_error());
}
// ArrayData -> PrimitiveArg-shaped view; the buffers must be addressable by the device
static amdarrow_array ToAbi(const ArrayData& a) {
  amdarrow_array d;
  d.values = a.buffers.size() > 1 && a.buffers[1] ? a.buffers[1]->data() : nullptr;
  d.validity = a.buffers[0] ? a.buffers[0]->data() : nullptr;
  d.length = a.length;
  d.offset = a.offset;
  d.type_id = static_cast<int32_t>(a.type->id());  // the ABI's type ids ARE arrow::Type::type (type_fwd.h:277-313)
  d.null_count = a.null_count.load();
  return d;
}
static int64_t ValueBytes(const DataType& t, int64_t n) {
  const int bw = ::arrow::internal::checked_cast<const FixedWidthType&>(t).bit_width();
  return bw == 1 ? BitUtil::BytesForBits(n) : n * (bw / 8);
}

// state shared by the kernels: one amdarrow_ctx per ExecContext
struct DeviceState { amdarrow_ctx* ctx = nullptr; };

// array_filter: PrimitiveFilter (vector_selection.cc:786-835)
static Status FilterExec(DeviceState* dev, KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto& opts = internal::OptionsWrapper<FilterOptions>::Get(ctx);
  const ArrayData& values = *batch[0].array();
  const ArrayData& filter = *batch[1].array();
  amdarrow_array v = ToAbi(values), f = ToAbi(filter);
  int64_t n = 0;
  RETURN_NOT_OK(FromAbi(amdarrow_filter_output_size(dev->ctx, &f, static_cast<int>(opts.null_selection_behavior), &n)));
  ArrayData* out_arr = out->mutable_array();
  // PreallocateData (:244-259): the validity bitmap only when an input may hold nulls (:809)
  const bool allocate_validity = (values.buffers[0] ? values.null_count.load() : 0) != 0 || (filter.buffers[0] ? filter.null_count.load() : 0) != 0;
  ARROW_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*values.type, n) + 8));
  std::shared_ptr<ResizableBuffer> nbuf;
  if (allocate_validity) { ARROW_ASSIGN_OR_RAISE(nbuf, ctx->AllocateBitmap(n + 64)); }
  amdarrow_out o;
  RETURN_NOT_OK(FromAbi(amdarrow_filter(dev->ctx, &v, &f, static_cast<int>(opts.null_selection_behavior), vbuf->mutable_data(),
                                        nbuf ? nbuf->mutable_data() : nullptr, n, &o)));
  RETURN_NOT_OK(FromAbi(amdarrow_ctx_synchronize(dev->ctx)));  // (the arrays are complete in stream order; this caller reads them on the host)
  out_arr->buffers.resize(2);
  out_arr->buffers[0] = o.has_validity ? std::shared_ptr<Buffer>(nbuf) : nullptr;
  out_arr->buffers[1] = std::move(vbuf);
  out_arr->length = o.length;
  out_arr->null_count = o.null_count;
  out_arr->type = values.type;
  return Status::OK();
}
// array_take: PrimitiveTake (:493-530)
static Status TakeExec(DeviceState* dev, KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto& opts = internal::OptionsWrapper<TakeOptions>::Get(ctx);
  const ArrayData& values = *batch[0].array();
  const ArrayData& indices = *batch[1].array();
  amdarrow_array v = ToAbi(values), i = ToAbi(indices);
  ARROW_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*values.type, indices.length) + 8));
  ARROW_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(indices.length + 64));
  amdarrow_out o;
  RETURN_NOT_OK(FromAbi(amdarrow_take(dev->ctx, &v, &i, opts.boundscheck, vbuf->mutable_data(), nbuf->mutable_data(), &o)));
  ArrayData* out_arr = out->mutable_array();
  out_arr->buffers = {std::shared_ptr<Buffer>(nbuf), std::shared_ptr<Buffer>(vbuf)};
  out_arr->length = o.length;
  out_arr->null_count = o.null_count;
  out_arr->type = values.type;
  return Status::OK();
}
// array_sort_indices (vector_sort.cc:558-575): PREALLOCATE + OUTPUT_NOT_NULL, the executor hands the uint64 buffer over
static Status SortExec(DeviceState* dev, KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  const auto& opts = internal::OptionsWrapper<ArraySortOptions>::Get(ctx);
  amdarrow_array v = ToAbi(*batch[0].array());
  ArrayData* out_arr = out->mutable_array();
  uint64_t* dst = out_arr->GetMutableValues<uint64_t>(1);
  return FromAbi(amdarrow_sort_indices(dev->ctx, &v, 1, opts.order == SortOrder::Descending ? AMDARROW_DESCENDING : AMDARROW_ASCENDING, dst));
}
// equal .. less_equal (scalar_compare.cc:100-171) and add .. multiply_checked (scalar_arithmetic.cc:69-203): the device
// kernel computes values AND validity (NullPropagator's result), so the kernels are registered COMPUTED_NO_PREALLOCATE
static Status BinaryExec(DeviceState* dev, bool compare, int op, KernelContext* ctx, const ExecBatch& batch, Datum* out) {
  if (!batch[0].is_array() || !batch[1].is_array()) return Status::NotImplemented("plugin: array (x) array only");
  const ArrayData& l = *batch[0].array();
  const ArrayData& r = *batch[1].array();
  amdarrow_array a = ToAbi(l), b = ToAbi(r);
  const int64_t n = batch.length;
  auto out_type = compare ? boolean() : l.type;
  ARROW_ASSIGN_OR_RAISE(auto vbuf, ctx->Allocate(ValueBytes(*out_type, n) + 8));
  ARROW_ASSIGN_OR_RAISE(auto nbuf, ctx->AllocateBitmap(n + 64));
  amdarrow_out o;
  const int rc = compare ? amdarrow_compare(dev->ctx, op, &a, nullptr, &b, nullptr, vbuf->mutable_data(), nbuf->mutable_data(), &o)
                         : amdarrow_arith(dev->ctx, op, &a, nullptr, &b, nullptr, vbuf->mutable_data(), nbuf->mutable_data(), &o);
  RETURN_NOT_OK(FromAbi(rc));
  RETURN_NOT_OK(FromAbi(amdarrow_ctx_synchronize(dev->ctx)));  // (values are complete in stream order; this caller reads them on the host)
  ArrayData* out_arr = out->mutable_array();
  out_arr->buffers = {o.has_validity ? std::shared_ptr<Buffer>(nbuf) : nullptr, std::shared_ptr<Buffer>(vbuf)};
  out_arr->length = o.length;
  out_arr->null_count = o.null_count;
  out_arr->type = out_type;
  return Status::OK();
}

static const FunctionDoc kDoc{"MI355X-backed kernel", "see include/amdarrow.h", {"a", "b"}};
static const FunctionDoc kDoc1{"MI355X-backed kernel", "see include/amdarrow.h", {"a"}};
static const FilterOptions kFilterDefaults = FilterOptions::Defaults();
static const TakeOptions kTakeDefaults = TakeOptions::Defaults();
static const ArraySortOptions kSortDefaults = ArraySortOptions::Defaults();

// One registry holding hot-path functions under the SAME names, device-backed.  The MetaFunctions "filter", "take",
// "sort_indices" are the reference's own: they only route by Datum kind and call CallFunction("array_*", ..., ctx) on
// the same ExecContext (vector_selection.cc:1877-1903, 2026-2072; vector_sort.cc:1565-1711).
std::unique_ptr<FunctionRegistry> MakeRegistry(DeviceState* dev) {
  auto registry = FunctionRegistry::Make();
  // the reference's meta functions + its other kernels first; the device-backed ones then replace array_* / compare / arithmetic
  internal::RegisterVectorSelection(registry.get());
  internal::RegisterVectorSort(registry.get());
  internal::RegisterScalarComparison(registry.get());
  internal::RegisterScalarArithmetic(registry.get());
  {
    auto f = std::make_shared<VectorFunction>("array_filter", Arity::Binary(), &kDoc, &kFilterDefaults);
    VectorKernel k(std::vector<InputType>{InputType(match::Primitive(), ValueDescr::ARRAY), InputType::Array(boolean())}, OutputType(internal::FirstType),
                   ArrayKernelExec([dev](KernelContext* c, const ExecBatch& b, Datum* o) { return FilterExec(dev, c, b, o); }),
                   internal::OptionsWrapper<FilterOptions>::Init);
    DCHECK_OK(f->AddKernel(std::move(k)));
    DCHECK_OK(registry->AddFunction(std::move(f), /*allow_overwrite=*/true));
  }
  {
    auto f = std::make_shared<VectorFunction>("array_take", Arity::Binary(), &kDoc, &kTakeDefaults);
    VectorKernel k(std::vector<InputType>{InputType(match::Primitive(), ValueDescr::ARRAY), InputType(match::Integer(), ValueDescr::ARRAY)}, OutputType(internal::FirstType),
                   ArrayKernelExec([dev](KernelContext* c, const ExecBatch& b, Datum* o) { return TakeExec(dev, c, b, o); }),
                   internal::OptionsWrapper<TakeOptions>::Init);
    k.can_execute_chunkwise = false;  // vector_selection.cc:2156
    DCHECK_OK(f->AddKernel(std::move(k)));
    DCHECK_OK(registry->AddFunction(std::move(f), true));
  }
  {
    auto f = std::make_shared<VectorFunction>("array_sort_indices", Arity::Unary(), &kDoc1, &kSortDefaults);
    for (const auto& ty : ::arrow::compute::internal::NumericTypes()) {
      VectorKernel k(std::vector<InputType>{InputType::Array(ty)}, OutputType(uint64()),
                     ArrayKernelExec([dev](KernelContext* c, const ExecBatch& b, Datum* o) { return SortExec(dev, c, b, o); }),
                     internal::OptionsWrapper<ArraySortOptions>::Init);
      k.mem_allocation = MemAllocation::PREALLOCATE;       // vector_sort.cc:1745-1747
      k.null_handling = NullHandling::OUTPUT_NOT_NULL;
      k.can_execute_chunkwise = false;
      DCHECK_OK(f->AddKernel(std::move(k)));
    }
    DCHECK_OK(registry->AddFunction(std::move(f), true));
  }
  const char* cmp_names[] = {"equal", "not_equal", "greater", "greater_equal", "less", "less_equal"};
  for (int op = 0; op < 6; ++op) {
    auto f = std::make_shared<ScalarFunction>(cmp_names[op], Arity::Binary(), &kDoc);
    for (const auto& ty : ::arrow::compute::internal::NumericTypes()) {
      ScalarKernel k(std::vector<InputType>{InputType::Array(ty), InputType::Array(ty)}, OutputType(boolean()),
                     ArrayKernelExec([dev, op](KernelContext* c, const ExecBatch& b, Datum* o) { return BinaryExec(dev, true, op, c, b, o); }));
      k.null_handling = NullHandling::COMPUTED_NO_PREALLOCATE;
      k.mem_allocation = MemAllocation::NO_PREALLOCATE;
      k.can_write_into_slices = false;
      DCHECK_OK(f->AddKernel(std::move(k)));
    }
    DCHECK_OK(registry->AddFunction(std::move(f), true));
  }
  struct { const char* name; int op; } arith[] = {{"add", AMDARROW_ADD}, {"subtract", AMDARROW_SUB}, {"multiply", AMDARROW_MUL},
                                                 {"add_checked", AMDARROW_ADD_CHECKED}, {"subtract_checked", AMDARROW_SUB_CHECKED},
                                                 {"multiply_checked", AMDARROW_MUL_CHECKED}};
  for (const auto& a : arith) {
    auto f = std::make_shared<ScalarFunction>(a.name, Arity::Binary(), &kDoc);
    const int op = a.op;
    for (const auto& ty : ::arrow::compute::internal::NumericTypes()) {
      ScalarKernel k(std::vector<InputType>{InputType::Array(ty), InputType::Array(ty)}, OutputType(internal::FirstType),
                     ArrayKernelExec([dev, op](KernelContext* c, const ExecBatch& b, Datum* o) { return BinaryExec(dev, false, op, c, b, o); }));
      k.null_handling = NullHandling::COMPUTED_NO_PREALLOCATE;
      k.mem_allocation = MemAllocation::NO_PREALLOCATE;
      k.can_write_into_slices = false;
      DCHECK_OK(f->AddKernel(std::move(k)));
    }
    DCHECK_OK(registry->AddFunction(std::move(f), true));
  }
  return registry;
}

}  // namespace mi355x
}  // namespace compute
}  // namespace arrow

// ---------------------------------------------------------------------------------------------------------------------
using namespace arrow;           // NOLINT
using namespace arrow::compute;  // NOLINT

static int g_checks = 0, g_failures = 0;
#define CHECK(cond) do { ++g_checks; if (!(cond)) { ++g_failures; fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); } } while (0)

static int RunCpu() {
  mi355x::DeviceState dev;  // no device here: registration and dispatch only, no kernel runs
  auto registry = mi355x::MakeRegistry(&dev);
  for (const char* name : {"array_filter", "array_take", "array_sort_indices", "filter", "take", "sort_indices", "equal", "not_equal", "greater",
                           "greater_equal", "less", "less_equal", "add", "subtract", "multiply", "add_checked", "subtract_checked",
                           "multiply_checked"}) {
    auto f = registry->GetFunction(name);
    CHECK(f.ok());
  }
  CHECK(registry->GetFunction("no_such_function").status().IsKeyError());
  auto check_dispatch = [&](const char* name, const std::vector<ValueDescr>& args, bool want) {
    auto f = registry->GetFunction(name);
    if (!f.ok()) { CHECK(false); return; }
    auto k = (*f)->DispatchExact(args);
    CHECK(k.ok() == want);
  };
  std::vector<std::shared_ptr<DataType>> prim = ::arrow::compute::internal::NumericTypes();
  prim.push_back(boolean());
  for (const auto& ty : prim) {
    check_dispatch("array_filter", std::vector<ValueDescr>{ValueDescr::Array(ty), ValueDescr::Array(boolean())}, true);
    for (const auto& it : ::arrow::compute::internal::IntTypes()) check_dispatch("array_take", std::vector<ValueDescr>{ValueDescr::Array(ty), ValueDescr::Array(it)}, true);
  }
  for (const auto& ty : ::arrow::compute::internal::NumericTypes()) {
    check_dispatch("array_sort_indices", std::vector<ValueDescr>{ValueDescr::Array(ty)}, true);
    for (const char* name : {"equal", "less", "greater_equal", "add", "multiply_checked"}) check_dispatch(name, std::vector<ValueDescr>{ValueDescr::Array(ty), ValueDescr::Array(ty)}, true);
  }
  check_dispatch("array_filter", std::vector<ValueDescr>{ValueDescr::Array(int64()), ValueDescr::Array(int64())}, false);   // the filter must be boolean
  check_dispatch("array_take", std::vector<ValueDescr>{ValueDescr::Array(int64()), ValueDescr::Array(float64())}, false);  // indices must be integers
  check_dispatch("array_sort_indices", std::vector<ValueDescr>{ValueDescr::Array(boolean())}, false);
  CHECK((*registry->GetFunction("array_filter"))->num_kernels() == 1 && (*registry->GetFunction("array_sort_indices"))->num_kernels() == 10);
  CHECK((*registry->GetFunction("filter"))->kind() == Function::META);  // the reference's own meta function, reused
  ExecContext ctx(default_memory_pool(), registry.get());
  CHECK(ctx.func_registry() == registry.get());
  printf("plugin_test cpu: %d checks, %d failures\n", g_checks, g_failures);
  return g_failures != 0;
}

// arrow::MemoryPool over pinned, device-addressable host memory: the reference's executor (validity propagation, result
// checks) and the HIP kernels both address it -- what a Mi355xMemoryManager (device.h:40-151) does with HBM + copies.
class PinnedPool : public MemoryPool {
 public:
  Status Allocate(int64_t size, uint8_t** out) override {
    void* p = nullptr;
    if (hipHostMalloc(&p, static_cast<size_t>(size < 64 ? 64 : size), hipHostMallocDefault) != hipSuccess) return Status::OutOfMemory("hipHostMalloc");
    *out = static_cast<uint8_t*>(p);
    bytes_ += size;
    if (bytes_ > max_) max_ = bytes_;
    return Status::OK();
  }
  Status Reallocate(int64_t old_size, int64_t new_size, uint8_t** ptr) override {
    uint8_t* n = nullptr;
    RETURN_NOT_OK(Allocate(new_size, &n));
    memcpy(n, *ptr, static_cast<size_t>(old_size < new_size ? old_size : new_size));
    Free(*ptr, old_size);
    *ptr = n;
    return Status::OK();
  }
  void Free(uint8_t* buffer, int64_t size) override { (void)hipHostFree(buffer); bytes_ -= size; }
  int64_t bytes_allocated() const override { return bytes_; }
  int64_t max_memory() const override { return max_; }
  std::string backend_name() const override { return "hip-pinned"; }

 private:
  int64_t bytes_ = 0, max_ = 0;
};

template <typename T>
static std::shared_ptr<ArrayData> MakeArrayData(MemoryPool* pool, std::shared_ptr<DataType> type, int64_t n, double p_null, std::mt19937_64* rng,
                                                int64_t lo, int64_t hi) {
  auto vbuf = *AllocateBuffer(static_cast<int64_t>(sizeof(T)) * n + 64, pool);
  auto nbuf = *AllocateBitmap(n + 64, pool);
  memset(nbuf->mutable_data(), 0, static_cast<size_t>(nbuf->size()));
  T* v = reinterpret_cast<T*>(vbuf->mutable_data());
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; ++i) {
    v[i] = static_cast<T>(lo + static_cast<int64_t>((*rng)() % static_cast<uint64_t>(hi - lo + 1)));
    const bool valid = p_null == 0 || (((*rng)() >> 11) * (1.0 / 9007199254740992.0)) >= p_null;
    if (valid) BitUtil::SetBit(nbuf->mutable_data(), i); else ++nulls;
  }
  return ArrayData::Make(type, n, {p_null == 0 ? nullptr : std::shared_ptr<Buffer>(std::move(nbuf)), std::shared_ptr<Buffer>(std::move(vbuf))}, nulls);
}
static std::shared_ptr<ArrayData> MakeMask(MemoryPool* pool, int64_t n, double p_true, double p_null, std::mt19937_64* rng) {
  auto vbuf = *AllocateBitmap(n + 64, pool);
  auto nbuf = *AllocateBitmap(n + 64, pool);
  memset(vbuf->mutable_data(), 0, static_cast<size_t>(vbuf->size()));
  memset(nbuf->mutable_data(), 0, static_cast<size_t>(nbuf->size()));
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; ++i) {
    if ((((*rng)() >> 11) * (1.0 / 9007199254740992.0)) < p_true) BitUtil::SetBit(vbuf->mutable_data(), i);
    if (p_null == 0 || (((*rng)() >> 11) * (1.0 / 9007199254740992.0)) >= p_null) BitUtil::SetBit(nbuf->mutable_data(), i); else ++nulls;
  }
  return ArrayData::Make(boolean(), n, {p_null == 0 ? nullptr : std::shared_ptr<Buffer>(std::move(nbuf)), std::shared_ptr<Buffer>(std::move(vbuf))}, nulls);
}
static bool SameArrays(const Datum& a, const Datum& b) {
  if (!a.is_array() || !b.is_array()) return false;
  return MakeArray(a.array())->Equals(*MakeArray(b.array()));
}

static int RunGpu() {
  mi355x::DeviceState dev;
  if (amdarrow_ctx_create(0, nullptr, 0, &dev.ctx) != 0) { fprintf(stderr, "no device: %s\n", amdarrow_last_error()); return 2; }
  PinnedPool pool;
  auto registry = mi355x::MakeRegistry(&dev);
  ExecContext gpu_ctx(&pool, registry.get());      // the drop-in: same CallFunction, another registry
  // the reference's own kernels on the same buffers (its default registry needs the generated config.h; the hot path's
  // Register* hooks are what oracle/ref_driver.cc uses too)
  auto cpu_registry = FunctionRegistry::Make();
  compute::internal::RegisterVectorSelection(cpu_registry.get());
  compute::internal::RegisterVectorSort(cpu_registry.get());
  compute::internal::RegisterScalarComparison(cpu_registry.get());
  compute::internal::RegisterScalarArithmetic(cpu_registry.get());
  ExecContext cpu_ctx(&pool, cpu_registry.get());
  std::mt19937_64 rng(0x0ff1ce);
  for (int64_t n : {1000, 70000, 300000}) {
    for (double p_null : {0.0, 0.05}) {
      auto values = MakeArrayData<int64_t>(&pool, int64(), n, p_null, &rng, -100, 100);
      auto other = MakeArrayData<int64_t>(&pool, int64(), n, p_null, &rng, -100, 100);
      auto mask = MakeMask(&pool, n, 0.5, p_null, &rng);
      for (auto sel : {FilterOptions::DROP, FilterOptions::EMIT_NULL}) {
        FilterOptions fo(sel);
        auto want = CallFunction("filter", {Datum(values), Datum(mask)}, &fo, &cpu_ctx);
        auto got = CallFunction("filter", {Datum(values), Datum(mask)}, &fo, &gpu_ctx);
        CHECK(want.ok() && got.ok());
        if (want.ok() && got.ok()) CHECK(SameArrays(*want, *got));
      }
      auto idx = MakeArrayData<int32_t>(&pool, int32(), n / 2 + 1, p_null, &rng, 0, n - 1);
      TakeOptions to;
      auto want = CallFunction("take", {Datum(values), Datum(idx)}, &to, &cpu_ctx);
      auto got = CallFunction("take", {Datum(values), Datum(idx)}, &to, &gpu_ctx);
      CHECK(want.ok() && got.ok());
      if (want.ok() && got.ok()) CHECK(SameArrays(*want, *got));
      for (auto order : {SortOrder::Ascending, SortOrder::Descending}) {
        SortOptions so({SortKey("", order)});
        ArraySortOptions ao(order);
        auto w = CallFunction("array_sort_indices", {Datum(values)}, &ao, &cpu_ctx);
        auto g = CallFunction("array_sort_indices", {Datum(values)}, &ao, &gpu_ctx);
        CHECK(w.ok() && g.ok());
        if (w.ok() && g.ok()) CHECK(SameArrays(*w, *g));
      }
      for (const char* fn : {"less", "equal", "add", "multiply_checked"}) {
        auto w = CallFunction(fn, {Datum(values), Datum(other)}, &cpu_ctx);
        auto g = CallFunction(fn, {Datum(values), Datum(other)}, &gpu_ctx);
        CHECK(w.ok() && g.ok());
        if (!w.ok()) fprintf(stderr, "  %s (reference kernels): %s\n", fn, w.status().ToString().c_str());
        if (!g.ok()) fprintf(stderr, "  %s (plugin registry): %s\n", fn, g.status().ToString().c_str());
        if (w.ok() && g.ok()) {
          CHECK(SameArrays(*w, *g));
          if (!SameArrays(*w, *g)) {
            auto x = w->array(), y = g->array();
            fprintf(stderr, "  %s n=%lld: type %s vs %s, length %lld vs %lld, nulls %lld vs %lld, validity %d vs %d\n", fn, (long long)n,
                    x->type->ToString().c_str(), y->type->ToString().c_str(), (long long)x->length, (long long)y->length,
                    (long long)x->GetNullCount(), (long long)y->GetNullCount(), x->buffers[0] != nullptr, y->buffers[0] != nullptr);
            auto ax = MakeArray(x), ay = MakeArray(y);
            for (int64_t i = 0, shown = 0; i < x->length && shown < 4; ++i)
              if (!ax->RangeEquals(i, i + 1, i, *ay)) { ++shown; fprintf(stderr, "    row %lld: %s vs %s\n", (long long)i, ax->GetScalar(i).ValueOrDie()->ToString().c_str(), ay->GetScalar(i).ValueOrDie()->ToString().c_str()); }
          }
        }
      }
    }
  }
  {  // errors travel as the reference's Status: a bad index is IndexError with the reference's text
    auto values = MakeArrayData<int64_t>(&pool, int64(), 10, 0.0, &rng, 0, 5);
    auto idx = MakeArrayData<int32_t>(&pool, int32(), 3, 0.0, &rng, 11, 11);
    TakeOptions to;
    auto got = CallFunction("take", {Datum(values), Datum(idx)}, &to, &gpu_ctx);
    auto want = CallFunction("take", {Datum(values), Datum(idx)}, &to, &cpu_ctx);
    CHECK(got.status().IsIndexError() && got.status().message() == want.status().message());
  }
  amdarrow_ctx_destroy(dev.ctx);
  printf("plugin_test gpu: %d checks, %d failures\n", g_checks, g_failures);
  return g_failures != 0;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "cpu";
  return mode == "gpu" ? RunGpu() : RunCpu();
}

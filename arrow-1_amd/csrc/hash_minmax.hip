// hash_minmax.hip -- hash_min_max: per-group minimum and maximum (GroupedMinMaxImpl,
// compute/kernels/hash_aggregate.cc:627-768), the last member of the aggregate family
// (SURVEY 8f.2).  Same split as the reference's HashAggregateKernel: caller-owned state that
// grows with the groups (resize -> init), consume (accumulate), finalize.
//
// State, exactly the reference's four builders (:704-707): mins / maxes in the VALUE type,
// new groups holding the anti-extreme (Extrema<T>::max() / ::min(), :629-642, :653-661), and the
// has_values / has_nulls bitmaps.  Rows reach their group's slots with atomics: native 32/64-bit
// integer min/max, a compare-and-swap on the containing word for 8/16-bit values and for floats.
// A float update happens only on a strict improvement, like std::min / std::max (:667-668): NaN
// never replaces anything, and of two zeros of different sign the one that is stored first stays --
// which of them that is depends on row order in the reference and on scheduling here, so float
// extremes are equal to the reference's as VALUES (-0.0 == +0.0), everything else bit for bit.
#include "common.h"

#include <cmath>
#include <limits>
#include <type_traits>

namespace amdarrow {

constexpr int kMThreads = 256;

template <typename T> struct Extrema {
  __host__ __device__ static T lowest() { return std::numeric_limits<T>::min(); }
  __host__ __device__ static T highest() { return std::numeric_limits<T>::max(); }
};
template <> struct Extrema<float> {
  __host__ __device__ static float lowest() { return -INFINITY; }
  __host__ __device__ static float highest() { return INFINITY; }
};
template <> struct Extrema<double> {
  __host__ __device__ static double lowest() { return -static_cast<double>(INFINITY); }
  __host__ __device__ static double highest() { return static_cast<double>(INFINITY); }
};

template <typename T>
__global__ __launch_bounds__(kMThreads) void k_minmax_init(T* mins, T* maxes, int64_t first, int64_t count) {
  for (int64_t i = int64_t(blockIdx.x) * kMThreads + threadIdx.x; i < count; i += int64_t(gridDim.x) * kMThreads) {
    mins[first + i] = Extrema<T>::highest();
    maxes[first + i] = Extrema<T>::lowest();
  }
}

// slot = better(slot, v) where better is min (IS_MIN) or max.  Reads first: once a group's extreme
// has settled most rows need no atomic at all.
template <typename T, bool IS_MIN>
__device__ __forceinline__ void AtomicExtreme(T* slot, T v) {
  auto better = [](T a, T b) { return IS_MIN ? a < b : a > b; };  // strictly: ties and NaN leave the slot alone
  if (!better(v, *slot)) return;
  if constexpr (sizeof(T) == 8 && !std::is_floating_point<T>::value) {
    using A = typename std::conditional<std::is_signed<T>::value, long long, unsigned long long>::type;
    if (IS_MIN) atomicMin(reinterpret_cast<A*>(slot), static_cast<A>(v));
    else atomicMax(reinterpret_cast<A*>(slot), static_cast<A>(v));
  } else if constexpr (sizeof(T) == 4 && !std::is_floating_point<T>::value) {
    using A = typename std::conditional<std::is_signed<T>::value, int, unsigned int>::type;
    if (IS_MIN) atomicMin(reinterpret_cast<A*>(slot), static_cast<A>(v));
    else atomicMax(reinterpret_cast<A*>(slot), static_cast<A>(v));
  } else if constexpr (sizeof(T) == 8) {  // double
    auto* w = reinterpret_cast<unsigned long long*>(slot);
    unsigned long long old = *w;
    while (better(v, __longlong_as_double(static_cast<long long>(old)))) {
      const unsigned long long seen = atomicCAS(w, old, static_cast<unsigned long long>(__double_as_longlong(v)));
      if (seen == old) break;
      old = seen;
    }
  } else {  // float and the 8/16-bit integers: compare-and-swap on the aligned 32-bit word holding the slot
    const uintptr_t addr = reinterpret_cast<uintptr_t>(slot);
    auto* w = reinterpret_cast<unsigned int*>(addr & ~uintptr_t(3));
    const int shift = static_cast<int>(addr & 3) * 8;
    const unsigned int field = sizeof(T) == 4 ? 0xFFFFFFFFu : ((1u << (8 * sizeof(T))) - 1u) << shift;
    unsigned int raw_v = 0;
    __builtin_memcpy(&raw_v, &v, sizeof(T));
    unsigned int old = *w;
    for (;;) {
      const unsigned int cur_bits = (old & field) >> shift;
      T cur;
      __builtin_memcpy(&cur, &cur_bits, sizeof(T));
      if (!better(v, cur)) break;
      const unsigned int seen = atomicCAS(w, old, (old & ~field) | (raw_v << shift));
      if (seen == old) break;
      old = seen;
    }
  }
}

__device__ __forceinline__ void SetBitAtomic(uint8_t* bm, uint32_t i) {
  auto* w = reinterpret_cast<unsigned int*>(bm) + (i >> 5);
  const unsigned int b = 1u << (i & 31);
  if (!(*w & b)) atomicOr(w, b);
}

// GetImpl::Visit consume_impl (:652-672): a valid value updates both extremes and has_values, a
// null marks has_nulls.
template <typename T>
__global__ __launch_bounds__(kMThreads) void k_minmax_consume(const T* values, const uint8_t* valid, int64_t off, int64_t n,
                                                              const uint32_t* ids, T* mins, T* maxes, uint8_t* has_values,
                                                              uint8_t* has_nulls) {
  for (int64_t i = int64_t(blockIdx.x) * kMThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kMThreads) {
    const uint32_t g = ids[i];
    if (valid && !GetBit(valid, off + i)) { SetBitAtomic(has_nulls, g); continue; }
    const T v = values[off + i];
    AtomicExtreme<T, false>(&maxes[g], v);
    AtomicExtreme<T, true>(&mins[g], v);
    SetBitAtomic(has_values, g);
  }
}

// Finalize (:729-748): valid = has_values, and under EMIT_NULL also not has_nulls.  One word per thread.
__global__ __launch_bounds__(kMThreads) void k_minmax_finalize(const uint64_t* has_values, const uint64_t* has_nulls, int emit_null,
                                                               int64_t num_groups, uint64_t* out_validity,
                                                               unsigned long long* counters) {
  const int64_t nwords = (num_groups + 63) >> 6;
  unsigned long long nulls = 0;
  for (int64_t w = int64_t(blockIdx.x) * kMThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kMThreads) {
    uint64_t v = has_values[w];
    if (emit_null) v &= ~has_nulls[w];
    const int64_t left = num_groups - (w << 6);
    const uint64_t live = left >= 64 ? ~uint64_t(0) : (uint64_t(1) << left) - 1;
    v &= live;
    out_validity[w] = v;
    nulls += __popcll(live & ~v);
  }
  nulls = WaveReduceSum64(nulls);
  if ((threadIdx.x & 63) == 0 && nulls) atomicAdd(CounterSlot(counters), nulls);
}

template <typename F>
static int DispatchNumeric(int type_id, F&& f) {
  switch (type_id) {
    case AMDARROW_UINT8: return f(uint8_t{});
    case AMDARROW_INT8: return f(int8_t{});
    case AMDARROW_UINT16: return f(uint16_t{});
    case AMDARROW_INT16: return f(int16_t{});
    case AMDARROW_UINT32: return f(uint32_t{});
    case AMDARROW_INT32: return f(int32_t{});
    case AMDARROW_UINT64: return f(uint64_t{});
    case AMDARROW_INT64: return f(int64_t{});
    case AMDARROW_FLOAT: return f(float{});
    case AMDARROW_DOUBLE: return f(double{});
    default: return SetError(AMDARROW_NOT_IMPLEMENTED, "Grouped MinMax data of this type");  // :674-684
  }
}

}  // namespace amdarrow

using namespace amdarrow;

extern "C" {

int amdarrow_hash_minmax_init(amdarrow_ctx* ctx, int value_type_id, void* mins, void* maxes, int64_t first, int64_t count) {
  if (count < 0 || first < 0) return SetError(AMDARROW_INVALID, "hash_min_max: negative group range");
  return DispatchNumeric(value_type_id, [&](auto tag) -> int {
    using T = decltype(tag);
    if (count == 0) return AMDARROW_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipLaunchKernelGGL(k_minmax_init<T>, dim3(GridFor(count, kMThreads, ctx->num_cus * 8)), dim3(kMThreads), 0, ctx->stream,
                       static_cast<T*>(mins), static_cast<T*>(maxes), first, count);
    AA_HIP(hipGetLastError());
    return AMDARROW_OK;
  });
}

int amdarrow_hash_minmax_accumulate(amdarrow_ctx* ctx, const amdarrow_array* values, const uint32_t* ids, void* mins,
                                    void* maxes, uint8_t* has_values, uint8_t* has_nulls) {
  return DispatchNumeric(values->type_id, [&](auto tag) -> int {
    using T = decltype(tag);
    if (values->length == 0) return AMDARROW_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    hipLaunchKernelGGL(k_minmax_consume<T>, dim3(GridFor(values->length, kMThreads, ctx->num_cus * 8)), dim3(kMThreads), 0,
                       ctx->stream, static_cast<const T*>(values->values), MayHaveNulls(values) ? values->validity : nullptr,
                       values->offset, values->length, ids, static_cast<T*>(mins), static_cast<T*>(maxes), has_values, has_nulls);
    AA_HIP(hipGetLastError());
    return AMDARROW_OK;
  });
}

int amdarrow_hash_minmax_finalize(amdarrow_ctx* ctx, int64_t num_groups, int null_handling, const uint8_t* has_values,
                                  const uint8_t* has_nulls, uint8_t* out_validity, amdarrow_out* out) {
  out->length = num_groups;
  out->null_count = 0;
  out->has_validity = 1;  // the struct's children always carry the bitmap (:741-742)
  if (num_groups == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  void* base;
  AA_TRY(Scratch(ctx, kCounterBytes, &base));
  auto* counters = static_cast<unsigned long long*>(base);
  AA_TRY(ClearCounters(ctx, counters));
  hipLaunchKernelGGL(k_minmax_finalize, dim3(GridFor((num_groups + 63) / 64, kMThreads, ctx->num_cus * 8)), dim3(kMThreads), 0,
                     ctx->stream, reinterpret_cast<const uint64_t*>(has_values), reinterpret_cast<const uint64_t*>(has_nulls),
                     null_handling == AMDARROW_MINMAX_EMIT_NULL, num_groups, reinterpret_cast<uint64_t*>(out_validity), counters);
  AA_HIP(hipGetLastError());
  unsigned long long nn = 0;
  AA_TRY(ReadCounters(ctx, counters, 1, &nn));
  out->null_count = static_cast<int64_t>(nn);
  return AMDARROW_OK;
}

}  // extern "C"

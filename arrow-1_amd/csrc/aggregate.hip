// aggregate.hip -- scalar aggregates over one array: count / sum / mean / min_max.
//
// Reference semantics: compute/kernels/aggregate_basic.cc:58-130 (CountImpl, SumInit, MinMaxInit),
// aggregate_basic_internal.h:54-330 (SumImpl / MeanImpl / MinMaxState / MinMaxImpl /
// BooleanMinMaxImpl), aggregate_internal.h:29-139 (FindAccumulatorType, SumArray).
// One pass produces everything a Consume of any of the four functions needs:
//   valid_count, sum (uint64 / int64 wrap-around, double for floats, true count for booleans),
//   min, max over the valid slots in the accumulator family (fmin / fmax for floats: NaN ignored).
// The caller (host layer) derives mean = double(sum) / valid_count, the null results and
// MinMaxOptions::EMIT_NULL from valid_count and length, and merges chunks / batches
// (MergeFrom: += for count and sum, min / max of the extrema).
// Float sums: the reference adds pairwise in blocks of 16; here every lane adds its strided rows
// sequentially and the partials are combined by a tree -- same error class, different rounding
// (tests compare with a tolerance relative to the sum of magnitudes).
// HBM stream: 1 value + 1/8 validity byte per row, one 32-byte partial per workgroup, a second
// one-workgroup kernel folds the partials (no same-address atomics, see common.h).
#include "common.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace amdarrow {

constexpr int kAggThreads = 1024;

struct AggPartial { long long valid; unsigned long long sum, min, max; };  // bits in the accumulator family

enum { kAggUnsigned = 0, kAggSigned = 1, kAggFloat = 2, kAggBool = 3 };

template <int KIND> struct AggOps;
template <> struct AggOps<kAggUnsigned> {
  using A = unsigned long long;
  __device__ static A Zero() { return 0; }
  __device__ static A Add(A a, A b) { return a + b; }
  __device__ static A Min(A a, A b) { return a < b ? a : b; }
  __device__ static A Max(A a, A b) { return a > b ? a : b; }
};
template <> struct AggOps<kAggSigned> {
  using A = long long;
  __device__ static A Zero() { return 0; }
  __device__ static A Add(A a, A b) { return static_cast<A>(static_cast<unsigned long long>(a) + static_cast<unsigned long long>(b)); }
  __device__ static A Min(A a, A b) { return a < b ? a : b; }
  __device__ static A Max(A a, A b) { return a > b ? a : b; }
};
template <> struct AggOps<kAggFloat> {
  using A = double;
  __device__ static A Zero() { return 0.0; }
  __device__ static A Add(A a, A b) { return a + b; }
  __device__ static A Min(A a, A b) { return fmin(a, b); }  // MinMaxState<floating>::MergeOne :197-200
  __device__ static A Max(A a, A b) { return fmax(a, b); }
};

template <typename A> __device__ __forceinline__ A ShflXor(A v, int d) {
  unsigned long long b;
  __builtin_memcpy(&b, &v, 8);
  b = __shfl_xor(b, d, 64);
  A r;
  __builtin_memcpy(&r, &b, 8);
  return r;
}
template <typename A> __device__ __forceinline__ unsigned long long Bits(A v) {
  unsigned long long b;
  __builtin_memcpy(&b, &v, 8);
  return b;
}
template <typename A> __device__ __forceinline__ A FromBits(unsigned long long b) {
  A v;
  __builtin_memcpy(&v, &b, 8);
  return v;
}

// Fold (valid, sum, min, max) over the workgroup; thread 0 returns the result.
template <int KIND>
__device__ __forceinline__ void BlockFold(long long& valid, typename AggOps<KIND>::A& sum, typename AggOps<KIND>::A& mn,
                                          typename AggOps<KIND>::A& mx) {
  using Ops = AggOps<KIND>;
  using A = typename Ops::A;
  __shared__ long long s_valid[kAggThreads / 64];
  __shared__ unsigned long long s_sum[kAggThreads / 64], s_min[kAggThreads / 64], s_max[kAggThreads / 64];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    valid += __shfl_xor(valid, d, 64);
    sum = Ops::Add(sum, ShflXor(sum, d));
    mn = Ops::Min(mn, ShflXor(mn, d));
    mx = Ops::Max(mx, ShflXor(mx, d));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_valid[wave] = valid; s_sum[wave] = Bits(sum); s_min[wave] = Bits(mn); s_max[wave] = Bits(mx); }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < static_cast<int>(blockDim.x >> 6); ++w) {
      valid += s_valid[w];
      sum = Ops::Add(sum, FromBits<A>(s_sum[w]));
      mn = Ops::Min(mn, FromBits<A>(s_min[w]));
      mx = Ops::Max(mx, FromBits<A>(s_max[w]));
    }
  }
}

template <typename T, int KIND>
__global__ __launch_bounds__(kAggThreads) void k_aggregate(const T* values, const uint8_t* valid_bits, int64_t off, int64_t n,
                                                           unsigned long long min_identity, unsigned long long max_identity,
                                                           AggPartial* partials) {
  using Ops = AggOps<KIND>;
  using A = typename Ops::A;
  long long valid = 0;
  A sum = Ops::Zero(), mn = FromBits<A>(min_identity), mx = FromBits<A>(max_identity);
  // A wave takes 8 x 64 consecutive rows per step: eight independent value loads (and their
  // validity words) are in flight together -- with one load per step the kernel ran at 1.9 TB/s.
  constexpr int kU = 8;
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t(blockIdx.x) * kAggThreads + threadIdx.x) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kAggThreads) >> 6;
  for (int64_t base = wave_id * (64 * kU); base < n; base += nwaves * (64 * kU)) {
    T v[kU];
    uint64_t vw[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {  // the value loads first: no control flow between them
      const int64_t r = base + 64 * u + lane;
      v[u] = values[off + (r < n ? r : n - 1)];  // clamped: never past the buffer, masked out by vw
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {  // (LoadBits64 branches on the word straddle, which serialises what follows it)
      const int64_t row0 = base + 64 * u;
      const int64_t left = n - row0;
      const int nbits = left >= 64 ? 64 : (left > 0 ? static_cast<int>(left) : 0);
      vw[u] = nbits > 0 ? LoadBitsOrOnes(valid_bits, off + row0, nbits) : 0;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if ((vw[u] >> lane) & 1) {
        const A x = static_cast<A>(v[u]);
        ++valid;
        sum = Ops::Add(sum, x);
        mn = Ops::Min(mn, x);
        mx = Ops::Max(mx, x);
      }
    }
  }
  BlockFold<KIND>(valid, sum, mn, mx);
  if (threadIdx.x == 0) partials[blockIdx.x] = AggPartial{valid, Bits(sum), Bits(mn), Bits(mx)};
}

// No validity buffer, values on a 16-byte boundary: 16-byte loads, four of them in flight per lane (the
// one-element-per-lane kernel above reads 800 MB of int64 in 235 us; the group-by runs this pass over a 64-bit
// key column before anything else, to range-compress it).
template <typename T, int KIND>
__global__ __launch_bounds__(kAggThreads) void k_aggregate_vec(const T* values, int64_t n, unsigned long long min_identity,
                                                               unsigned long long max_identity, AggPartial* partials) {
  using Ops = AggOps<KIND>;
  using A = typename Ops::A;
  constexpr int kPer = 16 / sizeof(T);  // elements per 16-byte load
  constexpr int kU = 4;
  using Vec = __attribute__((ext_vector_type(kPer))) T;
  long long valid = 0;
  A sum = Ops::Zero(), mn = FromBits<A>(min_identity), mx = FromBits<A>(max_identity);
  const int64_t nvec = n / kPer;
  const Vec* vp = reinterpret_cast<const Vec*>(values);
  const int64_t tid = int64_t(blockIdx.x) * kAggThreads + threadIdx.x, nthreads = int64_t(gridDim.x) * kAggThreads;
  for (int64_t i0 = tid; i0 < nvec; i0 += nthreads * kU) {
    Vec v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + u * nthreads;
      v[u] = __builtin_nontemporal_load(vp + (i < nvec ? i : nvec - 1));
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (i0 + u * nthreads >= nvec) continue;
#pragma unroll
      for (int e = 0; e < kPer; ++e) {
        const A x = static_cast<A>(v[u][e]);
        sum = Ops::Add(sum, x);
        mn = Ops::Min(mn, x);
        mx = Ops::Max(mx, x);
      }
      valid += kPer;
    }
  }
  if (tid == 0) {  // the elements behind the last whole vector
    for (int64_t r = nvec * kPer; r < n; ++r) {
      const A x = static_cast<A>(values[r]);
      ++valid;
      sum = Ops::Add(sum, x);
      mn = Ops::Min(mn, x);
      mx = Ops::Max(mx, x);
    }
  }
  BlockFold<KIND>(valid, sum, mn, mx);
  if (threadIdx.x == 0) partials[blockIdx.x] = AggPartial{valid, Bits(sum), Bits(mn), Bits(mx)};
}

// The same with a validity bitmap: the 16 / sizeof(T) rows of a lane's vector take their validity bits from one
// or two bytes of the bitmap (any bit offset), fetched next to the vector.
template <typename T, int KIND>
__global__ __launch_bounds__(kAggThreads) void k_aggregate_vec_valid(const T* values, const uint8_t* __restrict__ valid_bits,
                                                                     int64_t bit_off, int64_t n, unsigned long long min_identity,
                                                                     unsigned long long max_identity, AggPartial* partials) {
  using Ops = AggOps<KIND>;
  using A = typename Ops::A;
  constexpr int kPer = 16 / sizeof(T);  // elements per 16-byte load: 2 or 4
  constexpr int kU = 4;
  using Vec = __attribute__((ext_vector_type(kPer))) T;
  long long valid = 0;
  A sum = Ops::Zero(), mn = FromBits<A>(min_identity), mx = FromBits<A>(max_identity);
  const int64_t nvec = n / kPer;
  const int64_t last_byte = (bit_off + n - 1) >> 3;
  const Vec* vp = reinterpret_cast<const Vec*>(values);
  const int64_t tid = int64_t(blockIdx.x) * kAggThreads + threadIdx.x, nthreads = int64_t(gridDim.x) * kAggThreads;
  for (int64_t i0 = tid; i0 < nvec; i0 += nthreads * kU) {
    Vec v[kU];
    uint32_t b0[kU], b1[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {  // every load of the round first
      const int64_t i = i0 + u * nthreads < nvec ? i0 + u * nthreads : nvec - 1;
      v[u] = __builtin_nontemporal_load(vp + i);
      const int64_t byte = (bit_off + i * kPer) >> 3;
      b0[u] = valid_bits[byte];
      b1[u] = valid_bits[byte + 1 <= last_byte ? byte + 1 : last_byte];
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + u * nthreads;
      if (i >= nvec) continue;
      const uint32_t bits = ((b0[u] | (b1[u] << 8)) >> ((bit_off + i * kPer) & 7)) & ((1u << kPer) - 1u);
#pragma unroll
      for (int e = 0; e < kPer; ++e) {
        if ((bits >> e) & 1u) {
          const A x = static_cast<A>(v[u][e]);
          ++valid;
          sum = Ops::Add(sum, x);
          mn = Ops::Min(mn, x);
          mx = Ops::Max(mx, x);
        }
      }
    }
  }
  if (tid == 0) {  // the elements behind the last whole vector
    for (int64_t r = nvec * kPer; r < n; ++r) {
      if (!((valid_bits[(bit_off + r) >> 3] >> ((bit_off + r) & 7)) & 1)) continue;
      const A x = static_cast<A>(values[r]);
      ++valid;
      sum = Ops::Add(sum, x);
      mn = Ops::Min(mn, x);
      mx = Ops::Max(mx, x);
    }
  }
  BlockFold<KIND>(valid, sum, mn, mx);
  if (threadIdx.x == 0) partials[blockIdx.x] = AggPartial{valid, Bits(sum), Bits(mn), Bits(mx)};
}

// booleans: one 64-bit word per lane (BooleanArray::true_count, BooleanMinMaxImpl :317-343)
__global__ __launch_bounds__(kAggThreads) void k_aggregate_bool(const uint8_t* data, const uint8_t* valid_bits, int64_t off,
                                                                int64_t n, AggPartial* partials) {
  long long valid = 0;
  unsigned long long trues = 0, mn = 1, mx = 0;
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = int64_t(blockIdx.x) * kAggThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kAggThreads) {
    const int64_t row = w << 6;
    const int64_t left = n - row;
    const int nbits = left >= 64 ? 64 : static_cast<int>(left);
    const uint64_t v = LoadBitsOrOnes(valid_bits, off + row, nbits);
    const uint64_t d = LoadBits64(data, off + row, nbits) & v;
    valid += __popcll(v);
    trues += __popcll(d);
  }
  // min = no valid false, max = some valid true -- decided from the totals by the final kernel
  BlockFold<kAggUnsigned>(valid, trues, mn, mx);
  if (threadIdx.x == 0) partials[blockIdx.x] = AggPartial{valid, trues, 1, 0};
}

template <int KIND>
__global__ __launch_bounds__(256) void k_aggregate_final(const AggPartial* partials, int nblocks, unsigned long long min_identity,
                                                         unsigned long long max_identity, int is_bool, AggPartial* out) {
  using Ops = AggOps<KIND>;
  using A = typename Ops::A;
  long long valid = 0;
  A sum = Ops::Zero(), mn = FromBits<A>(min_identity), mx = FromBits<A>(max_identity);
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    const AggPartial p = partials[b];
    valid += p.valid;
    sum = Ops::Add(sum, FromBits<A>(p.sum));
    mn = Ops::Min(mn, FromBits<A>(p.min));
    mx = Ops::Max(mx, FromBits<A>(p.max));
  }
  BlockFold<KIND>(valid, sum, mn, mx);
  if (threadIdx.x == 0) {
    AggPartial r{valid, Bits(sum), Bits(mn), Bits(mx)};
    if (is_bool) {  // local.max = true_count > 0; local.min = false_count == 0 (:337-340)
      r.max = r.sum > 0 ? 1 : 0;
      r.min = static_cast<unsigned long long>(valid) == r.sum ? 1 : 0;
    }
    *out = r;
  }
}

// The body of amdarrow_aggregate; the caller holds ctx->mu (group-by uses it to range-compress int64 keys).
int AggregateLocked(amdarrow_ctx* ctx, const amdarrow_array* a, amdarrow_agg* out) {
  const int t = a->type_id;
  const int bw = BitWidth(t);
  if (bw < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "No sum implemented");  // SumLikeInit::Visit(const DataType&) :108
  const bool is_f = IsFloat(t), is_s = IsSignedInt(t);
  const int kind = t == AMDARROW_BOOL ? kAggBool : is_f ? kAggFloat : is_s ? kAggSigned : kAggUnsigned;
  // identities of MinMaxState (:140-219)
  unsigned long long min_id, max_id;
  if (kind == kAggFloat) {
    const double pinf = std::numeric_limits<double>::infinity(), ninf = -pinf;
    memcpy(&min_id, &pinf, 8); memcpy(&max_id, &ninf, 8);
  } else if (kind == kAggSigned) {
    min_id = static_cast<unsigned long long>(bw == 64 ? INT64_MAX : (int64_t(1) << (bw - 1)) - 1);
    max_id = static_cast<unsigned long long>(bw == 64 ? INT64_MIN : -(int64_t(1) << (bw - 1)));
  } else if (kind == kAggBool) {
    min_id = 1; max_id = 0;
  } else {
    min_id = bw == 64 ? ~0ull : (1ull << bw) - 1;
    max_id = 0;
  }
  out->valid_count = 0; out->sum = 0; out->min = min_id; out->max = max_id;
  if (a->length == 0) return AMDARROW_OK;
  const int64_t items = kind == kAggBool ? (a->length + 63) / 64 : a->length;
  const int nblocks = GridFor(items, kAggThreads, ctx->num_cus * 2);
  void* base;
  AA_TRY(Scratch(ctx, sizeof(AggPartial) * (nblocks + 1), &base));
  auto* partials = static_cast<AggPartial*>(base);
  AggPartial* result = partials + nblocks;
  const uint8_t* valid = MayHaveNulls(a) ? a->validity : nullptr;
  const bool vec_ok = a->length >= (int64_t(1) << 16) && bw >= 32;
#define AA_AGG(T, K)                                                                                                              \
  do {                                                                                                                             \
    const T* first = static_cast<const T*>(a->values) + a->offset;                                                                 \
    if (vec_ok && !valid && reinterpret_cast<uintptr_t>(first) % 16 == 0)                                                          \
      hipLaunchKernelGGL((k_aggregate_vec<T, K>), dim3(nblocks), dim3(kAggThreads), 0, ctx->stream, first, a->length, min_id,    \
                         max_id, partials);                                                                                        \
    else if (vec_ok && valid && reinterpret_cast<uintptr_t>(first) % 16 == 0)                                                      \
      hipLaunchKernelGGL((k_aggregate_vec_valid<T, K>), dim3(nblocks), dim3(kAggThreads), 0, ctx->stream, first, valid,           \
                         a->offset, a->length, min_id, max_id, partials);                                                          \
    else                                                                                                                           \
      hipLaunchKernelGGL((k_aggregate<T, K>), dim3(nblocks), dim3(kAggThreads), 0, ctx->stream, static_cast<const T*>(a->values), \
                         valid, a->offset, a->length, min_id, max_id, partials);                                                   \
  } while (0)
  switch (t) {
    case AMDARROW_BOOL:
      hipLaunchKernelGGL(k_aggregate_bool, dim3(nblocks), dim3(kAggThreads), 0, ctx->stream,
                         static_cast<const uint8_t*>(a->values), valid, a->offset, a->length, partials);
      break;
    case AMDARROW_UINT8: AA_AGG(uint8_t, kAggUnsigned); break;
    case AMDARROW_INT8: AA_AGG(int8_t, kAggSigned); break;
    case AMDARROW_UINT16: AA_AGG(uint16_t, kAggUnsigned); break;
    case AMDARROW_INT16: AA_AGG(int16_t, kAggSigned); break;
    case AMDARROW_UINT32: AA_AGG(uint32_t, kAggUnsigned); break;
    case AMDARROW_INT32: AA_AGG(int32_t, kAggSigned); break;
    case AMDARROW_UINT64: AA_AGG(uint64_t, kAggUnsigned); break;
    case AMDARROW_INT64: AA_AGG(int64_t, kAggSigned); break;
    case AMDARROW_FLOAT: AA_AGG(float, kAggFloat); break;
    default: AA_AGG(double, kAggFloat); break;
  }
#undef AA_AGG
  AA_HIP(hipGetLastError());
  if (kind == kAggFloat)
    hipLaunchKernelGGL(k_aggregate_final<kAggFloat>, dim3(1), dim3(256), 0, ctx->stream, partials, nblocks, min_id, max_id, 0, result);
  else if (kind == kAggSigned)
    hipLaunchKernelGGL(k_aggregate_final<kAggSigned>, dim3(1), dim3(256), 0, ctx->stream, partials, nblocks, min_id, max_id, 0, result);
  else
    hipLaunchKernelGGL(k_aggregate_final<kAggUnsigned>, dim3(1), dim3(256), 0, ctx->stream, partials, nblocks, min_id, max_id,
                       kind == kAggBool ? 1 : 0, result);
  AA_HIP(hipGetLastError());
  AggPartial host;
  AA_TRY(ReadBack(ctx, result, sizeof(host), &host));
  out->valid_count = host.valid;
  out->sum = host.sum;
  out->min = host.min;
  out->max = host.max;
  return AMDARROW_OK;
}

}  // namespace amdarrow

extern "C" int amdarrow_aggregate(amdarrow_ctx* ctx, const amdarrow_array* a, amdarrow_agg* out) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  return amdarrow::AggregateLocked(ctx, a, out);
}

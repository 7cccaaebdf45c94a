// scalar.hip -- elementwise compare / arithmetic kernels and validity propagation.
//
// Reference semantics:
//   compute/kernels/scalar_compare.cc:31-57,100-171   (six comparisons; less = flipped greater)
//   compute/kernels/scalar_arithmetic.cc:69-251        (add/sub/mul wrap through unsigned;
//                                                       divide + *_checked error codes)
//   compute/kernels/codegen_internal.h:758-824         (ScalarBinary: op on EVERY slot)
//   compute/kernels/codegen_internal.h:828-932         (ScalarBinaryNotNull: nulls -> 0)
//   compute/exec.cc:255-461                            (NullGeneralization / NullPropagator)
//   util/bitmap_generate.h:62-108                      (boolean output LSB-first, tail zero)
//
// All kernels are HBM streams: one element per lane, 4 independent 64-row groups per
// workgroup iteration; boolean results are packed with a wave ballot (one 8-byte store per
// 64 rows); output validity is produced a 64-bit word per lane from funnel-shifted input words.
#include "common.h"

#include <cstring>
#include <limits>
#include <string>
#include <type_traits>

namespace amdarrow {

constexpr int kThreads = 256;

// ---- validity: out = a_valid & b_valid at arbitrary bit offsets (BitmapAnd / CopyBitmap) -------
__global__ __launch_bounds__(kThreads) void k_validity(const uint8_t* a, int64_t a_off, const uint8_t* b, int64_t b_off,
                                                       int all_null, int64_t length, uint64_t* out) {
  const int64_t nwords = (length + 63) >> 6;
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) {
    const int64_t row = w << 6;
    const int64_t left = length - row;
    const int nbits = left >= 64 ? 64 : static_cast<int>(left);
    uint64_t x = 0;
    if (!all_null) x = LoadBitsOrOnes(a, a_off + row, nbits) & LoadBitsOrOnes(b, b_off + row, nbits);
    out[w] = x;  // SetBitsTo(.., length, true) leaves the tail of the last word zero
  }
}

struct Operand {
  const void* data;   // nullptr => scalar
  int64_t off;
  uint64_t scalar;    // raw bits of the scalar value
};

template <typename T> __device__ __forceinline__ T LoadOp(const Operand& o, int64_t i) {
  if (o.data) return reinterpret_cast<const T*>(o.data)[o.off + i];
  T v;
  __builtin_memcpy(&v, &o.scalar, sizeof(T));
  return v;
}
template <> __device__ __forceinline__ bool LoadOp<bool>(const Operand& o, int64_t i) {
  if (o.data) return GetBit(reinterpret_cast<const uint8_t*>(o.data), o.off + i);
  return (o.scalar & 0xFF) != 0;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_compare(int op, Operand l, Operand r, int64_t n, uint64_t* out_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t nround = (n + 63) & ~int64_t(63);
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < nround; i += int64_t(gridDim.x) * kThreads) {
    bool v = false;
    if (i < n) {
      const T a = LoadOp<T>(l, i), b = LoadOp<T>(r, i);
      switch (op) {
        case AMDARROW_EQ: v = a == b; break;
        case AMDARROW_NE: v = a != b; break;
        case AMDARROW_GT: v = a > b; break;
        case AMDARROW_GE: v = a >= b; break;
        case AMDARROW_LT: v = b > a; break;   // scalar_compare.cc:161-171
        default: v = b >= a; break;
      }
    }
    const uint64_t m = __ballot(v);
    if (lane == 0) out_bits[i >> 6] = m;
  }
}

// error word: (row << 2 | kind), max over rows => the LAST failing row decides the message,
// as *st is overwritten per failing element in the reference loops.
enum { kErrOverflow = 1, kErrDivZero = 2, kErrNegativePower = 3 };

template <typename T, typename U, bool kSigned>
__device__ __forceinline__ T IntOp(int op, T l, T r, bool valid, int* err) {
  switch (op) {
    case AMDARROW_ADD: return static_cast<T>(static_cast<U>(static_cast<U>(l) + static_cast<U>(r)));
    case AMDARROW_SUB: return static_cast<T>(static_cast<U>(static_cast<U>(l) - static_cast<U>(r)));
    case AMDARROW_MUL: return static_cast<T>(static_cast<uint64_t>(static_cast<U>(l)) * static_cast<uint64_t>(static_cast<U>(r)));
    case AMDARROW_NEGATE: return static_cast<T>(static_cast<U>(~static_cast<U>(l) + 1));  // :253-268 (SafeSignedNegate wraps)
    case AMDARROW_POWER: {  // Power::IntegerPower :296-318: right to left in uint64, every slot
      if (kSigned && r < 0) { *err = kErrNegativePower; return 0; }
      uint64_t base = static_cast<uint64_t>(l), exp = static_cast<uint64_t>(r), pow = 1;
      while (exp) {
        pow *= (exp & 1) ? base : 1;
        base *= base;
        exp >>= 1;
      }
      return static_cast<T>(pow);
    }
    default: break;
  }
  if (!valid) return 0;  // ScalarBinaryNotNull::WriteNull
  T res = 0;
  switch (op) {
    case AMDARROW_NEGATE_CHECKED:  // NegateWithOverflow :270-279 (signed only)
      if (__builtin_sub_overflow(static_cast<T>(0), l, &res)) *err = kErrOverflow;
      break;
    case AMDARROW_POWER_CHECKED: {  // :322-351: left to right with overflow checks
      if (kSigned && r < 0) { *err = kErrNegativePower; return 0; }
      if (r == 0) return 1;
      bool overflow = false;
      uint64_t bitmask = 1ull << (63 - __clzll(static_cast<long long>(static_cast<uint64_t>(r))));
      T pow = 1;
      while (bitmask) {
        overflow |= __builtin_mul_overflow(pow, pow, &pow);
        if (static_cast<uint64_t>(r) & bitmask) overflow |= __builtin_mul_overflow(pow, l, &pow);
        bitmask >>= 1;
      }
      if (overflow) *err = kErrOverflow;
      res = pow;
      break;
    }
    case AMDARROW_ADD_CHECKED: if (__builtin_add_overflow(l, r, &res)) *err = kErrOverflow; break;
    case AMDARROW_SUB_CHECKED: if (__builtin_sub_overflow(l, r, &res)) *err = kErrOverflow; break;
    case AMDARROW_MUL_CHECKED: if (__builtin_mul_overflow(l, r, &res)) *err = kErrOverflow; break;
    default:  // DIV, DIV_CHECKED (scalar_arithmetic.cc:205-251)
      if (r == 0) { *err = kErrDivZero; res = 0; }
      else if (kSigned && l == std::numeric_limits<T>::min() && r == static_cast<T>(-1)) {
        if (op == AMDARROW_DIV_CHECKED) *err = kErrOverflow;
        res = 0;
      } else res = static_cast<T>(l / r);
      break;
  }
  return res;
}
// std::pow (:314-318).  float32 goes through the double-precision pow and one rounding: within 1 ulp of the host
// libm's powf; float64 is ocml's pow (<= 2 ulp from the correctly rounded value glibc returns).
template <typename T> __device__ __forceinline__ T PowAs(T l, T r) { return static_cast<T>(pow(static_cast<double>(l), static_cast<double>(r))); }
template <typename T>
__device__ __forceinline__ T FloatOp(int op, T l, T r, bool valid, int* err) {
  switch (op) {
    case AMDARROW_ADD: return l + r;
    case AMDARROW_SUB: return l - r;
    case AMDARROW_MUL: return l * r;
    case AMDARROW_NEGATE: return -l;
    case AMDARROW_POWER: return PowAs<T>(l, r);
    default: break;
  }
  if (!valid) return 0;
  switch (op) {
    case AMDARROW_NEGATE_CHECKED: return -l;
    case AMDARROW_POWER_CHECKED: return PowAs<T>(l, r);
    case AMDARROW_ADD_CHECKED: return l + r;
    case AMDARROW_SUB_CHECKED: return l - r;
    case AMDARROW_MUL_CHECKED: return l * r;
    case AMDARROW_DIV: return l / r;
    default:
      if (r == 0) { *err = kErrDivZero; return 0; }
      return l / r;
  }
}

template <typename T, typename U, int kKind /*0 unsigned, 1 signed, 2 float*/>
__global__ __launch_bounds__(kThreads) void k_arith(int op, Operand l, Operand r, const uint8_t* out_valid, int64_t n,
                                                    T* out, unsigned long long* err_word) {
  unsigned long long my_err = 0;
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads) {
    const T a = LoadOp<T>(l, i), b = LoadOp<T>(r, i);
    const bool valid = (op < AMDARROW_DIV || op == AMDARROW_POWER || op == AMDARROW_NEGATE) ? true : GetBit(out_valid, i);
    int err = 0;
    T res;
    if constexpr (kKind == 2) res = FloatOp<T>(op, a, b, valid, &err);
    else res = IntOp<T, U, kKind == 1>(op, a, b, valid, &err);
    out[i] = res;
    if (err) my_err = (static_cast<unsigned long long>(i) << 2) | err;
  }
  if (my_err) atomicMax(err_word, my_err);
}

// NullGeneralization::Get, compute/exec.cc:258-281
static int Generalization(const amdarrow_array* a) {
  if (!a) return 1;  // valid scalar
  if (a->null_count == 0 || a->validity == nullptr) return 1;
  if (a->null_count == a->length) return 2;
  return 0;
}

// NullPropagator with a preallocated bitmap (ScalarExecutor always preallocates validity
// for INTERSECTION kernels, compute/exec.cc:690-693).
static int Propagate(amdarrow_ctx* ctx, const amdarrow_array* lhs, const amdarrow_array* rhs, int64_t n,
                     uint8_t* out_validity, amdarrow_out* out) {
  const amdarrow_array* with_nulls[2];
  int nw = 0;
  bool all_null = false;
  const amdarrow_array* in[2] = {lhs, rhs};
  for (int k = 0; k < 2; ++k) {
    const int g = Generalization(in[k]);
    if (g == 2) all_null = true;
    if (g != 1 && in[k]) with_nulls[nw++] = in[k];
  }
  out->length = n;
  out->has_validity = 1;
  if (all_null) out->null_count = n;                       // AllNullShortCircuit :326-348
  else if (nw == 0) out->null_count = 0;                   // :433-439
  else if (nw == 1) out->null_count = with_nulls[0]->null_count;  // PropagateSingle :350-380
  else out->null_count = -1;                               // PropagateMultiple :382-408
  if (n == 0) {  // a zero-length call runs no batch: the empty result keeps kUnknownNullCount
    out->null_count = -1;
    return AMDARROW_OK;
  }
  const uint8_t* a = nw > 0 ? with_nulls[0]->validity : nullptr;
  const int64_t a_off = nw > 0 ? with_nulls[0]->offset : 0;
  const uint8_t* b = nw > 1 ? with_nulls[1]->validity : nullptr;
  const int64_t b_off = nw > 1 ? with_nulls[1]->offset : 0;
  const int grid = GridFor((n + 63) / 64, kThreads, ctx->num_cus * 8);
  hipLaunchKernelGGL(k_validity, dim3(grid), dim3(kThreads), 0, ctx->stream, a, a_off, b, b_off, all_null ? 1 : 0, n,
                     reinterpret_cast<uint64_t*>(out_validity));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

static int CheckBinary(const amdarrow_array* lhs, const void* ls, const amdarrow_array* rhs, const void* rs, int* type,
                       int64_t* n) {
  if ((!lhs && !ls) || (!rhs && !rs) || (!lhs && !rhs))
    return SetError(AMDARROW_INVALID, "binary kernel: each operand is an array or a scalar, at least one an array");
  if (lhs && rhs) {
    if (lhs->length != rhs->length)
      return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");  // compute/exec.cc:190-192
    if (lhs->type_id != rhs->type_id)
      return SetError(AMDARROW_NOT_IMPLEMENTED, "binary kernel: operands must already share a type "
                                                "(implicit casts are done by DispatchBest on the host)");
  }
  *type = lhs ? lhs->type_id : rhs->type_id;
  *n = lhs ? lhs->length : rhs->length;
  if (BitWidth(*type) < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "binary kernel: unsupported type");
  return AMDARROW_OK;
}

static Operand MakeOperand(const amdarrow_array* a, const void* scalar, int type) {
  Operand o{};
  if (a) { o.data = a->values; o.off = a->offset; }
  else memcpy(&o.scalar, scalar, type == AMDARROW_BOOL ? 1 : BitWidth(type) / 8);
  return o;
}

// ---- mask-building scalars ----------------------------------------------------------------
// Boolean kernels (compute/kernels/scalar_boolean.cc), is_valid / is_null / is_nan
// (scalar_validity.cc), fill_null (scalar_fill_null.cc).  Bitmap work: one 64-bit output word per
// lane, inputs funnel-shifted from arbitrary bit offsets; 2 x N/8 bytes in, N/8 out per operand.
struct BoolSrc { const uint8_t* data; const uint8_t* valid; int64_t off; int scalar; };  // scalar < 0: array

__device__ __forceinline__ void FetchBool(const BoolSrc& s, int64_t row, int nbits, uint64_t live, uint64_t* d, uint64_t* v) {
  if (s.scalar < 0) {
    *d = LoadBits64(s.data, s.off + row, nbits);
    *v = s.valid ? LoadBits64(s.valid, s.off + row, nbits) : live;
  } else {
    *d = s.scalar == AMDARROW_SCALAR_TRUE ? live : 0;
    *v = s.scalar == AMDARROW_SCALAR_NULL ? 0 : live;
  }
}

enum { kDataPlain = 0, kDataKleene, kDataZero, kDataOnes, kDataCopy, kDataCopyInverted, kDataFill };
enum { kValidNone = 0, kValidKleene, kValidCopy, kValidDecides };

// `arr` = the array operand the copy / decides modes refer to (0 = left, 1 = right).
// null_count (optional, slotted counters of common.h): [0] null slots of the result, [1] / [2] null slots of the left / right array
// operand -- one pass also answers ArrayData::GetNullCount() for inputs whose count is unknown.
__global__ __launch_bounds__(kThreads) void k_boolean(int op, int data_mode, int valid_mode, int arr, BoolSrc l, BoolSrc r,
                                                      int64_t n, uint64_t* out_bits, uint64_t* out_valid,
                                                      unsigned long long* null_count) {
  const int64_t nwords = (n + 63) >> 6;
  unsigned long long nulls = 0, lnulls = 0, rnulls = 0;
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) {
    const int64_t row = w << 6;
    const int64_t left = n - row;
    const int nbits = left >= 64 ? 64 : static_cast<int>(left);
    const uint64_t live = nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0);
    uint64_t ld, lv, rd, rv;
    FetchBool(l, row, nbits, live, &ld, &lv);
    FetchBool(r, row, nbits, live, &rd, &rv);
    if (l.scalar < 0) lnulls += __popcll(~lv & live);
    if (r.scalar < 0) rnulls += __popcll(~rv & live);
    const bool or_type = op == AMDARROW_OR || op == AMDARROW_OR_KLEENE;
    const bool not_rhs = op == AMDARROW_AND_NOT || op == AMDARROW_AND_NOT_KLEENE;
    uint64_t data = 0, valid = live;
    switch (data_mode) {
      case kDataPlain:
        if (op == AMDARROW_INVERT) data = ~ld;                       // CopyFromInverted :103-106
        else if (op == AMDARROW_XOR) data = ld ^ rd;                 // :349-356
        else if (or_type) data = ld | rd;                            // :224-231
        else data = ld & (not_rhs ? ~rd : rd);                       // :132-139, :380-387
        break;
      case kDataKleene: {                                            // ComputeKleene :34-91
        const uint64_t rdd = not_rhs ? ~rd : rd;
        const uint64_t lt = lv & ld, lf = lv & ~ld, rt = rv & rdd, rf = rv & ~rdd;
        if (or_type) { data = lt | rt; valid = lt | rt | (lf & rf); }      // :311-319
        else { data = lt & rt; valid = lf | rf | (lt & rt); }              // :211-217, :447-453
        break;
      }
      case kDataZero: data = 0; break;
      case kDataOnes: data = live; break;
      case kDataCopy: data = arr ? rd : ld; break;
      case kDataFill: data = (lv & ld) | (rd & ~lv); break;  // fill_null on bits: rd = the fill bit replicated
      default: data = ~(arr ? rd : ld); break;
    }
    data &= live;
    if (valid_mode == kValidCopy) valid = arr ? rv : lv;
    else if (valid_mode == kValidDecides) valid = (arr ? rv : lv) & (or_type ? data : ~data);  // :178-190, :282-293, :417-430
    valid &= live;
    if (out_bits) out_bits[w] = data;
    if (out_valid) out_valid[w] = valid;
    nulls += __popcll(~valid & live);
  }
  if (null_count) {
    nulls = WaveReduceSum64(nulls);
    lnulls = WaveReduceSum64(lnulls);
    rnulls = WaveReduceSum64(rnulls);
    if ((threadIdx.x & 63) == 0) {
      unsigned long long* slot = CounterSlot(null_count);
      if (nulls) atomicAdd(&slot[0], nulls);
      if (lnulls) atomicAdd(&slot[1], lnulls);
      if (rnulls) atomicAdd(&slot[2], rnulls);
    }
  }
}

// is_valid (func 0) / is_null (func 1): the validity bitmap or its complement; constant when `valid` is null.
__global__ __launch_bounds__(kThreads) void k_validity_test(int func, const uint8_t* valid, int64_t off, int64_t n,
                                                            uint64_t* out_bits) {
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) {
    const int64_t row = w << 6;
    const int64_t left = n - row;
    const int nbits = left >= 64 ? 64 : static_cast<int>(left);
    const uint64_t live = nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0);
    const uint64_t v = LoadBitsOrOnes(valid, off + row, nbits);
    out_bits[w] = (func == 0 ? v : ~v) & live;
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_is_nan(const T* values, int64_t off, int64_t n, uint64_t* out_bits) {
  const int lane = threadIdx.x & 63;
  const int64_t nround = (n + 63) & ~int64_t(63);
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < nround; i += int64_t(gridDim.x) * kThreads) {
    bool v = false;
    if (i < n) { const T x = values[off + i]; v = x != x; }
    const uint64_t m = __ballot(v);
    if (lane == 0) out_bits[i >> 6] = m;
  }
}

// fill_null on fixed-width values of W bytes: out[i] = valid ? in[i] : fill (valid == nullptr: plain copy)
template <typename U>
__global__ __launch_bounds__(kThreads) void k_fill_null(const U* values, const uint8_t* valid, int64_t off, int64_t n, U fill,
                                                        U* out) {
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads) {
    const bool ok = valid ? GetBit(valid, off + i) : true;
    out[i] = ok ? values[off + i] : fill;
  }
}

// ---- cast between numbers (scalar_cast_numeric.cc:38-58, 203-262) ---------------------------------------
// One pass: convert and, for CastOptions::Safe, test the SOURCE value against [lo, hi] (the reference
// checks the whole input first -- IntegersCanFit / CheckForIntegerToFloatingTruncation -- and then
// converts; a failing call returns no output either way).  first_bad = lowest offending valid slot.
template <typename In, typename Out>
__global__ __launch_bounds__(kThreads) void k_cast(const In* values, const uint8_t* valid, int64_t off, int64_t n, Out* out,
                                                   int check, In lo, In hi, unsigned long long* first_bad) {
  unsigned long long bad = ~0ull;
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads) {
    const In v = values[off + i];
    out[i] = static_cast<Out>(v);
    if (check && (v < lo || v > hi) && (valid == nullptr || GetBit(valid, off + i)) && bad == ~0ull) bad = i;
  }
  if (bad != ~0ull) atomicMin(first_bad, bad);
}

}  // namespace amdarrow

using namespace amdarrow;

// cast (numbers).  Bounds of the safe mode, in the source type: integer -> integer GetSafeMinMax
// (util/int_util.cc:826-912: the intersection of the two ranges), integer -> floating the whole
// numbers the target represents exactly (FloatingIntegerBound, scalar_cast_numeric.cc:188-209).
namespace {
template <typename In> struct CastBounds { In lo, hi; bool check; };
template <typename In>
CastBounds<In> SafeBounds(int to) {
  using L = std::numeric_limits<In>;
  CastBounds<In> b{L::lowest(), L::max(), false};
  if constexpr (std::is_integral<In>::value) {
    __int128 lo = L::lowest(), hi = L::max(), tlo = 0, thi = 0;
    switch (to) {
      case AMDARROW_UINT8: thi = UINT8_MAX; break;
      case AMDARROW_INT8: tlo = INT8_MIN; thi = INT8_MAX; break;
      case AMDARROW_UINT16: thi = UINT16_MAX; break;
      case AMDARROW_INT16: tlo = INT16_MIN; thi = INT16_MAX; break;
      case AMDARROW_UINT32: thi = UINT32_MAX; break;
      case AMDARROW_INT32: tlo = INT32_MIN; thi = INT32_MAX; break;
      case AMDARROW_UINT64: thi = UINT64_MAX; break;
      case AMDARROW_INT64: tlo = INT64_MIN; thi = INT64_MAX; break;
      case AMDARROW_FLOAT: thi = 1ll << 24; tlo = -thi; if (sizeof(In) < 4) { tlo = lo; thi = hi; } break;
      default: thi = 1ll << 53; tlo = -thi; if (sizeof(In) < 8) { tlo = lo; thi = hi; } break;
    }
    if (tlo > lo) lo = tlo;
    if (thi < hi) hi = thi;
    b.check = lo != L::lowest() || hi != L::max();
    b.lo = static_cast<In>(lo); b.hi = static_cast<In>(hi);
  }
  return b;
}
template <typename In, typename Out>
int CastTo(amdarrow_ctx* ctx, const amdarrow_array* a, int to, int safe, void* out_values) {
  const CastBounds<In> b = SafeBounds<In>(to);
  const int check = safe && b.check;
  void* base;
  AA_TRY(Scratch(ctx, 256, &base));
  auto* first_bad = static_cast<unsigned long long*>(base);
  if (check) AA_HIP(hipMemsetAsync(first_bad, 0xff, 8, ctx->stream));
  const int64_t n = a->length;
  hipLaunchKernelGGL((k_cast<In, Out>), dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     static_cast<const In*>(a->values), MayHaveNulls(a) ? a->validity : nullptr, a->offset, n,
                     static_cast<Out*>(out_values), check, b.lo, b.hi, first_bad);
  AA_HIP(hipGetLastError());
  if (check) {
    unsigned long long bad = 0;
    AA_TRY(ReadBack(ctx, first_bad, 8, &bad));
    if (bad != ~0ull) {
      In v;
      AA_TRY(ReadBack(ctx, static_cast<const In*>(a->values) + a->offset + bad, sizeof(In), &v));
      return SetError(AMDARROW_INVALID, "Integer value " + std::to_string(v) + " not in range: " + std::to_string(b.lo) +
                                            " to " + std::to_string(b.hi));  // int_util.cc:614-618
    }
  }
  return AMDARROW_OK;
}
template <typename In>
int CastFrom(amdarrow_ctx* ctx, const amdarrow_array* a, int to, int safe, void* out_values) {
  if (std::is_floating_point<In>::value && to != AMDARROW_FLOAT && to != AMDARROW_DOUBLE)
    return SetError(AMDARROW_NOT_IMPLEMENTED, "cast: floating point to integer");
  switch (to) {
    case AMDARROW_UINT8: return CastTo<In, uint8_t>(ctx, a, to, safe, out_values);
    case AMDARROW_INT8: return CastTo<In, int8_t>(ctx, a, to, safe, out_values);
    case AMDARROW_UINT16: return CastTo<In, uint16_t>(ctx, a, to, safe, out_values);
    case AMDARROW_INT16: return CastTo<In, int16_t>(ctx, a, to, safe, out_values);
    case AMDARROW_UINT32: return CastTo<In, uint32_t>(ctx, a, to, safe, out_values);
    case AMDARROW_INT32: return CastTo<In, int32_t>(ctx, a, to, safe, out_values);
    case AMDARROW_UINT64: return CastTo<In, uint64_t>(ctx, a, to, safe, out_values);
    case AMDARROW_INT64: return CastTo<In, int64_t>(ctx, a, to, safe, out_values);
    case AMDARROW_FLOAT: return CastTo<In, float>(ctx, a, to, safe, out_values);
    case AMDARROW_DOUBLE: return CastTo<In, double>(ctx, a, to, safe, out_values);
    default: return SetError(AMDARROW_NOT_IMPLEMENTED, "cast: target type");
  }
}
}  // namespace


extern "C" {

int amdarrow_compare(amdarrow_ctx* ctx, int op, const amdarrow_array* lhs, const void* lhs_scalar,
                     const amdarrow_array* rhs, const void* rhs_scalar, uint8_t* out_bits, uint8_t* out_validity,
                     amdarrow_out* out) {
  int type; int64_t n;
  AA_TRY(CheckBinary(lhs, lhs_scalar, rhs, rhs_scalar, &type, &n));
  if (op < AMDARROW_EQ || op > AMDARROW_LE) return SetError(AMDARROW_INVALID, "compare: bad operator");
  std::lock_guard<std::mutex> lock(ctx->mu);
  AA_TRY(Propagate(ctx, lhs, rhs, n, out_validity, out));
  if (n == 0) return AMDARROW_OK;
  const Operand l = MakeOperand(lhs, lhs_scalar, type), r = MakeOperand(rhs, rhs_scalar, type);
  const int grid = GridFor(n, kThreads, ctx->num_cus * 8);
  uint64_t* ob = reinterpret_cast<uint64_t*>(out_bits);
#define AA_CMP(T) hipLaunchKernelGGL((k_compare<T>), dim3(grid), dim3(kThreads), 0, ctx->stream, op, l, r, n, ob)
  switch (type) {
    case AMDARROW_BOOL: AA_CMP(bool); break;
    case AMDARROW_UINT8: AA_CMP(uint8_t); break;
    case AMDARROW_INT8: AA_CMP(int8_t); break;
    case AMDARROW_UINT16: AA_CMP(uint16_t); break;
    case AMDARROW_INT16: AA_CMP(int16_t); break;
    case AMDARROW_UINT32: AA_CMP(uint32_t); break;
    case AMDARROW_INT32: AA_CMP(int32_t); break;
    case AMDARROW_UINT64: AA_CMP(uint64_t); break;
    case AMDARROW_INT64: AA_CMP(int64_t); break;
    case AMDARROW_FLOAT: AA_CMP(float); break;
    default: AA_CMP(double); break;
  }
#undef AA_CMP
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_arith(amdarrow_ctx* ctx, int op, const amdarrow_array* lhs, const void* lhs_scalar,
                   const amdarrow_array* rhs, const void* rhs_scalar, void* out_values, uint8_t* out_validity,
                   amdarrow_out* out) {
  int type; int64_t n;
  if (op < AMDARROW_ADD || op > AMDARROW_NEGATE_CHECKED) return SetError(AMDARROW_INVALID, "arith: bad operator");
  static const uint64_t zero = 0;
  if (op >= AMDARROW_NEGATE) { rhs = nullptr; rhs_scalar = &zero; }  // unary: the right operand is not read
  AA_TRY(CheckBinary(lhs, lhs_scalar, rhs, rhs_scalar, &type, &n));
  if (type == AMDARROW_BOOL) return SetError(AMDARROW_NOT_IMPLEMENTED, "arithmetic on boolean");
  if (op == AMDARROW_NEGATE_CHECKED && (type == AMDARROW_UINT8 || type == AMDARROW_UINT16 || type == AMDARROW_UINT32 || type == AMDARROW_UINT64))
    return SetError(AMDARROW_NOT_IMPLEMENTED, "Function negate_checked has no kernel matching input types (array[uint" +
                                                  std::to_string(BitWidth(type)) + "])");  // :454-466: none registered
  std::lock_guard<std::mutex> lock(ctx->mu);
  AA_TRY(Propagate(ctx, lhs, rhs, n, out_validity, out));
  if (n == 0) return AMDARROW_OK;
  void* base;
  AA_TRY(Scratch(ctx, 256, &base));
  auto* err_word = static_cast<unsigned long long*>(base);
  const bool can_fail = op >= AMDARROW_DIV;
  if (can_fail) AA_HIP(hipMemsetAsync(err_word, 0, 8, ctx->stream));
  const Operand l = MakeOperand(lhs, lhs_scalar, type), r = MakeOperand(rhs, rhs_scalar, type);
  const int grid = GridFor(n, kThreads, ctx->num_cus * 8);
#define AA_AR(T, U, K)                                                                                      \
  hipLaunchKernelGGL((k_arith<T, U, K>), dim3(grid), dim3(kThreads), 0, ctx->stream, op, l, r, out_validity, n, \
                     static_cast<T*>(out_values), err_word)
  switch (type) {
    case AMDARROW_UINT8: AA_AR(uint8_t, uint8_t, 0); break;
    case AMDARROW_INT8: AA_AR(int8_t, uint8_t, 1); break;
    case AMDARROW_UINT16: AA_AR(uint16_t, uint16_t, 0); break;
    case AMDARROW_INT16: AA_AR(int16_t, uint16_t, 1); break;
    case AMDARROW_UINT32: AA_AR(uint32_t, uint32_t, 0); break;
    case AMDARROW_INT32: AA_AR(int32_t, uint32_t, 1); break;
    case AMDARROW_UINT64: AA_AR(uint64_t, uint64_t, 0); break;
    case AMDARROW_INT64: AA_AR(int64_t, uint64_t, 1); break;
    case AMDARROW_FLOAT: AA_AR(float, uint32_t, 2); break;
    default: AA_AR(double, uint64_t, 2); break;
  }
#undef AA_AR
  AA_HIP(hipGetLastError());
  if (can_fail) {
    unsigned long long e = 0;
    AA_TRY(ReadBack(ctx, err_word, 8, &e));
    if (e) return SetError(AMDARROW_INVALID, (e & 3) == kErrDivZero ? "divide by zero" :
                           (e & 3) == kErrNegativePower ? "integers to negative integer powers are not allowed" : "overflow");
  }
  return AMDARROW_OK;
}

// ---- SURVEY 8f.1 entry points ------------------------------------------------------------------------
static int AllValid(amdarrow_ctx* ctx, int64_t n, uint8_t* out_validity) {  // ones for n bits, tail zero
  if (n == 0 || !out_validity) return AMDARROW_OK;
  hipLaunchKernelGGL(k_validity, dim3(GridFor((n + 63) / 64, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     static_cast<const uint8_t*>(nullptr), int64_t(0), static_cast<const uint8_t*>(nullptr), int64_t(0), 0, n,
                     reinterpret_cast<uint64_t*>(out_validity));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_boolean(amdarrow_ctx* ctx, int op, const amdarrow_array* lhs, int lhs_scalar, const amdarrow_array* rhs,
                     int rhs_scalar, uint8_t* out_bits, uint8_t* out_validity, amdarrow_out* out) {
  if (op < AMDARROW_INVERT || op > AMDARROW_OR_KLEENE) return SetError(AMDARROW_KEY_ERROR, "boolean: no such function");
  if (op == AMDARROW_INVERT) { rhs = nullptr; rhs_scalar = AMDARROW_SCALAR_TRUE; }
  if (!lhs && !rhs) return SetError(AMDARROW_INVALID, "boolean: need at least one array operand");
  if ((lhs && lhs->type_id != AMDARROW_BOOL) || (rhs && rhs->type_id != AMDARROW_BOOL))
    return SetError(AMDARROW_NOT_IMPLEMENTED, "boolean: operands must be boolean");
  if (lhs && rhs && lhs->length != rhs->length)
    return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");  // compute/exec.cc:190-192
  if ((!lhs && (lhs_scalar < 0 || lhs_scalar > 2)) || (!rhs && (rhs_scalar < 0 || rhs_scalar > 2)))
    return SetError(AMDARROW_INVALID, "boolean: bad scalar state");
  const int64_t n = lhs ? lhs->length : rhs->length;
  std::lock_guard<std::mutex> lock(ctx->mu);
  BoolSrc l{lhs ? static_cast<const uint8_t*>(lhs->values) : nullptr, lhs ? lhs->validity : nullptr, lhs ? lhs->offset : 0,
            lhs ? -1 : lhs_scalar};
  BoolSrc r{rhs ? static_cast<const uint8_t*>(rhs->values) : nullptr, rhs ? rhs->validity : nullptr, rhs ? rhs->offset : 0,
            rhs ? -1 : rhs_scalar};
  const int grid = GridFor((n + 63) / 64, kThreads, ctx->num_cus * 8);
  const size_t bm_bytes = static_cast<size_t>((n + 7) / 8);
  if (op <= AMDARROW_XOR) {  // INTERSECTION kernels
    if ((!lhs && lhs_scalar == AMDARROW_SCALAR_NULL) || (op != AMDARROW_INVERT && !rhs && rhs_scalar == AMDARROW_SCALAR_NULL)) {
      // a null scalar: every slot null, the zero-filled data bitmap stays untouched (exec.cc:326-348)
      out->length = n; out->has_validity = 1; out->null_count = n ? n : -1;
      if (n) {
        AA_HIP(hipMemsetAsync(out_bits, 0, bm_bytes, ctx->stream));
        AA_HIP(hipMemsetAsync(out_validity, 0, bm_bytes, ctx->stream));
      }
      return AMDARROW_OK;
    }
    AA_TRY(Propagate(ctx, lhs, rhs, n, out_validity, out));
    if (n == 0) return AMDARROW_OK;
    hipLaunchKernelGGL(k_boolean, dim3(grid), dim3(kThreads), 0, ctx->stream, op, kDataPlain, kValidNone, 0, l, r, n,
                       reinterpret_cast<uint64_t*>(out_bits), static_cast<uint64_t*>(nullptr),
                       static_cast<unsigned long long*>(nullptr));
    AA_HIP(hipGetLastError());
    return AMDARROW_OK;
  }
  // Kleene kernels compute validity themselves (COMPUTED_PREALLOCATE, scalar_boolean.cc:531-541)
  out->length = n; out->has_validity = 0; out->null_count = 0;
  if (n == 0) return AMDARROW_OK;  // SimpleBinary returns at once, codegen_internal.h:474
  // ArrayData::GetNullCount() of an input: the stored count, or (count unknown) whatever the
  // kernel itself counts on the way -- no extra pass, one read-back of three counters.
  auto known_no_nulls = [](const amdarrow_array* a) { return a->validity == nullptr || a->null_count == 0; };
  int data_mode, valid_mode, arr = 0;
  bool decide_validity_from_counts = false;  // has_validity = (an array operand has nulls)
  if (lhs && rhs) {
    if (known_no_nulls(lhs) && known_no_nulls(rhs)) { data_mode = kDataPlain; valid_mode = kValidNone; }  // :205-209
    else { data_mode = kDataKleene; valid_mode = kValidKleene; decide_validity_from_counts = true; }
    // (with unknown counts that turn out to be zero the Kleene formula yields the plain result
    // and an all-ones validity: only the has_validity flag is left to decide)
  } else {
    arr = lhs ? 0 : 1;
    const amdarrow_array* a = lhs ? lhs : rhs;
    int sc = lhs ? rhs_scalar : lhs_scalar;
    bool invert_array = false;
    if (op == AMDARROW_AND_NOT_KLEENE) {
      if (!lhs) invert_array = true;                                   // KleeneAndNot(scalar, array) :396-431
      else if (sc != AMDARROW_SCALAR_NULL) sc = sc == AMDARROW_SCALAR_TRUE ? AMDARROW_SCALAR_FALSE : AMDARROW_SCALAR_TRUE;  // :433-436
    }
    const int absorbing = op == AMDARROW_OR_KLEENE ? AMDARROW_SCALAR_TRUE : AMDARROW_SCALAR_FALSE;
    if (sc == absorbing) {
      data_mode = op == AMDARROW_OR_KLEENE ? kDataOnes : kDataZero;
      valid_mode = kValidNone;
    } else {
      data_mode = invert_array ? kDataCopyInverted : kDataCopy;
      if (sc != AMDARROW_SCALAR_NULL) {
        valid_mode = known_no_nulls(a) ? kValidNone : kValidCopy;
        decide_validity_from_counts = valid_mode == kValidCopy;
      } else {
        valid_mode = kValidDecides;
      }
    }
  }
  unsigned long long* counters = nullptr;
  if (valid_mode != kValidNone) {
    void* base;
    AA_TRY(Scratch(ctx, kCounterBytes, &base));
    counters = static_cast<unsigned long long*>(base);
    AA_TRY(ClearCounters(ctx, counters));
  }
  hipLaunchKernelGGL(k_boolean, dim3(grid), dim3(kThreads), 0, ctx->stream, op, data_mode, valid_mode, arr, l, r, n,
                     reinterpret_cast<uint64_t*>(out_bits), reinterpret_cast<uint64_t*>(out_validity), counters);
  AA_HIP(hipGetLastError());
  if (valid_mode != kValidNone) {
    unsigned long long c[3] = {0, 0, 0};
    AA_TRY(ReadCounters(ctx, counters, 3, c));
    auto stored_nulls = [](const amdarrow_array* a) { return a && a->validity && a->null_count > 0; };
    const bool input_nulls = c[1] > 0 || c[2] > 0 || stored_nulls(lhs) || stored_nulls(rhs);  // GetNullCount trusts a stored count
    if (!decide_validity_from_counts || input_nulls) {
      out->has_validity = 1;
      out->null_count = static_cast<int64_t>(c[0]);
    }  // else: no input has a null -> the reference drops the buffer; out_validity already holds all ones
  }
  return AMDARROW_OK;
}

int amdarrow_validity_test(amdarrow_ctx* ctx, int func, const amdarrow_array* a, uint8_t* out_bits, amdarrow_out* out) {
  if (func != 0 && func != 1) return SetError(AMDARROW_KEY_ERROR, "validity_test: 0 = is_valid, 1 = is_null");
  out->length = a->length; out->has_validity = 0; out->null_count = 0;  // OUTPUT_NOT_NULL, scalar_validity.cc:100
  if (a->length == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  hipLaunchKernelGGL(k_validity_test, dim3(GridFor((a->length + 63) / 64, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0,
                     ctx->stream, func, MayHaveNulls(a) ? a->validity : nullptr, a->offset, a->length,
                     reinterpret_cast<uint64_t*>(out_bits));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_is_nan(amdarrow_ctx* ctx, const amdarrow_array* a, uint8_t* out_bits, uint8_t* out_validity, amdarrow_out* out) {
  if (a->type_id != AMDARROW_FLOAT && a->type_id != AMDARROW_DOUBLE)
    return SetError(AMDARROW_NOT_IMPLEMENTED, "is_nan: floating point only");  // scalar_validity.cc:111-126
  std::lock_guard<std::mutex> lock(ctx->mu);
  AA_TRY(Propagate(ctx, a, nullptr, a->length, out_validity, out));
  if (a->length == 0) return AMDARROW_OK;
  const int grid = GridFor(a->length, kThreads, ctx->num_cus * 8);
  if (a->type_id == AMDARROW_FLOAT)
    hipLaunchKernelGGL(k_is_nan<float>, dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const float*>(a->values),
                       a->offset, a->length, reinterpret_cast<uint64_t*>(out_bits));
  else
    hipLaunchKernelGGL(k_is_nan<double>, dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const double*>(a->values),
                       a->offset, a->length, reinterpret_cast<uint64_t*>(out_bits));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_fill_null(amdarrow_ctx* ctx, const amdarrow_array* a, const void* fill, void* out_values, uint8_t* out_validity,
                       amdarrow_out* out) {
  const int bw = BitWidth(a->type_id);
  if (bw < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "fill_null: type");
  const int64_t n = a->length;
  out->length = n; out->has_validity = 0; out->null_count = 0;
  if (n == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const bool may = MayHaveNulls(a);
  const bool replace = may && fill != nullptr;  // scalar_fill_null.cc:55, :107
  const uint8_t* valid = replace ? a->validity : nullptr;
  if (bw == 1) {
    // bits: out = valid ? data : fill (scalar_fill_null.cc:100-139)
    const int fill_bit = fill && *static_cast<const uint8_t*>(fill) != 0;
    BoolSrc src{static_cast<const uint8_t*>(a->values), valid, a->offset, -1};
    BoolSrc fs{nullptr, nullptr, 0, fill_bit ? AMDARROW_SCALAR_TRUE : AMDARROW_SCALAR_FALSE};
    hipLaunchKernelGGL(k_boolean, dim3(GridFor((n + 63) / 64, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                       AMDARROW_AND, kDataFill, kValidNone, 0, src, fs, n, reinterpret_cast<uint64_t*>(out_values),
                       static_cast<uint64_t*>(nullptr), static_cast<unsigned long long*>(nullptr));
  } else {
    const int grid = GridFor(n, kThreads, ctx->num_cus * 8);
    uint64_t f = 0;
    if (fill) memcpy(&f, fill, bw / 8);
#define AA_FILL(U) hipLaunchKernelGGL(k_fill_null<U>, dim3(grid), dim3(kThreads), 0, ctx->stream, static_cast<const U*>(a->values), valid, a->offset, n, static_cast<U>(f), static_cast<U*>(out_values))
    switch (bw) {
      case 8: AA_FILL(uint8_t); break;
      case 16: AA_FILL(uint16_t); break;
      case 32: AA_FILL(uint32_t); break;
      default: AA_FILL(uint64_t); break;
    }
#undef AA_FILL
  }
  AA_HIP(hipGetLastError());
  if (replace || !may) return AllValid(ctx, n, out_validity);
  // the result IS the input: validity normalised to offset 0, count as stored (ArrayData copy :84)
  out->has_validity = 1;
  out->null_count = a->null_count;
  hipLaunchKernelGGL(k_validity, dim3(GridFor((n + 63) / 64, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     a->validity, a->offset, static_cast<const uint8_t*>(nullptr), int64_t(0), 0, n,
                     reinterpret_cast<uint64_t*>(out_validity));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_cast_numeric(amdarrow_ctx* ctx, const amdarrow_array* a, int to_type_id, int safe, void* out_values) {
  if (a->length == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  switch (a->type_id) {
    case AMDARROW_UINT8: return CastFrom<uint8_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_INT8: return CastFrom<int8_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_UINT16: return CastFrom<uint16_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_INT16: return CastFrom<int16_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_UINT32: return CastFrom<uint32_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_INT32: return CastFrom<int32_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_UINT64: return CastFrom<uint64_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_INT64: return CastFrom<int64_t>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_FLOAT: return CastFrom<float>(ctx, a, to_type_id, safe, out_values);
    case AMDARROW_DOUBLE: return CastFrom<double>(ctx, a, to_type_id, safe, out_values);
    default: return SetError(AMDARROW_NOT_IMPLEMENTED, "cast: source type");
  }
}

}  // extern "C"

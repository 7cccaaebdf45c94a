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
enum { kErrOverflow = 1, kErrDivZero = 2 };

template <typename T, typename U, bool kSigned>
__device__ __forceinline__ T IntOp(int op, T l, T r, bool valid, int* err) {
  switch (op) {
    case AMDARROW_ADD: return static_cast<T>(static_cast<U>(static_cast<U>(l) + static_cast<U>(r)));
    case AMDARROW_SUB: return static_cast<T>(static_cast<U>(static_cast<U>(l) - static_cast<U>(r)));
    case AMDARROW_MUL: return static_cast<T>(static_cast<uint64_t>(static_cast<U>(l)) * static_cast<uint64_t>(static_cast<U>(r)));
    default: break;
  }
  if (!valid) return 0;  // ScalarBinaryNotNull::WriteNull
  T res = 0;
  switch (op) {
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
template <typename T>
__device__ __forceinline__ T FloatOp(int op, T l, T r, bool valid, int* err) {
  switch (op) {
    case AMDARROW_ADD: return l + r;
    case AMDARROW_SUB: return l - r;
    case AMDARROW_MUL: return l * r;
    default: break;
  }
  if (!valid) return 0;
  switch (op) {
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
    const bool valid = op < AMDARROW_DIV ? true : GetBit(out_valid, i);
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
  if (n == 0) return AMDARROW_OK;
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

}  // namespace amdarrow

using namespace amdarrow;

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
  AA_TRY(CheckBinary(lhs, lhs_scalar, rhs, rhs_scalar, &type, &n));
  if (type == AMDARROW_BOOL) return SetError(AMDARROW_NOT_IMPLEMENTED, "arithmetic on boolean");
  if (op < AMDARROW_ADD || op > AMDARROW_DIV_CHECKED) return SetError(AMDARROW_INVALID, "arith: bad operator");
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
    if (e) return SetError(AMDARROW_INVALID, (e & 3) == kErrDivZero ? "divide by zero" : "overflow");
  }
  return AMDARROW_OK;
}

}  // extern "C"

/*
 * amdarrow.h -- C ABI of the MI355X (gfx950) compute-kernel backend for the Arrow
 * hot path: Filter/Take, sort_indices, hash_count/hash_sum/hash_mean over GroupBy,
 * compare and arithmetic.  This is the drop-in boundary: it sits directly below
 * `kernel->exec` of the reference's kernels (compute/kernel.h:94 ArrayKernelExec) and
 * is what the host C++ layer (arrow-1_amd/host/, or the reference itself through the
 * stub in INTEGRATION.md) binds.  Plain pointers and sizes only.
 *
 * All array pointers are DEVICE pointers (HBM) unless a parameter says "host".
 * Every entry point returns an arrow::StatusCode value (arrow/status.h:77-99); the
 * message of the last failure on the calling thread is amdarrow_last_error().
 *
 * Reference files cited as file:line are relative to /root/reference/cpp/src/arrow/.
 */
#ifndef AMDARROW_H_
#define AMDARROW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMDARROW_ABI_VERSION 1

/* arrow::StatusCode, status.h:77-99 */
enum {
  AMDARROW_OK = 0,
  AMDARROW_OUT_OF_MEMORY = 1,
  AMDARROW_KEY_ERROR = 2,
  AMDARROW_TYPE_ERROR = 3,
  AMDARROW_INVALID = 4,
  AMDARROW_CAPACITY_ERROR = 6,
  AMDARROW_INDEX_ERROR = 7,
  AMDARROW_UNKNOWN_ERROR = 9,
  AMDARROW_NOT_IMPLEMENTED = 10
};

/* arrow::Type::type, type_fwd.h:277-313 (the primitive types on the path) */
enum {
  AMDARROW_BOOL = 1, AMDARROW_UINT8 = 2, AMDARROW_INT8 = 3, AMDARROW_UINT16 = 4,
  AMDARROW_INT16 = 5, AMDARROW_UINT32 = 6, AMDARROW_INT32 = 7, AMDARROW_UINT64 = 8,
  AMDARROW_INT64 = 9, AMDARROW_FLOAT = 11, AMDARROW_DOUBLE = 12
};

enum { AMDARROW_DROP = 0, AMDARROW_EMIT_NULL = 1 };      /* FilterOptions::NullSelectionBehavior, compute/api_vector.h:35-42 */
enum { AMDARROW_ASCENDING = 0, AMDARROW_DESCENDING = 1 }; /* SortOrder, compute/api_vector.h:61-64 */
enum { AMDARROW_COUNT_NON_NULL = 0, AMDARROW_COUNT_NULL = 1 }; /* CountOptions::Mode, compute/api_aggregate.h:47-52 */
/* CompareOperator, compute/api_scalar.h:124-131 */
enum { AMDARROW_EQ = 0, AMDARROW_NE, AMDARROW_GT, AMDARROW_GE, AMDARROW_LT, AMDARROW_LE };
/* add/subtract/multiply/divide[_checked], compute/kernels/scalar_arithmetic.cc:69-251 */
enum {
  AMDARROW_ADD = 0, AMDARROW_SUB, AMDARROW_MUL, AMDARROW_DIV,
  AMDARROW_ADD_CHECKED, AMDARROW_SUB_CHECKED, AMDARROW_MUL_CHECKED, AMDARROW_DIV_CHECKED,
  AMDARROW_POWER, AMDARROW_POWER_CHECKED,  /* scalar_arithmetic.cc:296-356 */
  AMDARROW_NEGATE, AMDARROW_NEGATE_CHECKED /* :253-294; unary: rhs / rhs_scalar are ignored (may be NULL) */
};
enum { AMDARROW_HASH_SUM = 0, AMDARROW_HASH_COUNT = 1, AMDARROW_HASH_MEAN = 2 };

/* One primitive array = compute::internal::PrimitiveArg (compute/kernels/util_internal.h:35-47)
 * before GetPrimitiveArg pre-offsets `data`: `values` is the base of buffers[1] (bit-packed
 * for BOOL), `validity` the base of buffers[0] or NULL, `offset` in elements applies to
 * both, `null_count` is ArrayData::null_count as stored (-1 = kUnknownNullCount; ignored
 * when validity == NULL).  Buffers must be readable up to the next 8-byte boundary past
 * their last element (Arrow buffers are padded to 64 bytes, memory_pool.cc:101). */
typedef struct {
  const void* values;
  const uint8_t* validity;
  int64_t length;
  int64_t offset;
  int32_t type_id;
  int64_t null_count;
} amdarrow_array;

/* Shape of a produced array (what the reference sets on the output ArrayData). */
typedef struct {
  int64_t length;
  int64_t null_count;   /* -1 = kUnknownNullCount */
  int32_t has_validity; /* 0 => the reference leaves buffers[0] absent */
} amdarrow_out;

typedef struct amdarrow_ctx amdarrow_ctx;

/* ---- runtime ------------------------------------------------------------------------- */
int amdarrow_abi_version(void);
const char* amdarrow_last_error(void);
int amdarrow_device_count(int* out_count);
/* One context = one device + one HIP stream + a grow-only HBM scratch arena + a pinned
 * host word for size read-backs.  `stream` is an existing hipStream_t (the caller's
 * current stream; NULL = the device's default stream); with `private_stream` != 0 the
 * argument is ignored and the context creates (and owns) a non-blocking stream.  A context serialises its
 * callers (KernelExecutor::Execute is "not thread-safe", compute/exec_internal.h:116-118);
 * use one context per host thread for concurrent CallFunction. */
int amdarrow_ctx_create(int device, void* stream, int private_stream, amdarrow_ctx** out_ctx);
int amdarrow_ctx_destroy(amdarrow_ctx* ctx);
int amdarrow_ctx_synchronize(amdarrow_ctx* ctx);
void* amdarrow_ctx_stream(amdarrow_ctx* ctx);
/* Device memory pool with MemoryPool's two counters (memory_pool.h:38-105). Allocations
 * are 256-byte aligned and zero-padded to a multiple of 64 bytes. */
int amdarrow_malloc(amdarrow_ctx* ctx, int64_t nbytes, void** out_ptr);
int amdarrow_free(amdarrow_ctx* ctx, void* ptr);
int64_t amdarrow_bytes_allocated(amdarrow_ctx* ctx);
int64_t amdarrow_max_memory(amdarrow_ctx* ctx);
/* amdarrow_free keeps freed blocks by size class for reuse (a hipFree synchronises the device); the two counters
 * above count live buffers only.  The cache is bounded (an eighth of the device's memory unless
 * AMDARROW_POOL_CACHE_LIMIT / amdarrow_pool_set_cache_limit say otherwise), is given back by itself whenever an
 * allocation of the library would otherwise fail, and on request: amdarrow_pool_trim is MemoryPool::ReleaseUnused
 * (memory_pool.h:92-96). */
int64_t amdarrow_pool_cached_bytes(amdarrow_ctx* ctx);
int amdarrow_pool_trim(amdarrow_ctx* ctx, int64_t* out_freed_bytes /* may be NULL */);
int amdarrow_pool_set_cache_limit(amdarrow_ctx* ctx, int64_t bytes);
int amdarrow_memcpy_h2d(amdarrow_ctx* ctx, void* dst_dev, const void* src_host, int64_t nbytes);
int amdarrow_memcpy_d2h(amdarrow_ctx* ctx, void* dst_host, const void* src_dev, int64_t nbytes);
int amdarrow_memcpy_d2d(amdarrow_ctx* ctx, void* dst_dev, const void* src_dev, int64_t nbytes); /* async on the stream */
int amdarrow_memset(amdarrow_ctx* ctx, void* dst_dev, int value, int64_t nbytes);

/* ---- bit utilities (util/bitmap_ops.h) -------------------------------------------------- */
/* CountSetBits(bitmap, offset, length) -> *out_count (host). */
int amdarrow_count_set_bits(amdarrow_ctx* ctx, const uint8_t* bitmap, int64_t offset,
                            int64_t length, int64_t* out_count);

/* ---- vector_selection.cc ---------------------------------------------------------------- */
/* GetFilterOutputSize, compute/kernels/vector_selection.cc:61-88 -> *out_length (host). */
int amdarrow_filter_output_size(amdarrow_ctx* ctx, const amdarrow_array* filter,
                                int null_selection, int64_t* out_length);
/* PrimitiveFilter, vector_selection.cc:786-835 (+PrimitiveFilterImpl :569-784): stream
 * compaction of 1/8/16/32/64-bit values and their validity.  out_values must hold
 * `out_capacity` elements, out_validity BytesForBits(out_capacity) bytes; both are
 * fully written (bitmap tails zeroed).  Fails with AMDARROW_CAPACITY_ERROR when the
 * selection does not fit.  out_validity may be NULL iff out->has_validity would be 0
 * (neither input has a validity buffer with nonzero null_count). */
int amdarrow_filter(amdarrow_ctx* ctx, const amdarrow_array* values,
                    const amdarrow_array* filter, int null_selection, void* out_values,
                    uint8_t* out_validity, int64_t out_capacity, amdarrow_out* out);
/* GetTakeIndices, vector_selection.cc:92-237: uint16 indices when filter.length <= 65535,
 * else uint32 (*out_index_width = 16|32). */
int amdarrow_take_indices(amdarrow_ctx* ctx, const amdarrow_array* filter, int null_selection,
                          void* out_indices, uint8_t* out_validity, int64_t out_capacity,
                          int* out_index_width, amdarrow_out* out);
/* BinaryArray / StringArray with 32-bit offsets (array/array_binary.h; buffers validity | offsets | data). */
typedef struct amdarrow_binary_array {
  const int32_t* offsets;   /* (offset + length + 1) entries */
  const uint8_t* data;
  const uint8_t* validity;  /* LSB bitmap or NULL */
  int64_t length;
  int64_t offset;
  int64_t null_count;       /* -1 = unknown */
} amdarrow_binary_array;
/* Take of variable-width values -- VarBinaryImpl over Selection::VisitTake, vector_selection.cc:
 * 1203-1320, 1493-1573: out[i] = values[indices[i]]; null and zero bytes long if the index or the
 * value is null; validity always written; IndexError as amdarrow_take; "Take operation overflowed binary
 * array capacity" (Invalid) past INT32_MAX - 1 bytes.  Two calls, because the byte count is a result:
 *   offsets  out_offsets[indices.length + 1], out_validity, out->null_count, *out_data_bytes;
 *   data     the bytes into out_data (>= *out_data_bytes), with the offsets of the first call.
 * Filter of such values is this Take over amdarrow_take_indices(filter) (the identity the reference's
 * ValidateFilter checks, vector_selection_test.cc:74-100); BinaryFilter's validity / null_count
 * conventions (:1085-1124) are applied by the host kernel. */
int amdarrow_binary_take_offsets(amdarrow_ctx* ctx, const amdarrow_binary_array* values,
                                 const amdarrow_array* indices, int boundscheck, int32_t* out_offsets,
                                 uint8_t* out_validity, int64_t* out_data_bytes, amdarrow_out* out);
int amdarrow_binary_take_data(amdarrow_ctx* ctx, const amdarrow_binary_array* values,
                              const amdarrow_array* indices, const int32_t* out_offsets,
                              uint8_t* out_data);

/* ---- the other array_take / array_filter kernels (vector_selection.cc:2133-2176) ------------------------------
 * fixed_size_binary / decimal128 / decimal256 (FSBImpl :1575-1606), large_binary / large_utf8
 * (VarBinaryImpl<LargeBinaryType> :1493-1573), list / large_list / map (ListImpl :1608-1680), fixed_size_list (FSLImpl
 * :1682-1724) and struct (StructImpl :1736-1763) all run on Selection::VisitTake (:1214-1262): an output row is valid
 * iff its index is valid and the value it names is valid, the output ALWAYS carries a validity bitmap and an exact
 * null count (FinishCommon :1205-1210), indices are bounds-checked first (TakeExec :1330-1345, IndexError).  Here
 * that visit is ONE call, amdarrow_take_rows, which resolves every output row to the source row it copies (-1 = null)
 * -- for a struct that is the whole kernel, its fields are taken with the same indices --, and the per-layout calls
 * below consume the resolved rows.  Filter of these types is the same Take over amdarrow_take_indices(filter)
 * (FilterExec visits the same rows; StructFilter :1765-1781 does literally that).  dictionary / extension / null
 * arrays need no device call of their own (DictionaryTake :1151-1160 takes the indices, NullTake :1132-1140). */
int amdarrow_take_rows(amdarrow_ctx* ctx, const uint8_t* values_validity /* may be NULL */, int64_t values_length,
                       int64_t values_offset, int64_t values_null_count, const amdarrow_array* indices, int boundscheck,
                       int64_t* out_src_rows /* indices.length */, uint8_t* out_validity, amdarrow_out* out);
/* FSBImpl: out row i = the byte_width bytes of values row (values_offset + src_rows[i]), zero bytes for a null row. */
int amdarrow_gather_fixed_width(amdarrow_ctx* ctx, const void* values, int64_t values_offset, int byte_width,
                                const int64_t* src_rows, int64_t n, void* out_values);
/* An offsets-based array: (large_)binary / utf8 (`data` = the bytes) or (large_)list / map (`data` unused). */
typedef struct amdarrow_var_array {
  const void* offsets;      /* int32_t or int64_t, (offset + length + 1) entries */
  int32_t offset_bits;      /* 32 or 64 */
  const uint8_t* data;
  const uint8_t* validity;  /* LSB bitmap or NULL */
  int64_t length;
  int64_t offset;
  int64_t null_count;
} amdarrow_var_array;
/* out_offsets[n + 1] (of values->offset_bits) = exclusive scan of the taken slots' lengths, a null row being empty;
 * *out_total = the bytes (binary) / child rows (list) to allocate.  Invalid("Take operation overflowed binary array
 * capacity") when a 32-bit total passes INT32_MAX - 1 (:1540-1544). */
int amdarrow_var_take_offsets(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n,
                              void* out_offsets, int64_t* out_total);
/* VarBinaryImpl: the bytes behind those offsets. */
int amdarrow_var_take_data(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n,
                           const void* out_offsets, uint8_t* out_data);
/* ListImpl: the child indices (int32 for 32-bit offsets, int64 for 64-bit ones: TypeTraits<Type>::OffsetBuilderType)
 * the caller hands to Take(values.values(), child_indices, NoBoundsCheck) (:1667-1670). */
int amdarrow_list_take_child_indices(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n,
                                     const void* out_offsets, void* out_child_indices);
/* FSLImpl: n * list_size int64 child indices, (values_offset + src) * list_size + t, NULL (value 0) for every element
 * of a null row; out_index_validity in whole 8-byte words.  The reference computes index * list_size without the
 * array's offset (:1704) -- a defect for sliced inputs that is not replicated; parity is pinned at offset 0. */
int amdarrow_fixed_size_list_child_indices(amdarrow_ctx* ctx, const int64_t* src_rows, int64_t n, int list_size,
                                           int64_t values_offset, int64_t* out_indices, uint8_t* out_index_validity);

/* Concatenate of BinaryArrays / StringArrays -- ConcatenateImpl::Visit(BinaryType) + ConcatenateOffsets,
 * array/concatenate.cc:100-160, 203-214: `size` reports the rows and data bytes (only the byte range every
 * chunk references is copied), then the call proper fills out_offsets[rows + 1], out_data and (iff the summed
 * null count != 0) out_validity in whole 8-byte words. */
int amdarrow_binary_concatenate_size(amdarrow_ctx* ctx, const amdarrow_binary_array* chunks, int nchunks,
                                     int64_t* out_rows, int64_t* out_data_bytes);
int amdarrow_binary_concatenate(amdarrow_ctx* ctx, const amdarrow_binary_array* chunks, int nchunks,
                                int32_t* out_offsets, uint8_t* out_data, uint8_t* out_validity,
                                amdarrow_out* out);

/* Concatenate, array/concatenate.cc:171-201 -- what TakeCA/TakeCC (vector_selection.cc:1923-
 * 1959) do to chunked values first.  out_values: sum(length) elements (bits for BOOL).
 * out->null_count = sum of the chunks' counts, -1 if any is unknown (:172-180); the validity
 * bitmap is written iff that is non-zero (:190) and out_validity is given.  Bitmaps are
 * written in whole 8-byte words: size them to a multiple of 8 bytes. */
int amdarrow_concatenate(amdarrow_ctx* ctx, const amdarrow_array* chunks, int nchunks,
                         void* out_values, uint8_t* out_validity, amdarrow_out* out);
/* PrimitiveTake, vector_selection.cc:493-530 (+CheckIndexBounds util/int_util.cc:534-561):
 * out[i] = values[indices[i]], null if index or value is null; validity always written;
 * AMDARROW_INDEX_ERROR "Index N out of bounds" when boundscheck != 0. */
int amdarrow_take(amdarrow_ctx* ctx, const amdarrow_array* values,
                  const amdarrow_array* indices, int boundscheck, void* out_values,
                  uint8_t* out_validity, amdarrow_out* out);

/* ---- vector_sort.cc ------------------------------------------------------------------------ */
/* array_sort_indices / ChunkedArraySorter, vector_sort.cc:373-575, 657-832: stable argsort
 * of the logical concatenation of `chunks`; non-null non-NaN values by order, then NaNs,
 * then nulls, ties by ascending index in both orders.  out_indices: sum(length) uint64. */
int amdarrow_sort_indices(amdarrow_ctx* ctx, const amdarrow_array* chunks, int nchunks,
                          int order, uint64_t* out_indices);
/* sort_indices over a record batch / table with several sort keys:
 * RadixRecordBatchSorter vector_sort.cc:955-1045, MultipleKeyRecordBatchSorter :1184-1351,
 * MultipleKeyTableSorter :1381-1549 (one order, defined by MultipleKeyComparator :1049-1180):
 * lexicographic over the keys; within a key values by orders[k], then NaNs, then nulls; rows
 * equal on every key by ascending index.  `chunks` holds key 0's nchunks[0] chunks, then
 * key 1's, ...; all keys span the same number of rows.  out_indices: that many uint64. */
int amdarrow_sort_indices_multi(amdarrow_ctx* ctx, const amdarrow_array* chunks,
                                const int* nchunks, const int* orders, int nkeys,
                                uint64_t* out_indices);

/* ---- scalar_compare.cc / scalar_arithmetic.cc ---------------------------------------------- */
/* Operand = array (lhs / rhs) or valid scalar (lhs_scalar / rhs_scalar: HOST pointer to one
 * value of the array operand's type).  Output at bit/element offset 0 as ScalarExecutor
 * preallocates it (compute/exec.cc:686-716); out_validity is always written
 * (NullPropagator, exec.cc:295-461). */
int amdarrow_compare(amdarrow_ctx* ctx, int op, const amdarrow_array* lhs,
                     const void* lhs_scalar, const amdarrow_array* rhs, const void* rhs_scalar,
                     uint8_t* out_bits, uint8_t* out_validity, amdarrow_out* out);
/* Unchecked add/sub/mul write every slot (ScalarBinary, codegen_internal.h:758-824);
 * divide and *_checked write 0 at null slots (ScalarBinaryNotNull, :828-932) and fail with
 * AMDARROW_INVALID "overflow" / "divide by zero".  power is unchecked in the same sense (every slot, integer
 * powers wrap in uint64) yet fails with "integers to negative integer powers are not allowed" for a negative
 * exponent in ANY slot, as the reference does; floating point powers are pow(): float64 within 2 ulp, float32
 * within 1 ulp of the reference's libm result (everything else on this entry point is bit-exact). */
int amdarrow_arith(amdarrow_ctx* ctx, int op, const amdarrow_array* lhs,
                   const void* lhs_scalar, const amdarrow_array* rhs, const void* rhs_scalar,
                   void* out_values, uint8_t* out_validity, amdarrow_out* out);

/* cast between numbers: compute/kernels/scalar_cast_numeric.cc:38-50 (CastIntegerToInteger,
 * CastFloatingToFloating) and :203-262 (CastIntegerToFloating) -- what Function::Execute's implicit casts
 * (function.cc:175-176, cast.cc:228-243, after DispatchBest / CommonNumeric codegen_internal.cc:220-268)
 * and an explicit "cast" run.  out_values: values.length elements of to_type_id at element offset 0; the
 * result's validity IS the input's (a slice of it), so none is written.  safe != 0 = CastOptions::Safe:
 * AMDARROW_INVALID "Integer value V not in range: L to U" for the first valid slot that the target cannot
 * hold (int_util.cc:600-618; integer -> floating: |V| <= 2^24 / 2^53).  Floating point -> integer is
 * AMDARROW_NOT_IMPLEMENTED (its out-of-range conversion is undefined behaviour in the reference, :53-60). */
int amdarrow_cast_numeric(amdarrow_ctx* ctx, const amdarrow_array* values, int to_type_id, int safe,
                          void* out_values);

/* ---- hash_aggregate.cc ---------------------------------------------------------------------- */
/* Grouper (hash_aggregate.cc:313-437): keys -> dense uint32 ids in FIRST-APPEARANCE order,
 * stateful across Consume calls.  Key columns: fixed width, total <= 8 value bytes. */
typedef struct amdarrow_grouper amdarrow_grouper;
int amdarrow_grouper_create(amdarrow_ctx* ctx, int nkeys, const int* key_type_ids,
                            amdarrow_grouper** out);
int amdarrow_grouper_destroy(amdarrow_grouper* g);
int amdarrow_grouper_consume(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids);
/* Ids of rows whose keys the grouper has all seen (no counterpart in the reference, whose Consume always may insert,
 * :358-408): nothing is inserted and no table is sized for `length` new groups -- what a caller that knows the groups
 * (the fused GroupBy serving hash_min_max through ids) needs.  AMDARROW_KEY_ERROR if some key is unknown. */
int amdarrow_grouper_lookup(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids);
int amdarrow_grouper_num_groups(amdarrow_grouper* g, uint32_t* out_num_groups);
/* GetUniques (:410-428): key column k -> num_groups values + validity bitmap. */
int amdarrow_grouper_uniques(amdarrow_grouper* g, int k, void* out_values,
                             uint8_t* out_validity, amdarrow_out* out);
/* Grouper over ONE variable-width key column -- VarLengthKeyEncoder<BinaryType / LargeBinaryType>, hash_aggregate.cc:
 * 237-311, behind GrouperImpl :313-437 (binary / utf8: offset_bits 32, large_binary / large_utf8: 64).  Keys are equal
 * iff their bytes and lengths are; a null key is its own group; ids in first-appearance order across Consume calls;
 * the uniques come back in id order with a validity bitmap only when the null group exists (DecodeNulls :64-94).
 * Rows are grouped by a 64-bit hash of their bytes and EVERY row is then compared with its group's stored key: should
 * two different keys ever share a hash the table is rebuilt under another seed from the stored keys and the batch
 * redone (exact, or -- after four seeds -- a loud error).  Key columns of several kinds combine on the host: a
 * variable-width column enters the fixed-width amdarrow_grouper as its uint32 ids (equal rows <=> equal id tuples),
 * a dictionary column as its indices (DictionaryKeyEncoder :198-235). */
typedef struct amdarrow_binary_grouper amdarrow_binary_grouper;
int amdarrow_binary_grouper_create(amdarrow_ctx* ctx, int offset_bits, amdarrow_binary_grouper** out);
int amdarrow_binary_grouper_destroy(amdarrow_binary_grouper* g);
int amdarrow_binary_grouper_consume(amdarrow_binary_grouper* g, const amdarrow_var_array* keys, uint32_t* out_ids);
int amdarrow_binary_grouper_num_groups(amdarrow_binary_grouper* g, uint32_t* out_num_groups);
int amdarrow_binary_grouper_uniques_size(amdarrow_binary_grouper* g, int64_t* out_rows, int64_t* out_data_bytes);
int amdarrow_binary_grouper_uniques(amdarrow_binary_grouper* g, void* out_offsets /* rows + 1, of offset_bits */,
                                    uint8_t* out_data, uint8_t* out_validity /* whole 8-byte words */, amdarrow_out* out);
/* Grouper::MakeGroupings (hash_aggregate.cc:979-1018, api_aggregate.h:360-377): the ListArray<int32> that lists, per
 * group, the rows carrying its id in row order -- given as its two buffers.  ids: device uint32[n], every id <
 * num_groups (AMDARROW_INVALID otherwise; the reference only DCHECKs); out_offsets: int32[num_groups + 1];
 * out_indices: int32[n].  ApplyGroupings (:968-977) is take(values, out_indices) under the same offsets. */
int amdarrow_make_groupings(amdarrow_ctx* ctx, const uint32_t* ids, int64_t n, uint32_t num_groups,
                            int32_t* out_offsets, int32_t* out_indices);
/* One HashAggregateKernel consume+finalize over one batch (hash_aggregate.cc:466-622,
 * 928-960): func = AMDARROW_HASH_{SUM,COUNT,MEAN}; `ids` device uint32[values.length];
 * out_values: num_groups x 8 bytes (int64 / uint64 / double per FindAccumulatorType,
 * aggregate_internal.h:29-50; hash_mean = double(sum)/count, null when count == 0). */
int amdarrow_hash_aggregate(amdarrow_ctx* ctx, int func, const amdarrow_array* values,
                            const uint32_t* ids, int64_t num_groups, int count_mode,
                            void* out_values, uint8_t* out_validity, amdarrow_out* out);

/* The fused Filter -> GroupBy over SEVERAL key columns (GrouperImpl with one KeyEncoder per column,
 * hash_aggregate.cc:313-437; rows of the result as amdarrow_filter_groupby, one keys / validity buffer per key column
 * plus its null count).  The columns' values are packed into one 32-bit key word: column c contributes
 * value - min(column c), with one more code for null, in just enough bits for its range (one min / max pass per key
 * column), the packed column is materialised once (4 bytes per row) and the single-key machinery runs on it; the
 * group keys are decoded back into their columns at the end.  AMDARROW_NOT_IMPLEMENTED when the columns cannot be
 * packed -- float keys, more than 4 columns, ranges that need more than 32 bits together: the caller then takes the
 * Grouper + hash_accumulate route, which handles every shape. */
int amdarrow_filter_groupby_multi(amdarrow_ctx* ctx, const amdarrow_array* values,
                                  const amdarrow_array* keys, int nkeys, const amdarrow_array* mask,
                                  int64_t group_capacity, void* out_sums, int64_t* out_counts,
                                  void* const* out_keys, uint8_t* const* out_key_validity,
                                  int64_t* out_key_null_counts, uint32_t* out_first_rows,
                                  int64_t* out_num_groups);

/* The device half of the multi-GPU group-by merge (protocol in arrow-1_amd/dist.py): the
 * reference has no counterpart -- its GroupBy is single-process (hash_aggregate.cc:890-966); the
 * merged table is what that GroupBy returns on the concatenated shards, first-appearance order
 * (:391-401) included.  Dense protocol for integer keys of a small global range [kmin, kmin + R):
 *   scatter  this rank's partial table (keys / validity bytes / 8-byte sums / int64 counts, in local
 *            first-appearance order) into acc[R + 2][2] ({sum, count} per key, 16 bytes side by side;
 *            column R = the null key, acc[R + 1][0] = keys outside the range) and owner[R + 2]: owner[col] = rank << 32 | local id
 *            (INT64_MAX where absent), owner[R + 1] = -(local group count), which the MIN turns into
 *            the stride of the slot array.  The caller then all-reduces acc with SUM and owner with MIN;
 *   order    reads the reduced arrays back out in global order (rank-major, local id minor) into
 *            out_* (capacity rows) and stores the group count to a DEVICE word: no host round trip.
 * counts_as_double: acc holds float64 (sums of a float column; counts ride along as exact doubles).
 * Both calls are stream-ordered. */
int amdarrow_dense_merge_scatter(amdarrow_ctx* ctx, const void* keys, int key_type,
                                 const uint8_t* key_valid /* NULL = all valid */,
                                 int valid_is_bitmap /* 1: LSB-first bits, 0: one byte per group */,
                                 const void* sums,
                                 const int64_t* counts, int64_t num_groups, int64_t kmin, int64_t R,
                                 int rank, int counts_as_double, void* acc, void* owner);
int amdarrow_dense_merge_order(amdarrow_ctx* ctx, const void* acc, const void* owner, int64_t kmin,
                               int64_t R, int world, int64_t capacity, int key_type,
                               int counts_as_double, int64_t out_capacity, void* out_keys,
                               uint8_t* out_key_valid_bytes, void* out_sums, int64_t* out_counts,
                               int64_t* out_num_groups_dev);

/* dictionary_encode's index column under DictionaryEncodeOptions::MASK (vector_hash.cc:165-225):
 * `ids` = the Grouper's ids of the n rows, `uniques_validity` = the bitmap amdarrow_grouper_uniques
 * wrote (NULL when no key is null), num_groups its length.  out_indices[i] = the row's position in
 * the dictionary WITHOUT the null entry (ids past the null group move down by one), int32; rows of
 * the null group get a null index (value 0).  out_validity: whole 8-byte words; out->has_validity
 * iff some row is null.  unique / value_counts / ENCODE need no extra kernel: they are the
 * uniques, the uniques + hash_count, and the ids themselves. */
int amdarrow_dictionary_indices(amdarrow_ctx* ctx, const uint32_t* ids, int64_t n,
                                const uint8_t* uniques_validity, int64_t num_groups,
                                int32_t* out_indices, uint8_t* out_validity, amdarrow_out* out);

/* hash_min_max -- GroupedMinMaxImpl, hash_aggregate.cc:627-768, as the reference's init / resize /
 * consume / finalize (kernel.h:681-721).  The caller owns the state of the reference's four
 * builders (:704-707): mins / maxes (one element of the VALUE type per group) and the has_values /
 * has_nulls bitmaps (zero-filled, sized in whole 8-byte words).
 *   init        groups [first, first+count) get the anti-extremes (mins = max of the type / +inf,
 *               maxes = min of the type / -inf, :629-642, :653-661) -- call when groups are added;
 *   accumulate  one batch: a valid value updates mins[id], maxes[id] and has_values; a null marks
 *               has_nulls (:652-672).  NaN never replaces an extreme (std::min / std::max);
 *   finalize    validity of BOTH children: has_values, under AMDARROW_MINMAX_EMIT_NULL and not
 *               has_nulls (:729-740); mins / maxes are the output buffers as they stand.
 * Integers are bit-exact; float extremes are equal as values (the sign of a zero extreme follows
 * row order in the reference and scheduling here).  BOOL / HALF_FLOAT: NotImplemented (:674-684). */
enum { AMDARROW_MINMAX_SKIP = 0, AMDARROW_MINMAX_EMIT_NULL = 1 }; /* MinMaxOptions::Mode, api_aggregate.h:78-90 */
int amdarrow_hash_minmax_init(amdarrow_ctx* ctx, int value_type_id, void* mins, void* maxes,
                              int64_t first, int64_t count);
int amdarrow_hash_minmax_accumulate(amdarrow_ctx* ctx, const amdarrow_array* values,
                                    const uint32_t* ids, void* mins, void* maxes,
                                    uint8_t* has_values, uint8_t* has_nulls);
int amdarrow_hash_minmax_finalize(amdarrow_ctx* ctx, int64_t num_groups, int null_handling,
                                  const uint8_t* has_values, const uint8_t* has_nulls,
                                  uint8_t* out_validity, amdarrow_out* out);

/* The same kernel split as the reference's HashAggregateKernel consume / finalize pair
 * (compute/kernel.h:681-721) for multi-batch GroupBy: accumulate adds one batch into caller-owned
 * device state -- sums (num_groups x 8 bytes) and counts (num_groups x int64), zeroed by the caller
 * when a group is created; finalize turns the state into the output array IN PLACE in `sums`
 * (hash_count: the counts; hash_mean: double(sum)/count) and writes the validity bitmap. */
int amdarrow_hash_accumulate(amdarrow_ctx* ctx, int func, const amdarrow_array* values,
                             const uint32_t* ids, int count_mode, void* sums, int64_t* counts);
int amdarrow_hash_finalize(amdarrow_ctx* ctx, int func, int value_type_id, int64_t num_groups,
                           void* sums, const int64_t* counts, uint8_t* out_validity,
                           amdarrow_out* out);

/* ---- mask-building scalars ------------------------------------------------------------
 * Boolean kernels: compute/kernels/scalar_boolean.cc:30-566 (registered :521-542 as "invert", "and",
 * "and_not", "or", "xor", "and_kleene", "and_not_kleene", "or_kleene").  Operand i is a boolean
 * array or a scalar given by its state; at least one is an array; AMDARROW_INVERT ignores the right
 * operand.  invert..xor: validity / null_count as amdarrow_compare (NullPropagator; a null scalar
 * makes every slot null and leaves the data bits zero), data bits = the op on the DATA bits of
 * every slot.  *_kleene: validity computed by the kernel (ComputeKleene :34-91 and the scalar
 * special cases); out->has_validity = 0 (and out_validity all ones) exactly where the reference
 * drops the buffer; otherwise out->null_count is exact.  out_bits / out_validity: BytesForBits(n)
 * rounded up to 8 bytes, both required. */
enum { AMDARROW_INVERT = 0, AMDARROW_AND, AMDARROW_AND_NOT, AMDARROW_OR, AMDARROW_XOR, AMDARROW_AND_KLEENE,
       AMDARROW_AND_NOT_KLEENE, AMDARROW_OR_KLEENE };
enum { AMDARROW_SCALAR_NONE = -1, AMDARROW_SCALAR_FALSE = 0, AMDARROW_SCALAR_TRUE = 1, AMDARROW_SCALAR_NULL = 2 };
int amdarrow_boolean(amdarrow_ctx* ctx, int op, const amdarrow_array* lhs, int lhs_scalar, const amdarrow_array* rhs,
                     int rhs_scalar, uint8_t* out_bits, uint8_t* out_validity, amdarrow_out* out);

/* is_valid (func 0) / is_null (func 1): scalar_validity.cc:34-82, any type, result never null
 * (NullHandling::OUTPUT_NOT_NULL :100); constant when the input cannot have nulls
 * (ArrayData::MayHaveNulls). */
int amdarrow_validity_test(amdarrow_ctx* ctx, int func, const amdarrow_array* array, uint8_t* out_bits, amdarrow_out* out);
/* is_nan: scalar_validity.cc:84-89, 111-126 (float32 / float64; op on every slot, validity by
 * NullPropagator). */
int amdarrow_is_nan(amdarrow_ctx* ctx, const amdarrow_array* array, uint8_t* out_bits, uint8_t* out_validity,
                    amdarrow_out* out);
/* fill_null(values, scalar): scalar_fill_null.cc:44-147 (numeric and boolean).  fill = one value of
 * values' type in HOST memory, or NULL for a null scalar.  values may have nulls and fill valid:
 * null slots take the fill value, no validity, null_count 0; otherwise the result is the input
 * (validity normalised to bit offset 0, null_count as stored). */
int amdarrow_fill_null(amdarrow_ctx* ctx, const amdarrow_array* values, const void* fill, void* out_values,
                       uint8_t* out_validity, amdarrow_out* out);

/* ---- scalar aggregates ----------------------------------------------------------------
 * One Consume of count / sum / mean / min_max over an array: compute/kernels/aggregate_basic.cc:58-130,
 * aggregate_basic_internal.h:54-330 (CountImpl, SumImpl, MeanImpl, MinMaxImpl, BooleanMinMaxImpl).
 * valid_count = length - null count; sum / min / max over the valid slots as 64-bit patterns in the
 * accumulator family of the input type (FindAccumulatorType, aggregate_internal.h:29-50): uint64 for
 * unsigned and boolean (sum = true count, min / max = 0 / 1), int64 for signed (sums wrap), double
 * for float32 / float64 (min / max by fmin / fmax: NaN ignored).  No valid slot leaves the identities
 * (sum 0, min = type max / +inf / true, max = type min / -inf / false); the caller turns that into
 * the null scalars of SumImpl / MeanImpl / MinMaxImpl::Finalize and applies MinMaxOptions.  `out` is
 * HOST memory. */
typedef struct amdarrow_agg {
  int64_t valid_count;
  uint64_t sum, min, max;
} amdarrow_agg;
int amdarrow_aggregate(amdarrow_ctx* ctx, const amdarrow_array* array, amdarrow_agg* out);

/* Fused Filter -> GroupBy (BASELINE.json north_star / config 5 on one shard):
 *   Filter(values, mask, DROP), Filter(keys, mask, DROP), GroupBy({values}, {keys},
 *   {hash_sum, hash_count})  (vector_selection.cc:786-835 + hash_aggregate.cc:890-966)
 * in one pass over HBM.  `mask` may be NULL (plain GroupBy).  Outputs, in group
 * (first-appearance) order, each with capacity `group_capacity`:
 *   out_sums 8 B/group, out_counts int64 (valid values per group; sum is null where 0),
 *   out_keys (key type) + out_key_validity, out_first_rows (optional, may be NULL):
 *   row index, within this call's input, of the first selected row of each group --
 *   what a multi-GPU merge orders groups by.
 * *out_num_groups (host) is final on return; the output ARRAYS are complete in stream order
 * (synchronise the context, or consume them on the same stream). */
int amdarrow_filter_groupby(amdarrow_ctx* ctx, const amdarrow_array* values,
                            const amdarrow_array* keys, const amdarrow_array* mask,
                            int64_t group_capacity, void* out_sums, int64_t* out_counts,
                            void* out_keys, uint8_t* out_key_validity,
                            uint32_t* out_first_rows, int64_t* out_num_groups);

/* Diagnostics (tests, bench): how the context's last amdarrow_filter_groupby / _multi / grouped Grouper batch was
 * computed.  *path: 3 = per-workgroup LDS pre-aggregation, 2 = one-pass partition, 1 = exact two-pass partition,
 * 0 = global-atomic table; *record_words: 32-bit words per partitioned record (4 = {key, row, value}; 3 / 2 = the
 * compact records of a call whose predecessor of the same shape had a small key range / int32 values); 0 when no
 * records were written.  No reference counterpart. */
int amdarrow_groupby_last_path(amdarrow_ctx* ctx, int* path, int* record_words);

#ifdef __cplusplus
}
#endif
#endif /* AMDARROW_H_ */

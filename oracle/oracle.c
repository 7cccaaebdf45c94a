/*
 * oracle.c -- CPU ORACLE (test infrastructure only; see oracle.h).
 *
 * Plain-C restatement of the reference's single-threaded kernels.  The restatement is
 * semantic, element at a time: where the reference walks 64-bit BitBlockCounter blocks
 * with fast paths (all-set / none-set), every path computes the same per-element
 * function, which is what is written here.  Line citations are to
 * /root/reference/cpp/src/arrow/compute/kernels/ unless a directory is given.
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[256];
const char* orc_last_error(void) { return g_err; }
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* ---- util/bit_util.h ---------------------------------------------------------- */
static inline int get_bit(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline void set_bit(uint8_t* b, int64_t i) { b[i >> 3] |= (uint8_t)(1u << (i & 7)); }
static inline void clear_bit(uint8_t* b, int64_t i) { b[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
static inline void set_bit_to(uint8_t* b, int64_t i, int v) { if (v) set_bit(b, i); else clear_bit(b, i); }
static inline int64_t bytes_for_bits(int64_t n) { return (n + 7) >> 3; }

int orc_bit_width(int t) {
  switch (t) {
    case ORC_BOOL: return 1;
    case ORC_UINT8: case ORC_INT8: return 8;
    case ORC_UINT16: case ORC_INT16: return 16;
    case ORC_UINT32: case ORC_INT32: case ORC_FLOAT: return 32;
    case ORC_UINT64: case ORC_INT64: case ORC_DOUBLE: return 64;
    default: return -1;
  }
}
static int is_signed_int(int t) { return t == ORC_INT8 || t == ORC_INT16 || t == ORC_INT32 || t == ORC_INT64; }
static int is_unsigned_int(int t) { return t == ORC_UINT8 || t == ORC_UINT16 || t == ORC_UINT32 || t == ORC_UINT64; }
static int is_float(int t) { return t == ORC_FLOAT || t == ORC_DOUBLE; }

/* util/bitmap_ops.cc CountSetBits */
int64_t orc_count_set_bits(const uint8_t* bm, int64_t off, int64_t len) {
  int64_t n = 0;
  for (int64_t i = 0; i < len; ++i) n += get_bit(bm, off + i);
  return n;
}

/* GetPrimitiveArg (util_internal.cc:45-58): null_count is 0 when there is no validity
 * buffer, otherwise the STORED count (possibly -1 = unknown). */
static inline int64_t arg_null_count(const orc_array* a) { return a->validity ? a->null_count : 0; }
/* ArrayData::MayHaveNulls (array/data.h): null_count != 0 && buffers[0] != nullptr */
static inline int may_have_nulls(const orc_array* a) { return a->validity && a->null_count != 0; }
static inline int slot_valid(const orc_array* a, int64_t i) {
  return a->validity ? get_bit(a->validity, a->offset + i) : 1;
}
static inline int bool_at(const orc_array* a, int64_t i) {
  return get_bit((const uint8_t*)a->values, a->offset + i);
}

/* ---- a1: GetFilterOutputSize, vector_selection.cc:61-88 ------------------------- */
int64_t orc_filter_output_size(const orc_array* f, int null_selection) {
  int64_t n = 0;
  if (may_have_nulls(f)) {
    for (int64_t i = 0; i < f->length; ++i) {
      int v = get_bit(f->validity, f->offset + i), d = bool_at(f, i);
      n += (null_selection == ORC_EMIT_NULL) ? (d | !v) /* NextOrNotWord :70-75 */
                                             : (d & v); /* NextAndWord   :77-81 */
    }
  } else {
    n = orc_count_set_bits((const uint8_t*)f->values, f->offset, f->length); /* :85 */
  }
  return n;
}

/* copy element `src_i` of values (bit width bw, element offset applied) to out[dst_i] */
static inline void copy_elem(const orc_array* v, int bw, int64_t src_i, void* out, int64_t dst_i) {
  if (bw == 1) {
    set_bit_to((uint8_t*)out, dst_i, bool_at(v, src_i));
  } else {
    int w = bw / 8;
    memcpy((uint8_t*)out + dst_i * w, (const uint8_t*)v->values + (v->offset + src_i) * w, w);
  }
}
static inline void zero_elem(int bw, void* out, int64_t dst_i) {
  if (bw == 1) clear_bit((uint8_t*)out, dst_i);
  else memset((uint8_t*)out + dst_i * (bw / 8), 0, bw / 8);
}

/* ---- a2: PrimitiveFilter + PrimitiveFilterImpl, vector_selection.cc:569-835 ------- */
int orc_filter(const orc_array* values, const orc_array* filter, int null_selection,
               void* out_values, uint8_t* out_validity, orc_out* out) {
  const int bw = orc_bit_width(values->type_id);
  if (bw < 0) return fail(ORC_NOT_IMPLEMENTED, "filter: unsupported value type");
  if (filter->type_id != ORC_BOOL) return fail(ORC_NOT_IMPLEMENTED, "filter must be boolean");
  if (values->length != filter->length)
    return fail(ORC_INVALID, "Array arguments must all be the same length"); /* exec.cc:190-192 */
  const int64_t v_nc = arg_null_count(values), f_nc = arg_null_count(filter);
  const int64_t out_len = orc_filter_output_size(filter, null_selection); /* :792 */
  /* :799-804 */
  out->null_count = (v_nc == 0 && (null_selection == ORC_DROP || f_nc == 0)) ? 0 : -1;
  out->has_validity = (v_nc != 0 || f_nc != 0); /* :809 */
  out->length = out_len;
  if (bw == 1) memset(out_values, 0, bytes_for_bits(out_len)); /* AllocateBitmap zero-fills */
  if (out->has_validity) memset(out_validity, 0, bytes_for_bits(out_len));

  int64_t pos = 0;
  if (f_nc == 0 && v_nc == 0) {
    /* ExecNonNull :625-630: only the filter's data bits are consulted */
    for (int64_t i = 0; i < values->length; ++i)
      if (bool_at(filter, i)) copy_elem(values, bw, i, out_values, pos++);
    return ORC_OK;
  }
  /* Exec :632-744.  DropNullCounter keys on the PRESENCE of the filter validity buffer
   * (:540-552), as does the per-slot code below it. */
  for (int64_t i = 0; i < values->length; ++i) {
    const int fv = filter->validity ? get_bit(filter->validity, filter->offset + i) : 1;
    const int fd = bool_at(filter, i);
    if (fv && fd) {
      /* WriteNotNull / WriteMaybeNull :646-657: validity follows the value's validity,
       * the value bytes are copied even when the value is null */
      set_bit_to(out_validity, pos, slot_valid(values, i));
      copy_elem(values, bw, i, out_values, pos++);
    } else if (!fv && null_selection == ORC_EMIT_NULL) {
      /* :706-709 / :733-736: emitted null -> validity 0, zeroed value (:743-746) */
      clear_bit(out_validity, pos);
      zero_elem(bw, out_values, pos++);
    }
  }
  return ORC_OK;
}

/* ---- a3: GetTakeIndices[Impl], vector_selection.cc:92-237 ------------------------ */
int orc_take_indices(const orc_array* f, int null_selection, void* out_idx,
                     uint8_t* out_validity, int* out_width, orc_out* out) {
  if (f->length > 0xFFFFFFFFLL)
    return fail(ORC_NOT_IMPLEMENTED, "Filter length exceeds UINT32_MAX, consider a different strategy for selecting elements");
  const int w = (f->length <= 65535) ? 16 : 32; /* :224-229 */
  *out_width = w;
  int64_t n = 0, nulls = 0;
  const int have_nulls = may_have_nulls(f);
#define PUT(v) do { if (w == 16) ((uint16_t*)out_idx)[n] = (uint16_t)(v); else ((uint32_t*)out_idx)[n] = (uint32_t)(v); } while (0)
  if (have_nulls && null_selection == ORC_EMIT_NULL) {
    /* :103-160: builder with validity; null slot value is T{} (UnsafeAppendNull) */
    const int64_t total = orc_filter_output_size(f, null_selection);
    memset(out_validity, 0, bytes_for_bits(total));
    for (int64_t i = 0; i < f->length; ++i) {
      if (get_bit(f->validity, f->offset + i)) {
        if (bool_at(f, i)) { PUT(i); set_bit(out_validity, n); ++n; }
      } else { PUT(0); ++n; ++nulls; }
    }
    out->length = n; out->null_count = nulls; out->has_validity = 1;
    return ORC_OK;
  }
  for (int64_t i = 0; i < f->length; ++i) {
    const int sel = have_nulls ? (get_bit(f->validity, f->offset + i) && bool_at(f, i)) /* :165-197 */
                               : bool_at(f, i);                                          /* :198-209 */
    if (sel) { PUT(i); ++n; }
  }
#undef PUT
  out->length = n; out->null_count = 0; out->has_validity = 0; /* :214-216 */
  return ORC_OK;
}

static inline int64_t load_index(const orc_array* idx, int64_t i, int* negative) {
  const int64_t k = idx->offset + i;
  *negative = 0;
  switch (idx->type_id) {
    case ORC_INT8: { int8_t v = ((const int8_t*)idx->values)[k]; *negative = v < 0; return v; }
    case ORC_INT16: { int16_t v = ((const int16_t*)idx->values)[k]; *negative = v < 0; return v; }
    case ORC_INT32: { int32_t v = ((const int32_t*)idx->values)[k]; *negative = v < 0; return v; }
    case ORC_INT64: { int64_t v = ((const int64_t*)idx->values)[k]; *negative = v < 0; return v; }
    case ORC_UINT8: return ((const uint8_t*)idx->values)[k];
    case ORC_UINT16: return ((const uint16_t*)idx->values)[k];
    case ORC_UINT32: return ((const uint32_t*)idx->values)[k];
    default: return (int64_t)((const uint64_t*)idx->values)[k]; /* u64 >= 2^63 shows as negative */
  }
}

/* ---- a4: PrimitiveTake, vector_selection.cc:273-530 -------------------------------- */
int orc_take(const orc_array* values, const orc_array* idx, int boundscheck, void* out_values,
             uint8_t* out_validity, orc_out* out) {
  const int bw = orc_bit_width(values->type_id);
  if (bw < 0) return fail(ORC_NOT_IMPLEMENTED, "take: unsupported value type");
  if (!is_signed_int(idx->type_id) && !is_unsigned_int(idx->type_id))
    return fail(ORC_NOT_IMPLEMENTED, "take: indices must be integers");
  if (boundscheck) {
    /* CheckIndexBounds, util/int_util.cc:534-561: only VALID index slots are checked */
    for (int64_t i = 0; i < idx->length; ++i) {
      if (!slot_valid(idx, i)) continue;
      int neg; int64_t v = load_index(idx, i, &neg);
      const int oob = idx->type_id == ORC_UINT64 ? ((uint64_t)v >= (uint64_t)values->length)
                                                 : (neg || v >= values->length);
      if (oob) {
        char m[96];
        if (idx->type_id == ORC_UINT64) snprintf(m, sizeof m, "Index %llu out of bounds", (unsigned long long)v);
        else snprintf(m, sizeof m, "Index %lld out of bounds", (long long)v);
        return fail(ORC_INDEX_ERROR, m);
      }
    }
  }
  const int64_t n = idx->length;
  memset(out_validity, 0, bytes_for_bits(n));
  if (bw == 1) memset(out_values, 0, bytes_for_bits(n)); /* :387-388 */
  const int64_t v_nc = arg_null_count(values);
  int64_t valid = 0;
  for (int64_t i = 0; i < n; ++i) {
    int ok = slot_valid(idx, i);
    int64_t j = 0;
    if (ok) {
      int neg; j = load_index(idx, i, &neg);
      /* values' validity is consulted only when values.null_count != 0 (:306, :338) */
      if (v_nc != 0 && !get_bit(values->validity, values->offset + j)) ok = 0;
    }
    if (ok) { copy_elem(values, bw, j, out_values, i); set_bit(out_validity, i); ++valid; }
    else zero_elem(bw, out_values, i); /* out[position] = ValueCType{} */
  }
  out->length = n; out->null_count = n - valid; out->has_validity = 1; /* :367, :505-510 */
  return ORC_OK;
}

/* ---- 8f.4: variable-width Take / Filter, vector_selection.cc:840-1124, 1203-1320, 1493-1573 ---- */
static inline int bin_valid(const orc_binary* v, int64_t i) { return v->validity ? get_bit(v->validity, v->offset + i) : 1; }
static void bin_append(const orc_binary* v, int64_t i, int32_t* out_offsets, uint8_t* out_data, int64_t pos, int32_t* fill) {
  const int32_t o0 = v->offsets[v->offset + i], o1 = v->offsets[v->offset + i + 1];
  out_offsets[pos] = *fill;
  memcpy(out_data + *fill, v->data + o0, (size_t)(o1 - o0)); /* data_builder.UnsafeAppend :1550 */
  *fill += o1 - o0;
}
int orc_binary_concatenate(const orc_binary* chunks, int nchunks, int32_t* out_offsets, uint8_t* out_data,
                           uint8_t* out_validity, orc_out* out) {
  if (nchunks < 1) return fail(ORC_INVALID, "Must pass at least one array");
  int64_t rows = 0, nulls = 0, bytes = 0;
  for (int c = 0; c < nchunks; ++c) {
    rows += chunks[c].length;
    const int64_t nc = chunks[c].validity ? chunks[c].null_count : 0; /* :172-180 */
    nulls = (nulls < 0 || nc < 0) ? -1 : nulls + nc;
  }
  out->length = rows; out->null_count = nulls; out->has_validity = nulls != 0; /* :190 */
  memset(out_validity, 0, (size_t)bytes_for_bits(rows));
  int64_t pos = 0;
  for (int c = 0; c < nchunks; ++c) {
    const orc_binary* a = &chunks[c];
    const int32_t first = a->length ? a->offsets[a->offset] : 0;
    for (int64_t i = 0; i < a->length; ++i) {
      out_offsets[pos + i] = (int32_t)(a->offsets[a->offset + i] - first + bytes); /* PutOffsets :143-160 */
      if (out->has_validity && bin_valid(a, i)) set_bit(out_validity, pos + i);
    }
    if (a->length) {
      const int32_t last = a->offsets[a->offset + a->length];
      if (bytes + (last - first) > 2147483647LL) return fail(ORC_INVALID, "offset overflow while concatenating arrays");
      memcpy(out_data + bytes, a->data + first, (size_t)(last - first)); /* value_ranges :209-212 */
      bytes += last - first;
    }
    pos += a->length;
  }
  out_offsets[rows] = (int32_t)bytes;
  if (!out->has_validity) for (int64_t i = 0; i < rows; ++i) set_bit(out_validity, i);
  return ORC_OK;
}
int orc_binary_take(const orc_binary* values, const orc_array* indices, int boundscheck, int32_t* out_offsets,
                    uint8_t* out_data, uint8_t* out_validity, orc_out* out) {
  const int64_t n = indices->length;
  if (boundscheck) { /* CheckIndexBounds first (TakeExec :2080-2090; int_util.cc:534-561) */
    for (int64_t i = 0; i < n; ++i) {
      if (!slot_valid(indices, i)) continue;
      int neg; int64_t v = load_index(indices, i, &neg);
      const int oob = indices->type_id == ORC_UINT64 ? ((uint64_t)v >= (uint64_t)values->length) : (neg || v >= values->length);
      if (oob) {
        char msg[96];
        if (indices->type_id == ORC_UINT64) snprintf(msg, sizeof msg, "Index %llu out of bounds", (unsigned long long)v);
        else snprintf(msg, sizeof msg, "Index %lld out of bounds", (long long)v);
        return fail(ORC_INDEX_ERROR, msg);
      }
    }
  }
  memset(out_validity, 0, (size_t)bytes_for_bits(n));
  out->length = n; out->null_count = 0; out->has_validity = 1; /* FinishCommon :1250-1255 */
  int32_t fill = 0;
  const int values_have_nulls = values->validity && values->null_count != 0; /* MayHaveNulls :1265 */
  for (int64_t i = 0; i < n; ++i) {
    int neg = 0;
    int ok = slot_valid(indices, i);
    int64_t u = 0;
    if (ok) {
      u = load_index(indices, i, &neg);
      if (indices->type_id == ORC_UINT64 ? ((uint64_t)u >= (uint64_t)values->length) : (neg || u >= values->length)) ok = 0;
    }
    if (ok && values_have_nulls && !bin_valid(values, u)) ok = 0;
    if (ok) {
      if ((int64_t)fill + (values->offsets[values->offset + u + 1] - values->offsets[values->offset + u]) > 2147483646LL)
        return fail(ORC_INVALID, "Take operation overflowed binary array capacity"); /* :1540-1544 */
      bin_append(values, u, out_offsets, out_data, i, &fill);
      set_bit(out_validity, i);
    } else {
      out_offsets[i] = fill; /* visit_null :1556-1559 */
      ++out->null_count;
    }
  }
  out_offsets[n] = fill;
  return ORC_OK;
}
int orc_binary_filter(const orc_binary* values, const orc_array* filter, int null_selection, int32_t* out_offsets,
                      uint8_t* out_data, uint8_t* out_validity, orc_out* out) {
  if (filter->length != values->length) return fail(ORC_INVALID, "Array arguments must all be the same length");
  const int64_t vnc = values->validity ? values->null_count : 0, fnc = arg_null_count(filter);
  const int with_validity = vnc != 0 || fnc != 0; /* :1099-1112 */
  int64_t pos = 0;
  int32_t fill = 0;
  memset(out_validity, 0, (size_t)bytes_for_bits(values->length));
  for (int64_t i = 0; i < values->length; ++i) {
    const int fvalid = slot_valid(filter, i), fset = fvalid && bool_at(filter, i);
    if (fset) {
      if (bin_valid(values, i)) { bin_append(values, i, out_offsets, out_data, pos, &fill); set_bit(out_validity, pos); }
      else out_offsets[pos] = fill;
      ++pos;
    } else if (!fvalid && null_selection == ORC_EMIT_NULL) {
      out_offsets[pos] = fill;
      ++pos;
    }
  }
  out_offsets[pos] = fill;
  out->length = pos;
  out->has_validity = with_validity;
  out->null_count = (vnc == 0 && (null_selection == ORC_DROP || fnc == 0)) ? 0 : -1; /* :1090-1096 */
  if (!with_validity) for (int64_t i = 0; i < pos; ++i) set_bit(out_validity, i);
  return ORC_OK;
}

/* ---- b: sort_indices, vector_sort.cc:224-575, 657-832 ------------------------------ */
typedef struct { uint64_t key; uint64_t idx; } sort_rec;
static int cmp_rec(const void* a, const void* b) {
  const sort_rec* x = (const sort_rec*)a; const sort_rec* y = (const sort_rec*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}
/* Order-preserving map value -> uint64 such that (a < b) <=> key(a) < key(b) and values
 * that compare equal under `<` (incl. -0.0 vs +0.0) get the SAME key, so that a sort on
 * (key, index) reproduces std::stable_sort with `lhs < rhs` (:386-401). */
static uint64_t sort_key(int t, const void* base, int64_t k) {
  switch (t) {
    case ORC_BOOL: return get_bit((const uint8_t*)base, k);
    case ORC_UINT8: return ((const uint8_t*)base)[k];
    case ORC_UINT16: return ((const uint16_t*)base)[k];
    case ORC_UINT32: return ((const uint32_t*)base)[k];
    case ORC_UINT64: return ((const uint64_t*)base)[k];
    case ORC_INT8: return (uint64_t)((int64_t)((const int8_t*)base)[k]) ^ 0x8000000000000000ULL;
    case ORC_INT16: return (uint64_t)((int64_t)((const int16_t*)base)[k]) ^ 0x8000000000000000ULL;
    case ORC_INT32: return (uint64_t)((int64_t)((const int32_t*)base)[k]) ^ 0x8000000000000000ULL;
    case ORC_INT64: return (uint64_t)(((const int64_t*)base)[k]) ^ 0x8000000000000000ULL;
    case ORC_FLOAT: case ORC_DOUBLE: {
      double d = (t == ORC_FLOAT) ? (double)((const float*)base)[k] : ((const double*)base)[k];
      if (d == 0.0) d = 0.0; /* -0.0 ties with +0.0 under `<` */
      uint64_t u; memcpy(&u, &d, 8);
      return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    }
  }
  return 0;
}
int orc_sort_indices(const orc_array* chunks, int nchunks, int order, uint64_t* out) {
  int64_t total = 0;
  for (int c = 0; c < nchunks; ++c) total += chunks[c].length;
  if (nchunks > 0 && orc_bit_width(chunks[0].type_id) < 0) return fail(ORC_NOT_IMPLEMENTED, "sort: type");
  sort_rec* recs = (sort_rec*)malloc(sizeof(sort_rec) * (size_t)(total ? total : 1));
  uint64_t* nans = (uint64_t*)malloc(8 * (size_t)(total ? total : 1));
  uint64_t* nulls = (uint64_t*)malloc(8 * (size_t)(total ? total : 1));
  int64_t nr = 0, nn = 0, nz = 0, base = 0;
  for (int c = 0; c < nchunks; ++c) {
    const orc_array* a = &chunks[c];
    for (int64_t i = 0; i < a->length; ++i) {
      const uint64_t gi = (uint64_t)(base + i); /* logical index = concat order (:95-160) */
      if (!slot_valid(a, i)) { nulls[nz++] = gi; continue; } /* PartitionNullsOnly :224-234 */
      if (is_float(a->type_id)) {
        double d = a->type_id == ORC_FLOAT ? (double)((const float*)a->values)[a->offset + i]
                                           : ((const double*)a->values)[a->offset + i];
        if (isnan(d)) { nans[nn++] = gi; continue; } /* PartitionNullLikes :275-283 */
      }
      uint64_t k = sort_key(a->type_id, a->values, a->offset + i);
      /* Descending = complement the key; ties still resolve by ascending index (:393-401,
       * counting sort :466-475, golden vector_sort_test.cc:455-458) */
      recs[nr].key = (order == ORC_DESCENDING) ? ~k : k;
      recs[nr].idx = gi; ++nr;
    }
    base += a->length;
  }
  qsort(recs, (size_t)nr, sizeof(sort_rec), cmp_rec);
  int64_t p = 0;
  for (int64_t i = 0; i < nr; ++i) out[p++] = recs[i].idx;
  for (int64_t i = 0; i < nn; ++i) out[p++] = nans[i];  /* NaNs, index order, before nulls */
  for (int64_t i = 0; i < nz; ++i) out[p++] = nulls[i]; /* nulls last, index order */
  free(recs); free(nans); free(nulls);
  return ORC_OK;
}

/* ---- 8f.4: vector_hash.cc -- unique / value_counts / dictionary_encode ----------------- */
/* The memo table (util/hashing.h ScalarMemoTable / SmallScalarMemoTable): memo index = order of
 * first appearance; entries match on equal bits; a null is its own entry.  The grouper above keeps
 * exactly such a table over [null byte][value bytes] keys, so it serves as the memo table here. */
static int memo_run(const orc_array* a, uint32_t** ids, orc_grouper** g) {
  if (orc_bit_width(a->type_id) < 0) return fail(ORC_NOT_IMPLEMENTED, "hash kernels: type");
  *g = orc_grouper_new(1, &a->type_id);
  *ids = (uint32_t*)malloc(4 * (size_t)(a->length + 1));
  return orc_grouper_consume(*g, a, *ids); /* RegularHashKernel::DoAppend :300-330, row by row */
}
static void memo_dictionary(const orc_grouper* g, void* out_values, uint8_t* out_validity, orc_out* out) {
  int64_t nulls = 0;
  orc_grouper_uniques(g, 0, out_values, out_validity, &nulls); /* GetDictionary :345-351 + DictionaryTraits */
  out->length = orc_grouper_num_groups(g);
  out->null_count = nulls;
  out->has_validity = nulls > 0; /* the bitmap exists only when the table holds a null (dict_internal.h) */
}
int orc_unique(const orc_array* a, void* out_values, uint8_t* out_validity, orc_out* out) {
  uint32_t* ids; orc_grouper* g;
  int rc = memo_run(a, &ids, &g);
  if (rc == ORC_OK) memo_dictionary(g, out_values, out_validity, out); /* UniqueFinalize :617-623 */
  free(ids); orc_grouper_free(g);
  return rc;
}
int orc_value_counts(const orc_array* a, void* out_values, uint8_t* out_validity, int64_t* out_counts, orc_out* out) {
  uint32_t* ids; orc_grouper* g;
  int rc = memo_run(a, &ids, &g);
  if (rc == ORC_OK) {
    memo_dictionary(g, out_values, out_validity, out);
    memset(out_counts, 0, 8 * (size_t)out->length);
    for (int64_t i = 0; i < a->length; ++i) ++out_counts[ids[i]]; /* ObserveFound / ObserveNullFound :117-146 */
  }
  free(ids); orc_grouper_free(g);
  return rc;
}
int orc_dictionary_encode(const orc_array* a, int null_encoding, int32_t* out_indices, uint8_t* out_index_validity,
                          orc_out* om, void* out_dict, uint8_t* out_dict_validity, orc_out* dm) {
  om->length = a->length; om->null_count = 0; om->has_validity = 0;
  if (null_encoding == 0) { /* ENCODE: nulls go through the memo table like values (ShouldEncodeNulls :210-212) */
    uint32_t* ids; orc_grouper* g;
    int rc = memo_run(a, &ids, &g);
    if (rc == ORC_OK) {
      memo_dictionary(g, out_dict, out_dict_validity, dm);
      for (int64_t i = 0; i < a->length; ++i) out_indices[i] = (int32_t)ids[i];
      memset(out_index_validity, 0xFF, (size_t)bytes_for_bits(a->length));
    }
    free(ids); orc_grouper_free(g);
    return rc;
  }
  /* MASK: a null row appends a null index and never reaches the table (DoAppend with
   * !ShouldEncodeNulls, :318-326; ObserveNullFound :187-194): the dictionary is the table of the
   * valid rows alone. */
  if (orc_bit_width(a->type_id) < 0) return fail(ORC_NOT_IMPLEMENTED, "hash kernels: type");
  orc_grouper* g = orc_grouper_new(1, &a->type_id);
  memset(out_index_validity, 0, (size_t)bytes_for_bits(a->length));
  for (int64_t i = 0; i < a->length; ++i) {
    if (!slot_valid(a, i)) { out_indices[i] = 0; ++om->null_count; continue; }
    orc_array one = *a;
    one.offset = a->offset + i; one.length = 1;
    uint32_t id;
    orc_grouper_consume(g, &one, &id);
    out_indices[i] = (int32_t)id;
    set_bit(out_index_validity, i);
  }
  om->has_validity = om->null_count > 0;
  memo_dictionary(g, out_dict, out_dict_validity, dm);
  orc_grouper_free(g);
  return ORC_OK;
}

/* ---- c4: GroupedMinMaxImpl, hash_aggregate.cc:627-768 -------------------------------- */
#define MINMAX_CASE(ID, T, LO, HI)                                                          \
  case ID: {                                                                                \
    T* mn = (T*)out_mins; T* mx = (T*)out_maxes; const T* v = (const T*)values->values;     \
    for (int64_t g = 0; g < num_groups; ++g) { mn[g] = HI; mx[g] = LO; } /* anti-extremes :653-661 */ \
    for (int64_t i = 0; i < values->length; ++i) {                                          \
      const uint32_t g = ids[i];                                                            \
      if (!slot_valid(values, i)) { set_bit(has_nulls, g); continue; }                      \
      const T x = v[values->offset + i];                                                    \
      mx[g] = (mx[g] < x) ? x : mx[g]; /* std::max(a, b) = (a < b) ? b : a  (:667) */       \
      mn[g] = (x < mn[g]) ? x : mn[g]; /* std::min(a, b) = (b < a) ? b : a  (:668) */       \
      set_bit(has_values, g);                                                               \
    }                                                                                       \
  } break;
int orc_hash_min_max(const orc_array* values, const uint32_t* ids, int64_t num_groups, int null_handling,
                     void* out_mins, void* out_maxes, uint8_t* out_validity, orc_out* out) {
  const size_t nb = (size_t)((num_groups + 7) / 8) + 1;
  uint8_t* has_values = (uint8_t*)calloc(nb, 1);
  uint8_t* has_nulls = (uint8_t*)calloc(nb, 1);
  switch (values->type_id) {
    MINMAX_CASE(ORC_UINT8, uint8_t, 0, UINT8_MAX)
    MINMAX_CASE(ORC_INT8, int8_t, INT8_MIN, INT8_MAX)
    MINMAX_CASE(ORC_UINT16, uint16_t, 0, UINT16_MAX)
    MINMAX_CASE(ORC_INT16, int16_t, INT16_MIN, INT16_MAX)
    MINMAX_CASE(ORC_UINT32, uint32_t, 0, UINT32_MAX)
    MINMAX_CASE(ORC_INT32, int32_t, INT32_MIN, INT32_MAX)
    MINMAX_CASE(ORC_UINT64, uint64_t, 0, UINT64_MAX)
    MINMAX_CASE(ORC_INT64, int64_t, INT64_MIN, INT64_MAX)
    MINMAX_CASE(ORC_FLOAT, float, -INFINITY, INFINITY)
    MINMAX_CASE(ORC_DOUBLE, double, -INFINITY, INFINITY)
    default:
      free(has_values); free(has_nulls);
      return fail(ORC_NOT_IMPLEMENTED, "Grouped MinMax data of this type"); /* :674-684 */
  }
  out->length = num_groups; out->null_count = 0; out->has_validity = 1;
  memset(out_validity, 0, (size_t)((num_groups + 7) / 8));
  for (int64_t g = 0; g < num_groups; ++g) { /* Finalize :729-740 */
    const int ok = get_bit(has_values, g) && !(null_handling == 1 && get_bit(has_nulls, g));
    if (ok) set_bit(out_validity, g); else ++out->null_count;
  }
  free(has_values); free(has_nulls);
  return ORC_OK;
}

/* ---- b5: multi-key sort_indices, vector_sort.cc:955-1549 ---------------------------- */
/* One resolved sort key (ResolvedSortKey :1187-1205 / :1384-1420): its chunks, the logical
 * row at which each starts (ChunkedArrayResolver :95-160) and its order. */
typedef struct { const orc_array* chunks; int nchunks; int order; int64_t* starts; } mk_key;
static const mk_key* g_mk_keys; static int g_mk_nkeys;
static const orc_array* mk_resolve(const mk_key* k, int64_t row, int64_t* in_chunk) {
  int lo = 0, hi = k->nchunks - 1;
  while (lo < hi) { /* last chunk starting at or before row (empty chunks share a start with their successor) */
    int mid = (lo + hi + 1) / 2;
    if (k->starts[mid] <= row) lo = mid; else hi = mid - 1;
  }
  *in_chunk = row - k->starts[lo];
  return &k->chunks[lo];
}
/* MultipleKeyComparator::CompareType / CompareTypeValue (:1105-1172): -1, 0, 1.  Null after
 * everything, NaN after every number, both whatever the order; equal values (incl. -0.0 vs
 * +0.0, NaN vs NaN, null vs null) compare 0 and the next key decides. */
static int mk_compare_key(const mk_key* k, uint64_t l, uint64_t r) {
  int64_t il, ir;
  const orc_array* al = mk_resolve(k, (int64_t)l, &il);
  const orc_array* ar = mk_resolve(k, (int64_t)r, &ir);
  const int nl = !slot_valid(al, il), nr = !slot_valid(ar, ir);
  if (nl && nr) return 0;
  if (nl) return 1;
  if (nr) return -1;
  int c;
  const int t = al->type_id;
  if (is_float(t)) {
    double x = t == ORC_FLOAT ? (double)((const float*)al->values)[al->offset + il] : ((const double*)al->values)[al->offset + il];
    double y = t == ORC_FLOAT ? (double)((const float*)ar->values)[ar->offset + ir] : ((const double*)ar->values)[ar->offset + ir];
    const int xn = isnan(x), yn = isnan(y);
    if (xn && yn) return 0;
    if (xn) return 1;
    if (yn) return -1;
    c = x == y ? 0 : (x > y ? 1 : -1);
  } else {
    /* sort_key is order preserving and maps equal values to equal keys */
    const uint64_t x = sort_key(t, al->values, al->offset + il), y = sort_key(t, ar->values, ar->offset + ir);
    c = x == y ? 0 : (x > y ? 1 : -1);
  }
  return k->order == ORC_DESCENDING ? -c : c;
}
static int mk_less(uint64_t l, uint64_t r) { /* MultipleKeyComparator::Compare :1062-1077 */
  for (int k = 0; k < g_mk_nkeys; ++k) {
    const int c = mk_compare_key(&g_mk_keys[k], l, r);
    if (c != 0) return c < 0;
  }
  return 0;
}
/* std::stable_sort over the iota-initialised indices (:1640-1652, :1686-1704) */
static void mk_merge_sort(uint64_t* a, uint64_t* tmp, int64_t n) {
  if (n < 2) return;
  const int64_t h = n / 2;
  mk_merge_sort(a, tmp, h);
  mk_merge_sort(a + h, tmp, n - h);
  int64_t i = 0, j = h, o = 0;
  while (i < h && j < n) tmp[o++] = mk_less(a[j], a[i]) ? a[j++] : a[i++]; /* right wins only if strictly less */
  while (i < h) tmp[o++] = a[i++];
  while (j < n) tmp[o++] = a[j++];
  memcpy(a, tmp, 8 * (size_t)n);
}
int orc_sort_indices_multi(const orc_array* chunks, const int* nchunks, const int* orders, int nkeys, uint64_t* out) {
  if (nkeys < 1) return fail(ORC_INVALID, "Must specify one or more sort keys"); /* :1634-1636 */
  if (nkeys > 64) return fail(ORC_INVALID, "too many sort keys");
  mk_key keys[64];
  int64_t n = -1;
  const orc_array* p = chunks;
  for (int k = 0; k < nkeys; ++k) {
    keys[k].chunks = p; keys[k].nchunks = nchunks[k]; keys[k].order = orders[k];
    keys[k].starts = (int64_t*)malloc(8 * (size_t)(nchunks[k] + 1));
    int64_t len = 0;
    for (int c = 0; c < nchunks[k]; ++c) { keys[k].starts[c] = len; len += p[c].length; }
    if (n >= 0 && len != n) { for (int q = 0; q <= k; ++q) free(keys[q].starts); return fail(ORC_INVALID, "sort keys differ in length"); }
    n = len;
    p += nchunks[k];
  }
  uint64_t* tmp = (uint64_t*)malloc(8 * (size_t)(n ? n : 1));
  for (int64_t i = 0; i < n; ++i) out[i] = (uint64_t)i;
  g_mk_keys = keys; g_mk_nkeys = nkeys;
  mk_merge_sort(out, tmp, n);
  free(tmp);
  for (int k = 0; k < nkeys; ++k) free(keys[k].starts);
  return ORC_OK;
}

/* ---- Concatenate, array/concatenate.cc:70-92, 171-201 -------------------------------- */
int orc_concatenate(const orc_array* chunks, int nchunks, void* out_values, uint8_t* out_validity, orc_out* out) {
  if (nchunks < 1) return fail(ORC_INVALID, "Must pass at least one array"); /* :428-430 */
  const int t = chunks[0].type_id, bw = orc_bit_width(t);
  int64_t total = 0, nulls = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (chunks[c].type_id != t) return fail(ORC_INVALID, "arrays to be concatenated must be identically typed"); /* :436-441 */
    total += chunks[c].length;
    const int64_t nc = arg_null_count(&chunks[c]); /* :172-180 */
    nulls = (nulls < 0 || nc < 0) ? -1 : nulls + nc;
  }
  out->length = total; out->null_count = nulls; out->has_validity = nulls != 0; /* :190 */
  int64_t pos = 0;
  if (bw == 1) memset(out_values, 0, (size_t)((total + 7) / 8));
  if (out->has_validity) memset(out_validity, 0, (size_t)((total + 7) / 8));
  for (int c = 0; c < nchunks; ++c) {
    const orc_array* a = &chunks[c];
    for (int64_t i = 0; i < a->length; ++i) {
      /* ConcatenateBitmaps :70-92: an absent bitmap contributes set bits */
      if (out->has_validity && slot_valid(a, i)) set_bit(out_validity, pos + i);
      if (bw == 1 && get_bit((const uint8_t*)a->values, a->offset + i)) set_bit((uint8_t*)out_values, pos + i);
    }
    if (bw != 1 && a->length > 0)
      memcpy((char*)out_values + pos * (bw / 8), (const char*)a->values + a->offset * (bw / 8), (size_t)(a->length * (bw / 8)));
    pos += a->length;
  }
  return ORC_OK;
}

/* ---- c1: GrouperImpl, hash_aggregate.cc:313-437 ------------------------------------- */
struct orc_grouper {
  int nkeys; int types[8]; int widths[8]; /* value byte width per key column */
  int key_len;                             /* sum(1 + width) */
  uint8_t* key_bytes; int64_t key_cap;     /* key_bytes_ (:434) */
  uint32_t num_groups;                     /* num_groups_ (:435) */
  uint32_t* table; uint64_t table_size;    /* open addressing; stands in for map_ (:432) */
};
static uint64_t hash_bytes(const uint8_t* p, int n) {
  uint64_t h = 1469598103934665603ULL;
  for (int i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ULL; }
  return h ^ (h >> 29);
}
orc_grouper* orc_grouper_new(int nkeys, const int* types) {
  if (nkeys < 1 || nkeys > 8) return NULL;
  orc_grouper* g = (orc_grouper*)calloc(1, sizeof *g);
  g->nkeys = nkeys;
  for (int k = 0; k < nkeys; ++k) {
    int bw = orc_bit_width(types[k]);
    if (bw < 0) { free(g); return NULL; }
    g->types[k] = types[k];
    g->widths[k] = bw == 1 ? 1 : bw / 8; /* BooleanKeyEncoder kByteWidth=1 (:96), FixedWidth :145 */
    g->key_len += 1 + g->widths[k];      /* kExtraByteForNull (:50) */
  }
  g->table_size = 1024;
  g->table = (uint32_t*)malloc(4 * g->table_size);
  memset(g->table, 0xFF, 4 * g->table_size);
  g->key_cap = 1024 * g->key_len;
  g->key_bytes = (uint8_t*)malloc((size_t)g->key_cap);
  return g;
}
void orc_grouper_free(orc_grouper* g) { if (g) { free(g->table); free(g->key_bytes); free(g); } }
uint32_t orc_grouper_num_groups(const orc_grouper* g) { return g->num_groups; }
static void grouper_rehash(orc_grouper* g) {
  uint64_t ns = g->table_size * 2;
  uint32_t* nt = (uint32_t*)malloc(4 * ns);
  memset(nt, 0xFF, 4 * ns);
  for (uint32_t id = 0; id < g->num_groups; ++id) {
    uint64_t h = hash_bytes(g->key_bytes + (int64_t)id * g->key_len, g->key_len) & (ns - 1);
    while (nt[h] != 0xFFFFFFFFu) h = (h + 1) & (ns - 1);
    nt[h] = id;
  }
  free(g->table); g->table = nt; g->table_size = ns;
}
int orc_grouper_consume(orc_grouper* g, const orc_array* keys, uint32_t* out_ids) {
  const int64_t n = keys[0].length;
  uint8_t enc[8 * 9];
  for (int64_t i = 0; i < n; ++i) {
    /* Encode :142-168 / :98-113: [null byte][value bytes, zeros when null] per column */
    uint8_t* p = enc;
    for (int k = 0; k < g->nkeys; ++k) {
      const orc_array* a = &keys[k];
      const int w = g->widths[k];
      if (slot_valid(a, i)) {
        *p++ = 0; /* kValidByte */
        if (g->types[k] == ORC_BOOL) *p++ = (uint8_t)bool_at(a, i);
        else { memcpy(p, (const uint8_t*)a->values + (a->offset + i) * w, w); p += w; }
      } else {
        *p++ = 1; /* kNullByte */
        memset(p, 0, w); p += w;
      }
    }
    /* map_.emplace(key, num_groups_) :391-401: new key <=> next dense id, in row order */
    uint64_t h = hash_bytes(enc, g->key_len) & (g->table_size - 1);
    uint32_t id;
    for (;;) {
      id = g->table[h];
      if (id == 0xFFFFFFFFu) break;
      if (memcmp(g->key_bytes + (int64_t)id * g->key_len, enc, g->key_len) == 0) break;
      h = (h + 1) & (g->table_size - 1);
    }
    if (id == 0xFFFFFFFFu) {
      id = g->num_groups++;
      if ((int64_t)(id + 1) * g->key_len > g->key_cap) {
        g->key_cap *= 2; g->key_bytes = (uint8_t*)realloc(g->key_bytes, (size_t)g->key_cap);
      }
      memcpy(g->key_bytes + (int64_t)id * g->key_len, enc, g->key_len);
      g->table[h] = id;
      if ((uint64_t)g->num_groups * 2 > g->table_size) grouper_rehash(g);
    }
    out_ids[i] = id;
  }
  return ORC_OK;
}
int orc_grouper_uniques(const orc_grouper* g, int k, void* out_values, uint8_t* out_validity,
                        int64_t* out_null_count) {
  int off = 0;
  for (int j = 0; j < k; ++j) off += 1 + g->widths[j];
  const int w = g->widths[k];
  int64_t nulls = 0;
  const int64_t n = g->num_groups;
  memset(out_validity, 0, bytes_for_bits(n));
  if (g->types[k] == ORC_BOOL) memset(out_values, 0, bytes_for_bits(n));
  for (int64_t i = 0; i < n; ++i) {
    const uint8_t* e = g->key_bytes + i * g->key_len + off;
    if (e[0] == 1) ++nulls; else set_bit(out_validity, i); /* DecodeNulls :64-92 */
    if (g->types[k] == ORC_BOOL) set_bit_to((uint8_t*)out_values, i, e[1] != 0); /* :127-131 */
    else memcpy((uint8_t*)out_values + i * w, e + 1, w);                        /* :178-184 */
  }
  *out_null_count = nulls;
  return ORC_OK;
}

/* ---- c2/c3: GroupedSumImpl / GroupedCountImpl, hash_aggregate.cc:466-622 ------------ */
int orc_hash_sum(const orc_array* v, const uint32_t* ids, void* sums, int64_t* counts) {
  const int t = v->type_id;
  if (orc_bit_width(t) < 0) return fail(ORC_NOT_IMPLEMENTED, "Summing data of this type");
  for (int64_t i = 0; i < v->length; ++i) {
    if (!slot_valid(v, i)) continue; /* [&] { ++group; } :545 */
    const uint32_t g = ids[i];
    const int64_t k = v->offset + i;
    /* sums[*group] += value; counts[*group] += 1 (:539-543); accumulator per
     * FindAccumulatorType (aggregate_internal.h:29-50); integer adds wrap mod 2^64 */
    switch (t) {
      case ORC_BOOL: ((uint64_t*)sums)[g] += (uint64_t)get_bit((const uint8_t*)v->values, k); break;
      case ORC_UINT8: ((uint64_t*)sums)[g] += ((const uint8_t*)v->values)[k]; break;
      case ORC_UINT16: ((uint64_t*)sums)[g] += ((const uint16_t*)v->values)[k]; break;
      case ORC_UINT32: ((uint64_t*)sums)[g] += ((const uint32_t*)v->values)[k]; break;
      case ORC_UINT64: ((uint64_t*)sums)[g] += ((const uint64_t*)v->values)[k]; break;
      case ORC_INT8: ((uint64_t*)sums)[g] += (uint64_t)(int64_t)((const int8_t*)v->values)[k]; break;
      case ORC_INT16: ((uint64_t*)sums)[g] += (uint64_t)(int64_t)((const int16_t*)v->values)[k]; break;
      case ORC_INT32: ((uint64_t*)sums)[g] += (uint64_t)(int64_t)((const int32_t*)v->values)[k]; break;
      case ORC_INT64: ((uint64_t*)sums)[g] += (uint64_t)((const int64_t*)v->values)[k]; break;
      case ORC_FLOAT: ((double*)sums)[g] += (double)((const float*)v->values)[k]; break;
      case ORC_DOUBLE: ((double*)sums)[g] += ((const double*)v->values)[k]; break;
    }
    counts[g] += 1;
  }
  return ORC_OK;
}
/* NOTE (reference defect, not replicated): for COUNT_NON_NULL the reference indexes
 * group_ids with `begin - input->offset` (:497-499) although VisitSetBitRunsVoid already
 * reports run positions relative to the offset (util/bit_run_reader.h:484-499), so on a
 * SLICED input (offset != 0) it reads ids out of bounds.  The restatement uses the
 * logical row index; parity for hash_count is pinned on offset == 0 inputs only. */
int orc_hash_count(const orc_array* v, const uint32_t* ids, int mode, int64_t* counts) {
  for (int64_t i = 0; i < v->length; ++i) {
    const int valid = slot_valid(v, i);
    if (mode == ORC_COUNT_NULL) counts[ids[i]] += !valid; /* :485-492 */
    else if (valid) counts[ids[i]] += 1;                  /* :495-503 */
  }
  return ORC_OK;
}
int orc_hash_mean_finalize(int t, const void* sums, const int64_t* counts, int64_t n,
                           double* out, uint8_t* out_validity) {
  memset(out_validity, 0, bytes_for_bits(n));
  for (int64_t g = 0; g < n; ++g) {
    if (counts[g] <= 0) { out[g] = 0.0; continue; }
    double s = is_float(t) ? ((const double*)sums)[g]
             : is_signed_int(t) ? (double)((const int64_t*)sums)[g] : (double)((const uint64_t*)sums)[g];
    out[g] = s / (double)counts[g]; /* aggregate_basic_internal.h:94-103 */
    set_bit(out_validity, g);
  }
  return ORC_OK;
}
int orc_hash_aggregate(int func, const orc_array* v, const uint32_t* ids, int64_t ng, int mode,
                       void* out_values, uint8_t* out_validity, orc_out* out) {
  out->length = ng;
  memset(out_values, 0, (size_t)ng * 8); /* sums_.Append(added*8, 0) :568-569 */
  memset(out_validity, 0xFF, bytes_for_bits(ng));
  if (ng & 7) out_validity[bytes_for_bits(ng) - 1] = (uint8_t)((1u << (ng & 7)) - 1);
  if (func == 1) {
    out->has_validity = 0; out->null_count = 0; /* Int64Array(num_groups_, counts) :505-508 */
    return orc_hash_count(v, ids, mode, (int64_t*)out_values);
  }
  int64_t* counts = (int64_t*)calloc((size_t)(ng ? ng : 1), 8);
  int rc = orc_hash_sum(v, ids, out_values, counts);
  if (rc == ORC_OK) {
    /* Finalize :581-604: bitmap allocated only when some group saw no valid value */
    int64_t nulls = 0;
    for (int64_t g = 0; g < ng; ++g) if (counts[g] <= 0) { ++nulls; clear_bit(out_validity, g); }
    out->null_count = nulls; out->has_validity = nulls > 0;
    if (func == 2) {
      double* m = (double*)malloc((size_t)(ng ? ng : 1) * 8);
      orc_hash_mean_finalize(v->type_id, out_values, counts, ng, m, out_validity);
      memcpy(out_values, m, (size_t)ng * 8); free(m);
    }
  }
  free(counts);
  return rc;
}

/* ---- f1: NullPropagator with a preallocated bitmap, exec.cc:255-461 ------------------ */
static int generalization(const orc_array* a) { /* NullGeneralization::Get :258-281 */
  if (!a) return 1;                              /* valid scalar: ALL_VALID */
  if (a->null_count == 0 || a->validity == NULL) return 1;
  if (a->null_count == a->length) return 2;      /* ALL_NULL */
  return 0;                                      /* PERHAPS_NULL */
}
static void propagate(const orc_array* a, const orc_array* b, int64_t len, uint8_t* out_validity,
                      orc_out* out) {
  const orc_array* with_nulls[2]; int nw = 0; int all_null = 0;
  const orc_array* in[2] = {a, b};
  for (int k = 0; k < 2; ++k) {
    int g = generalization(in[k]);
    if (g == 2) all_null = 1;
    if (g != 1 && in[k]) with_nulls[nw++] = in[k];
  }
  out->length = len; out->has_validity = 1; /* validity_preallocated_ :690-693 */
  memset(out_validity, 0, bytes_for_bits(len));
  /* a zero-length call runs no batch: the (empty) result keeps kUnknownNullCount (observed on
   * the reference build for every INTERSECTION scalar kernel; exec.cc:704-716, :760-768) */
  if (len == 0) { out->null_count = -1; return; }
  if (all_null) { out->null_count = len; return; }                     /* :326-348 */
  if (nw == 0) {                                                       /* :433-439 */
    out->null_count = 0;
    for (int64_t i = 0; i < len; ++i) set_bit(out_validity, i);
    return;
  }
  if (nw == 1) {                                                       /* PropagateSingle :350-380 */
    out->null_count = with_nulls[0]->null_count;
    for (int64_t i = 0; i < len; ++i) if (slot_valid(with_nulls[0], i)) set_bit(out_validity, i);
    return;
  }
  out->null_count = -1;                                                /* PropagateMultiple :382-408 */
  for (int64_t i = 0; i < len; ++i)
    if (slot_valid(with_nulls[0], i) && slot_valid(with_nulls[1], i)) set_bit(out_validity, i);
}
int orc_propagate_nulls(const orc_array* a, const orc_array* b, uint8_t* out_validity, orc_out* out) {
  propagate(a, b, a ? a->length : b->length, out_validity, out);
  return ORC_OK;
}

/* ---- d: compare, scalar_compare.cc:31-57 -------------------------------------------- */
#define LOAD_AS(T, arr, scal, i) ((arr) ? ((const T*)(arr)->values)[(arr)->offset + (i)] : *(const T*)(scal))
#define CMP_LOOP(T)                                                                       \
  for (int64_t i = 0; i < n; ++i) {                                                       \
    const T l = LOAD_AS(T, lhs, ls, i), r = LOAD_AS(T, rhs, rs, i);                       \
    int v = 0;                                                                            \
    switch (op) {                                                                         \
      case ORC_EQ: v = l == r; break; case ORC_NE: v = l != r; break;                     \
      case ORC_GT: v = l > r; break;  case ORC_GE: v = l >= r; break;                     \
      case ORC_LT: v = r > l; break;  /* less = flipped greater, :161-171 */               \
      case ORC_LE: v = r >= l; break;                                                     \
    }                                                                                     \
    if (v) set_bit(out_bits, i);                                                          \
  }
/* ---- e3: implicit promotion and the casts behind it ---------------------------------------- */
/* CommonNumeric, codegen_internal.cc:220-268 (both types numeric, no half float; else -1). */
static int int_bits(int t) {
  switch (t) {
    case ORC_UINT8: case ORC_INT8: return 8;
    case ORC_UINT16: case ORC_INT16: return 16;
    case ORC_UINT32: case ORC_INT32: return 32;
    default: return 64;
  }
}
int orc_common_numeric(int a, int b) {
  const int ts[2] = {a, b};
  for (int i = 0; i < 2; ++i)
    if (!is_float(ts[i]) && !is_signed_int(ts[i]) && !is_unsigned_int(ts[i])) return -1;
  if (a == ORC_DOUBLE || b == ORC_DOUBLE) return ORC_DOUBLE;
  if (a == ORC_FLOAT || b == ORC_FLOAT) return ORC_FLOAT;
  int ws = 0, wu = 0;
  for (int i = 0; i < 2; ++i) {
    int* w = is_signed_int(ts[i]) ? &ws : &wu;
    if (int_bits(ts[i]) > *w) *w = int_bits(ts[i]);
  }
  if (ws == 0) { /* :247-253 */
    if (wu >= 64) return ORC_UINT64;
    if (wu == 32) return ORC_UINT32;
    if (wu == 16) return ORC_UINT16;
    return ORC_INT8; /* sic, :252 */
  }
  if (ws <= wu) { /* NextPower2(wu + 1), :255-257 */
    ws = 1;
    while (ws < wu + 1) ws <<= 1;
  }
  if (ws >= 64) return ORC_INT64;
  if (ws == 32) return ORC_INT32;
  if (ws == 16) return ORC_INT16;
  return ORC_INT8;
}

/* Number -> number casts, scalar_cast_numeric.cc: CastIntegerToInteger :38-45 (IntegersCanFit,
 * util/int_util.cc:914-948 with the bounds of GetSafeMinMax :826-912), CastFloatingToFloating
 * :47-50, CastIntegerToFloating :251-262 (CheckForIntegerToFloatingTruncation :211-249); the
 * conversion itself is static_cast (CastNumberToNumberUnsafe, scalar_cast_internal.cc).  The check
 * runs over the whole input first and names the first valid offending slot (int_util.cc:600-692). */
typedef struct { __int128 lo, hi; } range128;
static range128 int_range(int t) {
  range128 r = {0, 0};
  switch (t) {
    case ORC_UINT8: r.hi = UINT8_MAX; break;
    case ORC_INT8: r.lo = INT8_MIN; r.hi = INT8_MAX; break;
    case ORC_UINT16: r.hi = UINT16_MAX; break;
    case ORC_INT16: r.lo = INT16_MIN; r.hi = INT16_MAX; break;
    case ORC_UINT32: r.hi = UINT32_MAX; break;
    case ORC_INT32: r.lo = INT32_MIN; r.hi = INT32_MAX; break;
    case ORC_UINT64: r.hi = UINT64_MAX; break;
    default: r.lo = INT64_MIN; r.hi = INT64_MAX; break;
  }
  return r;
}
static __int128 load_int128(const orc_array* a, int64_t i) {
  const int64_t k = a->offset + i;
  switch (a->type_id) {
    case ORC_UINT8: return ((const uint8_t*)a->values)[k];
    case ORC_INT8: return ((const int8_t*)a->values)[k];
    case ORC_UINT16: return ((const uint16_t*)a->values)[k];
    case ORC_INT16: return ((const int16_t*)a->values)[k];
    case ORC_UINT32: return ((const uint32_t*)a->values)[k];
    case ORC_INT32: return ((const int32_t*)a->values)[k];
    case ORC_UINT64: return ((const uint64_t*)a->values)[k];
    default: return ((const int64_t*)a->values)[k];
  }
}
static void print_int128(char* buf, __int128 v) { /* std::to_string of the source type */
  if (v < 0) sprintf(buf, "%lld", (long long)v);
  else sprintf(buf, "%llu", (unsigned long long)v);
}
static void store_number(void* out, int64_t i, int to, __int128 iv, double fv, int from_float, int src_unsigned64) {
  switch (to) {
    case ORC_UINT8: ((uint8_t*)out)[i] = (uint8_t)(uint64_t)iv; break;
    case ORC_INT8: ((int8_t*)out)[i] = (int8_t)(uint8_t)(uint64_t)iv; break;
    case ORC_UINT16: ((uint16_t*)out)[i] = (uint16_t)(uint64_t)iv; break;
    case ORC_INT16: ((int16_t*)out)[i] = (int16_t)(uint16_t)(uint64_t)iv; break;
    case ORC_UINT32: ((uint32_t*)out)[i] = (uint32_t)(uint64_t)iv; break;
    case ORC_INT32: ((int32_t*)out)[i] = (int32_t)(uint32_t)(uint64_t)iv; break;
    case ORC_UINT64: ((uint64_t*)out)[i] = (uint64_t)iv; break;
    case ORC_INT64: ((int64_t*)out)[i] = (int64_t)(uint64_t)iv; break;
    case ORC_FLOAT: /* one rounding, from the source's own 64-bit widening */
      ((float*)out)[i] = from_float ? (float)fv : src_unsigned64 ? (float)(uint64_t)iv : (float)(int64_t)iv; break;
    default:
      ((double*)out)[i] = from_float ? fv : src_unsigned64 ? (double)(uint64_t)iv : (double)(int64_t)iv; break;
  }
}
int orc_cast_numeric(const orc_array* a, int to, int safe, void* out_values) {
  const int from = a->type_id;
  const int from_int = is_signed_int(from) || is_unsigned_int(from);
  const int to_int = is_signed_int(to) || is_unsigned_int(to);
  if (!from_int && !is_float(from)) return fail(ORC_NOT_IMPLEMENTED, "cast: source type");
  if (!to_int && !is_float(to)) return fail(ORC_NOT_IMPLEMENTED, "cast: target type");
  if (is_float(from) && to_int) return fail(ORC_NOT_IMPLEMENTED, "cast: floating point to integer");
  if (from_int && safe) {
    range128 src = int_range(from), dst;
    if (to_int) dst = int_range(to);
    else { /* FloatingIntegerBound :188-199; small integers are always exact :213-230 */
      const int exact_bits = to == ORC_FLOAT ? 24 : 53;
      dst.hi = (__int128)1 << exact_bits; dst.lo = -dst.hi;
      if (int_bits(from) < (to == ORC_FLOAT ? 32 : 64)) dst = src;
    }
    range128 b = src;
    if (dst.lo > b.lo) b.lo = dst.lo;
    if (dst.hi < b.hi) b.hi = dst.hi;
    if (b.lo != src.lo || b.hi != src.hi) {
      for (int64_t i = 0; i < a->length; ++i) {
        const __int128 v = load_int128(a, i);
        if ((v < b.lo || v > b.hi) && slot_valid(a, i)) {
          char msg[224], sv[48], sl[48], sh[48];
          print_int128(sv, v); print_int128(sl, b.lo); print_int128(sh, b.hi);
          sprintf(msg, "Integer value %s not in range: %s to %s", sv, sl, sh);
          return fail(ORC_INVALID, msg);
        }
      }
    }
  }
  for (int64_t i = 0; i < a->length; ++i) {
    if (from_int) store_number(out_values, i, to, load_int128(a, i), 0, 0, from == ORC_UINT64);
    else {
      const double fv = from == ORC_FLOAT ? (double)((const float*)a->values)[a->offset + i] : ((const double*)a->values)[a->offset + i];
      store_number(out_values, i, to, 0, fv, 1, 0);
    }
  }
  return ORC_OK;
}

/* Function::Execute's implicit casts for a binary call on two arrays of different numeric types
 * (function.cc:175-176 -> cast.cc:228-243 with CastOptions::Safe, after CompareFunction /
 * ArithmeticFunction::DispatchBest, scalar_compare.cc:75-98 / scalar_arithmetic.cc:389-412):
 * both operands become arrays of the common type sharing their validity.  0 = nothing to do. */
static int promote_pair(const orc_array* lhs, const orc_array* rhs, orc_array* pl, orc_array* pr, void** bl, void** br, int* rc) {
  *bl = *br = NULL; *rc = ORC_OK;
  if (!lhs || !rhs || lhs->type_id == rhs->type_id) return 0;
  const int t = orc_common_numeric(lhs->type_id, rhs->type_id);
  if (t < 0) { *rc = fail(ORC_NOT_IMPLEMENTED, "no kernel matching input types"); return 1; }
  const orc_array* in[2] = {lhs, rhs};
  orc_array* outp[2] = {pl, pr};
  void** bufs[2] = {bl, br};
  for (int k = 0; k < 2; ++k) {
    *outp[k] = *in[k];
    if (in[k]->type_id == t) continue;
    *bufs[k] = calloc((size_t)in[k]->length + 1, 8);
    *rc = orc_cast_numeric(in[k], t, 1, *bufs[k]);
    if (*rc != ORC_OK) return 1;
    /* the cast output starts at element 0, the validity view keeps its offset: re-base the values */
    const size_t width = (size_t)(is_float(t) ? (t == ORC_FLOAT ? 32 : 64) : int_bits(t)) / 8;
    outp[k]->type_id = t;
    outp[k]->values = (const char*)*bufs[k] - (size_t)in[k]->offset * width;
  }
  return 1;
}

int orc_compare(int op, const orc_array* lhs, const void* ls, const orc_array* rhs, const void* rs,
                uint8_t* out_bits, uint8_t* out_validity, orc_out* out) {
  if (!lhs && !rhs) return fail(ORC_INVALID, "compare: need at least one array operand");
  if (lhs && rhs && lhs->length != rhs->length)
    return fail(ORC_INVALID, "Array arguments must all be the same length");
  {
    orc_array pl, pr; void *bl, *br; int rc;
    if (promote_pair(lhs, rhs, &pl, &pr, &bl, &br, &rc)) {
      if (rc == ORC_OK) rc = orc_compare(op, &pl, ls, &pr, rs, out_bits, out_validity, out);
      free(bl); free(br);
      return rc;
    }
  }
  const int64_t n = lhs ? lhs->length : rhs->length;
  const int t = lhs ? lhs->type_id : rhs->type_id;
  propagate(lhs, rhs, n, out_validity, out);
  memset(out_bits, 0, bytes_for_bits(n)); /* GenerateBitsUnrolled leaves trailing bits 0 */
  switch (t) {
    case ORC_BOOL:
      for (int64_t i = 0; i < n; ++i) {
        const int l = lhs ? bool_at(lhs, i) : (*(const uint8_t*)ls != 0);
        const int r = rhs ? bool_at(rhs, i) : (*(const uint8_t*)rs != 0);
        int v = 0;
        switch (op) { case ORC_EQ: v = l == r; break; case ORC_NE: v = l != r; break;
          case ORC_GT: v = l > r; break; case ORC_GE: v = l >= r; break;
          case ORC_LT: v = r > l; break; case ORC_LE: v = r >= l; break; }
        if (v) set_bit(out_bits, i);
      }
      break;
    case ORC_UINT8: CMP_LOOP(uint8_t) break;   case ORC_INT8: CMP_LOOP(int8_t) break;
    case ORC_UINT16: CMP_LOOP(uint16_t) break; case ORC_INT16: CMP_LOOP(int16_t) break;
    case ORC_UINT32: CMP_LOOP(uint32_t) break; case ORC_INT32: CMP_LOOP(int32_t) break;
    case ORC_UINT64: CMP_LOOP(uint64_t) break; case ORC_INT64: CMP_LOOP(int64_t) break;
    case ORC_FLOAT: CMP_LOOP(float) break;     case ORC_DOUBLE: CMP_LOOP(double) break;
    default: return fail(ORC_NOT_IMPLEMENTED, "compare: type");
  }
  return ORC_OK;
}

/* ---- e: arithmetic, scalar_arithmetic.cc:69-251 ------------------------------------- */
/* INT: T signed?, U = unsigned twin.  Unchecked signed ops go through the unsigned type
 * (SafeSignedAdd etc., util/int_util_internal.h). */
#define ARITH_INT(T, U, SIGNED, TMIN)                                                     \
  for (int64_t i = 0; i < n; ++i) {                                                       \
    const int valid = get_bit(out_validity, i);                                           \
    const T l = LOAD_AS(T, lhs, ls, i), r = LOAD_AS(T, rhs, rs, i);                       \
    T res = 0;                                                                            \
    switch (op) {                                                                         \
      case ORC_ADD: res = (T)((U)l + (U)r); break;                                        \
      case ORC_SUB: res = (T)((U)l - (U)r); break;                                        \
      case ORC_MUL: res = (T)((uint64_t)(U)l * (uint64_t)(U)r); break; /* :145-185 */      \
      case ORC_NEGATE: res = (T)(U)(~(U)l + 1); break; /* :259-267 */                      \
      case ORC_NEGATE_CHECKED: if (!valid) break; /* :270-279 */                          \
        if (__builtin_sub_overflow((T)0, l, &res)) err = "overflow"; break;               \
      case ORC_POWER: /* :296-318, every slot */                                          \
        if (SIGNED && r < (T)0) { err = kNegPower; res = 0; break; }                      \
        res = (T)integer_power((uint64_t)l, (uint64_t)r); break;                          \
      case ORC_POWER_CHECKED: if (!valid) break; /* :322-351 */                           \
        if (SIGNED && r < (T)0) { err = kNegPower; res = 0; break; }                      \
        if (r == 0) { res = 1; break; }                                                   \
        { int ovf = 0; T pw = 1;                                                          \
          uint64_t bitmask = 1ull << (63 - __builtin_clzll((uint64_t)r));                 \
          while (bitmask) {                                                               \
            ovf |= __builtin_mul_overflow(pw, pw, &pw);                                   \
            if ((uint64_t)r & bitmask) ovf |= __builtin_mul_overflow(pw, l, &pw);         \
            bitmask >>= 1;                                                                \
          }                                                                               \
          if (ovf) err = "overflow";                                                      \
          res = pw; }                                                                     \
        break;                                                                            \
      case ORC_ADD_CHECKED: if (!valid) break;                                            \
        if (__builtin_add_overflow(l, r, &res)) err = "overflow"; break;                  \
      case ORC_SUB_CHECKED: if (!valid) break;                                            \
        if (__builtin_sub_overflow(l, r, &res)) err = "overflow"; break;                  \
      case ORC_MUL_CHECKED: if (!valid) break;                                            \
        if (__builtin_mul_overflow(l, r, &res)) err = "overflow"; break;                  \
      case ORC_DIV: case ORC_DIV_CHECKED: if (!valid) break;                              \
        /* DivideWithOverflow: r == 0 or (MIN / -1) */                                    \
        if (r == 0) { err = "divide by zero"; res = 0; }                                  \
        else if (SIGNED && l == (T)TMIN && r == (T)-1) {                                  \
          if (op == ORC_DIV_CHECKED) err = "overflow"; res = 0; } /* :222-228, :239-243 */ \
        else res = (T)(l / r);                                                            \
        break;                                                                            \
    }                                                                                     \
    ((T*)out_values)[i] = res;                                                            \
  }
#define ARITH_FLT(T, POWFN)                                                                    \
  for (int64_t i = 0; i < n; ++i) {                                                       \
    const int valid = get_bit(out_validity, i);                                           \
    const T l = LOAD_AS(T, lhs, ls, i), r = LOAD_AS(T, rhs, rs, i);                       \
    T res = 0;                                                                            \
    switch (op) {                                                                         \
      case ORC_ADD: res = l + r; break; case ORC_SUB: res = l - r; break;                 \
      case ORC_MUL: res = l * r; break;                                                   \
      case ORC_NEGATE: res = -l; break;                                                   \
      case ORC_NEGATE_CHECKED: if (valid) res = -l; break;                                \
      case ORC_POWER: res = (T)POWFN(l, r); break;                           /* :314-318 */ \
      case ORC_POWER_CHECKED: if (valid) res = (T)POWFN(l, r); break;                     \
      case ORC_ADD_CHECKED: if (valid) res = l + r; break;                                \
      case ORC_SUB_CHECKED: if (valid) res = l - r; break;                                \
      case ORC_MUL_CHECKED: if (valid) res = l * r; break;                                \
      case ORC_DIV: if (valid) res = l / r; break;                          /* :206-210 */ \
      case ORC_DIV_CHECKED: if (!valid) break;                                            \
        if (r == 0) { err = "divide by zero"; res = 0; } else res = l / r; break; /* :244-250 */ \
    }                                                                                     \
    ((T*)out_values)[i] = res;                                                            \
  }
static const char* const kNegPower = "integers to negative integer powers are not allowed";
static uint64_t integer_power(uint64_t base, uint64_t exp) { /* Power::IntegerPower :298-307 */
  uint64_t pow = 1;
  while (exp) {
    pow *= (exp & 1) ? base : 1;
    base *= base;
    exp >>= 1;
  }
  return pow;
}
int orc_arith(int op, const orc_array* lhs, const void* ls, const orc_array* rhs, const void* rs,
              void* out_values, uint8_t* out_validity, orc_out* out) {
  static const uint64_t zero = 0;
  if (op < ORC_ADD || op > ORC_NEGATE_CHECKED) return fail(ORC_INVALID, "arith: bad operator");
  if (op >= ORC_NEGATE) { rhs = NULL; rs = &zero; } /* unary kernels, :437-466 */
  if (op == ORC_NEGATE_CHECKED && lhs && is_unsigned_int(lhs->type_id))
  { /* :454-466: no kernel is registered -> DispatchExact's NoMatchingKernel (function.cc:112-125) */
    static char msg[128];
    sprintf(msg, "Function negate_checked has no kernel matching input types (array[uint%d])", int_bits(lhs->type_id));
    return fail(ORC_NOT_IMPLEMENTED, msg);
  }
  if (!lhs && !rhs) return fail(ORC_INVALID, "arith: need at least one array operand");
  if (lhs && rhs && lhs->length != rhs->length)
    return fail(ORC_INVALID, "Array arguments must all be the same length");
  {
    orc_array pl, pr; void *bl, *br; int rc;
    if (promote_pair(lhs, rhs, &pl, &pr, &bl, &br, &rc)) {
      if (rc == ORC_OK) rc = orc_arith(op, &pl, ls, &pr, rs, out_values, out_validity, out);
      free(bl); free(br);
      return rc;
    }
  }
  const int64_t n = lhs ? lhs->length : rhs->length;
  const int t = lhs ? lhs->type_id : rhs->type_id;
  propagate(lhs, rhs, n, out_validity, out);
  const char* err = NULL; /* *st is overwritten per failing element: the LAST one wins */
  switch (t) {
    case ORC_UINT8: ARITH_INT(uint8_t, uint8_t, 0, 0) break;
    case ORC_INT8: ARITH_INT(int8_t, uint8_t, 1, INT8_MIN) break;
    case ORC_UINT16: ARITH_INT(uint16_t, uint16_t, 0, 0) break;
    case ORC_INT16: ARITH_INT(int16_t, uint16_t, 1, INT16_MIN) break;
    case ORC_UINT32: ARITH_INT(uint32_t, uint32_t, 0, 0) break;
    case ORC_INT32: ARITH_INT(int32_t, uint32_t, 1, INT32_MIN) break;
    case ORC_UINT64: ARITH_INT(uint64_t, uint64_t, 0, 0) break;
    case ORC_INT64: ARITH_INT(int64_t, uint64_t, 1, INT64_MIN) break;
    case ORC_FLOAT: ARITH_FLT(float, powf) break;
    case ORC_DOUBLE: ARITH_FLT(double, pow) break;
    default: return fail(ORC_NOT_IMPLEMENTED, "arith: type");
  }
  if (err) return fail(ORC_INVALID, err);
  return ORC_OK;
}

/* ---- 8f.1: boolean kernels, scalar_boolean.cc ------------------------------------------ */
static int64_t exact_null_count(const orc_array* a) { /* ArrayData::GetNullCount */
  if (!a->validity) return 0;
  if (a->null_count >= 0) return a->null_count;
  int64_t n = 0;
  for (int64_t i = 0; i < a->length; ++i) n += !slot_valid(a, i);
  return n;
}
static void all_valid(uint8_t* v, int64_t n) { for (int64_t i = 0; i < n; ++i) set_bit(v, i); }
/* finish a Kleene result whose validity bits were computed into out_validity */
static void kleene_done(int64_t n, const uint8_t* out_validity, orc_out* out) {
  int64_t nulls = 0;
  for (int64_t i = 0; i < n; ++i) nulls += !get_bit(out_validity, i);
  out->has_validity = 1; out->null_count = nulls;
}
static void kleene_no_validity(int64_t n, uint8_t* out_validity, orc_out* out) {
  all_valid(out_validity, n); /* buffers[0] = nullptr, null_count = 0 */
  out->has_validity = 0; out->null_count = 0;
}
int orc_boolean(int op, const orc_array* lhs, int lsc, const orc_array* rhs, int rsc,
                uint8_t* out_bits, uint8_t* out_validity, orc_out* out) {
  if (op < ORC_INVERT || op > ORC_OR_KLEENE) return fail(ORC_KEY_ERROR, "boolean: no such function");
  if (op == ORC_INVERT) { rhs = NULL; rsc = ORC_SCALAR_TRUE; }
  if (!lhs && !rhs) return fail(ORC_INVALID, "boolean: need at least one array operand");
  if ((lhs && lhs->type_id != ORC_BOOL) || (rhs && rhs->type_id != ORC_BOOL))
    return fail(ORC_NOT_IMPLEMENTED, "boolean: operands must be boolean");
  if (lhs && rhs && lhs->length != rhs->length)
    return fail(ORC_INVALID, "Array arguments must all be the same length");
  const int64_t n = lhs ? lhs->length : rhs->length;
  memset(out_bits, 0, bytes_for_bits(n));
  if (op <= ORC_XOR) {
    /* INTERSECTION: a null scalar is NullGeneralization ALL_NULL (exec.cc:258-281) */
    if ((!lhs && lsc == ORC_SCALAR_NULL) || (op != ORC_INVERT && !rhs && rsc == ORC_SCALAR_NULL)) {
      out->length = n; out->has_validity = 1; out->null_count = n ? n : -1;
      memset(out_validity, 0, bytes_for_bits(n));
      return ORC_OK; /* Op::Call with an invalid scalar writes nothing (:122-130 etc.) */
    }
    propagate(lhs, rhs, n, out_validity, out);
    for (int64_t i = 0; i < n; ++i) { /* the op on the data bits of every slot */
      const int l = lhs ? bool_at(lhs, i) : lsc, r = rhs ? bool_at(rhs, i) : rsc;
      int v;
      switch (op) {
        case ORC_INVERT: v = !l; break;           /* CopyFromInverted :103-106 */
        case ORC_AND: v = l & r; break;           /* :132-139 */
        case ORC_AND_NOT: v = l & !r; break;      /* :357-387 */
        case ORC_OR: v = l | r; break;            /* :224-231 */
        default: v = l ^ r; break;                /* :349-356 */
      }
      if (v) set_bit(out_bits, i);
    }
    return ORC_OK;
  }
  /* Kleene: and_not_kleene(x, y) = and_kleene(x, invert y) slot by slot (:389-458) */
  out->length = n;
  memset(out_validity, 0, bytes_for_bits(n));
  if (n == 0) { out->has_validity = 0; out->null_count = 0; return ORC_OK; } /* SimpleBinary returns at once, codegen_internal.h:474 */
  const int lnulls = lhs ? exact_null_count(lhs) > 0 : lsc == ORC_SCALAR_NULL;
  const int rnulls = rhs ? exact_null_count(rhs) > 0 : rsc == ORC_SCALAR_NULL;
  for (int64_t i = 0; i < n; ++i) {
    int lv = lhs ? slot_valid(lhs, i) : lsc != ORC_SCALAR_NULL, ld = lhs ? bool_at(lhs, i) : lsc == ORC_SCALAR_TRUE;
    int rv = rhs ? slot_valid(rhs, i) : rsc != ORC_SCALAR_NULL, rd = rhs ? bool_at(rhs, i) : rsc == ORC_SCALAR_TRUE;
    if (op == ORC_AND_NOT_KLEENE) rd = !rd;
    const int lt = lv & ld, lf = lv & !ld, rt = rv & rd, rf = rv & !rd;
    int data, valid;
    if (op == ORC_OR_KLEENE) { data = lt | rt; valid = lt | rt | (lf & rf); }   /* :311-319 */
    else { data = lt & rt; valid = lf | rf | (lt & rt); }                      /* :211-217, :447-453 */
    if (data) set_bit(out_bits, i);
    if (valid) set_bit(out_validity, i);
  }
  /* Where the reference copies the data bitmap instead of computing it (scalar cases) the
   * copied bits of NULL slots differ from lt&rt; restate those cases literally. */
  const orc_array* arr = lhs ? lhs : rhs;
  if (!lhs || !rhs) {
    int sc = lhs ? rsc : lsc;                       /* the scalar operand */
    const int scalar_is_left = !lhs;
    int invert_array = 0;
    if (op == ORC_AND_NOT_KLEENE) {
      if (scalar_is_left) invert_array = 1;         /* KleeneAndNot(scalar, array) :396-431 */
      else if (sc != ORC_SCALAR_NULL) sc = !sc;     /* KleeneAnd(array, InvertScalar) :433-436 */
    }
    const int absorbing = op == ORC_OR_KLEENE ? ORC_SCALAR_TRUE : ORC_SCALAR_FALSE;
    memset(out_bits, 0, bytes_for_bits(n));
    if (sc == absorbing) {                          /* all false / all true, no validity */
      if (op == ORC_OR_KLEENE) for (int64_t i = 0; i < n; ++i) set_bit(out_bits, i);
      kleene_no_validity(n, out_validity, out);
      return ORC_OK;
    }
    for (int64_t i = 0; i < n; ++i)                 /* data = (inverted) copy of the array's data bits */
      if (bool_at(arr, i) ^ invert_array) set_bit(out_bits, i);
    if (sc != ORC_SCALAR_NULL) {                    /* neutral scalar: validity = the array's */
      if (exact_null_count(arr) == 0) { kleene_no_validity(n, out_validity, out); return ORC_OK; }
      memset(out_validity, 0, bytes_for_bits(n));
      for (int64_t i = 0; i < n; ++i) if (slot_valid(arr, i)) set_bit(out_validity, i);
      kleene_done(n, out_validity, out);
      return ORC_OK;
    }
    /* null scalar: valid iff the array slot is valid and decides the result by itself */
    memset(out_validity, 0, bytes_for_bits(n));
    for (int64_t i = 0; i < n; ++i) {
      const int d = bool_at(arr, i) ^ invert_array;  /* and: decides when false; or: when true */
      const int decides = op == ORC_OR_KLEENE ? d : !d;
      if (slot_valid(arr, i) && decides) set_bit(out_validity, i);
    }
    kleene_done(n, out_validity, out);
    return ORC_OK;
  }
  if (!lnulls && !rnulls) {                         /* plain And/AndNot/Or on the data bits (:205-209) */
    memset(out_bits, 0, bytes_for_bits(n));
    for (int64_t i = 0; i < n; ++i) {
      const int l = bool_at(lhs, i), r = bool_at(rhs, i);
      const int v = op == ORC_OR_KLEENE ? (l | r) : op == ORC_AND_KLEENE ? (l & r) : (l & !r);
      if (v) set_bit(out_bits, i);
    }
    kleene_no_validity(n, out_validity, out);
    return ORC_OK;
  }
  kleene_done(n, out_validity, out);
  return ORC_OK;
}

/* ---- is_valid / is_null / is_nan, scalar_validity.cc ------------------------------------- */
int orc_validity_test(int func, const orc_array* a, uint8_t* out_bits, orc_out* out) {
  const int64_t n = a->length;
  out->length = n; out->has_validity = 0; out->null_count = 0; /* OUTPUT_NOT_NULL :100 */
  memset(out_bits, 0, bytes_for_bits(n));
  const int may_have_nulls = a->validity != NULL && a->null_count != 0; /* ArrayData::MayHaveNulls */
  for (int64_t i = 0; i < n; ++i) {
    const int valid = may_have_nulls ? slot_valid(a, i) : 1;
    if (func == 0 ? valid : !valid) set_bit(out_bits, i);
  }
  return ORC_OK;
}
int orc_is_nan(const orc_array* a, uint8_t* out_bits, uint8_t* out_validity, orc_out* out) {
  if (a->type_id != ORC_FLOAT && a->type_id != ORC_DOUBLE) return fail(ORC_NOT_IMPLEMENTED, "is_nan: floating point only");
  const int64_t n = a->length;
  propagate(a, NULL, n, out_validity, out);
  memset(out_bits, 0, bytes_for_bits(n));
  for (int64_t i = 0; i < n; ++i) {
    const int v = a->type_id == ORC_FLOAT ? isnan(((const float*)a->values)[a->offset + i])
                                          : isnan(((const double*)a->values)[a->offset + i]);
    if (v) set_bit(out_bits, i);
  }
  return ORC_OK;
}

/* ---- fill_null, scalar_fill_null.cc:44-147 ---------------------------------------------- */
int orc_fill_null(const orc_array* a, const void* fill, void* out_values, uint8_t* out_validity, orc_out* out) {
  const int bw = orc_bit_width(a->type_id);
  if (bw < 0) return fail(ORC_NOT_IMPLEMENTED, "fill_null: type");
  const int64_t n = a->length;
  const int may_have_nulls = a->validity != NULL && a->null_count != 0;
  out->length = n;
  memset(out_validity, 0, bytes_for_bits(n));
  if (bw == 1) memset(out_values, 0, bytes_for_bits(n));
  if (n == 0) { out->has_validity = 0; out->null_count = 0; return ORC_OK; } /* empty result of a zero-length call */
  const int replace = may_have_nulls && fill != NULL; /* :55, :107 */
  for (int64_t i = 0; i < n; ++i) {
    const int take_fill = replace && !slot_valid(a, i);
    if (bw == 1) {
      const int v = take_fill ? (*(const uint8_t*)fill != 0) : bool_at(a, i);
      if (v) set_bit((uint8_t*)out_values, i);
    } else {
      const size_t w = (size_t)bw / 8;
      memcpy((uint8_t*)out_values + (size_t)i * w,
             take_fill ? (const uint8_t*)fill : (const uint8_t*)a->values + (size_t)(a->offset + i) * w, w);
    }
  }
  if (replace) { out->has_validity = 0; out->null_count = 0; all_valid(out_validity, n); return ORC_OK; }
  /* *output = data :84 -- ArrayData::Make drops a validity buffer whose stored count is 0 (array/data.cc:50-60) */
  out->has_validity = may_have_nulls;
  out->null_count = may_have_nulls ? a->null_count : 0;
  for (int64_t i = 0; i < n; ++i) if (!may_have_nulls || slot_valid(a, i)) set_bit(out_validity, i);
  return ORC_OK;
}

/* ---- 8f.2: scalar aggregates, aggregate_basic_internal.h / aggregate_internal.h ---------- */
typedef struct { double sum[64]; uint64_t mask; int root; } pairwise; /* SumArray state :80-103 */
static void pw_reduce(pairwise* p, double block_sum) {
  int lvl = 0; uint64_t m = 1;
  p->sum[lvl] += block_sum; p->mask ^= m;
  while ((p->mask & m) == 0) {
    block_sum = p->sum[lvl]; p->sum[lvl] = 0; ++lvl; m <<= 1;
    p->sum[lvl] += block_sum; p->mask ^= m;
  }
  if (lvl > p->root) p->root = lvl;
}
static double value_as_double(const orc_array* a, int64_t i) {
  return a->type_id == ORC_FLOAT ? (double)((const float*)a->values)[a->offset + i] : ((const double*)a->values)[a->offset + i];
}
int orc_aggregate(const orc_array* a, orc_agg* out) {
  const int t = a->type_id;
  if (orc_bit_width(t) < 0) return fail(ORC_NOT_IMPLEMENTED, "No sum implemented");
  const int64_t n = a->length;
  const int is_f = t == ORC_FLOAT || t == ORC_DOUBLE;
  const int is_signed = t == ORC_INT8 || t == ORC_INT16 || t == ORC_INT32 || t == ORC_INT64;
  int64_t valid = 0;
  for (int64_t i = 0; i < n; ++i) valid += slot_valid(a, i);
  out->valid_count = valid;
  /* sum */
  if (is_f) {
    pairwise p; memset(&p, 0, sizeof(p));
    double s = 0;
    if (valid > 0) {
      int64_t i = 0;
      while (i < n) { /* VisitSetBitRunsVoid: maximal runs of valid slots */
        while (i < n && !slot_valid(a, i)) ++i;
        int64_t j = i;
        while (j < n && slot_valid(a, j)) ++j;
        int64_t k = i;
        for (; k + 16 <= j; k += 16) { double b = 0; for (int q = 0; q < 16; ++q) b += value_as_double(a, k + q); pw_reduce(&p, b); }
        if (k < j) { double b = 0; for (; k < j; ++k) b += value_as_double(a, k); pw_reduce(&p, b); }
        i = j;
      }
      for (int l = 1; l <= p.root; ++l) p.sum[l] += p.sum[l - 1];
      s = p.sum[p.root];
    }
    memcpy(&out->sum, &s, 8);
  } else {
    uint64_t s = 0; /* signed sums wrap the same way in two's complement */
    for (int64_t i = 0; i < n; ++i) if (slot_valid(a, i)) {
      switch (t) {
        case ORC_BOOL: s += (uint64_t)bool_at(a, i); break;
        case ORC_UINT8: s += ((const uint8_t*)a->values)[a->offset + i]; break;
        case ORC_INT8: s += (uint64_t)(int64_t)((const int8_t*)a->values)[a->offset + i]; break;
        case ORC_UINT16: s += ((const uint16_t*)a->values)[a->offset + i]; break;
        case ORC_INT16: s += (uint64_t)(int64_t)((const int16_t*)a->values)[a->offset + i]; break;
        case ORC_UINT32: s += ((const uint32_t*)a->values)[a->offset + i]; break;
        case ORC_INT32: s += (uint64_t)(int64_t)((const int32_t*)a->values)[a->offset + i]; break;
        case ORC_UINT64: s += ((const uint64_t*)a->values)[a->offset + i]; break;
        default: s += (uint64_t)((const int64_t*)a->values)[a->offset + i]; break;
      }
    }
    out->sum = s;
  }
  /* min / max over valid slots */
  if (is_f) {
    double mn = INFINITY, mx = -INFINITY;
    for (int64_t i = 0; i < n; ++i) if (slot_valid(a, i)) { const double v = value_as_double(a, i); mn = fmin(mn, v); mx = fmax(mx, v); }
    memcpy(&out->min, &mn, 8); memcpy(&out->max, &mx, 8);
  } else if (t == ORC_BOOL) {
    uint64_t mn = 1, mx = 0;
    for (int64_t i = 0; i < n; ++i) if (slot_valid(a, i)) { const uint64_t v = (uint64_t)bool_at(a, i); if (v < mn) mn = v; if (v > mx) mx = v; }
    out->min = mn; out->max = mx;
  } else if (is_signed) {
    const int bw = orc_bit_width(t);
    int64_t mn = bw == 64 ? INT64_MAX : ((int64_t)1 << (bw - 1)) - 1, mx = bw == 64 ? INT64_MIN : -((int64_t)1 << (bw - 1));
    for (int64_t i = 0; i < n; ++i) if (slot_valid(a, i)) {
      int64_t v;
      switch (t) { case ORC_INT8: v = ((const int8_t*)a->values)[a->offset + i]; break;
        case ORC_INT16: v = ((const int16_t*)a->values)[a->offset + i]; break;
        case ORC_INT32: v = ((const int32_t*)a->values)[a->offset + i]; break;
        default: v = ((const int64_t*)a->values)[a->offset + i]; break; }
      if (v < mn) mn = v; if (v > mx) mx = v;
    }
    out->min = (uint64_t)mn; out->max = (uint64_t)mx;
  } else {
    const int bw = orc_bit_width(t);
    uint64_t mn = bw == 64 ? UINT64_MAX : ((uint64_t)1 << bw) - 1, mx = 0;
    for (int64_t i = 0; i < n; ++i) if (slot_valid(a, i)) {
      uint64_t v;
      switch (t) { case ORC_UINT8: v = ((const uint8_t*)a->values)[a->offset + i]; break;
        case ORC_UINT16: v = ((const uint16_t*)a->values)[a->offset + i]; break;
        case ORC_UINT32: v = ((const uint32_t*)a->values)[a->offset + i]; break;
        default: v = ((const uint64_t*)a->values)[a->offset + i]; break; }
      if (v < mn) mn = v; if (v > mx) mx = v;
    }
    out->min = mn; out->max = mx;
  }
  return ORC_OK;
}

/* ---- headline composition: Filter -> GroupBy{hash_sum} ------------------------------- */
int64_t orc_filter_groupby_sum(const orc_array* values, const orc_array* keys, const orc_array* mask,
                               void* out_sums, int64_t* out_counts, void* out_keys,
                               uint8_t* out_key_validity) {
  const int64_t n = values->length;
  const int vw = orc_bit_width(values->type_id) / 8, kw = orc_bit_width(keys->type_id) / 8;
  orc_out o1, o2;
  const int64_t m = orc_filter_output_size(mask, ORC_DROP);
  /* boolean columns are bit-packed (bit width 1 -> vw / kw == 0) */
  uint8_t* fv = (uint8_t*)calloc(vw ? (size_t)(m ? m : 1) * vw : (size_t)bytes_for_bits(m) + 8, 1);
  uint8_t* fk = (uint8_t*)calloc(kw ? (size_t)(m ? m : 1) * kw : (size_t)bytes_for_bits(m) + 8, 1);
  uint8_t* fvv = (uint8_t*)malloc((size_t)bytes_for_bits(m) + 1);
  uint8_t* fkv = (uint8_t*)malloc((size_t)bytes_for_bits(m) + 1);
  (void)n;
  orc_filter(values, mask, ORC_DROP, fv, fvv, &o1); /* FilterRecordBatch-style: same mask on each column */
  orc_filter(keys, mask, ORC_DROP, fk, fkv, &o2);
  orc_array V = {fv, o1.has_validity ? fvv : NULL, m, 0, values->type_id, o1.null_count};
  orc_array K = {fk, o2.has_validity ? fkv : NULL, m, 0, keys->type_id, o2.null_count};
  uint32_t* ids = (uint32_t*)malloc((size_t)(m ? m : 1) * 4);
  orc_grouper* g = orc_grouper_new(1, &keys->type_id);
  orc_grouper_consume(g, &K, ids);
  const int64_t ng = orc_grouper_num_groups(g);
  memset(out_sums, 0, (size_t)ng * 8);
  memset(out_counts, 0, (size_t)ng * 8);
  orc_hash_sum(&V, ids, out_sums, out_counts);
  int64_t knc;
  orc_grouper_uniques(g, 0, out_keys, out_key_validity, &knc);
  orc_grouper_free(g);
  free(fv); free(fk); free(fvv); free(fkv); free(ids);
  return ng;
}

/* Grouper::MakeGroupings, hash_aggregate.cc:979-1018: count the rows of every id (:990-994), exclusive
 * prefix sum (:995-1001), then a stable scatter of the row numbers through a copy of the offsets
 * (:1004-1013) -- a counting sort of the rows by id. */
int orc_make_groupings(const uint32_t* ids, int64_t n, uint32_t num_groups, int32_t* out_offsets,
                       int32_t* out_indices) {
  for (int64_t i = 0; i < n; ++i)
    if (ids[i] >= num_groups) return fail(ORC_INVALID, "MakeGroupings: an id is not below num_groups");
  memset(out_offsets, 0, sizeof(int32_t) * ((size_t)num_groups + 1));
  for (int64_t i = 0; i < n; ++i) out_offsets[ids[i]] += 1;
  int32_t length = 0;
  for (uint32_t g = 0; g < num_groups; ++g) {
    const int32_t c = out_offsets[g];
    out_offsets[g] = length;
    length += c;
  }
  out_offsets[num_groups] = length;
  int32_t* cursor = (int32_t*)malloc(sizeof(int32_t) * ((size_t)num_groups + 1));
  memcpy(cursor, out_offsets, sizeof(int32_t) * ((size_t)num_groups + 1));
  for (int64_t i = 0; i < n; ++i) out_indices[cursor[ids[i]]++] = (int32_t)i;
  free(cursor);
  return ORC_OK;
}

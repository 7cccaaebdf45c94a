// groupby.hip -- Grouper, hash_sum / hash_count / hash_mean, and the fused
// Filter -> GroupBy of the north-star path.
//
// Reference semantics: compute/kernels/hash_aggregate.cc
//   KeyEncoder / FixedWidthKeyEncoder / BooleanKeyEncoder :48-196  (null key = its own group,
//       byte-wise equality: -0.0 != +0.0, NaNs with equal bits are one group)
//   GrouperImpl::Consume :358-408   (group id = order of FIRST APPEARANCE, map_.emplace(key, num_groups_))
//   GroupedCountImpl :466-517, GroupedSumImpl :522-622 (+FindAccumulatorType, aggregate_internal.h:29-50)
//   GroupBy :890-966                (output rows = groups in id order)
//
// Design (see DESIGN.md "group-by"):
//   * keys are packed into one 64-bit word x (value bits + one null flag per column) and
//     hashed with the BIJECTIVE murmur3 finaliser h = fmix64(x); tables store h only -- equal
//     h <=> equal key, and the key is recovered with the inverse at output time.
//   * build: open-addressed table of h; per selected row  first_row = min(row),
//     sum += v, count += 1.
//   * first-appearance ids without a sort: set bit first_row of every group in an N-bit
//     bitmap, exclusive popcount-scan the bitmap words, id(g) = rank of its first_row bit.
//     Traffic N/8 bytes instead of a radix sort of the groups.
//   * emit: scatter (x, sum, count, first_row) of each group to slot id, decode keys.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace amdarrow {

// The strategy ladder, one file per rung (all part of this translation unit):
#include "groupby_table.inc"      // key words, the global table, first-appearance numbering, aggregates over ids
#include "groupby_partition.inc"  // two-pass partition; one-pass scatters with staged tiles
#include "groupby_bins.inc"       // bin scatters: 16-byte and compact records
#include "groupby_aggregate.inc"  // partition aggregates
#include "groupby_local.inc"      // per-workgroup pre-aggregation, heavy hitters, gates and merges

}  // namespace amdarrow

using namespace amdarrow;

#include "groupby_grouper.inc"    // Grouper and the hash aggregates over ids


// The fused Filter -> GroupBy.  `locked`: the caller already holds ctx->mu (the multi-key entry point, which runs
// this on its column of packed key codes).
constexpr int kHintProbeEvery = 64;  // hinted group-by calls between two calls that try the per-workgroup pre-aggregation again
// ---- null keys, split off ------------------------------------------------------------------------------------
// The rows whose key is null are ONE group.  Carried through the scatters they cost the specialised kernels (whose loaders
// take 16 bytes of keys per lane and know no validity) and, the null key being on more than a fraction of a per cent of
// the rows, the heavy-hitter pass on top: 1.9 ms per 100 M rows with 1 % or 30 % null keys.  Instead one kernel over the
// bitmaps (k_null_split) writes mask' = mask AND key validity and aggregates the null group's rows by itself (sum, valid
// count, first row); the group-by proper runs on keys without validity under mask' -- every specialised path is open to
// it -- and the null group enters the finished table as one more entry before the first-appearance numbering
// (k_inject_null: a free slot, the key word every path uses for the null key).
struct NullAcc { unsigned long long sum, count; uint32_t first, pad; };
__global__ __launch_bounds__(kThreads) void k_null_split(const uint8_t* m_data, const uint8_t* m_valid, int64_t m_off,
                                                         const uint8_t* k_valid, int64_t k_off, ValueCol v, int has_values,
                                                         int64_t n, unsigned long long* out_mask, NullAcc* acc) {
  __shared__ unsigned long long s_sum[kThreads / 64], s_cnt[kThreads / 64];
  __shared__ uint32_t s_first[kThreads / 64];
  const int64_t nwords = (n + 63) >> 6;
  unsigned long long isum = 0, cnt = 0;
  double dsum = 0.0;
  uint32_t first = kNoRow;
  bool is_double = false;
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) {
    const int nbits = n - 64 * w >= 64 ? 64 : static_cast<int>(n - 64 * w);
    uint64_t sel = LoadBitsOrOnes(m_data, m_off + 64 * w, nbits);
    if (m_valid) sel &= LoadBits64(m_valid, m_off + 64 * w, nbits);
    const uint64_t kv = LoadBitsOrOnes(k_valid, k_off + 64 * w, nbits);  // (no key validity: the kernel only folds the mask's validity into the mask)
    out_mask[w] = sel & kv;
    uint64_t nulls = sel & ~kv;
    if (nulls && first == kNoRow) first = static_cast<uint32_t>(64 * w + __builtin_ctzll(nulls));  // (a thread's words ascend)
    while (nulls && has_values) {
      const int64_t row = 64 * w + __builtin_ctzll(nulls);
      nulls &= nulls - 1;
      if (v.valid && !GetBit(v.valid, v.off + row)) continue;
      const uint64_t bits = LoadAccum(v, row, &is_double);
      if (is_double) dsum += __longlong_as_double(static_cast<long long>(bits)); else isum += bits;
      ++cnt;
    }
  }
  is_double = v.type_id == AMDARROW_FLOAT || v.type_id == AMDARROW_DOUBLE;
  unsigned long long sum = is_double ? static_cast<unsigned long long>(__double_as_longlong(dsum)) : isum;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long os = __shfl_xor(sum, d), oc = __shfl_xor(cnt, d);
    const uint32_t of = __shfl_xor(first, d);
    sum = is_double ? static_cast<unsigned long long>(__double_as_longlong(__longlong_as_double(static_cast<long long>(sum)) + __longlong_as_double(static_cast<long long>(os)))) : sum + os;
    cnt += oc;
    first = of < first ? of : first;
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_sum[wave] = sum; s_cnt[wave] = cnt; s_first[wave] = first; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < kThreads / 64; ++k) {
      sum = is_double ? static_cast<unsigned long long>(__double_as_longlong(__longlong_as_double(static_cast<long long>(sum)) + __longlong_as_double(static_cast<long long>(s_sum[k])))) : sum + s_sum[k];
      cnt += s_cnt[k];
      first = s_first[k] < first ? s_first[k] : first;
    }
    if (first != kNoRow) {  // (one set of atomics per workgroup)
      atomicMin(&acc->first, first);
      if (cnt) {
        atomicAdd(&acc->count, cnt);
        if (is_double) atomicAdd(reinterpret_cast<double*>(&acc->sum), __longlong_as_double(static_cast<long long>(sum)));
        else atomicAdd(&acc->sum, sum);
      }
    }
  }
}
// One wave: the null group into the first free slot of the finished table, its first row marked for the numbering.
__global__ __launch_bounds__(64) void k_inject_null(GroupTable t, unsigned long long* first_bitmap, const NullAcc* acc,
                                                    unsigned long long hkey, unsigned int* overflow, int wide_slot) {
  const uint32_t first = acc->first;
  if (first == kNoRow) return;  // no selected row has a null key
  const int lane = threadIdx.x;
  const uint64_t T = t.mask + 1;
  if (wide_slot) {  // a single 64-bit key that did not range-compress: its null lives in the special slot behind the table
    if (lane == 0) {
      t.hkeys[T] = 0; t.first[T] = first; t.sums[T] = acc->sum; t.counts[T] = acc->count;
      atomicOr(&first_bitmap[first >> 6], 1ull << (first & 63));
    }
    return;
  }
  for (uint64_t base = 0; base < T; base += 64) {
    const bool free_slot = base + lane < T && t.hkeys[base + lane] == kEmpty && t.first[base + lane] == kNoRow;
    const uint64_t found = __ballot(free_slot);
    if (!found) continue;
    if (lane == __builtin_ctzll(found)) {
      t.hkeys[base + lane] = hkey; t.first[base + lane] = first; t.sums[base + lane] = acc->sum; t.counts[base + lane] = acc->count;
      atomicOr(&first_bitmap[first >> 6], 1ull << (first & 63));
    }
    return;
  }
  if (lane == 0) __hip_atomic_store(overflow, kOverflowTable, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a table without a free slot)
}

static int FilterGroupByCore(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                             const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                             void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                             int64_t* out_num_groups, bool locked, ulonglong4** packed_out, uint64_t* bias_out, const NullAcc* null_inject);
// Values narrower than 64 bits as their accumulator type (FindAccumulatorType: signed -> int64, unsigned / boolean -> uint64,
// float -> double), element i at the same offset: the specialised loaders take 8-byte values as 16-byte vectors, and an
// int32 or float32 value column sent the call to the generic kernels (1.18 ms per 100 M rows against 0.52).  The copy is 4 + 8
// bytes per row, 0.25 ms.
__global__ __launch_bounds__(kThreads) void k_widen_values(ValueCol v, int64_t n, unsigned long long* out /* element 0 = row -v.off */) {
  constexpr int kU = 4;
  const int64_t stride = int64_t(gridDim.x) * kThreads;
  for (int64_t i0 = int64_t(blockIdx.x) * kThreads + threadIdx.x; i0 < n; i0 += kU * stride) {
    uint64_t w[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + u * stride;
      bool is_double;
      w[u] = LoadAccum(v, i < n ? i : n - 1, &is_double);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n) out[v.off + i] = w[u];
    }
  }
}

static int FilterGroupByImpl(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                             const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                             void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                             int64_t* out_num_groups, bool locked, ulonglong4** packed_out, uint64_t* bias_out) {
  const int64_t n = keys->length;
  const int kbw = BitWidth(keys->type_id);
  // (the same pass folds a NULLABLE MASK's validity into the mask -- what a comparison over a nullable column yields: the bin
  // scatters take a mask without validity, and such a call ran on the generic scatter, 1.43 ms per 100 M rows instead of 0.55)
  const bool keys_nullable = keys->validity && keys->null_count != 0 && kbw >= 8 && kbw <= 64;
  const bool mask_nullable = mask && mask->validity && mask->null_count != 0;
  const bool shapes_ok = n >= (int64_t(1) << 18) && n < (int64_t(1) << 32) - 1 && (!values || values->length == n) &&
                         (!mask || (mask->type_id == AMDARROW_BOOL && mask->length == n));
  const bool null_split = (keys_nullable || mask_nullable) && shapes_ok && !getenv("AMDARROW_GROUPBY_NO_NULL_SPLIT");
  const int vbw = values ? BitWidth(values->type_id) : 64;
  const bool widen = values && vbw >= 1 && vbw < 64 && (kbw == 32 || kbw == 64) && shapes_ok && !getenv("AMDARROW_GROUPBY_NO_WIDEN");
  if (!null_split && !widen)
    return FilterGroupByCore(ctx, values, keys, mask, group_capacity, out_sums, out_counts, out_keys, out_key_validity, out_first_rows,
                             out_num_groups, locked, packed_out, bias_out, nullptr);
  std::unique_lock<std::mutex> lock(ctx->mu, std::defer_lock);
  if (!locked) lock.lock();
  const int64_t nwords = (n + 63) >> 6;
  void* block = nullptr;
  void* wblock = nullptr;
  if (null_split) AA_TRY(amdarrow_malloc(ctx, 64 + 8 * (nwords + 2), &block));
  int rc = AMDARROW_OK;
  amdarrow_array values_w{};
  if (widen) {
    rc = amdarrow_malloc(ctx, 8 * (values->offset + n) + 64, &wblock);
    if (rc == AMDARROW_OK) {
      // (row 0 of the slice on a 16-byte boundary: the bin scatters' condition)
      auto* wide = reinterpret_cast<unsigned long long*>(static_cast<char*>(wblock) + ((values->offset & 1) ? 8 : 0));
      const ValueCol v{values->values, values->validity, values->offset, values->type_id};
      hipLaunchKernelGGL(k_widen_values, dim3(GridFor(n, kThreads * 4, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, v, n, wide);
      if (hipGetLastError() != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, "group-by: value widening launch failed");
      values_w = *values;
      values_w.values = wide;
      values_w.type_id = IsFloat(values->type_id) ? AMDARROW_DOUBLE : IsSignedInt(values->type_id) ? AMDARROW_INT64 : AMDARROW_UINT64;
      values = &values_w;
    }
  }
  if (rc == AMDARROW_OK && !null_split)
    rc = FilterGroupByCore(ctx, values, keys, mask, group_capacity, out_sums, out_counts, out_keys, out_key_validity, out_first_rows,
                           out_num_groups, /*locked=*/true, packed_out, bias_out, nullptr);
  if (rc == AMDARROW_OK && null_split) {
    auto* acc = static_cast<NullAcc*>(block);
    auto* mask2 = reinterpret_cast<unsigned long long*>(static_cast<char*>(block) + 64);
    const NullAcc init{0, 0, kNoRow, 0};
    memcpy(static_cast<char*>(ctx->pinned) + kStatusOffset + 192, &init, sizeof init);
    if (hipMemcpyAsync(acc, static_cast<char*>(ctx->pinned) + kStatusOffset + 192, sizeof init, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemsetAsync(mask2 + nwords, 0, 16, ctx->stream) != hipSuccess)
      rc = SetError(AMDARROW_UNKNOWN_ERROR, "group-by: null split setup failed");
    if (rc == AMDARROW_OK) {
      ValueCol v{};
      if (values) v = ValueCol{values->values, values->validity, values->offset, values->type_id};
      hipLaunchKernelGGL(k_null_split, dim3(GridFor(nwords, kThreads, ctx->num_cus * 4)), dim3(kThreads), 0, ctx->stream,
                         mask ? static_cast<const uint8_t*>(mask->values) : nullptr, mask ? mask->validity : nullptr, mask ? mask->offset : 0,
                         keys_nullable ? keys->validity : nullptr, keys->offset, v, values ? 1 : 0, n, mask2, acc);
      if (hipGetLastError() != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, "group-by: null split launch failed");
    }
    if (rc == AMDARROW_OK) {
      amdarrow_array keys_nn = *keys;
      if (keys_nullable) { keys_nn.validity = nullptr; keys_nn.null_count = 0; }
      amdarrow_array mask_nn{};
      mask_nn.type_id = AMDARROW_BOOL; mask_nn.values = mask2; mask_nn.validity = nullptr; mask_nn.length = n; mask_nn.offset = 0; mask_nn.null_count = 0;
      rc = FilterGroupByCore(ctx, values, &keys_nn, &mask_nn, group_capacity, out_sums, out_counts, out_keys, out_key_validity, out_first_rows,
                             out_num_groups, /*locked=*/true, packed_out, bias_out, keys_nullable ? acc : nullptr);
    }
  }
  (void)hipStreamSynchronize(ctx->stream);  // (the blocks go back to the pool: nothing queued may still read them)
  if (block) (void)amdarrow_free(ctx, block);
  if (wblock) (void)amdarrow_free(ctx, wblock);
  return rc;
}

static int FilterGroupByCore(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                             const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                             void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                             int64_t* out_num_groups, bool locked, ulonglong4** packed_out, uint64_t* bias_out, const NullAcc* null_inject) {
  const int64_t n = keys->length;
  *out_num_groups = 0;
  if (values && values->length != n) return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");
  if (mask) {
    if (mask->type_id != AMDARROW_BOOL) return SetError(AMDARROW_NOT_IMPLEMENTED, "Filter argument must be boolean type");
    if (mask->length != n) return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");
  }
  if (values && BitWidth(values->type_id) < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "Summing data of this type");
  if (n >= (int64_t(1) << 32) - 1) return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: 2^32 rows or more per call");
  if (n == 0) return AMDARROW_OK;
  KeySpec ks;
  AA_TRY(MakeKeySpec(1, keys, nullptr, &ks));
  std::unique_lock<std::mutex> lock(ctx->mu, std::defer_lock);
  if (!locked) lock.lock();
  const KeySpec ks_generic = ks;  // as the reference encodes the column; used by the generic path
  // An int64 / uint64 key column whose values span less than 2^32 (ids, dates, small integers --
  // most integer key columns) is range-compressed onto 32 bits: x = key - min.  One extra pass over
  // the keys (min / max) buys the partitioned LDS paths instead of the global-atomic table.
  int k_load_bits = ks.col[0].bits;
  uint64_t k_bias = 0;
  bool bias_from_hint = false;  // the min / max pass skipped: the compact scatter checks every selected key against the bias
  if (ks.total_bits == 64 && (keys->type_id == AMDARROW_INT64 || keys->type_id == AMDARROW_UINT64) &&
      n >= (int64_t(1) << 18) && n < (int64_t(1) << 30) && !getenv("AMDARROW_GROUPBY_NO_RANGE_COMPRESSION")) {
    // a call of a shape whose last call ran on compact records takes that call's bias on trust (it WILL run on compact
    // records again, or start over: see below)
    if (!ctx->gb_no_bias_hint && !keys->validity && !getenv("AMDARROW_GROUPBY_PATH") && !getenv("AMDARROW_GROUPBY_NO_HINT") &&
        !getenv("AMDARROW_GROUPBY_NO_COMPACT")) {
      int every = kHintProbeEvery;
      if (const char* env = getenv("AMDARROW_GROUPBY_PROBE_EVERY")) every = std::max(2, atoi(env));
      for (auto& h : ctx->gb_hints) {
        if (h.n == n && h.caller == ctx->gb_caller && h.key_type == keys->type_id && h.value_type == (values ? values->type_id : 0) && h.has_mask == (mask != nullptr) &&
            h.has_value_validity == (values && values->validity != nullptr) && h.stats && h.compact_off == 0 && h.key_lo <= h.key_hi && h.key_lo >= h.k_bias &&
            h.key_hi - h.key_lo <= (uint64_t(1) << 21) - 1024 && n <= (int64_t(1) << 28) &&  // (the compact records will take it)
            (h.uses + 1) % every != 0) {
          k_bias = h.k_bias;
          bias_from_hint = true;
        }
      }
    }
    // a shape whose last call found keys spread over more than 2^32 values (ids, hashes) goes the wide-key way without
    // looking: that path assumes nothing about the keys
    bool wide_from_hint = false;
    if (!bias_from_hint && !(keys->validity && keys->null_count != 0) && !getenv("AMDARROW_GROUPBY_PATH") && !getenv("AMDARROW_GROUPBY_NO_HINT") &&
        !getenv("AMDARROW_GROUPBY_NO_WIDE")) {
      int every = kHintProbeEvery;
      if (const char* env = getenv("AMDARROW_GROUPBY_PROBE_EVERY")) every = std::max(2, atoi(env));
      for (auto& h : ctx->gb_hints)
        if (h.n == n && h.caller == ctx->gb_caller && h.key_type == keys->type_id && h.value_type == (values ? values->type_id : 0) && h.has_mask == (mask != nullptr) &&
            h.has_value_validity == (values && values->validity != nullptr) && h.wide_keys)
          wide_from_hint = (++h.wide_uses % every) != 0;
    }
    amdarrow_agg kr{};
    if (!bias_from_hint && !wide_from_hint) AA_TRY(AggregateLocked(ctx, keys, &kr));
    if (bias_from_hint || (!wide_from_hint && kr.valid_count > 0 && kr.max - kr.min < 0xFFFFFFFFull)) {  // (wrap-around difference: exact for both signednesses)
      if (!bias_from_hint) k_bias = kr.min;
      ks.total_bits = 32;
      ks.col[0].bits = 32;
      ks.wide_null_slot = 0;
    }
  }
  // (a call on a hinted bias that cannot run on compact records after all, or whose scatter finds a key the bias does
  // not hold for, starts over with the min / max pass)
  auto start_over_measured = [&]() -> int {
    for (auto& h : ctx->gb_hints) if (h.n == n && h.caller == ctx->gb_caller && h.key_type == keys->type_id) h.n = -1;
    ctx->gb_no_bias_hint = true;
    const int rc = FilterGroupByCore(ctx, values, keys, mask, group_capacity, out_sums, out_counts, out_keys, out_key_validity, out_first_rows,
                                     out_num_groups, /*locked=*/true, packed_out, bias_out, null_inject);
    ctx->gb_no_bias_hint = false;
    return rc;
  };

  ValueCol v{};
  if (values) v = ValueCol{values->values, values->validity, values->offset, values->type_id};
  const uint8_t* m_data = mask ? static_cast<const uint8_t*>(mask->values) : nullptr;
  const uint8_t* m_valid = mask ? mask->validity : nullptr;
  const int64_t m_off = mask ? mask->offset : 0;

  // path 2: one-pass partition into private regions; 1: exact two-pass partition; 0: generic
  // global table.  Each falls back to the next when its capacity assumption fails.
  const bool fast_ok = ks.total_bits <= 32 && n >= (int64_t(1) << 18) && n < (int64_t(1) << 30);
  int path = fast_ok ? 3 : 0;  // 3 = local pre-aggregation with the partition path queued behind it
  // 64-bit keys that did not range-compress (ids, hashes; doubles by their bits) and hold no null: the exact two-pass
  // partition path with the key's high word in a parallel array, instead of the global-atomic table
  const bool wide = !fast_ok && ks.total_bits == 64 && !(keys->validity && keys->null_count != 0) &&
                    n >= (int64_t(1) << 18) && n < (int64_t(1) << 30) && !getenv("AMDARROW_GROUPBY_NO_WIDE");
  if (wide) path = 1;
  // wide keys, first attempt: ONE pass -- the bin scatter on five-word records {row, key, value} (no histogram pass, no
  // scan) and the stream aggregate over them; regions that overflow (skewed keys) fall back to the exact two-pass variant
  // (the bin scatter's loaders: 8-byte values as 16-byte vectors, a mask without validity; a call without values reads the key
  // buffer where it would read values and must not have a mask either -- as for the 32-bit keys' bins)
  bool wide_onepass = wide && !m_valid && (values ? BitWidth(values->type_id) == 64 : (!mask && n >= 64)) && !getenv("AMDARROW_GROUPBY_WIDE_TWO_PASS");
  if (const char* env = getenv("AMDARROW_GROUPBY_PATH")) path = std::min(path, atoi(env));
  // Path 3 first tries to pre-aggregate every row range in LDS and looks for heavy hitters; with many
  // evenly spread groups both give up (~60 us of kernels and launches for nothing).  A call of the
  // same shape as the last one that went that way starts at the partition path directly; if its
  // capacity assumption then fails (the data changed character) the call starts over at path 3, and
  // every kHintProbeEvery-th hinted call takes path 3 anyway so that a change for the better is noticed too.
  amdarrow_ctx::GroupByHint shape;
  shape.n = n; shape.key_type = keys->type_id; shape.value_type = values ? values->type_id : 0;
  shape.has_mask = mask != nullptr; shape.has_value_validity = values && values->validity != nullptr;
  shape.caller = ctx->gb_caller;
  // the hint of this shape, or the least recently used slot (a hint learnt by this call goes there)
  amdarrow_ctx::GroupByHint* slot = &ctx->gb_hints[0];
  {
    bool found = false;
    for (auto& h : ctx->gb_hints) {
      if (h.n == shape.n && h.caller == shape.caller && h.key_type == shape.key_type && h.value_type == shape.value_type && h.has_mask == shape.has_mask &&
          h.has_value_validity == shape.has_value_validity) { slot = &h; found = true; break; }
    }
    if (!found)
      for (auto& h : ctx->gb_hints) if (h.n < 0 ? slot->n >= 0 : (slot->n >= 0 && h.last_used < slot->last_used)) slot = &h;
    slot->last_used = ++ctx->gb_hint_clock;
  }
  auto& hint = *slot;
  bool hinted = false;
  // the back-off state of this shape survives the hint being forgotten and learnt again within the call
  const bool shape_known = hint.n == shape.n && hint.caller == shape.caller && hint.key_type == shape.key_type && hint.value_type == shape.value_type &&
                           hint.has_mask == shape.has_mask && hint.has_value_validity == shape.has_value_validity;
  int carry_misses = shape_known ? hint.compact_misses : 0, carry_off = shape_known ? hint.compact_off : 0;
  bool compact_allowed = carry_off == 0;
  if (carry_off > 0) --carry_off;
  if (ctx->gb_carry_misses >= 0) {  // the retry of a call whose hinted bias did not hold
    carry_misses = ctx->gb_carry_misses; carry_off = ctx->gb_carry_off; compact_allowed = false;
    ctx->gb_carry_misses = -1;
  }
  // few groups per row range last time: the pre-aggregation alone is queued (no gated scatter / aggregate launches
  // behind it, ~5 us each); if it gives up this time the call starts over with everything
  bool local_only = false;
  int hinted_sub_bits = 0;
  if (path == 3 && shape_known &&
      !getenv("AMDARROW_GROUPBY_PATH") && !getenv("AMDARROW_GROUPBY_NO_HINT")) {
    int every = kHintProbeEvery;
    if (const char* env = getenv("AMDARROW_GROUPBY_PROBE_EVERY")) every = std::max(2, atoi(env));
    if (hint.local_ok) local_only = true;
    else if (++hint.uses % every != 0) { path = 2; hinted = true; hinted_sub_bits = hint.sub_bits; }
  }
  // Compact records (k_part_scatter_compact): a hinted call whose predecessor's selected keys spanned few enough
  // values carries 12-byte records, 8-byte ones if that call's (integer) values were all int32; the kernel checks
  // both for every selected row and a miss starts the call over without the hint.
  int compact_rw = 0, compact_pb = 9;  // words per record; log2(partitions): see CGeo
  uint32_t c_base = 0;
  const bool headline_shape = ks.total_bits == 32 && k_load_bits == 32 && !keys->validity && values &&
                              BitWidth(values->type_id) == 64 && !m_valid;
  // the bin scatter loads 16-byte vectors: keys and values must start on a 16-byte boundary
  // (also for a 64-bit key column that range-compressed onto 32 bits: the kernel loads the 64-bit keys and subtracts the bias)
  // (and for a call without values -- the Grouper's first batch -- as long as there is no mask either: the kernel
  // then reads the first bytes of the key buffer where it would read values, and marks no record as carrying one)
  const bool bins_shape = ks.total_bits == 32 && (k_load_bits == 32 || k_load_bits == 64) && !keys->validity && !m_valid &&
                          (values ? BitWidth(values->type_id) == 64 : (!mask && n >= 64));
  bool use_bins = bins_shape && !getenv("AMDARROW_GROUPBY_NO_BINS") &&
                        (reinterpret_cast<uintptr_t>(keys->values) + (k_load_bits / 8) * static_cast<uintptr_t>(keys->offset)) % 16 == 0 &&
                        (values ? (reinterpret_cast<uintptr_t>(values->values) + 8 * static_cast<uintptr_t>(values->offset)) % 16 == 0
                                : reinterpret_cast<uintptr_t>(keys->values) % 16 == 0);
  // Clustered keys (sorted input, time-ordered ids: a tile's 4096 rows hold a few dozen keys) defeat the bins -- a tile's
  // records meet in a few of the 512 bins, dozens each where a bin takes sixteen, and the records that find their bin full
  // overflow the room at the region tops --: an attempt that ends that way is repeated with the staged scatter, which ranks a
  // tile's records by partition whatever their spread, and the shape's hint keeps the next calls off the bins for a while.
  bool bins_overflowed = false;
  const int no_bins_before = shape_known ? hint.no_bins : 0, no_bins_next_before = shape_known ? hint.no_bins_next : 16;
  const bool bins_off_by_hint = use_bins && no_bins_before > 0;
  if (bins_off_by_hint) use_bins = false;
  // (key_lo / key_hi: the selected keys' range in the scatter's key order plus the bias; narrow: the selected values were all int32)
  auto plan_compact = [&](uint64_t key_lo, uint64_t key_hi, bool narrow) -> bool {
    if (!(use_bins && compact_allowed && key_lo <= key_hi && key_lo >= k_bias && key_hi - k_bias <= 0xFFFFFFFFull) ||
        getenv("AMDARROW_GROUPBY_NO_COMPACT"))
      return false;
    int rw = values && !IsFloat(values->type_id) && narrow ? 2 : 3;
    if (const char* env = getenv("AMDARROW_GROUPBY_COMPACT_WORDS")) rw = atoi(env) == 2 && rw == 2 ? 2 : 3;
    int pb = rw == 2 ? 9 : 8;  // 8-byte records: 512 partitions, bins of 64 words; 12-byte records: 256 partitions, bins of 128
    if (const char* env = getenv("AMDARROW_GROUPBY_COMPACT_PARTS")) pb = atoi(env) == 256 ? 8 : atoi(env) == 512 ? 9 : pb;
    const int slot_bits = 21 - pb;
    const uint32_t first = static_cast<uint32_t>(key_lo - k_bias) >> pb;
    const uint32_t span = (static_cast<uint32_t>(key_hi - k_bias) >> pb) - first + 1;
    if (span > (1u << slot_bits)) return false;
    const uint32_t slack = ((1u << slot_bits) - span) / 2;  // the window is centred on the range seen last time
    c_base = first > slack ? first - slack : 0;
    compact_rw = rw;
    compact_pb = pb;
    return true;
  };
  if (hinted && hint.stats) plan_compact(hint.key_lo, hint.key_hi, hint.values_narrow != 0);
  // A call that knows nothing about its data (the first of a shape, every kHintProbeEvery-th, every call of a caller
  // whose data keep changing) used to pay for that with 16-byte records, 0.68-0.78 ms against 0.52.  The optimistic
  // per-workgroup pre-aggregation it starts with IS a sample of the input -- 256 row ranges, up to 2048 distinct keys of
  // each -- so the host now waits for that one kernel (one round trip, ~20 us), and if it gave up without heavy
  // hitters the call goes on with compact records planned from the sample's key range and value width: GUESSED, and
  // checked per row like a hinted call's; a miss starts over on the full path, once.
  bool guess_tried = false, guessed = false;
  // ... and when it gave up WITH heavy hitters (a skewed key column never earns a hint: every call starts here): the hot
  // rows are folded next to the scatter as before, and the scatter of the remainder carries 8-byte compact records when
  // the sample's values are int32 -- same workgroups and regions as the 16-byte bin scatter, so the call goes on in place
  bool hot_compact = false;
  GroupTable t{};
  RankMem r{};
  unsigned long long* gx = nullptr;
  unsigned int* misc = nullptr;  // status words, see RankFirstRows; [64..] one flag per workgroup of the local path
  constexpr int kMaxFlags = 1024;
  int sub_bits = hinted_sub_bits;  // 2 after a partition's LDS table overflowed: four sub-problems, up to ~7 M groups; then 4 (bin scatter: rounds of the aggregate)
  bool fa_val_wide = false;  // the last attempt's scatter reported whether the values were all int32 (status[10])
  for (;; --path) {
    fa_val_wide = false;
    const bool fast = path > 0;
    // (sub-problems: the partition attempt of path 2 -- and of path 3 where the aggregate does them as rounds over one scatter:
    // a table overflow with heavy hitters taken out goes through the heavy-hitter pass again, not through a scatter that
    // would send a third of the rows to one region)
    const bool rounds_on_path3 = path == 3 && sub_bits > 0 && use_bins && !getenv("AMDARROW_GROUPBY_SUB_PASSES");
    const int nsub = (fast && (path == 2 || rounds_on_path3)) ? 1 << sub_bits : 1;
    const uint64_t T = fast ? uint64_t(nsub) * kP * kSlots
                            : NextPow2(2 * static_cast<uint64_t>(group_capacity < n ? group_capacity : n));
    Carver c;
    TableMem tm;
    CarveTable(&c, T, false, &tm);
    size_t roff[4];
    CarveRank(&c, n, roff);
    const size_t off_gx = c.Take(32 * static_cast<size_t>(group_capacity));  // packed group records
    const size_t off_misc = c.Take(256 + 4 * kMaxFlags);
    // partition scratch
    int nblocks = 0, nblocks_local = 0;
    int64_t rows_per_block = 0, rows_per_block_local = 0, cap = 0, hist_n = 0;
    bool compact = false;
    if (path >= 2) {
      // the specialised scatter runs two 512-thread workgroups per CU on 3584-row tiles
      const int64_t tile = use_bins ? kBThreads * 4 : headline_shape ? kFRows : kDRows;
      nblocks = static_cast<int>(std::min<int64_t>(ctx->num_cus * (headline_shape && !use_bins ? 1024 / kFThreads : 1), (n + tile - 1) / tile));
      rows_per_block = (((n + nblocks - 1) / nblocks + tile - 1) / tile) * tile;
      nblocks_local = static_cast<int>(std::min<int64_t>(ctx->num_cus, (n + kDRows - 1) / kDRows));
      rows_per_block_local = (((n + nblocks_local - 1) / nblocks_local + kDRows - 1) / kDRows) * kDRows;
      // (sub-problems behind the bin scatter: ONE scatter of every selected row, the aggregate folds each partition in nsub rounds)
      const bool sub_rounds_sizing = nsub > 1 && use_bins && !getenv("AMDARROW_GROUPBY_SUB_PASSES");
      const double mean = static_cast<double>(rows_per_block) / kP / (sub_rounds_sizing ? 1 : nsub);
      // room for a key of up to ~0.25 % of the rows next to the uniform share (hotter keys are taken out before the scatter)
      cap = (static_cast<int64_t>(mean * 2.0 + 8.0 * std::sqrt(mean) + 32.0) + 7) & ~int64_t(7);  // whole 128-byte lines
      if (compact_rw != 0 && hinted && path == 2 && nsub == 1) {
        // a compact record carries its row as an offset into the workgroup's row range (19 / 18 bits): more workgroups
        // (whole multiples of the CUs) for large inputs
        const int64_t limit = int64_t(1) << (10 + compact_pb);
        const int64_t rounds = (n + ctx->num_cus * limit - 1) / (ctx->num_cus * limit);
        const int64_t nb = std::min<int64_t>(ctx->num_cus * rounds, (n + tile - 1) / tile);
        const int64_t rpb = (((n + nb - 1) / nb + tile - 1) / tile) * tile;
        if (nb <= kMaxFlags && rpb <= limit) {
          compact = true;
          nblocks = static_cast<int>(nb);
          rows_per_block = rpb;
          const double cmean = static_cast<double>(rpb) / (1 << compact_pb);
          cap = (static_cast<int64_t>(cmean * 2.0 + 8.0 * std::sqrt(cmean) + 32.0) + 7) & ~int64_t(7);
          cap = ((cap * compact_rw + 3) / 4 + 7) & ~int64_t(7);  // that number of records, in 16-byte units
        }
      }
      if (bias_from_hint && !compact) return start_over_measured();
      if (use_bins) cap += kCSpillUnits;  // room at the top of every region for the records that found their bin full
      hist_n = int64_t(kP) * (nblocks + (use_bins ? 1 : 0));  // bins: one more (always empty) segment per partition, where the spill list was
    } else if (path == 1 && wide_onepass) {
      const int64_t tile = kBThreads * 4;
      nblocks = static_cast<int>(std::min<int64_t>(ctx->num_cus, (n + tile - 1) / tile));
      rows_per_block = (((n + nblocks - 1) / nblocks + tile - 1) / tile) * tile;
      // (every row may be selected: the mask's density is not known here)
      const double mean = static_cast<double>(rows_per_block) / kP * (m_data ? 0.75 : 1.0);
      cap = static_cast<int64_t>(mean * 1.5 + 8.0 * std::sqrt(mean) + 32.0);       // records
      cap = (((cap * 5 + 3) / 4 + 7) & ~int64_t(7)) + 2 * kWSpillRecs;             // 16-byte units, whole lines, room for the spilled records
      hist_n = int64_t(kP) * (nblocks + 1);
    } else if (path == 1) {
      nblocks = static_cast<int>(std::min<int64_t>(ctx->num_cus, (n + kPartRows - 1) / kPartRows));
      rows_per_block = (((n + nblocks - 1) / nblocks + kPartRows - 1) / kPartRows) * kPartRows;
      hist_n = int64_t(kP) * nblocks + 1;
    }
    const bool w1 = path == 1 && wide_onepass;
    const int64_t scan_nb = (hist_n + 2047) / 2048;
    const size_t off_hist = c.Take(4 * hist_n);
    const size_t off_sums = c.Take(4 * scan_nb);
    const size_t off_sumoff = c.Take(8 * (scan_nb + 1));
    const int nseg = nblocks + (path >= 2 && use_bins ? 1 : 0);  // segments per partition the aggregate walks
    const size_t rec_bytes = path >= 2 ? std::max(16 * static_cast<size_t>(cap) * (compact ? size_t(1) << compact_pb : size_t(kP)) * nseg, LocalStageBytes(nblocks_local))
                           : w1 ? 16 * static_cast<size_t>(cap) * kP * (nblocks + 1)
                           : path == 1 ? 16 * static_cast<size_t>(n) : 0;
    const size_t off_rec = c.Take(rec_bytes);
    const size_t off_hi = c.Take(path == 1 && wide && !w1 ? 4 * static_cast<size_t>(n) : 0);
    // heavy-hitter scratch (path 3 only)
    const int nwg_hot = path == 3 ? nblocks_local : 0;
    const size_t off_hot_cand = c.Take(4 * size_t(nwg_hot) * kHotPerWg);
    const size_t off_hot_candn = c.Take(4 * size_t(kMaxFlags));
    const size_t off_hot_tags = c.Take(4 * size_t(kHotCap));
    const size_t off_hot_count = c.Take(256);
    const size_t off_hot_psum = c.Take(8 * size_t(nwg_hot) * (kHotCap + 1));
    const size_t off_hot_pcnt = c.Take(4 * size_t(nwg_hot) * (kHotCap + 1));
    const size_t off_hot_pfirst = c.Take(4 * size_t(nwg_hot) * (kHotCap + 1));
    const size_t off_hot_mask = c.Take(path == 3 ? static_cast<size_t>(n / 8 + 64) : 0);
    const size_t off_hot_stats = c.Take(16 * size_t(kMaxFlags));
    void* base;
    AA_TRY(Scratch(ctx, c.off, &base));
    char* b = static_cast<char*>(base);
    HotMem hot{};
    hot.cand = reinterpret_cast<uint32_t*>(b + off_hot_cand);
    hot.cand_n = reinterpret_cast<unsigned int*>(b + off_hot_candn);
    hot.tags = reinterpret_cast<uint32_t*>(b + off_hot_tags);
    hot.count = reinterpret_cast<unsigned int*>(b + off_hot_count);
    hot.part_sum = reinterpret_cast<unsigned long long*>(b + off_hot_psum);
    hot.part_cnt = reinterpret_cast<unsigned int*>(b + off_hot_pcnt);
    hot.part_first = reinterpret_cast<uint32_t*>(b + off_hot_pfirst);
    hot.alt_mask = reinterpret_cast<uint8_t*>(b + off_hot_mask);
    hot.wg_stats = reinterpret_cast<uint32_t*>(b + off_hot_stats);
    t = BindTable(b, tm, T, false);
    r = BindRank(b, roff, n);
    gx = reinterpret_cast<unsigned long long*>(b + off_gx);
    misc = reinterpret_cast<unsigned int*>(b + off_misc);
    unsigned int* wg_flags = misc + 64;
    if (nblocks > kMaxFlags || nblocks_local > kMaxFlags) return SetError(AMDARROW_INVALID, "group-by: more workgroups than flag words");
    hipLaunchKernelGGL(k_prepare, dim3(GridFor(r.nwords, kThreads, ctx->num_cus * 4)), dim3(kThreads), 0, ctx->stream, r.bitmap,
                       r.nwords, misc, wg_flags, nblocks_local > nblocks ? nblocks_local : nblocks,
                       path == 3 ? hot.cand_n : static_cast<unsigned int*>(nullptr),
                       (path >= 2 && use_bins) || w1 ? reinterpret_cast<uint32_t*>(b + off_hist) : static_cast<uint32_t*>(nullptr), w1 ? nblocks + 1 : nseg,
                       (path >= 2 && use_bins) || w1 ? misc + 8 : static_cast<unsigned int*>(nullptr));
    AA_HIP(hipGetLastError());

    if (fast) {
      auto* hist = reinterpret_cast<uint32_t*>(b + off_hist);
      auto* sums = reinterpret_cast<uint32_t*>(b + off_sums);
      auto* sumoff = reinterpret_cast<int64_t*>(b + off_sumoff);
      auto* records = reinterpret_cast<uint4*>(b + off_rec);
      FastArgs fa{};
      fa.k_data = keys->values; fa.k_valid = keys->validity; fa.k_off = keys->offset; fa.k_bits = ks.total_bits;
      fa.k_load_bits = k_load_bits; fa.k_bias = k_bias;
      fa.k_flip = keys->type_id == AMDARROW_INT32 ? 0x80000000u : 0u;  // (a range-compressed column: x = key - min >= 0 as it is)
      fa.key_minmax = misc + 8;  // (status words are misc[0..11]; the workgroup flags start at misc + 64)
      fa.val_wide = path >= 2 && use_bins && values && !IsFloat(values->type_id) ? misc + 10 : nullptr;
      fa.c_base = c_base;
      fa_val_wide = fa.val_wide != nullptr;
      fa.v = v; fa.has_values = values ? 1 : 0;
      if (!values && use_bins) fa.v = ValueCol{keys->values, nullptr, 0, AMDARROW_INT64};  // (never used as values: see BinLoadTile)
      fa.m_data = m_data; fa.m_valid = m_valid; fa.m_off = m_off;
      fa.n = n; fa.nblocks = nblocks; fa.rows_per_block = rows_per_block;
      const int acc_double = values && IsFloat(values->type_id) ? 1 : 0;
      if (path >= 2) {
        const bool M = m_data != nullptr, V = values && values->validity != nullptr;
        const unsigned int* run_if = nullptr;
        if (path == 3) {
          // optimistic local pre-aggregation; everything after it is gated on misc[1] on the device
          char* sb = reinterpret_cast<char*>(records);
          FastArgs fl = fa;
          fl.nblocks = nblocks_local; fl.rows_per_block = rows_per_block_local;
          const size_t ne = static_cast<size_t>(kStageSlots) * nblocks_local;
          LocalStage st;
          st.sum = reinterpret_cast<unsigned long long*>(sb);
          st.tag = reinterpret_cast<unsigned int*>(sb + 8 * ne);
          st.first = reinterpret_cast<uint32_t*>(sb + 12 * ne);
          st.cnt = reinterpret_cast<unsigned int*>(sb + 16 * ne);
          unsigned int* abort_flag = misc + 1;  // written by k_clear_table_unless (OR of the workgroup flags)
#define AA_LOCAL(MM, VV) hipLaunchKernelGGL((k_local_aggregate_fast<MM, VV>), dim3(nblocks_local), dim3(kDThreads), 0, ctx->stream, fl, acc_double, st, wg_flags, hot)
#define AA_LVEC(MM, VV) hipLaunchKernelGGL((k_local_aggregate_vec<MM, VV>), dim3(nblocks_local), dim3(kBThreads), 0, ctx->stream, fl, acc_double, st, wg_flags, hot)
          if (!headline_shape) hipLaunchKernelGGL(k_local_aggregate, dim3(nblocks_local), dim3(kDThreads), 0, ctx->stream, fl, acc_double, st, wg_flags, hot);
          else if (use_bins && k_load_bits == 32 && !getenv("AMDARROW_GROUPBY_OLD_LOCAL")) {
            if (M && V) AA_LVEC(true, true); else if (M) AA_LVEC(true, false); else if (V) AA_LVEC(false, true); else AA_LVEC(false, false);
          }
          else if (M && V) AA_LOCAL(true, true); else if (M) AA_LOCAL(true, false); else if (V) AA_LOCAL(false, true); else AA_LOCAL(false, false);
#undef AA_LOCAL
#undef AA_LVEC
          AA_HIP(hipGetLastError());
          const int64_t nslots = static_cast<int64_t>(T) + 2;
          // (the guess needs the vectorised pre-aggregation, which reports what it saw, and a shape the compact scatter takes)
          const bool probe = !local_only && !guess_tried && headline_shape && use_bins && k_load_bits == 32 && !getenv("AMDARROW_GROUPBY_OLD_LOCAL") &&
                             compact_allowed && sub_bits == 0 && n <= (int64_t(1) << 28) && !getenv("AMDARROW_GROUPBY_PATH") &&
                             !getenv("AMDARROW_GROUPBY_NO_COMPACT") && !getenv("AMDARROW_GROUPBY_NO_GUESS");
          volatile unsigned int* probe_host = reinterpret_cast<volatile unsigned int*>(static_cast<char*>(ctx->pinned) + kStatusOffset + 128);
          if (probe) probe_host[0] = 0;
          hipLaunchKernelGGL(k_clear_table_unless, dim3(GridFor(nslots, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0,
                             ctx->stream, t, nslots, wg_flags, nblocks_local, abort_flag, hot, misc + 5,
                             local_only ? misc : static_cast<unsigned int*>(nullptr),
                             probe ? reinterpret_cast<unsigned int*>(static_cast<char*>(ctx->pinned_dev) + kStatusOffset + 128) : static_cast<unsigned int*>(nullptr));
          if (probe) {
            AA_HIP(hipGetLastError());
            AA_HIP(hipEventRecord(ctx->status_event, ctx->stream));
            AA_HIP(hipEventSynchronize(ctx->status_event));
            guess_tried = true;
            if (probe_host[0] == 1 && probe_host[1] == 0) {
              local_only = true;  // every row range held its groups: nothing but the merge is queued behind (the table is cleared)
            } else if (probe_host[0] == 1 && probe_host[2] == 0 &&
                       plan_compact(uint64_t(probe_host[3]) + k_bias, uint64_t(probe_host[4]) + k_bias, probe_host[5] == 0)) {
              guessed = true;
              hinted = true;
              path = 3;  // -> 2 after the loop's decrement: the partition path on compact records, from the top
              continue;
            } else if (probe_host[0] == 1 && probe_host[1] != 0 && probe_host[2] != 0 && ctx->gb_hot_guess_off > 0) {
              --ctx->gb_hot_guess_off;  // (such a guess missed not long ago: the 16-byte flow as queued)
            } else if (probe_host[0] == 1 && probe_host[1] != 0 && probe_host[2] != 0 && nsub == 1 && m_valid == nullptr &&
                       !getenv("AMDARROW_GROUPBY_OLD_HOT") &&
                       plan_compact(uint64_t(probe_host[3]) + k_bias, uint64_t(probe_host[4]) + k_bias, probe_host[5] == 0)) {
              if (compact_rw == 2 && compact_pb == 9 && rows_per_block <= (int64_t(1) << 19)) {
                hot_compact = true;  // (8-byte records: the geometry of this iteration's bin scatter)
                fa.c_base = c_base;
              } else {
                compact_rw = 0;  // (12-byte records run on 256 partitions and twice the workgroups: the 16-byte flow as queued)
              }
            }
          }
          if (headline_shape && !getenv("AMDARROW_GROUPBY_OLD_MERGE"))
            hipLaunchKernelGGL(k_local_merge_slots, dim3(kSlots / kMergeSlots), dim3(kMergeThreads), 0, ctx->stream, st,
                               nblocks_local, ks.total_bits, acc_double, t, abort_flag, misc);
          else
          hipLaunchKernelGGL(k_local_merge, dim3(GridFor(static_cast<int64_t>(ne), kThreads, ctx->num_cus * 8)), dim3(kThreads),
                             0, ctx->stream, st, static_cast<int64_t>(ne), ks.total_bits, acc_double, t, abort_flag, misc);
          AA_HIP(hipGetLastError());
          // heavy hitters found by the workgroups that gave up: fold their rows here, hand the
          // scatter a mask without them (no-op when status[5] is 0)
          if (!local_only) {
            if (headline_shape && use_bins && k_load_bits == 32 && !getenv("AMDARROW_GROUPBY_OLD_LOCAL") && !getenv("AMDARROW_GROUPBY_OLD_HOT")) {
#define AA_HOTV(MM, VV) hipLaunchKernelGGL((k_hot_aggregate_vec<MM, VV>), dim3(nblocks_local), dim3(kBThreads), 0, ctx->stream, fl, acc_double, hot, misc + 5)
              if (M && V) AA_HOTV(true, true); else if (M) AA_HOTV(true, false); else if (V) AA_HOTV(false, true); else AA_HOTV(false, false);
#undef AA_HOTV
            } else {
              hipLaunchKernelGGL(k_hot_aggregate, dim3(nblocks_local), dim3(kDThreads), 0, ctx->stream, fl, acc_double, hot, misc + 5);
            }
          }
          AA_HIP(hipGetLastError());
          run_if = abort_flag;
        }
        const unsigned int* use_hot = path == 3 ? misc + 5 : nullptr;
        const bool sub_rounds = nsub > 1 && use_bins && !getenv("AMDARROW_GROUPBY_SUB_PASSES");
        for (int sub = 0; sub < (local_only && path == 3 ? 0 : sub_rounds ? 1 : nsub); ++sub) {  // one pass per sub-problem (nsub == 1 unless a table overflowed before)
        fa.sub_bits = nsub > 1 && !sub_rounds ? sub_bits : 0; fa.sub_id = static_cast<uint32_t>(sub);
        FastArgs fh = fa;  // the same scatter on the mask k_hot_aggregate wrote
        fh.m_data = hot.alt_mask; fh.m_valid = nullptr; fh.m_off = 0;
        if (hot_compact) {  // the remainder of a skewed input (the mask k_hot_aggregate_vec wrote), on 8-byte records
          if (V) hipLaunchKernelGGL((k_part_scatter_compact<true, true, false, 2, 9>), dim3(nblocks), dim3(kBThreads), 0, ctx->stream, fh, cap, records, hist, misc);
          else hipLaunchKernelGGL((k_part_scatter_compact<true, false, false, 2, 9>), dim3(nblocks), dim3(kBThreads), 0, ctx->stream, fh, cap, records, hist, misc);
        } else if (compact) {
#define AA_CBINS4(MM, VV, KK, RR, PP) hipLaunchKernelGGL((k_part_scatter_compact<MM, VV, KK, RR, PP>), dim3(nblocks), dim3(kBThreads), 0, ctx->stream, fa, cap, records, hist, misc)
#define AA_CBINS3(MM, VV, KK) do { if (compact_rw == 2 && compact_pb == 9) AA_CBINS4(MM, VV, KK, 2, 9); else if (compact_rw == 2) AA_CBINS4(MM, VV, KK, 2, 8); \
          else if (compact_pb == 9) AA_CBINS4(MM, VV, KK, 3, 9); else AA_CBINS4(MM, VV, KK, 3, 8); } while (0)
#define AA_CBINS(MM, VV) do { if (k_load_bits == 64) AA_CBINS3(MM, VV, true); else AA_CBINS3(MM, VV, false); } while (0)
          if (M && V) AA_CBINS(true, true); else if (M) AA_CBINS(true, false); else if (V) AA_CBINS(false, true); else AA_CBINS(false, false);
#undef AA_CBINS
#undef AA_CBINS4
#undef AA_CBINS3
        } else if (use_bins) {
#define AA_BINS(MM, VV, ARGS, RUN, SKIP) do { if (k_load_bits == 64) hipLaunchKernelGGL((k_part_scatter_bins<MM, VV, true>), dim3(nblocks), dim3(kBThreads), 0, ctx->stream, ARGS, cap, records, hist, misc, RUN, SKIP); \
          else hipLaunchKernelGGL((k_part_scatter_bins<MM, VV, false>), dim3(nblocks), dim3(kBThreads), 0, ctx->stream, ARGS, cap, records, hist, misc, RUN, SKIP); } while (0)
          if (M && V) AA_BINS(true, true, fa, run_if, use_hot); else if (M) AA_BINS(true, false, fa, run_if, use_hot);
          else if (V) AA_BINS(false, true, fa, run_if, use_hot); else AA_BINS(false, false, fa, run_if, use_hot);
          if (use_hot) { if (V) AA_BINS(true, true, fh, use_hot, nullptr); else AA_BINS(true, false, fh, use_hot, nullptr); }
#undef AA_BINS
        } else if (headline_shape) {
#define AA_FAST(MM, VV, ARGS, RUN, SKIP) hipLaunchKernelGGL((k_part_scatter_fast<MM, VV>), dim3(nblocks), dim3(kFThreads), 0, ctx->stream, ARGS, cap, records, hist, misc, RUN, SKIP)
          if (M && V) AA_FAST(true, true, fa, run_if, use_hot); else if (M) AA_FAST(true, false, fa, run_if, use_hot);
          else if (V) AA_FAST(false, true, fa, run_if, use_hot); else AA_FAST(false, false, fa, run_if, use_hot);
          if (use_hot) { if (V) AA_FAST(true, true, fh, use_hot, nullptr); else AA_FAST(true, false, fh, use_hot, nullptr); }
#undef AA_FAST
        } else {
          hipLaunchKernelGGL(k_part_scatter_direct, dim3(nblocks), dim3(kDThreads), 0, ctx->stream, fa, cap, records, hist,
                             misc, run_if, use_hot);
          if (use_hot)
            hipLaunchKernelGGL(k_part_scatter_direct, dim3(nblocks), dim3(kDThreads), 0, ctx->stream, fh, cap, records, hist,
                               misc, use_hot, static_cast<const unsigned int*>(nullptr));
        }
        AA_HIP(hipGetLastError());
        if (compact || hot_compact) {
#define AA_CAGG(RR, PP) hipLaunchKernelGGL((k_part_aggregate_compact<RR, PP>), dim3(1 << PP), dim3(kPartThreads), 0, ctx->stream, records, hist, nblocks, cap, \
                               acc_double, t, r.bitmap, misc, static_cast<int64_t>(T), c_base, fa.k_flip, static_cast<uint32_t>(rows_per_block))
          if (compact_rw == 2 && compact_pb == 9) AA_CAGG(2, 9); else if (compact_rw == 2) AA_CAGG(2, 8);
          else if (compact_pb == 9) AA_CAGG(3, 9); else AA_CAGG(3, 8);
#undef AA_CAGG
        } else if (use_bins)
          hipLaunchKernelGGL(k_part_aggregate_piped, dim3(kP, sub_rounds ? nsub : 1), dim3(kPartThreads), 0, ctx->stream, records, hist, nseg,
                             cap, ks.total_bits, acc_double, t, r.bitmap, misc, run_if, int64_t(sub) * kP * kSlots,
                             sub == 0 ? static_cast<int64_t>(T) : int64_t(0), fa.key_minmax, fa.k_flip, use_hot, sub_rounds ? sub_bits : 0);
        else
        hipLaunchKernelGGL(k_part_aggregate_direct, dim3(kP), dim3(kPartThreads), 0, ctx->stream, records, hist, nseg,
                           cap, ks.total_bits, acc_double, t, r.bitmap, misc, run_if, int64_t(sub) * kP * kSlots,
                           sub == 0 ? static_cast<int64_t>(T) : int64_t(0));
        AA_HIP(hipGetLastError());
        }  // sub-problems
      } else if (w1) {
        const bool M = m_data != nullptr, V = values && values->validity != nullptr;
        if (!values) fa.v = ValueCol{keys->values, nullptr, 0, AMDARROW_INT64};  // (never used as values: see BinLoadTile)
#define AA_WBINS(MM, VV) hipLaunchKernelGGL((k_part_scatter_compact<MM, VV, true, 5, 9>), dim3(nblocks), dim3(kBThreads), 0, ctx->stream, fa, cap, records, hist, misc)
        if (M && V) AA_WBINS(true, true); else if (M) AA_WBINS(true, false); else if (V) AA_WBINS(false, true); else AA_WBINS(false, false);
#undef AA_WBINS
        hipLaunchKernelGGL(k_part_aggregate_wide, dim3(kP), dim3(kPartThreads), 0, ctx->stream, records, hist, nblocks, cap, acc_double, t, misc,
                           static_cast<uint32_t>(rows_per_block));
        AA_HIP(hipGetLastError());
      } else {
        AA_HIP(hipMemsetAsync(hist + hist_n - 1, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_part_hist, dim3(nblocks), dim3(kPartThreads), 0, ctx->stream, fa, hist);
        AA_HIP(hipGetLastError());
        AA_TRY(ScanU32InPlace(ctx, hist, hist_n, sums, sumoff));
        auto* hi_words = reinterpret_cast<uint32_t*>(b + off_hi);
        if (wide) {
          hipLaunchKernelGGL(k_part_scatter<true>, dim3(nblocks), dim3(kPartThreads), 0, ctx->stream, fa, hist, records, hi_words);
          hipLaunchKernelGGL(k_part_aggregate<true>, dim3(kP), dim3(kPartThreads), 0, ctx->stream, records, hi_words, hist, nblocks,
                             ks.total_bits, acc_double, t, misc);
        } else {
          hipLaunchKernelGGL(k_part_scatter<false>, dim3(nblocks), dim3(kPartThreads), 0, ctx->stream, fa, hist, records, hi_words);
          hipLaunchKernelGGL(k_part_aggregate<false>, dim3(kP), dim3(kPartThreads), 0, ctx->stream, records, hi_words, hist, nblocks,
                             ks.total_bits, acc_double, t, misc);
        }
        AA_HIP(hipGetLastError());
      }
    } else {
      AA_TRY(ClearTable(ctx, t, false));
      hipLaunchKernelGGL(k_groupby_build, dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                         ks_generic, v, values ? 1 : 0, m_data, m_valid, m_off, n, t, misc);
      AA_HIP(hipGetLastError());
    }
    if (null_inject) {  // the null group (k_null_split) joins the finished table
      const KeySpec& kinj = fast ? ks : ks_generic;  // (how this attempt's table spells its keys: the generic path takes a 64-bit column as it is)
      hipLaunchKernelGGL(k_inject_null, dim3(1), dim3(64), 0, ctx->stream, t, r.bitmap, null_inject, Fmix64(uint64_t(1) << (kinj.total_bits & 63)), misc,
                         kinj.wide_null_slot ? 1 : 0);
      AA_HIP(hipGetLastError());
    }
    int64_t G = 0;
    const HotInject inject{hot, nblocks_local, ks.total_bits, values && IsFloat(values->type_id) ? 1 : 0};
    AA_TRY(RankFirstRows(ctx, t, 0, r, &G, /*mark_mode=*/path == 3 ? 2 : path == 2 ? 1 : 3, misc,
                          path == 3 && !local_only ? &inject : nullptr));
    // Emit and decode are queued BEFORE the group count is known on the host (they take it from
    // the status words and are bounded by group_capacity), so the one host round trip of the
    // call overlaps with them instead of sitting between two kernels.
    EmitArgs e{};
    e.t = t; e.nslots = static_cast<int64_t>(t.mask) + 3; e.bitmap = r.bitmap; e.prefix = r.prefix;
    e.capacity = group_capacity; e.id_base = 0; e.only_unassigned = 0;
    e.packed = reinterpret_cast<ulonglong4*>(gx); e.wide_null_id = misc + 4; e.skip_if = misc;
    hipLaunchKernelGGL(k_emit_groups, dim3(GridFor(e.nslots, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, e);
    AA_HIP(hipGetLastError());
    const int64_t bound = group_capacity < n ? group_capacity : n;
    const KeySpec& ku = fast ? ks : ks_generic;  // how x was built on this path
    if (packed_out) *packed_out = e.packed;  // {key word, sum, count, first row} per group id, in the context's scratch
    if (bias_out) *bias_out = fast ? k_bias : 0;  // (a range-compressed 64-bit key column: key = key word + bias)
    if (out_keys)
    hipLaunchKernelGGL(k_unpack_groups, dim3(GridFor(bound, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                       e.packed, bound, ku.col[0].bits, ks_generic.col[0].bits, fast ? k_bias : 0ull,
                       ku.wide_null_slot ? -1 : ku.total_bits, static_cast<unsigned long long*>(out_sums), reinterpret_cast<long long*>(out_counts), out_first_rows,
                       out_keys, reinterpret_cast<uint64_t*>(out_key_validity), misc);
    AA_HIP(hipGetLastError());
    unsigned int status[12] = {0};
    AA_TRY(AwaitStatus(ctx, status));
#ifdef SC_PHASES
    if (getenv("AMDARROW_PHASES")) {
      unsigned long long ph[16];
      AA_HIP(hipStreamSynchronize(ctx->stream));
      AA_HIP(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase_cycles), sizeof(ph)));
      unsigned long long z[16] = {0};
      AA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)));
      fprintf(stderr, "phases(kcycles/wg):");
      for (int i = 0; i < 9; ++i) fprintf(stderr, " %d:%.1f", i, double(ph[i]) / nblocks / 1e3);
      fprintf(stderr, "\n");
    }
#endif
    memcpy(&G, status + 6, 8);
    const unsigned int overflow = status[0];
    if (getenv("AMDARROW_GROUPBY_TRACE"))  // (diagnostics: one line per attempt of a call)
      fprintf(stderr, "group-by attempt: path %d sub_bits %d hinted %d guessed %d compact %d(%d words) hot_compact %d local_only %d wide_onepass %d -> overflow %u gave_up %u hot %u groups %lld key range [%u, %u]\n",
              path, sub_bits, int(hinted), int(guessed), int(compact), compact_rw, int(hot_compact), int(local_only), int(w1), overflow, status[1], status[5],
              static_cast<long long>(G), status[8], status[9]);
    if (local_only && path == 3 && status[1] != 0) {  // the pre-aggregation gave up after all: everything, from the start
      local_only = false;
      hint.n = -1;
      path = 4;  // -> 3 after the loop's decrement
      continue;
    }
    if (compact && (overflow & kOverflowCompact)) {
      // a miss weighs two, a compact call that goes through takes one off: a shape that keeps missing (data sets of one
      // shape that alternate, say) reaches four and its next 16 calls stay on 16-byte records
      carry_misses += 2;
      if (carry_misses >= 4) { carry_misses = 0; carry_off = 16; }
    }
    if (overflow && bias_from_hint) {
      ctx->gb_carry_misses = carry_misses; ctx->gb_carry_off = carry_off;  // (start_over_measured forgets the shape's hints; the retry picks the back-off up here)
      return start_over_measured();
    }
    if (hot_compact && (overflow & kOverflowCompact)) {  // a row outside the sampled key range / a wide value: the same call on 16-byte records
      hot_compact = false;
      compact_rw = 0;
      ctx->gb_hot_guess_off = 16;  // the next calls on skewed input do not guess
      path = 4;  // -> 3 after the loop's decrement (no second guess: guess_tried)
      continue;
    }
    if (overflow && fast && hinted) {  // the hint (or the guess) was wrong for this input: forget it, start over with the full path 3
      guessed = false;
      guess_tried = true;  // (one compact attempt per call: what broke a hinted call's assumption -- a few far keys, one wide value -- hides from a sample too)
      hinted = false;
      hint.n = -1;
      sub_bits = 0;
      path = 4;  // -> 3 after the loop's decrement
      continue;
    }
    if (overflow && w1) {  // a region or a table too small for these keys: the exact two-pass variant
      wide_onepass = false;
      path += 1;  // (the loop's decrement brings it back)
      continue;
    }
    if ((overflow & kOverflowRegion) && fast && path >= 2 && use_bins) {  // clustered keys? the same path once more, on the staged scatter
      use_bins = false;
      bins_overflowed = true;
      hot_compact = false;
      path += 1;  // (the loop's decrement brings it back)
      continue;
    }
    if (overflow && fast) {  // capacity assumption failed: next (more general) path
      if (path == 3) path = 2;  // the partition attempt was part of path 3
      if (overflow == kOverflowTable) {
        // the two-pass variant has the same LDS tables.  More groups than 512 partitions hold: once
        // more with four sub-problems (four passes over the input, 2048 partitions in effect),
        // then the generic global-atomic table
        const bool had_hot = status[5] != 0 && use_bins && !getenv("AMDARROW_GROUPBY_SUB_PASSES");  // heavy hitters were taken out: path 3 again (-> 3 after the loop's decrement)
        if (sub_bits == 0) { sub_bits = 2; path = had_hot ? 4 : 3; }  // -> 2 after the loop's decrement
        else if (sub_bits >= 2 && sub_bits < 4 && use_bins && !getenv("AMDARROW_GROUPBY_SUB_PASSES")) { ++sub_bits; path = had_hot ? 4 : 3; }  // eight, then sixteen rounds over the records: up to ~14 M / ~28 M groups
        else path = 1;
      }
      continue;
    }
    if (overflow || G > group_capacity)
      return SetError(AMDARROW_CAPACITY_ERROR,
                      "group-by: more than group_capacity = " + std::to_string(group_capacity) + " groups");
    if (path == 3 && !hinted) {
      // status[1]: the local pre-aggregation gave up; status[5]: heavy hitters were taken out
      if (status[1] != 0 && status[5] == 0 && sub_bits == 0) {
        shape.uses = hint.n == shape.n ? hint.uses : 0;
        // what the bin scatter saw: the range of the selected keys, and whether the values were all int32
        shape.stats = use_bins && status[8] <= status[9] ? 1 : 0;
        shape.key_lo = status[8] + k_bias; shape.key_hi = status[9] + k_bias;
        shape.values_narrow = values && !IsFloat(values->type_id) && status[10] == 0 ? 1 : 0;
        shape.k_bias = k_bias;
        shape.last_used = hint.last_used;
        hint = shape;
      }
      else if (status[1] == 0 && !overflow && sub_bits == 0) {  // the pre-aggregation held every group
        shape.uses = 0; shape.stats = 0; shape.local_ok = 1;
        shape.last_used = hint.last_used;
        hint = shape;
      }
      else hint.n = -1;
    }
    if (path == 2 && !hinted && sub_bits > 0 && use_bins && status[5] == 0) {  // the next calls of this shape start with as many rounds
      shape.uses = 0; shape.stats = 0; shape.sub_bits = sub_bits;
      shape.last_used = hint.last_used;
      hint = shape;
    }
    if (guessed) {  // the guess held: the next call of this shape takes over what this one's scatter measured, like after any unhinted call
      shape.uses = shape_known ? hint.uses : 0;
      shape.stats = status[8] <= status[9] ? 1 : 0;
      shape.key_lo = status[8] + k_bias; shape.key_hi = status[9] + k_bias;
      shape.values_narrow = compact_rw == 2 ? 1 : (values && !IsFloat(values->type_id) && fa_val_wide && status[10] == 0 ? 1 : 0);
      shape.k_bias = k_bias;
      shape.last_used = hint.last_used;
      hint = shape;
    }
    // a hinted call that found few groups: the local attempt it skipped would have held them all -- next call, try it again
    if (hinted && G <= static_cast<int64_t>(kLocalLimit)) hint.n = -1;
    // a hinted call's scatter measured the same things: the next call's records follow the data
    if (hinted && use_bins && hint.n == shape.n && status[8] <= status[9]) {
      if (!compact) { hint.stats = 1; hint.key_lo = status[8] + k_bias; hint.key_hi = status[9] + k_bias; hint.k_bias = k_bias; }
      if (fa_val_wide && !(compact && compact_rw == 2)) hint.values_narrow = status[10] == 0 ? 1 : 0;
    }
    if (hint.n == shape.n) {  // (how long the shape stays off the bins: sixteen calls after an overflow, twice as many after the next, ...; then the bins get another try)
      const int next_before = shape_known ? no_bins_next_before : 16;
      if (bins_overflowed) {
        hint.no_bins = next_before;
        hint.no_bins_next = next_before < 1024 ? 2 * next_before : 1024;
      } else if (bins_off_by_hint) {
        hint.no_bins = no_bins_before - 1;
        hint.no_bins_next = next_before;
      } else {
        hint.no_bins = no_bins_before;
        hint.no_bins_next = use_bins ? 16 : next_before;  // (the bins held this call)
      }
    }
    if (hint.n == shape.n) {  // the back-off state stays with the shape; a compact call that went through ends a run of misses
      hint.compact_misses = compact && carry_misses > 0 ? carry_misses - 1 : carry_misses;
      hint.compact_off = carry_off;
    }
    if (wide && path == 1) {  // the next calls of this shape skip the min / max pass
      if (hint.n != shape.n) { shape.last_used = hint.last_used; hint = shape; }
      hint.wide_keys = 1;
    } else if (hint.n == shape.n) {
      hint.wide_keys = 0;
    }
    *out_num_groups = G;
    ctx->gb_last_path = path == 3 && status[1] != 0 ? 2 : path;
    ctx->gb_last_words = ctx->gb_last_path == 0 || ctx->gb_last_path == 3 ? 0 : w1 ? 5 : (compact || hot_compact) ? compact_rw : 4;
    break;
  }
  // no trailing synchronise: *out_num_groups is final, the output arrays are stream-ordered
  return AMDARROW_OK;
}

int amdarrow_groupby_last_path(amdarrow_ctx* ctx, int* path, int* record_words) {
  if (!ctx) return SetError(AMDARROW_INVALID, "null context");
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (path) *path = ctx->gb_last_path;
  if (record_words) *record_words = ctx->gb_last_words;
  return AMDARROW_OK;
}

static int FilterGroupByImplCall(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                                 const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                                 void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                                 int64_t* out_num_groups, bool locked) {
  return FilterGroupByImpl(ctx, values, keys, mask, group_capacity, out_sums, out_counts, out_keys, out_key_validity,
                           out_first_rows, out_num_groups, locked, nullptr);
}

// The packed key column: one uint32 per row (PackedKey).  Four rows per thread and step; the raw values of every
// column are requested before any is used (the width switch is uniform: one branch per column, four loads inside).
// Every column 32 bits wide, no validity, every slice on a 16-byte boundary: four rows per thread as one 16-byte
// load per column and one 16-byte store (the general kernel below moves 1.2 GB for two int32 columns at 3.75 TB/s).
__global__ __launch_bounds__(256) void k_pack_keys_vec32(KeyPack p, int64_t n, uint32_t* out) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  const int64_t nvec = n / 4;
  int bad = 0;
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < nvec; i += int64_t(gridDim.x) * 256) {
    u32x4 c[kMaxPackedKeys];
#pragma unroll
    for (int k = 0; k < kMaxPackedKeys; ++k) {
      if (k >= p.ncols) break;
      c[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const uint32_t*>(p.data[k]) + p.off[k]) + i);
    }
    u32x4 x = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < kMaxPackedKeys; ++k) {
      if (k >= p.ncols) break;
      // (value - bias in 32 bits: the signed / unsigned reading of the column is in the bias; the code fits its field)
      const uint32_t b = static_cast<uint32_t>(p.bias[k]);
      const u32x4 d = c[k] - b;
      if (p.out_of_range) bad |= (d.x > p.span[k]) | (d.y > p.span[k]) | (d.z > p.span[k]) | (d.w > p.span[k]);  // (uniform branch)
      x |= d << static_cast<unsigned int>(p.shift[k]);
    }
    __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out) + i);
  }
  if (p.out_of_range && __syncthreads_or(bad) && threadIdx.x == 0) __hip_atomic_store(p.out_of_range, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (blockIdx.x == 0 && threadIdx.x < n - nvec * 4) {  // the rows behind the last whole vector
    const int64_t r = nvec * 4 + threadIdx.x;
    uint32_t x = 0;
    for (int k = 0; k < p.ncols; ++k) {
      const uint32_t d = reinterpret_cast<const uint32_t*>(p.data[k])[p.off[k] + r] - static_cast<uint32_t>(p.bias[k]);
      if (p.out_of_range && d > p.span[k]) __hip_atomic_store(p.out_of_range, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x |= d << p.shift[k];
    }
    out[r] = x;
  }
}

__global__ __launch_bounds__(256) void k_pack_keys(KeyPack p, int64_t n, uint32_t* out) {
  constexpr int kU = 4;
  for (int64_t base = (int64_t(blockIdx.x) * 256) * kU; base < n; base += int64_t(gridDim.x) * 256 * kU) {
    uint64_t raw[kMaxPackedKeys][kU];
    bool ok[kMaxPackedKeys][kU];
#pragma unroll
    for (int c = 0; c < kMaxPackedKeys; ++c) {
      if (c >= p.ncols) break;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t r = base + u * 256 + threadIdx.x;
        ok[c][u] = !p.valid[c] || GetBit(p.valid[c], p.off[c] + (r < n ? r : n - 1));
      }
      const int64_t o = p.off[c];
#define AA_RAW(T, EXT)                                                                                   \
  _Pragma("unroll") for (int u = 0; u < kU; ++u) {                                                       \
    const int64_t r = base + u * 256 + threadIdx.x;                                                      \
    raw[c][u] = static_cast<uint64_t>(static_cast<EXT>(reinterpret_cast<const T*>(p.data[c])[o + (r < n ? r : n - 1)])); \
  }
      switch (p.load_bits[c]) {
        case 1:
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const int64_t r = base + u * 256 + threadIdx.x;
            raw[c][u] = GetBit(reinterpret_cast<const uint8_t*>(p.data[c]), o + (r < n ? r : n - 1));
          }
          break;
        case 8: if (p.is_signed[c]) { AA_RAW(int8_t, int64_t) } else { AA_RAW(uint8_t, uint64_t) } break;
        case 16: if (p.is_signed[c]) { AA_RAW(int16_t, int64_t) } else { AA_RAW(uint16_t, uint64_t) } break;
        case 32: if (p.is_signed[c]) { AA_RAW(int32_t, int64_t) } else { AA_RAW(uint32_t, uint64_t) } break;
        default: AA_RAW(uint64_t, uint64_t) break;
      }
#undef AA_RAW
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t r = base + u * 256 + threadIdx.x;
      uint64_t x = 0;
#pragma unroll
      for (int c = 0; c < kMaxPackedKeys; ++c) {
        if (c >= p.ncols) break;
        const uint64_t code = ok[c][u] ? raw[c][u] - p.bias[c] : uint64_t(p.null_code[c]);
        x |= code << p.shift[c];
      }
      if (r < n) out[r] = static_cast<uint32_t>(x);
    }
  }
}

// Packed group keys -> the key columns (GetUniques per column, hash_aggregate.cc:410-428): one group per lane,
// a wave's validity word by ballot.
struct UnpackCols { void* values[kMaxPackedKeys]; uint64_t* validity[kMaxPackedKeys]; int width[kMaxPackedKeys]; };
__global__ __launch_bounds__(256) void k_decode_packed_keys(const uint32_t* packed, int64_t G, KeyPack p, UnpackCols o,
                                                            unsigned long long* null_counts /* slotted, one counter per column */) {
  const int lane = threadIdx.x & 63;
  const int64_t nwords = (G + 63) >> 6;
  for (int64_t w = (int64_t(blockIdx.x) * 256 + threadIdx.x) >> 6; w < nwords; w += (int64_t(gridDim.x) * 256) >> 6) {
    const int64_t g = (w << 6) + lane;
    const uint32_t x = g < G ? packed[g] : 0u;
    for (int c = 0; c < p.ncols; ++c) {
      const uint32_t code = p.bits[c] >= 32 ? x : ((x >> p.shift[c]) & ((1u << p.bits[c]) - 1u));
      const bool valid = g < G && !(p.valid[c] && code == p.null_code[c]);
      const uint64_t v = valid ? uint64_t(code) + p.bias[c] : 0;
      const uint64_t m = __ballot(valid);
      if (o.width[c] == 1) {
        const uint64_t bits = __ballot(valid && (v & 1));
        if (lane == 0) reinterpret_cast<uint64_t*>(o.values[c])[w] = bits;
      } else if (g < G) {
        switch (o.width[c]) {
          case 8: static_cast<uint8_t*>(o.values[c])[g] = static_cast<uint8_t>(v); break;
          case 16: static_cast<uint16_t*>(o.values[c])[g] = static_cast<uint16_t>(v); break;
          case 32: static_cast<uint32_t*>(o.values[c])[g] = static_cast<uint32_t>(v); break;
          default: static_cast<uint64_t*>(o.values[c])[g] = v; break;
        }
      }
      if (lane == 0) {
        o.validity[c][w] = m;
        const int64_t left = G - (w << 6);
        const unsigned long long nulls = (left >= 64 ? 64 : left) - __popcll(m);
        if (nulls) atomicAdd(CounterSlot(null_counts) + c, nulls);
      }
    }
  }
}

extern "C" {

int amdarrow_filter_groupby(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                            const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                            void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                            int64_t* out_num_groups) {
  return FilterGroupByImplCall(ctx, values, keys, mask, group_capacity, out_sums, out_counts, out_keys, out_key_validity, out_first_rows,
                           out_num_groups, /*locked=*/false);
}

int amdarrow_filter_groupby_multi(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys, int nkeys,
                                  const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                                  void* const* out_keys, uint8_t* const* out_key_validity, int64_t* out_key_null_counts,
                                  uint32_t* out_first_rows, int64_t* out_num_groups) {
  *out_num_groups = 0;
  if (nkeys < 1 || nkeys > kMaxPackedKeys) return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: 1 to 4 packed key columns");
  const int64_t n = keys[0].length;
  for (int c = 0; c < nkeys; ++c) {
    if (keys[c].length != n) return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");
    const int t = keys[c].type_id;
    if (t == AMDARROW_FLOAT || t == AMDARROW_DOUBLE || BitWidth(t) < 0)
      return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: packed keys are integer or boolean columns");
    out_key_null_counts[c] = 0;
  }
  if (n == 0) return AMDARROW_OK;
  if (n >= (int64_t(1) << 32) - 1) return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: 2^32 rows or more per call");
  std::lock_guard<std::mutex> lock(ctx->mu);
  // the ranges of the last call of this shape (32-bit columns without a validity buffer, every slice on a 16-byte
  // boundary): taken on trust, every value checked while packing, the call redone with measured ranges on a miss
  auto& ph = ctx->pack_hint;
  bool hintable = n >= 1024;
  for (int c = 0; c < nkeys; ++c)
    hintable = hintable && BitWidth(keys[c].type_id) == 32 && keys[c].validity == nullptr &&
               (reinterpret_cast<uintptr_t>(keys[c].values) + 4 * static_cast<uintptr_t>(keys[c].offset)) % 16 == 0;
  bool hinted = hintable && ph.n == n && ph.ncols == nkeys && !getenv("AMDARROW_GROUPBY_NO_HINT");
  for (int c = 0; c < nkeys && hinted; ++c) hinted = ph.types[c] == keys[c].type_id;
  for (;;) {  // (second trip: the hinted ranges did not hold)
  // plan: one pass per key column for its range (the aggregate kernel's min / max), bits = log2(range [+ a code for null])
  KeyPack p{};
  p.ncols = nkeys;
  int total = 0;
  for (int c = 0; c < nkeys; ++c) {
    const amdarrow_array& k = keys[c];
    amdarrow_agg r;
    if (hinted) { r = amdarrow_agg{}; r.valid_count = n; r.min = ph.bias[c]; r.max = ph.bias[c] + ph.span[c]; }
    else
    AA_TRY(AggregateLocked(ctx, &k, &r));
    const bool has_null = k.validity && k.null_count != 0 && r.valid_count < n;
    const uint64_t span = r.valid_count > 0 ? r.max - r.min : 0;  // wrap-around difference: exact for both signednesses
    if (span >= 0xFFFFFFFEull) return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: key range too wide to pack");
    const uint64_t codes = (r.valid_count > 0 ? span + 1 : 0) + (has_null ? 1 : 0);
    int bits = 0;
    while (bits < 33 && (uint64_t(1) << bits) < codes) ++bits;
    p.data[c] = k.values; p.valid[c] = has_null ? k.validity : nullptr; p.off[c] = k.offset;
    p.load_bits[c] = BitWidth(k.type_id);
    p.is_signed[c] = k.type_id == AMDARROW_INT8 || k.type_id == AMDARROW_INT16 || k.type_id == AMDARROW_INT32 || k.type_id == AMDARROW_INT64;
    p.bias[c] = r.valid_count > 0 ? r.min : 0;
    p.null_code[c] = static_cast<uint32_t>(r.valid_count > 0 ? span + 1 : 0);
    p.bits[c] = bits;
    p.shift[c] = total;
    p.span[c] = static_cast<uint32_t>(span);
    if (!hinted && hintable) { ph.types[c] = k.type_id; ph.bias[c] = p.bias[c]; ph.span[c] = static_cast<uint32_t>(span); }
    total += bits;
    if (total > 32) return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: the key columns need more than 32 bits together");
  }
  // the single-key machinery on the column of packed codes (grow-only buffer of the context: codes of the rows, then of
  // the groups, then a validity bitmap nobody reads)
  const int64_t bound = group_capacity < n ? group_capacity : n;
  const size_t need = 4 * static_cast<size_t>(n) + 4 * static_cast<size_t>(bound) + static_cast<size_t>(bound) / 8 + 1024;
  if (ctx->pack_cap < need) {
    if (ctx->pack_buf) (void)hipFree(ctx->pack_buf);
    ctx->pack_buf = nullptr;
    ctx->pack_cap = 0;
    AA_TRY(DeviceMalloc(ctx, &ctx->pack_buf, need));
    ctx->pack_cap = need;
  }
  auto* codes = static_cast<uint32_t*>(ctx->pack_buf);
  uint32_t* packed = codes + ((n + 63) / 64) * 64;
  uint8_t* dummy_validity = reinterpret_cast<uint8_t*>(packed + ((bound + 63) / 64) * 64);
  unsigned int* range_flag = reinterpret_cast<unsigned int*>(static_cast<char*>(ctx->pack_buf) + ((need - 256) & ~size_t(255)));
  p.out_of_range = nullptr;
  if (hinted) { AA_HIP(hipMemsetAsync(range_flag, 0, 4, ctx->stream)); p.out_of_range = range_flag; }
  bool vec32 = n >= 1024 && reinterpret_cast<uintptr_t>(codes) % 16 == 0;
  for (int k = 0; k < p.ncols; ++k)
    vec32 = vec32 && p.load_bits[k] == 32 && !p.valid[k] &&
            (reinterpret_cast<uintptr_t>(p.data[k]) + 4 * static_cast<uintptr_t>(p.off[k])) % 16 == 0;
  if (hinted && !vec32) { hinted = false; ph.n = -1; continue; }  // (only the vector kernel checks the hinted ranges)
  if (vec32) hipLaunchKernelGGL(k_pack_keys_vec32, dim3(GridFor(n / 4, 256, ctx->num_cus * 16)), dim3(256), 0, ctx->stream, p, n, codes);
  else
  hipLaunchKernelGGL(k_pack_keys, dim3(GridFor(n, 1024, ctx->num_cus * 8)), dim3(256), 0, ctx->stream, p, n, codes);
  AA_HIP(hipGetLastError());
  amdarrow_array stand_in{};
  stand_in.values = codes; stand_in.validity = nullptr; stand_in.length = n; stand_in.offset = 0;
  stand_in.type_id = AMDARROW_UINT32; stand_in.null_count = 0;
  int64_t G = 0;
  CallerScope who(ctx, kCallerPackedKeys);
  int rc = FilterGroupByImplCall(ctx, values, &stand_in, mask, group_capacity, out_sums, out_counts, packed, dummy_validity, out_first_rows,
                             &G, /*locked=*/true);
  if (hinted) {  // (the group-by has synchronised: this read-back finds the stream idle)
    unsigned int missed = 0;
    AA_TRY(ReadBack(ctx, range_flag, 4, &missed));
    if (missed) { hinted = false; ph.n = -1; continue; }
  } else if (hintable && rc == AMDARROW_OK) {
    ph.n = n; ph.ncols = nkeys;
  }
  if (rc == AMDARROW_OK && G > 0) {
    void* base;
    rc = Scratch(ctx, kCounterBytes, &base);
    if (rc == AMDARROW_OK) {
      auto* counters = static_cast<unsigned long long*>(base);
      rc = ClearCounters(ctx, counters);
      UnpackCols o{};
      for (int c = 0; c < nkeys; ++c) {
        o.values[c] = out_keys[c];
        o.validity[c] = reinterpret_cast<uint64_t*>(out_key_validity[c]);
        o.width[c] = BitWidth(keys[c].type_id);
      }
      if (rc == AMDARROW_OK) {
        hipLaunchKernelGGL(k_decode_packed_keys, dim3(GridFor(G, 256, ctx->num_cus * 8)), dim3(256), 0, ctx->stream, packed, G, p, o,
                           counters);
        unsigned long long nulls[kMaxPackedKeys] = {0, 0, 0, 0};
        rc = ReadCounters(ctx, counters, nkeys, nulls);
        for (int c = 0; c < nkeys; ++c) out_key_null_counts[c] = static_cast<int64_t>(nulls[c]);
      }
    }
  }
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  AA_TRY(rc);
  AA_HIP(e);
  *out_num_groups = G;
  return AMDARROW_OK;
  }  // (trips)
}

}  // extern "C"

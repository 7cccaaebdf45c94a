// grouper_var.hip -- Grouper over ONE variable-width key column (binary / utf8 / large_binary / large_utf8):
// VarLengthKeyEncoder + GrouperImpl, compute/kernels/hash_aggregate.cc:237-311, 358-437.
//
// Reference semantics: a key is its bytes and its length (two keys are one group iff both match), a null key is its
// own group, ids are handed out in order of first appearance, GetUniques returns the keys in id order with a validity
// bitmap only if the null group exists (DecodeNulls :64-94).
//
// Design: every row's bytes are hashed to 64 bits, the hash column (with the key column's validity) is grouped by
// the fixed-width Grouper of groupby.hip -- ids in first-appearance order, the partitioned paths for large batches --
// and the grouper keeps ONE representative string per group in HBM (the unique store: int64 offsets + bytes; any row
// of a new group can be the representative).  Then EVERY row of the batch is compared, byte for byte, with the
// representative of the group it was given: a mismatch means two different strings met in one hash.  That is never
// assumed away: the grouper then rebuilds its hash table under another seed FROM THE UNIQUE STORE (the strings of
// groups 0..G-1 in id order, so they get their ids back -- checked) and consumes the batch again; four seeds, then a
// loud error.  Results are therefore exact or the call fails; a 64-bit collision among 10^8 distinct keys has
// probability ~3e-4 per seed.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace amdarrow {

constexpr int kGThreads = 256;
constexpr uint32_t kNoGroup = 0xFFFFFFFFu;

__device__ __forceinline__ uint64_t Mix64(uint64_t h) {
  h ^= h >> 32; h *= 0xd6e8feb86659fd93ULL; h ^= h >> 32; h *= 0xd6e8feb86659fd93ULL; h ^= h >> 32;
  return h;
}
// bytes -> 64 bits: eight bytes per step (unaligned loads), the tail byte by byte, the length folded in last
__device__ __forceinline__ uint64_t HashBytes(const uint8_t* p, int64_t len, uint64_t seed) {
  uint64_t h = seed ^ (static_cast<uint64_t>(len) * 0x9E3779B97F4A7C15ULL);
  int64_t t = 0;
  for (; t + 8 <= len; t += 8) {
    uint64_t q;
    __builtin_memcpy(&q, p + t, 8);
    h = Mix64(h ^ q) + 0x9E3779B97F4A7C15ULL;
  }
  uint64_t tail = 0;
  for (int s = 0; t < len; ++t, s += 8) tail |= static_cast<uint64_t>(p[t]) << s;
  return Mix64(h ^ tail ^ 0xA5A5A5A5A5A5A5A5ULL);
}

template <typename OffT>
__global__ __launch_bounds__(kGThreads) void k_hash_rows(const OffT* offsets, const uint8_t* data, const uint8_t* valid, int64_t off,
                                                         int64_t n, uint64_t seed, uint64_t keep_mask, unsigned long long* hashes,
                                                         uint64_t* out_valid) {
  const int lane = threadIdx.x & 63;
  const int64_t ground = (n + 63) & ~int64_t(63);
  for (int64_t i = int64_t(blockIdx.x) * kGThreads + threadIdx.x; i < ground; i += int64_t(gridDim.x) * kGThreads) {
    bool v = false;
    if (i < n) {
      v = valid ? GetBit(valid, off + i) : true;
      uint64_t h = 0;
      if (v) {
        const int64_t a = static_cast<int64_t>(offsets[off + i]), b = static_cast<int64_t>(offsets[off + i + 1]);
        h = HashBytes(data + a, b - a, seed) & keep_mask;
      }
      hashes[i] = h;
    }
    const uint64_t m = __ballot(v);
    if (lane == 0) out_valid[i >> 6] = m;
  }
}

// any row of a new group (id >= first_new) is its representative: racing writers all name a row of that group
__global__ __launch_bounds__(kGThreads) void k_pick_representatives(const uint32_t* ids, int64_t n, uint32_t first_new, long long* rep) {
  for (int64_t i = int64_t(blockIdx.x) * kGThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kGThreads) {
    const uint32_t g = ids[i];
    if (g >= first_new) __hip_atomic_store(&rep[g - first_new], static_cast<long long>(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// lengths of the representatives (0 for the null key, whose group id is reported)
template <typename OffT>
__global__ __launch_bounds__(kGThreads) void k_rep_lengths(const OffT* offsets, const uint8_t* valid, int64_t off, const long long* rep,
                                                           int64_t count, uint32_t first_new, long long* lengths, uint32_t* null_id) {
  for (int64_t j = int64_t(blockIdx.x) * kGThreads + threadIdx.x; j < count; j += int64_t(gridDim.x) * kGThreads) {
    const int64_t r = rep[j];
    const bool v = valid ? GetBit(valid, off + r) : true;
    lengths[j] = v ? static_cast<long long>(offsets[off + r + 1]) - static_cast<long long>(offsets[off + r]) : 0;
    if (!v) *null_id = first_new + static_cast<uint32_t>(j);
  }
}
// store_offsets[first_new + j] = base + local[j] (j <= count), and the representatives' bytes behind them
template <typename OffT>
__global__ __launch_bounds__(kGThreads) void k_append_store(const OffT* offsets, const uint8_t* data, int64_t off, const long long* rep,
                                                            int64_t count, const long long* local, long long base, long long* store_off,
                                                            uint8_t* store_data) {
  for (int64_t j = int64_t(blockIdx.x) * kGThreads + threadIdx.x; j <= count; j += int64_t(gridDim.x) * kGThreads) {
    store_off[j] = base + local[j];
    if (j == count) continue;
    const long long len = local[j + 1] - local[j];
    const uint8_t* sp = data + static_cast<int64_t>(offsets[off + rep[j]]);
    uint8_t* dp = store_data + base + local[j];
    long long t = 0;
    for (; t + 8 <= len; t += 8) { unsigned long long q; __builtin_memcpy(&q, sp + t, 8); __builtin_memcpy(dp + t, &q, 8); }
    for (; t < len; ++t) dp[t] = sp[t];
  }
}
// every row against the representative of its group; mismatches are counted (any value != 0 is a collision)
template <typename OffT>
__global__ __launch_bounds__(kGThreads) void k_verify_rows(const OffT* offsets, const uint8_t* data, const uint8_t* valid, int64_t off,
                                                           int64_t n, const uint32_t* ids, const long long* store_off,
                                                           const uint8_t* store_data, uint32_t null_id, unsigned long long* mismatches) {
  unsigned long long bad = 0;
  for (int64_t i = int64_t(blockIdx.x) * kGThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kGThreads) {
    const uint32_t g = ids[i];
    const bool v = valid ? GetBit(valid, off + i) : true;
    if (!v) { bad += g != null_id; continue; }
    if (g == null_id) { ++bad; continue; }
    const int64_t a = static_cast<int64_t>(offsets[off + i]), len = static_cast<int64_t>(offsets[off + i + 1]) - a;
    const long long sa = store_off[g], slen = store_off[g + 1] - sa;
    if (len != slen) { ++bad; continue; }
    const uint8_t* p = data + a;
    const uint8_t* q = store_data + sa;
    int64_t t = 0;
    bool same = true;
    for (; t + 8 <= len; t += 8) {
      unsigned long long x, y;
      __builtin_memcpy(&x, p + t, 8); __builtin_memcpy(&y, q + t, 8);
      same &= x == y;
    }
    for (; t < len; ++t) same &= p[t] == q[t];
    bad += !same;
  }
  bad = WaveReduceSum64(bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(CounterSlot(mismatches), bad);
}
__global__ __launch_bounds__(kGThreads) void k_check_iota(const uint32_t* ids, int64_t n, unsigned long long* mismatches) {
  unsigned long long bad = 0;
  for (int64_t i = int64_t(blockIdx.x) * kGThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kGThreads) bad += ids[i] != static_cast<uint32_t>(i);
  bad = WaveReduceSum64(bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(CounterSlot(mismatches), bad);
}
// the uniques as the key type has them: offsets of OffT, a validity bitmap (whole words) that clears the null group's bit
template <typename OffT>
__global__ __launch_bounds__(kGThreads) void k_store_to_output(const long long* store_off, int64_t G, uint32_t null_id, OffT* out_off,
                                                               uint64_t* out_valid) {
  const int lane = threadIdx.x & 63;
  const int64_t ground = (G + 1 + 63) & ~int64_t(63);
  for (int64_t g = int64_t(blockIdx.x) * kGThreads + threadIdx.x; g < ground; g += int64_t(gridDim.x) * kGThreads) {
    if (out_off && g <= G) out_off[g] = static_cast<OffT>(store_off[g]);
    const uint64_t m = __ballot(g < G && static_cast<uint32_t>(g) != null_id);
    if (lane == 0 && out_valid && (g >> 6) < ((G + 63) >> 6)) out_valid[g >> 6] = m;
  }
}

}  // namespace amdarrow

using namespace amdarrow;

struct amdarrow_binary_grouper {
  amdarrow_ctx* ctx = nullptr;
  int offset_bits = 32;
  amdarrow_grouper* inner = nullptr;
  int seed_index = 0;
  uint32_t num_groups = 0;
  uint32_t null_id = kNoGroup;
  // the unique store
  long long* store_off = nullptr; int64_t store_rows_cap = 0;
  uint8_t* store_data = nullptr; int64_t store_bytes_cap = 0; int64_t store_bytes = 0;
  // per-call temporaries (grow-only)
  void* tmp = nullptr; size_t tmp_cap = 0;
  unsigned long long* counters = nullptr;  // kCounterBytes + 256
};

static const uint64_t kSeeds[4] = {0x243F6A8885A308D3ULL, 0x13198A2E03707344ULL, 0xA4093822299F31D0ULL, 0x082EFA98EC4E6C89ULL};

static int EnsureTmp(amdarrow_binary_grouper* g, size_t bytes, char** out) {
  if (bytes > g->tmp_cap) {
    AA_HIP(hipStreamSynchronize(g->ctx->stream));
    if (g->tmp) (void)hipFree(g->tmp);
    g->tmp = nullptr; g->tmp_cap = 0;
    const size_t cap = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    AA_TRY(DeviceMalloc(g->ctx, &g->tmp, cap));
    g->tmp_cap = cap;
  }
  *out = static_cast<char*>(g->tmp);
  return AMDARROW_OK;
}
static int GrowStore(amdarrow_binary_grouper* g, int64_t rows, int64_t bytes) {
  amdarrow_ctx* ctx = g->ctx;
  if (rows + 1 > g->store_rows_cap) {
    const int64_t cap = std::max<int64_t>(2 * (rows + 1), 1024);
    long long* n;
    AA_TRY(DeviceMalloc(ctx, reinterpret_cast<void**>(&n), 8 * static_cast<size_t>(cap)));
    if (g->store_off) {
      AA_HIP(hipMemcpyAsync(n, g->store_off, 8 * (static_cast<size_t>(g->num_groups) + 1), hipMemcpyDeviceToDevice, ctx->stream));
      AA_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(g->store_off);
    } else {
      AA_HIP(hipMemsetAsync(n, 0, 8, ctx->stream));
    }
    g->store_off = n; g->store_rows_cap = cap;
  }
  if (bytes > g->store_bytes_cap) {
    const int64_t cap = std::max<int64_t>(2 * bytes, 4096);
    uint8_t* n;
    AA_TRY(DeviceMalloc(ctx, reinterpret_cast<void**>(&n), static_cast<size_t>(cap) + 16));
    if (g->store_data) {
      AA_HIP(hipMemcpyAsync(n, g->store_data, static_cast<size_t>(g->store_bytes), hipMemcpyDeviceToDevice, ctx->stream));
      AA_HIP(hipStreamSynchronize(ctx->stream));
      (void)hipFree(g->store_data);
    }
    g->store_data = n; g->store_bytes_cap = cap;
  }
  return AMDARROW_OK;
}

static int NewInner(amdarrow_binary_grouper* g) {
  if (g->inner) amdarrow_grouper_destroy(g->inner);
  g->inner = nullptr;
  const int t = AMDARROW_UINT64;
  return amdarrow_grouper_create(g->ctx, 1, &t, &g->inner);
}

// hash `n` rows of (offsets, data, validity) and group them; -> ids
template <typename OffT>
static int HashAndGroup(amdarrow_binary_grouper* g, const OffT* offsets, const uint8_t* data, const uint8_t* valid, int64_t off, int64_t n,
                        int64_t null_count, unsigned long long* hashes, uint64_t* hvalid, uint32_t* out_ids) {
  amdarrow_ctx* ctx = g->ctx;
  // (testing knob: the FIRST seed keeps only its low bits, so that distinct keys do collide and the rebuild runs)
  uint64_t keep = ~uint64_t(0);
  if (g->seed_index == 0) if (const char* env = getenv("AMDARROW_BINARY_GROUPER_WEAK_FIRST_HASH")) keep = (uint64_t(1) << std::max(1, atoi(env))) - 1;
  hipLaunchKernelGGL((k_hash_rows<OffT>), dim3(GridFor(n, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream, offsets, data, valid,
                     off, n, kSeeds[g->seed_index], keep, hashes, hvalid);
  AA_HIP(hipGetLastError());
  amdarrow_array col{};
  col.values = hashes; col.validity = valid ? reinterpret_cast<const uint8_t*>(hvalid) : nullptr;
  col.length = n; col.offset = 0; col.type_id = AMDARROW_UINT64; col.null_count = valid ? null_count : 0;
  return amdarrow_grouper_consume(g->inner, &col, out_ids);
}

template <typename OffT>
static int ConsumeImpl(amdarrow_binary_grouper* g, const amdarrow_var_array* keys, uint32_t* out_ids) {
  amdarrow_ctx* ctx = g->ctx;
  const int64_t n = keys->length;
  if (n == 0) return AMDARROW_OK;
  if (n >= (int64_t(1) << 31)) return SetError(AMDARROW_NOT_IMPLEMENTED, "Grouper: 2^31 rows or more per batch");
  const OffT* offsets = static_cast<const OffT*>(keys->offsets);
  const uint8_t* valid = keys->validity && keys->null_count != 0 ? keys->validity : nullptr;
  for (;;) {
    Carver cv;
    const size_t o_hash = cv.Take(8 * static_cast<size_t>(n));
    const size_t o_hvalid = cv.Take(8 * static_cast<size_t>((n + 63) / 64 + 1));
    const size_t o_rep = cv.Take(8 * static_cast<size_t>(n));
    const size_t o_len = cv.Take(8 * static_cast<size_t>(n + 1));
    const size_t o_local = cv.Take(8 * static_cast<size_t>(n + 1));
    const size_t o_sums = cv.Take(8 * static_cast<size_t>(Scan64Blocks(n) + 2));
    const size_t o_word = cv.Take(256);
    const size_t o_cnt = cv.Take(kCounterBytes);
    char* b;
    AA_TRY(EnsureTmp(g, cv.off, &b));
    auto* hashes = reinterpret_cast<unsigned long long*>(b + o_hash);
    auto* hvalid = reinterpret_cast<uint64_t*>(b + o_hvalid);
    auto* rep = reinterpret_cast<long long*>(b + o_rep);
    auto* lengths = reinterpret_cast<long long*>(b + o_len);
    auto* local = reinterpret_cast<long long*>(b + o_local);
    auto* sums = reinterpret_cast<long long*>(b + o_sums);
    auto* word = reinterpret_cast<uint32_t*>(b + o_word);  // [0] null id found among the new groups, [1] scan overflow (unused: 64-bit)
    auto* counters = reinterpret_cast<unsigned long long*>(b + o_cnt);

    const uint32_t G0 = g->num_groups;
    AA_TRY(HashAndGroup<OffT>(g, offsets, keys->data, valid, keys->offset, n, keys->null_count, hashes, hvalid, out_ids));
    uint32_t G1 = 0;
    AA_TRY(amdarrow_grouper_num_groups(g->inner, &G1));
    const int64_t fresh = static_cast<int64_t>(G1) - G0;
    {  // (the context's pinned read-back words are shared with every other call on it)
    std::lock_guard<std::mutex> lock(ctx->mu);
    uint32_t null_id = g->null_id;
    int64_t new_bytes = 0;
    if (fresh > 0) {
      AA_HIP(hipMemsetAsync(word, 0xFF, 8, ctx->stream));
      hipLaunchKernelGGL(k_pick_representatives, dim3(GridFor(n, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream, out_ids, n, G0, rep);
      hipLaunchKernelGGL((k_rep_lengths<OffT>), dim3(GridFor(fresh, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream, offsets, valid,
                         keys->offset, rep, fresh, G0, lengths, word);
      AA_HIP(hipGetLastError());
      AA_TRY(LaunchScan64(ctx, lengths, fresh, sums, 64, local, word + 1));
      long long total = 0;
      uint32_t found_null = kNoGroup;
      AA_TRY(ReadBack(ctx, sums + Scan64Blocks(fresh), 8, &total));
      AA_TRY(ReadBack(ctx, word, 4, &found_null));
      new_bytes = total;
      if (found_null != kNoGroup) null_id = found_null;
      AA_TRY(GrowStore(g, G1, g->store_bytes + new_bytes));
      hipLaunchKernelGGL((k_append_store<OffT>), dim3(GridFor(fresh + 1, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream, offsets,
                         keys->data, keys->offset, rep, fresh, local, static_cast<long long>(g->store_bytes), g->store_off + G0, g->store_data);
      AA_HIP(hipGetLastError());
    }
    AA_TRY(ClearCounters(ctx, counters));
    hipLaunchKernelGGL((k_verify_rows<OffT>), dim3(GridFor(n, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream, offsets, keys->data,
                       valid, keys->offset, n, out_ids, g->store_off, g->store_data, null_id, counters);
    AA_HIP(hipGetLastError());
    unsigned long long bad = 0;
    AA_TRY(ReadCounters(ctx, counters, 1, &bad));
    if (bad == 0) {  // every row carries its group's bytes: commit
      g->num_groups = G1; g->null_id = null_id; g->store_bytes += new_bytes;
      return AMDARROW_OK;
    }
    }
    // Two different keys met in one hash.  Another seed: the table is rebuilt from the unique store (groups 0..G0-1 in
    // id order get their ids back -- checked), the tentative appends of this attempt are dropped, the batch is redone.
    for (;;) {
      if (++g->seed_index >= 4) return SetError(AMDARROW_UNKNOWN_ERROR, "Grouper: variable-width keys collided under four hash seeds");
      AA_TRY(NewInner(g));
      if (G0 == 0) break;
      Carver c2;
      const size_t p_hash = c2.Take(8 * static_cast<size_t>(G0));
      const size_t p_sv = c2.Take(8 * static_cast<size_t>((G0 + 63) / 64 + 1));   // the store's validity: all ones but the null group
      const size_t p_hv = c2.Take(8 * static_cast<size_t>((G0 + 63) / 64 + 1));   // the hash column's (k_hash_rows writes its own copy)
      const size_t p_ids = c2.Take(4 * static_cast<size_t>(G0));
      const size_t p_cnt = c2.Take(kCounterBytes);
      char* b2;
      AA_TRY(EnsureTmp(g, c2.off, &b2));  // (nothing of the failed attempt is needed any more)
      auto* h2 = reinterpret_cast<unsigned long long*>(b2 + p_hash);
      auto* sv_words = reinterpret_cast<uint64_t*>(b2 + p_sv);
      auto* ids2 = reinterpret_cast<uint32_t*>(b2 + p_ids);
      auto* cnt2 = reinterpret_cast<unsigned long long*>(b2 + p_cnt);
      const uint8_t* sv = nullptr;
      if (g->null_id != kNoGroup) {
        hipLaunchKernelGGL((k_store_to_output<long long>), dim3(GridFor(int64_t(G0) + 1, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream,
                           g->store_off, static_cast<int64_t>(G0), g->null_id, static_cast<long long*>(nullptr), sv_words);
        AA_HIP(hipGetLastError());
        sv = reinterpret_cast<const uint8_t*>(sv_words);
      }
      const int rc = HashAndGroup<long long>(g, g->store_off, g->store_data, sv, 0, G0, sv ? 1 : 0, h2, reinterpret_cast<uint64_t*>(b2 + p_hv), ids2);
      AA_TRY(rc);
      std::lock_guard<std::mutex> lock(ctx->mu);
      AA_TRY(ClearCounters(ctx, cnt2));
      hipLaunchKernelGGL(k_check_iota, dim3(GridFor(G0, kGThreads, ctx->num_cus * 8)), dim3(kGThreads), 0, ctx->stream, ids2, static_cast<int64_t>(G0), cnt2);
      AA_HIP(hipGetLastError());
      unsigned long long off_order = 0;
      AA_TRY(ReadCounters(ctx, cnt2, 1, &off_order));
      if (off_order == 0) break;  // the known groups have their ids back under this seed
    }
  }
}

extern "C" {

int amdarrow_binary_grouper_create(amdarrow_ctx* ctx, int offset_bits, amdarrow_binary_grouper** out) {
  if (offset_bits != 32 && offset_bits != 64) return SetError(AMDARROW_INVALID, "Grouper: offsets must be 32 or 64 bits wide");
  auto* g = new amdarrow_binary_grouper;
  g->ctx = ctx;
  g->offset_bits = offset_bits;
  int rc = NewInner(g);
  if (rc == AMDARROW_OK) rc = GrowStore(g, 0, 0);
  if (rc != AMDARROW_OK) { amdarrow_binary_grouper_destroy(g); return rc; }
  *out = g;
  return AMDARROW_OK;
}

int amdarrow_binary_grouper_destroy(amdarrow_binary_grouper* g) {
  if (!g) return AMDARROW_OK;
  (void)hipStreamSynchronize(g->ctx->stream);
  if (g->inner) amdarrow_grouper_destroy(g->inner);
  if (g->store_off) (void)hipFree(g->store_off);
  if (g->store_data) (void)hipFree(g->store_data);
  if (g->tmp) (void)hipFree(g->tmp);
  delete g;
  return AMDARROW_OK;
}

int amdarrow_binary_grouper_consume(amdarrow_binary_grouper* g, const amdarrow_var_array* keys, uint32_t* out_ids) {
  if (keys->offset_bits != g->offset_bits) return SetError(AMDARROW_TYPE_ERROR, "Grouper: key column of another offset width");
  return g->offset_bits == 32 ? ConsumeImpl<int32_t>(g, keys, out_ids) : ConsumeImpl<long long>(g, keys, out_ids);
}

int amdarrow_binary_grouper_num_groups(amdarrow_binary_grouper* g, uint32_t* out_num_groups) {
  *out_num_groups = g->num_groups;
  return AMDARROW_OK;
}

int amdarrow_binary_grouper_uniques_size(amdarrow_binary_grouper* g, int64_t* out_rows, int64_t* out_data_bytes) {
  *out_rows = g->num_groups;
  *out_data_bytes = g->store_bytes;
  return AMDARROW_OK;
}

int amdarrow_binary_grouper_uniques(amdarrow_binary_grouper* g, void* out_offsets, uint8_t* out_data, uint8_t* out_validity,
                                    amdarrow_out* out) {
  amdarrow_ctx* ctx = g->ctx;
  const int64_t G = g->num_groups;
  out->length = G;
  out->null_count = g->null_id != kNoGroup ? 1 : 0;
  out->has_validity = out->null_count > 0;  // DecodeNulls :64-94
  if (g->offset_bits == 32 && g->store_bytes > 2147483647LL) return SetError(AMDARROW_INVALID, "Grouper: unique keys overflow 32-bit offsets");
  uint64_t* vw = out->has_validity ? reinterpret_cast<uint64_t*>(out_validity) : nullptr;
  if (out->has_validity && !out_validity) return SetError(AMDARROW_INVALID, "Grouper: validity buffer required");
  const int grid = GridFor(G + 1, kGThreads, ctx->num_cus * 8);
  if (g->offset_bits == 32)
    hipLaunchKernelGGL((k_store_to_output<int32_t>), dim3(grid), dim3(kGThreads), 0, ctx->stream, g->store_off, G, g->null_id,
                       static_cast<int32_t*>(out_offsets), vw);
  else
    hipLaunchKernelGGL((k_store_to_output<long long>), dim3(grid), dim3(kGThreads), 0, ctx->stream, g->store_off, G, g->null_id,
                       static_cast<long long*>(out_offsets), vw);
  AA_HIP(hipGetLastError());
  if (g->store_bytes > 0) AA_HIP(hipMemcpyAsync(out_data, g->store_data, static_cast<size_t>(g->store_bytes), hipMemcpyDeviceToDevice, ctx->stream));
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}

}  // extern "C"

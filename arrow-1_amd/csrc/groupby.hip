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

constexpr int kThreads = 256;
constexpr uint64_t kEmpty = ~uint64_t(0);
constexpr uint32_t kNoRow = 0xFFFFFFFFu;

__host__ __device__ __forceinline__ uint64_t Fmix64(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33;
  return h;
}
__host__ __device__ __forceinline__ uint64_t Fmix64Inv(uint64_t h) {
  h ^= h >> 33; h *= 0x9cb4b2f8129337dbULL; h ^= h >> 33; h *= 0x4f74430c22a54005ULL; h ^= h >> 33;
  return h;
}

// 32-bit murmur3 finaliser: also a bijection; used where keys fit 32 bits (cheaper than fmix64).
__host__ __device__ __forceinline__ uint32_t Fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
__host__ __device__ __forceinline__ uint32_t Fmix32Inv(uint32_t h) {
  h ^= h >> 16; h *= 0x7ed1b41du; h ^= h >> 13; h ^= h >> 26; h *= 0xa5cb9243u; h ^= h >> 16;
  return h;
}

// ---- key packing ------------------------------------------------------------------------------
constexpr int kMaxKeys = 4;
struct KeyCol { const void* data; const uint8_t* valid; int64_t off; int bits; int shift; };
struct KeySpec {
  int nkeys;
  int total_bits;      // sum of value bits
  int wide_null_slot;  // single 64-bit key: a null key goes to the dedicated slot
  KeyCol col[kMaxKeys];
};

__device__ __forceinline__ uint64_t LoadBitsOfWidth(const void* data, int64_t i, int bits) {
  switch (bits) {
    case 1: return GetBit(reinterpret_cast<const uint8_t*>(data), i);
    case 8: return reinterpret_cast<const uint8_t*>(data)[i];
    case 16: return reinterpret_cast<const uint16_t*>(data)[i];
    case 32: return reinterpret_cast<const uint32_t*>(data)[i];
    default: return reinterpret_cast<const uint64_t*>(data)[i];
  }
}
// Returns true when the row's key is the null of a single 64-bit key column.
__device__ __forceinline__ bool EncodeKey(const KeySpec& ks, int64_t row, uint64_t* x_out) {
  uint64_t x = 0;
#pragma unroll
  for (int k = 0; k < kMaxKeys; ++k) {
    if (k >= ks.nkeys) break;
    const KeyCol& c = ks.col[k];
    const bool valid = c.valid ? GetBit(c.valid, c.off + row) : true;
    if (valid) x |= LoadBitsOfWidth(c.data, c.off + row, c.bits) << c.shift;   // [kValidByte][bytes]
    else if (ks.wide_null_slot) return true;
    else x |= uint64_t(1) << (ks.total_bits + k);                               // [kNullByte][zeros]
  }
  *x_out = x;
  return false;
}

// ---- the group table --------------------------------------------------------------------------
// T power-of-two slots + 2 special slots: [T] the null of a single 64-bit key, [T+1] the one
// key whose hash equals the empty marker.
struct GroupTable {
  unsigned long long* hkeys;  // kEmpty = free
  uint32_t* first;            // first selected row of the group; kNoRow = none yet
  unsigned long long* sums;   // int64/uint64 (wrapping) or double bits
  unsigned long long* counts; // valid values per group
  uint32_t* ids;              // Grouper only: assigned dense id, kNoRow = unassigned
  uint64_t mask;              // T - 1
};

__device__ __forceinline__ int64_t FindOrInsert(const GroupTable& t, uint64_t h, bool insert) {
  if (h == kEmpty) return static_cast<int64_t>(t.mask) + 2;
  uint64_t idx = h & t.mask;
  // A chain this long does not occur below ~90 % load: the table is (nearly) full -- report it instead of walking it
  const uint64_t limit = t.mask < 4095 ? t.mask : 4095;
  for (uint64_t probes = 0; probes <= limit; ++probes) {
    unsigned long long cur = t.hkeys[idx];
    if (cur == h) return static_cast<int64_t>(idx);
    if (cur == kEmpty) {
      if (!insert) return -1;
      cur = atomicCAS(&t.hkeys[idx], static_cast<unsigned long long>(kEmpty), static_cast<unsigned long long>(h));
      if (cur == kEmpty || cur == h) return static_cast<int64_t>(idx);
    }
    idx = (idx + 1) & t.mask;
  }
  return -1;  // table full
}

struct ValueCol { const void* data; const uint8_t* valid; int64_t off; int type_id; };

// value -> accumulator bits (FindAccumulatorType): signed -> int64, unsigned/bool -> uint64,
// float/double -> double
__device__ __forceinline__ uint64_t LoadAccum(const ValueCol& v, int64_t row, bool* is_double) {
  const int64_t i = v.off + row;
  *is_double = false;
  switch (v.type_id) {
    case AMDARROW_BOOL: return GetBit(reinterpret_cast<const uint8_t*>(v.data), i);
    case AMDARROW_UINT8: return reinterpret_cast<const uint8_t*>(v.data)[i];
    case AMDARROW_UINT16: return reinterpret_cast<const uint16_t*>(v.data)[i];
    case AMDARROW_UINT32: return reinterpret_cast<const uint32_t*>(v.data)[i];
    case AMDARROW_UINT64: return reinterpret_cast<const uint64_t*>(v.data)[i];
    case AMDARROW_INT8: return static_cast<uint64_t>(static_cast<int64_t>(reinterpret_cast<const int8_t*>(v.data)[i]));
    case AMDARROW_INT16: return static_cast<uint64_t>(static_cast<int64_t>(reinterpret_cast<const int16_t*>(v.data)[i]));
    case AMDARROW_INT32: return static_cast<uint64_t>(static_cast<int64_t>(reinterpret_cast<const int32_t*>(v.data)[i]));
    case AMDARROW_INT64: return static_cast<uint64_t>(reinterpret_cast<const int64_t*>(v.data)[i]);
    case AMDARROW_FLOAT: { *is_double = true; double d = reinterpret_cast<const float*>(v.data)[i]; return __double_as_longlong(d); }
    default: { *is_double = true; return reinterpret_cast<const uint64_t*>(v.data)[i]; }
  }
}
__device__ __forceinline__ void Accumulate(unsigned long long* slot, uint64_t bits, bool is_double) {
  if (is_double) atomicAdd(reinterpret_cast<double*>(slot), __longlong_as_double(bits));
  else atomicAdd(slot, static_cast<unsigned long long>(bits));
}

// ---- build: generic path (global table) ---------------------------------------------------------
// One row per lane.  mask: selected iff (mask_valid ? valid : 1) && bit  (Filter DROP semantics).
__global__ __launch_bounds__(kThreads) void k_groupby_build(KeySpec ks, ValueCol v, int has_values,
                                                            const uint8_t* m_data, const uint8_t* m_valid, int64_t m_off,
                                                            int64_t n, GroupTable t, unsigned int* overflow) {
  for (int64_t row = int64_t(blockIdx.x) * kThreads + threadIdx.x; row < n; row += int64_t(gridDim.x) * kThreads) {
    if (m_data) {
      if (!GetBit(m_data, m_off + row)) continue;
      if (m_valid && !GetBit(m_valid, m_off + row)) continue;
    }
    uint64_t x;
    int64_t idx;
    if (EncodeKey(ks, row, &x)) idx = static_cast<int64_t>(t.mask) + 1;
    else idx = FindOrInsert(t, Fmix64(x), true);
    if (idx < 0) { *overflow = 1; continue; }
    atomicMin(&t.first[idx], static_cast<uint32_t>(row));
    if (has_values) {
      const bool valid = v.valid ? GetBit(v.valid, v.off + row) : true;
      if (valid) {
        bool is_double;
        const uint64_t bits = LoadAccum(v, row, &is_double);
        Accumulate(&t.sums[idx], bits, is_double);
        atomicAdd(&t.counts[idx], 1ull);
      }
    }
  }
}

// ---- first-appearance numbering -----------------------------------------------------------------
// Set bit first_row for every group that needs an id.
// scratch of the heavy-hitter path (see "heavy hitters" below)
struct HotMem {
  uint32_t* cand;          // [nwg][kHotPerWg] tags reported by the local attempt
  unsigned int* cand_n;    // [nwg] how many; bit 31: the null key is hot in this row range
  uint32_t* tags;          // [kHotCap] the hot set
  unsigned int* count;     // [0] keys in the set, [1] null key is hot
  unsigned long long* part_sum;  // [nwg][kHotCap + 1] partial sums (index kHotCap: the null key)
  unsigned int* part_cnt;        // [nwg][kHotCap + 1]
  uint32_t* part_first;          // [nwg][kHotCap + 1]
  uint8_t* alt_mask;       // selection mask without the hot rows, bit offset 0
  uint32_t* wg_stats;      // [nwg][4] what a workgroup of k_local_aggregate_vec saw: smallest / largest key of its table (the scatter's key order), a value that is not an int32, -
};
__device__ void InjectHotGroups(const HotMem& hot, int nwg, int k_bits, int acc_double, const GroupTable& t,
                                unsigned long long* bitmap, int64_t gtid, int64_t gthreads);
__global__ __launch_bounds__(kThreads) void k_mark_first(GroupTable t, int64_t nslots, int only_unassigned,
                                                         unsigned long long* bitmap, const unsigned int* skip_if,
                                                         HotMem hot, const unsigned int* use_hot, int nwg,
                                                         int k_bits, int acc_double) {
  // heavy hitters kept out of the partitions join the table here (see "heavy hitters" below)
  if (use_hot && *use_hot)
    InjectHotGroups(hot, nwg, k_bits, acc_double, t, bitmap, int64_t(blockIdx.x) * kThreads + threadIdx.x,
                    int64_t(gridDim.x) * kThreads);
  if (skip_if && *skip_if) return;  // the groups were marked by the kernel that built them
  for (int64_t s = int64_t(blockIdx.x) * kThreads + threadIdx.x; s < nslots; s += int64_t(gridDim.x) * kThreads) {
    const uint32_t fr = t.first[s];
    if (fr == kNoRow) continue;
    if (only_unassigned && t.ids[s] != kNoRow) continue;
    atomicOr(&bitmap[fr >> 6], 1ull << (fr & 63));
  }
}
// counts[b] = popcount of 1024 consecutive bitmap words
__global__ __launch_bounds__(kThreads) void k_popc_blocks(const unsigned long long* bitmap, int64_t nwords, uint32_t* counts) {
  __shared__ uint32_t s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * 1024;
  uint32_t c = 0;
  for (int k = 0; k < 4; ++k) {
    const int64_t w = base + k * kThreads + threadIdx.x;
    if (w < nwords) c += __popcll(bitmap[w]);
  }
  c = static_cast<uint32_t>(WaveReduceSum64(c));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_sum, c);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_sum;
}
// prefix[w] = number of set bits in words [0, w)
__global__ __launch_bounds__(kThreads) void k_word_prefix(const unsigned long long* bitmap, int64_t nwords,
                                                          const int64_t* block_offsets, uint32_t* prefix,
                                                          unsigned int* status, unsigned int* host_status) {
  // status words (see RankFirstRows): store the total, then mirror all of them into mapped host
  // memory -- the host waits for an event behind this kernel instead of paying a copy launch.
  if (status && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long total = block_offsets[gridDim.x];
    *reinterpret_cast<long long*>(status + 6) = total;
    if (host_status) {
      for (int i = 0; i < 6; ++i) host_status[i] = status[i];
      host_status[6] = static_cast<unsigned int>(total);
      host_status[7] = static_cast<unsigned int>(static_cast<unsigned long long>(total) >> 32);
      for (int i = 8; i < 12; ++i) host_status[i] = status[i];
    }
  }
  __shared__ uint32_t s_wave[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = int64_t(blockIdx.x) * 1024;
  uint32_t carry = static_cast<uint32_t>(block_offsets[blockIdx.x]);
  for (int k = 0; k < 4; ++k) {
    const int64_t w = base + k * kThreads + threadIdx.x;
    const uint32_t pc = w < nwords ? __popcll(bitmap[w]) : 0;
    const uint32_t incl = WaveInclusiveScan(pc);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int j = 0; j < 4; ++j) { if (j < wave) woff += s_wave[j]; total += s_wave[j]; }
    if (w < nwords) prefix[w] = carry + woff + incl - pc;
    carry += total;
    __syncthreads();
  }
}

// k_word_prefix without the scan launch in front: a block adds up the counts of the blocks before it itself (at
// most kFusedScanBlocks of them, 16 KB out of the L2 -- cheaper than one more dependent launch); the last block
// knows the total and mirrors the status words.
constexpr int64_t kFusedScanBlocks = 4096;
__global__ __launch_bounds__(kThreads) void k_word_prefix_scan(const unsigned long long* bitmap, int64_t nwords,
                                                               const uint32_t* block_counts, uint32_t* prefix,
                                                               unsigned int* status, unsigned int* host_status) {
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_before[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t mine = 0;
  for (int b = threadIdx.x; b < static_cast<int>(blockIdx.x); b += kThreads) mine += block_counts[b];
  mine = static_cast<uint32_t>(WaveReduceSum64(mine));
  if (lane == 0) s_before[wave] = mine;
  __syncthreads();
  uint32_t carry = s_before[0] + s_before[1] + s_before[2] + s_before[3];
  if (status && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const long long total = static_cast<long long>(carry) + block_counts[blockIdx.x];
    *reinterpret_cast<long long*>(status + 6) = total;
    if (host_status) {
      for (int i = 0; i < 6; ++i) host_status[i] = status[i];
      host_status[6] = static_cast<unsigned int>(total);
      host_status[7] = static_cast<unsigned int>(static_cast<unsigned long long>(total) >> 32);
      for (int i = 8; i < 12; ++i) host_status[i] = status[i];
    }
  }
  const int64_t base = int64_t(blockIdx.x) * 1024;
  for (int k = 0; k < 4; ++k) {
    const int64_t w = base + k * kThreads + threadIdx.x;
    const uint32_t pc = w < nwords ? __popcll(bitmap[w]) : 0;
    const uint32_t incl = WaveInclusiveScan(pc);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int j = 0; j < 4; ++j) { if (j < wave) woff += s_wave[j]; total += s_wave[j]; }
    if (w < nwords) prefix[w] = carry + woff + incl - pc;
    carry += total;
    __syncthreads();
  }
}

struct EmitArgs {
  GroupTable t; int64_t nslots; const unsigned long long* bitmap; const uint32_t* prefix;
  int64_t capacity; uint32_t id_base; int only_unassigned;
  unsigned long long* gx;     // key word per id
  unsigned long long* sums;   // may be null
  long long* counts;          // may be null
  uint32_t* first_rows;       // may be null
  uint32_t* wide_null_id;     // id of the null-key group of a single 64-bit key (kNoRow if none)
  const unsigned int* skip_if; // optional device word: non-zero = the table is not valid, emit nothing
  ulonglong4* packed;          // optional: ONE 32-byte record {key word, sum, count, first row} per id instead of
                               // the four arrays above (one full-sector scattered store instead of four partial ones)
};
__global__ __launch_bounds__(kThreads) void k_emit_groups(EmitArgs a) {
  if (a.skip_if && *a.skip_if) return;
  // Four slots per thread and trip, every stage's loads issued together: one slot at a time is a chain of three
  // dependent memory round trips per trip (first row -> bitmap word / prefix -> store), 29 us for 2 M slots.
  constexpr int U = 4;
  const int64_t stride = int64_t(gridDim.x) * kThreads;
  const uint64_t T = a.t.mask + 1;
  for (int64_t s0 = int64_t(blockIdx.x) * kThreads + threadIdx.x; s0 < a.nslots; s0 += U * stride) {
    uint32_t fr[U], pre[U];
    uint64_t hk[U], sum[U], cnt[U], word[U];
    bool live[U];
    uint32_t had_id[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // (raw loads only: a value looked at here would be waited for before the next slot's loads)
      const int64_t s = s0 + u * stride;
      const int64_t sc = s < a.nslots ? s : a.nslots - 1;
      fr[u] = a.t.first[sc];
      had_id[u] = kNoRow;
      if (a.only_unassigned) had_id[u] = a.t.ids[sc];  // (uniform)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = s0 + u * stride;
      live[u] = s < a.nslots && fr[u] != kNoRow && had_id[u] == kNoRow;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {  // second stage: the live slots' columns (half the slots of a directly indexed table are empty) and their ranks' words
      const int64_t s = s0 + u * stride;
      const uint32_t w = live[u] ? fr[u] >> 6 : 0u;
      word[u] = a.bitmap[w];
      pre[u] = a.prefix[w];
      hk[u] = 0; sum[u] = 0; cnt[u] = 0;
      if (live[u]) {
        hk[u] = a.t.hkeys[s];
        if (a.packed || a.sums || a.counts) { sum[u] = a.t.sums[s]; cnt[u] = a.t.counts[s]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) continue;
      const int64_t s = s0 + u * stride;
      const uint32_t id = a.id_base + pre[u] + __popcll(word[u] & ((1ull << (fr[u] & 63)) - 1));
      if (a.t.ids) { a.t.ids[s] = id; a.t.first[s] = kNoRow; }
      if (static_cast<int64_t>(id) >= a.capacity) continue;
      uint64_t x;
      if (static_cast<uint64_t>(s) == T) { x = 0; *a.wide_null_id = id; }
      else if (static_cast<uint64_t>(s) == T + 1) x = Fmix64Inv(kEmpty);
      else x = Fmix64Inv(hk[u]);
      if (a.packed) {
        a.packed[id] = make_ulonglong4(x, sum[u], cnt[u], fr[u]);
        continue;
      }
      a.gx[id] = x;
      if (a.sums) a.sums[id] = sum[u];
      if (a.counts) a.counts[id] = static_cast<long long>(cnt[u]);
      if (a.first_rows) a.first_rows[id] = fr[u];
    }
  }
}

// Decode key column `col` of ids [0, G): values + validity bitmap (one word per wave).
__global__ __launch_bounds__(kThreads) void k_decode_keys(const unsigned long long* gx, int64_t G, int bits, int shift,
                                                          int null_bit /* <0: none */, uint32_t wide_null_id,
                                                          void* out_values, uint64_t* out_validity,
                                                          unsigned long long* null_count,
                                                          const unsigned int* status /* optional, see RankFirstRows */) {
  if (status) {  // group count and wide-null id still live on the device; G is the capacity here
    if (status[0]) return;
    const long long total = *reinterpret_cast<const long long*>(status + 6);
    if (total < G) G = total;
    wide_null_id = status[4];
  }
  const int lane = threadIdx.x & 63;
  const int64_t ground = (G + 63) & ~int64_t(63);
  for (int64_t g = int64_t(blockIdx.x) * kThreads + threadIdx.x; g < ground; g += int64_t(gridDim.x) * kThreads) {
    bool valid = false;
    uint64_t v = 0;
    if (g < G) {
      const uint64_t x = gx[g];
      valid = (null_bit >= 0) ? !((x >> null_bit) & 1) : (static_cast<uint32_t>(g) != wide_null_id);
      v = bits >= 64 ? x : ((x >> shift) & ((uint64_t(1) << bits) - 1));
      if (!valid) v = 0;
    }
    const uint64_t vm = __ballot(valid);
    if (bits == 1) {
      const uint64_t bm = __ballot(v != 0);
      if (lane == 0) reinterpret_cast<uint64_t*>(out_values)[g >> 6] = bm;
    } else if (g < G) {
      switch (bits) {
        case 8: reinterpret_cast<uint8_t*>(out_values)[g] = static_cast<uint8_t>(v); break;
        case 16: reinterpret_cast<uint16_t*>(out_values)[g] = static_cast<uint16_t>(v); break;
        case 32: reinterpret_cast<uint32_t*>(out_values)[g] = static_cast<uint32_t>(v); break;
        default: reinterpret_cast<uint64_t*>(out_values)[g] = v; break;
      }
    }
    if (lane == 0) {
      out_validity[g >> 6] = vm;
      const int64_t live = G - g >= 64 ? 64 : G - g;
      const int nulls = static_cast<int>(live) - __popcll(vm);
      if (nulls) atomicAdd(null_count, static_cast<unsigned long long>(nulls));
    }
  }
}

// Packed group records (k_emit_groups) -> the caller's arrays, all coalesced; decodes the single
// key column like k_decode_keys.  The group count comes from the status words (see RankFirstRows).
// out_bits / bias: a range-compressed int64 key column comes back as x + bias in 64 bits.
__global__ __launch_bounds__(kThreads) void k_unpack_groups(const ulonglong4* packed, int64_t bound, int bits, int out_bits,
                                                            unsigned long long bias,
                                                            int null_bit /* <0: none */, unsigned long long* out_sums,
                                                            long long* out_counts, uint32_t* out_first_rows,
                                                            void* out_keys, uint64_t* out_validity,
                                                            const unsigned int* status) {
  if (status[0]) return;
  int64_t G = *reinterpret_cast<const long long*>(status + 6);
  if (G > bound) G = bound;
  const uint32_t wide_null_id = status[4];
  const int lane = threadIdx.x & 63;
  const int64_t ground = (G + 63) & ~int64_t(63);
  for (int64_t g = int64_t(blockIdx.x) * kThreads + threadIdx.x; g < ground; g += int64_t(gridDim.x) * kThreads) {
    bool valid = false;
    uint64_t v = 0;
    if (g < G) {
      const ulonglong4 r = packed[g];
      if (out_sums) out_sums[g] = r.y;
      if (out_counts) out_counts[g] = static_cast<long long>(r.z);
      if (out_first_rows) out_first_rows[g] = static_cast<uint32_t>(r.w);
      const uint64_t x = r.x;
      valid = (null_bit >= 0) ? !((x >> null_bit) & 1) : (static_cast<uint32_t>(g) != wide_null_id);
      v = bits >= 64 ? x : (x & ((uint64_t(1) << bits) - 1));
      v += bias;
      if (!valid) v = 0;
      switch (out_bits) {
        case 1: break;  // packed below
        case 8: reinterpret_cast<uint8_t*>(out_keys)[g] = static_cast<uint8_t>(v); break;
        case 16: reinterpret_cast<uint16_t*>(out_keys)[g] = static_cast<uint16_t>(v); break;
        case 32: reinterpret_cast<uint32_t*>(out_keys)[g] = static_cast<uint32_t>(v); break;
        default: reinterpret_cast<uint64_t*>(out_keys)[g] = v; break;
      }
    }
    // whole waves reach this point together (the loop bound is a multiple of 64)
    const uint64_t vm = __ballot(valid);
    const uint64_t bm = __ballot(v != 0);
    if (lane == 0) {
      out_validity[g >> 6] = vm;
      if (out_bits == 1) reinterpret_cast<uint64_t*>(out_keys)[g >> 6] = bm;
    }
  }
}

// Grouper pass 2: out_ids[row] = ids[slot(key(row))]
__global__ __launch_bounds__(kThreads) void k_lookup_ids(KeySpec ks, int64_t n, GroupTable t, uint32_t* out_ids,
                                                         unsigned long long* misses /* slotted counter or null */) {
  unsigned long long missed = 0;
  for (int64_t row = int64_t(blockIdx.x) * kThreads + threadIdx.x; row < n; row += int64_t(gridDim.x) * kThreads) {
    uint64_t x;
    int64_t idx;
    if (EncodeKey(ks, row, &x)) idx = static_cast<int64_t>(t.mask) + 1;
    else idx = FindOrInsert(t, Fmix64(x), false);
    const uint32_t id = idx >= 0 ? t.ids[idx] : kNoRow;
    out_ids[row] = id;
    missed += id == kNoRow;
  }
  if (misses) {
    missed = WaveReduceSum64(missed);
    if ((threadIdx.x & 63) == 0 && missed) atomicAdd(CounterSlot(misses), missed);
  }
}
// Grouper growth: re-insert the known groups (x per id) into a fresh table.
__global__ __launch_bounds__(kThreads) void k_reinsert(const unsigned long long* gx, uint32_t G, uint32_t wide_null_id,
                                                       GroupTable t) {
  for (uint32_t g = blockIdx.x * kThreads + threadIdx.x; g < G; g += gridDim.x * kThreads) {
    int64_t idx;
    if (g == wide_null_id) idx = static_cast<int64_t>(t.mask) + 1;
    else idx = FindOrInsert(t, Fmix64(gx[g]), true);
    if (idx >= 0) t.ids[idx] = g;
  }
}

// ---- hash_sum / hash_count / hash_mean over given ids ------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_hash_agg(ValueCol v, const uint32_t* ids, int64_t n, int func, int count_mode,
                                                       unsigned long long* sums, unsigned long long* counts) {
  for (int64_t row = int64_t(blockIdx.x) * kThreads + threadIdx.x; row < n; row += int64_t(gridDim.x) * kThreads) {
    const bool valid = v.valid ? GetBit(v.valid, v.off + row) : true;
    const uint32_t g = ids[row];
    if (func == AMDARROW_HASH_COUNT) {
      if (count_mode == AMDARROW_COUNT_NULL ? !valid : valid) atomicAdd(&counts[g], 1ull);  // :485-503
    } else if (valid) {                                                                     // :539-545
      bool is_double;
      const uint64_t bits = LoadAccum(v, row, &is_double);
      Accumulate(&sums[g], bits, is_double);
      atomicAdd(&counts[g], 1ull);
    }
  }
}
// Finalize: validity = count > 0 (GroupedSumImpl::Finalize :581-604); mean = double(sum)/count.
__global__ __launch_bounds__(kThreads) void k_agg_finalize(int func, int acc_kind /*0 u64, 1 i64, 2 f64*/, int64_t G,
                                                           unsigned long long* values, const unsigned long long* counts,
                                                           uint64_t* out_validity, unsigned long long* null_count) {
  const int lane = threadIdx.x & 63;
  const int64_t ground = (G + 63) & ~int64_t(63);
  for (int64_t g = int64_t(blockIdx.x) * kThreads + threadIdx.x; g < ground; g += int64_t(gridDim.x) * kThreads) {
    bool valid = false;
    if (g < G) {
      const unsigned long long c = counts[g];
      valid = func == AMDARROW_HASH_COUNT ? true : c > 0;
      if (func == AMDARROW_HASH_COUNT) values[g] = c;
      else if (func == AMDARROW_HASH_MEAN) {
        double m = 0.0;
        if (c > 0) {
          const unsigned long long s = values[g];
          const double sd = acc_kind == 2 ? __longlong_as_double(s)
                          : acc_kind == 1 ? static_cast<double>(static_cast<long long>(s)) : static_cast<double>(s);
          m = sd / static_cast<double>(c);  // MeanImpl::Finalize, aggregate_basic_internal.h:94-103
        }
        values[g] = __double_as_longlong(m);
      }
    }
    const uint64_t vm = __ballot(valid);
    if (lane == 0) {
      out_validity[g >> 6] = vm;
      const int64_t live = G - g >= 64 ? 64 : G - g;
      const int nulls = static_cast<int>(live) - __popcll(vm);
      if (nulls) atomicAdd(null_count, static_cast<unsigned long long>(nulls));
    }
  }
}

// ---- host helpers ---------------------------------------------------------------------------------
int LaunchScanU32(amdarrow_ctx* ctx, const uint32_t* counts, int64_t n, int64_t* offsets);  // selection.hip

static uint64_t NextPow2(uint64_t v) {
  uint64_t p = 1024;
  while (p < v) p <<= 1;
  return p;
}

static int MakeKeySpec(int nkeys, const amdarrow_array* keys, const int* types, KeySpec* ks) {
  if (nkeys < 1 || nkeys > kMaxKeys)
    return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: 1.." + std::to_string(kMaxKeys) + " key columns supported");
  memset(ks, 0, sizeof *ks);
  ks->nkeys = nkeys;
  int bits = 0;
  for (int k = 0; k < nkeys; ++k) {
    const int t = keys ? keys[k].type_id : types[k];
    const int bw = BitWidth(t);
    if (bw < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "Keys of this type");  // hash_aggregate.cc:355
    ks->col[k].bits = bw;
    ks->col[k].shift = bits;
    if (keys) { ks->col[k].data = keys[k].values; ks->col[k].valid = keys[k].validity; ks->col[k].off = keys[k].offset; }
    bits += bw;
  }
  ks->total_bits = bits;
  if (nkeys == 1 && bits == 64) ks->wide_null_slot = 1;
  else if (bits + nkeys > 64)
    return SetError(AMDARROW_NOT_IMPLEMENTED, "group-by: key columns wider than 64 bits in total (incl. one null flag each)");
  return AMDARROW_OK;
}

struct TableMem { size_t hkeys, first, sums, counts, ids; };
static size_t CarveTable(Carver* c, uint64_t T, bool with_ids, TableMem* m) {
  const uint64_t n = T + 2;
  m->hkeys = c->Take(8 * n); m->first = c->Take(4 * n); m->sums = c->Take(8 * n); m->counts = c->Take(8 * n);
  m->ids = with_ids ? c->Take(4 * n) : 0;
  return c->off;
}
static GroupTable BindTable(char* base, const TableMem& m, uint64_t T, bool with_ids) {
  GroupTable t;
  t.hkeys = reinterpret_cast<unsigned long long*>(base + m.hkeys);
  t.first = reinterpret_cast<uint32_t*>(base + m.first);
  t.sums = reinterpret_cast<unsigned long long*>(base + m.sums);
  t.counts = reinterpret_cast<unsigned long long*>(base + m.counts);
  t.ids = with_ids ? reinterpret_cast<uint32_t*>(base + m.ids) : nullptr;
  t.mask = T - 1;
  return t;
}
static int ClearTable(amdarrow_ctx* ctx, const GroupTable& t, bool with_ids) {
  const uint64_t n = t.mask + 3;
  AA_HIP(hipMemsetAsync(t.hkeys, 0xFF, 8 * n, ctx->stream));
  AA_HIP(hipMemsetAsync(t.first, 0xFF, 4 * n, ctx->stream));
  AA_HIP(hipMemsetAsync(t.sums, 0, 8 * n, ctx->stream));
  AA_HIP(hipMemsetAsync(t.counts, 0, 8 * n, ctx->stream));
  if (with_ids) AA_HIP(hipMemsetAsync(t.ids, 0xFF, 4 * n, ctx->stream));
  return AMDARROW_OK;
}

// bitmap(first rows) -> per-word prefix; returns number of marked groups in *total.
struct RankMem { unsigned long long* bitmap; uint32_t* prefix; uint32_t* block_counts; int64_t* block_offsets; int64_t nwords; int64_t nblocks; };
static void CarveRank(Carver* c, int64_t n_rows, size_t off[4]) {
  const int64_t nwords = (n_rows + 63) / 64 + 1;
  const int64_t nblocks = (nwords + 1023) / 1024;
  off[0] = c->Take(8 * nwords); off[1] = c->Take(4 * nwords); off[2] = c->Take(4 * nblocks); off[3] = c->Take(8 * (nblocks + 1));
}
static RankMem BindRank(char* base, const size_t off[4], int64_t n_rows) {
  RankMem r;
  r.nwords = (n_rows + 63) / 64 + 1;
  r.nblocks = (r.nwords + 1023) / 1024;
  r.bitmap = reinterpret_cast<unsigned long long*>(base + off[0]);
  r.prefix = reinterpret_cast<uint32_t*>(base + off[1]);
  r.block_counts = reinterpret_cast<uint32_t*>(base + off[2]);
  r.block_offsets = reinterpret_cast<int64_t*>(base + off[3]);
  return r;
}
// Wait for the status words queued by RankFirstRows(status_dev != nullptr).
static int AwaitStatus(amdarrow_ctx* ctx, unsigned int (&status)[12]) {
  AA_HIP(hipEventSynchronize(ctx->status_event));
  memcpy(status, static_cast<const char*>(ctx->pinned) + kStatusOffset, sizeof(status));
  return AMDARROW_OK;
}

// mark_mode: 0 = clear the bitmap and mark every group's first row here; 1 = already marked by
// the kernel that built the table; 2 = bitmap already cleared, mark here unless status[1] is
// set on the device (then the partition path built the table and marked on the way); 3 = bitmap
// already cleared, mark here.
// `status_dev` (optional): 12 device words {overflow, local-abort, null count x2, wide-null id, -,
// total lo, total hi, smallest / largest selected key (bin scatter), a value wider than 32 bits was seen, -};
// k_word_prefix stores the total there and mirrors the words into mapped host memory.  With status_dev the caller collects them with AwaitStatus (after queueing
// whatever else does not need them on the host).
struct HotInject { HotMem hot; int nwg, k_bits, acc_double; };  // heavy hitters to add while marking (mode 2)
static int RankFirstRows(amdarrow_ctx* ctx, const GroupTable& t, int only_unassigned, const RankMem& r, int64_t* total,
                         int mark_mode = 0, unsigned int* status_dev = nullptr, const HotInject* inject = nullptr) {
  const int64_t nslots = static_cast<int64_t>(t.mask) + 3;
  if (mark_mode != 1) {
    if (mark_mode == 0) AA_HIP(hipMemsetAsync(r.bitmap, 0, 8 * r.nwords, ctx->stream));
    hipLaunchKernelGGL(k_mark_first, dim3(GridFor(nslots, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, t,
                       nslots, only_unassigned, r.bitmap, mark_mode == 2 ? status_dev + 1 : nullptr,
                       inject ? inject->hot : HotMem{}, inject ? status_dev + 5 : nullptr, inject ? inject->nwg : 0,
                       inject ? inject->k_bits : 0, inject ? inject->acc_double : 0);
  }
  hipLaunchKernelGGL(k_popc_blocks, dim3(static_cast<unsigned>(r.nblocks)), dim3(kThreads), 0, ctx->stream, r.bitmap,
                     r.nwords, r.block_counts);
  AA_HIP(hipGetLastError());
  if (status_dev && r.nblocks <= kFusedScanBlocks && !getenv("AMDARROW_GROUPBY_SCAN_LAUNCH")) {
    hipLaunchKernelGGL(k_word_prefix_scan, dim3(static_cast<unsigned>(r.nblocks)), dim3(kThreads), 0, ctx->stream, r.bitmap,
                       r.nwords, r.block_counts, r.prefix, status_dev,
                       reinterpret_cast<unsigned int*>(static_cast<char*>(ctx->pinned_dev) + kStatusOffset));
    AA_HIP(hipGetLastError());
    AA_HIP(hipEventRecord(ctx->status_event, ctx->stream));
    return AMDARROW_OK;
  }
  AA_TRY(LaunchScanU32(ctx, r.block_counts, r.nblocks, r.block_offsets));
  hipLaunchKernelGGL(k_word_prefix, dim3(static_cast<unsigned>(r.nblocks)), dim3(kThreads), 0, ctx->stream, r.bitmap,
                     r.nwords, r.block_offsets, r.prefix, status_dev,
                     status_dev ? reinterpret_cast<unsigned int*>(static_cast<char*>(ctx->pinned_dev) + kStatusOffset) : nullptr);
  AA_HIP(hipGetLastError());
  if (status_dev) {
    AA_HIP(hipEventRecord(ctx->status_event, ctx->stream));
    return AMDARROW_OK;
  }
  AA_TRY(ReadBack(ctx, r.block_offsets + r.nblocks, 8, total));
  return AMDARROW_OK;
}

static int AccKind(int t) { return IsFloat(t) ? 2 : (IsSignedInt(t) ? 1 : 0); }

// ==== fast path: hash-partition to LDS-sized pieces, aggregate each piece in LDS =================
// Global atomics run at the memory side on MI355X (~20 G scattered atomics/s chip-wide), far
// below what the 12 B/row input stream needs.  So for large inputs the rows are first
// partitioned by the top bits of h into kP pieces (one 16-byte record {x, row|flags, value}
// per selected row), and each piece is then aggregated by ONE workgroup in an open-addressed
// LDS table (LDS atomics only).  A piece that does not fit its LDS table raises `overflow`
// and the caller re-runs the generic path, so the fast path is an optimisation, never a
// restriction.
constexpr int kPBits = 9;
constexpr int kP = 1 << kPBits;          // partitions
constexpr int kSlots = 4096;             // LDS table slots per partition (96 KiB)
constexpr int kSlotLimit = 3584;         // groups per partition before giving up (87.5 % load)
constexpr int kPartThreads = 1024;
constexpr int kPartRows = 8192;          // rows per tile of the partition pass (8 per thread)
constexpr uint32_t kRecKeyNull = 0x80000000u, kRecValValid = 0x40000000u, kRecRowMask = 0x3FFFFFFFu;

// Several key columns packed into one key word (amdarrow_filter_groupby_multi): column c contributes
// code = value - bias (or null_code for a null) at bit `shift`; the codes of all columns fit 32 bits.
// k_pack_keys materialises the packed column once (12 bytes per row for two int32 columns) and the
// single-key machinery -- the specialised scatter included -- runs on it unchanged.
constexpr int kMaxPackedKeys = 4;
struct KeyPack {
  int ncols;
  const void* data[kMaxPackedKeys]; const uint8_t* valid[kMaxPackedKeys]; int64_t off[kMaxPackedKeys];
  int load_bits[kMaxPackedKeys]; int is_signed[kMaxPackedKeys]; int shift[kMaxPackedKeys]; int bits[kMaxPackedKeys];
  uint32_t null_code[kMaxPackedKeys]; uint64_t bias[kMaxPackedKeys];
  // hinted ranges (k_pack_keys_vec32): a value with value - bias > span[c] sets *out_of_range (null: not checked)
  uint32_t span[kMaxPackedKeys]; unsigned int* out_of_range;
};
struct FastArgs {
  const void* k_data; const uint8_t* k_valid; int64_t k_off; int k_bits;
  int k_load_bits;    // width of the key column in memory (== k_bits, or 64 for a range-compressed int64 column)
  uint64_t k_bias;    // subtracted from a 64-bit key: x = key - min(key) fits k_bits = 32
  ValueCol v; int has_values;
  const uint8_t* m_data; const uint8_t* m_valid; int64_t m_off;
  int64_t n; int64_t rows_per_block; int nblocks;
  // Sub-problems (more than ~1.8 M groups): the top sub_bits hash bits select one of 2^sub_bits
  // passes over the input, each with its own 512 partitions (the NEXT 9 bits) and table slice.
  int sub_bits; uint32_t sub_id;
  // bin scatter only: keys are compared / partitioned as k ^ k_flip (0x80000000 for signed keys: value order as
  // unsigned order); the smallest / largest selected key lands in key_minmax[0 / 1] (atomicMin / atomicMax)
  uint32_t k_flip; unsigned int* key_minmax;
  // bin scatter: optional device word set when a selected, valid INTEGER value does not fit 32 bits (what a later
  // call of the same shape needs to know before it may carry the values as 32-bit words, see the compact records)
  unsigned int* val_wide;
  // compact records only: (key ^ k_flip) >> kPBits - c_base is the record's 12-bit slot offset
  uint32_t c_base;
};
__device__ __forceinline__ uint32_t PartOf(uint32_t h, int sub_bits) { return (h << sub_bits) >> (32 - kPBits); }
__device__ __forceinline__ bool InSub(uint32_t h, int sub_bits, uint32_t sub_id) {
  return sub_bits == 0 || (h >> (32 - sub_bits)) == sub_id;
}

__device__ __forceinline__ bool FastSelected(const FastArgs& a, int64_t row) {
  if (!a.m_data) return true;
  if (!GetBit(a.m_data, a.m_off + row)) return false;
  return !a.m_valid || GetBit(a.m_valid, a.m_off + row);
}
// Selection mask of the 64 consecutive rows starting at `row0` (one word per wave, every
// lane loads the same address): bit i set <=> row0 + i is selected and < hi.
__device__ __forceinline__ uint64_t FastMaskWord(const FastArgs& a, int64_t row0, int64_t hi) {
  const int64_t left = hi - row0;
  if (left <= 0) return 0;
  const int nbits = left >= 64 ? 64 : static_cast<int>(left);
  uint64_t w = nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0);
  if (a.m_data) {
    w = LoadBits64(a.m_data, a.m_off + row0, nbits);
    if (a.m_valid) w &= LoadBits64(a.m_valid, a.m_off + row0, nbits);
  }
  return w;
}
__device__ __forceinline__ uint64_t PackedKey(const KeyPack& p, int64_t row) {
  uint64_t x = 0;
#pragma unroll
  for (int c = 0; c < kMaxPackedKeys; ++c) {
    if (c >= p.ncols) break;
    const int64_t i = p.off[c] + row;
    uint64_t code;
    if (p.valid[c] && !GetBit(p.valid[c], i)) {
      code = p.null_code[c];
    } else {
      uint64_t raw;
      switch (p.load_bits[c]) {
        case 1: raw = GetBit(reinterpret_cast<const uint8_t*>(p.data[c]), i); break;
        case 8: raw = p.is_signed[c] ? uint64_t(int64_t(reinterpret_cast<const int8_t*>(p.data[c])[i])) : reinterpret_cast<const uint8_t*>(p.data[c])[i]; break;
        case 16: raw = p.is_signed[c] ? uint64_t(int64_t(reinterpret_cast<const int16_t*>(p.data[c])[i])) : reinterpret_cast<const uint16_t*>(p.data[c])[i]; break;
        case 32: raw = p.is_signed[c] ? uint64_t(int64_t(reinterpret_cast<const int32_t*>(p.data[c])[i])) : reinterpret_cast<const uint32_t*>(p.data[c])[i]; break;
        default: raw = reinterpret_cast<const uint64_t*>(p.data[c])[i]; break;
      }
      code = raw - p.bias[c];
    }
    x |= code << p.shift[c];
  }
  return x;
}
__device__ __forceinline__ uint64_t FastKeyBits(const FastArgs& a, int64_t row) {
  switch (a.k_load_bits) {
    case 1: return GetBit(reinterpret_cast<const uint8_t*>(a.k_data), a.k_off + row);
    case 8: return reinterpret_cast<const uint8_t*>(a.k_data)[a.k_off + row];
    case 16: return reinterpret_cast<const uint16_t*>(a.k_data)[a.k_off + row];
    case 64: return reinterpret_cast<const uint64_t*>(a.k_data)[a.k_off + row] - a.k_bias;
    default: return reinterpret_cast<const uint32_t*>(a.k_data)[a.k_off + row];
  }
}

// hist[p * nblocks + b] = selected rows of row-range b that hash to partition p
__global__ __launch_bounds__(kPartThreads) void k_part_hist(FastArgs a, uint32_t* hist) {
  __shared__ unsigned int s_cnt[kP];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < kP) s_cnt[tid] = 0;
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  constexpr int kIters = kPartRows / kPartThreads;  // 8 independent loads in flight per thread
  for (int64_t tile0 = lo; tile0 < hi; tile0 += kPartRows) {
    uint64_t x[kIters];
    bool sel[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t row = tile0 + j * kPartThreads + tid;
      const uint64_t w = FastMaskWord(a, row - lane, hi);
      sel[j] = (w >> lane) & 1;
      x[j] = 0;
      if (sel[j]) {
        const bool kn = a.k_valid && !GetBit(a.k_valid, a.k_off + row);  // (64-bit keys come here without nulls)
        x[j] = kn ? (uint64_t(1) << (a.k_bits & 63)) : FastKeyBits(a, row);
      }
    }
#pragma unroll
    for (int j = 0; j < kIters; ++j)
      if (sel[j]) atomicAdd(&s_cnt[Fmix64(x[j]) >> (64 - kPBits)], 1u);
  }
  __syncthreads();
  if (tid < kP) hist[int64_t(tid) * a.nblocks + blockIdx.x] = s_cnt[tid];
}

// Scatter the selected rows of row-range b into the partition regions.  Per 8192-row tile the
// records are staged in LDS grouped by partition, so every partition leaves as one contiguous
// run; successive tiles of a block append to the same regions, which keeps the partially
// written lines in this CU's L2 until they are complete.
// WIDE: 64-bit keys that do not range-compress (ids, hashes).  The record keeps its 16 bytes (x = the key's
// low word) and the key's high word travels in a parallel array `hi_words`, same order; tiles are half as
// long so that both stages fit the LDS.
template <bool WIDE>
__global__ __launch_bounds__(kPartThreads) void k_part_scatter(FastArgs a, const uint32_t* offsets, uint4* records, uint32_t* hi_words) {
  constexpr int kTileRows = WIDE ? kPartRows / 2 : kPartRows;
  __shared__ unsigned int s_cnt[kP];
  __shared__ uint32_t s_off[kP];
  __shared__ uint32_t s_cur[kP];
  __shared__ uint32_t s_scan[8];
  __shared__ uint4 s_rec[kTileRows];
  __shared__ uint32_t s_hi[WIDE ? kTileRows : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kIters = kTileRows / kPartThreads;  // 8 (4 when WIDE)
  if (tid < kP) { s_cnt[tid] = 0; s_cur[tid] = offsets[int64_t(tid) * a.nblocks + blockIdx.x]; }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  for (int64_t tile0 = lo; tile0 < hi; tile0 += kTileRows) {
    uint32_t rx[kIters], rflags[kIters], rrank[kIters];
    uint32_t rhi[WIDE ? kIters : 1];
    uint64_t rval[kIters];
    int rpart[kIters];
    // phase 1a: issue every load of the tile (mask word, key, value) before anything waits
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t row = tile0 + j * kPartThreads + tid;
      const uint64_t w = FastMaskWord(a, row - lane, hi);
      rpart[j] = -1;
      rx[j] = 0; rflags[j] = 0; rval[j] = 0;
      if ((w >> lane) & 1) {
        rpart[j] = 0;
        const bool kn = a.k_valid && !GetBit(a.k_valid, a.k_off + row);
        const uint64_t kx = kn ? 0u : FastKeyBits(a, row);
        rx[j] = static_cast<uint32_t>(kx);
        if (WIDE) rhi[WIDE ? j : 0] = static_cast<uint32_t>(kx >> 32);
        uint32_t f = static_cast<uint32_t>(row) | (kn ? kRecKeyNull : 0u);
        if (a.has_values && (!a.v.valid || GetBit(a.v.valid, a.v.off + row))) {
          bool is_double;
          rval[j] = LoadAccum(a.v, row, &is_double);
          f |= kRecValValid;
        }
        rflags[j] = f;
      }
    }
    // phase 1b: partition + rank inside the tile (LDS atomics)
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      if (rpart[j] >= 0) {
        const uint64_t x = WIDE ? ((uint64_t(rhi[WIDE ? j : 0]) << 32) | rx[j])
                                : ((rflags[j] & kRecKeyNull) ? (uint64_t(1) << a.k_bits) : uint64_t(rx[j]));
        const int p = static_cast<int>(Fmix64(x) >> (64 - kPBits));
        rpart[j] = p;
        rrank[j] = atomicAdd(&s_cnt[p], 1u);
      }
    }
    __syncthreads();
    uint32_t c = 0, incl = 0;
    if (tid < kP) {
      c = s_cnt[tid];
      incl = WaveInclusiveScan(c);
      if (lane == 63) s_scan[wave] = incl;
    }
    __syncthreads();
    if (tid < kP) {
      uint32_t woff = 0;
      for (int w = 0; w < wave; ++w) woff += s_scan[w];
      s_off[tid] = woff + incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      if (rpart[j] >= 0) {
        uint4 r;
        r.x = rx[j]; r.y = rflags[j];
        r.z = static_cast<uint32_t>(rval[j]); r.w = static_cast<uint32_t>(rval[j] >> 32);
        s_rec[s_off[rpart[j]] + rrank[j]] = r;
        if (WIDE) s_hi[WIDE ? s_off[rpart[j]] + rrank[j] : 0] = rhi[WIDE ? j : 0];
      }
    }
    __syncthreads();
    const uint32_t total = s_off[kP - 1] + s_cnt[kP - 1];
    for (uint32_t i = tid; i < total; i += kPartThreads) {
      const uint4 r = s_rec[i];
      const uint64_t x = WIDE ? ((uint64_t(s_hi[WIDE ? i : 0]) << 32) | r.x)
                              : ((r.y & kRecKeyNull) ? (uint64_t(1) << a.k_bits) : uint64_t(r.x));
      const int p = static_cast<int>(Fmix64(x) >> (64 - kPBits));
      records[s_cur[p] + (i - s_off[p])] = r;
      if (WIDE) hi_words[s_cur[p] + (i - s_off[p])] = s_hi[WIDE ? i : 0];
    }
    __syncthreads();
    if (tid < kP) { s_cur[tid] += s_cnt[tid]; s_cnt[tid] = 0; }
    __syncthreads();
  }
}

// One workgroup per partition: aggregate its records in an LDS hash table, then publish the
// table as slice p of the global GroupTable (T = kP * kSlots) for the shared numbering/emit.
template <bool WIDE>
__global__ __launch_bounds__(kPartThreads) void k_part_aggregate(const uint4* records, const uint32_t* hi_words, const uint32_t* offsets,
                                                                 int nblocks, int k_bits, int acc_double, GroupTable t,
                                                                 unsigned int* overflow) {
  __shared__ unsigned long long s_h[kSlots];
  __shared__ unsigned long long s_sum[kSlots];
  __shared__ uint32_t s_first[kSlots];
  __shared__ unsigned int s_cnt[kSlots];
  __shared__ unsigned int s_groups;
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  for (int s = tid; s < kSlots; s += kPartThreads) { s_h[s] = kEmpty; s_sum[s] = 0; s_first[s] = kNoRow; s_cnt[s] = 0; }
  if (tid == 0) s_groups = 0;
  __syncthreads();
  const int64_t lo = offsets[int64_t(p) * nblocks];
  const int64_t hi = offsets[int64_t(p + 1) * nblocks];  // hist carries one extra zero entry at the end
  bool failed = false;
  constexpr int kUnroll = 4;
  for (int64_t i0 = lo + tid; i0 < hi; i0 += int64_t(kUnroll) * kPartThreads) {
    uint4 rec[kUnroll];
    uint32_t rec_hi[WIDE ? kUnroll : 1];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = i0 + int64_t(u) * kPartThreads;
      rec[u] = make_uint4(0, 0, 0, 0);
      if (i < hi) rec[u] = records[i];
      if (WIDE) rec_hi[WIDE ? u : 0] = i < hi ? hi_words[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t i = i0 + int64_t(u) * kPartThreads;
      if (i >= hi) break;
      const uint4 r = rec[u];
      const uint64_t x = WIDE ? ((uint64_t(rec_hi[WIDE ? u : 0]) << 32) | r.x)
                              : ((r.y & kRecKeyNull) ? (uint64_t(1) << k_bits) : uint64_t(r.x));
      const uint64_t h = Fmix64(x);
      if (WIDE && h == kEmpty) { failed = true; continue; }  // the one key that hashes onto the free marker: generic path
      uint32_t slot = static_cast<uint32_t>(h) & (kSlots - 1);
      bool found = false;
      for (int probes = 0; probes < kSlots; ++probes) {
        unsigned long long cur = s_h[slot];
        if (cur == h) { found = true; break; }
        if (cur == kEmpty) {
          if (s_groups >= kSlotLimit) break;
          cur = atomicCAS(&s_h[slot], static_cast<unsigned long long>(kEmpty), static_cast<unsigned long long>(h));
          if (cur == kEmpty) { atomicAdd(&s_groups, 1u); found = true; break; }
          if (cur == h) { found = true; break; }
        }
        slot = (slot + 1) & (kSlots - 1);
      }
      if (!found) { failed = true; continue; }
      atomicMin(&s_first[slot], r.y & kRecRowMask);
      if (r.y & kRecValValid) {
        const unsigned long long bits = (static_cast<unsigned long long>(r.w) << 32) | r.z;
        if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot]), __longlong_as_double(bits));
        else atomicAdd(&s_sum[slot], bits);
        atomicAdd(&s_cnt[slot], 1u);
      }
    }
  }
  if (failed) *overflow = 1;
  __syncthreads();
  const int64_t g0 = int64_t(p) * kSlots;
  for (int s = tid; s < kSlots; s += kPartThreads) {
    t.hkeys[g0 + s] = s_h[s];
    t.first[g0 + s] = s_first[s];
    t.sums[g0 + s] = s_sum[s];
    t.counts[g0 + s] = s_cnt[s];
  }
  if (p == 0 && tid < 2) {  // the two special slots stay empty on this path
    const int64_t T = int64_t(kP) * kSlots;
    t.hkeys[T + tid] = kEmpty; t.first[T + tid] = kNoRow; t.sums[T + tid] = 0; t.counts[T + tid] = 0;
  }
}


// ==== fast path, one pass: private (row-range block, partition) regions ==========================
// Each scatter workgroup owns, for every partition, a fixed-capacity region (laid out
// [workgroup][partition][cap], so one workgroup writes within a few MB = a few TLB pages); a
// record goes to region[b][p][cursor++] with the cursor kept in LDS, so there is no histogram pass and no
// global scan.  Hash partitioning makes the per-region fill Poisson-like, so a capacity of
// mean*1.25 + 8*sqrt(mean) + 16 overflows only on heavily skewed keys -- then `overflow` is
// raised and the exact two-pass variant above (k_part_hist / k_part_scatter) is used instead.
// Partial 128-byte lines of a region are completed by later rows of the same workgroup while
// the line is still in this XCD's L2.
constexpr int kDThreads = kPartThreads;  // 1024
constexpr int kDRows = kPartRows;         // 8192 rows per tile: 8 per thread

__device__ __forceinline__ uint32_t FastPartition32(uint32_t x, bool is_null, int sub_bits) {
  return is_null ? 0u : PartOf(Fmix32(x), sub_bits);  // the null key lives in partition 0 of sub-problem 0
}

struct TileRegs {
  uint32_t x[kDRows / kDThreads], flags[kDRows / kDThreads];
  uint64_t val[kDRows / kDThreads];
  bool sel[kDRows / kDThreads];
};
// Issue every load of one tile (mask word, key, value); nothing here waits on memory.
__device__ __forceinline__ void LoadTile(const FastArgs& a, int64_t tile0, int64_t hi, int tid, int lane, TileRegs* t) {
#pragma unroll
  for (int j = 0; j < kDRows / kDThreads; ++j) {
    const int64_t row = tile0 + j * kDThreads + tid;
    const uint64_t w = FastMaskWord(a, row - lane, hi);
    t->sel[j] = (w >> lane) & 1;
    t->x[j] = 0; t->flags[j] = 0; t->val[j] = 0;
    if (t->sel[j]) {
      const bool kn = a.k_valid && !GetBit(a.k_valid, a.k_off + row);
      t->x[j] = kn ? 0u : static_cast<uint32_t>(FastKeyBits(a, row));
      uint32_t f = static_cast<uint32_t>(row) | (kn ? kRecKeyNull : 0u);
      if (a.has_values && (!a.v.valid || GetBit(a.v.valid, a.v.off + row))) {
        bool is_double;
        t->val[j] = LoadAccum(a.v, row, &is_double);
        f |= kRecValValid;
      }
      t->flags[j] = f;
    }
  }
}

// Per 8192-row tile: rank records per partition with LDS atomics, stage them in LDS grouped by
// partition, then copy every partition's run to the tail of this workgroup's private region.
// The loads of tile i+1 are issued before the copy-out of tile i, so HBM latency hides behind it.
__global__ __launch_bounds__(kDThreads) void k_part_scatter_direct(FastArgs a, int64_t cap, uint4* regions,
                                                                   uint32_t* counts, unsigned int* overflow,
                                                                   const unsigned int* run_if, const unsigned int* skip_if) {
  if (run_if && !*run_if) return;  // the local path already built the table
  if (skip_if && *skip_if) return;  // the launch on the mask without the hot rows does the work
  __shared__ unsigned int s_cnt[2][kP];
  __shared__ uint32_t s_off[kP];
  __shared__ uint32_t s_cur[kP];
  __shared__ uint32_t s_scan[8];
  __shared__ uint4 s_rec[kDRows];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kIters = kDRows / kDThreads;
  if (tid < kP) { s_cnt[0][tid] = 0; s_cnt[1][tid] = 0; s_cur[tid] = 0; }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  bool over = false;
  TileRegs t;
  LoadTile(a, lo, hi, tid, lane, &t);
  int buf = 0;
  for (int64_t tile0 = lo; tile0 < hi; tile0 += kDRows, buf ^= 1) {
    uint32_t rrank[kIters];
    uint32_t rpart[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      if (t.sel[j]) {
        const bool kn = (t.flags[j] & kRecKeyNull) != 0;
        if (kn ? a.sub_id != 0 : !InSub(Fmix32(t.x[j]), a.sub_bits, a.sub_id)) { t.sel[j] = false; continue; }  // another pass takes it
        rpart[j] = FastPartition32(t.x[j], kn, a.sub_bits);
        rrank[j] = atomicAdd(&s_cnt[buf][rpart[j]], 1u);
      }
    }
    __syncthreads();
    uint32_t c = 0, incl = 0;
    if (tid < kP) {
      c = s_cnt[buf][tid];
      incl = WaveInclusiveScan(c);
      if (lane == 63) s_scan[wave] = incl;
    }
    __syncthreads();
    if (tid < kP) {
      uint32_t woff = 0;
      for (int w = 0; w < wave; ++w) woff += s_scan[w];
      s_off[tid] = woff + incl - c;
      s_cnt[buf ^ 1][tid] = 0;  // the other counter set is idle until the next tile
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      if (t.sel[j]) {
        uint4 r;
        r.x = t.x[j]; r.y = t.flags[j];
        r.z = static_cast<uint32_t>(t.val[j]); r.w = static_cast<uint32_t>(t.val[j] >> 32);
        s_rec[s_off[rpart[j]] + rrank[j]] = r;
      }
    }
    __syncthreads();
    if (tile0 + kDRows < hi) LoadTile(a, tile0 + kDRows, hi, tid, lane, &t);  // in flight during the copy-out
    const uint32_t total = s_off[kP - 1] + s_cnt[buf][kP - 1];
    for (uint32_t i = tid; i < total; i += kDThreads) {
      const uint4 r = s_rec[i];
      const uint32_t p = FastPartition32(r.x, r.y & kRecKeyNull, a.sub_bits);
      const uint32_t idx = s_cur[p] + (i - s_off[p]);
      if (idx < cap) regions[(int64_t(blockIdx.x) * kP + p) * cap + idx] = r;
      else over = true;
    }
    __syncthreads();
    if (tid < kP) s_cur[tid] += s_cnt[buf][tid];
  }
  if (over) *overflow = 1;
  __syncthreads();
  if (tid < kP) {
    const uint32_t c = s_cur[tid];
    counts[int64_t(tid) * a.nblocks + blockIdx.x] = c < cap ? c : static_cast<uint32_t>(cap);
  }
}

// Row mapping of the fast scatter: wave w owns rows [tile0 + 512 w, +512); step j covers 64
// consecutive rows, one per lane.  The 8 mask (and value-validity) words of a wave are then 64
// consecutive bytes: lane l keeps word (l & 7) and step j fetches it with a shuffle -- 2 VGPRs
// instead of 16.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it
// would wait for the prefetched tile's global loads at every barrier and serialise HBM latency
// with the LDS phases; the staged records never leave LDS between these barriers.
__device__ __forceinline__ void LdsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Branch-free raw fetch of the (up to) two aligned words that hold 64 bits starting at bit
// `pos`; combined later by CombineBits.  No control flow and no use of the loaded data here,
// so the loads stay in flight (a data-dependent branch would force `s_waitcnt vmcnt(0)` and
// drain the prefetched tile).
struct RawBits { uint64_t lo, hi; };
__device__ __forceinline__ RawBits LoadRawBits(const uint8_t* __restrict__ bm, int64_t pos, int nbits) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(bm) + static_cast<uintptr_t>(pos >> 3);
  const uint64_t* w = reinterpret_cast<const uint64_t*>(addr & ~uintptr_t(7));
  const int shift = static_cast<int>((addr & 7) << 3) + static_cast<int>(pos & 7);
  RawBits r;
  r.lo = w[0];
  r.hi = w[(shift + nbits > 64) ? 1 : 0];  // never touches a word that holds no requested bit
  return r;
}
__device__ __forceinline__ uint64_t CombineBits(const uint8_t* bm, int64_t pos, int nbits, RawBits r) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(bm) + static_cast<uintptr_t>(pos >> 3);
  const int shift = static_cast<int>((addr & 7) << 3) + static_cast<int>(pos & 7);
  uint64_t v = r.lo >> shift;
  if (shift + nbits > 64) v |= r.hi << (64 - shift);
  if (nbits < 64) v &= (uint64_t(1) << nbits) - 1;
  return nbits > 0 ? v : 0;
}

// Tile geometry of the fast scatter: 7 steps of 64 rows per wave (7168 rows per tile), which
// leaves LDS room for the per-partition carry buffers below.
constexpr int kFSteps = 7;
// SC_THREADS = 512 runs two workgroups per CU on half-size tiles (staging in rounds of kFStage
// records).  Measured: slower (578 vs 466 us) -- the per-tile work of the 512 partition counters
// (scan, carry flush) is paid twice as often and outweighs the overlap.  -DSC_PHASES adds
// per-phase cycle counters (AMDARROW_PHASES=1 prints them per call); at 1024 threads a tile costs
// ~20 k cycles: 36 % waiting at the first barrier for the prefetched loads and the previous
// tile's stores to retire, 16 % scan + carry flush, 10 % stage, 12 % copy-out.
#ifndef SC_THREADS
#define SC_THREADS 1024
#endif
constexpr int kFThreads = SC_THREADS;
constexpr int kFRows = kFSteps * kFThreads;  // 3584 rows per tile
constexpr int kFStage = SC_THREADS == 512 ? 2304 : kFRows;  // staged records per round: one round up to 64 % selectivity
constexpr int kLRows = kFSteps * kDThreads;  // 7168: tile of the 1024-thread local-aggregate kernel (same row mapping)
constexpr int kSector = 4;                   // records per 64-byte sector

struct ScatterShared {
  unsigned int cnt[2][kP];
  uint32_t off[kP];
  uint32_t cur[2][kP];      // records already written to this workgroup's region of partition p (multiple of 4)
  uint32_t wcnt[kP];        // records of (carry + this tile) that go out this tile (multiple of 4); bit 31 = overflow
  uint32_t carry_cnt[2][kP];  // 0..3 records waiting for their sector to fill; [buf] = before this tile, [buf ^ 1] = after it
  uint32_t scan[8];
  uint4 carry[kP][kSector - 1];
  uint4 rec[kFStage];
};

template <bool MASK, bool VVALID>
__device__ __forceinline__ void FastLoadTile(const FastArgs& a, const uint32_t* __restrict__ keys,
                                             const uint64_t* __restrict__ vals, int64_t tile0, int64_t hi, int wave,
                                             int lane, uint32_t (&kx)[kFSteps], uint64_t (&kv)[kFSteps], RawBits& mraw,
                                             RawBits& vraw) {
  const int64_t wrow0 = tile0 + wave * (kFSteps * 64);
  const int64_t mrow = wrow0 + 64 * (lane & 7);
  const int64_t left = hi - mrow;
  const int nbits = left >= 64 ? 64 : (left > 0 ? static_cast<int>(left) : 0);
  const int64_t mrow_c = nbits > 0 ? mrow : hi - 1;  // keep the address inside the buffers
  if (MASK) mraw = LoadRawBits(a.m_data, a.m_off + mrow_c, nbits > 0 ? nbits : 1);
  if (VVALID) vraw = LoadRawBits(a.v.valid, a.v.off + mrow_c, nbits > 0 ? nbits : 1);
#pragma unroll
  for (int j = 0; j < kFSteps; ++j) {
    const int64_t row = wrow0 + j * 64 + lane;
    const int64_t rc = row < hi ? row : hi - 1;  // always a valid row of this block
    kx[j] = keys[rc];
    kv[j] = vals[rc];
  }
}

// Phase 1 of a tile: partition id and rank (LDS atomic) of every selected row.  This is the
// first consumer of the tile's registers, i.e. where the wave waits for its prefetched loads.
__device__ __forceinline__ void FastRankTile(ScatterShared& sh, int buf, int lane, const uint32_t (&kx)[kFSteps],
                                             uint64_t mword, uint32_t (&rpart)[kFSteps], uint32_t (&rrank)[kFSteps],
                                             bool (&sel)[kFSteps], int sub_bits, uint32_t sub_id) {
#pragma unroll
  for (int j = 0; j < kFSteps; ++j) {
    const uint32_t h = Fmix32(kx[j]);
    sel[j] = ((__shfl(mword, j, 64) >> lane) & 1) && InSub(h, sub_bits, sub_id);
    rpart[j] = PartOf(h, sub_bits);
    rrank[j] = 0;
    if (sel[j]) rrank[j] = atomicAdd(&sh.cnt[buf][rpart[j]], 1u);
  }
}
// Phases 2-4: scan the partition counts, stage the records in LDS grouped by partition, copy
// every partition's run to the tail of this workgroup's private region -- in whole 64-byte
// sectors only: up to 3 trailing records per partition wait in an LDS carry buffer for the
// next tile, so HBM never sees a partially written sector (partial sectors evicted from L2
// made the region stores cost as much as the rest of the kernel).
#ifdef SC_PHASES
__device__ unsigned long long g_phase_cycles[16];
#define PH(i) do { const unsigned long long _n = __builtin_readcyclecounter(); ph_acc[i] += _n - ph_last; ph_last = _n; } while (0)
#define PH_ARGS , unsigned long long (&ph_acc)[16], unsigned long long& ph_last
#define PH_PASS , ph_acc, ph_last
#else
#define PH(i)
#define PH_ARGS
#define PH_PASS
#endif
template <bool VVALID>
__device__ __forceinline__ void FastFinishTile(ScatterShared& sh, int buf, int64_t tile0, int64_t cap, int nblocks,
                                               uint4* __restrict__ regions, int tid, int wave, int lane,
                                               const uint32_t (&kx)[kFSteps], const uint64_t (&kv)[kFSteps],
                                               uint64_t vword, const uint32_t (&rpart)[kFSteps],
                                               const uint32_t (&rrank)[kFSteps], const bool (&sel)[kFSteps], int sub_bits PH_ARGS) {
  const int64_t wrow0 = tile0 + wave * (kFSteps * 64);
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  // this workgroup's regions as one buffer: 32-bit offsets, cache policy per store (aux)
  uint4* const block_base = regions + int64_t(blockIdx.x) * kP * cap;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(block_base, 0, static_cast<int>(kP * cap * 16), 0x00020000);
  PH(1);
  LdsBarrier();
  PH(2);
  uint32_t c = 0, incl = 0;
  if (tid < kP) {
    c = sh.cnt[buf][tid];
    incl = WaveInclusiveScan(c);
    if (lane == 63) sh.scan[wave] = incl;
  }
  LdsBarrier();
  PH(3);
  if (tid < kP) {
    // all eight wave totals in two reads (a loop over `wave` dependent reads took 7 LDS round trips)
    const uint4 s0 = *reinterpret_cast<const uint4*>(&sh.scan[0]), s1 = *reinterpret_cast<const uint4*>(&sh.scan[4]);
    const uint32_t tot[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) woff += w < wave ? tot[w] : 0u;
    sh.off[tid] = woff + incl - c;
    sh.cnt[buf ^ 1][tid] = 0;
    // sectors that become complete with this tile; the old carry leads them out
    const uint32_t cc = sh.carry_cnt[buf][tid];
    const uint32_t w = (cc + c) & ~uint32_t(kSector - 1);
    const uint32_t cur = sh.cur[buf][tid];
    const bool ovf = cur + w > cap;
    sh.wcnt[tid] = w | (ovf ? 0x80000000u : 0u);
    // the state after this tile is known now: no trailing phase (and barrier) to update it
    sh.carry_cnt[buf ^ 1][tid] = cc + c - w;
    sh.cur[buf ^ 1][tid] = cur + w;
    if (w > 0 && !ovf && cc > 0) {
      uint4* dst = regions + (int64_t(blockIdx.x) * kP + tid) * cap + cur;
      const uint4 c0 = sh.carry[tid][0], c1 = sh.carry[tid][1], c2 = sh.carry[tid][2];  // read together
      dst[0] = c0;
      if (cc > 1) dst[1] = c1;
      if (cc > 2) dst[2] = c2;
    }
  }
  LdsBarrier();
  PH(4);
  const uint32_t total = sh.off[kP - 1] + sh.cnt[buf][kP - 1];
  // The staging buffer holds kFStage records; a denser tile goes out in several rounds.
  for (uint32_t base = 0; base < total; base += kFStage) {
    if (base) LdsBarrier();  // the previous round's copy-out has read the buffer
#pragma unroll
    for (int j = 0; j < kFSteps; ++j) {
      // the shuffle must run with the whole wave active (the source lane may be unselected)
      const bool vvalid = VVALID ? ((__shfl(vword, j, 64) >> lane) & 1) : true;
      if (sel[j]) {
        const uint32_t at = sh.off[rpart[j]] + rrank[j] - base;
        if (at < static_cast<uint32_t>(kFStage)) {
          const uint64_t v = vvalid ? kv[j] : 0;
          uint4 q;
          q.x = kx[j];
          q.y = static_cast<uint32_t>(wrow0 + j * 64 + lane) | (vvalid ? kRecValValid : 0u);
          q.z = static_cast<uint32_t>(v); q.w = static_cast<uint32_t>(v >> 32);
          sh.rec[at] = q;
        }
      }
    }
    PH(5);
    LdsBarrier();
    PH(6);
    const uint32_t round = total - base < static_cast<uint32_t>(kFStage) ? total - base : static_cast<uint32_t>(kFStage);
    for (uint32_t i = tid; i < round; i += kFThreads) {
      const uint4 q = sh.rec[i];
      const uint32_t p = PartOf(Fmix32(q.x), sub_bits);
      const uint32_t k = sh.carry_cnt[buf][p] + (base + i - sh.off[p]);  // position in (old carry + this tile's run)
      const uint32_t w = sh.wcnt[p];
      if (k < (w & 0x7FFFFFFFu)) {
        if (!(w >> 31)) {
          const uint32_t voff = static_cast<uint32_t>((int64_t(p) * cap + sh.cur[buf][p] + k) * 16);
          u32x4 v = {q.x, q.y, q.z, q.w};
          // nt: the region lines are written once and read once by another CU -- streaming them past
          // the L2 replacement queue measured 10 % faster than default-policy stores (sc1 write-through: 45 % slower)
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voff, 0, 2);
        }
      } else {
        sh.carry[p][k - (w & 0x7FFFFFFFu)] = q;  // < 3 by construction
      }
    }
  }
  PH(7);
  // No barrier here: the next tile's rank phase only touches cnt[buf ^ 1] (cleared above), and three
  // barriers separate this copy-out from the next writes to rec / carry / wcnt.
}

// Specialisation of k_part_scatter_direct for the headline shape: 32-bit keys without nulls and
// 8-byte values accumulated as raw bits (int64 / uint64 / double).  Straight-line code, loads
// unconditional, and TWO register sets so that the next tile's loads are in flight while the
// current tile goes through its LDS phases.
template <bool MASK, bool VVALID>
__global__ __launch_bounds__(kFThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_part_scatter_fast(FastArgs a, int64_t cap, uint4* regions, uint32_t* counts,
                                                                 unsigned int* overflow, const unsigned int* run_if,
                                                                 const unsigned int* skip_if) {
  if (run_if && !*run_if) return;  // the local path already built the table
  if (skip_if && *skip_if) return;  // the launch on the mask without the hot rows does the work
  __shared__ ScatterShared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < kP) { sh.cnt[0][tid] = 0; sh.cnt[1][tid] = 0; sh.cur[0][tid] = 0; sh.carry_cnt[0][tid] = 0; sh.wcnt[tid] = 0; }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(a.k_data) + a.k_off;
  const uint64_t* vals = reinterpret_cast<const uint64_t*>(a.v.data) + a.v.off;
  uint32_t kxA[kFSteps], kxB[kFSteps];
  uint64_t kvA[kFSteps], kvB[kFSteps];
  RawBits mA{}, vA{}, mB{}, vB{};
  // mask / validity word of this lane's (lane & 7)-th 64-row step of the tile starting at t0
  auto words = [&](int64_t t0, const RawBits& mraw, const RawBits& vraw, uint64_t* mword, uint64_t* vword) {
    const int64_t mrow = t0 + wave * (kFSteps * 64) + 64 * (lane & 7);
    const int64_t left = hi - mrow;
    const int nbits = left >= 64 ? 64 : (left > 0 ? static_cast<int>(left) : 0);
    const int64_t mrow_c = nbits > 0 ? mrow : hi - 1;
    const uint64_t live = nbits == 0 ? 0 : (nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0));
    *mword = MASK ? (CombineBits(a.m_data, a.m_off + mrow_c, nbits > 0 ? nbits : 1, mraw) & live) : live;
    *vword = VVALID ? (CombineBits(a.v.valid, a.v.off + mrow_c, nbits > 0 ? nbits : 1, vraw) & live) : live;
  };
  // Order per tile X: rank(X) [first use of X's registers = the only vmcnt wait], then issue the
  // loads of the NEXT tile, then scan / stage / copy-out of X.  The next tile's loads are the
  // youngest outstanding memory operations when they are issued, so they fly during three LDS
  // phases and the copy-out.  (Issuing them any earlier makes the compiler wait vmcnt(0) on
  // them: vmcnt retires in order and the copy-out issues a data-dependent number of stores.)
  uint32_t rpart[kFSteps], rrank[kFSteps];
  bool sel[kFSteps];
  uint64_t mw, vw;
  int fin = 0;  // which copy of cur / carry_cnt holds the state after the last tile
#ifdef SC_PHASES
  unsigned long long ph_acc[16] = {0};
  unsigned long long ph_last = __builtin_readcyclecounter();
#endif
  if (lo < hi) FastLoadTile<MASK, VVALID>(a, keys, vals, lo, hi, wave, lane, kxA, kvA, mA, vA);
  for (int64_t tile0 = lo; tile0 < hi; tile0 += 2 * kFRows) {
    const bool has_b = tile0 + kFRows < hi;
    words(tile0, mA, vA, &mw, &vw);
    FastRankTile(sh, 0, lane, kxA, mw, rpart, rrank, sel, a.sub_bits, a.sub_id);
    PH(0);
    if (has_b) FastLoadTile<MASK, VVALID>(a, keys, vals, tile0 + kFRows, hi, wave, lane, kxB, kvB, mB, vB);
    FastFinishTile<VVALID>(sh, 0, tile0, cap, a.nblocks, regions, tid, wave, lane, kxA, kvA, vw, rpart, rrank, sel, a.sub_bits PH_PASS);
    fin = 1;
    if (!has_b) break;
    words(tile0 + kFRows, mB, vB, &mw, &vw);
    FastRankTile(sh, 1, lane, kxB, mw, rpart, rrank, sel, a.sub_bits, a.sub_id);
    PH(0);
    if (tile0 + 2 * kFRows < hi) FastLoadTile<MASK, VVALID>(a, keys, vals, tile0 + 2 * kFRows, hi, wave, lane, kxA, kvA, mA, vA);
    FastFinishTile<VVALID>(sh, 1, tile0 + kFRows, cap, a.nblocks, regions, tid, wave, lane, kxB, kvB, vw, rpart, rrank, sel, a.sub_bits PH_PASS);
    fin = 0;
  }
#ifdef SC_PHASES
  if (tid == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_phase_cycles[i], ph_acc[i]);
#endif
  __syncthreads();
  if (tid < kP) {
    // flush the last (partial) sector of every partition and publish the region fill
    const uint32_t cc = sh.carry_cnt[fin][tid], cur = sh.cur[fin][tid];
    bool over = (sh.wcnt[tid] >> 31) != 0 || cur + cc > cap;
    if (!over) {
      uint4* dst = regions + (int64_t(blockIdx.x) * kP + tid) * cap + cur;
      for (uint32_t k = 0; k < cc; ++k) dst[k] = sh.carry[tid][k];
    }
    counts[int64_t(tid) * a.nblocks + blockIdx.x] = over ? 0 : cur + cc;
    if (over) *overflow = 1;
  }
}

// ---- scatter with per-partition LDS bins --------------------------------------------------------
// Same contract as k_part_scatter_fast (headline shape, private [workgroup][partition][cap] regions,
// whole 64-byte sectors only), without the per-tile scan / compact staging / per-record partition
// look-ups: every partition owns a BIN of kBinSlots records in LDS.  A selected row's LDS atomic on
// the bin's counter IS its slot; after one barrier the bin's complete 128-byte lines go to the region tail
// (BinFlush) and the (fewer than eight) left-over records move to the bin's front.  Two LDS barriers per tile
// instead of four, no scan, and the loads are 16 bytes per lane: a thread owns R consecutive rows (one key vector,
// R / 2 value vectors, one nibble of the mask).
// A row whose bin is full (arrivals are Poisson with mean tile rows x selectivity / 512; more than
// 13 in one tile is rare by choice of R) goes to the TOP of the workgroup's region for the partition instead,
// growing downwards, its slot from an LDS counter (kCSpillUnits records per region; counts[p][b] carries their
// number in its top byte).  The aggregate kernel folds them after the region's stream.
constexpr int kBinSlots = 16;
constexpr int kBThreads = 1024;
// status[0]: why a partitioned attempt failed
constexpr unsigned int kOverflowRegion = 1;  // a (workgroup, partition) region was too small: the exact two-pass variant helps
constexpr unsigned int kOverflowTable = 2;   // a partition holds more groups than LDS slots: only the generic table helps
constexpr int kCSpillUnits = 32;  // records per (workgroup, partition) region that may find their bin full, see below
struct BinShared {
  unsigned int cnt[kP];  // records in the bin (runs past kBinSlots while a tile is ranked: the excess was spilled)
  uint32_t cur[kP];      // records of this workgroup's region already written (a multiple of 8)
  uint32_t list[kP];     // per wave (32 entries each): the partitions with a line to flush, see BinFlush
  unsigned int spill[kP];  // records of this workgroup that found their bin full (they sit at the top of the region)
  uint4 bin[kP * kBinSlots];
};
// A bin is four 64-byte sectors; sector s of bin p lives at physical sector s ^ (p & 2 ? 3 : 0).  The flush
// reads bins with eight lanes per partition, and the hardware serves a ds_read_b128 in the lane groups
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32): with this swizzle the four (partition, sector) pairs
// of every group fall into four different 16-bank groups.
__device__ __forceinline__ uint32_t BinIndex(uint32_t p, uint32_t slot) {
  return p * kBinSlots + ((((slot >> 2) ^ ((p & 2u) ? 3u : 0u)) & 3u) << 2) + (slot & 3u);
}

// One tile's loads as they arrive: nothing is computed from them before the tile is staged (a use would
// make the wave wait for the load right where it was issued).
using BinU32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using BinU32x2 = __attribute__((ext_vector_type(2))) unsigned int;
using BinU64x2 = __attribute__((ext_vector_type(2))) unsigned long long;
// Partition of a key in the bin scatter: its low 9 bits XOR nine hashed bits of the rest.  (partition, key >> 9)
// names the key, and key >> 9 spans a small range when the keys do -- small enough, for key ranges up to
// 2 M, to index the partition's LDS table directly (k_part_aggregate_piped) instead of hashing into it --
// while keys that share their low bits (multiples of 512, say) are still spread by the hashed part.
__device__ __forceinline__ uint32_t BinPartOf(uint32_t kk) { return (kk & (kP - 1)) ^ (((kk >> kPBits) * 0x9E3779B1u) >> (32 - kPBits)); }
// The keys of a tile as they were loaded: 32-bit keys, or 64-bit keys of a range-compressed column
// (x = key - bias fits 32 bits; the subtraction happens when the tile is staged).
template <int R, bool K64> struct BinKeys;
template <> struct BinKeys<4, false> {
  BinU32x4 kv;
  __device__ __forceinline__ uint32_t hi(int, uint64_t) const { return 0u; }
  __device__ __forceinline__ uint32_t k(int j, uint64_t) const { return kv[j]; }
  __device__ __forceinline__ void load(const void* base, int64_t r0) { kv = __builtin_nontemporal_load(reinterpret_cast<const BinU32x4*>(static_cast<const uint32_t*>(base) + r0)); }
  __device__ __forceinline__ void set(int j, const void* base, int64_t r) { kv[j] = static_cast<const uint32_t*>(base)[r]; }
  __device__ __forceinline__ void touch() const { asm volatile("" :: "v"(kv)); }
};
template <> struct BinKeys<2, false> {
  BinU32x2 kv;
  __device__ __forceinline__ uint32_t hi(int, uint64_t) const { return 0u; }
  __device__ __forceinline__ uint32_t k(int j, uint64_t) const { return kv[j]; }
  __device__ __forceinline__ void load(const void* base, int64_t r0) { kv = __builtin_nontemporal_load(reinterpret_cast<const BinU32x2*>(static_cast<const uint32_t*>(base) + r0)); }
  __device__ __forceinline__ void set(int j, const void* base, int64_t r) { kv[j] = static_cast<const uint32_t*>(base)[r]; }
  __device__ __forceinline__ void touch() const { asm volatile("" :: "v"(kv)); }
};
template <> struct BinKeys<4, true> {
  BinU64x2 ka, kb;
  // (the upper half of key - bias: non-zero when a hinted bias does not hold for this key)
  __device__ __forceinline__ uint32_t hi(int j, uint64_t bias) const { return static_cast<uint32_t>(((j < 2 ? ka[j] : kb[j - 2]) - bias) >> 32); }
  __device__ __forceinline__ uint32_t k(int j, uint64_t bias) const { return static_cast<uint32_t>((j < 2 ? ka[j] : kb[j - 2]) - bias); }
  __device__ __forceinline__ void load(const void* base, int64_t r0) {
    ka = __builtin_nontemporal_load(reinterpret_cast<const BinU64x2*>(static_cast<const uint64_t*>(base) + r0));
    kb = __builtin_nontemporal_load(reinterpret_cast<const BinU64x2*>(static_cast<const uint64_t*>(base) + r0 + 2));
  }
  __device__ __forceinline__ void set(int j, const void* base, int64_t r) {
    if (j < 2) ka[j] = static_cast<const uint64_t*>(base)[r]; else kb[j - 2] = static_cast<const uint64_t*>(base)[r];
  }
  __device__ __forceinline__ void touch() const { asm volatile("" :: "v"(ka), "v"(kb)); }
};
template <> struct BinKeys<2, true> {
  BinU64x2 ka;
  __device__ __forceinline__ uint32_t hi(int j, uint64_t bias) const { return static_cast<uint32_t>((ka[j] - bias) >> 32); }
  __device__ __forceinline__ uint32_t k(int j, uint64_t bias) const { return static_cast<uint32_t>(ka[j] - bias); }
  __device__ __forceinline__ void load(const void* base, int64_t r0) { ka = __builtin_nontemporal_load(reinterpret_cast<const BinU64x2*>(static_cast<const uint64_t*>(base) + r0)); }
  __device__ __forceinline__ void set(int j, const void* base, int64_t r) { ka[j] = static_cast<const uint64_t*>(base)[r]; }
  __device__ __forceinline__ void touch() const { asm volatile("" :: "v"(ka)); }
};
template <int R, bool K64> struct BinTile;
template <bool K64> struct BinTile<4, K64> {
  BinKeys<4, K64> key; BinU64x2 va, vb; uint32_t m0, m1, v0, v1;
  __device__ __forceinline__ uint64_t v(int j) const { return j < 2 ? va[j] : vb[j - 2]; }
};
template <bool K64> struct BinTile<2, K64> {
  BinKeys<2, K64> key; BinU64x2 va; uint32_t m0, m1, v0, v1;
  __device__ __forceinline__ uint64_t v(int j) const { return va[j]; }
};

// R consecutive bits of a bitmap starting at bit `pos` (R <= 4, any alignment) arrive as two bytes;
// `last_byte` is the last byte that holds a bit of the buffer.
__device__ __forceinline__ void LoadBitBytes(const uint8_t* __restrict__ bm, int64_t pos, int64_t last_byte, uint32_t* b0,
                                             uint32_t* b1) {
  const int64_t i0 = pos >> 3;
  const int64_t i1 = i0 + 1 <= last_byte ? i0 + 1 : last_byte;
  *b0 = bm[i0];
  *b1 = bm[i1];
}
template <int R>
__device__ __forceinline__ uint32_t FewBits(uint32_t b0, uint32_t b1, int64_t pos) {
  return ((b0 | (b1 << 8)) >> (pos & 7)) & ((1u << R) - 1u);
}

// A whole tile (every row below hi): 16-byte loads, unconditional.
template <bool MASK, bool VVALID, int R, bool K64>
__device__ __forceinline__ void BinLoadTile(const FastArgs& a, const void* __restrict__ keys,
                                            const uint64_t* __restrict__ vals, int64_t tile0, int tid, BinTile<R, K64>& t) {
  const int64_t r0 = tile0 + int64_t(tid) * R;
  t.key.load(keys, r0);
  // (a call without a value column: `vals` is any readable buffer and every lane loads its first bytes -- the
  // load stays unconditional, which keeps the count of loads in flight a constant; see BinRun)
  const int64_t rv = r0 * a.has_values;
  t.va = __builtin_nontemporal_load(reinterpret_cast<const BinU64x2*>(vals + rv));
  if constexpr (R == 4) t.vb = __builtin_nontemporal_load(reinterpret_cast<const BinU64x2*>(vals + rv + 2));
  if (MASK) LoadBitBytes(a.m_data, a.m_off + r0, (a.m_off + a.n - 1) >> 3, &t.m0, &t.m1);
  if (VVALID) LoadBitBytes(a.v.valid, a.v.off + r0, (a.v.off + a.n - 1) >> 3, &t.v0, &t.v1);
}
// The last, partial tile of the last block: one row at a time, addresses kept inside the buffers.
template <bool MASK, bool VVALID, int R, bool K64>
__device__ __forceinline__ void BinLoadTail(const FastArgs& a, const void* __restrict__ keys,
                                            const uint64_t* __restrict__ vals, int64_t tile0, int64_t hi, int tid,
                                            BinTile<R, K64>& t) {
  const int64_t r0 = tile0 + int64_t(tid) * R;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int64_t rc = r0 + j < hi ? r0 + j : hi - 1;
    t.key.set(j, keys, rc);
    if (j < 2) t.va[j] = vals[rc * a.has_values];
    if constexpr (R == 4) { if (j >= 2) t.vb[j - 2] = vals[rc * a.has_values]; }
  }
  const int64_t rb = r0 < hi ? r0 : hi - 1;
  if (MASK) LoadBitBytes(a.m_data, a.m_off + rb, (a.m_off + a.n - 1) >> 3, &t.m0, &t.m1);
  if (VVALID) LoadBitBytes(a.v.valid, a.v.off + rb, (a.v.off + a.n - 1) >> 3, &t.v0, &t.v1);
}

// Rank + stage: the slot of a selected row is what the LDS atomic on its bin's counter returns.
template <bool MASK, bool VVALID, int R, bool K64>
__device__ __forceinline__ void BinStageTile(BinShared& sh, const FastArgs& a, const BinTile<R, K64>& t, int64_t tile0, int64_t hi,
                                             int tid, int64_t cap, uint4* __restrict__ regions,
                                             uint32_t* __restrict__ counts, unsigned int* overflow, uint32_t& kmin,
                                             uint32_t& kmax, uint32_t& vwide) {
  const int64_t r0 = tile0 + int64_t(tid) * R;
  const int64_t left = hi - r0;
  const uint32_t live = left >= R ? (1u << R) - 1u : (left > 0 ? (1u << left) - 1u : 0u);
  const int64_t rb = left > 0 ? r0 : hi - 1;  // where the bitmap bytes were fetched (bits beyond hi are dropped by `live`)
  const uint32_t mbits = MASK ? (FewBits<R>(t.m0, t.m1, a.m_off + rb) & live) : live;
  const uint32_t vbits = VVALID ? FewBits<R>(t.v0, t.v1, a.v.off + rb) : live;
  uint32_t part[R], slot[R];
  bool sel[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint32_t x = t.key.k(j, a.k_bias);
    const uint32_t kk = x ^ a.k_flip;
    sel[j] = (mbits >> j) & 1u;
    if (a.sub_bits) sel[j] = sel[j] && InSub(Fmix32(x), a.sub_bits, a.sub_id);  // (uniform branch: sub-problems only)
    part[j] = BinPartOf(kk);
    slot[j] = 0;
    if (sel[j]) {
      slot[j] = atomicAdd(&sh.cnt[part[j]], 1u);
      kmin = kk < kmin ? kk : kmin;
      kmax = kk > kmax ? kk : kmax;
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (!sel[j]) continue;
    const bool vvalid = a.has_values && ((vbits >> j) & 1u);
    const uint64_t v = vvalid ? t.v(j) : 0;
    if (a.val_wide) vwide |= static_cast<uint32_t>((v + 0x80000000ull) >> 32);  // (uniform branch; non-zero: not an int32)
    uint4 q;
    q.x = t.key.k(j, a.k_bias);
    q.y = static_cast<uint32_t>(r0 + j) | (vvalid ? kRecValValid : 0u);
    q.z = static_cast<uint32_t>(v); q.w = static_cast<uint32_t>(v >> 32);
    if (slot[j] < static_cast<uint32_t>(kBinSlots)) {
      sh.bin[BinIndex(part[j], slot[j])] = q;
    } else {  // the bin is full until the next flush: the record goes to the top of the region (LDS counter, no device atomic)
      const uint32_t at = atomicAdd(&sh.spill[part[j]], 1u);
      if (at < static_cast<uint32_t>(kCSpillUnits)) regions[(int64_t(blockIdx.x) * kP + part[j]) * cap + (cap - 1 - at)] = q;
      else __hip_atomic_store(overflow, kOverflowRegion, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Flush: whole 128-byte lines (kBinUnit records) leave, the rest moves to the front of the bin.  Measured with the
// kernel's access pattern and no LDS work at all (exp/membench.hip): the input stream next to 64-byte region writes
// runs at 4.7 TB/s, next to 128-byte writes at 6.0 TB/s.  A wave owns 32 partitions: one lane per partition decides,
// the partitions with a line to go are compacted into a wave-private list, eight lanes per listed partition copy
// (see CBinFlush, where this scheme was measured against eight lanes per partition deciding in four rounds).
constexpr int kBinUnit = 8;
__device__ __forceinline__ void BinFlush(BinShared& sh, int tid, int64_t cap, uint4* __restrict__ my_regions,
                                         unsigned int* overflow) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  constexpr int kPerWave = kP / (kBThreads / 64);
  static_assert(kPerWave == 32 && kBinSlots == 2 * kBinUnit, "one lane of the low half per partition; two lines per bin");
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t nl = 0, entry = 0;
  if (lane < kPerWave) {
    const uint32_t p = wave * kPerWave + lane;
    const uint32_t raw = sh.cnt[p], cur = sh.cur[p];
    const uint32_t c = raw < static_cast<uint32_t>(kBinSlots) ? raw : static_cast<uint32_t>(kBinSlots);
    nl = c / kBinUnit;
    const uint32_t left = c & uint32_t(kBinUnit - 1);
    if (nl > 0) {
      sh.cnt[p] = left;
      if (cur + kBinUnit * nl <= cap - kCSpillUnits) {
        sh.cur[p] = cur + kBinUnit * nl;
        entry = p | nl << 9 | left << 11 | (cur >> 3) << 16;
      } else {
        nl = 0;
        __hip_atomic_store(overflow, kOverflowRegion, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  const uint64_t going = __ballot(nl > 0);
  const int count = __popcll(going);
  if (nl > 0) sh.list[wave * kPerWave + __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(going), 0u)] = entry;
  const uint32_t ll = lane & 7;
  const int trips = (count + 7) >> 3;  // (wave-uniform)
  constexpr int kTrips = kPerWave / 8;
  uint32_t e[kTrips];
#pragma unroll
  for (int t = 0; t < kTrips; ++t) {
    e[t] = 0;
    if (t < trips) {
      const int k = 8 * t + (lane >> 3);
      if (k < count) e[t] = sh.list[wave * kPerWave + k];
    }
  }
  uint4 lo[kTrips], hi[kTrips];
#pragma unroll
  for (int t = 0; t < kTrips; ++t) {
    if (t < trips && ((e[t] >> 9) & 3u) != 0) {
      const uint32_t p = e[t] & (kP - 1);
      lo[t] = sh.bin[BinIndex(p, ll)];
      hi[t] = sh.bin[BinIndex(p, kBinUnit + ll)];  // the second line, or the records that move to the front
    }
  }
#pragma unroll
  for (int t = 0; t < kTrips; ++t) {
    const uint32_t lines = (e[t] >> 9) & 3u;
    if (t < trips && lines != 0) {
      const uint32_t p = e[t] & (kP - 1), left = (e[t] >> 11) & 31u, cur = (e[t] >> 16) << 3;
      uint4* dst = my_regions + int64_t(p) * cap + cur + ll;
      const u32x4 v = {lo[t].x, lo[t].y, lo[t].z, lo[t].w};
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
      if (lines > 1) {
        const u32x4 w = {hi[t].x, hi[t].y, hi[t].z, hi[t].w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(dst + kBinUnit));
      }
      if (lines == 1 && ll < left) sh.bin[BinIndex(p, ll)] = hi[t];
    }
  }
}

// ---- compact records ------------------------------------------------------------------------------------
// The 16-byte record {key, row, value} is what bounds the partitioned group-by: every selected row's record is
// written once and read once (0.8 GB each way next to the 1.2 GB input at 50 % selectivity).  When the selected
// keys span a small range -- the case in which the aggregate indexes its LDS table directly -- a record needs
// only 12 key bits (its slot offset (key >> 9) - c_base; the partition supplies the low 9 bits), and with every
// workgroup writing its own regions only the row's offset inside the workgroup's row range (19 bits):
//     word 0 = row offset << 13 | value valid << 12 | slot offset,  then the value: 64 bits (RW = 3 words per
//     record) or, when every selected value of an integer column is an int32, 32 bits (RW = 2).
// Whether a call qualifies is known only after a pass over its keys and values, so the context's hint (the
// last call of this shape went the partition way without heavy hitters) also carries that call's key range
// and whether its values were narrow; a hinted call scatters compact records and CHECKS every selected row
// against both assumptions -- a row that breaks one raises kOverflowCompact, and the call starts over on the
// 16-byte path like any hinted call whose assumption failed.
// A partition's records form a stream of 32-bit words: the bin holds up to 64 of them, whole 128-byte lines
// (32 words, records may straddle them) leave at a flush and the words behind them move to the bin's front.
// A record that does not fit the bin goes to the top of its region as a 16-byte record, as in the 16-byte scatter.
constexpr unsigned int kOverflowCompact = 4;  // a selected row did not fit the compact record of a hinted call
// Two geometries, PB = log2(partitions): 512 partitions with bins of 64 words (12 slot bits, 19 row bits; the
// 8-byte records) or 256 partitions with bins of 128 words (13 slot bits, 18 row bits -- twice the workgroups --;
// the 12-byte records: a bin of 64 words takes one tile's twelve words on average behind what a flush leaves, a
// bin of 128 takes two tiles' 48, so the flush and its two barriers run every other tile and decide half as many
// bins).  Both keep 2 M table slots (partitions x slots) and the same number and size of regions.
template <int PB> struct CGeo {
  static constexpr int kParts = 1 << PB;
  static constexpr int kBinWords = PB == 9 ? 64 : 128;
  static constexpr int kStride = kBinWords + 4;     // consecutive bins start four banks apart
  static constexpr int kSlotBits = 21 - PB;         // 12 / 13: the partition's directly indexed table
  static constexpr int kSlots = 1 << kSlotBits;
  static constexpr int kRowBits = 31 - kSlotBits;   // 19 / 18: a workgroup's row range
  static constexpr uint32_t kValid = 1u << kSlotBits;
  static constexpr int kPerWave = kParts / (kBThreads / 64);  // 32 / 16 partitions per wave of the flush
  static constexpr int kMaxLines = kBinWords / 32;  // 2 / 4 lines of 128 bytes per bin
  static constexpr int kListPerWave = kPerWave * kMaxLines;  // 64
  __device__ static __forceinline__ uint32_t PartOf(uint32_t kk) {
    return (kk & (kParts - 1)) ^ (((kk >> PB) * 0x9E3779B1u) >> (32 - PB));
  }
  // offset o lives in slot bitreverse(o): a slice filled to a fraction holds its groups spread over all its slots
  __device__ static __forceinline__ uint32_t Slot(uint32_t o) { return __brev(o) >> (32 - kSlotBits); }
};
static_assert(CGeo<9>::kSlots == kSlots && CGeo<9>::kParts == kP, "the 512-partition geometry is the 16-byte path's");
template <int PB> struct CBinShared {
  using G = CGeo<PB>;
  unsigned int cnt[G::kParts];  // low half: the stream's position in words (runs past what the bin holds while a tile is ranked: the excess was spilled); high half: the position at the last flush
  uint32_t cur[G::kParts];      // 16-byte units of this workgroup's region already written (a multiple of 8)
  unsigned int spill[G::kParts];  // records of this workgroup that found their bin full (they sit at the top of the region)
  uint32_t list[(kBThreads / 64) * G::kListPerWave];  // per wave: the lines to flush, see CBinFlush
  __attribute__((aligned(16))) uint32_t bin[G::kParts * G::kStride];
};
// A record that finds its bin full goes straight to the TOP of the workgroup's region for its partition, as a
// 16-byte record {key word, row | flags, value}, growing downwards: its place comes from an LDS counter.  (Such
// records used to go to a per-partition list filled through a device atomic: the returning atomic makes the wave
// wait for every load it has in flight -- two tiles -- and the workgroup waits for that wave at the next barrier,
// ~1-2 us each time.)  kCSpillUnits records per region; more raise kOverflowRegion.  Both bin scatters do this.
template <bool MASK, bool VVALID, int R, bool K64, int RW, int PB>
__device__ __forceinline__ void CBinStageTile(CBinShared<PB>& sh, const FastArgs& a, const BinTile<R, K64>& t, int64_t lo, int64_t tile0,
                                              int64_t hi, int tid, int64_t cap, uint4* __restrict__ my_regions,
                                              unsigned int* overflow, uint32_t& kmin,
                                              uint32_t& kmax, uint32_t& vwide, uint32_t& bad) {
  using G = CGeo<PB>;
  const int64_t r0 = tile0 + int64_t(tid) * R;
  const int64_t left = hi - r0;
  const uint32_t live = left >= R ? (1u << R) - 1u : (left > 0 ? (1u << left) - 1u : 0u);
  const int64_t rb = left > 0 ? r0 : hi - 1;
  const uint32_t mbits = MASK ? (FewBits<R>(t.m0, t.m1, a.m_off + rb) & live) : live;
  const uint32_t vbits = VVALID ? FewBits<R>(t.v0, t.v1, a.v.off + rb) : live;
  uint32_t part[R], at[R], off[R];
  bool sel[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint32_t kk = t.key.k(j, a.k_bias) ^ a.k_flip;
    sel[j] = (mbits >> j) & 1u;
    part[j] = G::PartOf(kk);
    off[j] = (kk >> PB) - a.c_base;
    at[j] = 0;
    if (sel[j]) {
      at[j] = atomicAdd(&sh.cnt[part[j]], static_cast<unsigned int>(RW));  // (stream position | position at the last flush << 16)
      kmin = kk < kmin ? kk : kmin;
      kmax = kk > kmax ? kk : kmax;
      bad |= off[j] >> G::kSlotBits;  // (a key below c_base wraps around to a large offset)
      if (K64) bad |= t.key.hi(j, a.k_bias);  // (a 64-bit key outside the 2^32 above the -- possibly hinted -- bias)
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (!sel[j]) continue;
    const bool vvalid = a.has_values && ((vbits >> j) & 1u);
    const uint64_t v = vvalid ? t.v(j) : 0;
    if (RW == 2) bad |= static_cast<uint32_t>((v + 0x80000000ull) >> 32);
    else if (a.val_wide) vwide |= static_cast<uint32_t>((v + 0x80000000ull) >> 32);  // (could the next call carry 32-bit values?)
    // the bin is a ring of kBinWords words: the stream's words [flushed, flushed + kBinWords) live at their position
    // mod kBinWords, `flushed` being the last whole line before the position at the last flush (a flush leaves
    // less than a line behind)
    const uint32_t pos = at[j] & 0xFFFFu, flushed = (at[j] >> 16) & ~31u;
    if (pos - flushed + RW <= static_cast<uint32_t>(G::kBinWords)) {
      uint32_t* bin = &sh.bin[part[j] * G::kStride];
      bin[pos & (G::kBinWords - 1)] = static_cast<uint32_t>(r0 + j - lo) << (G::kSlotBits + 1) | (vvalid ? G::kValid : 0u) | (off[j] & (G::kValid - 1u));
      bin[(pos + 1) & (G::kBinWords - 1)] = static_cast<uint32_t>(v);
      if (RW == 3) bin[(pos + 2) & (G::kBinWords - 1)] = static_cast<uint32_t>(v >> 32);
    } else {  // the bin is full until the next flush: the row goes to the top of the region as a 16-byte record
      uint4 q;
      q.x = t.key.k(j, a.k_bias);
      q.y = static_cast<uint32_t>(r0 + j) | (vvalid ? kRecValValid : 0u);
      q.z = static_cast<uint32_t>(v); q.w = static_cast<uint32_t>(v >> 32);
      const uint32_t s = atomicAdd(&sh.spill[part[j]], 1u);
      if (s < static_cast<uint32_t>(kCSpillUnits)) my_regions[int64_t(part[j]) * cap + (cap - 1 - s)] = q;
      else __hip_atomic_store(overflow, kOverflowRegion, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// Flush: the stream's whole 128-byte lines leave; what is behind them stays where it is (the bin is a ring).  A wave
// owns kPerWave partitions.  ONE lane per partition decides (counter read, clamp, lines to go, new counter) -- with
// eight lanes per partition doing that in four rounds the flush logic was a third of the kernel (116 of 372 us; the
// global stores it feeds: 10 us) -- and the lines to go are compacted into a wave-private list (a wave scan of the
// line counts); eight lanes per listed line then copy it (16 bytes each), eight lines per trip, the LDS reads of
// four trips issued before the first store.  Everything stays inside the wave: LDS operations of one wave execute
// in order, no barrier between the steps.
template <int RW, int PB>
__device__ __forceinline__ void CBinFlush(CBinShared<PB>& sh, int tid, int64_t cap, uint4* __restrict__ my_regions,
                                          unsigned int* overflow) {
  using G = CGeo<PB>;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t nl = 0, first_line = 0, cur = 0, p = 0;
  if (lane < G::kPerWave) {
    p = wave * G::kPerWave + lane;
    const uint32_t raw = sh.cnt[p];
    cur = sh.cur[p];
    const uint32_t start = raw >> 16, flushed = start & ~31u, had = start - flushed;  // (had < 32)
    const uint32_t came = (raw - start) & 0xFFFFu;  // words of the arrivals since the last flush, accepted or not
    // (arrivals were accepted while their last word still fitted)
    const uint32_t took = had + came <= static_cast<uint32_t>(G::kBinWords) ? came : ((G::kBinWords - had) / RW) * RW;
    const uint32_t now = (start + took) & (G::kBinWords - 1);  // (positions restart inside the ring: the low half never carries into the high one)
    nl = (had + took) >> 5;
    first_line = (flushed & (G::kBinWords - 1)) >> 5;
    sh.cnt[p] = now << 16 | now;
    if (nl > 0) {
      if (cur + 8 * nl <= cap - kCSpillUnits) {
        sh.cur[p] = cur + 8 * nl;
      } else {
        nl = 0;
        __hip_atomic_store(overflow, kOverflowRegion, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // the lines' places in the wave's list: an inclusive scan of nl over the deciding lanes
  uint32_t incl = nl;
#pragma unroll
  for (int d = 1; d < G::kPerWave; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  const int count = __builtin_amdgcn_readlane(incl, G::kPerWave - 1);  // lines of this wave (wave-uniform)
  uint32_t* list = &sh.list[wave * G::kListPerWave];
  for (uint32_t l = 0; l < nl; ++l)  // entry: partition | the line's quarter of the ring << 9 | its 128-byte unit of the region << 12
    list[incl - nl + l] = p | ((first_line + l) & (G::kMaxLines - 1)) << 9 | ((cur >> 3) + l) << 12;
  const uint32_t ll = lane & 7;
  constexpr int kTrips = 4;  // lines of up to four trips in flight
  for (int base = 0; base < count; base += 8 * kTrips) {  // (wave-uniform; one pass unless more than 32 lines go)
    uint32_t e[kTrips];
    bool on[kTrips];
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
      const int k = base + 8 * t + (lane >> 3);
      on[t] = k < count;
      e[t] = 0;
      if (base + 8 * t < count && on[t]) e[t] = list[k];
    }
    uint4 q[kTrips];
#pragma unroll
    for (int t = 0; t < kTrips; ++t)
      if (base + 8 * t < count && on[t])
        q[t] = *reinterpret_cast<const uint4*>(&sh.bin[(e[t] & (G::kParts - 1)) * G::kStride + ((e[t] >> 9) & 7u) * 32 + 4 * ll]);
#pragma unroll
    for (int t = 0; t < kTrips; ++t) {
      if (base + 8 * t < count && on[t]) {
        uint4* dst = my_regions + int64_t(e[t] & (G::kParts - 1)) * cap + ((e[t] >> 12) << 3) + ll;
        const u32x4 v = {q[t].x, q[t].y, q[t].z, q[t].w};
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst));
      }
    }
  }
}

// The tile loop of one workgroup over its row range, R rows per thread and tile.  RW = 0: 16-byte records
// (BinShared); RW = 2 / 3: compact records of RW words (CBinShared).
// FE: the bins are flushed after every FE-th tile (2 for the 8-byte records: a tile brings a bin eight words on
// average, two tiles fit behind what a flush leaves, and the flush with its two barriers is paid half as often).
template <bool MASK, bool VVALID, int R, bool K64, int RW = 0, class SH = BinShared, int FE = 1>
__device__ __forceinline__ void BinRun(SH& sh, const FastArgs& a, int64_t lo, int64_t hi, int64_t cap,
                                       uint4* __restrict__ regions, uint4* __restrict__ my_regions,
                                       uint32_t* __restrict__ counts, unsigned int* overflow, uint32_t& kmin, uint32_t& kmax,
                                       uint32_t& vwide, uint32_t& bad) {
  constexpr int kRows = kBThreads * R;
  const int tid = threadIdx.x;
  const void* keys = static_cast<const char*>(a.k_data) + a.k_off * (K64 ? 8 : 4);
  const uint64_t* vals = reinterpret_cast<const uint64_t*>(a.v.data) + a.v.off;
  // Three register sets: tile X is staged while the loads of X+1 and X+2 are in flight.  Per tile:
  //   stage(X) | barrier | wait for X+1 | flush(X) | issue X+3 | barrier
  // The wait sits BEFORE the flush stores, where the only younger memory operations are the loads of
  // X+2 (an exact vmcnt): behind the flush it would also wait for the flush's own stores.
  BinTile<R, K64> t0{}, t1{}, t2{};
#define AA_TOUCH(T) do { T.key.touch(); asm volatile("" :: "v"(T.va)); if constexpr (R == 4) asm volatile("" :: "v"(T.vb)); \
                         if (MASK) asm volatile("" :: "v"(T.m0), "v"(T.m1)); if (VVALID) asm volatile("" :: "v"(T.v0), "v"(T.v1)); } while (0)
  // Every prefetch is issued unconditionally -- past the last whole tile it fetches that tile again
  // and the result is dropped -- so that the number of loads in flight at each wait is a constant
  // the compiler can count (a conditional load makes it wait for everything).
#define AA_BIN_STEP(CUR, NEXT, TILE0)                                                                  \
  do {                                                                                                   \
    if constexpr (RW == 0) BinStageTile<MASK, VVALID, R, K64>(sh, a, CUR, (TILE0), hi, tid, cap, regions, counts, overflow, kmin, kmax, vwide); \
    else CBinStageTile<MASK, VVALID, R, K64, RW>(sh, a, CUR, lo, (TILE0), hi, tid, cap, my_regions, overflow, kmin, kmax, vwide, bad); \
    const bool flush_now = FE == 1 || (++step_no % FE) == 0;  /* (uniform) */                            \
    if (flush_now) LdsBarrier();                                                                         \
    AA_TOUCH(NEXT);                                                                                      \
    if (flush_now) {                                                                                     \
      if constexpr (RW == 0) BinFlush(sh, tid, cap, my_regions, overflow);                               \
      else CBinFlush<RW>(sh, tid, cap, my_regions, overflow);                                            \
    }                                                                                                    \
    BinLoadTile<MASK, VVALID, R, K64>(a, keys, vals, (TILE0) + 3 * kRows <= last_whole ? (TILE0) + 3 * kRows : last_whole, tid, CUR); \
    if (flush_now) LdsBarrier();                                                                         \
  } while (0)
  uint32_t step_no = 0;
  int64_t tile0 = lo;
  if (lo + kRows <= hi) {
    const int64_t last_whole = lo + ((hi - lo) / kRows - 1) * kRows;
    BinLoadTile<MASK, VVALID, R, K64>(a, keys, vals, lo, tid, t0);
    BinLoadTile<MASK, VVALID, R, K64>(a, keys, vals, lo + kRows <= last_whole ? lo + kRows : last_whole, tid, t1);
    BinLoadTile<MASK, VVALID, R, K64>(a, keys, vals, lo + 2 * kRows <= last_whole ? lo + 2 * kRows : last_whole, tid, t2);
    AA_TOUCH(t0);  // (the loop body starts with its first tile complete on every path into it)
    // whole tiles, three per trip (a trip ends with the register sets back in phase); no exit from
    // inside a trip: with one the compiler sees paths into the loop head on which t0 was just reloaded
    for (; tile0 + 2 * kRows <= last_whole; tile0 += 3 * kRows) {
      AA_BIN_STEP(t0, t1, tile0);
      AA_BIN_STEP(t1, t2, tile0 + kRows);
      AA_BIN_STEP(t2, t0, tile0 + 2 * kRows);
    }
    if (tile0 <= last_whole) {
      AA_BIN_STEP(t0, t1, tile0);
      tile0 += kRows;
      if (tile0 <= last_whole) {
        AA_BIN_STEP(t1, t2, tile0);
        tile0 += kRows;
      }
    }
  }
  if (tile0 < hi) {  // the partial tile at the end of the input
    BinLoadTail<MASK, VVALID, R, K64>(a, keys, vals, tile0, hi, tid, t0);
    if constexpr (RW == 0) BinStageTile<MASK, VVALID, R, K64>(sh, a, t0, tile0, hi, tid, cap, regions, counts, overflow, kmin, kmax, vwide);
    else CBinStageTile<MASK, VVALID, R, K64, RW>(sh, a, t0, lo, tile0, hi, tid, cap, my_regions, overflow, kmin, kmax, vwide, bad);
    LdsBarrier();
    if constexpr (RW == 0) BinFlush(sh, tid, cap, my_regions, overflow);
    else CBinFlush<RW>(sh, tid, cap, my_regions, overflow);
  } else if (FE > 1) {  // (the last whole tile may have gone without a flush: the write-out behind expects less than a line per bin)
    LdsBarrier();
    if constexpr (RW == 0) BinFlush(sh, tid, cap, my_regions, overflow);
    else CBinFlush<RW>(sh, tid, cap, my_regions, overflow);
  }
#undef AA_BIN_STEP
#undef AA_TOUCH
}

// A bin takes what is left of the last line (< 8 records) plus a tile's arrivals, Poisson with mean
// tile rows x selectivity / 512.  4096-row tiles spill 0.06 % of the records at 50 % selectivity and
// 0.6 % at 75 % (mean 6); denser input runs on 2048-row tiles (mean <= 4).  Every workgroup decides for
// itself from the mask bits of its first rows.
constexpr int kBinSampleWords = 256;  // 64-bit words of the workgroup's mask range that are counted
template <bool MASK, bool VVALID, bool K64>
__global__ __launch_bounds__(kBThreads) void k_part_scatter_bins(FastArgs a, int64_t cap, uint4* regions, uint32_t* counts,
                                                                 unsigned int* overflow, const unsigned int* run_if,
                                                                 const unsigned int* skip_if) {
  if (run_if && !*run_if) return;  // the local path already built the table
  if (skip_if && *skip_if) return;  // the launch on the mask without the hot rows does the work
  __shared__ BinShared sh;
  __shared__ unsigned int s_sample, s_kmin, s_kmax;
  const int tid = threadIdx.x;
  if (tid < kP) { sh.cnt[tid] = 0; sh.cur[tid] = 0; sh.spill[tid] = 0; }
  if (tid == 0) { s_sample = 0; s_kmin = 0xFFFFFFFFu; s_kmax = 0; }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  uint4* const my_regions = regions + int64_t(blockIdx.x) * kP * cap;
  bool dense = true;  // no mask: every row is selected
  if (MASK) {
    const int64_t sample_rows = hi - lo < 64 * kBinSampleWords ? hi - lo : 64 * kBinSampleWords;
    if (tid < kBinSampleWords) {
      const int64_t r = lo + 64 * int64_t(tid);
      const int64_t left = lo + sample_rows - r;
      if (left > 0) {
        const uint64_t w = LoadBits64(a.m_data, a.m_off + r, left >= 64 ? 64 : static_cast<int>(left));
        atomicAdd(&s_sample, static_cast<unsigned int>(__popcll(w)));
      }
    }
    __syncthreads();
    dense = 3 * int64_t(s_sample) > 2 * sample_rows;  // more than two thirds of the rows selected
  }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0, vwide = 0, bad = 0;
  if (dense) BinRun<MASK, VVALID, 2, K64>(sh, a, lo, hi, cap, regions, my_regions, counts, overflow, kmin, kmax, vwide, bad);
  else BinRun<MASK, VVALID, 4, K64>(sh, a, lo, hi, cap, regions, my_regions, counts, overflow, kmin, kmax, vwide, bad);
  if (a.val_wide && __syncthreads_or(vwide != 0) && tid == 0) __hip_atomic_store(a.val_wide, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {  // the range of the selected keys: wave, workgroup (LDS), device -- two device atomics per workgroup
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t omin = __shfl_xor(kmin, d, 64), omax = __shfl_xor(kmax, d, 64);
      kmin = omin < kmin ? omin : kmin;
      kmax = omax > kmax ? omax : kmax;
    }
    if ((tid & 63) == 0 && kmin <= kmax) { atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); }
  }
  __syncthreads();
  if (tid == 0 && s_kmin <= s_kmax) { atomicMin(&a.key_minmax[0], s_kmin); atomicMax(&a.key_minmax[1], s_kmax); }
  if (tid < kP) {
    // what is left of every partition's last line, and the region fills
    const uint32_t left = sh.cnt[tid], cur = sh.cur[tid], spilled = sh.spill[tid];  // left < kBinUnit after a flush
    const bool ok = cur + left <= cap - kCSpillUnits && spilled <= static_cast<uint32_t>(kCSpillUnits);
    if (ok) {
      uint4* dst = my_regions + int64_t(tid) * cap + cur;
      for (uint32_t k = 0; k < left; ++k) dst[k] = sh.bin[BinIndex(tid, k)];
    } else {
      __hip_atomic_store(overflow, kOverflowRegion, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    counts[int64_t(tid) * (a.nblocks + 1) + blockIdx.x] = ok ? (cur + left) | spilled << 24 : 0;  // (records | records at the region's top << 24)
  }
}

// The bin scatter with compact records of RW words (see above).  counts[p][b] = RECORDS in the region of
// workgroup b for partition p (RW words each, packed from the region's start) | 16-byte records at its top << 24.
// FE: tiles per flush.
template <bool MASK, bool VVALID, bool K64, int RW, int PB>
__global__ __launch_bounds__(kBThreads) void k_part_scatter_compact(FastArgs a, int64_t cap, uint4* regions, uint32_t* counts,
                                                                    unsigned int* overflow) {
  using G = CGeo<PB>;
  __shared__ CBinShared<PB> sh;
  __shared__ unsigned int s_sample, s_kmin, s_kmax;
  const int tid = threadIdx.x;
  if (tid < G::kParts) { sh.cnt[tid] = 0; sh.cur[tid] = 0; sh.spill[tid] = 0; }
  if (tid == 0) { s_sample = 0; s_kmin = 0xFFFFFFFFu; s_kmax = 0; }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  uint4* const my_regions = regions + int64_t(blockIdx.x) * G::kParts * cap;
  bool dense = true;  // no mask: every row is selected
  if (MASK) {
    const int64_t sample_rows = hi - lo < 64 * kBinSampleWords ? hi - lo : 64 * kBinSampleWords;
    if (tid < kBinSampleWords) {
      const int64_t r = lo + 64 * int64_t(tid);
      const int64_t left = lo + sample_rows - r;
      if (left > 0) {
        const uint64_t w = LoadBits64(a.m_data, a.m_off + r, left >= 64 ? 64 : static_cast<int>(left));
        atomicAdd(&s_sample, static_cast<unsigned int>(__popcll(w)));
      }
    }
    __syncthreads();
    dense = RW == 2 && PB == 9 ? 20 * int64_t(s_sample) > 11 * sample_rows  // (two tiles per flush: the half-size tiles from 55 % on)
                               : 3 * int64_t(s_sample) > 2 * sample_rows;   // more than two thirds of the rows selected
  }
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0, vwide = 0, bad = 0;
  // tiles per flush: what two (three) tiles bring on average fits behind what a flush leaves (< 32 words)
  constexpr int FE = PB == 9 ? (RW == 2 ? 2 : 1) : (RW == 2 ? 3 : 2);
  if (dense) BinRun<MASK, VVALID, 2, K64, RW, CBinShared<PB>, FE>(sh, a, lo, hi, cap, regions, my_regions, counts, overflow, kmin, kmax, vwide, bad);
  else BinRun<MASK, VVALID, 4, K64, RW, CBinShared<PB>, FE>(sh, a, lo, hi, cap, regions, my_regions, counts, overflow, kmin, kmax, vwide, bad);
  if (RW == 3 && a.val_wide && __syncthreads_or(vwide != 0) && tid == 0) __hip_atomic_store(a.val_wide, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // a selected row outside the hinted key range, or (RW == 2) a value that is not an int32: the call starts over
  if (__syncthreads_or(bad != 0) && tid == 0) __hip_atomic_store(overflow, kOverflowCompact, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t omin = __shfl_xor(kmin, d, 64), omax = __shfl_xor(kmax, d, 64);
      kmin = omin < kmin ? omin : kmin;
      kmax = omax > kmax ? omax : kmax;
    }
    if ((tid & 63) == 0 && kmin <= kmax) { atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); }
  }
  __syncthreads();
  if (tid == 0 && s_kmin <= s_kmax) { atomicMin(&a.key_minmax[0], s_kmin); atomicMax(&a.key_minmax[1], s_kmax); }
  if (tid < G::kParts) {
    // what is left of every partition's stream (< 32 words after a flush, behind the last whole line), and the region fills
    const uint32_t now = sh.cnt[tid] & 0xFFFFu, left = now & 31u, from = (now & ~31u) & (G::kBinWords - 1);
    const uint32_t cur = sh.cur[tid], spilled = sh.spill[tid];
    const bool ok = 4 * cur + left <= 4 * (cap - kCSpillUnits) && spilled <= static_cast<uint32_t>(kCSpillUnits);
    if (ok) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(my_regions + int64_t(tid) * cap + cur);
      for (uint32_t k = 0; k < left; ++k) dst[k] = sh.bin[tid * G::kStride + from + k];
    } else {
      __hip_atomic_store(overflow, kOverflowRegion, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    counts[int64_t(tid) * (a.nblocks + 1) + blockIdx.x] = ok ? (4 * cur + left) / RW | spilled << 24 : 0;  // (records | spilled << 24)
  }
}

// One workgroup per partition; every wave walks whole regions (segments) of the partition.
// LDS table: 32-bit tags (fmix32 of the key, a bijection, so tag equality is key equality),
// 0xFFFFFFFF = free; the null key and the one key whose tag equals the free marker live in
// two side slots.  Per step a lane takes 4 records: 4 unconditional 16-byte loads, then the
// 4 first probes are issued together; only mismatches enter the probe loop.
constexpr uint32_t kTagFree = 0xFFFFFFFFu;
constexpr int kSideNull = kSlots, kSideFree = kSlots + 1;

// The LDS table is bucketised: 1024 buckets of 4 tags (one ds_read_b128 fetches a bucket).  At
// <= 50 % load a key is almost always found in, or inserted into, its home bucket, so the
// expensive divergent probe loop of plain linear probing disappears.
constexpr int kBucket = 4;
constexpr int kBuckets = kSlots / kBucket;
constexpr int kBucketBits = 10;  // second bucket: the next kBucketBits bits of the tag
static_assert(kBuckets == 1 << kBucketBits, "bucket index width");

__device__ __forceinline__ int MatchBucket(const uint4& b, uint32_t h) {
  return b.x == h ? 0 : b.y == h ? 1 : b.z == h ? 2 : b.w == h ? 3 : -1;
}
// Slow path: (re)read buckets until the key is found or inserted.  Returns slot or -1 (full).
// Probe order of a key: its first bucket, its second bucket, then linearly on from the second.
// Entries of a bucket fill from index 0 and are never removed, so every inserter of a key
// targets the same entry and the loser of a CAS race re-reads the bucket.
__device__ __forceinline__ int ProbeInsert(unsigned int* s_tag, unsigned int* s_groups, uint32_t h, uint32_t b1, uint32_t b2,
                                           unsigned int limit) {
  uint32_t bucket = b1;
  for (int probes = 0; probes <= kBuckets; ++probes) {
    for (;;) {
      const uint4 b = *reinterpret_cast<const uint4*>(&s_tag[bucket * kBucket]);
      int j = MatchBucket(b, h);
      if (j >= 0) return static_cast<int>(bucket * kBucket + j);
      j = MatchBucket(b, kTagFree);
      if (j < 0) break;  // bucket full: next bucket
      if (*s_groups >= limit) return -1;
      const unsigned int old = atomicCAS(&s_tag[bucket * kBucket + j], kTagFree, h);
      if (old == kTagFree) { atomicAdd(s_groups, 1u); return static_cast<int>(bucket * kBucket + j); }
      if (old == h) return static_cast<int>(bucket * kBucket + j);
      // another key took the free entry: look at the bucket again
    }
    bucket = probes == 0 ? b2 : (bucket + 1) & (kBuckets - 1);
  }
  return -1;
}

constexpr int kAggUnroll = 4;                  // records per lane and step (local path)
constexpr int kPaUnroll = 4;                   // records per lane and step (partition aggregate; 5: +5 %, 6 spills: +10 %)
constexpr int kAggStep = 64 * kPaUnroll;       // records per wave and step of the partition aggregate
constexpr int kMaxSegments = 1024 + 1;        // scatter workgroups (kMaxFlags) + one (once the bin scatters' spill list: always empty now, records that find their bin full sit at the region tops)

// Issue the (unconditional, clamped) streaming loads of one step: records [i0, i0 + kAggStep) of `seg`.
template <int N>
__device__ __forceinline__ void AggLoad(uint4 (&rec)[N], const uint4* seg, uint32_t i0, uint32_t cnt, int lane) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  const uint32_t last = cnt ? cnt - 1 : 0;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const uint32_t i = i0 + lane + 64 * u;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(&seg[i < cnt ? i : last]));
    rec[u] = make_uint4(v.x, v.y, v.z, v.w);
  }
}

// Fold one step into the LDS table.  Returns false when the table overflowed.
template <int N>
__device__ __forceinline__ bool AggFold(const uint4 (&rec)[N], const bool (&live)[N], int acc_double,
                                        unsigned int* s_tag, unsigned long long* s_sum, uint32_t* s_first,
                                        unsigned int* s_cnt, unsigned int* s_groups, unsigned int limit = kSlotLimit) {
  bool ok = true;
  uint32_t h[N];
  int slot[N];
  uint4 home1[N], home2[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    h[u] = Fmix32(rec[u].x);
    home1[u] = *reinterpret_cast<const uint4*>(&s_tag[(h[u] & (kBuckets - 1)) * kBucket]);
    home2[u] = *reinterpret_cast<const uint4*>(&s_tag[((h[u] >> kBucketBits) & (kBuckets - 1)) * kBucket]);
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (!live[u]) { slot[u] = -2; continue; }
    if (rec[u].y & kRecKeyNull) slot[u] = kSideNull;
    else if (h[u] == kTagFree) slot[u] = kSideFree;
    else {
      const uint32_t b1 = h[u] & (kBuckets - 1), b2 = (h[u] >> kBucketBits) & (kBuckets - 1);
      const int j1 = MatchBucket(home1[u], h[u]), j2 = MatchBucket(home2[u], h[u]);
      if (j1 >= 0) slot[u] = static_cast<int>(b1 * kBucket) + j1;
      else if (j2 >= 0) slot[u] = static_cast<int>(b2 * kBucket) + j2;
      else slot[u] = ProbeInsert(s_tag, s_groups, h[u], b1, b2, limit);
    }
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (slot[u] == -2) continue;
    if (slot[u] < 0) { ok = false; continue; }
    const uint4 r = rec[u];
    atomicMin(&s_first[slot[u]], r.y & kRecRowMask);
    if (r.y & kRecValValid) {
      const unsigned long long bits = (static_cast<unsigned long long>(r.w) << 32) | r.z;
      if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot[u]]), __longlong_as_double(bits));
      else atomicAdd(&s_sum[slot[u]], bits);
      atomicAdd(&s_cnt[slot[u]], 1u);
    }
  }
  return ok;
}

// The local pre-aggregation's table for the headline shape: tags are the RAW keys, the slot of a key is
// (key ^ key >> 12) & 4095 -- dense small keys (ids, codes, dates: what few-groups-per-range input mostly
// is) sit at their own value without a collision -- and a miss probes linearly, at most kRawProbes slots.
// ~15 vector instructions per record where the hashed, bucketised fold above needs ~100 (two multiplies of
// the finaliser, eight tag compares), which is what bounds that fold.  A chain longer than kRawProbes, or
// the one key that equals the free marker, fails the attempt: the caller gives up and the partition path,
// which takes any key, runs instead.
constexpr int kRawProbes = 8;
__device__ __forceinline__ uint32_t RawSlot(uint32_t x) { return (x ^ (x >> 12)) & (kSlots - 1); }
__device__ __attribute__((noinline)) int ProbeRaw(unsigned int* s_tag, unsigned int* s_groups, uint32_t x, uint32_t slot,
                                                  unsigned int limit) {
  for (int probes = 0; probes < kRawProbes; ++probes) {
    for (;;) {
      const unsigned int t = s_tag[slot];
      if (t == x) return static_cast<int>(slot);
      if (t != kTagFree) break;  // another key lives here: next slot
      if (*s_groups >= limit) return -1;
      const unsigned int old = atomicCAS(&s_tag[slot], kTagFree, x);
      if (old == kTagFree) { atomicAdd(s_groups, 1u); return static_cast<int>(slot); }
      if (old == x) return static_cast<int>(slot);
      // another key took it: look again (and move on)
    }
    slot = (slot + 1) & (kSlots - 1);
  }
  return -1;
}
template <int N>
__device__ __forceinline__ bool AggFoldRaw(const uint4 (&rec)[N], const bool (&live)[N], int acc_double,
                                           unsigned int* s_tag, unsigned long long* s_sum, uint32_t* s_first,
                                           unsigned int* s_cnt, unsigned int* s_groups, unsigned int limit) {
  bool ok = true;
  uint32_t home[N];
  unsigned int t0[N];
  int slot[N];
#pragma unroll
  for (int u = 0; u < N; ++u) { home[u] = RawSlot(rec[u].x); t0[u] = s_tag[home[u]]; }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (!live[u]) { slot[u] = -2; continue; }
    if (t0[u] == rec[u].x && rec[u].x != kTagFree) slot[u] = static_cast<int>(home[u]);
    else if (rec[u].x == kTagFree) slot[u] = -1;
    else slot[u] = ProbeRaw(s_tag, s_groups, rec[u].x, home[u], limit);
  }
#pragma unroll
  for (int u = 0; u < N; ++u) {
    if (slot[u] == -2) continue;
    if (slot[u] < 0) { ok = false; continue; }
    const uint4 r = rec[u];
    atomicMin(&s_first[slot[u]], r.y & kRecRowMask);
    if (r.y & kRecValValid) {
      const unsigned long long bits = (static_cast<unsigned long long>(r.w) << 32) | r.z;
      if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot[u]]), __longlong_as_double(bits));
      else atomicAdd(&s_sum[slot[u]], bits);
      atomicAdd(&s_cnt[slot[u]], 1u);
    }
  }
  return ok;
}

template <int N>
__device__ __forceinline__ bool AggFoldStep(const uint4 (&rec)[N], uint32_t i0, uint32_t cnt, int lane,
                                            int acc_double, unsigned int* s_tag, unsigned long long* s_sum,
                                            uint32_t* s_first, unsigned int* s_cnt, unsigned int* s_groups) {
  bool live[N];
#pragma unroll
  for (int u = 0; u < N; ++u) live[u] = i0 + lane + 64 * u < cnt;
  return AggFold(rec, live, acc_double, s_tag, s_sum, s_first, s_cnt, s_groups);
}

// A wave owns segments wave, wave + 16, ...; their fills are fetched once (one per lane) and
// the steps are software-pipelined: the loads of step s + 1 are in flight while step s is
// folded into LDS, so a wave never sits idle for a dependent fill -> records round trip.
__global__ __launch_bounds__(kPartThreads) void k_part_aggregate_direct(const uint4* regions, const uint32_t* counts,
                                                                        int nblocks, int64_t cap, int k_bits,
                                                                        int acc_double, GroupTable t,
                                                                        unsigned long long* first_bitmap,
                                                                        unsigned int* overflow, const unsigned int* run_if,
                                                                        int64_t slice0 /* first slot of this sub-problem's table slice */,
                                                                        int64_t total_slots /* of all slices; 0 = not the first slice */) {
  __shared__ __attribute__((aligned(16))) unsigned int s_tag[kSlots + 4];
  __shared__ unsigned long long s_sum[kSlots + 2];
  __shared__ uint32_t s_first[kSlots + 2];
  __shared__ unsigned int s_cnt[kSlots + 2];
  __shared__ unsigned int s_groups;
  __shared__ unsigned int s_abort;
  if (run_if && !*run_if) return;  // the local path already built the table
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = blockIdx.x;
  constexpr int kWaves = kPartThreads / 64;
  // fills of this wave's first 64 segments, fetched before the table is cleared
  uint32_t my_cnt = 0;
  {
    const int b = wave + kWaves * lane;
    if (b < nblocks) my_cnt = counts[int64_t(p) * nblocks + b];
    if (my_cnt > cap) my_cnt = static_cast<uint32_t>(cap);  // (a spill list that overflowed; the attempt is abandoned anyway)
  }
  for (int s = tid; s < kSlots + 2; s += kPartThreads) { s_tag[s] = kTagFree; s_sum[s] = 0; s_first[s] = kNoRow; s_cnt[s] = 0; }
  if (tid == 0) { s_groups = 0; s_abort = 0; }
  __syncthreads();
  bool failed = false;
  // A table that overflows (more groups in the partition than LDS slots) dooms the attempt: the
  // wave raises kOverflowTable, its workgroup stops at the next step and every other workgroup at
  // its next segment, so a hopeless attempt costs ~0.1 ms instead of a full pass.
  volatile unsigned int* abort_lds = &s_abort;
  auto give_up = [&]() {
    if (lane == 0) { *abort_lds = 1; __hip_atomic_store(overflow, kOverflowTable, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  };
  for (int jb = 0; wave + kWaves * jb < nblocks; jb += 64) {
    if (jb) {
      const int b = wave + kWaves * (jb + lane);
      my_cnt = b < nblocks ? counts[int64_t(p) * nblocks + b] : 0;
      if (my_cnt > cap) my_cnt = static_cast<uint32_t>(cap);
    }
    const int left = (nblocks - wave - kWaves * jb + kWaves - 1) / kWaves;
    const int nseg = left < 64 ? left : 64;
    const int64_t seg_stride = int64_t(kWaves) * kP * cap;
    const uint4* seg = regions + (int64_t(wave + kWaves * jb) * kP + p) * cap;  // segment k: seg + k * seg_stride
    int k = 0;
    uint32_t i0 = 0, cnt = __shfl(my_cnt, 0);
    uint4 ra[kPaUnroll], rb[kPaUnroll];
    AggLoad(ra, seg, i0, cnt, lane);
    for (;;) {
      // step after (k, i0): the rest of the segment, else the next segment (a step of an empty
      // or out-of-range segment loads one valid record and folds nothing)
      int k2 = k;
      uint32_t i2 = i0 + kAggStep, cnt2 = cnt;
      if (i2 >= cnt) { k2 = k + 1; i2 = 0; cnt2 = k2 < nseg ? __shfl(my_cnt, k2) : 0; }
      AggLoad(rb, seg + (k2 < nseg ? k2 : k) * seg_stride, i2, cnt2, lane);
      failed |= !AggFoldStep(ra, i0, cnt, lane, acc_double, s_tag, s_sum, s_first, s_cnt, &s_groups);
      if (__any(failed)) { give_up(); break; }
      if (*abort_lds || (i2 == 0 && (k2 & 3) == 0 && __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) break;
      if (k2 >= nseg) break;
      k = k2;
      i0 = i2 + kAggStep; cnt = cnt2;
      if (i0 >= cnt2) { k = k2 + 1; i0 = 0; cnt = k < nseg ? __shfl(my_cnt, k) : 0; }
      AggLoad(ra, seg + (k < nseg ? k : k2) * seg_stride, i0, cnt, lane);
      failed |= !AggFoldStep(rb, i2, cnt2, lane, acc_double, s_tag, s_sum, s_first, s_cnt, &s_groups);
      if (__any(failed)) { give_up(); break; }
      if (*abort_lds) break;
      if (k >= nseg) break;
    }
    if (*abort_lds) break;
  }
  __syncthreads();
  if (*abort_lds || __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // nothing reads the table
  const int64_t g0 = slice0 + int64_t(p) * kSlots;
  for (int s = tid; s < kSlots; s += kPartThreads) {
    const unsigned int tag = s_tag[s];
    // the shared numbering / emit kernels speak fmix64 of the packed key word
    t.hkeys[g0 + s] = tag == kTagFree ? kEmpty : Fmix64(uint64_t(Fmix32Inv(tag)));
    const uint32_t fr = s_first[s];
    t.first[g0 + s] = fr;
    t.sums[g0 + s] = s_sum[s];
    t.counts[g0 + s] = s_cnt[s];
    // mark the group's first row for the first-appearance numbering (was a separate kernel)
    if (fr != kNoRow) atomicOr(&first_bitmap[fr >> 6], 1ull << (fr & 63));
  }
  // side slots -> two spare regular slots of the global table: partition p's slice is only
  // kSlotLimit full, so the (at most two) extra groups go to free slots of the slice.
  __syncthreads();
  if (tid == 0) {
    for (int side = kSideNull; side <= kSideFree; ++side) {
      if (s_first[side] == kNoRow) continue;
      const uint64_t x = side == kSideNull ? (uint64_t(1) << k_bits) : uint64_t(Fmix32Inv(kTagFree));
      for (int s = 0; s < kSlots; ++s) {
        if (s_tag[s] == kTagFree) {
          s_tag[s] = 0;  // claimed (value irrelevant from here on)
          t.hkeys[g0 + s] = Fmix64(x); t.first[g0 + s] = s_first[side]; t.sums[g0 + s] = s_sum[side];
          t.counts[g0 + s] = s_cnt[side];
          atomicOr(&first_bitmap[s_first[side] >> 6], 1ull << (s_first[side] & 63));
          break;
        }
      }
    }
  }
  if (p == 0 && tid < 2 && total_slots > 0) {  // the two special slots behind the last slice stay empty on this path
    const int64_t T = total_slots;
    t.hkeys[T + tid] = kEmpty; t.first[T + tid] = kNoRow; t.sums[T + tid] = 0; t.counts[T + tid] = 0;
  }
}


// ---- the partition aggregate, pipelined three steps deep -------------------------------------------------------
// Same table, same output as k_part_aggregate_direct (which it replaces behind the bin scatter).  A wave keeps
// three 256-record steps in flight -- the loads are unconditional, so the compiler counts them: vmcnt(11..8)
// before a fold -- and the LDS abort flag is read with relaxed atomic loads: a `volatile` read made the compiler
// drain every counter, i.e. the prefetched steps, once per step (k_part_aggregate_direct has it).  Keys of a small
// range index the table directly (see `direct` below); the others go through the hashed fold (AggFold).
// Measured and NOT kept: a parallel first-attempt insert (compare-and-swap on the first free entry of the
// fetched buckets, probe loop as a cold call) was 20 % slower than the serial probe loop.
struct AggIter { int k; uint32_t i0, cnt; };
// Directly indexed table: offset o = (key >> 9) - base lives in slot bitreverse12(o), so that a slice filled to
// a fraction holds its groups spread over all of its slots (the kernels that walk the finished table split
// it by slot ranges: with the groups packed at the front a few workgroups did all the work, 29 -> 44 us).
static_assert(kSlots == 4096, "DirectSlot reverses 12 bits");
__device__ __forceinline__ uint32_t DirectSlot(uint32_t o) { return __brev(o) >> 20; }
__global__ __launch_bounds__(kPartThreads) void k_part_aggregate_piped(const uint4* regions, const uint32_t* counts,
                                                                       int nblocks, int64_t cap, int k_bits,
                                                                       int acc_double, GroupTable t,
                                                                       unsigned long long* first_bitmap,
                                                                       unsigned int* overflow, const unsigned int* run_if,
                                                                       int64_t slice0, int64_t total_slots,
                                                                       const unsigned int* key_minmax, uint32_t k_flip,
                                                                       const unsigned int* hashed_if /* optional device word */) {
  __shared__ __attribute__((aligned(16))) unsigned int s_tag[kSlots + 4];
  __shared__ unsigned long long s_sum[kSlots + 2];
  __shared__ uint32_t s_first[kSlots + 2];
  __shared__ unsigned int s_cnt[kSlots + 2];
  __shared__ uint32_t s_fill[kMaxSegments];
  __shared__ unsigned int s_groups;
  __shared__ unsigned int s_abort;
  if (run_if && !*run_if) return;  // the local path already built the table
  if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // the scatter gave up
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform values stay in scalar registers: uniform loop control)
  const int p = blockIdx.x;
  constexpr int kWaves = kPartThreads / 64;
  for (int b = tid; b < nblocks; b += kPartThreads) s_fill[b] = counts[int64_t(p) * nblocks + b];  // records | records at the region's top << 24
  for (int s = tid; s < kSlots + 2; s += kPartThreads) { s_tag[s] = kTagFree; s_sum[s] = 0; s_first[s] = kNoRow; s_cnt[s] = 0; }
  if (tid == 0) { s_groups = 0; s_abort = 0; }
  // Keys of a small range (key >> 9 spans at most kSlots values): the slot IS (key >> 9) - base, no tags, no
  // probing -- the fold is three LDS atomics per record.  The hashed fold costs ~100 vector instructions per
  // record (two bucket compares, the finaliser's multiplies): 127 us of issue slots per 100 M rows, which is
  // what kept it from hiding under the record stream.
  const uint32_t key_lo = key_minmax[0], key_hi = key_minmax[1];
  const uint32_t dbase = key_lo >> kPBits;
  // (heavy hitters taken out before the scatter are injected into free slots of the finished table later: a
  // directly indexed slice may have none, so calls with heavy hitters keep the hashed table, at most 87.5 % full)
  const bool direct = key_lo <= key_hi && (key_hi >> kPBits) - dbase < static_cast<uint32_t>(kSlots) &&
                      !(hashed_if && *hashed_if);
  __syncthreads();
  // (relaxed atomic loads, not volatile ones: a volatile access makes the compiler drain every counter, i.e. the prefetched steps)
  auto aborted = [&]() { return __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };
  const int nseg = wave < nblocks ? (nblocks - wave + kWaves - 1) / kWaves : 0;  // this wave's segments: wave, wave + 16, ...
  if (nseg > 0) {
    const int64_t seg_stride = int64_t(kWaves) * kP * cap;
    const uint4* seg0 = regions + (int64_t(wave) * kP + p) * cap;  // segment k: seg0 + k * seg_stride
    // the step after `it`; empty segments are skipped; k == nseg: past the end (such a step loads record 0 of
    // the wave's first segment slot and folds nothing)
    auto advance = [&](AggIter it) -> AggIter {
      it.i0 += kAggStep;
      if (it.i0 >= it.cnt) {
        it.i0 = 0;
        do { ++it.k; it.cnt = it.k < nseg ? __builtin_amdgcn_readfirstlane(s_fill[wave + kWaves * it.k]) & 0xFFFFFFu : 0u; } while (it.k < nseg && it.cnt == 0);
      }
      return it;
    };
    auto load = [&](uint4 (&r)[kPaUnroll], const AggIter& it) {
      const bool in = it.k < nseg;
      AggLoad(r, seg0 + (in ? it.k : 0) * seg_stride, in ? it.i0 : 0u, in ? it.cnt : 0u, lane);
    };
    bool failed = false;
    auto fold = [&](const uint4 (&r)[kPaUnroll], const AggIter& it) {
      bool live[kPaUnroll];
#pragma unroll
      for (int u = 0; u < kPaUnroll; ++u) live[u] = it.k < nseg && it.i0 + lane + 64 * u < it.cnt;
      if (direct) {
#pragma unroll
        for (int u = 0; u < kPaUnroll; ++u) {
          if (!live[u]) continue;
          const uint32_t slot = DirectSlot(((r[u].x ^ k_flip) >> kPBits) - dbase);
          atomicMin(&s_first[slot], r[u].y & kRecRowMask);
          if (r[u].y & kRecValValid) {
            const unsigned long long bits = (static_cast<unsigned long long>(r[u].w) << 32) | r[u].z;
            if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot]), __longlong_as_double(bits));
            else atomicAdd(&s_sum[slot], bits);
            atomicAdd(&s_cnt[slot], 1u);
          }
        }
      } else {
        failed |= !AggFold(r, live, acc_double, s_tag, s_sum, s_first, s_cnt, &s_groups);
      }
    };
    AggIter i0{-1, 0, 0};
    i0 = advance(i0);  // the first step of the first non-empty segment
    AggIter i1 = advance(i0), i2 = advance(i1);
    uint4 r0[kPaUnroll], r1[kPaUnroll], r2[kPaUnroll];
    load(r0, i0); load(r1, i1); load(r2, i2);
    while (i0.k < nseg) {  // three steps per trip, no exit inside (steps past the end fold nothing)
      fold(r0, i0); i0 = advance(i2); load(r0, i0);
      fold(r1, i1); i1 = advance(i0); load(r1, i1);
      fold(r2, i2); i2 = advance(i1); load(r2, i2);
      const bool full = aborted() || __hip_atomic_load(&s_groups, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > kSlotLimit;
      if (__any(failed || full)) {
        if (lane == 0) {
          __hip_atomic_store(&s_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(overflow, kOverflowTable, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
      }
    }
  }
  // the records that found their bin full: at the top of each region, a few per region at most
  for (int b = wave; b < nblocks && !aborted(); b += kWaves) {
    const uint32_t ns = __builtin_amdgcn_readfirstlane(s_fill[b]) >> 24;
    if (ns == 0) continue;
    const uint4* top = regions + (int64_t(b) * kP + p) * cap + (cap - 1);
    uint4 one[1];
    bool live1[1];
    live1[0] = static_cast<uint32_t>(lane) < ns;
    one[0] = live1[0] ? *(top - lane) : make_uint4(0, 0, 0, 0);
    bool ok1 = true;
    if (direct) {
      if (live1[0]) {
        const uint32_t slot = DirectSlot(((one[0].x ^ k_flip) >> kPBits) - dbase);
        atomicMin(&s_first[slot], one[0].y & kRecRowMask);
        if (one[0].y & kRecValValid) {
          const unsigned long long bits = (static_cast<unsigned long long>(one[0].w) << 32) | one[0].z;
          if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot]), __longlong_as_double(bits));
          else atomicAdd(&s_sum[slot], bits);
          atomicAdd(&s_cnt[slot], 1u);
        }
      }
    } else {
      ok1 = AggFold(one, live1, acc_double, s_tag, s_sum, s_first, s_cnt, &s_groups);
    }
    if (__any(!ok1) && lane == 0) {
      __hip_atomic_store(&s_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_store(overflow, kOverflowTable, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  if (aborted() || __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // nothing reads the table
  const int64_t g0 = slice0 + int64_t(p) * kSlots;
  for (int s = tid; s < kSlots; s += kPartThreads) {
    const uint32_t fr = s_first[s];
    // the shared numbering / emit kernels speak fmix64 of the packed key word
    if (direct) {
      // the key of (partition p, slot s): high part s + base, low part p XOR the high part's hashed bits (BinPartOf)
      const uint32_t high = DirectSlot(static_cast<uint32_t>(s)) + dbase;  // (the bit reversal is its own inverse)
      const uint32_t kk = (high << kPBits) | (static_cast<uint32_t>(p) ^ ((high * 0x9E3779B1u) >> (32 - kPBits)));
      t.hkeys[g0 + s] = fr == kNoRow ? kEmpty : Fmix64(uint64_t(kk ^ k_flip));
    } else {
      const unsigned int tag = s_tag[s];
      t.hkeys[g0 + s] = tag == kTagFree ? kEmpty : Fmix64(uint64_t(Fmix32Inv(tag)));
    }
    t.first[g0 + s] = fr;
    t.sums[g0 + s] = s_sum[s];
    t.counts[g0 + s] = s_cnt[s];
    // mark the group's first row for the first-appearance numbering
    if (fr != kNoRow) atomicOr(&first_bitmap[fr >> 6], 1ull << (fr & 63));
  }
  // side slots -> two spare regular slots of the global table (see k_part_aggregate_direct)
  __syncthreads();
  if (tid == 0) {
    for (int side = kSideNull; side <= kSideFree; ++side) {
      if (s_first[side] == kNoRow) continue;
      const uint64_t x = side == kSideNull ? (uint64_t(1) << k_bits) : uint64_t(Fmix32Inv(kTagFree));
      for (int s = 0; s < kSlots; ++s) {
        if (s_tag[s] == kTagFree) {
          s_tag[s] = 0;  // claimed (value irrelevant from here on)
          t.hkeys[g0 + s] = Fmix64(x); t.first[g0 + s] = s_first[side]; t.sums[g0 + s] = s_sum[side];
          t.counts[g0 + s] = s_cnt[side];
          atomicOr(&first_bitmap[s_first[side] >> 6], 1ull << (s_first[side] & 63));
          break;
        }
      }
    }
  }
  if (p == 0 && tid < 2 && total_slots > 0) {  // the two special slots behind the last slice stay empty on this path
    const int64_t T = total_slots;
    t.hkeys[T + tid] = kEmpty; t.first[T + tid] = kNoRow; t.sums[T + tid] = 0; t.counts[T + tid] = 0;
  }
}

// ---- the partition aggregate over compact records (k_part_scatter_compact) ----------------------------------
// Always the directly indexed table (the compact record IS a slot offset): no tags, 68 KB (512 partitions) or
// 132 KB (256 partitions) of LDS.  A wave walks the regions of workgroups wave, wave + kWaves, ...; region b's rows
// start at b * rows_per_block.  RW = 2: a lane loads 16 bytes = two records; RW = 3: one 12-byte record.  Three
// steps in flight as in k_part_aggregate_piped; the records at the tops of the regions (16-byte records of rows
// that found their bin full, a fraction of a per cent) are folded at the end.
template <int RW> struct CUnit;
template <> struct CUnit<2> { BinU32x4 w; };
template <> struct CUnit<3> { __attribute__((ext_vector_type(3), aligned(4))) unsigned int w; };
template <int RW, int PB>
__global__ __launch_bounds__(kPartThreads) void k_part_aggregate_compact(const uint4* regions, const uint32_t* counts,
                                                                         int nblocks, int64_t cap, int acc_double, GroupTable t,
                                                                         unsigned long long* first_bitmap,
                                                                         unsigned int* overflow, int64_t total_slots,
                                                                         uint32_t c_base, uint32_t k_flip, uint32_t rows_per_block) {
  using G = CGeo<PB>;
  __shared__ unsigned long long s_sum[G::kSlots];
  __shared__ uint32_t s_first[G::kSlots];
  __shared__ unsigned int s_cnt[G::kSlots];
  __shared__ uint32_t s_fill[kMaxSegments];
  if (__hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // the scatter gave up
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = blockIdx.x;
  constexpr int kWaves = kPartThreads / 64;
  constexpr int kUnitRecs = RW == 2 ? 2 : 1;                     // records per loaded unit
  constexpr uint32_t kStepUnits = 64 * kPaUnroll;                // units per wave and step
  const int nseg_all = nblocks + 1;
  for (int b = tid; b < nblocks; b += kPartThreads) s_fill[b] = counts[int64_t(p) * nseg_all + b];  // records | spilled << 24
  for (int s = tid; s < G::kSlots; s += kPartThreads) { s_sum[s] = 0; s_first[s] = kNoRow; s_cnt[s] = 0; }
  __syncthreads();
  auto fold_one = [&](uint32_t meta, uint32_t lo32, uint32_t hi32, uint32_t seg_lo) {
    const uint32_t slot = G::Slot(meta & (G::kValid - 1u));
    atomicMin(&s_first[slot], seg_lo + (meta >> (G::kSlotBits + 1)));
    if (meta & G::kValid) {
      const unsigned long long bits = RW == 2 ? static_cast<unsigned long long>(static_cast<long long>(static_cast<int>(lo32)))
                                              : (static_cast<unsigned long long>(hi32) << 32) | lo32;
      if (RW == 3 && acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot]), __longlong_as_double(bits));
      else atomicAdd(&s_sum[slot], bits);
      atomicAdd(&s_cnt[slot], 1u);
    }
  };
  const int nseg = wave < nblocks ? (nblocks - wave + kWaves - 1) / kWaves : 0;  // this wave's regions: wave, wave + kWaves, ...
  if (nseg > 0) {
    const int64_t seg_stride = int64_t(kWaves) * G::kParts * cap;
    const uint4* seg0 = regions + (int64_t(wave) * G::kParts + p) * cap;
    struct It { int k; uint32_t i0, units, recs; };
    auto advance = [&](It it) -> It {
      it.i0 += kStepUnits;
      if (it.i0 >= it.units) {
        it.i0 = 0;
        do {
          ++it.k;
          it.recs = it.k < nseg ? __builtin_amdgcn_readfirstlane(s_fill[wave + kWaves * it.k]) & 0xFFFFFFu : 0u;
          it.units = (it.recs + kUnitRecs - 1) / kUnitRecs;
        } while (it.k < nseg && it.recs == 0);
      }
      return it;
    };
    auto load = [&](CUnit<RW> (&r)[kPaUnroll], const It& it) {
      const bool in = it.k < nseg;
      const uint32_t* base = reinterpret_cast<const uint32_t*>(seg0 + (in ? it.k : 0) * seg_stride);
      const uint32_t units = in ? it.units : 0u, i0 = in ? it.i0 : 0u;
      const uint32_t last = units ? units - 1 : 0;
#pragma unroll
      for (int u = 0; u < kPaUnroll; ++u) {
        const uint32_t i = i0 + lane + 64 * u;
        const uint32_t ii = i < units ? i : last;
        r[u].w = __builtin_nontemporal_load(reinterpret_cast<const decltype(r[u].w)*>(base + (RW == 2 ? 4 : 3) * ii));
      }
    };
    auto fold = [&](const CUnit<RW> (&r)[kPaUnroll], const It& it) {
      if (it.k >= nseg) return;  // (wave-uniform)
      const uint32_t seg_lo = static_cast<uint32_t>(wave + kWaves * it.k) * rows_per_block;
#pragma unroll
      for (int u = 0; u < kPaUnroll; ++u) {
        const uint32_t i = it.i0 + lane + 64 * u;
        if constexpr (RW == 2) {
          if (2 * i < it.recs) fold_one(r[u].w.x, r[u].w.y, 0u, seg_lo);
          if (2 * i + 1 < it.recs) fold_one(r[u].w.z, r[u].w.w, 0u, seg_lo);
        } else {
          if (i < it.recs) fold_one(r[u].w.x, r[u].w.y, r[u].w.z, seg_lo);
        }
      }
    };
    It i0{-1, 0, 0, 0};
    i0 = advance(i0);
    It i1 = advance(i0), i2 = advance(i1);
    CUnit<RW> r0[kPaUnroll], r1[kPaUnroll], r2[kPaUnroll];
    load(r0, i0); load(r1, i1); load(r2, i2);
    while (i0.k < nseg) {  // three steps per trip, no exit inside (steps past the end fold nothing)
      fold(r0, i0); i0 = advance(i2); load(r0, i0);
      fold(r1, i1); i1 = advance(i0); load(r1, i1);
      fold(r2, i2); i2 = advance(i1); load(r2, i2);
    }
  }
  // the records that found their bin full: 16-byte records {key word, row | flags, value} at the top of each region
  for (int b = wave; b < nblocks; b += kWaves) {
    const uint32_t ns = __builtin_amdgcn_readfirstlane(s_fill[b]) >> 24;
    if (ns == 0) continue;
    const uint4* top = regions + (int64_t(b) * G::kParts + p) * cap + (cap - 1);
    if (static_cast<uint32_t>(lane) < ns) {
      const uint4 q = *(top - lane);
      const uint32_t slot = G::Slot((((q.x ^ k_flip) >> PB) - c_base) & (G::kValid - 1u));
      atomicMin(&s_first[slot], q.y & kRecRowMask);
      if (q.y & kRecValValid) {
        const unsigned long long bits = (static_cast<unsigned long long>(q.w) << 32) | q.z;
        if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[slot]), __longlong_as_double(bits));
        else atomicAdd(&s_sum[slot], bits);
        atomicAdd(&s_cnt[slot], 1u);
      }
    }
  }
  __syncthreads();
  const int64_t g0 = int64_t(p) * G::kSlots;
  for (int s = tid; s < G::kSlots; s += kPartThreads) {
    const uint32_t fr = s_first[s];
    // the key of (partition p, slot s): high part = slot offset + base, low part p XOR the high part's hashed bits (CGeo::PartOf)
    const uint32_t high = G::Slot(static_cast<uint32_t>(s)) + c_base;  // (the bit reversal is its own inverse)
    const uint32_t kk = (high << PB) | (static_cast<uint32_t>(p) ^ ((high * 0x9E3779B1u) >> (32 - PB)));
    t.hkeys[g0 + s] = fr == kNoRow ? kEmpty : Fmix64(uint64_t(kk ^ k_flip));
    t.first[g0 + s] = fr;
    t.sums[g0 + s] = s_sum[s];
    t.counts[g0 + s] = s_cnt[s];
    if (fr != kNoRow) atomicOr(&first_bitmap[fr >> 6], 1ull << (fr & 63));
  }
  if (p == 0 && tid < 2) {  // the two special slots behind the table stay empty on this path
    const int64_t T = total_slots;
    t.hkeys[T + tid] = kEmpty; t.first[T + tid] = kNoRow; t.sums[T + tid] = 0; t.counts[T + tid] = 0;
  }
}

// ---- local path: per-workgroup LDS pre-aggregation ----------------------------------------------------
// When a workgroup's row range holds few distinct keys (low-cardinality or clustered keys) the
// records never need to leave the CU: every workgroup folds its rows straight into the LDS table
// of k_part_aggregate_direct and the (small) tables are merged into the global table afterwards.
// The attempt is optimistic: a workgroup whose table exceeds kLocalLimit raises ITS OWN flag word
// and stops (a flag shared by all workgroups costs ~30 ns per store: same-address device-scope
// stores are serialised at the memory side -- 130 us when every wave of a doomed attempt raises
// it).  The first kernel behind ORs the flags into status[1]; the merge kernels then turn into
// no-ops and the partition path (already queued, gated on the same word) takes over.  No host
// round trip is spent on the decision.
struct LocalStage { unsigned int* tag; uint32_t* first; unsigned long long* sum; unsigned int* cnt; };
constexpr int kStageSlots = kSlots + 2;
// The local attempt gives up at half load: beyond it the inserts crawl (long probe chains, CAS
// races of 16 waves) and a doomed attempt must fail fast, it is pure overhead for the partition path.
#ifndef LOCAL_LIMIT
#define LOCAL_LIMIT (kSlots / 2)
#endif
constexpr unsigned int kLocalLimit = LOCAL_LIMIT;
static size_t LocalStageBytes(int nblocks) { return size_t(20) * kStageSlots * nblocks; }

struct LocalShared {
  __attribute__((aligned(16))) unsigned int tag[kSlots + 4];
  unsigned long long sum[kSlots + 2];
  uint32_t first[kSlots + 2];
  unsigned int cnt[kSlots + 2];
  unsigned int groups;
  unsigned int stop;  // some wave of this workgroup gave up
};
__device__ __forceinline__ void LocalInit(LocalShared& sh, int tid) {
  for (int s = tid; s < kSlots + 2; s += kDThreads) { sh.tag[s] = kTagFree; sh.sum[s] = 0; sh.first[s] = kNoRow; sh.cnt[s] = 0; }
  if (tid == 0) { sh.groups = 0; sh.stop = 0; }
  __syncthreads();
}
__device__ __forceinline__ void LocalGiveUp(LocalShared& sh, unsigned int* wg_flags) {
  if ((threadIdx.x & 63) == 0) { __hip_atomic_store(&sh.stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); wg_flags[blockIdx.x] = 1; }
}
// raw_tags: the table holds raw keys (AggFoldRaw); the staging table, like everything behind it, speaks
// fmix32 tags.  (k_local_merge tells a free slot by its first row, not by its tag, and recovers the key as
// Fmix32Inv(tag): the one key whose tag equals the free marker comes through like any other.)
__device__ __forceinline__ void LocalWriteOut(const LocalShared& sh, const LocalStage& st, int tid, bool raw_tags = false) {
  const int64_t base = int64_t(blockIdx.x) * kStageSlots;
  for (int s = tid; s < kStageSlots; s += kDThreads) {
    const unsigned int tag = s < kSlots ? sh.tag[s] : 0u;
    st.tag[base + s] = raw_tags && s < kSlots && tag != kTagFree ? Fmix32(tag) : tag;
    st.first[base + s] = sh.first[s];
    st.sum[base + s] = sh.sum[s];
    st.cnt[base + s] = sh.cnt[s];
  }
}

// ---- heavy hitters -------------------------------------------------------------------------------------
// With one workgroup per partition in the aggregate, a key that carries 1 % / 10 % / 40 % of the
// rows used to cost 3.5 / 15 / 53 ms per 100 M rows (uniform: 0.85 ms): its partition overflows the
// fixed regions (two-pass fallback) and ONE workgroup then folds all of its records.  So hot keys
// never reach the partitions:
//   1. the optimistic local attempt (above) already counts keys per row range; when it gives up,
//      every workgroup reports the keys it saw >= kHotMin times among its first ~2-4 k selected rows
//      (frequency >~ 0.1 %); the null key counts as a key
//   2. block 0 of k_clear_table_unless deduplicates the candidates into the hot set (<= kHotCap)
//   3. k_hot_aggregate: every workgroup folds the hot rows of its row range into an LDS table
//      indexed by the hot set and writes a copy of the selection mask without them
//   4. the scatter runs on that mask (second launch gated on status[5]; the first, on the caller's
//      mask, is gated off) -- partitions see a uniform remainder
//   5. k_hot_inject (inside k_mark_first's launch) merges the per-workgroup partials and puts each
//      hot group into a free slot of its partition's slice of the table, marking its first row.
// Nothing here costs a host round trip; with no hot key steps 3-5 are two empty launches.
constexpr int kHotPerWg = 128;    // candidates one workgroup may report
constexpr int kHotCap = 2048;     // keys in the hot set
constexpr unsigned int kHotMin = 4;
constexpr int kHotMapSlots = 4096;  // LDS open-addressing map tag -> hot index (load <= 50 %)


// Called by every thread of a local-aggregate workgroup that gave up.  Reports the MOST frequent
// keys of the sample (power-of-two count classes, from the top until kHotPerWg would be exceeded):
// slot order is the same in every workgroup, so "the first kHotPerWg found" would drop the same
// keys everywhere, however hot.
__device__ __forceinline__ void ReportHotCandidates(LocalShared& sh, const HotMem& hot, int tid, bool raw_tags = false) {
  __shared__ unsigned int s_n, s_class[33], s_min_class;
  if (tid == 0) s_n = 0;
  if (tid < 33) s_class[tid] = 0;
  __syncthreads();
  for (int s = tid; s < kSlots; s += kDThreads)
    if (sh.tag[s] != kTagFree && sh.cnt[s] >= kHotMin) atomicAdd(&s_class[32 - __clz(sh.cnt[s])], 1u);
  __syncthreads();
  if (tid == 0) {
    unsigned int total = 0;
    int c = 32;
    for (; c > 0; --c) {
      if (total + s_class[c] > static_cast<unsigned int>(kHotPerWg)) break;
      total += s_class[c];
    }
    s_min_class = c + 1;  // classes >= this fit
  }
  __syncthreads();
  const unsigned int min_class = s_min_class;
  for (int s = tid; s < kSlots; s += kDThreads) {
    if (sh.tag[s] != kTagFree && sh.cnt[s] >= kHotMin && 32u - __clz(sh.cnt[s]) >= min_class) {
      const unsigned int i = atomicAdd(&s_n, 1u);
      if (i < static_cast<unsigned int>(kHotPerWg)) hot.cand[int64_t(blockIdx.x) * kHotPerWg + i] = raw_tags ? Fmix32(sh.tag[s]) : sh.tag[s];
    }
  }
  __syncthreads();
  if (tid == 0)
    hot.cand_n[blockIdx.x] = (s_n < static_cast<unsigned int>(kHotPerWg) ? s_n : kHotPerWg) | (sh.cnt[kSideNull] >= kHotMin ? 0x80000000u : 0u);
}

// Block 0 of k_clear_table_unless, when the local attempt was given up: candidates -> hot set.
// A key is hot only if MANY row ranges report it (>= a quarter of those that gave up, at least 2):
// sorted / clustered keys repeat inside one range and are reported by that range alone -- they
// are not heavy hitters, and treating them as such cost 15 ms on sorted input.
constexpr int kHotVoteSlots = 8192;  // LDS vote table (tags + counts = 64 KiB); inserts stop at 75 % load
__device__ __forceinline__ void BuildHotSet(const HotMem& hot, int nwg, unsigned int* use_hot) {
  __shared__ uint32_t s_tag[kHotVoteSlots];
  __shared__ unsigned int s_votes[kHotVoteSlots];
  __shared__ unsigned int s_used, s_count, s_null_votes, s_reporting;
  const int tid = threadIdx.x;
  for (int i = tid; i < kHotVoteSlots; i += blockDim.x) { s_tag[i] = kTagFree; s_votes[i] = 0; }
  if (tid == 0) { s_used = 0; s_count = 0; s_null_votes = 0; s_reporting = 0; }
  __syncthreads();
  for (int w = tid; w < nwg; w += blockDim.x) {
    const unsigned int cn = hot.cand_n[w];
    const unsigned int n = cn & 0x7FFFFFFFu;
    if (cn) atomicAdd(&s_reporting, 1u);
    if (cn >> 31) atomicAdd(&s_null_votes, 1u);
    for (unsigned int i = 0; i < n && i < static_cast<unsigned int>(kHotPerWg); ++i) {
      const uint32_t tag = hot.cand[int64_t(w) * kHotPerWg + i];
      if (tag == kTagFree) continue;  // (the one key whose tag is the free marker stays cold)
      uint32_t slot = tag & (kHotVoteSlots - 1);
      for (int probes = 0; probes < 64; ++probes) {  // bounded: a crowded table just drops the candidate
        uint32_t old = s_tag[slot];
        if (old == kTagFree) {
          if (s_used >= kHotVoteSlots * 3 / 4) break;
          old = atomicCAS(&s_tag[slot], kTagFree, tag);
          if (old == kTagFree) { atomicAdd(&s_used, 1u); old = tag; }
        }
        if (old == tag) { atomicAdd(&s_votes[slot], 1u); break; }
        slot = (slot + 1) & (kHotVoteSlots - 1);
      }
    }
  }
  __syncthreads();
  const unsigned int quorum = s_reporting / 4 > 2 ? s_reporting / 4 : 2;
  for (int i = tid; i < kHotVoteSlots; i += blockDim.x) {
    if (s_tag[i] != kTagFree && s_votes[i] >= quorum) {
      const unsigned int idx = atomicAdd(&s_count, 1u);
      if (idx < static_cast<unsigned int>(kHotCap)) hot.tags[idx] = s_tag[i];
    }
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned int n = s_count < static_cast<unsigned int>(kHotCap) ? s_count : kHotCap;
    const unsigned int null_hot = s_null_votes >= quorum ? 1u : 0u;
    hot.count[0] = n;
    hot.count[1] = null_hot;
    *use_hot = (n > 0 || null_hot) ? 1u : 0u;
  }
}

// Step 3.  One workgroup per row range of the local attempt (same geometry).
__global__ __launch_bounds__(kDThreads) void k_hot_aggregate(FastArgs a, int acc_double, HotMem hot, const unsigned int* use_hot) {
  if (!*use_hot) return;
  __shared__ uint32_t s_map_tag[kHotMapSlots];
  __shared__ unsigned short s_map_idx[kHotMapSlots];
  __shared__ unsigned long long s_sum[kHotCap + 1];
  __shared__ unsigned int s_cnt[kHotCap + 1];
  __shared__ uint32_t s_first[kHotCap + 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned int nhot = hot.count[0];
  const bool null_hot = hot.count[1] != 0;
  for (int i = tid; i < kHotMapSlots; i += kDThreads) s_map_tag[i] = kTagFree;
  for (int i = tid; i <= kHotCap; i += kDThreads) { s_sum[i] = 0; s_cnt[i] = 0; s_first[i] = kNoRow; }
  __syncthreads();
  for (unsigned int h = tid; h < nhot; h += kDThreads) {  // tags are distinct: plain linear-probing insert
    const uint32_t tag = hot.tags[h];
    uint32_t slot = tag & (kHotMapSlots - 1);
    while (atomicCAS(&s_map_tag[slot], kTagFree, tag) != kTagFree) slot = (slot + 1) & (kHotMapSlots - 1);
    s_map_idx[slot] = static_cast<unsigned short>(h);
  }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  unsigned long long* alt = reinterpret_cast<unsigned long long*>(hot.alt_mask);
  // A wave takes 4 x 64 consecutive rows per step: the four mask words and (unconditional, clamped)
  // key loads are issued together; values are loaded only for hot rows.
  constexpr int kU = 4;
  const int wave = tid >> 6;
  const int64_t stride = int64_t(kDThreads / 64) * (64 * kU);
  uint64_t mw[kU], mw_next[kU];
  uint32_t kx[kU], kx_next[kU];
  int64_t base0 = lo + int64_t(wave) * (64 * kU);
  auto load = [&](int64_t b0, uint64_t (&m)[kU], uint32_t (&k)[kU]) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t row = b0 + 64 * u + lane;
      k[u] = static_cast<uint32_t>(FastKeyBits(a, row < hi ? row : hi - 1));
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) m[u] = FastMaskWord(a, b0 + 64 * u, hi);
  };
  if (base0 < hi) load(base0, mw, kx);
  for (; base0 < hi; base0 += stride) {
    // the next step's keys and mask words are requested before this step is folded
    const int64_t next = base0 + stride;
    load(next < hi ? next : base0, mw_next, kx_next);
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t row0 = base0 + 64 * u;
      if (row0 >= hi) break;  // wave-uniform
      const int64_t row = row0 + lane;
      bool keep = (mw[u] >> lane) & 1;
      if (keep) {
        const bool kn = a.k_valid && !GetBit(a.k_valid, a.k_off + row);
        int h = -1;
        if (kn) {
          if (null_hot) h = kHotCap;
        } else {
          const uint32_t tag = Fmix32(kx[u]);
          uint32_t slot = tag & (kHotMapSlots - 1);
          for (;;) {
            const uint32_t t = s_map_tag[slot];
            if (t == tag) { h = s_map_idx[slot]; break; }
            if (t == kTagFree) break;
            slot = (slot + 1) & (kHotMapSlots - 1);
          }
        }
        if (h >= 0) {
          keep = false;
          atomicMin(&s_first[h], static_cast<uint32_t>(row));
          if (a.has_values && (!a.v.valid || GetBit(a.v.valid, a.v.off + row))) {
            bool is_double;
            const uint64_t bits = LoadAccum(a.v, row, &is_double);
            if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[h]), __longlong_as_double(bits));
            else atomicAdd(&s_sum[h], static_cast<unsigned long long>(bits));
            atomicAdd(&s_cnt[h], 1u);
          }
        }
      }
      const uint64_t out = __ballot(keep);
      if (lane == 0) alt[row0 >> 6] = out;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) { mw[u] = mw_next[u]; kx[u] = kx_next[u]; }
  }
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * (kHotCap + 1);
  for (int i = tid; i <= kHotCap; i += kDThreads) {
    hot.part_sum[base + i] = s_sum[i];
    hot.part_cnt[base + i] = s_cnt[i];
    hot.part_first[base + i] = s_first[i];
  }
}

// Step 3 with the bin scatter's loader (32-bit keys without validity, 64-bit values, a mask without validity, all on
// 16-byte boundaries -- the shapes k_local_aggregate_vec takes): a thread owns four consecutive rows, three tiles in
// flight, the values arrive with the keys instead of one dependent load per hot row.  A key that carries a tenth of the
// rows would have a tenth of a wave's lanes add to ONE LDS word, served one lane after the other: the rows of the (up
// to three) most frequent hot keys of a wave step are reduced across the wave first -- one set of atomics per key and
// step -- and only what is left goes lane by lane.  k_hot_aggregate (one row per lane, generic loaders): 0.82 ms per
// 100 M rows with one hot key; this kernel streams at the scatter's rate.
template <bool MASK, bool VVALID>
__global__ __launch_bounds__(kBThreads) void k_hot_aggregate_vec(FastArgs a, int acc_double, HotMem hot, const unsigned int* use_hot) {
  if (!*use_hot) return;
  __shared__ uint32_t s_map_tag[kHotMapSlots];
  __shared__ unsigned short s_map_idx[kHotMapSlots];
  __shared__ unsigned long long s_sum[kHotCap + 1];
  __shared__ unsigned int s_cnt[kHotCap + 1];
  __shared__ uint32_t s_first[kHotCap + 1];
  constexpr int R = 4, kRows = kBThreads * R;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned int nhot = hot.count[0];
  for (int i = tid; i < kHotMapSlots; i += kBThreads) s_map_tag[i] = kTagFree;
  for (int i = tid; i <= kHotCap; i += kBThreads) { s_sum[i] = 0; s_cnt[i] = 0; s_first[i] = kNoRow; }
  __syncthreads();
  // the map holds the RAW keys (the hot set speaks fmix32 tags, a bijection: inverted once here instead of hashing four
  // keys per thread and tile), at slot RawSlot(key).  The one key that equals the free marker is left out: its rows
  // stay in the mask and reach their group through the partitions like any cold key's (InjectHot skips a hot key no
  // row was folded for).
  static_assert(kHotMapSlots == kSlots, "RawSlot");
  for (unsigned int h = tid; h < nhot; h += kBThreads) {  // keys are distinct: plain linear-probing insert
    const uint32_t key = Fmix32Inv(hot.tags[h]);
    if (key == kTagFree) continue;
    uint32_t slot = RawSlot(key);
    while (atomicCAS(&s_map_tag[slot], kTagFree, key) != kTagFree) slot = (slot + 1) & (kHotMapSlots - 1);
    s_map_idx[slot] = static_cast<unsigned short>(h);
  }
  __syncthreads();
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  const void* keys = static_cast<const char*>(a.k_data) + a.k_off * 4;
  const uint64_t* vals = reinterpret_cast<const uint64_t*>(a.v.data) + a.v.off;
  unsigned long long* alt = reinterpret_cast<unsigned long long*>(hot.alt_mask);
  auto fold = [&](const BinTile<R, false>& t, int64_t tile0) {
    const int64_t r0 = tile0 + int64_t(tid) * R;
    const int64_t left = hi - r0;
    const uint32_t live_bits = left >= R ? (1u << R) - 1u : (left > 0 ? (1u << left) - 1u : 0u);
    const int64_t rb = left > 0 ? r0 : hi - 1;
    const uint32_t mbits = MASK ? (FewBits<R>(t.m0, t.m1, a.m_off + rb) & live_bits) : live_bits;
    const uint32_t vbits = VVALID ? FewBits<R>(t.v0, t.v1, a.v.off + rb) : live_bits;
    int h[R];
    uint32_t tag[R], first_probe[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {  // (every row's first probe is on its way before any is looked at)
      tag[j] = t.key.k(j, 0);
      first_probe[j] = s_map_tag[RawSlot(tag[j])];
    }
    uint32_t keep = mbits;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      h[j] = -1;
      if (((mbits >> j) & 1u) && tag[j] != kTagFree) {
        uint32_t slot = RawSlot(tag[j]), seen = first_probe[j];
        for (;;) {
          if (seen == tag[j]) { h[j] = s_map_idx[slot]; break; }
          if (seen == kTagFree) break;
          slot = (slot + 1) & (kHotMapSlots - 1);
          seen = s_map_tag[slot];
        }
        if (h[j] >= 0) keep &= ~(1u << j);
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const bool vv = (vbits >> j) & 1u;
      const unsigned long long bits = t.v(j);
      // the wave's most frequent hot keys of this step, one after the other: their rows are folded across the wave
      for (int round = 0; round < 3; ++round) {
        const uint64_t pending = __ballot(h[j] >= 0);
        if (__popcll(pending) < 8) break;  // (wave-uniform) a handful of rows: lane by lane is cheaper than a reduction
        const int cur = __builtin_amdgcn_readlane(h[j], __builtin_ctzll(pending));
        const bool mine = h[j] == cur;
        const uint64_t m = __ballot(mine);
        if (__popcll(m) < 6) break;      // the step's hot rows are spread over many keys
        const uint64_t mv = __ballot(mine && vv);
        unsigned long long part = 0;
        if (acc_double) {
          double d = mine && vv ? __longlong_as_double(bits) : 0.0;
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) d += __longlong_as_double(__shfl_xor(static_cast<unsigned long long>(__double_as_longlong(d)), o, 64));
          part = static_cast<unsigned long long>(__double_as_longlong(d));
        } else {
          part = WaveReduceSum64(mine && vv ? bits : 0ull);
        }
        if (lane == __builtin_ctzll(m)) {  // the lowest lane of the key's rows holds its smallest row of this step
          atomicMin(&s_first[cur], static_cast<uint32_t>(r0 + j));
          if (mv) {
            if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[cur]), __longlong_as_double(part));
            else atomicAdd(&s_sum[cur], part);
            atomicAdd(&s_cnt[cur], static_cast<unsigned int>(__popcll(mv)));
          }
        }
        if (mine) h[j] = -1;
      }
      if (h[j] >= 0) {  // what is left: lane by lane
        atomicMin(&s_first[h[j]], static_cast<uint32_t>(r0 + j));
        if (vv) {
          if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[h[j]]), __longlong_as_double(bits));
          else atomicAdd(&s_sum[h[j]], bits);
          atomicAdd(&s_cnt[h[j]], 1u);
        }
      }
    }
    // the mask without the hot rows: 16 lanes' nibbles make a word
    unsigned long long w = static_cast<unsigned long long>(keep) << (4 * (lane & 15));
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) w |= __shfl_xor(w, o, 64);
    if ((lane & 15) == 0 && r0 < hi) alt[r0 >> 6] = w;
  };
  BinTile<R, false> t0{}, t1{}, t2{};
  int64_t tile0 = lo;
  if (lo + kRows <= hi) {
    const int64_t last_whole = lo + ((hi - lo) / kRows - 1) * kRows;
    auto at = [&](int64_t t) { return t <= last_whole ? t : last_whole; };  // (past the end: the last whole tile again, dropped)
    BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, lo, tid, t0);
    BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, at(lo + kRows), tid, t1);
    BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, at(lo + 2 * kRows), tid, t2);
#define AA_HOT_STEP(CUR, TILE0)                                                              \
    do {                                                                                       \
      if ((TILE0) <= last_whole) fold(CUR, (TILE0));                                           \
      BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, at((TILE0) + 3 * kRows), tid, CUR);   \
    } while (0)
    for (; tile0 <= last_whole; tile0 += 3 * kRows) {
      AA_HOT_STEP(t0, tile0);
      AA_HOT_STEP(t1, tile0 + kRows);
      AA_HOT_STEP(t2, tile0 + 2 * kRows);
    }
#undef AA_HOT_STEP
    tile0 = last_whole + kRows;
  }
  if (tile0 < hi) {  // the partial tile at the end of the input
    BinLoadTail<MASK, VVALID, R, false>(a, keys, vals, tile0, hi, tid, t0);
    fold(t0, tile0);
  }
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * (kHotCap + 1);
  for (int i = tid; i <= kHotCap; i += kBThreads) {
    hot.part_sum[base + i] = s_sum[i];
    hot.part_cnt[base + i] = s_cnt[i];
    hot.part_first[base + i] = s_first[i];
  }
}

// Step 5: hot group h -> a free slot of its partition's slice.  Called grid-stride from k_mark_first.
__device__ __forceinline__ void InjectHot(const HotMem& hot, int nwg, int k_bits, int acc_double, const GroupTable& t,
                                          unsigned long long* bitmap, int64_t gtid, int64_t gthreads) {
  const unsigned int nhot = hot.count[0];
  const bool null_hot = hot.count[1] != 0;
  // one WAVE per hot key: the lanes split the workgroups' partials, a butterfly folds them
  const int lane = static_cast<int>(gtid & 63);
  for (int64_t h = gtid >> 6; h <= kHotCap; h += gthreads >> 6) {
    if (h < kHotCap ? h >= nhot : !null_hot) continue;  // wave-uniform
    unsigned long long sum = 0;
    double dsum = 0.0;
    unsigned long long cnt = 0;
    uint32_t first = kNoRow;
    for (int w = lane; w < nwg; w += 64) {
      const int64_t i = int64_t(w) * (kHotCap + 1) + h;
      const uint32_t f = hot.part_first[i];
      if (f == kNoRow) continue;
      first = f < first ? f : first;
      cnt += hot.part_cnt[i];
      if (acc_double) dsum += __longlong_as_double(hot.part_sum[i]);
      else sum += hot.part_sum[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t of = __shfl_xor(first, d, 64);
      first = of < first ? of : first;
      cnt += __shfl_xor(cnt, d, 64);
      sum += __shfl_xor(sum, d, 64);
      dsum += __longlong_as_double(__shfl_xor(static_cast<unsigned long long>(__double_as_longlong(dsum)), d, 64));
    }
    if (lane != 0) continue;
    if (first == kNoRow) continue;  // no selected row carried this key
    const uint32_t tag = h < kHotCap ? hot.tags[h] : 0u;
    const uint64_t x = h < kHotCap ? uint64_t(Fmix32Inv(tag)) : (uint64_t(1) << k_bits);
    const uint32_t p = h < kHotCap ? (tag >> (32 - kPBits)) : 0u;  // the null key lives in partition 0
    const int64_t g0 = int64_t(p) * kSlots;
    const unsigned long long hk = Fmix64(x);
    for (int probes = 0; probes < kSlots; ++probes) {
      const int64_t s = g0 + ((tag + probes) & (kSlots - 1));
      if (atomicCAS(&t.hkeys[s], static_cast<unsigned long long>(kEmpty), hk) == kEmpty) {
        t.first[s] = first;
        t.sums[s] = acc_double ? static_cast<unsigned long long>(__double_as_longlong(dsum)) : sum;
        t.counts[s] = cnt;
        atomicOr(&bitmap[first >> 6], 1ull << (first & 63));
        break;
      }
    }
  }
}

__device__ void InjectHotGroups(const HotMem& hot, int nwg, int k_bits, int acc_double, const GroupTable& t,
                                unsigned long long* bitmap, int64_t gtid, int64_t gthreads) {
  InjectHot(hot, nwg, k_bits, acc_double, t, bitmap, gtid, gthreads);
}

// Fold the 7 x 64 rows a wave holds of one fast-layout tile (see FastLoadTile).
template <bool VVALID>
__device__ __forceinline__ bool LocalFoldFast(LocalShared& sh, int64_t tile0, int wave, int lane, int acc_double,
                                              const uint32_t (&kx)[kFSteps], const uint64_t (&kv)[kFSteps], uint64_t mword,
                                              uint64_t vword) {
  const int64_t wrow0 = tile0 + wave * (kFSteps * 64);
  bool ok = true;
#pragma unroll
  for (int g = 0; g < kFSteps; g += kAggUnroll) {
    uint4 rec[kAggUnroll];
    bool live[kAggUnroll];
#pragma unroll
    for (int u = 0; u < kAggUnroll; ++u) {
      const int j = g + u;
      live[u] = false;
      rec[u] = make_uint4(0, 0, 0, 0);
      if (j < kFSteps) {
        live[u] = (__shfl(mword, j, 64) >> lane) & 1;
        const bool valid = VVALID ? ((__shfl(vword, j, 64) >> lane) & 1) != 0 : true;
        const uint32_t row = static_cast<uint32_t>(wrow0 + j * 64 + lane);
        rec[u] = make_uint4(kx[j], row | (valid ? kRecValValid : 0u), static_cast<uint32_t>(kv[j]),
                            static_cast<uint32_t>(kv[j] >> 32));
      }
    }
    ok &= AggFoldRaw(rec, live, acc_double, sh.tag, sh.sum, sh.first, sh.cnt, &sh.groups, kLocalLimit);
  }
  return ok;
}

template <bool MASK, bool VVALID>
__global__ __launch_bounds__(kDThreads) void k_local_aggregate_fast(FastArgs a, int acc_double, LocalStage st,
                                                                    unsigned int* wg_flags, HotMem hot) {
  __shared__ LocalShared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  LocalInit(sh, tid);
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  const uint32_t* keys = reinterpret_cast<const uint32_t*>(a.k_data) + a.k_off;
  const uint64_t* vals = reinterpret_cast<const uint64_t*>(a.v.data) + a.v.off;
  uint32_t kxA[kFSteps], kxB[kFSteps];
  uint64_t kvA[kFSteps], kvB[kFSteps];
  RawBits mA{}, vA{}, mB{}, vB{};
  auto words = [&](int64_t t0, const RawBits& mraw, const RawBits& vraw, uint64_t* mword, uint64_t* vword) {
    const int64_t mrow = t0 + wave * (kFSteps * 64) + 64 * (lane & 7);
    const int64_t left = hi - mrow;
    const int nbits = left >= 64 ? 64 : (left > 0 ? static_cast<int>(left) : 0);
    const int64_t mrow_c = nbits > 0 ? mrow : hi - 1;
    const uint64_t live = nbits == 0 ? 0 : (nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0));
    *mword = MASK ? (CombineBits(a.m_data, a.m_off + mrow_c, nbits > 0 ? nbits : 1, mraw) & live) : live;
    *vword = VVALID ? (CombineBits(a.v.valid, a.v.off + mrow_c, nbits > 0 ? nbits : 1, vraw) & live) : live;
  };
  uint64_t mw, vw;
  // (a relaxed atomic load, not a volatile one: a volatile access drains every counter, i.e. the prefetched tile)
  auto stopped = [&]() { return __hip_atomic_load(&sh.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };
  if (lo < hi) FastLoadTile<MASK, VVALID>(a, keys, vals, lo, hi, wave, lane, kxA, kvA, mA, vA);
  for (int64_t tile0 = lo; tile0 < hi; tile0 += 2 * kLRows) {
    const bool has_b = tile0 + kLRows < hi;
    if (has_b) FastLoadTile<MASK, VVALID>(a, keys, vals, tile0 + kLRows, hi, wave, lane, kxB, kvB, mB, vB);
    if (stopped()) break;
    words(tile0, mA, vA, &mw, &vw);
    if (__any(!LocalFoldFast<VVALID>(sh, tile0, wave, lane, acc_double, kxA, kvA, mw, vw))) { LocalGiveUp(sh, wg_flags); break; }
    if (!has_b) break;
    if (tile0 + 2 * kLRows < hi) FastLoadTile<MASK, VVALID>(a, keys, vals, tile0 + 2 * kLRows, hi, wave, lane, kxA, kvA, mA, vA);
    if (stopped()) break;
    words(tile0 + kLRows, mB, vB, &mw, &vw);
    if (__any(!LocalFoldFast<VVALID>(sh, tile0 + kLRows, wave, lane, acc_double, kxB, kvB, mw, vw))) { LocalGiveUp(sh, wg_flags); break; }
  }
  __syncthreads();
  if (stopped()) {  // a stale staging table is never read: the merge is gated on the flags
    ReportHotCandidates(sh, hot, tid, /*raw_tags=*/true);
    return;
  }
  LocalWriteOut(sh, st, tid, /*raw_tags=*/true);
}

// The same attempt with the bin scatter's loader: a thread owns four consecutive rows (one 16-byte key load,
// two 16-byte value loads, a nibble of the mask), three tiles in flight, nothing but LDS work between the
// loads (k_local_aggregate_fast reads one element per lane and step: 4.9 TB/s; the row range of a workgroup
// and its staged table are the same).  Needs keys and values on a 16-byte boundary, like the bin scatter.
template <bool MASK, bool VVALID>
__global__ __launch_bounds__(kBThreads) void k_local_aggregate_vec(FastArgs a, int acc_double, LocalStage st,
                                                                  unsigned int* wg_flags, HotMem hot) {
  __shared__ LocalShared sh;
  constexpr int R = 4, kRows = kBThreads * R;
  const int tid = threadIdx.x;
  LocalInit(sh, tid);
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  const void* keys = static_cast<const char*>(a.k_data) + a.k_off * 4;
  const uint64_t* vals = reinterpret_cast<const uint64_t*>(a.v.data) + a.v.off;
  auto stopped = [&]() { return __hip_atomic_load(&sh.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };
  uint32_t vwide = 0;
  auto fold = [&](const BinTile<R, false>& t, int64_t tile0) -> bool {
    const int64_t r0 = tile0 + int64_t(tid) * R;
    const int64_t left = hi - r0;
    const uint32_t live_bits = left >= R ? (1u << R) - 1u : (left > 0 ? (1u << left) - 1u : 0u);
    const int64_t rb = left > 0 ? r0 : hi - 1;
    const uint32_t mbits = MASK ? (FewBits<R>(t.m0, t.m1, a.m_off + rb) & live_bits) : live_bits;
    const uint32_t vbits = VVALID ? FewBits<R>(t.v0, t.v1, a.v.off + rb) : live_bits;
    uint4 rec[R];
    bool live[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      live[j] = (mbits >> j) & 1u;
      const bool vv = (vbits >> j) & 1u;
      const uint64_t v = t.v(j);
      if (live[j] && vv) vwide |= static_cast<uint32_t>((v + 0x80000000ull) >> 32);  // (a selected value that is not an int32)
      rec[j] = make_uint4(t.key.k(j, 0), static_cast<uint32_t>(r0 + j) | (vv ? kRecValValid : 0u), static_cast<uint32_t>(v),
                          static_cast<uint32_t>(v >> 32));
    }
    return AggFoldRaw(rec, live, acc_double, sh.tag, sh.sum, sh.first, sh.cnt, &sh.groups, kLocalLimit);
  };
  BinTile<R, false> t0{}, t1{}, t2{};
  int64_t tile0 = lo;
  bool gave_up = false;
  if (lo + kRows <= hi) {
    const int64_t last_whole = lo + ((hi - lo) / kRows - 1) * kRows;
    auto at = [&](int64_t t) { return t <= last_whole ? t : last_whole; };  // (past the end: the last whole tile again, dropped)
    BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, lo, tid, t0);
    BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, at(lo + kRows), tid, t1);
    BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, at(lo + 2 * kRows), tid, t2);
#define AA_LOCAL_STEP(CUR, TILE0)                                                            \
    do {                                                                                       \
      if ((TILE0) <= last_whole && !gave_up && __any(!fold(CUR, (TILE0)))) gave_up = true;     \
      BinLoadTile<MASK, VVALID, R, false>(a, keys, vals, at((TILE0) + 3 * kRows), tid, CUR);   \
    } while (0)
    for (; tile0 <= last_whole; tile0 += 3 * kRows) {
      AA_LOCAL_STEP(t0, tile0);
      AA_LOCAL_STEP(t1, tile0 + kRows);
      AA_LOCAL_STEP(t2, tile0 + 2 * kRows);
      if (gave_up) LocalGiveUp(sh, wg_flags);
      if (gave_up || stopped()) break;
    }
#undef AA_LOCAL_STEP
    if (tile0 > last_whole) tile0 = last_whole + kRows;
  }
  if (tile0 < hi && !gave_up && !stopped()) {  // the partial tile at the end of the input
    BinLoadTail<MASK, VVALID, R, false>(a, keys, vals, tile0, hi, tid, t0);
    if (__any(!fold(t0, tile0))) LocalGiveUp(sh, wg_flags);
  }
  __syncthreads();
  {
    // What this row range looked like, for a host that plans the partition path from it (FilterGroupByImpl, "guessed"
    // compact records): the key range of the table -- up to kLocalLimit distinct keys from the range's first rows when
    // the attempt gave up, all of its keys otherwise -- in the scatter's key order, and whether a selected value was
    // not an int32.  256 workgroups spread over the input: half a million keys sampled for nothing.
    __shared__ unsigned int s_kmin, s_kmax;
    if (tid == 0) { s_kmin = 0xFFFFFFFFu; s_kmax = 0; }
    const unsigned int any_wide = __syncthreads_or(vwide != 0);
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
    for (int s = tid; s < kSlots; s += kBThreads) {
      const unsigned int tag = sh.tag[s];
      if (tag == kTagFree) continue;
      const uint32_t kk = tag ^ a.k_flip;
      kmin = kk < kmin ? kk : kmin;
      kmax = kk > kmax ? kk : kmax;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t omin = __shfl_xor(kmin, d, 64), omax = __shfl_xor(kmax, d, 64);
      kmin = omin < kmin ? omin : kmin;
      kmax = omax > kmax ? omax : kmax;
    }
    if ((tid & 63) == 0 && kmin <= kmax) { atomicMin(&s_kmin, kmin); atomicMax(&s_kmax, kmax); }
    __syncthreads();
    if (tid == 0 && hot.wg_stats) {
      uint32_t* out = hot.wg_stats + 4 * blockIdx.x;
      out[0] = s_kmin; out[1] = s_kmax; out[2] = any_wide ? 1u : 0u; out[3] = 0;
    }
  }
  if (stopped()) {  // a stale staging table is never read: the merge is gated on the flags
    ReportHotCandidates(sh, hot, tid, /*raw_tags=*/true);
    return;
  }
  LocalWriteOut(sh, st, tid, /*raw_tags=*/true);
}

// Any key width <= 32 bits, key / value / mask validity, every value type (LoadTile).
__global__ __launch_bounds__(kDThreads) void k_local_aggregate(FastArgs a, int acc_double, LocalStage st,
                                                               unsigned int* wg_flags, HotMem hot) {
  __shared__ LocalShared sh;
  const int tid = threadIdx.x, lane = tid & 63;
  LocalInit(sh, tid);
  const int64_t lo = int64_t(blockIdx.x) * a.rows_per_block;
  const int64_t hi = lo + a.rows_per_block < a.n ? lo + a.rows_per_block : a.n;
  constexpr int kIters = kDRows / kDThreads;
  static_assert(kIters % kAggUnroll == 0, "tile rows per thread");
  // (a relaxed atomic load, not a volatile one: a volatile access drains every counter, i.e. the prefetched tile)
  auto stopped = [&]() { return __hip_atomic_load(&sh.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0; };
  TileRegs t;
  if (lo < hi) LoadTile(a, lo, hi, tid, lane, &t);
  for (int64_t tile0 = lo; tile0 < hi; tile0 += kDRows) {
    const TileRegs c = t;
    if (tile0 + kDRows < hi) LoadTile(a, tile0 + kDRows, hi, tid, lane, &t);
    if (stopped()) break;
    bool ok = true;
#pragma unroll
    for (int g = 0; g < kIters; g += kAggUnroll) {
      uint4 rec[kAggUnroll];
      bool live[kAggUnroll];
#pragma unroll
      for (int u = 0; u < kAggUnroll; ++u) {
        live[u] = c.sel[g + u];
        rec[u] = make_uint4(c.x[g + u], c.flags[g + u], static_cast<uint32_t>(c.val[g + u]),
                            static_cast<uint32_t>(c.val[g + u] >> 32));
      }
      ok &= AggFold(rec, live, acc_double, sh.tag, sh.sum, sh.first, sh.cnt, &sh.groups, kLocalLimit);
    }
    if (__any(!ok)) { LocalGiveUp(sh, wg_flags); break; }
  }
  __syncthreads();
  if (stopped()) {
    ReportHotCandidates(sh, hot, tid);
    return;
  }
  LocalWriteOut(sh, st, tid);
}

// Zero the first-row bitmap, the status words and the per-workgroup flags in one launch.
__global__ __launch_bounds__(kThreads) void k_prepare(unsigned long long* bitmap, int64_t nwords, unsigned int* status,
                                                      unsigned int* wg_flags, int nflags, unsigned int* hot_cand_n,
                                                      uint32_t* spill_counts /* counts[p][stride - 1], or null */, int spill_stride,
                                                      unsigned int* key_minmax /* bin scatter, or null */) {
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) bitmap[w] = 0;
  if (blockIdx.x == 1 % gridDim.x && spill_counts)
    for (int p = threadIdx.x; p < kP; p += kThreads) spill_counts[int64_t(p) * spill_stride + spill_stride - 1] = 0;
  if (blockIdx.x == 0) {
    if (threadIdx.x < 8) status[threadIdx.x] = threadIdx.x == 4 ? kNoRow : 0u;
    if (threadIdx.x >= 8 && threadIdx.x < 12 && !(key_minmax && threadIdx.x < 10)) status[threadIdx.x] = 0u;
    if (key_minmax && threadIdx.x < 2) key_minmax[threadIdx.x] = threadIdx.x == 0 ? 0xFFFFFFFFu : 0u;
    for (int i = threadIdx.x; i < nflags; i += kThreads) { wg_flags[i] = 0; if (hot_cand_n) hot_cand_n[i] = 0; }
  }
}

// OR the workgroup flags into *summary (status[1]); reset the global table -- only when the
// local path went through.
// `gave_up_skip` (a call that queued ONLY the pre-aggregation, see local_only): nothing behind this kernel builds the
// table when the attempt gives up, so the word every downstream kernel is gated on (status[0]) is raised as well --
// mark / emit / unpack then touch neither the uninitialised table nor the heavy hitters' scratch -- and the host
// starts the call over.
constexpr unsigned int kLocalGaveUp = 8;
// `probe` (optional, mapped host memory): {1, gave up, heavy hitters, smallest key, largest key, a wide value} for a host
// that waits for this kernel before it queues the rest (FilterGroupByImpl's guessed compact records).
__global__ __launch_bounds__(kThreads) void k_clear_table_unless(GroupTable t, int64_t nslots, const unsigned int* wg_flags,
                                                                 int nflags, unsigned int* summary, HotMem hot,
                                                                 unsigned int* use_hot, unsigned int* gave_up_skip,
                                                                 unsigned int* probe) {
  int mine = 0;
  for (int i = threadIdx.x; i < nflags; i += kThreads) mine |= wg_flags[i] != 0;
  const int any = __syncthreads_or(mine);
  if (blockIdx.x == 0 && threadIdx.x == 0) *summary = any ? 1u : 0u;
  __shared__ unsigned int s_pmin, s_pmax, s_pwide;
  if (probe && blockIdx.x == 0) {  // (uniform over the block)
    if (threadIdx.x == 0) { s_pmin = 0xFFFFFFFFu; s_pmax = 0; s_pwide = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < nflags; i += kThreads) {
      const uint32_t lo = hot.wg_stats[4 * i], hi = hot.wg_stats[4 * i + 1];
      if (lo <= hi) { atomicMin(&s_pmin, lo); atomicMax(&s_pmax, hi); }
      if (hot.wg_stats[4 * i + 2]) s_pwide = 1;
    }
    __syncthreads();
  }
  if (any) {
    if (gave_up_skip) {
      if (blockIdx.x == 0 && threadIdx.x == 0) *gave_up_skip = kLocalGaveUp;
      return;
    }
    if (blockIdx.x == 0) BuildHotSet(hot, nflags, use_hot);  // heavy hitters seen by the workgroups that gave up
  }
  if (probe && blockIdx.x == 0) {
    __syncthreads();
    if (threadIdx.x == 0) {
      probe[1] = any ? 1u : 0u; probe[2] = any ? *use_hot : 0u; probe[3] = s_pmin; probe[4] = s_pmax; probe[5] = s_pwide;
      __threadfence_system();
      probe[0] = 1u;
    }
  }
  if (any) return;
  for (int64_t s = int64_t(blockIdx.x) * kThreads + threadIdx.x; s < nslots; s += int64_t(gridDim.x) * kThreads) {
    t.hkeys[s] = kEmpty; t.first[s] = kNoRow; t.sums[s] = 0; t.counts[s] = 0;
  }
}

// Merge the staged workgroup tables into the global table (global atomics, few entries).
// The merge for the raw-key local tables (k_local_aggregate_fast): every workgroup's table puts a key at the
// same home slot (probing only displaces it when keys collide), so the entries of one slot across all staged
// tables are almost always ONE group.  A workgroup here takes kMergeSlots slots, folds each slot's entries over
// the staged tables in LDS (the first tag met owns the slot; an entry with another tag goes to the global table on
// its own) and inserts one reduced group per slot: G global-table updates instead of G x 256 (64 -> ~10 us at
// 1000 groups).
constexpr int kMergeSlots = 64;
constexpr int kMergeThreads = 1024;
__global__ __launch_bounds__(kMergeThreads) void k_local_merge_slots(LocalStage st, int nwg, int k_bits, int acc_double,
                                                                     GroupTable t, const unsigned int* skip_if,
                                                                     unsigned int* overflow) {
  if (*skip_if) return;
  __shared__ unsigned long long s_owner[kMergeSlots];  // 0 = nobody yet, else (1 << 32) | tag of the slot's group
  __shared__ uint32_t s_first[kMergeSlots];
  __shared__ unsigned long long s_sum[kMergeSlots], s_cnt[kMergeSlots];
  const int tid = threadIdx.x;
  const int sl = tid & (kMergeSlots - 1);
  const int slot = blockIdx.x * kMergeSlots + sl;  // (slots kSlots, kSlots + 1: the two side slots, never filled by the raw fold)
  if (tid < kMergeSlots) { s_owner[tid] = 0; s_first[tid] = kNoRow; s_sum[tid] = 0; s_cnt[tid] = 0; }
  __syncthreads();
  auto to_global = [&](unsigned int tag, uint32_t fr, unsigned long long sum, unsigned long long cnt) {
    const int64_t idx = FindOrInsert(t, Fmix64(uint64_t(Fmix32Inv(tag))), true);
    if (idx < 0) { *overflow = 1; return; }
    atomicMin(&t.first[idx], fr);
    if (cnt) {
      Accumulate(&t.sums[idx], sum, acc_double != 0);
      atomicAdd(&t.counts[idx], cnt);
    }
  };
  if (slot < kSlots) {
    for (int w = tid / kMergeSlots; w < nwg; w += kMergeThreads / kMergeSlots) {
      const int64_t i = int64_t(w) * kStageSlots + slot;
      const uint32_t fr = st.first[i];
      if (fr == kNoRow) continue;
      const unsigned int tag = st.tag[i];
      // the first entry to arrive owns the slot: one 64-bit compare-and-swap publishes (taken | tag), nobody waits
      const unsigned long long mine = (1ull << 32) | tag;
      const unsigned long long prev = atomicCAS(&s_owner[sl], 0ull, mine);
      const unsigned int owner_tag = prev == 0ull ? tag : static_cast<unsigned int>(prev);
      if (tag == owner_tag) {
        atomicMin(&s_first[sl], fr);
        const unsigned int c = st.cnt[i];
        if (c) {
          if (acc_double) atomicAdd(reinterpret_cast<double*>(&s_sum[sl]), __longlong_as_double(st.sum[i]));
          else atomicAdd(&s_sum[sl], st.sum[i]);
          atomicAdd(&s_cnt[sl], static_cast<unsigned long long>(c));
        }
      } else {
        to_global(tag, fr, st.sum[i], st.cnt[i]);
      }
    }
  }
  __syncthreads();
  if (tid < kMergeSlots && s_first[tid] != kNoRow) to_global(static_cast<unsigned int>(s_owner[tid]), s_first[tid], s_sum[tid], s_cnt[tid]);
}

__global__ __launch_bounds__(kThreads) void k_local_merge(LocalStage st, int64_t nentries, int k_bits, int acc_double,
                                                          GroupTable t, const unsigned int* skip_if, unsigned int* overflow) {
  if (*skip_if) return;
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < nentries; i += int64_t(gridDim.x) * kThreads) {
    const uint32_t fr = st.first[i];
    if (fr == kNoRow) continue;  // free slot
    const int s = static_cast<int>(i % kStageSlots);
    const uint64_t x = s == kSideNull ? (uint64_t(1) << k_bits)
                     : s == kSideFree ? uint64_t(Fmix32Inv(kTagFree)) : uint64_t(Fmix32Inv(st.tag[i]));
    const int64_t idx = FindOrInsert(t, Fmix64(x), true);
    if (idx < 0) { *overflow = 1; continue; }
    atomicMin(&t.first[idx], fr);
    const unsigned int c = st.cnt[i];
    if (c) {
      Accumulate(&t.sums[idx], st.sum[i], acc_double != 0);
      atomicAdd(&t.counts[idx], static_cast<unsigned long long>(c));
    }
  }
}

}  // namespace amdarrow

using namespace amdarrow;

// ---- Grouper ------------------------------------------------------------------------------------------
constexpr int kMaxWideKeys = 16;
struct WideGrouper;
struct amdarrow_grouper {
  amdarrow_ctx* ctx;
  int nkeys;
  int types[kMaxWideKeys];
  WideGrouper* wide = nullptr;  // key rows that do not fit one 64-bit word: see WideGrouper
  KeySpec spec;              // shifts/bits only (data pointers are per batch)
  uint64_t T = 0;            // table slots
  char* table_mem = nullptr; // hkeys | first | sums(unused) | counts(unused) | ids
  TableMem tm;
  unsigned long long* gx = nullptr;  // key word per group id
  uint64_t gx_cap = 0;
  uint32_t num_groups = 0;
  uint32_t* wide_null_id = nullptr;  // device word
};

extern "C" {
int amdarrow_grouper_create(amdarrow_ctx* ctx, int nkeys, const int* key_type_ids, amdarrow_grouper** out);
int amdarrow_grouper_destroy(amdarrow_grouper* g);
int amdarrow_grouper_consume(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids);
int amdarrow_grouper_lookup(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids);
int amdarrow_grouper_uniques(amdarrow_grouper* g, int k, void* out_values, uint8_t* out_validity, amdarrow_out* out);
}
static int GrouperAllocTable(amdarrow_grouper* g, uint64_t T, char** mem, TableMem* tm) {
  Carver c;
  CarveTable(&c, T, true, tm);
  AA_TRY(DeviceMalloc(g->ctx, reinterpret_cast<void**>(mem), c.off));
  return ClearTable(g->ctx, BindTable(*mem, *tm, T, true), true);
}

// ---- key rows wider than one 64-bit word (KeyEncoder rows of any width, hash_aggregate.cc:48-311, 313-437) ----
// The packed-word table above holds keys of up to 64 - nkeys value bits.  Wider rows are grouped by composition,
// every step exact: each key column gets its own single-column grouper (values -> dense ids), and the row's ids are
// folded left to right, pairs[j] grouping the 64-bit word (ids so far | ids of column j + 1 << 32).  Equal rows
// have equal id tuples and vice versa, every grouper numbers in first-appearance order and keeps its state across
// Consume calls, so the last pair grouper's ids ARE the reference's group ids.  GetUniques walks the pair
// dictionaries back down to the column's ids and gathers the column grouper's uniques through them.
struct WideGrouper {
  std::vector<amdarrow_grouper*> cols, pairs;
  uint32_t* ids = nullptr;             // (ncols + 1) x cap row ids
  unsigned long long* packed = nullptr;  // cap pair words
  int64_t cap = 0;
};
__global__ __launch_bounds__(kThreads) void k_pack_id_pair(const uint32_t* lo, const uint32_t* hi, int64_t n, unsigned long long* out) {
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads)
    out[i] = static_cast<unsigned long long>(lo[i]) | (static_cast<unsigned long long>(hi[i]) << 32);
}
// out[i] = one half of pair word of group (lst ? lst[i] : i)
__global__ __launch_bounds__(kThreads) void k_pair_half(const unsigned long long* pair_words, const uint32_t* lst, int64_t n, int high,
                                                        uint32_t* out) {
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads) {
    const unsigned long long x = pair_words[lst ? lst[i] : static_cast<uint32_t>(i)];
    out[i] = high ? static_cast<uint32_t>(x >> 32) : static_cast<uint32_t>(x);
  }
}
static void WideDestroy(amdarrow_grouper* g) {
  if (!g->wide) return;
  for (auto* c : g->wide->cols) amdarrow_grouper_destroy(c);
  for (auto* c : g->wide->pairs) amdarrow_grouper_destroy(c);
  if (g->wide->ids) (void)hipFree(g->wide->ids);
  if (g->wide->packed) (void)hipFree(g->wide->packed);
  delete g->wide;
  g->wide = nullptr;
}
static int WideCreate(amdarrow_ctx* ctx, int nkeys, const int* types, amdarrow_grouper** out) {
  auto* g = new amdarrow_grouper;
  g->ctx = ctx;
  g->nkeys = nkeys;
  for (int k = 0; k < nkeys; ++k) g->types[k] = types[k];
  g->wide = new WideGrouper;
  int rc = AMDARROW_OK;
  for (int k = 0; k < nkeys && rc == AMDARROW_OK; ++k) {
    amdarrow_grouper* c = nullptr;
    rc = amdarrow_grouper_create(ctx, 1, &types[k], &c);
    if (rc == AMDARROW_OK) g->wide->cols.push_back(c);
  }
  const int pair_type = AMDARROW_UINT64;
  for (int j = 0; j + 1 < nkeys && rc == AMDARROW_OK; ++j) {
    amdarrow_grouper* c = nullptr;
    rc = amdarrow_grouper_create(ctx, 1, &pair_type, &c);
    if (rc == AMDARROW_OK) g->wide->pairs.push_back(c);
  }
  if (rc != AMDARROW_OK) { WideDestroy(g); delete g; return rc; }
  *out = g;
  return AMDARROW_OK;
}
static int WideIds(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids, bool insert) {
  amdarrow_ctx* ctx = g->ctx;
  WideGrouper* w = g->wide;
  const int K = g->nkeys;
  const int64_t n = keys[0].length;
  for (int k = 0; k < K; ++k) {
    if (keys[k].type_id != g->types[k]) return SetError(AMDARROW_TYPE_ERROR, "grouper: key type differs from Make()");
    if (keys[k].length != n) return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");
  }
  if (n == 0) return AMDARROW_OK;
  if (n > w->cap) {
    AA_HIP(hipStreamSynchronize(ctx->stream));
    if (w->ids) (void)hipFree(w->ids);
    if (w->packed) (void)hipFree(w->packed);
    w->ids = nullptr; w->packed = nullptr; w->cap = 0;
    AA_TRY(DeviceMalloc(g->ctx, reinterpret_cast<void**>(&w->ids), 4 * static_cast<size_t>(K + 1) * static_cast<size_t>(n)));
    AA_TRY(DeviceMalloc(g->ctx, reinterpret_cast<void**>(&w->packed), 8 * static_cast<size_t>(n)));
    w->cap = n;
  }
  for (int k = 0; k < K; ++k)
    AA_TRY(insert ? amdarrow_grouper_consume(w->cols[k], &keys[k], w->ids + k * w->cap)
                  : amdarrow_grouper_lookup(w->cols[k], &keys[k], w->ids + k * w->cap));
  const uint32_t* cur = w->ids;
  for (int j = 0; j + 1 < K; ++j) {
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      hipLaunchKernelGGL(k_pack_id_pair, dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, cur,
                         w->ids + (j + 1) * w->cap, n, w->packed);
      AA_HIP(hipGetLastError());
    }
    amdarrow_array pa;
    pa.values = w->packed; pa.validity = nullptr; pa.length = n; pa.offset = 0; pa.type_id = AMDARROW_UINT64; pa.null_count = 0;
    uint32_t* dst = j + 2 == K ? out_ids : w->ids + K * w->cap;
    AA_TRY(insert ? amdarrow_grouper_consume(w->pairs[j], &pa, dst) : amdarrow_grouper_lookup(w->pairs[j], &pa, dst));
    cur = dst;
  }
  g->num_groups = w->pairs.back()->num_groups;
  return AMDARROW_OK;
}
static int WideUniques(amdarrow_grouper* g, int k, void* out_values, uint8_t* out_validity, amdarrow_out* out) {
  amdarrow_ctx* ctx = g->ctx;
  WideGrouper* w = g->wide;
  const int K = g->nkeys;
  const int64_t G = g->num_groups;
  out->length = G; out->null_count = 0; out->has_validity = 0;
  if (G == 0) return AMDARROW_OK;
  amdarrow_grouper* child = w->cols[k];
  const int64_t Gc = child->num_groups;
  char* mem = nullptr;
  const size_t words = 8 * (static_cast<size_t>(G) + 8), bits = (static_cast<size_t>(std::max(G, Gc)) / 64 + 2) * 8;
  const size_t lst_bytes = 8 * ((static_cast<size_t>(G) + 9) / 2), child_bytes = 8 * (static_cast<size_t>(Gc) + 8);
  AA_TRY(DeviceMalloc(g->ctx, reinterpret_cast<void**>(&mem), words + bits + 2 * lst_bytes + child_bytes + bits));
  auto* pair_words = reinterpret_cast<unsigned long long*>(mem);
  auto* scratch_bits = reinterpret_cast<uint8_t*>(mem + words);
  uint32_t* lst_buf[2] = {reinterpret_cast<uint32_t*>(mem + words + bits), reinterpret_cast<uint32_t*>(mem + words + bits + lst_bytes)};
  char* child_values = mem + words + bits + 2 * lst_bytes;
  auto* child_valid = reinterpret_cast<uint8_t*>(child_values + child_bytes);
  int rc = AMDARROW_OK;
  const uint32_t* lst = nullptr;  // identity over the G groups
  uint32_t* col_ids = nullptr;
  for (int j = K - 2; j >= 0 && rc == AMDARROW_OK && !col_ids; --j) {
    amdarrow_out po;
    rc = amdarrow_grouper_uniques(w->pairs[j], 0, pair_words, scratch_bits, &po);
    if (rc != AMDARROW_OK) break;
    const int high = k == j + 1;
    uint32_t* dst = lst_buf[j & 1];
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      hipLaunchKernelGGL(k_pair_half, dim3(GridFor(G, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, pair_words, lst, G,
                         high, dst);
      if (hipGetLastError() != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, "k_pair_half launch");
    }
    if (high || j == 0) col_ids = dst;  // j == 0 and not high: the low half is column 0's ids
    lst = dst;
  }
  if (rc == AMDARROW_OK) {
    amdarrow_out co;
    rc = amdarrow_grouper_uniques(child, 0, child_values, child_valid, &co);
    if (rc == AMDARROW_OK) {
      amdarrow_array va, ia;
      va.values = child_values; va.validity = co.has_validity ? child_valid : nullptr; va.length = Gc; va.offset = 0;
      va.type_id = g->types[k]; va.null_count = co.null_count;
      ia.values = col_ids; ia.validity = nullptr; ia.length = G; ia.offset = 0; ia.type_id = AMDARROW_UINT32; ia.null_count = 0;
      amdarrow_out to;
      rc = amdarrow_take(ctx, &va, &ia, /*boundscheck=*/0, out_values, out_validity, &to);
      if (rc == AMDARROW_OK) { out->null_count = to.null_count; out->has_validity = to.null_count > 0; }
    }
  }
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipFree(mem);
  return rc;
}

extern "C" {

int amdarrow_grouper_create(amdarrow_ctx* ctx, int nkeys, const int* key_type_ids, amdarrow_grouper** out) {
  KeySpec ks;
  {
    const int rc = MakeKeySpec(nkeys, nullptr, key_type_ids, &ks);
    if (rc != AMDARROW_OK) {
      bool fixed_width = nkeys >= 2 && nkeys <= kMaxWideKeys;
      for (int k = 0; fixed_width && k < nkeys; ++k) fixed_width = BitWidth(key_type_ids[k]) >= 0;
      return fixed_width ? WideCreate(ctx, nkeys, key_type_ids, out) : rc;
    }
  }
  auto* g = new amdarrow_grouper;
  g->ctx = ctx;
  g->nkeys = nkeys;
  for (int k = 0; k < nkeys; ++k) g->types[k] = key_type_ids[k];
  g->spec = ks;
  std::lock_guard<std::mutex> lock(ctx->mu);
  g->T = 4096;
  int rc = GrouperAllocTable(g, g->T, &g->table_mem, &g->tm);
  if (rc != AMDARROW_OK) { delete g; return rc; }
  rc = DeviceMalloc(ctx, reinterpret_cast<void**>(&g->wide_null_id), 256);
  if (rc != AMDARROW_OK) { delete g; return rc; }
  AA_HIP(hipMemsetAsync(g->wide_null_id, 0xFF, 256, ctx->stream));
  *out = g;
  return AMDARROW_OK;
}

int amdarrow_grouper_destroy(amdarrow_grouper* g) {
  if (!g) return AMDARROW_OK;
  (void)hipStreamSynchronize(g->ctx->stream);
  WideDestroy(g);
  if (g->table_mem) (void)hipFree(g->table_mem);
  if (g->gx) (void)hipFree(g->gx);
  if (g->wide_null_id) (void)hipFree(g->wide_null_id);
  delete g;
  return AMDARROW_OK;
}

int amdarrow_grouper_num_groups(amdarrow_grouper* g, uint32_t* out) {
  *out = g->num_groups;
  return AMDARROW_OK;
}

// Which entry point is asking (part of the hint's shape, see GroupByHint::caller); set under ctx->mu.
struct CallerScope {
  amdarrow_ctx* ctx; int saved;
  CallerScope(amdarrow_ctx* c, int id) : ctx(c), saved(c->gb_caller) { c->gb_caller = id; }
  ~CallerScope() { ctx->gb_caller = saved; }
};
enum { kCallerFilterGroupBy = 0, kCallerPackedKeys = 1, kCallerHashAccumulate = 2, kCallerGrouper = 3 };
static int FilterGroupByImpl(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                             const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                             void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                             int64_t* out_num_groups, bool locked, ulonglong4** packed_out, uint64_t* bias_out = nullptr);

__global__ __launch_bounds__(kThreads) void k_gx_from_packed(const ulonglong4* packed, uint32_t G, unsigned long long bias, unsigned long long* gx) {
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < G; i += gridDim.x * kThreads) gx[i] = packed[i].x + bias;
}

// Ids of the rows of a batch whose distinct keys span a small range: a table id_of[key - min] (4 bytes per
// possible key, L2 / Infinity-Cache resident) filled from the G key words, then one gather per row -- the hashed
// look-up (k_lookup_ids) fetches two sectors of a table several times that size per row: 3 ms per 100 M rows.
__global__ __launch_bounds__(kThreads) void k_gx_minmax(const unsigned long long* gx, uint32_t G, int null_bit, unsigned long long flip,
                                                        unsigned long long* out /* [0] min, [1] max; preset ~0, 0 */) {
  unsigned long long mn = ~0ull, mx = 0;
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < G; i += gridDim.x * kThreads) {
    const unsigned long long raw = gx[i];
    if (null_bit < 64 && ((raw >> null_bit) & 1)) continue;  // the null key has no value
    const unsigned long long x = raw ^ flip;  // (signed keys: value order)
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long a = __shfl_xor(mn, d, 64), b = __shfl_xor(mx, d, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0 && mn <= mx) { atomicMin(&out[0], mn); atomicMax(&out[1], mx); }
}
__global__ __launch_bounds__(kThreads) void k_dense_fill(uint32_t* dense, int64_t range, unsigned long long* minmax, uint32_t* null_id,
                                                         int init_minmax) {
  if (init_minmax) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { minmax[0] = ~0ull; minmax[1] = 0; *null_id = kNoRow; }
    return;
  }
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < range; i += int64_t(gridDim.x) * kThreads) dense[i] = kNoRow;
}
__global__ __launch_bounds__(kThreads) void k_dense_scatter(const unsigned long long* gx, uint32_t G, int null_bit, unsigned long long flip,
                                                            unsigned long long kmin, uint32_t* dense, uint32_t* null_id) {
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < G; i += gridDim.x * kThreads) {
    const unsigned long long x = gx[i];
    if (null_bit < 64 && ((x >> null_bit) & 1)) *null_id = i;
    else dense[(x ^ flip) - kmin] = i;
  }
}
__global__ __launch_bounds__(kThreads) void k_lookup_dense(KeyCol c, int64_t n, unsigned long long flip, unsigned long long kmin, const uint32_t* __restrict__ dense,
                                                           const uint32_t* null_id, uint32_t* __restrict__ out_ids, int vec32) {
  if (vec32) {  // 32-bit keys without validity on a 16-byte boundary: four rows per thread
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(c.data) + c.off;
    const int64_t nvec = n / 4;
    const uint32_t k0 = static_cast<uint32_t>(kmin), f = static_cast<uint32_t>(flip);
    for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < nvec; i += int64_t(gridDim.x) * kThreads) {
      const u32x4 k = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(keys) + i);
      u32x4 r;
      r.x = dense[(k.x ^ f) - k0]; r.y = dense[(k.y ^ f) - k0]; r.z = dense[(k.z ^ f) - k0]; r.w = dense[(k.w ^ f) - k0];
      __builtin_nontemporal_store(r, reinterpret_cast<u32x4*>(out_ids) + i);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - nvec * 4) {
      const int64_t r = nvec * 4 + threadIdx.x;
      out_ids[r] = dense[(keys[r] ^ f) - k0];
    }
    return;
  }
  const uint32_t nid = *null_id;
  for (int64_t row = int64_t(blockIdx.x) * kThreads + threadIdx.x; row < n; row += int64_t(gridDim.x) * kThreads) {
    const bool valid = c.valid ? GetBit(c.valid, c.off + row) : true;
    out_ids[row] = valid ? dense[(LoadBitsOfWidth(c.data, c.off + row, c.bits) ^ flip) - kmin] : nid;
  }
}
constexpr int64_t kDenseLookupMax = int64_t(1) << 23;  // keys of a range up to 8 M: a 32 MB table

// Known keys keep their ids, new keys are flagged (in the batch's first-appearance order, which is the order of `packed`).
__global__ __launch_bounds__(kThreads) void k_grouper_probe(const ulonglong4* packed, uint32_t G, unsigned long long bias, GroupTable t,
                                                            uint32_t* slot_of, uint32_t* is_new, unsigned int* full) {
  for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < G; j += gridDim.x * kThreads) {
    const int64_t idx = FindOrInsert(t, Fmix64(packed[j].x + bias), true);  // (distinct keys: no two threads meet on a slot)
    if (idx < 0) { *full = 1; slot_of[j] = 0; is_new[j] = 0; continue; }
    slot_of[j] = static_cast<uint32_t>(idx);
    is_new[j] = t.ids[idx] == kNoRow ? 1u : 0u;
  }
}
__global__ __launch_bounds__(kThreads) void k_grouper_commit(const ulonglong4* packed, uint32_t G, GroupTable t, const uint32_t* slot_of,
                                                             const uint32_t* is_new, const int64_t* new_before, uint32_t id_base,
                                                             unsigned long long bias, unsigned long long* gx) {
  for (uint32_t j = blockIdx.x * kThreads + threadIdx.x; j < G; j += gridDim.x * kThreads) {
    if (!is_new[j]) continue;
    const uint32_t id = id_base + static_cast<uint32_t>(new_before[j]);
    t.ids[slot_of[j]] = id;
    gx[id] = packed[j].x + bias;
  }
}

// A batch of >= 256 K rows of a grouper over one key column of at most 32 bits: the batch's distinct keys and their
// first-appearance order come from the partitioned group-by (no values, no mask: the local / bin / partition
// kernels of amdarrow_filter_groupby); only those G keys meet the grouper's table -- known ones keep their ids,
// new ones are numbered in that order behind the known groups -- and every row then looks its id up.  No insert
// pass of n rows with global atomics (7.4 ms per 100 M rows).  Returns 1 when the batch has to take the general
// path after all (more distinct keys than the attempt was sized for).
static int GrouperBatchGrouped(amdarrow_grouper* g, const amdarrow_array* keys, const KeySpec& ks, uint32_t* out_ids) {
  amdarrow_ctx* ctx = g->ctx;
  const int64_t n = keys[0].length;
  const int64_t cap = std::min<int64_t>(n, int64_t(1) << 22);
  int64_t G = 0;
  ulonglong4* packed = nullptr;
  uint64_t bias = 0;  // a 64-bit key column comes back range-compressed (key word + bias) or whole (bias 0)
  CallerScope who(ctx, kCallerGrouper);
  const int rc = FilterGroupByImpl(ctx, nullptr, &keys[0], nullptr, cap, nullptr, nullptr, nullptr, nullptr, nullptr, &G,
                                   /*locked=*/true, &packed, &bias);
  if (rc == AMDARROW_CAPACITY_ERROR) return 1;
  if (rc != AMDARROW_OK) return rc;
  // room for every distinct key of the batch being new (the table stays under 50 % load)
  const uint64_t known = g->num_groups;
  const uint64_t need = NextPow2(2 * std::max<uint64_t>(known + static_cast<uint64_t>(G), 2048));
  if (need > g->T) {
    char* mem; TableMem tm;
    AA_TRY(GrouperAllocTable(g, need, &mem, &tm));
    GroupTable nt = BindTable(mem, tm, need, true);
    if (known) {
      hipLaunchKernelGGL(k_reinsert, dim3(GridFor(static_cast<int64_t>(known), kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                         g->gx, static_cast<uint32_t>(known), kNoRow, nt);
      AA_HIP(hipGetLastError());
    }
    AA_HIP(hipStreamSynchronize(ctx->stream));
    AA_HIP(hipFree(g->table_mem));
    g->table_mem = mem; g->tm = tm; g->T = need;
  }
  if (known + static_cast<uint64_t>(G) > g->gx_cap) {
    const uint64_t gcap = NextPow2(2 * (known + static_cast<uint64_t>(G)));
    unsigned long long* ngx;
    AA_TRY(DeviceMalloc(ctx, reinterpret_cast<void**>(&ngx), 8 * gcap));
    if (known) AA_HIP(hipMemcpyAsync(ngx, g->gx, 8 * known, hipMemcpyDeviceToDevice, ctx->stream));
    AA_HIP(hipStreamSynchronize(ctx->stream));
    if (g->gx) AA_HIP(hipFree(g->gx));
    g->gx = ngx; g->gx_cap = gcap;
  }
  GroupTable t = BindTable(g->table_mem, g->tm, g->T, true);
  if (G > 0 && known == 0) {
    // an empty grouper: every key is new, ids in the order of `packed`
    hipLaunchKernelGGL(k_gx_from_packed, dim3(GridFor(G, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, packed,
                       static_cast<uint32_t>(G), bias, g->gx);
    hipLaunchKernelGGL(k_reinsert, dim3(GridFor(G, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, g->gx,
                       static_cast<uint32_t>(G), kNoRow, t);
    AA_HIP(hipGetLastError());
    g->num_groups = static_cast<uint32_t>(G);
  } else if (G > 0) {
    // side buffer (not the scratch arena: `packed` lives there): slot and new-flag per distinct key, their scan
    const size_t off_new = (static_cast<size_t>(G) * 4 + 255) & ~size_t(255);
    const size_t off_scan = off_new + ((static_cast<size_t>(G) * 4 + 255) & ~size_t(255));
    const size_t off_full = off_scan + ((static_cast<size_t>(G + 1) * 8 + 255) & ~size_t(255));
    const size_t need_side = off_full + 256;
    if (ctx->side_cap < need_side) {
      AA_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->side_buf) (void)hipFree(ctx->side_buf);
      ctx->side_buf = nullptr;
      ctx->side_cap = 0;
      AA_TRY(DeviceMalloc(ctx, &ctx->side_buf, need_side));
      ctx->side_cap = need_side;
    }
    char* sb = static_cast<char*>(ctx->side_buf);
    auto* slot_of = reinterpret_cast<uint32_t*>(sb);
    auto* is_new = reinterpret_cast<uint32_t*>(sb + off_new);
    auto* new_before = reinterpret_cast<int64_t*>(sb + off_scan);
    auto* full = reinterpret_cast<unsigned int*>(sb + off_full);
    AA_HIP(hipMemsetAsync(full, 0, 4, ctx->stream));
    const int grid = GridFor(G, kThreads, ctx->num_cus * 8);
    hipLaunchKernelGGL(k_grouper_probe, dim3(grid), dim3(kThreads), 0, ctx->stream, packed, static_cast<uint32_t>(G), bias, t, slot_of, is_new, full);
    AA_HIP(hipGetLastError());
    AA_TRY(LaunchScanU32(ctx, is_new, G, new_before));
    hipLaunchKernelGGL(k_grouper_commit, dim3(grid), dim3(kThreads), 0, ctx->stream, packed, static_cast<uint32_t>(G), t, slot_of, is_new,
                       new_before, static_cast<uint32_t>(known), bias, g->gx);
    AA_HIP(hipGetLastError());
    long long added = 0;
    AA_TRY(ReadBack(ctx, new_before + G, 8, &added));
    unsigned int was_full = 0;
    AA_TRY(ReadBack(ctx, full, 4, &was_full));
    if (was_full) return SetError(AMDARROW_CAPACITY_ERROR, "grouper: table full");  // (sized for every key being new: cannot happen)
    g->num_groups = static_cast<uint32_t>(known + static_cast<uint64_t>(added));
  }
  const uint32_t total = g->num_groups;
  // ids of the batch's rows: through a directly indexed table when ALL known keys span a small range
  bool dense_done = false;
  if (total > 0 && !getenv("AMDARROW_GROUPER_HASHED_LOOKUP")) {
    void* base;
    AA_TRY(Scratch(ctx, 256, &base));
    auto* minmax = static_cast<unsigned long long*>(base);
    auto* null_id = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + 64);
    // signed keys in value order (as unsigned: x ^ sign bit), so that a range around zero is a small range
    const unsigned long long flip = IsSignedInt(keys[0].type_id) && ks.col[0].bits > 1 ? 1ull << (ks.col[0].bits - 1) : 0ull;
    hipLaunchKernelGGL(k_dense_fill, dim3(1), dim3(kThreads), 0, ctx->stream, static_cast<uint32_t*>(nullptr), int64_t(0), minmax, null_id, 1);
    hipLaunchKernelGGL(k_gx_minmax, dim3(GridFor(total, kThreads, ctx->num_cus * 4)), dim3(kThreads), 0, ctx->stream, g->gx, total,
                       ks.total_bits, flip, minmax);
    AA_HIP(hipGetLastError());
    unsigned long long mm[2];
    AA_TRY(ReadBack(ctx, minmax, 16, mm));
    const bool any = mm[0] <= mm[1];
    const unsigned long long span = any ? mm[1] - mm[0] : 0;  // (unsigned: 64-bit keys may span more than 2^63)
    const int64_t range = span < static_cast<unsigned long long>(kDenseLookupMax) ? static_cast<int64_t>(span) + 1 : kDenseLookupMax + 1;
    if (range <= kDenseLookupMax) {
      AA_TRY(Scratch(ctx, 256 + 4 * static_cast<size_t>(range), &base));
      minmax = static_cast<unsigned long long*>(base);
      null_id = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + 64);
      auto* dense = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + 256);
      const unsigned long long kmin = any ? mm[0] : 0;
      hipLaunchKernelGGL(k_dense_fill, dim3(1), dim3(kThreads), 0, ctx->stream, dense, range, minmax, null_id, 1);  // (a grown arena: the null id again)
      hipLaunchKernelGGL(k_dense_fill, dim3(GridFor(range, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, dense, range,
                         minmax, null_id, 0);
      hipLaunchKernelGGL(k_dense_scatter, dim3(GridFor(total, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, g->gx, total,
                         ks.total_bits, flip, kmin, dense, null_id);
      const KeyCol c0{keys[0].values, MayHaveNulls(&keys[0]) ? keys[0].validity : nullptr, keys[0].offset, ks.col[0].bits, 0};
      const int vec32 = c0.bits == 32 && !c0.valid &&
                        (reinterpret_cast<uintptr_t>(c0.data) + 4 * static_cast<uintptr_t>(c0.off)) % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(out_ids) % 16 == 0;
      hipLaunchKernelGGL(k_lookup_dense, dim3(GridFor(vec32 ? n / 4 + 1 : n, kThreads, ctx->num_cus * 16)), dim3(kThreads), 0, ctx->stream, c0,
                         n, flip, kmin, dense, null_id, out_ids, vec32);
      AA_HIP(hipGetLastError());
      dense_done = true;
    }
  }
  if (!dense_done) {
    hipLaunchKernelGGL(k_lookup_ids, dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, ks, n, t, out_ids,
                       static_cast<unsigned long long*>(nullptr));
    AA_HIP(hipGetLastError());
  }
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}

int amdarrow_grouper_consume(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids) {
  if (g->wide) return WideIds(g, keys, out_ids, /*insert=*/true);
  amdarrow_ctx* ctx = g->ctx;
  const int64_t n = keys[0].length;
  for (int k = 0; k < g->nkeys; ++k) {
    if (keys[k].type_id != g->types[k]) return SetError(AMDARROW_TYPE_ERROR, "grouper: key type differs from Make()");
    if (keys[k].length != n) return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");
  }
  if (n == 0) return AMDARROW_OK;
  if (n >= (int64_t(1) << 32) - 1) return SetError(AMDARROW_NOT_IMPLEMENTED, "grouper: batches of 2^32 rows or more");
  KeySpec ks;
  AA_TRY(MakeKeySpec(g->nkeys, keys, nullptr, &ks));
  std::lock_guard<std::mutex> lock(ctx->mu);
  // (one key column of at most 32 bits, or of 64 bits without nulls: a null 64-bit key lives in the table's special slot)
  if (g->nkeys == 1 && (ks.total_bits <= 32 || (ks.total_bits == 64 && !MayHaveNulls(&keys[0]))) && n >= (int64_t(1) << 18) &&
      n < (int64_t(1) << 30) && !getenv("AMDARROW_GROUPER_GENERAL")) {
    const int rc = GrouperBatchGrouped(g, keys, ks, out_ids);
    if (rc != 1) return rc;
  }
  // Grow the table so that (known + possible new) groups stay under 50 % load.  Every row of the batch COULD be a
  // new group, but sizing for that costs gigabytes per 50 M-row batch (allocation, clearing, re-insertion) when
  // the batch usually brings few: start with room for 1 M (or as many as are known) new groups and come back
  // with 16x more when the insert pass reports a full table -- the groups it inserted on the way have no id yet
  // and vanish with the old table.
  uint64_t room = static_cast<uint64_t>(n);
  if (room > (uint64_t(1) << 20) && !getenv("AMDARROW_GROUPER_SIZE_FOR_BATCH"))
    room = std::max<uint64_t>(uint64_t(1) << 20, 2 * static_cast<uint64_t>(g->num_groups));
  if (room > static_cast<uint64_t>(n)) room = static_cast<uint64_t>(n);
  int64_t new_groups = 0;
  GroupTable t{};
  RankMem r{};
  for (bool retry = true; retry;) {
  retry = false;
  const uint64_t need = NextPow2(2 * (static_cast<uint64_t>(g->num_groups) + room));
  if (need > g->T) {
    char* mem; TableMem tm;
    AA_TRY(GrouperAllocTable(g, need, &mem, &tm));
    GroupTable nt = BindTable(mem, tm, need, true);
    if (g->num_groups) {
      uint32_t wide = kNoRow;
      AA_TRY(ReadBack(ctx, g->wide_null_id, 4, &wide));
      hipLaunchKernelGGL(k_reinsert, dim3(GridFor(g->num_groups, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0,
                         ctx->stream, g->gx, g->num_groups, wide, nt);
      AA_HIP(hipGetLastError());
    }
    AA_HIP(hipStreamSynchronize(ctx->stream));
    AA_HIP(hipFree(g->table_mem));
    g->table_mem = mem; g->tm = tm; g->T = need;
  }
  if (g->num_groups + room > g->gx_cap) {
    const uint64_t cap = NextPow2(2 * (g->num_groups + room));
    unsigned long long* ngx;
    AA_TRY(DeviceMalloc(ctx, reinterpret_cast<void**>(&ngx), 8 * cap));
    if (g->num_groups) AA_HIP(hipMemcpyAsync(ngx, g->gx, 8ull * g->num_groups, hipMemcpyDeviceToDevice, ctx->stream));
    AA_HIP(hipStreamSynchronize(ctx->stream));
    if (g->gx) AA_HIP(hipFree(g->gx));
    g->gx = ngx; g->gx_cap = cap;
  }
  t = BindTable(g->table_mem, g->tm, g->T, true);
  Carver c;
  size_t roff[4];
  CarveRank(&c, n, roff);
  const size_t off_flag = c.Take(256);
  void* base;
  AA_TRY(Scratch(ctx, c.off, &base));
  r = BindRank(static_cast<char*>(base), roff, n);
  auto* overflow = reinterpret_cast<unsigned int*>(static_cast<char*>(base) + off_flag);
  AA_HIP(hipMemsetAsync(overflow, 0, 4, ctx->stream));
  // pass 1: insert keys, first_row = min(row) on every slot touched
  ValueCol nov{};
  hipLaunchKernelGGL(k_groupby_build, dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, ks, nov,
                     0, static_cast<const uint8_t*>(nullptr), static_cast<const uint8_t*>(nullptr), int64_t(0), n, t,
                     overflow);
  AA_HIP(hipGetLastError());
  // rank the NEW groups (no id yet) by first row; old groups just reset their first_row
  AA_TRY(RankFirstRows(ctx, t, /*only_unassigned=*/1, r, &new_groups));
  unsigned int full = 0;
  AA_TRY(ReadBack(ctx, overflow, 4, &full));
  if (full || static_cast<uint64_t>(new_groups) > room) {
    if (room >= static_cast<uint64_t>(n)) return SetError(AMDARROW_CAPACITY_ERROR, "grouper: table full");
    room = std::min<uint64_t>(static_cast<uint64_t>(n), room * 16);
    // force a fresh table even if the size class does not change: the failed pass left id-less entries behind
    g->T = 0;
    retry = true;
  }
  }  // sizing attempts
  EmitArgs e{};
  e.t = t; e.nslots = static_cast<int64_t>(t.mask) + 3; e.bitmap = r.bitmap; e.prefix = r.prefix;
  e.capacity = static_cast<int64_t>(g->gx_cap); e.id_base = g->num_groups; e.only_unassigned = 1;
  e.gx = g->gx; e.wide_null_id = g->wide_null_id;
  hipLaunchKernelGGL(k_emit_groups, dim3(GridFor(e.nslots, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, e);
  AA_HIP(hipGetLastError());
  // old groups touched by this batch still carry a first_row: clear it for the next batch
  AA_HIP(hipMemsetAsync(t.first, 0xFF, 4 * (t.mask + 3), ctx->stream));
  // pass 2: ids per row
  hipLaunchKernelGGL(k_lookup_ids, dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, ks, n, t,
                     out_ids, static_cast<unsigned long long*>(nullptr));
  AA_HIP(hipGetLastError());
  g->num_groups += static_cast<uint32_t>(new_groups);
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}

/* Ids of rows whose keys are all known already (Grouper::Consume without the growth: no table is sized for n new
 * groups, nothing is inserted).  A key that is not in the table is an error (AMDARROW_KEY_ERROR). */
int amdarrow_grouper_lookup(amdarrow_grouper* g, const amdarrow_array* keys, uint32_t* out_ids) {
  if (g->wide) return WideIds(g, keys, out_ids, /*insert=*/false);
  amdarrow_ctx* ctx = g->ctx;
  const int64_t n = keys[0].length;
  for (int k = 0; k < g->nkeys; ++k) {
    if (keys[k].type_id != g->types[k]) return SetError(AMDARROW_TYPE_ERROR, "grouper: key type differs from Make()");
    if (keys[k].length != n) return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");
  }
  if (n == 0) return AMDARROW_OK;
  if (g->T == 0 || g->num_groups == 0) return SetError(AMDARROW_KEY_ERROR, "grouper: lookup in an empty grouper");
  KeySpec ks;
  AA_TRY(MakeKeySpec(g->nkeys, keys, nullptr, &ks));
  std::lock_guard<std::mutex> lock(ctx->mu);
  GroupTable t = BindTable(g->table_mem, g->tm, g->T, true);
  void* base;
  AA_TRY(Scratch(ctx, kCounterBytes, &base));
  auto* counters = static_cast<unsigned long long*>(base);
  AA_TRY(ClearCounters(ctx, counters));
  hipLaunchKernelGGL(k_lookup_ids, dim3(GridFor(n, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, ks, n, t, out_ids,
                     counters);
  AA_HIP(hipGetLastError());
  unsigned long long missed = 0;
  AA_TRY(ReadCounters(ctx, counters, 1, &missed));
  if (missed) return SetError(AMDARROW_KEY_ERROR, "grouper: " + std::to_string(missed) + " rows hold keys the grouper has not seen");
  return AMDARROW_OK;
}

int amdarrow_grouper_uniques(amdarrow_grouper* g, int k, void* out_values, uint8_t* out_validity, amdarrow_out* out) {
  amdarrow_ctx* ctx = g->ctx;
  if (k < 0 || k >= g->nkeys) return SetError(AMDARROW_INVALID, "grouper_uniques: no such key column");
  if (g->wide) return WideUniques(g, k, out_values, out_validity, out);
  out->length = g->num_groups;
  out->null_count = 0;
  out->has_validity = 0;
  if (g->num_groups == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  void* base;
  AA_TRY(Scratch(ctx, 256, &base));
  auto* nulls = static_cast<unsigned long long*>(base);
  AA_HIP(hipMemsetAsync(nulls, 0, 8, ctx->stream));
  uint32_t wide = kNoRow;
  AA_TRY(ReadBack(ctx, g->wide_null_id, 4, &wide));
  const KeyCol& c = g->spec.col[k];
  const int null_bit = g->spec.wide_null_slot ? -1 : g->spec.total_bits + k;
  hipLaunchKernelGGL(k_decode_keys, dim3(GridFor(g->num_groups, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     g->gx, static_cast<int64_t>(g->num_groups), c.bits, c.shift, null_bit, wide, out_values,
                     reinterpret_cast<uint64_t*>(out_validity), nulls, nullptr);
  AA_HIP(hipGetLastError());
  unsigned long long nn = 0;
  AA_TRY(ReadBack(ctx, nulls, 8, &nn));
  out->null_count = static_cast<int64_t>(nn);
  out->has_validity = nn > 0;  // DecodeNulls allocates the bitmap only when a null key exists (:64-92)
  return AMDARROW_OK;
}

// acc[id] += the batch's per-group result; every id occurs once in (gkeys, gsums, gcounts): plain read-modify-write
__global__ __launch_bounds__(kThreads) void k_add_by_id(const uint32_t* gkeys, const unsigned long long* gsums, const int64_t* gcounts,
                                                        int64_t G, int is_double, int with_sums, unsigned long long* sums,
                                                        unsigned long long* counts) {
  for (int64_t g = int64_t(blockIdx.x) * kThreads + threadIdx.x; g < G; g += int64_t(gridDim.x) * kThreads) {
    const uint32_t id = gkeys[g];
    if (with_sums) {
      if (is_double) sums[id] = static_cast<unsigned long long>(__double_as_longlong(__longlong_as_double(sums[id]) + __longlong_as_double(gsums[g])));
      else sums[id] += gsums[g];
    }
    counts[id] += static_cast<unsigned long long>(gcounts[g]);
  }
}

// hash_sum / hash_count / hash_mean of a batch of >= 256 K rows: the ids ARE a key column of the partitioned
// group-by (dense, 32 bits: bin scatter + directly indexed tables, or the local tables when a row range holds few
// of them), whose per-group sums and valid counts are then added to the accumulators -- one plain update per
// group instead of two memory-side atomics per ROW (k_hash_agg: 8.7 ms per 100 M rows with 1 M groups, 14 ms with
// 1000, where every row fights for the same few words).  Returns 1 when the batch has to take the atomics
// after all (more distinct ids than the attempt was sized for).
static int HashAccumulateGrouped(amdarrow_ctx* ctx, int func, const amdarrow_array* values, const uint32_t* ids,
                                 unsigned long long* sums, unsigned long long* counts) {
  const int64_t n = values->length;
  const int64_t cap = std::min<int64_t>(n, int64_t(1) << 22);
  const size_t off_valid = (static_cast<size_t>(cap) * 20 + 255) & ~size_t(255);
  const size_t need = off_valid + static_cast<size_t>((cap + 63) / 64) * 8 + 256;
  if (ctx->side_cap < need) {
    AA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->side_buf) (void)hipFree(ctx->side_buf);
    ctx->side_buf = nullptr;
    ctx->side_cap = 0;
    AA_TRY(DeviceMalloc(ctx, &ctx->side_buf, need));
    ctx->side_cap = need;
  }
  char* sb = static_cast<char*>(ctx->side_buf);
  auto* gsums = reinterpret_cast<unsigned long long*>(sb);
  auto* gcounts = reinterpret_cast<int64_t*>(sb + 8 * cap);
  auto* gkeys = reinterpret_cast<uint32_t*>(sb + 16 * cap);
  auto* gvalid = reinterpret_cast<uint8_t*>(sb + off_valid);
  amdarrow_array key{};
  key.values = const_cast<uint32_t*>(ids); key.validity = nullptr; key.length = n; key.offset = 0; key.type_id = AMDARROW_UINT32;
  key.null_count = 0;
  int64_t G = 0;
  CallerScope who(ctx, kCallerHashAccumulate);
  const int rc = FilterGroupByImpl(ctx, values, &key, nullptr, cap, gsums, gcounts, gkeys, gvalid, nullptr, &G, /*locked=*/true, nullptr);
  if (rc == AMDARROW_CAPACITY_ERROR) return 1;
  if (rc != AMDARROW_OK) return rc;
  if (G > 0) {
    hipLaunchKernelGGL(k_add_by_id, dim3(GridFor(G, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, gkeys, gsums, gcounts, G,
                       IsFloat(values->type_id) ? 1 : 0, func != AMDARROW_HASH_COUNT ? 1 : 0, sums, counts);
    AA_HIP(hipGetLastError());
  }
  return AMDARROW_OK;
}

int amdarrow_hash_accumulate(amdarrow_ctx* ctx, int func, const amdarrow_array* values, const uint32_t* ids,
                             int count_mode, void* sums, int64_t* counts) {
  if (func < AMDARROW_HASH_SUM || func > AMDARROW_HASH_MEAN) return SetError(AMDARROW_KEY_ERROR, "No such hash aggregate");
  const int t = values->type_id;
  if (BitWidth(t) < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "Summing data of this type");  // :556-562
  if (values->length == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (values->length >= (int64_t(1) << 18) && values->length < (int64_t(1) << 30) &&
      !(func == AMDARROW_HASH_COUNT && count_mode == AMDARROW_COUNT_NULL) && !getenv("AMDARROW_HASH_ATOMICS")) {
    const int rc = HashAccumulateGrouped(ctx, func, values, ids, static_cast<unsigned long long*>(sums),
                                         reinterpret_cast<unsigned long long*>(counts));
    if (rc != 1) return rc;
  }
  ValueCol v{values->values, values->validity, values->offset, t};
  hipLaunchKernelGGL(k_hash_agg, dim3(GridFor(values->length, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, v,
                     ids, values->length, func, count_mode, static_cast<unsigned long long*>(sums),
                     reinterpret_cast<unsigned long long*>(counts));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_hash_finalize(amdarrow_ctx* ctx, int func, int value_type_id, int64_t num_groups, void* sums,
                           const int64_t* counts, uint8_t* out_validity, amdarrow_out* out) {
  if (func < AMDARROW_HASH_SUM || func > AMDARROW_HASH_MEAN) return SetError(AMDARROW_KEY_ERROR, "No such hash aggregate");
  out->length = num_groups;
  out->null_count = 0;
  out->has_validity = 0;
  if (num_groups == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  void* base;
  AA_TRY(Scratch(ctx, 256, &base));
  auto* nulls = static_cast<unsigned long long*>(base);
  AA_HIP(hipMemsetAsync(nulls, 0, 8, ctx->stream));
  hipLaunchKernelGGL(k_agg_finalize, dim3(GridFor(num_groups, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     func, AccKind(value_type_id), num_groups, static_cast<unsigned long long*>(sums),
                     reinterpret_cast<const unsigned long long*>(counts), reinterpret_cast<uint64_t*>(out_validity), nulls);
  AA_HIP(hipGetLastError());
  unsigned long long nn = 0;
  AA_TRY(ReadBack(ctx, nulls, 8, &nn));
  out->null_count = static_cast<int64_t>(nn);
  out->has_validity = nn > 0;  // hash_sum: bitmap only if a group has no valid value (:585-600); hash_count: never
  return AMDARROW_OK;
}

int amdarrow_hash_aggregate(amdarrow_ctx* ctx, int func, const amdarrow_array* values, const uint32_t* ids,
                            int64_t num_groups, int count_mode, void* out_values, uint8_t* out_validity,
                            amdarrow_out* out) {
  out->length = num_groups;
  out->null_count = 0;
  out->has_validity = 0;
  if (num_groups == 0) return AMDARROW_OK;
  void* counts = nullptr;
  AA_TRY(amdarrow_malloc(ctx, 8 * num_groups, &counts));
  int rc = AMDARROW_OK;
  if (hipMemsetAsync(counts, 0, 8 * num_groups, ctx->stream) != hipSuccess ||
      hipMemsetAsync(out_values, 0, 8 * num_groups, ctx->stream) != hipSuccess)
    rc = SetError(AMDARROW_UNKNOWN_ERROR, "hipMemsetAsync failed");
  if (rc == AMDARROW_OK) rc = amdarrow_hash_accumulate(ctx, func, values, ids, count_mode, out_values, static_cast<int64_t*>(counts));
  if (rc == AMDARROW_OK)
    rc = amdarrow_hash_finalize(ctx, func, values->type_id, num_groups, out_values, static_cast<int64_t*>(counts), out_validity, out);
  (void)amdarrow_free(ctx, counts);
  return rc;
}

}  // extern "C"

// The fused Filter -> GroupBy.  `locked`: the caller already holds ctx->mu (the multi-key entry point, which runs
// this on its column of packed key codes).
constexpr int kHintProbeEvery = 64;  // hinted group-by calls between two calls that try the per-workgroup pre-aggregation again
static int FilterGroupByImpl(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* keys,
                             const amdarrow_array* mask, int64_t group_capacity, void* out_sums, int64_t* out_counts,
                             void* out_keys, uint8_t* out_key_validity, uint32_t* out_first_rows,
                             int64_t* out_num_groups, bool locked, ulonglong4** packed_out, uint64_t* bias_out) {
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
    amdarrow_agg kr{};
    if (!bias_from_hint) AA_TRY(AggregateLocked(ctx, keys, &kr));
    if (bias_from_hint || (kr.valid_count > 0 && kr.max - kr.min < 0xFFFFFFFFull)) {  // (wrap-around difference: exact for both signednesses)
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
    const int rc = FilterGroupByImpl(ctx, values, keys, mask, group_capacity, out_sums, out_counts, out_keys, out_key_validity, out_first_rows,
                                     out_num_groups, /*locked=*/true, packed_out, bias_out);
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
  if (path == 3 && shape_known &&
      !getenv("AMDARROW_GROUPBY_PATH") && !getenv("AMDARROW_GROUPBY_NO_HINT")) {
    int every = kHintProbeEvery;
    if (const char* env = getenv("AMDARROW_GROUPBY_PROBE_EVERY")) every = std::max(2, atoi(env));
    if (hint.local_ok) local_only = true;
    else if (++hint.uses % every != 0) { path = 2; hinted = true; }
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
  const bool use_bins = bins_shape && !getenv("AMDARROW_GROUPBY_NO_BINS") &&
                        (reinterpret_cast<uintptr_t>(keys->values) + (k_load_bits / 8) * static_cast<uintptr_t>(keys->offset)) % 16 == 0 &&
                        (values ? (reinterpret_cast<uintptr_t>(values->values) + 8 * static_cast<uintptr_t>(values->offset)) % 16 == 0
                                : reinterpret_cast<uintptr_t>(keys->values) % 16 == 0);
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
  GroupTable t{};
  RankMem r{};
  unsigned long long* gx = nullptr;
  unsigned int* misc = nullptr;  // status words, see RankFirstRows; [64..] one flag per workgroup of the local path
  constexpr int kMaxFlags = 1024;
  int sub_bits = 0;  // 2 after a partition's LDS table overflowed: four sub-problems, up to ~7 M groups
  bool fa_val_wide = false;  // the last attempt's scatter reported whether the values were all int32 (status[10])
  for (;; --path) {
    fa_val_wide = false;
    const bool fast = path > 0;
    const int nsub = (fast && path == 2) ? 1 << sub_bits : 1;
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
      const double mean = static_cast<double>(rows_per_block) / kP / nsub;
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
    } else if (path == 1) {
      nblocks = static_cast<int>(std::min<int64_t>(ctx->num_cus, (n + kPartRows - 1) / kPartRows));
      rows_per_block = (((n + nblocks - 1) / nblocks + kPartRows - 1) / kPartRows) * kPartRows;
      hist_n = int64_t(kP) * nblocks + 1;
    }
    const int64_t scan_nb = (hist_n + 2047) / 2048;
    const size_t off_hist = c.Take(4 * hist_n);
    const size_t off_sums = c.Take(4 * scan_nb);
    const size_t off_sumoff = c.Take(8 * (scan_nb + 1));
    const int nseg = nblocks + (path >= 2 && use_bins ? 1 : 0);  // segments per partition the aggregate walks
    const size_t rec_bytes = path >= 2 ? std::max(16 * static_cast<size_t>(cap) * (compact ? size_t(1) << compact_pb : size_t(kP)) * nseg, LocalStageBytes(nblocks_local))
                           : path == 1 ? 16 * static_cast<size_t>(n) : 0;
    const size_t off_rec = c.Take(rec_bytes);
    const size_t off_hi = c.Take(path == 1 && wide ? 4 * static_cast<size_t>(n) : 0);
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
                       path >= 2 && use_bins ? reinterpret_cast<uint32_t*>(b + off_hist) : static_cast<uint32_t*>(nullptr), nseg,
                       path >= 2 && use_bins ? misc + 8 : static_cast<unsigned int*>(nullptr));
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
        for (int sub = 0; sub < (local_only && path == 3 ? 0 : nsub); ++sub) {  // one pass per sub-problem (nsub == 1 unless a table overflowed before)
        fa.sub_bits = nsub > 1 ? sub_bits : 0; fa.sub_id = static_cast<uint32_t>(sub);
        FastArgs fh = fa;  // the same scatter on the mask k_hot_aggregate wrote
        fh.m_data = hot.alt_mask; fh.m_valid = nullptr; fh.m_off = 0;
        if (compact) {
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
        if (compact) {
#define AA_CAGG(RR, PP) hipLaunchKernelGGL((k_part_aggregate_compact<RR, PP>), dim3(1 << PP), dim3(kPartThreads), 0, ctx->stream, records, hist, nblocks, cap, \
                               acc_double, t, r.bitmap, misc, static_cast<int64_t>(T), c_base, fa.k_flip, static_cast<uint32_t>(rows_per_block))
          if (compact_rw == 2 && compact_pb == 9) AA_CAGG(2, 9); else if (compact_rw == 2) AA_CAGG(2, 8);
          else if (compact_pb == 9) AA_CAGG(3, 9); else AA_CAGG(3, 8);
#undef AA_CAGG
        } else if (use_bins)
          hipLaunchKernelGGL(k_part_aggregate_piped, dim3(kP), dim3(kPartThreads), 0, ctx->stream, records, hist, nseg,
                             cap, ks.total_bits, acc_double, t, r.bitmap, misc, run_if, int64_t(sub) * kP * kSlots,
                             sub == 0 ? static_cast<int64_t>(T) : int64_t(0), fa.key_minmax, fa.k_flip, use_hot);
        else
        hipLaunchKernelGGL(k_part_aggregate_direct, dim3(kP), dim3(kPartThreads), 0, ctx->stream, records, hist, nseg,
                           cap, ks.total_bits, acc_double, t, r.bitmap, misc, run_if, int64_t(sub) * kP * kSlots,
                           sub == 0 ? static_cast<int64_t>(T) : int64_t(0));
        AA_HIP(hipGetLastError());
        }  // sub-problems
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
    if (overflow && fast && hinted) {  // the hint (or the guess) was wrong for this input: forget it, start over with the full path 3
      guessed = false;
      guess_tried = true;  // (one compact attempt per call: what broke a hinted call's assumption -- a few far keys, one wide value -- hides from a sample too)
      hinted = false;
      hint.n = -1;
      path = 4;  // -> 3 after the loop's decrement
      continue;
    }
    if (overflow && fast) {  // capacity assumption failed: next (more general) path
      if (path == 3) path = 2;  // the partition attempt was part of path 3
      if (overflow == kOverflowTable) {
        // the two-pass variant has the same LDS tables.  More groups than 512 partitions hold: once
        // more with four sub-problems (four passes over the input, 2048 partitions in effect),
        // then the generic global-atomic table
        if (sub_bits == 0) { sub_bits = 2; path = 3; }  // -> 2 after the loop's decrement
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
    if (hint.n == shape.n) {  // the back-off state stays with the shape; a compact call that went through ends a run of misses
      hint.compact_misses = compact && carry_misses > 0 ? carry_misses - 1 : carry_misses;
      hint.compact_off = carry_off;
    }
    *out_num_groups = G;
    ctx->gb_last_path = path == 3 && status[1] != 0 ? 2 : path;
    ctx->gb_last_words = ctx->gb_last_path == 0 || ctx->gb_last_path == 3 ? 0 : compact ? compact_rw : 4;
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

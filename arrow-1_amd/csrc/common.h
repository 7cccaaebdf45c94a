// common.h -- shared host/device helpers for the gfx950 kernels.  Internal; the public
// surface is include/amdarrow.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "amdarrow.h"

namespace amdarrow {

constexpr int kWave = 64;  // CDNA4 wavefront

// ---- status / errors ---------------------------------------------------------------------
int SetError(int code, const std::string& msg);
int HipError(hipError_t e, const char* what, const char* file, int line);

#define AA_HIP(expr)                                                        \
  do {                                                                      \
    hipError_t _e = (expr);                                                 \
    if (_e != hipSuccess) return ::amdarrow::HipError(_e, #expr, __FILE__, __LINE__); \
  } while (0)
#define AA_TRY(expr)             \
  do {                           \
    int _rc = (expr);            \
    if (_rc != AMDARROW_OK) return _rc; \
  } while (0)

}  // namespace amdarrow

// ---- context -----------------------------------------------------------------------------
struct amdarrow_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* scratch = nullptr;  // grow-only HBM arena for per-call temporaries
  size_t scratch_cap = 0;
  void* pinned = nullptr;   // 8 KiB pinned, device-mapped host memory: [0, 4096) ReadBack, [4096, ..) status words
  void* pinned_dev = nullptr;  // the same memory as the device sees it (kernels may store status words there)
  hipEvent_t status_event = nullptr;  // marks the kernel that stored them
  int num_cus = 256;
  // groupby.hip: the shape of the last fused group-by call whose optimistic local pre-aggregation gave
  // up without finding heavy hitters -- the next call of that shape starts at the partition path
  struct GroupByHint {
    int64_t n = -1; int key_type = 0, value_type = 0, has_mask = 0, has_value_validity = 0, uses = 0;
    // what the call that set the hint saw (bin scatter only; `stats` = the three below are meaningful): smallest /
    // largest selected key in the scatter's unsigned key order plus the call's key bias, and whether every
    // selected valid value of an integer column was an int32 -- what a compact-record call assumes and checks
    int stats = 0, values_narrow = 0; uint64_t key_lo = 1, key_hi = 0;
    uint64_t k_bias = 0;  // the bias of that call's range-compressed 64-bit key column (the next compact call skips the min / max pass)
    int local_ok = 0;     // that call's per-workgroup pre-aggregation held every group: the next call queues nothing behind it
    // who asked (amdarrow_filter_groupby, the packed multi-key call, hash_accumulate over ids, a Grouper batch): part of the
    // shape, so that entry points whose calls happen to have equal n and types do not trade each other's assumptions
    int caller = 0;
    // a running score of compact-record calls of this shape that had to start over (+2 each, -1 per call that went
    // through), and how many calls stay on 16-byte records once it reached four (interleaved data sets of one shape
    // would otherwise pay for two attempts every other call)
    int compact_misses = 0, compact_off = 0;
    int no_bins = 0;  // calls of this shape that stay off the bin scatters (their bins overflowed: clustered keys), see FilterGroupByImpl
    int no_bins_next = 16;  // ... and how many it will be after the next overflow (doubles up to 1024; back to 16 once the bins hold)
    int sub_bits = 0;  // that call needed 4 / 16 rounds per partition (more groups than 512 LDS tables hold): the next calls start there
    int wide_keys = 0, wide_uses = 0;  // that call's 64-bit keys did not range-compress: the next calls skip the min / max pass and go the wide-key way (all but every 64th)
    uint64_t last_used = 0;
  };
  // multi-key sorts on the leading 64 bits of a wider composite key: the window kernels of the hybrid sort mark every row
  // whose key equals another row's (bit 63 of its output index; rows < 2^31) and report that they ran
  bool sort_mark_ties = false, sort_ties_marked = false;
  int gb_caller = 0;  // set by the entry point around FilterGroupByImpl (under `mu`)
  int gb_carry_misses = -1, gb_carry_off = 0;  // back-off state handed to the retry of a call whose hinted bias did not hold (-1: none)
  bool gb_no_bias_hint = false;  // set around the retry of a call whose hinted bias did not hold
  // one hint per call shape, the least recently used one gives way: queries that alternate keep their hints
  static constexpr int kGroupByHints = 4;
  GroupByHint gb_hints[kGroupByHints];
  uint64_t gb_hint_clock = 0;
  int gb_last_path = -1, gb_last_words = 0;  // amdarrow_groupby_last_path
  int gb_hot_guess_off = 0;  // calls that stay off the guessed compact records for skewed input (set when such a guess missed: a skewed shape keeps no hint that could remember it)
  // amdarrow_filter_groupby_multi: the key columns' ranges as the last call of this shape found them (32-bit columns
  // without nulls): the next call packs with them -- the pack kernel checks every value -- instead of a min / max pass
  // per column first
  struct PackHint { int64_t n = -1; int ncols = 0; int types[4] = {0, 0, 0, 0}; uint64_t bias[4] = {0, 0, 0, 0}; uint32_t span[4] = {0, 0, 0, 0}; } pack_hint;
  void* pack_buf = nullptr;  // groupby.hip amdarrow_filter_groupby_multi: packed key codes (grow-only)
  size_t pack_cap = 0;
  void* side_buf = nullptr;  // groupby.hip amdarrow_hash_accumulate: the inner group-by's per-group outputs (grow-only)
  size_t side_cap = 0;
  int lds_rank_ordered = -1;  // sort.hip LdsAtomicsLaneOrdered(): -1 not probed yet, 0 no, 1 yes
  int64_t bytes_allocated = 0;
  int64_t max_memory = 0;
  std::unordered_map<void*, int64_t> allocs;
  // amdarrow_malloc / amdarrow_free keep freed blocks by size class instead of returning them to the device: a
  // hipFree is a device-wide synchronisation and a hipMalloc of a few MB costs 0.1-0.3 ms -- two thirds of a
  // GroupBy call through the C++ API went there.  A cached block is handed out again only to work queued on the
  // context's stream, i.e. behind whatever still uses it.
  std::unordered_map<void*, int64_t> alloc_class;              // block -> size class (bytes actually allocated)
  std::unordered_map<int64_t, std::vector<void*>> free_blocks;  // size class -> cached blocks
  int64_t cached_bytes = 0;
  int64_t cache_limit = int64_t(16) << 30;  // cached bytes beyond which a freed block goes back to the device (set at creation)
  std::mutex pool_mu;  // guards the seven pool fields above; never held across a kernel launch, may be taken under `mu`
  std::mutex mu;
};

namespace amdarrow {

// Ensure the scratch arena holds `bytes`; returns its base (256-B aligned).
int Scratch(amdarrow_ctx* ctx, size_t bytes, void** out);
// hipMalloc for the library's own buffers (arena, grouper tables, side buffers, temporaries).  When the device is
// out of memory the pool's cached blocks (amdarrow_free keeps them) are given back and the allocation is tried once
// more: memory sitting idle in the cache must never fail a call.
int DeviceMalloc(amdarrow_ctx* ctx, void** out, size_t bytes);
// Give every cached block of the pool back to the device (synchronises the context's stream first); returns the bytes freed.
int64_t DrainPool(amdarrow_ctx* ctx);

// A bump allocator over the scratch arena (sizes rounded to 256 B).
struct Carver {
  size_t off = 0;
  size_t Take(size_t bytes) {
    size_t at = off;
    off += (bytes + 255) & ~size_t(255);
    return at;
  }
};

// Status words: kernels store small results (a total, an overflow flag) at pinned_dev +
// kStatusOffset, the host records status_event behind that kernel, queues whatever does not need
// the value, and only then waits for the event -- the round trip overlaps with the queued work.
// (ReadBack uses the head of the same pinned block.)
constexpr size_t kStatusOffset = 4096;  // behind the 4 KiB ReadBack area

// Copy `nbytes` (<= 4096) from device to the pinned word and wait for it.
int ReadBack(amdarrow_ctx* ctx, const void* dev, size_t nbytes, void* host_out);

// Slotted counters.  A device-scope atomic on ONE address is serialised at the memory side
// (~30 ns each on MI355X: 2048 workgroups adding to one word cost 60 us, more than the kernels
// that did it), so global sums go to kCounterSlots copies, each in its own 128-byte line, picked
// by the workgroup id; the host adds them up after ONE 4 KiB read-back.  Up to kCounterStride
// counters per slot.
constexpr int kCounterSlots = 32;
constexpr int kCounterStride = 16;  // unsigned long long per slot = 128 bytes
constexpr size_t kCounterBytes = size_t(kCounterSlots) * kCounterStride * 8;
// Zero the block (asynchronously); `base` is kCounterBytes of device memory.
int ClearCounters(amdarrow_ctx* ctx, unsigned long long* base);
// Sum counter `k` (k < ncounters <= kCounterStride) over the slots into out[k].
int ReadCounters(amdarrow_ctx* ctx, const unsigned long long* base, int ncounters, unsigned long long* out);

// amdarrow_aggregate without taking ctx->mu (aggregate.hip).
int AggregateLocked(amdarrow_ctx* ctx, const amdarrow_array* a, amdarrow_agg* out);

// Single-workgroup exclusive scan of u32 counts into int64 offsets (offsets[n] = total).
int LaunchScanU32(amdarrow_ctx* ctx, const uint32_t* counts, int64_t n, int64_t* offsets);
// Device-wide in-place exclusive scan of u32 (total < 2^32).  sums: ceil(n/2048) u32,
// sum_offsets: ceil(n/2048) + 1 int64 of scratch; sum_offsets[ceil(n/2048)] = total.
int ScanU32InPlace(amdarrow_ctx* ctx, uint32_t* data, int64_t n, uint32_t* sums, int64_t* sum_offsets);

// Device-wide exclusive scan of int64 `lengths` into offsets of 32 or 64 bits (out_offsets[n] = total; *overflow is
// raised when a 32-bit total passes INT32_MAX - 1).  sums: Scan64Blocks(n) + 1 int64 of scratch; sums[Scan64Blocks(n)] = total.
// (selection_nested.hip)
inline int64_t Scan64Blocks(int64_t n) { return (n + 2047) / 2048; }
int LaunchScan64(amdarrow_ctx* ctx, const long long* lengths, int64_t n, long long* sums, int offset_bits, void* out_offsets,
                 unsigned int* overflow);

inline int BitWidth(int type_id) {
  switch (type_id) {
    case AMDARROW_BOOL: return 1;
    case AMDARROW_UINT8: case AMDARROW_INT8: return 8;
    case AMDARROW_UINT16: case AMDARROW_INT16: return 16;
    case AMDARROW_UINT32: case AMDARROW_INT32: case AMDARROW_FLOAT: return 32;
    case AMDARROW_UINT64: case AMDARROW_INT64: case AMDARROW_DOUBLE: return 64;
    default: return -1;
  }
}
inline bool IsSignedInt(int t) { return t == AMDARROW_INT8 || t == AMDARROW_INT16 || t == AMDARROW_INT32 || t == AMDARROW_INT64; }
inline bool IsUnsignedInt(int t) { return t == AMDARROW_UINT8 || t == AMDARROW_UINT16 || t == AMDARROW_UINT32 || t == AMDARROW_UINT64; }
inline bool IsFloat(int t) { return t == AMDARROW_FLOAT || t == AMDARROW_DOUBLE; }
inline int64_t BytesForBits(int64_t n) { return (n + 7) >> 3; }
// GetPrimitiveArg (compute/kernels/util_internal.cc:45-58): 0 without a validity buffer,
// else the stored count (possibly -1).
inline int64_t ArgNullCount(const amdarrow_array* a) { return a->validity ? a->null_count : 0; }
// ArrayData::MayHaveNulls
inline bool MayHaveNulls(const amdarrow_array* a) { return a->validity && a->null_count != 0; }

inline int GridFor(int64_t work_items, int per_block, int max_blocks) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return static_cast<int>(b);
}

#ifdef __HIPCC__
// ---- device bit helpers --------------------------------------------------------------------
// 64 bits of `bm` starting at absolute bit position `pos`; bits at and beyond `nbits`
// (1..64) are returned as zero.  Only aligned 8-byte words that contain a requested bit
// are touched, so a buffer padded to the next 8-byte boundary is never over-read.
__device__ __forceinline__ uint64_t LoadBits64(const uint8_t* __restrict__ bm, int64_t pos, int nbits) {
  const uintptr_t addr = reinterpret_cast<uintptr_t>(bm) + static_cast<uintptr_t>(pos >> 3);
  const uint64_t* w = reinterpret_cast<const uint64_t*>(addr & ~uintptr_t(7));
  const int shift = static_cast<int>((addr & 7) << 3) + static_cast<int>(pos & 7);
  uint64_t v = w[0] >> shift;
  if (shift + nbits > 64) v |= w[1] << (64 - shift);
  if (nbits < 64) v &= (uint64_t(1) << nbits) - 1;
  return v;
}
// Same, but an absent bitmap reads as all ones.
__device__ __forceinline__ uint64_t LoadBitsOrOnes(const uint8_t* __restrict__ bm, int64_t pos, int nbits) {
  if (bm == nullptr) return nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0);
  return LoadBits64(bm, pos, nbits);
}
// This workgroup's copy of the slotted counters (see ClearCounters / ReadCounters).
__device__ __forceinline__ unsigned long long* CounterSlot(unsigned long long* base) {
  return base + (blockIdx.x & (kCounterSlots - 1)) * kCounterStride;
}
__device__ __forceinline__ bool GetBit(const uint8_t* __restrict__ bm, int64_t pos) {
  return (bm[pos >> 3] >> (pos & 7)) & 1;
}
__device__ __forceinline__ int LaneId() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0));
}
// popcount of the bits of `mask` below this lane
__device__ __forceinline__ int RankBelowLane(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0));
}
// Store bitmap word `word_index` (64 bits) of a bitmap holding `total_bits` bits without
// writing past BytesForBits(total_bits).  `bm` must be 8-byte aligned.
__device__ __forceinline__ void StoreBitmapWord(uint8_t* __restrict__ bm, int64_t word_index, uint64_t w,
                                                int64_t total_bits) {
  const int64_t nbytes = (total_bits + 7) >> 3;
  const int64_t byte0 = word_index << 3;
  if (byte0 + 8 <= nbytes) {
    reinterpret_cast<uint64_t*>(bm)[word_index] = w;
  } else {
    for (int64_t b = byte0; b < nbytes; ++b) bm[b] = static_cast<uint8_t>(w >> ((b - byte0) << 3));
  }
}
// Inclusive scan of one value per lane across the 64-lane wave.
__device__ __forceinline__ uint32_t WaveInclusiveScan(uint32_t v) {
  const int lane = LaneId();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ uint64_t WaveReduceSum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
#endif  // __HIPCC__

}  // namespace amdarrow

// sort.hip -- sort_indices for numeric arrays / chunked arrays: a stable LSD radix sort.
//
// Reference semantics: compute/kernels/vector_sort.cc
//   PartitionNulls[Only|Likes] :224-313  (nulls last, NaNs just before them, both in index order)
//   ArrayCompareSorter :373-405          (std::stable_sort on `lhs < rhs`; Descending uses `rhs < lhs`,
//                                          so ties keep ASCENDING index order in both directions)
//   ArrayCountSorter :407-478            (same order by construction)
//   ChunkedArraySorter :657-832          (logical index = position in the concatenation)
// The required output is therefore unique: sort the non-null, non-NaN rows by
// (order-preserving key, index), then NaN rows by index, then null rows by index.
//
// Pipeline (all HBM-bound, no comparisons):
//   1. classify + stable 3-way partition per 2048-row tile (ballot ranks): sortable rows ->
//      (key, index32) arrays, NaN / null rows -> their final slots at the tail of the output
//   2. one pass builds the histograms of every 8-bit digit; digits on which all keys agree
//      are skipped (an int64 column holding [-100, 100] needs 2 of 8 passes)
//   3. per remaining digit: block histograms -> device-wide scan -> stable scatter.  Inside a
//      4096-key block, ranks come from 8 ballots per 64-key step (match-any on the digit) and
//      keys are first sorted by digit in LDS so each digit leaves as one contiguous run.
// Keys: unsigned as is; signed with the sign bit flipped; IEEE floats with -0.0 folded onto
// +0.0 (they tie under `<`) then sign-magnitude -> biased; Descending complements the key.
#include "common.h"

#include <algorithm>
#include <cstring>
#include <type_traits>
#include <vector>

namespace amdarrow {

constexpr int kThreads = 256;
constexpr int kCTile = 2048;   // classify tile: 256 threads x 8
constexpr int kRTile = 4096;   // radix tile: 16 keys per lane of a 256-thread block
#ifndef ONESWEEP_TILE
#define ONESWEEP_TILE 8192
#define ONESWEEP_THREADS 512
#endif
constexpr int kOTileRows = ONESWEEP_TILE;  // one-sweep tile (= kOTile below)

struct SortChunk {
  const void* values; const uint8_t* valid; int64_t off; int64_t length;
  int64_t row_start;   // logical index of the chunk's first row
  int64_t tile_start;  // first classify tile of the chunk
  int64_t otile_start; // first one-sweep tile (kOTile rows) of the chunk: the classified raw first pass, see SortRawClassified
};

int LaunchScanU32(amdarrow_ctx* ctx, const uint32_t* counts, int64_t n, int64_t* offsets);  // selection.hip

template <typename KeyT> __device__ __forceinline__ KeyT MakeKey(int type_id, const void* data, int64_t i, bool* is_nan);
template <> __device__ __forceinline__ uint32_t MakeKey<uint32_t>(int t, const void* d, int64_t i, bool* is_nan) {
  *is_nan = false;
  switch (t) {
    case AMDARROW_UINT8: return reinterpret_cast<const uint8_t*>(d)[i];
    case AMDARROW_INT8: return static_cast<uint8_t>(reinterpret_cast<const uint8_t*>(d)[i] ^ 0x80u);
    case AMDARROW_UINT16: return reinterpret_cast<const uint16_t*>(d)[i];
    case AMDARROW_INT16: return static_cast<uint16_t>(reinterpret_cast<const uint16_t*>(d)[i] ^ 0x8000u);
    case AMDARROW_UINT32: return reinterpret_cast<const uint32_t*>(d)[i];
    case AMDARROW_INT32: return reinterpret_cast<const uint32_t*>(d)[i] ^ 0x80000000u;
    default: {  // FLOAT
      float f = reinterpret_cast<const float*>(d)[i];
      *is_nan = f != f;
      if (f == 0.0f) f = 0.0f;  // -0.0 == +0.0 is a tie
      const uint32_t u = __float_as_uint(f);
      return (u >> 31) ? ~u : (u | 0x80000000u);
    }
  }
}
template <> __device__ __forceinline__ uint64_t MakeKey<uint64_t>(int t, const void* d, int64_t i, bool* is_nan) {
  *is_nan = false;
  switch (t) {
    case AMDARROW_UINT64: return reinterpret_cast<const uint64_t*>(d)[i];
    case AMDARROW_INT64: return reinterpret_cast<const uint64_t*>(d)[i] ^ 0x8000000000000000ULL;
    default: {  // DOUBLE
      double f = reinterpret_cast<const double*>(d)[i];
      *is_nan = f != f;
      if (f == 0.0) f = 0.0;
      const uint64_t u = static_cast<uint64_t>(__double_as_longlong(f));
      return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    }
  }
}

__device__ __forceinline__ int FindChunk(const SortChunk* chunks, int nchunks, int64_t tile) {
  int lo = 0, hi = nchunks - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunks[mid].tile_start <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Chunk holding logical row `row` (the last chunk that starts at or before it; an empty chunk
// shares its start with its successor, so it is never the answer for a row that exists).
__device__ __forceinline__ int FindChunkOfRow(const SortChunk* chunks, int nchunks, int64_t row) {
  int lo = 0, hi = nchunks - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunks[mid].row_start <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// MakeKey in two halves, for loops that want all their loads in flight before the first use: N raw values (the
// switch on the width sits outside the loop over the values -- inside, every load ends up behind a branch and a wait),
// then the arithmetic.
template <typename KeyT, int N>
__device__ __forceinline__ void LoadRawKeys(int t, const void* d, const int64_t (&i)[N], KeyT (&raw)[N]) {
  if constexpr (sizeof(KeyT) == 8) {
#pragma unroll
    for (int j = 0; j < N; ++j) raw[j] = reinterpret_cast<const uint64_t*>(d)[i[j]];
  } else {
    if (t == AMDARROW_UINT8 || t == AMDARROW_INT8) {
#pragma unroll
      for (int j = 0; j < N; ++j) raw[j] = reinterpret_cast<const uint8_t*>(d)[i[j]];
    } else if (t == AMDARROW_UINT16 || t == AMDARROW_INT16) {
#pragma unroll
      for (int j = 0; j < N; ++j) raw[j] = reinterpret_cast<const uint16_t*>(d)[i[j]];
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) raw[j] = reinterpret_cast<const uint32_t*>(d)[i[j]];
    }
  }
}
template <typename KeyT> __device__ __forceinline__ KeyT KeyFromRaw(int t, KeyT raw, bool* is_nan);
template <> __device__ __forceinline__ uint32_t KeyFromRaw<uint32_t>(int t, uint32_t raw, bool* is_nan) {
  *is_nan = false;
  switch (t) {
    case AMDARROW_UINT8: case AMDARROW_UINT16: case AMDARROW_UINT32: return raw;
    case AMDARROW_INT8: return raw ^ 0x80u;
    case AMDARROW_INT16: return raw ^ 0x8000u;
    case AMDARROW_INT32: return raw ^ 0x80000000u;
    default: {  // FLOAT
      float f = __uint_as_float(raw);
      *is_nan = f != f;
      if (f == 0.0f) f = 0.0f;  // -0.0 == +0.0 is a tie
      const uint32_t u = __float_as_uint(f);
      return (u >> 31) ? ~u : (u | 0x80000000u);
    }
  }
}
template <> __device__ __forceinline__ uint64_t KeyFromRaw<uint64_t>(int t, uint64_t raw, bool* is_nan) {
  *is_nan = false;
  switch (t) {
    case AMDARROW_UINT64: return raw;
    case AMDARROW_INT64: return raw ^ 0x8000000000000000ULL;
    default: {  // DOUBLE
      double f = __longlong_as_double(static_cast<long long>(raw));
      *is_nan = f != f;
      if (f == 0.0) f = 0.0;
      const uint64_t u = static_cast<uint64_t>(__double_as_longlong(f));
      return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    }
  }
}

// class of a row: 0 sortable, 1 NaN, 2 null
template <typename KeyT>
__device__ __forceinline__ int Classify(const SortChunk& c, int type_id, int64_t r, KeyT key_mask, KeyT flip, KeyT* key) {
  if (c.valid && !GetBit(c.valid, c.off + r)) return 2;
  bool is_nan;
  const KeyT k = MakeKey<KeyT>(type_id, c.values, c.off + r, &is_nan);
  *key = (k ^ flip) & key_mask;
  return is_nan ? 1 : 0;
}

// counts[class * ntiles + tile] for class 0..2
// PERM: the rows are visited in the order perm[0..n) (the result of sorting by the later sort
// keys, RadixRecordBatchSorter vector_sort.cc:955-1045) instead of in logical order; tiles then
// cut the positions, not the chunks, and every row looks its chunk up.
template <typename KeyT, bool PERM>
__global__ __launch_bounds__(kThreads) void k_sort_count(const SortChunk* chunks, int nchunks, int type_id, int64_t ntiles,
                                                         const uint64_t* perm, int64_t n, uint32_t* counts) {
  __shared__ uint32_t s_c[3];
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x < 3) s_c[threadIdx.x] = 0;
    __syncthreads();
    SortChunk c;
    int64_t r0;
    if (PERM) { r0 = tile * kCTile; c.length = n; }
    else { c = chunks[FindChunk(chunks, nchunks, tile)]; r0 = (tile - c.tile_start) * kCTile; }
    uint32_t n1 = 0, n2 = 0, n0 = 0;
    if (PERM && nchunks > 1) {
      for (int k = 0; k < kCTile / kThreads; ++k) {
        const int64_t r = r0 + k * kThreads + threadIdx.x;
        if (r < c.length) {
          KeyT key;
          const int64_t row = static_cast<int64_t>(perm[r]);
          const SortChunk rc = chunks[FindChunkOfRow(chunks, nchunks, row)];
          const int cls = Classify<KeyT>(rc, type_id, row - rc.row_start, ~KeyT(0), 0, &key);
          n0 += cls == 0; n1 += cls == 1; n2 += cls == 2;
        }
      }
    } else if (c.length > 0) {
      // the tile's loads (values where a NaN is possible, validity bytes) all issued before the first is looked at
      // (through a permutation over ONE chunk as well: the positions' rows first, then the same)
      constexpr int kI = kCTile / kThreads;
      const bool floats = type_id == AMDARROW_FLOAT || type_id == AMDARROW_DOUBLE;
      int64_t ri[kI];
      bool in[kI];
      KeyT raw[kI];
      uint32_t vb[kI];
      if (PERM) { const int64_t len = c.length; c = chunks[0]; c.length = len; }
#pragma unroll
      for (int k = 0; k < kI; ++k) {
        const int64_t r = r0 + k * kThreads + threadIdx.x;
        in[k] = r < c.length;
        ri[k] = in[k] ? r : c.length - 1;
        if (PERM) ri[k] = static_cast<int64_t>(perm[ri[k]]) - c.row_start;
        raw[k] = 0;
        vb[k] = 0xFFu;
      }
#pragma unroll
      for (int k = 0; k < kI; ++k) ri[k] += c.off;
      if (floats) LoadRawKeys<KeyT, kI>(type_id, c.values, ri, raw);
      if (c.valid) {
#pragma unroll
        for (int k = 0; k < kI; ++k) vb[k] = c.valid[ri[k] >> 3];
      }
#pragma unroll
      for (int k = 0; k < kI; ++k) {
        if (!in[k]) continue;
        bool is_nan = false;
        if (!((vb[k] >> (ri[k] & 7)) & 1u)) { n2 += 1; continue; }
        if (floats) KeyFromRaw<KeyT>(type_id, raw[k], &is_nan);
        n0 += !is_nan; n1 += is_nan;
      }
    }
    n0 = static_cast<uint32_t>(WaveReduceSum64(n0));
    n1 = static_cast<uint32_t>(WaveReduceSum64(n1));
    n2 = static_cast<uint32_t>(WaveReduceSum64(n2));
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_c[0], n0); atomicAdd(&s_c[1], n1); atomicAdd(&s_c[2], n2); }
    __syncthreads();
    if (threadIdx.x < 3) counts[threadIdx.x * ntiles + tile] = s_c[threadIdx.x];
    __syncthreads();
  }
}

// k_sort_count for a column that can hold neither nulls nor NaNs (integers without a validity buffer): every row is
// sortable, the counts are the tiles' row counts -- no pass over the keys (64 us per 100 M uint64).
__global__ __launch_bounds__(kThreads) void k_sort_count_all_sortable(const SortChunk* chunks, int nchunks, int64_t ntiles, uint32_t* counts) {
  const int64_t tile = int64_t(blockIdx.x) * kThreads + threadIdx.x;
  if (tile >= ntiles) return;
  const SortChunk c = chunks[FindChunk(chunks, nchunks, tile)];
  const int64_t left = c.length - (tile - c.tile_start) * kCTile;
  counts[tile] = static_cast<uint32_t>(left >= kCTile ? kCTile : (left > 0 ? left : 0));
  counts[ntiles + tile] = 0;
  counts[2 * ntiles + tile] = 0;
}

// One more key for bucket `d` of an LDS histogram -- for every lane of the wave for which `active` holds.  The lanes
// that agree with the wave's first active lane are counted with ONE atomic when there are many of them: 64 lanes adding
// to one LDS word are served one after the other, and that is what the high digits of small integers, of sorted or of
// clustered keys are -- the same value in every lane (k_sort_hist took 2.1 ms instead of 0.16 for int64 values in
// [0, 1000], the counting phase of a one-sweep pass 0.69 ms instead of 0.50 for sorted keys).  Returns, where the
// caller ranks with the atomic's return value (RANK), the number of keys of bucket d counted before this one: the
// agreeing lanes are numbered in lane order behind their common base, which is the order the ranks have to have.
template <bool RANK>
__device__ __forceinline__ uint32_t WaveHistAdd(uint32_t* h, uint32_t d, bool active) {
  const uint64_t act = __ballot(active);
  uint32_t rank = 0;
  if (act == 0) return rank;  // (wave-uniform)
  const uint32_t first = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(d), __builtin_ctzll(act)));
  const bool same = active && d == first;
  const uint64_t m = __ballot(same);
  const uint32_t agree = static_cast<uint32_t>(__popcll(m));
  if (agree >= 16) {  // (wave-uniform)
    uint32_t base = 0;
    if (LaneId() == __builtin_ctzll(m)) base = atomicAdd(&h[first], agree);
    if (RANK) {
      base = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(base), __builtin_ctzll(m)));
      if (same) rank = base + static_cast<uint32_t>(RankBelowLane(m));
    }
    if (active && !same) rank = atomicAdd(&h[d], 1u);
  } else if (active) {
    rank = atomicAdd(&h[d], 1u);
  }
  return rank;
}

// The digit histograms of a column without nulls or NaNs, straight from its values: what k_sort_partition builds on
// the way, without its copy of the keys and indices -- the first radix pass then reads the column itself (RAW in
// k_radix_onesweep) and numbers the rows as it goes.
// NQ: only the histograms of the TOP NQ digits are built (the LDS atomics are what bounds this kernel: eight per 64-bit
// key take 0.21 ms per 100 M keys); which of the lower digits vary at all falls out of an OR / AND over the keys
// (orand[0] |= key, orand[1] &= key), which is all the hybrid plan needs to know about them -- they are sorted in LDS by
// the window kernel, never by a global pass.  A plan that does need their histograms runs the kernel again with NQ = all.
template <typename KeyT, int NQ>
__global__ __launch_bounds__(kThreads) void k_sort_hist(const void* values, int64_t off, int type_id, int64_t n, KeyT key_mask, KeyT flip,
                                                        unsigned int* ghist, unsigned long long* orand) {
  constexpr int kPasses = sizeof(KeyT);
  constexpr int kFirst = kPasses - NQ;
  __shared__ unsigned int s_h[kPasses * 256];
  for (int i = threadIdx.x; i < kPasses * 256; i += kThreads) s_h[i] = 0;
  __syncthreads();
  KeyT k_or = 0, k_and = ~KeyT(0);
  constexpr int kIters = kCTile / kThreads;
  const int64_t ntiles = (n + kCTile - 1) / kCTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    KeyT key[kIters];
    bool in[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t r = tile * kCTile + j * kThreads + threadIdx.x;
      in[j] = r < n;
      key[j] = reinterpret_cast<const KeyT*>(values)[off + (in[j] ? r : n - 1)];  // (unconditional: all of the tile's loads in flight together)
    }
#pragma unroll
    for (int j = 0; j < kIters; ++j) key[j] = (key[j] ^ flip) & key_mask;  // (flip: sign bit and order, see the caller)
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      if (in[j]) { k_or |= key[j]; k_and &= key[j]; }
      if (tile + 1 < ntiles) {  // a whole tile (wave-uniform): every lane holds a key
        // Where most lanes of the wave hold the same digit value -- the high digits of small integers, of sorted or of
        // clustered keys: all 64 -- they are counted with ONE atomic: 64 lanes adding to one LDS word are served one
        // after the other (int64 values in [0, 1000]: 2.1 ms for this kernel instead of 0.16).
#pragma unroll
        for (int q = kFirst; q < kPasses; ++q) {
          const uint32_t d = static_cast<uint32_t>((key[j] >> (8 * q)) & 255);
          const uint32_t f = __builtin_amdgcn_readfirstlane(d);
          const uint64_t m = __ballot(d == f);
          if (__popcll(m) >= 32) {  // (wave-uniform)
            if (LaneId() == 0) atomicAdd(&s_h[q * 256 + f], static_cast<unsigned int>(__popcll(m)));
            if (d != f) atomicAdd(&s_h[q * 256 + d], 1u);
          } else {
            atomicAdd(&s_h[q * 256 + d], 1u);
          }
        }
      } else {
#pragma unroll
        for (int q = kFirst; q < kPasses; ++q) WaveHistAdd<false>(&s_h[q * 256], static_cast<uint32_t>((key[j] >> (8 * q)) & 255), in[j]);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    k_or |= __shfl_xor(k_or, o);
    k_and &= __shfl_xor(k_and, o);
  }
  if ((threadIdx.x & 63) == 0 && orand) {  // slotted (common.h): same-address device atomics are serialised at ~35 ns each
    unsigned long long* slot = CounterSlot(orand);
    atomicOr(&slot[0], static_cast<unsigned long long>(k_or));                          // bits some key has set
    atomicOr(&slot[1], static_cast<unsigned long long>(~k_and) & static_cast<unsigned long long>(key_mask));  // bits some key has clear
  }
  __syncthreads();
  for (int i = threadIdx.x + kFirst * 256; i < kPasses * 256; i += kThreads)
    if (s_h[i]) atomicAdd(&ghist[i], s_h[i]);
}
struct HistShifts { int n; int shift[4]; };
__device__ __forceinline__ int FindChunkO(const SortChunk* chunks, int nchunks, int64_t otile) {
  int lo = 0, hi = nchunks - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunks[mid].otile_start <= otile) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// ---- the classified raw path (SortRawClassified): nullable / floating-point / chunked columns without a copy ----------
// One workgroup per one-sweep tile (kOTileRows rows of ONE chunk, 512 threads x 16 rows).  MODE 0: the rows' classes are
// counted -- counts[class][tile]: sortable | NaN | null -- and the sortable keys' OR / OR-of-complement collected (which
// bits vary); MODE 1: the histograms of the plan's digits (hs) over the sortable keys.  Loads raw for the whole tile
// before anything is looked at.
constexpr int kRcThreads = 1024;  // (8 rows per thread: 374 -> 260 us per 100 M float64 against 512 threads x 16 rows)
template <typename KeyT, int MODE>
__global__ __launch_bounds__(kRcThreads) void k_rawc_scan(const SortChunk* chunks, int nchunks, int type_id, int64_t notiles, int64_t tile_stride,
                                                          KeyT key_mask, KeyT flip, uint32_t* counts, unsigned long long* orand, HistShifts hs,
                                                          unsigned int* ghist, const HistShifts* hs_dev = nullptr) {
  constexpr bool kCount = (MODE & 1) != 0, kHist = (MODE & 2) != 0;
  if (kHist && hs_dev) hs = *hs_dev;  // (decided by k_guess_shifts, queued in front of this kernel)
  constexpr int kIters = kOTileRows / kRcThreads;  // 16
  constexpr int kWaves = kRcThreads / 64;
  __shared__ unsigned int s_h[kHist ? 4 * 256 : 1];
  __shared__ uint32_t s_cls[2][3][kWaves];  // double-buffered by tile parity: one barrier per tile
  if (kHist) { for (int i = threadIdx.x; i < 4 * 256; i += kRcThreads) s_h[i] = 0; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  KeyT k_or = 0, k_orn = 0;
  // Two tiles' loads in flight: the next tile's keys and validity bytes are requested before this tile is classified
  // (a tile at a time, every tile began with the round trips of its chunk record and its keys in a row).  The chunk
  // record is looked up again only when a tile leaves the chunk of the tile before it.
  SortChunk c_a{}, c_b{};
  KeyT key_a[kIters], key_b[kIters];
  uint32_t vb_a[kIters], vb_b[kIters];
  int chunk_at = -1;
  SortChunk chunk_now{};
  auto load = [&](int64_t tile, SortChunk& c, KeyT (&key)[kIters], uint32_t (&vb)[kIters]) {
    if (chunk_at < 0 || tile < chunk_now.otile_start || (chunk_at + 1 < nchunks && tile >= chunks[chunk_at + 1].otile_start)) {
      chunk_at = FindChunkO(chunks, nchunks, tile);
      chunk_now = chunks[chunk_at];
    }
    c = chunk_now;
    const int64_t r0 = (tile - c.otile_start) * kOTileRows;
    int64_t ri[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t r = r0 + j * kRcThreads + threadIdx.x;
      ri[j] = c.off + (r < c.length ? r : c.length - 1);
      vb[j] = 0xFFu;
    }
    LoadRawKeys<KeyT, kIters>(type_id, c.values, ri, key);
    if (c.valid) {
#pragma unroll
      for (int j = 0; j < kIters; ++j) vb[j] = c.valid[ri[j] >> 3];
    }
  };
  auto process = [&](int64_t tile, const SortChunk& c, const KeyT (&key)[kIters], const uint32_t (&vb)[kIters], int parity) {
    const int64_t r0 = (tile - c.otile_start) * kOTileRows;
    uint32_t tot[3] = {0, 0, 0};  // (per thread; reduced once per tile)
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t r = r0 + j * kRcThreads + threadIdx.x;
      const int64_t ri = c.off + (r < c.length ? r : c.length - 1);
      int cls = 3;
      KeyT k = 0;
      if (r < c.length) {
        if (!((vb[j] >> (ri & 7)) & 1u)) cls = 2;
        else {
          bool is_nan;
          k = (KeyFromRaw<KeyT>(type_id, key[j], &is_nan) ^ flip) & key_mask;
          cls = is_nan ? 1 : 0;
        }
      }
      if (kCount) {
        tot[1] += cls == 1; tot[2] += cls == 2;
        if (cls == 0) { ++tot[0]; k_or |= k; k_orn |= ~k & key_mask; }
      }
      if (kHist && cls == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (q < hs.n) atomicAdd(&s_h[q * 256 + ((k >> hs.shift[q]) & 255)], 1u);
      }
    }
    if (kCount) {
      unsigned long long packed = tot[0] | static_cast<unsigned long long>(tot[1]) << 16 | static_cast<unsigned long long>(tot[2]) << 32;  // (<= 1024 per wave each)
      packed = WaveReduceSum64(packed);
      if (lane == 0) { s_cls[parity][0][wave] = packed & 0xFFFFu; s_cls[parity][1][wave] = (packed >> 16) & 0xFFFFu; s_cls[parity][2][wave] = static_cast<uint32_t>(packed >> 32); }
      __syncthreads();
      if (threadIdx.x < 3) {
        uint32_t t = 0;
        for (int w = 0; w < kWaves; ++w) t += s_cls[parity][threadIdx.x][w];
        counts[threadIdx.x * notiles + tile] = t;
      }
    }
  };
  {
    const int64_t step = int64_t(gridDim.x) * tile_stride;
    int64_t t0 = int64_t(blockIdx.x) * tile_stride;
    if (t0 < notiles) {
      load(t0, c_a, key_a, vb_a);
      for (;;) {
        const int64_t t1 = t0 + step;
        if (t1 < notiles) load(t1, c_b, key_b, vb_b);
        process(t0, c_a, key_a, vb_a, 0);
        if (t1 >= notiles) break;
        t0 = t1 + step;
        if (t0 < notiles) load(t0, c_a, key_a, vb_a);
        process(t1, c_b, key_b, vb_b, 1);
        if (t0 >= notiles) break;
      }
    }
  }
  if (kCount) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { k_or |= __shfl_xor(k_or, o); k_orn |= __shfl_xor(k_orn, o); }
    if (lane == 0) {
      unsigned long long* slot = CounterSlot(orand);
      atomicOr(&slot[0], static_cast<unsigned long long>(k_or));
      atomicOr(&slot[1], static_cast<unsigned long long>(k_orn));
    }
  }
  if (kHist) {
    __syncthreads();
    for (int i = threadIdx.x; i < hs.n * 256; i += kRcThreads)
      if (s_h[i]) atomicAdd(&ghist[i], s_h[i]);
  }
}

// Histograms of up to four 8-bit digits at ARBITRARY bit positions of the (already transformed) keys: the hybrid plan
// aligns its top digit with the highest bit that varies among the keys instead of with a byte boundary (see SortImpl).
template <typename KeyT>
__global__ __launch_bounds__(kThreads) void k_hist_shifts(const KeyT* keys, int64_t n, HistShifts hs, unsigned int* ghist /* [4][256] */) {
  __shared__ unsigned int s_h[4 * 256];
  for (int i = threadIdx.x; i < 4 * 256; i += kThreads) s_h[i] = 0;
  __syncthreads();
  constexpr int kIters = kCTile / kThreads;
  const int64_t ntiles = (n + kCTile - 1) / kCTile;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    KeyT key[kIters];
    bool in[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t r = tile * kCTile + j * kThreads + threadIdx.x;
      in[j] = r < n;
      key[j] = keys[in[j] ? r : n - 1];
    }
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (q < hs.n) WaveHistAdd<false>(&s_h[q * 256], static_cast<uint32_t>((key[j] >> hs.shift[q]) & 255), in[j]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hs.n * 256; i += kThreads)
    if (s_h[i]) atomicAdd(&ghist[i], s_h[i]);
}
__global__ __launch_bounds__(kThreads) void k_iota_indices(int64_t n, uint64_t* out) {
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads) out[i] = static_cast<uint64_t>(i);
}

// Stable 3-way partition.  offsets = exclusive scan of counts laid out [class][tile]: it is
// directly the global output position (sortable rows first, then NaNs, then nulls).
template <typename KeyT, bool PERM>
__global__ __launch_bounds__(kThreads) void k_sort_partition(const SortChunk* chunks, int nchunks, int type_id, int64_t ntiles,
                                                             const uint64_t* perm, int64_t n,
                                                             const uint32_t* offsets, KeyT key_mask, KeyT flip, KeyT* keys,
                                                             uint32_t* idx, uint64_t* out_indices, unsigned int* ghist) {
  // the digit histograms of all passes ride along (the keys are in registers anyway): hist[pass * 256 + digit]
  constexpr int kPasses = sizeof(KeyT);
  __shared__ unsigned int s_h[kPasses * 256];
  for (int i = threadIdx.x; i < kPasses * 256; i += kThreads) s_h[i] = 0;
  __shared__ uint32_t s_wave[3][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kIters = kCTile / kThreads;  // 8; wave w owns rows [w*512, (w+1)*512) of the tile, in order
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    SortChunk c;
    int64_t r0;
    if (PERM) { r0 = tile * kCTile + wave * (kIters * 64); c.length = n; c.row_start = 0; }
    else { c = chunks[FindChunk(chunks, nchunks, tile)]; r0 = (tile - c.tile_start) * kCTile + wave * (kIters * 64); }
    KeyT key[kIters];
    int cls[kIters];
    uint64_t rows[PERM ? kIters : 1];
    uint32_t tot[3] = {0, 0, 0};
    if (PERM) {
#pragma unroll
      for (int j = 0; j < kIters; ++j) {  // the gathers below depend on these: issue them together
        const int64_t r = r0 + j * 64 + lane;
        rows[j] = perm[r < n ? r : n - 1];
      }
    }
    // (visiting the rows through a permutation: one chunk -- the usual case -- takes the batched loads too, the rows
    // being its rows; several chunks look every row's chunk up, below)
    const bool batched = PERM ? nchunks == 1 : c.length > 0;
    if (PERM && batched) { const int64_t len = c.length; c = chunks[0]; c.length = len; }
    if (batched) {
      // the tile's loads (values, validity bytes) all issued before the first is looked at: with the classification
      // inside each row's bounds check every load was waited for before the next was issued
      int64_t ri[kIters];
      bool in[kIters];
      uint32_t vb[kIters];
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        const int64_t r = r0 + j * 64 + lane;
        in[j] = r < c.length;
        if (PERM) ri[j] = c.off + (static_cast<int64_t>(rows[PERM ? j : 0]) - c.row_start);  // (rows[] past the end repeat the last row)
        else ri[j] = c.off + (in[j] ? r : c.length - 1);
        vb[j] = 0xFFu;
      }
      LoadRawKeys<KeyT, kIters>(type_id, c.values, ri, key);
      if (c.valid) {
#pragma unroll
        for (int j = 0; j < kIters; ++j) vb[j] = c.valid[ri[j] >> 3];
      }
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        cls[j] = 3;
        if (in[j]) {
          if (!((vb[j] >> (ri[j] & 7)) & 1u)) { cls[j] = 2; key[j] = 0; }
          else {
            bool is_nan;
            key[j] = (KeyFromRaw<KeyT>(type_id, key[j], &is_nan) ^ flip) & key_mask;
            cls[j] = is_nan ? 1 : 0;
          }
        } else {
          key[j] = 0;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t r = r0 + j * 64 + lane;
      if (PERM && !batched) {
        cls[j] = 3;
        key[j] = 0;
        if (r < n) {
          const int64_t row = static_cast<int64_t>(rows[j]);
          const SortChunk rc = chunks[FindChunkOfRow(chunks, nchunks, row)];
          cls[j] = Classify<KeyT>(rc, type_id, row - rc.row_start, key_mask, flip, &key[j]);
        }
      } else if (!batched) {
        cls[j] = 3;
        key[j] = 0;
      }
      tot[0] += __popcll(__ballot(cls[j] == 0));
      tot[1] += __popcll(__ballot(cls[j] == 1));
      tot[2] += __popcll(__ballot(cls[j] == 2));
    }
    if (lane == 0) { s_wave[0][wave] = tot[0]; s_wave[1][wave] = tot[1]; s_wave[2][wave] = tot[2]; }
    __syncthreads();
    int64_t base[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint32_t w_off = 0;
      for (int w = 0; w < wave; ++w) w_off += s_wave[k][w];
      base[k] = int64_t(offsets[k * ntiles + tile]) + w_off;
    }
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int64_t r = r0 + j * 64 + lane;
      const uint64_t m0 = __ballot(cls[j] == 0), m1 = __ballot(cls[j] == 1), m2 = __ballot(cls[j] == 2);
      const uint64_t gidx = PERM ? rows[PERM ? j : 0] : static_cast<uint64_t>(c.row_start + r);
      if (cls[j] == 0) {
        const int64_t p = base[0] + RankBelowLane(m0);
        keys[p] = key[j];
        idx[p] = static_cast<uint32_t>(gidx);
      } else if (cls[j] == 1) {
        out_indices[base[1] + RankBelowLane(m1)] = gidx;
      } else if (cls[j] == 2) {
        out_indices[base[2] + RankBelowLane(m2)] = gidx;
      }
#pragma unroll
      for (int q = 0; q < kPasses; ++q) WaveHistAdd<false>(&s_h[q * 256], static_cast<uint32_t>((key[j] >> (8 * q)) & 255), cls[j] == 0);
      base[0] += __popcll(m0); base[1] += __popcll(m1); base[2] += __popcll(m2);
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < kPasses * 256; i += kThreads)
    if (s_h[i]) atomicAdd(&ghist[i], s_h[i]);
}

// block_hist[digit * nblocks + block]
template <typename KeyT>
__global__ __launch_bounds__(kThreads) void k_radix_block_hist(const KeyT* keys, int64_t n, int shift, int64_t nblocks,
                                                               uint32_t* block_hist) {
  __shared__ unsigned int s_h[256];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * kRTile;
#pragma unroll 4
  for (int k = 0; k < kRTile / kThreads; ++k) {
    const int64_t i = base + k * kThreads + threadIdx.x;
    if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & 255], 1u);
  }
  __syncthreads();
  block_hist[int64_t(threadIdx.x) * nblocks + blockIdx.x] = s_h[threadIdx.x];
}

// Device-wide exclusive scan of u32 (in place allowed): block sums -> single-block scan -> apply.
__global__ __launch_bounds__(kThreads) void k_scan_block_sums(const uint32_t* in, int64_t n, uint32_t* sums) {
  __shared__ unsigned int s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * 2048;
  uint32_t c = 0;
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + k * kThreads + threadIdx.x;
    if (i < n) c += in[i];
  }
  c = static_cast<uint32_t>(WaveReduceSum64(c));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_sum, c);
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = s_sum;
}
__global__ __launch_bounds__(kThreads) void k_scan_apply(const uint32_t* in, int64_t n, const int64_t* block_offsets,
                                                         uint32_t* out) {
  __shared__ uint32_t s_wave[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = int64_t(blockIdx.x) * 2048;
  uint32_t carry = static_cast<uint32_t>(block_offsets[blockIdx.x]);
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + k * kThreads + threadIdx.x;
    const uint32_t v = i < n ? in[i] : 0;
    const uint32_t incl = WaveInclusiveScan(v);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0, total = 0;
    for (int j = 0; j < 4; ++j) { if (j < wave) woff += s_wave[j]; total += s_wave[j]; }
    if (i < n) out[i] = carry + woff + incl - v;
    carry += total;
    __syncthreads();
  }
}

// Stable scatter of one digit.  LAST: write the final uint64 indices instead of (key, idx).
template <typename KeyT, bool LAST>
__global__ __launch_bounds__(kThreads) void k_radix_scatter(const KeyT* keys_in, const uint32_t* idx_in, int64_t n, int shift,
                                                            int64_t nblocks, const uint32_t* block_offsets, KeyT* keys_out,
                                                            uint32_t* idx_out, uint64_t* out64) {
  constexpr int kIters = kRTile / kThreads;  // 16; wave w owns keys [w*1024, (w+1)*1024) of the tile, in order
  __shared__ uint32_t s_whist[4][256];   // per-wave digit counts, then per-wave running local positions
  __shared__ uint32_t s_dstart[256];     // first local position of each digit
  __shared__ uint32_t s_gbase[256];      // global position of this block's run of each digit
  __shared__ uint32_t s_scan[4];
  __shared__ KeyT s_keys[kRTile];
  __shared__ uint32_t s_idx[kRTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = int64_t(blockIdx.x) * kRTile;
  const int64_t left = n - base;
  const uint32_t count = left >= kRTile ? kRTile : static_cast<uint32_t>(left);

  for (int w = 0; w < 4; ++w) s_whist[w][tid] = 0;
  __syncthreads();
  KeyT key[kIters];
  uint32_t idx[kIters];
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    key[j] = 0; idx[j] = 0;
    if (li < count) {
      key[j] = keys_in[base + li];
      idx[j] = idx_in[base + li];
      atomicAdd(&s_whist[wave][(key[j] >> shift) & 255], 1u);
    }
  }
  __syncthreads();
  {  // thread d: totals of digit d, exclusive scan over digits, per-wave bases
    const uint32_t c0 = s_whist[0][tid], c1 = s_whist[1][tid], c2 = s_whist[2][tid], c3 = s_whist[3][tid];
    const uint32_t tot = c0 + c1 + c2 + c3;
    const uint32_t incl = WaveInclusiveScan(tot);
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_scan[w];
    const uint32_t dstart = woff + incl - tot;
    s_dstart[tid] = dstart;
    s_gbase[tid] = block_offsets[int64_t(tid) * nblocks + blockIdx.x];
    s_whist[0][tid] = dstart;
    s_whist[1][tid] = dstart + c0;
    s_whist[2][tid] = dstart + c0 + c1;
    s_whist[3][tid] = dstart + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    const bool live = li < count;
    const uint32_t d = (key[j] >> shift) & 255;
    // match-any on the 8 digit bits among live lanes
    uint64_t m = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bb = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bb : ~bb;
    }
    if (live) {
      const uint32_t rank = RankBelowLane(m);
      const uint32_t old = s_whist[wave][d];
      __builtin_amdgcn_wave_barrier();
      if (rank == 0) s_whist[wave][d] = old + __popcll(m);
      const uint32_t lpos = old + rank;
      s_keys[lpos] = key[j];
      s_idx[lpos] = idx[j];
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  for (uint32_t i = tid; i < count; i += kThreads) {
    const KeyT k = s_keys[i];
    const uint32_t d = (k >> shift) & 255;
    const uint32_t gpos = s_gbase[d] + (i - s_dstart[d]);
    if (LAST) {
      out64[gpos] = s_idx[i];
    } else {
      keys_out[gpos] = k;
      idx_out[gpos] = s_idx[i];
    }
  }
}

// ---- one-sweep pass ---------------------------------------------------------------------------------
// One kernel per digit: a block takes the next tile (ticket), counts its digits, publishes the
// counts, ranks and stages its keys in LDS, and only then needs the number of equal digits in
// ALL earlier tiles -- obtained by decoupled look-back over the published counts (chained scan:
// a tile adds predecessors' aggregates until it meets one whose inclusive prefix is known).
// The keys are read once per pass; the separate block-histogram read and the device-wide scan of
// 256 x nblocks counters of the three-kernel variant are gone.
// lookback word: bits 31..30 = 0 nothing yet, 1 aggregate of this tile, 2 inclusive prefix;
// bits 29..0 = count (n < 2^30).  Tile = workgroup id: workgroups are dispatched in id order, so
// every tile a block waits for is already running.  (An atomic ticket would make that a
// guarantee, but one same-address device-scope atomic per tile is serialised at ~40 ns -- the
// whole pass then runs at ticket speed: 1.02 ms for 24 k tiles, 3.5 ms for 49 k.)  The wait is
// bounded by kSpinLimit; if it ever expires the pass reports it and the host redoes the sort
// with the three-kernel passes, which need no inter-block ordering.
constexpr uint32_t kLbAggregate = 1u << 30, kLbInclusive = 2u << 30, kLbValue = (1u << 30) - 1;
constexpr int kLbWindow = 8;  // (16 / 32 words per round trip, or the first window's loads issued ahead of the key staging: slower -- the wait is for predecessors that have not counted yet, not for round trips)
constexpr int kSpinLimit = 1 << 18;  // polls of one predecessor word (~0.3 s) before the pass gives up

// dbase[pass][d] = number of keys whose digit `pass` is < d (exclusive scan of the global histogram)
__global__ __launch_bounds__(kThreads) void k_digit_bases(const unsigned int* ghist, uint32_t* dbase) {
  __shared__ uint32_t s_scan[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t c = ghist[blockIdx.x * 256 + tid];
  const uint32_t incl = WaveInclusiveScan(c);
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += s_scan[w];
  dbase[blockIdx.x * 256 + tid] = woff + incl - c;
}

// 8192 keys per tile and 512 threads: a digit gets a run of 32 keys (256 bytes of keys, 128 of indices) instead of
// 16 from a 4096-key tile of 256 threads -- 671 -> 595 us per pass; 8192 keys on 256 threads: 705 us.
constexpr int kOThreads = ONESWEEP_THREADS;  // one-sweep block
constexpr int kOTile = ONESWEEP_TILE;        // keys per one-sweep tile (<= 2 * kRTile: the look-back array is sized by kRTile tiles)
static_assert(kOTile >= kRTile, "look-back array");
static_assert(kOThreads >= 256 && kOTile / kOThreads == 16, "digit threads; sixteen keys per thread");
constexpr int kOWaves = kOThreads / 64;

// ARANK: a key's rank among the keys of its wave with the same digit is what the LDS atomic that
// counts it returns -- the hardware serves the lanes of one ds_add_rtn in ascending lane order and a
// wave's LDS operations in program order, so the returned counts follow (step, lane) order, which
// is the stable order.  That ordering is not an architectural promise: LdsAtomicsLaneOrdered()
// checks it on the device once per context, and the ballot-based match-any ranking (!ARANK, ~10x
// the instructions) stays as the variant for a device that fails the check.
// RAW: the first pass over a column without nulls or NaNs -- keys_in is the column itself (the key transformation is
// applied here: raw_type, raw_flip, raw_mask), the row numbers are the positions.
// RAW == 2: the first pass over a column that may hold nulls / NaNs and may come in chunks (SortRawClassified): a tile
// is kOTile ROWS of one chunk; the rows are classified here -- nulls and NaNs go straight to their final places at the
// end of out64 (their positions come from the scanned class counts, rc_offsets[class][tile]), the sortable keys take
// part in the pass; the row numbers are the logical positions.
struct RawcArgs { const SortChunk* chunks; int nchunks; const uint32_t* offsets; int64_t notiles; };
#ifdef WS_PHASES  // -DWS_PHASES: cycles per phase, summed over the workgroups (profiles/bench_src/ws_phases.py reads them)
__device__ unsigned long long g_os_phase[64][16];
#define OPH(i) do { if (threadIdx.x == 0) { const unsigned long long _n = __builtin_readcyclecounter(); oph[i] += _n - oph_last; oph_last = _n; } } while (0)
#define OPH_FLUSH() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 12; ++_i) atomicAdd(&g_os_phase[blockIdx.x & 63][_i], oph[_i]); atomicAdd(&g_os_phase[blockIdx.x & 63][15], 1ull); } } while (0)
#else
#define OPH(i)
#define OPH_FLUSH()
#endif
template <typename KeyT, bool LAST, bool ARANK, int RAW = 0>
__global__ __launch_bounds__(kOThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_radix_onesweep(const KeyT* keys_in, const uint32_t* idx_in, int64_t n, int shift,
                                                              const uint32_t* dbase /* [256] of this pass */,
                                                              uint32_t* lookback /* [nblocks][256], zeroed */,
                                                              unsigned int* error, KeyT* keys_out,
                                                              uint32_t* idx_out, uint64_t* out64, int raw_type = 0,
                                                              KeyT raw_flip = 0, KeyT raw_mask = 0, uint32_t xcd_run = 0,
                                                              RawcArgs rc = RawcArgs{nullptr, 0, nullptr, 0}) {
  constexpr int kIters = kOTile / kOThreads;  // 16; wave w owns keys [w*1024, (w+1)*1024) of the tile, in order
  __shared__ uint32_t s_whist[kOWaves][256];  // per-wave digit counts, then per-wave running local positions
  __shared__ uint32_t s_dstart[256];          // first local position of each digit
  __shared__ uint32_t s_gbase[256];           // global position of this tile's run of each digit
  __shared__ uint32_t s_scan[4];
  // ONE staging buffer of 8-byte words, used twice: first the keys go through it (LDS position -> global position), then
  // the row indices -- each with the place in the tile it came from in its high half, for the order check below.  With
  // a buffer for each the kernel needed 137 KB of LDS, one workgroup per CU, and the CU sat idle in memory while the
  // tile was ranked and idle in LDS while it streamed; 73 KB lets two workgroups share a CU and overlap.
  __shared__ unsigned long long s_buf[kOTile];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef WS_PHASES
  unsigned long long oph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long oph_last = __builtin_readcyclecounter();
#endif
  for (int i = tid; i < kOWaves * 256; i += kOThreads) (&s_whist[0][0])[i] = 0;
  __syncthreads();
  // Tile of this workgroup.  Workgroups go to the eight XCDs round robin, each XCD with its own L2, and the
  // runs two neighbouring tiles write for a digit share the cache line where they meet: with tile = blockIdx
  // both halves of that line leave as partial writes from two different L2s.  kXcdRun consecutive tiles are
  // therefore handed to the SAME XCD (inside every group of 8 * kXcdRun workgroups), so that the halves meet in
  // one L2.  A tile's predecessors are dispatched at most one such group later, which the look-back tolerates
  // (the blocks of a group never wait for a later group).
  // The permutation makes a tile wait, in the look-back, for tiles whose workgroups are dispatched up to a group
  // LATER than its own: that is only safe while a whole group (8 * xcd_run workgroups) is resident at once, so the
  // host passes xcd_run = 0 (tile = workgroup id: every predecessor was dispatched earlier) unless the occupancy
  // query says the device holds a group, and the device has the eight XCDs the runs are dealt to.
  uint32_t tile = blockIdx.x;
  if (xcd_run) {
    const uint32_t group = 8 * xcd_run;
    const uint32_t g0 = tile / group * group, in = tile - g0;
    if (g0 + group <= gridDim.x) tile = g0 + (in % 8) * xcd_run + in / 8;  // (the ragged last group keeps tile = blockIdx)
  }
  const int64_t base = int64_t(tile) * kOTile;
  const int64_t left = n - base;
  uint32_t count = RAW == 2 ? 0u : (left >= kOTile ? kOTile : static_cast<uint32_t>(left));  // RAW == 2: the tile's ROWS, set below

  KeyT key[kIters];
  uint32_t idx[kIters];
  uint32_t alive = 0;  // bit j: key[j] takes part in the pass
  __shared__ uint32_t s_tail[2][kOWaves];  // RAW == 2: NaN / null rows per wave
  uint32_t tail_cls = 0, tail_rank[RAW == 2 ? kIters : 1];  // RAW == 2: this thread's NaN / null rows (written behind the first barrier)
  if constexpr (RAW == 2) {
    const SortChunk c = rc.chunks[FindChunkO(rc.chunks, rc.nchunks, tile)];
    const int64_t r0 = (int64_t(tile) - c.otile_start) * kOTile;
    const int64_t rows_left = c.length - r0;
    count = rows_left >= kOTile ? kOTile : static_cast<uint32_t>(rows_left > 0 ? rows_left : 0);
    int64_t ri[kIters];
    uint32_t vb[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
      ri[j] = c.off + r0 + (li < count ? li : (count ? count - 1 : 0));
      vb[j] = 0xFFu;
      idx[j] = static_cast<uint32_t>(c.row_start + r0 + li);
    }
    LoadRawKeys<KeyT, kIters>(raw_type, c.values, ri, key);
    if (c.valid) {
#pragma unroll
      for (int j = 0; j < kIters; ++j) vb[j] = c.valid[ri[j] >> 3];
    }
    // classes, and the NaN / null rows' ranks inside the tile (row order = (wave, step, lane) order)
    uint32_t run[2] = {0, 0};
    uint32_t trank[kIters];
    uint32_t tcls = 0;  // two bits per step: 0 sortable / dead, 1 NaN, 2 null
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
      int cls = 3;
      if (li < count) {
        if (!((vb[j] >> (ri[j] & 7)) & 1u)) cls = 2;
        else {
          bool is_nan;
          key[j] = (KeyFromRaw<KeyT>(raw_type, key[j], &is_nan) ^ raw_flip) & raw_mask;
          cls = is_nan ? 1 : 0;
        }
      }
      if (cls == 0) alive |= 1u << j; else key[j] = 0;
      const uint64_t m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
      trank[j] = cls == 1 ? run[0] + RankBelowLane(m1) : cls == 2 ? run[1] + RankBelowLane(m2) : 0u;
      if (cls == 1 || cls == 2) tcls |= static_cast<uint32_t>(cls) << (2 * j);
      run[0] += __popcll(m1);
      run[1] += __popcll(m2);
    }
    if (lane == 0) { s_tail[0][wave] = run[0]; s_tail[1][wave] = run[1]; }
    tail_cls = tcls;
#pragma unroll
    for (int j = 0; j < kIters; ++j) tail_rank[RAW == 2 ? j : 0] = trank[j];
  } else {
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
      key[j] = 0; idx[j] = 0;
      if (li < count) {
        alive |= 1u << j;
        key[j] = keys_in[base + li];
        idx[j] = RAW ? static_cast<uint32_t>(base + li) : idx_in[base + li];
      }
    }
    if (RAW) {  // integers as wide as the key: the transformation is one XOR (sign bit and order) and the mask -- behind
                // the loop, so that no load is waited for before the next one is issued
#pragma unroll
      for (int j = 0; j < kIters; ++j) key[j] = (key[j] ^ raw_flip) & raw_mask;
    }
  }
  uint32_t rk[ARANK ? kIters : 1];
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    // (one atomic per lane also where most lanes agree -- sorted input: 0.69 ms per pass instead of 0.50 --: the wave-wide
    // variant, WaveHistAdd<true>, cost this kernel its registers and every input 0.1 ms per pass)
    if ((alive >> j) & 1u) {
      const uint32_t r = atomicAdd(&s_whist[wave][(key[j] >> shift) & 255], 1u);
      if (ARANK) rk[ARANK ? j : 0] = r;
    }
  }
  __syncthreads();
  OPH(0);  // loads + counting atomics
  if constexpr (RAW == 2) {  // the NaN / null rows go to their final places (tile order, row order inside the tile)
    uint32_t any_tail = 0;
#pragma unroll
    for (int w = 0; w < kOWaves; ++w) any_tail |= s_tail[0][w] | s_tail[1][w];
    if (__builtin_amdgcn_readfirstlane(any_tail)) {
      uint32_t woff[2] = {0, 0};
      for (int w = 0; w < wave; ++w) { woff[0] += s_tail[0][w]; woff[1] += s_tail[1][w]; }
      const uint32_t tail_base[2] = {rc.offsets[1 * rc.notiles + tile], rc.offsets[2 * rc.notiles + tile]};
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        const uint32_t cls = (tail_cls >> (2 * j)) & 3u;
        if (cls) out64[tail_base[cls - 1] + woff[cls - 1] + tail_rank[RAW == 2 ? j : 0]] = idx[j];
      }
    }
  }
  uint32_t tot = 0;
  if (tid < 256) {  // thread d: totals of digit d -> publish; exclusive scan over digits, per-wave bases
    uint32_t c[kOWaves];
#pragma unroll
    for (int w = 0; w < kOWaves; ++w) { c[w] = s_whist[w][tid]; tot += c[w]; }
    __hip_atomic_store(&lookback[int64_t(tile) * 256 + tid], tot | (tile == 0 ? kLbInclusive : kLbAggregate), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t incl = WaveInclusiveScan(tot);
    if (lane == 63) s_scan[wave] = incl;
    // only waves 0..3 are in here: a named barrier would do; the block barrier below is shared with the rest
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < kOWaves; ++w) { const uint32_t cw = c[w]; c[w] = run; run += cw; }
    // positions relative to the digit start; the start itself is added after the barrier
#pragma unroll
    for (int w = 0; w < kOWaves; ++w) s_whist[w][tid] = c[w];
    s_dstart[tid] = incl - tot;  // within this wave of digits; completed below
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_scan[w];
    const uint32_t dstart = s_dstart[tid] + woff;
    s_dstart[tid] = dstart;
#pragma unroll
    for (int w = 0; w < kOWaves; ++w) s_whist[w][tid] += dstart;
  }
  __syncthreads();
  OPH(1);  // digit totals, publish, scan
  uint32_t lpos[kIters];  // where each of this thread's keys is staged (both staging rounds use it)
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const bool live = (alive >> j) & 1u;
    const uint32_t d = (key[j] >> shift) & 255;
    lpos[j] = 0;
    if (ARANK) {
      if (live) {
        lpos[j] = s_whist[wave][d] + rk[ARANK ? j : 0];
        s_buf[lpos[j]] = static_cast<unsigned long long>(key[j]);
      }
      continue;
    }
    // match-any on the 8 digit bits among live lanes
    uint64_t m = __ballot(live);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const uint64_t bb = __ballot((d >> b) & 1);
      m &= ((d >> b) & 1) ? bb : ~bb;
    }
    if (live) {
      const uint32_t rank = RankBelowLane(m);
      const uint32_t old = s_whist[wave][d];
      __builtin_amdgcn_wave_barrier();
      if (rank == 0) s_whist[wave][d] = old + __popcll(m);
      lpos[j] = old + rank;
      s_buf[lpos[j]] = static_cast<unsigned long long>(key[j]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  OPH(2);  // keys staged
  // The tile's keys and row indices go through the staging buffer BEFORE the look-back is waited for: what the look-back
  // yields -- the global place of each digit's run -- is only needed by the stores, and the predecessors it waits for
  // publish their counts while this tile is busy in LDS.  (With the look-back between the two staging rounds a tile sat
  // idle for a fifth of its time: its predecessors in tile order are dispatched up to a group of workgroups later.)
  if (RAW == 2) count = s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];  // the keys that took part (the digit totals' sum)
  // round 1: the keys in digit order (thread t takes the staged positions t, t + 512, ...: the same ones in both rounds)
  KeyT skey[kIters];
  uint32_t inrun = 0;     // bit k: position i is not the first of its digit's run (the order check compares it with i - 1)
#pragma unroll
  for (int k = 0; k < kIters; ++k) {
    const uint32_t i = k * kOThreads + tid;
    skey[k] = 0;
    if (i < count) {
      skey[k] = static_cast<KeyT>(s_buf[i]);
      if (i > s_dstart[(skey[k] >> shift) & 255]) inrun |= 1u << k;
    }
  }
  __syncthreads();
  OPH(5);  // round 1: keys back in digit order
  // round 2: the row indices, each with its origin in the tile
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    if ((alive >> j) & 1u) s_buf[lpos[j]] = static_cast<unsigned long long>(li) << 32 | idx[j];
  }
  __syncthreads();
  OPH(6);  // row indices staged
  bool disorder = false;
  uint32_t sidx[kIters];
#pragma unroll
  for (int k = 0; k < kIters; ++k) {  // (every lane takes part in the lane shift below)
    const uint32_t i = k * kOThreads + tid;
    const bool in = i < count;
    const unsigned long long pair = s_buf[in ? i : 0];
    sidx[k] = static_cast<uint32_t>(pair);
    // ARANK is VERIFIED, not trusted: inside a digit's run the staged keys must appear in the order they had in the
    // tile (that is what "stable" means for this pass; the atomics hand out distinct ranks by construction, so the
    // only thing that could go wrong is their order).  A tile that fails raises error word 2 and the host redoes the
    // sort with the ballot ranking -- correctness never rests on the order in which the hardware serves the lanes of
    // one ds_add_rtn, only the speed does.
    if (ARANK) {
      // the predecessor's origin: from the lane below (a DPP wave shift, no LDS traffic), lane 0 of a wave from LDS
      const uint32_t from = static_cast<uint32_t>(pair >> 32);
      uint32_t prev = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(from), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
      if (lane == 0 && i > 0) prev = static_cast<uint32_t>(s_buf[in ? i - 1 : 0] >> 32);
      if (in && ((inrun >> k) & 1u)) disorder |= prev > from;
    }
  }
  if (ARANK && disorder) __hip_atomic_store(error + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool gave_up = false;
  if (tid < 256) {  // thread d: keys with digit d in all earlier tiles
    uint32_t excl = 0;
    if (tile > 0) {
      // A window of kLbWindow predecessors per round trip: the words of one round are loaded
      // back to back and consumed nearest-first; the walk stops at the first inclusive word and
      // re-polls from the first word that is not published yet.
      // (Measured and dropped: 16 / 32 words per round trip; the first window's loads issued ahead of the staging
      // rounds; two levels -- a tile adds up its predecessors inside its group of sixteen, then walks over group words
      // published by each group's last tile: 526 / 487 -> 601 / 565 us per pass, the extra hop through a tile that is
      // dispatched last in its group costs more than the shorter walk saves; group words that every tile ADDS its count
      // to with one atomic, no hop: 544 / 500 us, the same as this flat walk.  The wait is not the walk: a tile cannot
      // finish before the SLOWEST of the ~460 tiles in flight in front of it has counted.)
      int64_t j = int64_t(tile) - 1;
      int spins = 0;
      for (bool done = false; !done;) {
        uint32_t v[kLbWindow];
#pragma unroll
        for (int s = 0; s < kLbWindow; ++s) {
          const int64_t t = j - s;
          v[s] = t >= 0 ? __hip_atomic_load(&lookback[t * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kLbInclusive;
        }
        int consumed = kLbWindow;
#pragma unroll
        for (int s = 0; s < kLbWindow; ++s) {
          if (consumed != kLbWindow) continue;         // stopped at an earlier word of this window
          if ((v[s] >> 30) == 0) { consumed = s; continue; }  // not published yet: poll again from here
          excl += v[s] & kLbValue;
          if ((v[s] >> 30) == 2) { done = true; consumed = s + 1; }
        }
        j -= consumed;
        if (!done && consumed != kLbWindow) {
          // never in a correct run (see header); the host checks the word.  Once one tile gave up
          // the others follow quickly instead of each waiting out its own bound.
          if (++spins >= kSpinLimit ||
              ((spins & 1023) == 0 && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gave_up = true;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      __hip_atomic_store(&lookback[int64_t(tile) * 256 + tid], ((excl + tot) & kLbValue) | kLbInclusive, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    s_gbase[tid] = dbase[tid] + excl;
  }
  OPH(3);  // look-back (thread 0's digit)
  if (__syncthreads_or(gave_up)) return;  // positions unknown: write nothing
  OPH(4);  // ... and everybody else's
#pragma unroll
  for (int k = 0; k < kIters; ++k) {
    const uint32_t i = k * kOThreads + tid;
    if (i >= count) continue;
    const uint32_t d = (skey[k] >> shift) & 255;  // (neighbouring positions hold the same digit: these two reads are broadcasts)
    const uint32_t gpos = s_gbase[d] + (i - s_dstart[d]);
    if (LAST) {
      out64[gpos] = sidx[k];
    } else {
      keys_out[gpos] = skey[k];
      idx_out[gpos] = sidx[k];
    }
  }
  OPH(7);  // round 2: row indices out
  OPH_FLUSH();
}

// ---- windowed LDS sort: the tail of the hybrid sort ---------------------------------------------------
// After one-sweep passes over the TOP digits only, the keys are ordered by those digits and every
// run of equal top digits ("segment") still has to be ordered by the remaining ones.  For spread-out
// keys the segments are short, so the rest of the sort never has to touch HBM again: a workgroup
// takes the segments that START inside its tile of kWTile positions (at most kWSpan keys: the tile
// plus the tail of its last segment), sorts them in LDS by every digit that varies among them --
// LSD passes with the same match-any ranking as the global passes, stable -- and writes the row
// indices out.  Whole segments only, hence independent of the neighbours.  A segment longer than
// kWSpan - kWTile is reported through `error`; the host then redoes the sort with LSD passes only.
constexpr int kWThreads = 256;  // 512 / 1024 threads per window measure 8 % slower
constexpr int kWWaves = kWThreads / 64;
constexpr int kWTile = 2048;
constexpr int kWSpan = 4096;
constexpr int kWBucketBits = 12;
constexpr int kWBuckets = 1 << kWBucketBits;  // buckets of the window's one-pass sort
constexpr int kWMaxBucket = 24;             // more keys than this in one bucket: LSD passes instead
struct WindowDigits { int n; int shift[10]; };  // (five digits below a global pass at bit 39 and four from there up)

// Workgroups are dealt to the eight XCDs round robin, each with its own L2.  A window reads its tile and, as the tail of
// its last run, up to a whole tile more -- which is the NEXT window's tile: with window = workgroup id those two reads
// come from two different L2s and the overlap is fetched from HBM twice (2.4 GB instead of 1.2 per 100 M keys).  Inside
// every group of 8 x kWinRun workgroups the window index is permuted so that kWinRun consecutive windows run on one
// XCD; windows do not depend on each other, so any order is safe.
constexpr uint32_t kWinRun = 16;
__device__ __forceinline__ uint32_t WindowOfBlock() {
  uint32_t w = blockIdx.x;
  constexpr uint32_t group = 8 * kWinRun;
  const uint32_t g0 = w / group * group, in = w - g0;
  if (g0 + group <= gridDim.x) w = g0 + (in % 8) * kWinRun + in / 8;
  return w;
}
// A crowded window (keys clustered inside it) is not sorted here: its flag is raised and k_window_lsd, a kernel of its
// own with the register budget of its ballot-ranked LSD passes, takes it -- kept in this kernel the fallback's registers
// cost every window a wave per SIMD (222 VGPRs: two workgroups per CU instead of three, 1.21 -> 1.52 ms).
#ifdef WS_PHASES  // -DWS_PHASES: cycles per phase of k_window_sort, summed over the workgroups (profiles/bench_src/ws_phases.py reads them)
__device__ unsigned long long g_ws_phase[64][16];
#define WPH(i) do { if (threadIdx.x == 0) { const unsigned long long _n = __builtin_readcyclecounter(); wph[i] += _n - wph_last; wph_last = _n; } } while (0)
#define WPH_FLUSH() do { if (threadIdx.x == 0) { for (int _i = 0; _i < 12; ++_i) atomicAdd(&g_ws_phase[blockIdx.x & 63][_i], wph[_i]); atomicAdd(&g_ws_phase[blockIdx.x & 63][15], 1ull); } } while (0)
#else
#define WPH(i)
#define WPH_FLUSH()
#endif
// Where the windows begin.  Boundary t (t = 0 .. windows): the first run head at or after position t * kWTile -- the
// keys are sorted by their top part, so "the first position whose top part differs from that of the key in front of the
// tile" is a binary search over at most kWSpan - kWTile positions, twelve dependent loads per thread and fifty thousand
// threads: ~20 us per 100 M keys, after which a window knows its keys without looking at a single candidate (it used
// to load 4096 of them, compare every one with its neighbour and reduce the heads over the workgroup: a third of the
// window kernel).  A boundary travels as two 64-bit words, position << 48 | the part of a key above the window's own
// digits: x = of the key AT the boundary (the window's first key), y = of the key in front of it (the last key of the
// window before).  A run that does not end within kWSpan - kWTile positions leaves the position at kWNoBound: the
// window before reports it.
constexpr uint32_t kWNoBound = 0xFFFFu;
template <typename KeyT>
__global__ __launch_bounds__(256) void k_window_bounds(const KeyT* keys, int64_t n, KeyT top_mask, int64_t nwin, ulonglong2* bounds) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (t > nwin) return;
  const int low_bits = sizeof(KeyT) == 8 ? __ffsll(static_cast<long long>(top_mask)) - 1 : __ffs(static_cast<int>(top_mask)) - 1;
  const int64_t p = t * kWTile;
  if (p >= n) {  // the input's end (boundary `windows`, and any boundary past it)
    const unsigned long long before = n > 0 ? static_cast<unsigned long long>(keys[n - 1]) >> low_bits : 0ull;
    const unsigned long long rel = static_cast<unsigned long long>(n - (t - 1) * kWTile);  // relative to the window BEFORE (what reads y)
    bounds[t] = make_ulonglong2(0ull, (rel <= kWSpan ? rel : kWNoBound) << 48 | before);
    return;
  }
  if (p == 0) {
    bounds[0] = make_ulonglong2(static_cast<unsigned long long>(keys[0]) >> low_bits, 0ull);
    return;
  }
  const KeyT v = keys[p - 1] & top_mask;
  // first position q in [p, lim] with a top part other than v (lim = where the search gives up, or the input's end)
  const int64_t reach = p + (kWSpan - kWTile);
  const int64_t lim = reach < n ? reach : n;
  int64_t lo = p, hi = lim;  // answer in [lo, hi]; keys[hi] differs (or hi == lim: unknown / the end)
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((keys[mid] & top_mask) != v) hi = mid; else lo = mid + 1;
  }
  // lo == lim: no head inside the reach -- the run ends exactly at the input's end (lim == n) or is too long
  const bool found = lo < lim || (lim < n && (keys[lim] & top_mask) != v) || lim == n;
  const unsigned long long before = static_cast<unsigned long long>(keys[lo - 1]) >> low_bits;
  const unsigned long long at = lo < n ? static_cast<unsigned long long>(keys[lo]) >> low_bits : 0ull;
  // x: relative to this boundary's own window (t), y: relative to the window before (t - 1)
  const unsigned long long relx = static_cast<unsigned long long>(lo - p), rely = static_cast<unsigned long long>(lo - (p - kWTile));
  bounds[t] = make_ulonglong2((found ? relx : kWNoBound) << 48 | at, (found && rely <= kWSpan ? rely : kWNoBound) << 48 | before);
}

// 128 registers: four workgroups per CU (32 KB of LDS each) instead of three -- 904 -> 712 us per 100 M uint64
// MARK (rows < 2^31): bit 63 of a row's output = another row of the window carries the same key (k_fix_ties).  A template
// parameter: the flag as a run-time argument cost the kernel its register budget (spills, 0.5 -> 4.6 ms).
template <typename KeyT, bool MARK = false>
__global__ __launch_bounds__(kWThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_window_sort(
    const KeyT* keys, const uint32_t* idx_in, int64_t n, KeyT top_mask, unsigned int* error, unsigned int* crowded /* [windows], zeroed */,
    uint64_t* out64, const ulonglong2* bounds /* [windows + 1], k_window_bounds */) {
  constexpr int kIters = kWSpan / kWThreads;  // 16: step j takes the candidates [j * 256, (j + 1) * 256), a wave 64 consecutive ones
  // What bounds this kernel is instruction issue (four workgroups share a CU: a wave gets a quarter of its SIMD) and
  // the LDS traffic of lanes that go to unrelated addresses (64 lanes over 32 banks: an 8-byte access of a wave takes
  // ~30 cycles instead of 4).  So: no 64-bit key ever goes to LDS; a key costs as few scattered 32-bit accesses as the
  // algorithm allows -- its bucket's counter (an atomic), the bucket's start (a read), one packed word (a write), the
  // words of a bucket it shares (reads), its row index at its place in the sorted window (a write) --; and a wave
  // skips, with a scalar branch, every step whose 64 candidates lie outside the window (half of them on average).
  __shared__ uint32_t s_cnt[kWBuckets];  // per bucket: its count, then start << 16 | count
  __shared__ uint32_t s_sub[kWSpan];     // by bucket: sub-bucket bits << 12 | window position; later the row indices in sorted order
  __shared__ uint32_t s_scan[16];
  __shared__ uint32_t s_big;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t win = WindowOfBlock();
  const int64_t p0 = int64_t(win) * kWTile;
  const int64_t left = n - p0;
  const uint32_t cand = left >= kWSpan ? kWSpan : static_cast<uint32_t>(left);  // candidate keys [p0, p0 + cand)
  if (tid == 0) s_big = 0;
#ifdef WS_PHASES
  unsigned long long wph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long wph_last = __builtin_readcyclecounter();
#endif

  // The window: runs that START in this tile, [lo, hi) in candidate positions.  k_window_bounds has found both ends and
  // the upper parts of the window's first and last key (which is all the bucket pass needs to know about its key range):
  // no candidate is loaded to find them, and only the window's keys and row indices are loaded at all.
  const ulonglong2 b0 = bounds[win], b1 = bounds[win + 1];
  const unsigned long long plo = b0.x, phi = b1.y;
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(plo >> 48));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(phi >> 48));
  if (lo >= static_cast<uint32_t>(kWTile)) return;  // no segment starts here: an earlier tile owns these keys
  if (hi > static_cast<uint32_t>(kWSpan)) {          // the last segment runs past the window
    if (tid == 0) __hip_atomic_store(error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const uint32_t count = hi - lo;
  const int low_bits = sizeof(KeyT) == 8 ? __ffsll(static_cast<long long>(top_mask)) - 1 : __ffs(static_cast<int>(top_mask)) - 1;
  // does step j hold a key of the window in this wave?  (scalar: whole steps are skipped)
#define WS_STEP_LIVE(j) (static_cast<uint32_t>((j) * kWThreads + wave * 64 + 64) > lo && static_cast<uint32_t>((j) * kWThreads + wave * 64) < hi)
  KeyT key[kIters];
  uint32_t idx[kIters];
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    key[j] = 0; idx[j] = 0;
    if (WS_STEP_LIVE(j)) {
      const uint32_t li = j * kWThreads + tid;
      const int64_t at = p0 + (li < cand ? li : cand - 1);
      key[j] = keys[at];
      idx[j] = idx_in[at];
    }
  }
  {
    uint4* z = reinterpret_cast<uint4*>(s_cnt);
    for (int i = tid; i < kWBuckets / 4; i += kWThreads) z[i] = make_uint4(0, 0, 0, 0);
  }
  __syncthreads();
  WPH(0);

  // ---- bucket pass: ONE counting pass instead of one LSD pass per varying digit ----------------------
  // bucket = (key - base) >> shift, monotone in the key: keys of different buckets are in order once every bucket knows
  // where it starts (a scan of the counts), and a key that is alone in its bucket -- six in ten when the keys are
  // spread out -- goes straight to its place.  The keys of a bucket that holds several are ranked inside it on
  // (key, window position): each stages ONE 32-bit word, the 20 key bits below the bucket's << 12 | its position, and
  // counts the bucket's words that are smaller than its own -- its place in the bucket, stable whatever order the
  // counting atomics handed out.  Two words that agree in those 20 bits while the keys have more (one pair in a
  // million among spread-out keys) compare by the keys themselves, fetched from global memory.
  // The window's key range needs no reduction over its keys: the window is ordered by the digits the global passes
  // sorted, so its FIRST key carries the smallest and its LAST key the largest value of everything at and above the
  // lowest of those digits; below it anything goes.  base = first key with the low part cleared, range = (last -
  // first) at that granularity + the whole low part.  A bucket with more than kWMaxBucket keys (keys clustered
  // inside the window) sends the window to the LSD passes.
  constexpr int kSubBits = 20;
  uint32_t bb[kIters], sub[kIters];  // bb: the key's bucket | its arrival number in the bucket << 16
  const KeyT low_mask = (KeyT(1) << low_bits) - 1;
  const KeyT base = static_cast<KeyT>((plo & ((1ull << 48) - 1)) << low_bits);
  const KeyT range = (static_cast<KeyT>((phi & ((1ull << 48) - 1)) << low_bits) - base) | low_mask;
  const int bits = sizeof(KeyT) == 8 ? 64 - __clzll(static_cast<long long>(range)) : 32 - __clz(static_cast<int>(range));
  const int bshift = __builtin_amdgcn_readfirstlane(bits > kWBucketBits ? bits - kWBucketBits : 0);
  const bool sub_exact = bshift <= kSubBits;  // the packed word holds every key bit below the bucket's
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    bb[j] = 0; sub[j] = 0;
    if (!WS_STEP_LIVE(j)) continue;
    const uint32_t li = j * kWThreads + tid;
    if (li >= lo && li < hi) {
      const KeyT rel = key[j] - base;
      const uint32_t bk = static_cast<uint32_t>(rel >> bshift);
      const uint32_t below = sub_exact ? static_cast<uint32_t>(rel & ((KeyT(1) << bshift) - 1)) << (kSubBits - bshift)
                                       : static_cast<uint32_t>(rel >> (bshift - kSubBits)) & ((1u << kSubBits) - 1u);
      sub[j] = below << 12 | (li - lo);
      bb[j] = bk | atomicAdd(&s_cnt[bk], 1u) << 16;
    }
  }
  __syncthreads();
  WPH(2);
  {
    // exclusive scan of the bucket counts: thread t owns buckets [t * per, (t + 1) * per)
    constexpr int kPer = kWBuckets / kWThreads;  // 16
    uint32_t cnt[kPer], mine = 0, biggest = 0;
#pragma unroll
    for (int q = 0; q < kPer / 4; ++q) {
      const uint4 c4 = reinterpret_cast<const uint4*>(s_cnt)[tid * (kPer / 4) + q];
      cnt[4 * q] = c4.x; cnt[4 * q + 1] = c4.y; cnt[4 * q + 2] = c4.z; cnt[4 * q + 3] = c4.w;
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q) { mine += cnt[q]; biggest = cnt[q] > biggest ? cnt[q] : biggest; }
    const uint32_t incl = WaveInclusiveScan(mine);
    if (lane == 63) s_scan[wave] = incl;
    if (biggest > kWMaxBucket) s_big = 1;
    __syncthreads();
    WPH(3);
    if (s_big == 0) {
      uint32_t run = incl - mine;
      for (int w = 0; w < wave; ++w) run += s_scan[w];
#pragma unroll
      for (int q = 0; q < kPer / 4; ++q) {
        uint4 o;
        o.x = run << 16 | cnt[4 * q]; run += cnt[4 * q];
        o.y = run << 16 | cnt[4 * q + 1]; run += cnt[4 * q + 1];
        o.z = run << 16 | cnt[4 * q + 2]; run += cnt[4 * q + 2];
        o.w = run << 16 | cnt[4 * q + 3]; run += cnt[4 * q + 3];
        reinterpret_cast<uint4*>(s_cnt)[tid * (kPer / 4) + q] = o;  // start << 16 | count
      }
      __syncthreads();
      WPH(4);
      // a key's bucket: where it starts and how many keys it holds
      // (all of a thread's reads first: inside the liveness test they went one round trip at a time)
      uint32_t sc[kIters];
#pragma unroll
      for (int j = 0; j < kIters; ++j) { sc[j] = 0; if (WS_STEP_LIVE(j)) sc[j] = s_cnt[bb[j] & 0xFFFFu]; }
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        if (!WS_STEP_LIVE(j)) continue;
        const uint32_t li = j * kWThreads + tid;
        if (!(li >= lo && li < hi)) sc[j] = 0;  // (count 0: not a key of this window)
        if ((sc[j] & 0xFFFFu) > 1) s_sub[(sc[j] >> 16) + (bb[j] >> 16)] = sub[j];  // (a bucket's words sit in the order the atomics handed out)
      }
      __syncthreads();
      WPH(5);
      // place in the sorted window: bucket start + the bucket's OTHER keys that sort before this one.  The first
      // kFlat words of a shared bucket are read WITHOUT a branch: a candidate that does not share its bucket reads a
      // harmless word of its own (s_sub[tid]: no bank conflict) and ignores it -- with the reads inside `if (shared)`
      // every candidate cost the wave a dependent LDS round trip of its own.  A key's own word never counts (it is
      // not smaller than itself).  Larger buckets, and words that tie in the staged key bits, finish in a slow path
      // few steps enter.
      constexpr uint32_t kFlat = 4;
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        if (!WS_STEP_LIVE(j)) continue;
        const uint32_t bs = sc[j] >> 16, sz = sc[j] & 0xFFFFu, lim = sz > 1 ? sz : 0u;
        uint32_t other[kFlat];
#pragma unroll
        for (uint32_t q = 0; q < kFlat; ++q) other[q] = s_sub[q < lim ? bs + q : static_cast<uint32_t>(tid)];
        uint32_t before_me = 0;
        bool slow = lim > kFlat;
        bool tie = false;
#pragma unroll
        for (uint32_t q = 0; q < kFlat; ++q) {
          const uint32_t x = other[q] ^ sub[j];
          before_me += q < lim && other[q] < sub[j] ? 1u : 0u;
          slow = slow || (q < lim && !sub_exact && x != 0 && (x >> 12) == 0);  // another key with the same staged bits
          if constexpr (MARK) tie = tie || (q < lim && sub_exact && x != 0 && (x >> 12) == 0);  // (all the bits below the bucket: the same key)
        }
        if (slow) {  // everything again, ties by the keys themselves (both from global memory), then by the position
          before_me = 0;
          for (uint32_t q = 0; q < sz; ++q) {
            const uint32_t o = s_sub[bs + q];
            if (o == sub[j]) continue;
            if (!sub_exact && ((o ^ sub[j]) >> 12) == 0) {
              const KeyT ok = keys[p0 + lo + (o & 0xFFFu)], mk = keys[p0 + lo + (sub[j] & 0xFFFu)];
              before_me += (ok < mk || (ok == mk && o < sub[j])) ? 1u : 0u;
              if constexpr (MARK) tie = tie || ok == mk;
            } else {
              before_me += o < sub[j] ? 1u : 0u;
              if constexpr (MARK) tie = tie || ((o ^ sub[j]) >> 12) == 0;
            }
          }
        }
        if constexpr (MARK) sc[j] = (bs + before_me) | (tie ? 0x80000000u : 0u);
        else sc[j] = bs + before_me;  // (dead candidates: never used)
      }
      __syncthreads();
      WPH(6);
      // the staged words are done with: their space carries the row indices in sorted order
      uint32_t* s_rows = s_sub;
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        if (!WS_STEP_LIVE(j)) continue;
        const uint32_t li = j * kWThreads + tid;
        if constexpr (MARK) { if (li >= lo && li < hi) s_rows[sc[j] & 0x7FFFFFFFu] = idx[j] | (sc[j] & 0x80000000u); }
        else if (li >= lo && li < hi) s_rows[sc[j]] = idx[j];
      }
      __syncthreads();
      WPH(7);
      if constexpr (MARK) {
        for (uint32_t i = tid; i < count; i += kWThreads) {
          const uint32_t v = s_rows[i];
          out64[p0 + lo + i] = static_cast<uint64_t>(v & 0x7FFFFFFFu) | (static_cast<uint64_t>(v >> 31) << 63);
        }
      } else
      for (uint32_t i = tid; i < count; i += kWThreads) out64[p0 + lo + i] = s_rows[i];
      WPH(8);
      WPH_FLUSH();
      return;
    }
  }
#undef WS_STEP_LIVE
  if (tid == 0) {  // crowded: k_window_lsd sorts this window (error word 3 tells the host that there is work for it)
    crowded[win] = 1u;
    __hip_atomic_store(error + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The windows k_window_sort found crowded (more than kWMaxBucket keys in one bucket): stable LSD passes over the digits
// that vary in the window, ranked with ballots (match-any on the digit bits) -- nothing here rests on the order in which
// LDS atomics hand out their return values.  Same window geometry; a workgroup whose flag is down returns at once.
template <typename KeyT>
__global__ __launch_bounds__(kWThreads) void k_window_lsd(const KeyT* keys, const uint32_t* idx_in, int64_t n, KeyT top_mask,
                                                         WindowDigits digits, const unsigned int* crowded, uint64_t* out64,
                                                         int mark_ties = 0) {
  constexpr bool ARANK = false;
  constexpr int kIters = kWSpan / kWThreads;
  __shared__ KeyT s_keys[kWSpan];
  __shared__ uint32_t s_idx[kWSpan];
  __shared__ uint32_t s_whist[kWWaves][256];
  __shared__ uint32_t s_scan[16];
  __shared__ uint32_t s_lo, s_hi;
  __shared__ KeyT s_or, s_and;
  const uint32_t win = WindowOfBlock();
  if (!crowded[win]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p0 = int64_t(win) * kWTile;
  const int64_t left = n - p0;
  const uint32_t cand = left >= kWSpan ? kWSpan : static_cast<uint32_t>(left);
  if (tid == 0) { s_lo = kWSpan; s_hi = kWSpan; s_or = 0; s_and = ~KeyT(0); }
  KeyT key[kIters];
  uint32_t idx[kIters];
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    key[j] = li < cand ? keys[p0 + li] : KeyT(0);
    idx[j] = li < cand ? idx_in[p0 + li] : 0u;
  }
  const KeyT before = p0 > 0 ? keys[p0 - 1] : KeyT(0);
#pragma unroll
  for (int j = 0; j < kIters; ++j) s_keys[wave * (kIters * 64) + j * 64 + lane] = key[j];
  __syncthreads();
  uint32_t lo = kWSpan, hi = kWSpan;
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    if (li < cand) {
      const KeyT prev = li > 0 ? s_keys[li - 1] : before;
      const bool head = (p0 + li == 0) || ((prev ^ key[j]) & top_mask) != 0;
      if (head && li < kWTile && li < lo) lo = li;
      if (head && li >= kWTile && li < hi) hi = li;
    }
  }
  if (lo < kWSpan) atomicMin(&s_lo, lo);
  if (hi < kWSpan) atomicMin(&s_hi, hi);
  __syncthreads();
  lo = s_lo;
  hi = s_hi == kWSpan ? cand : s_hi;  // (k_window_sort has checked that the window holds its last run)
  const uint32_t count = hi - lo;
  // bring the window to local positions [0, count): the passes below index keys and rows that way
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    if (li >= lo && li < hi) { s_keys[li - lo] = key[j]; s_idx[li - lo] = idx[j]; }
  }
  __syncthreads();
  KeyT k_or = 0, k_and = ~KeyT(0);
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    const bool live = li < count;
    key[j] = live ? s_keys[li] : KeyT(0);
    idx[j] = live ? s_idx[li] : 0u;
    if (live) { k_or |= key[j]; k_and &= key[j]; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    k_or |= __shfl_xor(k_or, o);
    k_and &= __shfl_xor(k_and, o);
  }
  if (lane == 0) {
    if (sizeof(KeyT) == 8) {
      atomicOr(reinterpret_cast<unsigned long long*>(&s_or), static_cast<unsigned long long>(k_or));
      atomicAnd(reinterpret_cast<unsigned long long*>(&s_and), static_cast<unsigned long long>(k_and));
    } else {
      atomicOr(reinterpret_cast<unsigned int*>(&s_or), static_cast<unsigned int>(k_or));
      atomicAnd(reinterpret_cast<unsigned int*>(&s_and), static_cast<unsigned int>(k_and));
    }
  }
  __syncthreads();
  const KeyT varying = s_or ^ s_and;  // bits that differ somewhere in the window: a digit without any is already in order

  for (int p = 0; p < digits.n; ++p) {
    const int shift = digits.shift[p];
    if (((varying >> shift) & 255) == 0) continue;
    for (int i = tid; i < kWWaves * 256; i += kWThreads) (&s_whist[0][0])[i] = 0;
    __syncthreads();
    uint32_t rk[ARANK ? kIters : 1];
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
      if (li < count) {
        const uint32_t r = atomicAdd(&s_whist[wave][(key[j] >> shift) & 255], 1u);
        if (ARANK) rk[ARANK ? j : 0] = r;  // rank among this wave's keys of the digit, in (step, lane) order
      }
    }
    __syncthreads();
    {  // thread d < 256: exclusive scan over digits, then per-wave bases of digit d
      uint32_t c[kWWaves], tot = 0;
      uint32_t incl = 0;
      if (tid < 256) {
#pragma unroll
        for (int w = 0; w < kWWaves; ++w) { c[w] = s_whist[w][tid]; tot += c[w]; }
        incl = WaveInclusiveScan(tot);
        if (lane == 63) s_scan[wave] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t run = incl - tot;
        for (int w = 0; w < wave; ++w) run += s_scan[w];
#pragma unroll
        for (int w = 0; w < kWWaves; ++w) { s_whist[w][tid] = run; run += c[w]; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
      if (wave * (kIters * 64) + j * 64 >= count) continue;  // no live key in this step (wave-uniform)
      const bool live = li < count;
      const uint32_t d = (key[j] >> shift) & 255;
      if (ARANK) {
        if (live) {
          const uint32_t lpos = s_whist[wave][d] + rk[ARANK ? j : 0];
          s_keys[lpos] = key[j];
          s_idx[lpos] = idx[j];
        }
        continue;
      }
      uint64_t m = __ballot(live);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint64_t bb = __ballot((d >> b) & 1);
        m &= ((d >> b) & 1) ? bb : ~bb;
      }
      if (live) {
        const uint32_t rank = RankBelowLane(m);
        const uint32_t old = s_whist[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (rank == 0) s_whist[wave][d] = old + __popcll(m);
        const uint32_t lpos = old + rank;
        s_keys[lpos] = key[j];
        s_idx[lpos] = idx[j];
      }
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
      if (li < count) { key[j] = s_keys[li]; idx[j] = s_idx[li]; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint32_t li = wave * (kIters * 64) + j * 64 + lane;
    if (li >= count) continue;
    uint64_t v = idx[j];
    // (s_keys holds the window's keys in sorted order since the last pass: a key with an equal neighbour is tied)
    if (mark_ties && ((li > 0 && s_keys[li - 1] == key[j]) || (li + 1 < count && s_keys[li + 1] == key[j]))) v |= uint64_t(1) << 63;
    out64[p0 + lo + li] = v;
  }
}

__global__ __launch_bounds__(kThreads) void k_widen_indices(const uint32_t* idx, int64_t n, uint64_t* out) {
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kThreads) out[i] = idx[i];
}

int ScanU32InPlace(amdarrow_ctx* ctx, uint32_t* data, int64_t n, uint32_t* sums, int64_t* sum_offsets) {
  const int64_t nb = (n + 2047) / 2048;
  hipLaunchKernelGGL(k_scan_block_sums, dim3(static_cast<unsigned>(nb)), dim3(kThreads), 0, ctx->stream, data, n, sums);
  AA_HIP(hipGetLastError());
  AA_TRY(LaunchScanU32(ctx, sums, nb, sum_offsets));
  hipLaunchKernelGGL(k_scan_apply, dim3(static_cast<unsigned>(nb)), dim3(kThreads), 0, ctx->stream, data, n, sum_offsets, data);
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

// Do 4096 keys sampled at a fixed stride contain the same key twice?  A value that shows up twice in
// such a sample holds on the order of n / 4096 rows or more -- a run of equal keys no window can take.
// The digit histograms cannot see that (a thousand distinct heavy values fill every digit evenly),
// so the hybrid plan asks one workgroup first (LDS hash set, linear probing; part of k_sort_plan).
constexpr int kSampleKeys = 4096;

// Everything the host's plan needs to know about a column, in ONE small record that one workgroup writes straight into
// mapped host memory (the host waits for an event behind it): per digit histogram the largest count, whether a single
// value holds every key, and the OR / AND of the digit values that occur; which key bits vary; the number of sortable
// rows; whether the stride sample holds a value twice (all of a thread's sample loads are in flight before its first
// insertion: one dependent load after the other, the test alone took 22 us).  Five separate read-backs of 4 KiB each -- a stream synchronisation and
// an idle queue every time, ~30 us apiece -- used to carry the same information.
struct PlanDigit { uint32_t max, trivial, orv, andv; };
struct PlanSummary {
  uint32_t n0, dup;
  unsigned long long bits_set, bits_clear;
  int guess_shift0;  // SortRawClassified: the top digit's shift the histograms were built at (see k_guess_shifts)
  int pad;
  PlanDigit digit[8];
};
constexpr size_t kPlanOffset = kStatusOffset + 256;  // behind the group-by's status words in the pinned block
static_assert(kPlanOffset + sizeof(PlanSummary) <= 8192, "pinned block");
template <typename KeyT>
__global__ __launch_bounds__(256) void k_sort_plan(const unsigned int* ghist, int ndigits, const uint32_t* n0_dev, uint32_t n0_imm,
                                                   const unsigned long long* orand, const KeyT* sample, const HistShifts* hs_dev,
                                                   PlanSummary* out) {
  constexpr int kSlots = 2 * kSampleKeys;
  __shared__ KeyT s_set[kSlots];
  __shared__ unsigned int s_dup;
  __shared__ uint32_t s_red[4][4];
  using Word = typename std::conditional<sizeof(KeyT) == 8, unsigned long long, unsigned int>::type;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t n0 = n0_dev ? *n0_dev : n0_imm;
  const KeyT kEmpty = ~KeyT(0);
  if (tid == 0) s_dup = 0;
  if (sample && n0 >= kSampleKeys) {
    constexpr int kPer = kSampleKeys / 256;
    KeyT k[kPer];
    const int64_t stride = n0 / kSampleKeys;
#pragma unroll
    for (int j = 0; j < kPer; ++j) k[j] = sample[int64_t(j * 256 + tid) * stride];
    for (int i = tid; i < kSlots; i += 256) s_set[i] = kEmpty;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      if (k[j] == kEmpty) continue;  // the one value the set cannot hold; it does not decide anything
      const uint64_t h = static_cast<uint64_t>(k[j]) * 0x9E3779B97F4A7C15ull;
      uint32_t slot = static_cast<uint32_t>(h >> 40) & (kSlots - 1);
      for (;;) {
        const KeyT seen = static_cast<KeyT>(atomicCAS(reinterpret_cast<Word*>(&s_set[slot]), static_cast<Word>(kEmpty), static_cast<Word>(k[j])));
        if (seen == kEmpty) break;
        if (seen == k[j]) { s_dup = 1; break; }
        slot = (slot + 1) & (kSlots - 1);
      }
    }
  }
  __syncthreads();
  for (int q = 0; q < ndigits; ++q) {
    const uint32_t h = ghist[q * 256 + tid];
    uint32_t mx = h, tr = h == n0 ? 1u : 0u, orv = h ? tid : 0u, andv = h ? tid : 255u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t m2 = __shfl_xor(mx, o);
      mx = m2 > mx ? m2 : mx;
      tr |= __shfl_xor(tr, o); orv |= __shfl_xor(orv, o); andv &= __shfl_xor(andv, o);
    }
    if (lane == 0) { s_red[wave][0] = mx; s_red[wave][1] = tr; s_red[wave][2] = orv; s_red[wave][3] = andv; }
    __syncthreads();
    if (tid == 0) {
      PlanDigit d{s_red[0][0], s_red[0][1], s_red[0][2], s_red[0][3]};
      for (int w = 1; w < 4; ++w) {
        d.max = s_red[w][0] > d.max ? s_red[w][0] : d.max;
        d.trivial |= s_red[w][1]; d.orv |= s_red[w][2]; d.andv &= s_red[w][3];
      }
      out->digit[q] = d;
    }
    __syncthreads();
  }
  if (wave == 0) {
    unsigned long long bs = 0, bc = 0;
    if (orand && lane < kCounterSlots) { bs = orand[lane * kCounterStride]; bc = orand[lane * kCounterStride + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { bs |= __shfl_xor(bs, o); bc |= __shfl_xor(bc, o); }
    if (lane == 0) {
      out->n0 = n0; out->dup = s_dup; out->bits_set = bs; out->bits_clear = bc;
      out->guess_shift0 = hs_dev ? hs_dev->shift[0] : 0;
    }
  }
}
// Queue k_sort_plan and wait for its record.
template <typename KeyT>
static int ReadPlan(amdarrow_ctx* ctx, const unsigned int* ghist, int ndigits, const uint32_t* n0_dev, uint32_t n0_imm,
                    const unsigned long long* orand, const KeyT* sample, const HistShifts* hs_dev, PlanSummary* plan) {
  auto* out = reinterpret_cast<PlanSummary*>(static_cast<char*>(ctx->pinned_dev) + kPlanOffset);
  hipLaunchKernelGGL((k_sort_plan<KeyT>), dim3(1), dim3(256), 0, ctx->stream, ghist, ndigits, n0_dev, n0_imm, orand, sample, hs_dev, out);
  AA_HIP(hipGetLastError());
  AA_HIP(hipEventRecord(ctx->status_event, ctx->stream));
  AA_HIP(hipEventSynchronize(ctx->status_event));
  memcpy(plan, static_cast<const char*>(ctx->pinned) + kPlanOffset, sizeof(PlanSummary));
  return AMDARROW_OK;
}
// The chunk table goes to the device through the pinned block when it fits (a copy from pageable memory has to be
// waited for; this one is stream-ordered like everything behind it).
static int UploadChunks(amdarrow_ctx* ctx, SortChunk* d_chunks, const std::vector<SortChunk>& hc) {
  const size_t bytes = sizeof(SortChunk) * hc.size();
  if (bytes <= 4096) {
    AA_HIP(hipStreamSynchronize(ctx->stream));  // (nothing of an earlier call may still be reading the block; idle in practice)
    memcpy(ctx->pinned, hc.data(), bytes);
    AA_HIP(hipMemcpyAsync(d_chunks, ctx->pinned, bytes, hipMemcpyHostToDevice, ctx->stream));
    return AMDARROW_OK;
  }
  AA_HIP(hipMemcpyAsync(d_chunks, hc.data(), bytes, hipMemcpyHostToDevice, ctx->stream));
  AA_HIP(hipStreamSynchronize(ctx->stream));  // hc is pageable host memory
  return AMDARROW_OK;
}
// The highest varying key bit of the sampled tiles -> the digit positions the full pass builds its histograms at
// (SortRawClassified), decided on the device: the full pass is queued right behind the sample.
template <typename KeyT>
__global__ void k_guess_shifts(const unsigned long long* orand, HistShifts* hs_out) {
  unsigned long long bs = 0, bc = 0;
  for (int s = 0; s < kCounterSlots; ++s) { bs |= orand[s * kCounterStride]; bc |= orand[s * kCounterStride + 1]; }
  const unsigned long long varying = bs & bc;
  const int T = varying ? 63 - __clzll(static_cast<long long>(varying)) : int(8 * sizeof(KeyT)) - 1;
  HistShifts h{};
  h.n = 4;
  for (int k = 0; k < 4; ++k) { const int s = T + 1 - 8 * (k + 1); h.shift[k] = s > 0 ? s : 0; }
  *hs_out = h;
}

// One workgroup ranks pseudo-random digits (uniform over 256 values, over 4 values, all equal) both
// ways and counts the keys whose atomic rank differs from the match-any rank.
__global__ __launch_bounds__(256) void k_probe_lds_rank(unsigned int* mismatches) {
  __shared__ uint32_t s_cnt[4][256];
  const int tid = threadIdx.x, wave = tid >> 6;
  unsigned int bad = 0;
  for (int round = 0; round < 24; ++round) {
    for (int i = tid; i < 4 * 256; i += 256) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    uint32_t d[8], rk[8], seen[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t x = (round * 8 + j) * 256 + tid;
      x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
      d[j] = round % 3 == 0 ? (x & 255) : round % 3 == 1 ? (x & 3) : 7u;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) rk[j] = atomicAdd(&s_cnt[wave][d[j]], 1u);
    __syncthreads();
    for (int i = tid; i < 4 * 256; i += 256) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t m = ~uint64_t(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const uint64_t bb = __ballot((d[j] >> b) & 1);
        m &= ((d[j] >> b) & 1) ? bb : ~bb;
      }
      const uint32_t rank = RankBelowLane(m);
      const uint32_t old = s_cnt[wave][d[j]];
      __builtin_amdgcn_wave_barrier();
      if (rank == 0) s_cnt[wave][d[j]] = old + __popcll(m);
      __builtin_amdgcn_wave_barrier();
      seen[j] = old + rank;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bad += rk[j] != seen[j];
    __syncthreads();
  }
  if (bad) atomicAdd(mismatches, bad);
}

// Does an LDS atomic hand out its return values in lane order on this device (see k_radix_onesweep)?
// Checked once per context; any mismatch -- or a failed launch -- selects the ballot ranking.
static bool LdsAtomicsLaneOrdered(amdarrow_ctx* ctx) {
  if (ctx->lds_rank_ordered < 0) {
    ctx->lds_rank_ordered = 0;
    void* base;
    if (!getenv("AMDARROW_SORT_BALLOT_RANK") && Scratch(ctx, 256, &base) == AMDARROW_OK &&
        hipMemsetAsync(base, 0, 4, ctx->stream) == hipSuccess) {
      hipLaunchKernelGGL(k_probe_lds_rank, dim3(1), dim3(256), 0, ctx->stream, static_cast<unsigned int*>(base));
      unsigned int bad = 1;
      if (hipGetLastError() == hipSuccess && ReadBack(ctx, base, 4, &bad) == AMDARROW_OK && bad == 0) ctx->lds_rank_ordered = 1;
    }
  }
  return ctx->lds_rank_ordered == 1;
}

// Tiles per XCD run of the one-sweep passes (see k_radix_onesweep), or 0 = tile = workgroup id.  The permuted order
// needs a whole group of 8 x 16 workgroups resident at once -- asked of the occupancy calculator for this very kernel
// -- on a device with eight XCDs; anything else (a partitioned or CU-masked device, a smaller part) keeps dispatch order,
// where the look-back only ever waits for workgroups that were dispatched earlier.
static uint32_t XcdRunFor(amdarrow_ctx* ctx, const void* kernel) {
  if (getenv("AMDARROW_SORT_NO_XCD_RUNS")) return 0;
  uint32_t kXcdRun = 16;
  if (const char* env = getenv("AMDARROW_SORT_XCD_RUN")) kXcdRun = static_cast<uint32_t>(std::max(1, atoi(env)));  // (tuning knob)
  int per_cu = 0, xccs = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kOThreads, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
  if (hipDeviceGetAttribute(&xccs, hipDeviceAttributeNumberOfXccs, ctx->device) != hipSuccess) { (void)hipGetLastError(); xccs = ctx->num_cus == 256 ? 8 : 0; }
  return xccs == 8 && int64_t(per_cu) * ctx->num_cus >= int64_t(8) * kXcdRun ? kXcdRun : 0;
}

// Nullable / floating-point / chunked columns WITHOUT the partition's copy (0.63 ms per 100 M float64: it reads the
// column and writes 12 bytes per row that the first radix pass reads again).  The rows' classes are counted per
// one-sweep tile (k_rawc_scan<0>, which also tells which key bits vary), the plan's digits -- aligned with the highest
// varying bit -- get their histograms from the column itself (k_rawc_scan<1>), and the FIRST one-sweep pass reads the
// chunks directly: it classifies the rows again, drops NaNs and nulls at their final places behind the sortable rows
// (stable: tile order, row order inside the tile) and sorts the rest by the lowest of the plan's digits.  Only taken
// when the hybrid plan applies (large input, spread-out keys, runs that fit the windows); returns *done = false
// otherwise and SortImpl goes its usual way.
template <typename KeyT>
static int SortRawClassified(amdarrow_ctx* ctx, const std::vector<SortChunk>& hc, int type_id, int order, int64_t n, uint64_t* out,
                             bool arank, bool* done) {
  *done = false;
  const int nchunks = static_cast<int>(hc.size());
  const int key_bits = BitWidth(type_id);
  const KeyT key_mask = ~KeyT(0);
  const KeyT flip = order == AMDARROW_DESCENDING ? key_mask : 0;
  (void)key_bits;
  int64_t notiles = 0;
  for (const auto& c : hc) notiles += (c.length + kOTileRows - 1) / kOTileRows;
  if (notiles == 0) return AMDARROW_OK;
  const int64_t scan_nb = (3 * notiles + 2047) / 2048;
  const int64_t nwin = (n + kWTile - 1) / kWTile;
  const int64_t nb_max = std::max<int64_t>(notiles, (n + kOTile - 1) / kOTile);
  Carver cv;
  const size_t o_chunks = cv.Take(sizeof(SortChunk) * nchunks);
  const size_t o_counts = cv.Take(4 * 3 * notiles);
  const size_t o_sums = cv.Take(4 * scan_nb);
  const size_t o_sumoff = cv.Take(8 * (scan_nb + 1));
  const size_t o_keys_a = cv.Take(sizeof(KeyT) * n);
  const size_t o_keys_b = cv.Take(sizeof(KeyT) * n);
  const size_t o_idx_a = cv.Take(4 * n);
  const size_t o_idx_b = cv.Take(4 * n);
  const size_t o_ghist = cv.Take(4 * 4 * 256);
  const size_t o_dbase = cv.Take(4 * 4 * 256);
  const size_t o_orand = cv.Take(kCounterBytes);
  const size_t o_ctl = cv.Take(256);
  const size_t lb_words = 256 * static_cast<size_t>(nb_max);  // per pass: a word per tile and digit
  const size_t o_lookback = cv.Take(4 * lb_words * 4);
  const size_t o_crowded = cv.Take(4 * static_cast<size_t>(nwin));
  const size_t o_bounds = cv.Take(16 * static_cast<size_t>(nwin + 1));  // k_window_bounds
  void* base;
  AA_TRY(Scratch(ctx, cv.off, &base));
  char* b = static_cast<char*>(base);
  auto* d_chunks = reinterpret_cast<SortChunk*>(b + o_chunks);
  auto* counts = reinterpret_cast<uint32_t*>(b + o_counts);
  auto* sums = reinterpret_cast<uint32_t*>(b + o_sums);
  auto* sumoff = reinterpret_cast<int64_t*>(b + o_sumoff);
  KeyT* keys_a = reinterpret_cast<KeyT*>(b + o_keys_a);
  KeyT* keys_b = reinterpret_cast<KeyT*>(b + o_keys_b);
  uint32_t* idx_a = reinterpret_cast<uint32_t*>(b + o_idx_a);
  uint32_t* idx_b = reinterpret_cast<uint32_t*>(b + o_idx_b);
  auto* ghist = reinterpret_cast<unsigned int*>(b + o_ghist);
  auto* dbase = reinterpret_cast<uint32_t*>(b + o_dbase);
  auto* orand = reinterpret_cast<unsigned long long*>(b + o_orand);
  auto* ctl = reinterpret_cast<unsigned int*>(b + o_ctl);
  auto* lookback = reinterpret_cast<uint32_t*>(b + o_lookback);
  auto* crowded = reinterpret_cast<unsigned int*>(b + o_crowded);
  AA_TRY(UploadChunks(ctx, d_chunks, hc));
  // 1. classes per tile, and which key bits vary
  // (a first look at every 64th tile tells where the keys' highest varying bit probably is: the full pass then builds
  // the plan's histograms next to the counts, and only a wrong guess costs a second pass over the column.  The guess
  // stays on the device -- k_guess_shifts hands the digit positions to the full pass -- and the host learns everything
  // it plans with from ONE record, see k_sort_plan)
  auto shifts_for = [](int T) { HistShifts h{}; h.n = 4; for (int k = 0; k < 4; ++k) h.shift[k] = std::max(T + 1 - 8 * (k + 1), 0); return h; };
  const int grid = GridFor(notiles, 1, ctx->num_cus * 4);
  auto* hs_dev = reinterpret_cast<HistShifts*>(ctl + 32);  // (the control block is cleared only after the plan has been read)
  AA_TRY(ClearCounters(ctx, orand));
  hipLaunchKernelGGL((k_rawc_scan<KeyT, 1>), dim3(GridFor((notiles + 63) / 64, 1, ctx->num_cus * 4)), dim3(kRcThreads), 0, ctx->stream, d_chunks, nchunks,
                     type_id, notiles, int64_t(64), key_mask, flip, counts, orand, HistShifts{}, static_cast<unsigned int*>(nullptr), static_cast<const HistShifts*>(nullptr));
  hipLaunchKernelGGL((k_guess_shifts<KeyT>), dim3(1), dim3(1), 0, ctx->stream, orand, hs_dev);
  AA_HIP(hipGetLastError());
  AA_TRY(ClearCounters(ctx, orand));
  AA_HIP(hipMemsetAsync(ghist, 0, 4 * 4 * 256, ctx->stream));
  hipLaunchKernelGGL((k_rawc_scan<KeyT, 3>), dim3(grid), dim3(kRcThreads), 0, ctx->stream, d_chunks, nchunks, type_id, notiles, int64_t(1), key_mask, flip, counts,
                     orand, HistShifts{}, ghist, static_cast<const HistShifts*>(hs_dev));
  AA_HIP(hipGetLastError());
  AA_TRY(ScanU32InPlace(ctx, counts, 3 * notiles, sums, sumoff));
  PlanSummary plan;
  AA_TRY(ReadPlan<KeyT>(ctx, ghist, 4, counts + notiles /* start of the NaN region = number of sortable rows */, 0, orand,
                        static_cast<const KeyT*>(nullptr), hs_dev, &plan));
  const int64_t n0 = plan.n0;
  const unsigned long long varying = plan.bits_set & plan.bits_clear;
  if (n0 < (int64_t(1) << 21) || varying == 0) return AMDARROW_OK;  // small or constant: the usual way
  // 2. the plan: digits aligned with the highest varying bit
  const int T = 63 - __builtin_clzll(varying);
  if (T < 31) return AMDARROW_OK;
  HistShifts hs = shifts_for(T);
  if (hs.shift[0] != plan.guess_shift0) {  // the sample missed the keys that carry the top bit: the histograms again, at the right place
    AA_HIP(hipMemsetAsync(ghist, 0, 4 * 4 * 256, ctx->stream));
    hipLaunchKernelGGL((k_rawc_scan<KeyT, 2>), dim3(grid), dim3(kRcThreads), 0, ctx->stream, d_chunks, nchunks, type_id, notiles, int64_t(1), key_mask, flip,
                       counts, orand, hs, ghist, static_cast<const HistShifts*>(nullptr));
    AA_HIP(hipGetLastError());
    AA_TRY(ReadPlan<KeyT>(ctx, ghist, 4, counts + notiles, 0, static_cast<const unsigned long long*>(nullptr), static_cast<const KeyT*>(nullptr),
                          static_cast<const HistShifts*>(nullptr), &plan));
  }
  int top = 0;
  double expected = static_cast<double>(n0);
  for (int m = 1; m <= 4 && top == 0; ++m) {
    expected *= static_cast<double>(plan.digit[m - 1].max) / static_cast<double>(n0);
    if (m >= 2 && hs.shift[m - 1] >= 16 && expected <= 0.8 * (kWSpan - kWTile)) top = m;
  }
  if (top == 0) return AMDARROW_OK;
  // (a value that owns ~n / 4096 rows or more hides from the digit histograms when there are many such values: the
  // window kernel notices the run and the sort is redone without the hybrid -- SortImpl's sample check needs the
  // partitioned keys, which this path does not have; the retry below covers it)
  // 3. the passes: the first reads the chunks, the others the keys it wrote
  hipLaunchKernelGGL(k_digit_bases, dim3(4), dim3(kThreads), 0, ctx->stream, ghist, dbase);
  AA_HIP(hipGetLastError());
  AA_HIP(hipMemsetAsync(ctl, 0, 256, ctx->stream));
  AA_HIP(hipMemsetAsync(lookback, 0, 4 * lb_words * top, ctx->stream));
  const int low_shift = hs.shift[top - 1];
  for (int a = 0; a < top; ++a) {
    const int k = top - 1 - a;  // lowest digit first
    uint32_t* lb = lookback + lb_words * a;
    const uint32_t* db = dbase + 256 * k;
    if (a == 0) {
      auto* pass = arank ? k_radix_onesweep<KeyT, false, true, 2> : k_radix_onesweep<KeyT, false, false, 2>;
      hipLaunchKernelGGL(pass, dim3(static_cast<unsigned>(notiles)), dim3(kOThreads), 0, ctx->stream, static_cast<const KeyT*>(nullptr),
                         static_cast<const uint32_t*>(nullptr), n, hs.shift[k], db, lb, ctl + 15, keys_b, idx_b, out, type_id, flip, key_mask,
                         XcdRunFor(ctx, reinterpret_cast<const void*>(pass)), RawcArgs{d_chunks, nchunks, counts, notiles});
    } else {
      auto* pass = arank ? k_radix_onesweep<KeyT, false, true> : k_radix_onesweep<KeyT, false, false>;
      hipLaunchKernelGGL(pass, dim3(static_cast<unsigned>((n0 + kOTile - 1) / kOTile)), dim3(kOThreads), 0, ctx->stream, keys_a, idx_a, n0, hs.shift[k], db, lb,
                         ctl + 15, keys_b, idx_b, out, 0, KeyT(0), KeyT(0), XcdRunFor(ctx, reinterpret_cast<const void*>(pass)), RawcArgs{nullptr, 0, nullptr, 0});
    }
    AA_HIP(hipGetLastError());
    std::swap(keys_a, keys_b);
    std::swap(idx_a, idx_b);
  }
  // 4. the windows
  const int hi_bit = hs.shift[0] + 8;
  const KeyT top_mask = (hi_bit >= int(8 * sizeof(KeyT)) ? ~KeyT(0) : (KeyT(1) << hi_bit) - 1) & ~((KeyT(1) << low_shift) - 1);
  WindowDigits wd{};
  for (int sh = low_shift - 8; wd.n < 8; sh -= 8) {
    wd.shift[wd.n++] = std::max(sh, 0);
    if (sh <= 0) break;
  }
  std::reverse(wd.shift, wd.shift + wd.n);
  // ... and the digits of the global passes on top: a crowded window that holds several runs (the keys of a float column
  // on either side of zero: two clusters at the ends of the window's range) lost their order to the passes over the low
  // digits -- round 3's full-size float64 test missed it, its values are all positive.  A window of one run skips them
  // (no bit of them varies).
  for (int sh = low_shift; sh < int(8 * sizeof(KeyT)) && wd.n < 10; sh += 8) wd.shift[wd.n++] = sh;
  const unsigned nw = static_cast<unsigned>((n0 + kWTile - 1) / kWTile);
  AA_HIP(hipMemsetAsync(crowded, 0, 4 * static_cast<size_t>(nw), ctx->stream));
  auto* wbounds = reinterpret_cast<ulonglong2*>(b + o_bounds);
  hipLaunchKernelGGL((k_window_bounds<KeyT>), dim3((nw + 1 + 255) / 256), dim3(256), 0, ctx->stream, keys_a, n0, top_mask, int64_t(nw), wbounds);
  if (ctx->sort_mark_ties) hipLaunchKernelGGL((k_window_sort<KeyT, true>), dim3(nw), dim3(kWThreads), 0, ctx->stream, keys_a, idx_a, n0, top_mask, ctl + 14, crowded, out, wbounds);
  else hipLaunchKernelGGL((k_window_sort<KeyT>), dim3(nw), dim3(kWThreads), 0, ctx->stream, keys_a, idx_a, n0, top_mask, ctl + 14, crowded, out, wbounds);
  AA_HIP(hipGetLastError());
  unsigned int err[4] = {0, 0, 0, 0};
  AA_TRY(ReadBack(ctx, ctl + 14, 16, err));
  if (err[2]) { ctx->lds_rank_ordered = 0; return AMDARROW_OK; }  // (SortImpl redoes everything with the ballot ranking)
  if (err[0] || err[1]) return AMDARROW_OK;                       // a run too long for the windows / a pass gave up: the usual way
  if (err[3]) {
    hipLaunchKernelGGL((k_window_lsd<KeyT>), dim3(nw), dim3(kWThreads), 0, ctx->stream, keys_a, idx_a, n0, top_mask, wd, crowded, out, ctx->sort_mark_ties ? 1 : 0);
    AA_HIP(hipGetLastError());
  }
  if (ctx->sort_mark_ties) ctx->sort_ties_marked = true;
  *done = true;
  return AMDARROW_OK;
}

template <typename KeyT>
static int SortImpl(amdarrow_ctx* ctx, const amdarrow_array* chunks, int nchunks, int type_id, int order, int64_t n,
                    uint64_t* out, const uint64_t* perm = nullptr, bool allow_onesweep = true, bool allow_hybrid = true) {
  constexpr int kPasses = sizeof(KeyT);
  const bool arank = LdsAtomicsLaneOrdered(ctx);  // before the arena is carved: the probe borrows it
  const int key_bits = BitWidth(type_id);
  const KeyT key_mask = key_bits >= static_cast<int>(8 * sizeof(KeyT)) ? ~KeyT(0) : ((KeyT(1) << key_bits) - 1);
  const KeyT flip = order == AMDARROW_DESCENDING ? key_mask : 0;  // Descending = complemented key
  const int npasses = key_bits / 8;

  // chunk table + classify tiles
  std::vector<SortChunk> hc(nchunks);
  int64_t row = 0, tiles = 0, otiles = 0;
  for (int c = 0; c < nchunks; ++c) {
    hc[c] = SortChunk{chunks[c].values, MayHaveNulls(&chunks[c]) ? chunks[c].validity : nullptr, chunks[c].offset,
                      chunks[c].length, row, tiles, otiles};
    otiles += (chunks[c].length + kOTileRows - 1) / kOTileRows;
    row += chunks[c].length;
    tiles += (chunks[c].length + kCTile - 1) / kCTile;
  }
  {  // nullable / floating-point / chunked columns: the classified raw path first (no partition copy)
    bool plain = !IsFloat(type_id) && nchunks == 1;
    for (const auto& c : hc) plain = plain && c.valid == nullptr;
    if (!perm && !plain && allow_onesweep && allow_hybrid && n >= (int64_t(1) << 21) && n < (int64_t(1) << 30) &&
        BitWidth(type_id) == 8 * int(sizeof(KeyT)) && !getenv("AMDARROW_SORT_THREE_KERNEL") && !getenv("AMDARROW_SORT_NO_RAW_FIRST_PASS") &&
        !getenv("AMDARROW_SORT_NO_HYBRID") && !getenv("AMDARROW_SORT_HYBRID_MIN")) {
      bool done = false;
      AA_TRY(SortRawClassified<KeyT>(ctx, hc, type_id, order, n, out, LdsAtomicsLaneOrdered(ctx), &done));
      if (done) return AMDARROW_OK;
    }
  }
  const int64_t ntiles = perm ? (n + kCTile - 1) / kCTile : tiles;
  const int64_t nblocks = (n + kRTile - 1) / kRTile;
  const int64_t hist_n = 256 * nblocks;
  const int64_t scan_nb = (std::max<int64_t>(hist_n, 3 * ntiles) + 2047) / 2048;

  Carver cv;
  const size_t o_chunks = cv.Take(sizeof(SortChunk) * nchunks);
  const size_t o_counts = cv.Take(4 * 3 * ntiles);
  const size_t o_keys_a = cv.Take(sizeof(KeyT) * n);
  const size_t o_keys_b = cv.Take(sizeof(KeyT) * n);
  const size_t o_idx_a = cv.Take(4 * n);
  const size_t o_idx_b = cv.Take(4 * n);
  const size_t o_ghist = cv.Take(4 * 256 * kPasses);
  const size_t o_orand = cv.Take(kCounterBytes);  // k_sort_hist: which key bits are set / clear somewhere (slotted)
  const size_t o_bhist = cv.Take(4 * hist_n);
  const size_t o_sums = cv.Take(4 * scan_nb);
  const size_t o_sumoff = cv.Take(8 * (scan_nb + 1));
  const size_t o_dbase = cv.Take(4 * 256 * kPasses);
  const size_t o_ctl = cv.Take(256);  // [13] duplicate-sample flag, [14] error word of the window sort, [15] of the one-sweep passes
  const size_t lb_words = static_cast<size_t>(hist_n);  // per pass: a word per tile and digit
  const size_t o_lookback = cv.Take(4 * lb_words * kPasses);
  const size_t o_bounds = cv.Take(16 * static_cast<size_t>((n + kWTile - 1) / kWTile + 1));  // k_window_bounds
  void* base;
  AA_TRY(Scratch(ctx, cv.off, &base));
  char* b = static_cast<char*>(base);
  auto* d_chunks = reinterpret_cast<SortChunk*>(b + o_chunks);
  auto* counts = reinterpret_cast<uint32_t*>(b + o_counts);
  KeyT* keys_a = reinterpret_cast<KeyT*>(b + o_keys_a);
  KeyT* keys_b = reinterpret_cast<KeyT*>(b + o_keys_b);
  uint32_t* idx_a = reinterpret_cast<uint32_t*>(b + o_idx_a);
  uint32_t* idx_b = reinterpret_cast<uint32_t*>(b + o_idx_b);
  auto* ghist = reinterpret_cast<unsigned int*>(b + o_ghist);
  auto* bhist = reinterpret_cast<uint32_t*>(b + o_bhist);
  auto* sums = reinterpret_cast<uint32_t*>(b + o_sums);
  auto* sumoff = reinterpret_cast<int64_t*>(b + o_sumoff);
  auto* dbase = reinterpret_cast<uint32_t*>(b + o_dbase);
  auto* ctl = reinterpret_cast<unsigned int*>(b + o_ctl);
  auto* lookback = reinterpret_cast<uint32_t*>(b + o_lookback);

  AA_TRY(UploadChunks(ctx, d_chunks, hc));

  // 1. stable 3-way partition: sortable | NaN | null (+ the digit histograms of the sortable keys)
  AA_HIP(hipMemsetAsync(ghist, 0, 4 * 256 * kPasses, ctx->stream));
  const int grid_t = GridFor(ntiles, 1, ctx->num_cus * 8);
  bool all_sortable = !perm && !IsFloat(type_id);
  for (const auto& c : hc) all_sortable = all_sortable && c.valid == nullptr;
  auto count_and_scan = [&]() -> int {  // the tiles' class counts -> the partition's offsets (rows < 2^32)
    if (all_sortable)
      hipLaunchKernelGGL(k_sort_count_all_sortable, dim3(static_cast<unsigned>((ntiles + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream,
                         d_chunks, nchunks, ntiles, counts);
    else if (perm)
      hipLaunchKernelGGL((k_sort_count<KeyT, true>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, d_chunks, nchunks, type_id,
                         ntiles, perm, n, counts);
    else
      hipLaunchKernelGGL((k_sort_count<KeyT, false>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, d_chunks, nchunks, type_id,
                         ntiles, perm, n, counts);
    AA_HIP(hipGetLastError());
    return ScanU32InPlace(ctx, counts, 3 * ntiles, sums, sumoff);
  };
  // One chunk of integers without a validity buffer, as wide as the sort key: every row is sortable and the partition
  // below would be a copy.  Only its histograms are built (k_sort_hist) and the FIRST radix pass reads the column
  // itself; if the plan below leaves no pass that can (no pass at all, or a single last one), the partition runs
  // after all (and only then are the tiles counted).
  const bool raw_try = all_sortable && nchunks == 1 && allow_onesweep && n < (int64_t(1) << 30) && BitWidth(type_id) == 8 * int(sizeof(KeyT)) &&
                       !getenv("AMDARROW_SORT_THREE_KERNEL") && !getenv("AMDARROW_SORT_NO_RAW_FIRST_PASS");
  auto run_partition = [&]() -> int {
    if (perm)
      hipLaunchKernelGGL((k_sort_partition<KeyT, true>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, d_chunks, nchunks,
                         type_id, ntiles, perm, n, counts, key_mask, flip, keys_a, idx_a, out, ghist);
    else
      hipLaunchKernelGGL((k_sort_partition<KeyT, false>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, d_chunks, nchunks,
                         type_id, ntiles, perm, n, counts, key_mask, flip, keys_a, idx_a, out, ghist);
    AA_HIP(hipGetLastError());
    return AMDARROW_OK;
  };
  // MakeKey of a signed integer flips the sign bit; together with the order's `flip` one XOR constant
  const KeyT raw_xor = flip ^ (IsSignedInt(type_id) ? KeyT(1) << (8 * sizeof(KeyT) - 1) : KeyT(0));
  auto* orand = reinterpret_cast<unsigned long long*>(b + o_orand);
  constexpr int kTopDigits = kPasses < 4 ? kPasses : 4;  // digits whose histograms the first hist pass builds
  bool top_only = false;                                 // ... and whether the lower digits' histograms are missing
  const char* min_env = getenv("AMDARROW_SORT_HYBRID_MIN");  // testing knob: smallest input that takes the hybrid path
  const int64_t hybrid_min = min_env ? atoll(min_env) : (int64_t(1) << 21);
  // the stride sample (see k_sort_plan) is looked at whenever the hybrid plan can come into question at all
  const bool want_sample = allow_onesweep && allow_hybrid && n < (int64_t(1) << 30) && n >= hybrid_min && !min_env /* the testing knob forces the attempt */ &&
                           !getenv("AMDARROW_SORT_THREE_KERNEL") && !getenv("AMDARROW_SORT_NO_HYBRID");
  if (raw_try) {
    AA_TRY(ClearCounters(ctx, orand));
    top_only = kPasses > kTopDigits && allow_hybrid && !getenv("AMDARROW_SORT_FULL_HIST");
    if (top_only)
      hipLaunchKernelGGL((k_sort_hist<KeyT, kTopDigits>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, hc[0].values, hc[0].off, type_id, n, key_mask, raw_xor, ghist, orand);
    else
      hipLaunchKernelGGL((k_sort_hist<KeyT, kPasses>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, hc[0].values, hc[0].off, type_id, n, key_mask, raw_xor, ghist, orand);
    AA_HIP(hipGetLastError());
  } else {
    AA_TRY(count_and_scan());
    AA_TRY(run_partition());
  }
  // 2. what the histograms of every digit say (built by the partition kernel / k_sort_hist), in one record; skip digits on which all keys agree
  // (equal keys stay equal under the key transformation: the column itself serves as the sample where the partition did not run)
  PlanSummary plan;
  AA_TRY(ReadPlan<KeyT>(ctx, ghist, kPasses, raw_try ? nullptr : counts + ntiles /* start of the NaN region = number of sortable rows */,
                        static_cast<uint32_t>(raw_try ? n : 0), raw_try ? orand : nullptr,
                        want_sample ? (raw_try ? reinterpret_cast<const KeyT*>(hc[0].values) + hc[0].off : keys_a) : nullptr,
                        static_cast<const HistShifts*>(nullptr), &plan));
  const int64_t n0 = plan.n0;
  if (n0 == 0) return AMDARROW_OK;
  const unsigned long long varying_bits = plan.bits_set & plan.bits_clear;  // a bit varies iff some key has it set and some key has it clear
  auto find_active = [&](bool low_from_bits) {
    std::vector<int> act;
    for (int p = 0; p < npasses; ++p) {
      if (low_from_bits && p < kPasses - kTopDigits) {  // no histogram: the digit varies iff one of its bits does
        if ((varying_bits >> (8 * p)) & 255) act.push_back(p);
        continue;
      }
      if (!plan.digit[p].trivial) act.push_back(p);
    }
    return act;
  };
  std::vector<int> active = find_active(top_only);
  // the lower digits' histograms after all (a plan without the hybrid, or one whose global passes reach below the top digits)
  auto ensure_full_hist = [&]() -> int {
    if (!top_only) return AMDARROW_OK;
    top_only = false;
    AA_HIP(hipMemsetAsync(ghist, 0, 4 * 256 * kPasses, ctx->stream));
    hipLaunchKernelGGL((k_sort_hist<KeyT, kPasses>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, hc[0].values, hc[0].off, type_id, n, key_mask, raw_xor, ghist,
                       static_cast<unsigned long long*>(nullptr));
    AA_HIP(hipGetLastError());
    PlanSummary full;
    AA_TRY(ReadPlan<KeyT>(ctx, ghist, kPasses, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t>(n), static_cast<const unsigned long long*>(nullptr),
                          static_cast<const KeyT*>(nullptr), static_cast<const HistShifts*>(nullptr), &full));
    for (int p = 0; p < kPasses; ++p) plan.digit[p] = full.digit[p];
    active = find_active(false);
    return AMDARROW_OK;
  };
  if (top_only) {  // the (up to four) highest active digits are the ones a hybrid plan may pass over: their histograms must be real
    const size_t look = std::min<size_t>(4, active.size());
    for (size_t q = 0; q < look; ++q)
      if (active[active.size() - 1 - q] < kPasses - kTopDigits) { AA_TRY(ensure_full_hist()); break; }
  }
  if (active.empty()) {
    if (raw_try) hipLaunchKernelGGL(k_iota_indices, dim3(GridFor(n0, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, n0, out);
    else
    hipLaunchKernelGGL(k_widen_indices, dim3(GridFor(n0, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream, idx_a, n0,
                       out);
    AA_HIP(hipGetLastError());
    return AMDARROW_OK;
  }
  // 3. LSD passes
  const int64_t nb0 = (n0 + kRTile - 1) / kRTile;
  const int64_t hn0 = 256 * nb0;
  const bool onesweep = allow_onesweep && n0 < (int64_t(1) << 30) && !getenv("AMDARROW_SORT_THREE_KERNEL");
  // Hybrid: one-sweep passes over the top `top` active digits only, the rest in LDS (k_window_sort).
  // Worth it when it saves at least two passes over HBM, and possible when the runs of equal top
  // digits are short.  Their length is estimated from the digit histograms as if the digits were
  // independent (n0 x the largest share of each top digit); the window kernel checks the real
  // lengths and reports a run that does not fit, in which case the sort is redone without it.
  int top = 0;
  const bool heavy_values = want_sample && plan.dup != 0;  // (a sample of fewer than kSampleKeys sortable rows reports nothing: no hybrid below that anyway)
  if (onesweep && allow_hybrid && !heavy_values && n0 >= hybrid_min && !getenv("AMDARROW_SORT_NO_HYBRID")) {
    for (int m = 2; m <= 4 && top == 0; ++m) {
      if (static_cast<int>(active.size()) < m + 2) break;
      double expected = static_cast<double>(n0);
      for (int q = 0; q < m; ++q) {
        const int p = active[active.size() - 1 - q];
        expected *= static_cast<double>(plan.digit[p].max) / static_cast<double>(n0);
      }
      if (expected <= 0.8 * (kWSpan - kWTile)) top = m;
    }
  }
  // Bit-aligned digits (partition path: the transformed keys are in keys_a).  A byte boundary is an arbitrary place to cut:
  // float64 in [0, 1) has a top byte that is all but constant (two values, one of them rare) and pays a whole pass for it.
  // With the top digit starting at the highest bit that varies -- read off the top active byte's histogram -- the same
  // runs need one pass less (four -> three for that column).  The four digits' exact histograms cost one more pass over
  // the keys with four LDS atomics each (k_hist_shifts); the plan is taken only if it saves a global pass.
  std::vector<int> pass_shift;          // the global passes' digit positions, lowest first
  std::vector<const uint32_t*> pass_db; // ... and their digit bases
  int low_shift = -1;                   // bit-aligned plan: the lowest bit the global passes sort by
  if (onesweep && allow_hybrid && !raw_try && !heavy_values && n0 >= hybrid_min && static_cast<int>(active.size()) >= 4 &&
      !getenv("AMDARROW_SORT_NO_HYBRID") && !getenv("AMDARROW_SORT_BYTE_DIGITS") && !min_env) {
    const int ptop = active.back();
    const unsigned orv = plan.digit[ptop].orv, andv = plan.digit[ptop].andv;  // over the digit values that occur
    const int T = 8 * ptop + (31 - __builtin_clz((orv ^ andv) | 1u));  // the highest varying bit
    if ((T + 1) % 8 != 0 && T >= 31) {
      HistShifts hs{};
      hs.n = 4;
      for (int k = 0; k < 4; ++k) hs.shift[k] = std::max(T + 1 - 8 * (k + 1), 0);
      auto* ghist2 = reinterpret_cast<unsigned int*>(bhist);          // (the three-kernel passes' block histograms: free on this path)
      auto* dbase2 = reinterpret_cast<uint32_t*>(bhist) + 4 * 256;
      AA_HIP(hipMemsetAsync(ghist2, 0, 4 * 4 * 256, ctx->stream));
      hipLaunchKernelGGL((k_hist_shifts<KeyT>), dim3(grid_t), dim3(kThreads), 0, ctx->stream, keys_a, n0, hs, ghist2);
      AA_HIP(hipGetLastError());
      PlanSummary p2;
      AA_TRY(ReadPlan<KeyT>(ctx, ghist2, 4, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t>(n0), static_cast<const unsigned long long*>(nullptr),
                            static_cast<const KeyT*>(nullptr), static_cast<const HistShifts*>(nullptr), &p2));
      int top_c = 0;
      double expected = static_cast<double>(n0);
      for (int m = 1; m <= 4 && top_c == 0; ++m) {
        expected *= static_cast<double>(p2.digit[m - 1].max) / static_cast<double>(n0);
        if (m >= 2 && hs.shift[m - 1] >= 16 && expected <= 0.8 * (kWSpan - kWTile)) top_c = m;  // (at least two digits left for the window)
      }
      if (top_c && (top == 0 || top_c < top)) {
        hipLaunchKernelGGL(k_digit_bases, dim3(4), dim3(kThreads), 0, ctx->stream, ghist2, dbase2);
        AA_HIP(hipGetLastError());
        top = top_c;
        low_shift = hs.shift[top_c - 1];
        for (int k = top_c - 1; k >= 0; --k) { pass_shift.push_back(hs.shift[k]); pass_db.push_back(dbase2 + 256 * k); }
      }
    }
  }
  if (!top && top_only) AA_TRY(ensure_full_hist());  // every active digit becomes a global pass: all their histograms
  WindowDigits wd{};
  KeyT top_mask = 0;
  const KeyT* window_keys = nullptr;
  const uint32_t* window_idx = nullptr;
  unsigned window_n = 0;
  if (top && low_shift >= 0) {  // bit-aligned plan
    const int hi_bit = pass_shift.back() + 8;
    top_mask = (hi_bit >= int(8 * sizeof(KeyT)) ? ~KeyT(0) : (KeyT(1) << hi_bit) - 1) & ~((KeyT(1) << low_shift) - 1);
    for (int sh = low_shift - 8; wd.n < 8; sh -= 8) {  // the crowded windows' LSD digits: everything below the global passes
      wd.shift[wd.n++] = std::max(sh, 0);               // (digits may overlap at the bottom: harmless for a stable LSD sort)
      if (sh <= 0) break;
    }
    std::reverse(wd.shift, wd.shift + wd.n);            // lowest first
    for (int sh = low_shift; sh < int(8 * sizeof(KeyT)) && wd.n < 10; sh += 8) wd.shift[wd.n++] = sh;  // (several runs in a crowded window: see SortRawClassified)
  } else if (top) {
    for (int p : active) wd.shift[wd.n++] = 8 * p;
    for (int q = 0; q < top; ++q) top_mask |= KeyT(255) << (8 * active[active.size() - 1 - q]);
    active.erase(active.begin(), active.end() - top);  // the global passes: the top digits, lowest first
  }
  if (low_shift < 0) {
    for (int p : active) { pass_shift.push_back(8 * p); pass_db.push_back(dbase + 256 * p); }
  }
  if (onesweep) {
    hipLaunchKernelGGL(k_digit_bases, dim3(kPasses), dim3(kThreads), 0, ctx->stream, ghist, dbase);
    AA_HIP(hipGetLastError());
    AA_HIP(hipMemsetAsync(ctl, 0, 256, ctx->stream));
    AA_HIP(hipMemsetAsync(lookback, 0, 4 * lb_words * pass_shift.size(), ctx->stream));
  }
  const bool raw_first = raw_try && onesweep && !(pass_shift.size() == 1 && !top);  // (the first pass is not the one that writes the result)
  if (raw_try && !raw_first) {  // the partition after all (its histograms are the ones already read)
    AA_HIP(hipMemsetAsync(ghist, 0, 4 * 256 * kPasses, ctx->stream));
    AA_TRY(count_and_scan());
    AA_TRY(run_partition());
    if (onesweep) { hipLaunchKernelGGL(k_digit_bases, dim3(kPasses), dim3(kThreads), 0, ctx->stream, ghist, dbase); AA_HIP(hipGetLastError()); }
  }
  for (size_t a = 0; a < pass_shift.size(); ++a) {
    const int shift = pass_shift[a];
    const bool last = a + 1 == pass_shift.size() && !top;
    if (onesweep && a == 0 && raw_first) {
      const uint32_t* db = pass_db[a];
      auto* pass = arank ? k_radix_onesweep<KeyT, false, true, true> : k_radix_onesweep<KeyT, false, false, true>;
      const KeyT* raw_src = reinterpret_cast<const KeyT*>(hc[0].values) + hc[0].off;
      hipLaunchKernelGGL(pass, dim3(static_cast<unsigned>((n0 + kOTile - 1) / kOTile)), dim3(kOThreads), 0, ctx->stream,
                         raw_src, static_cast<const uint32_t*>(nullptr), n0, shift, db, lookback,
                         ctl + 15, keys_b, idx_b, out, type_id, raw_xor, key_mask, XcdRunFor(ctx, reinterpret_cast<const void*>(pass)),
                         RawcArgs{nullptr, 0, nullptr, 0});
      AA_HIP(hipGetLastError());
      std::swap(keys_a, keys_b);
      std::swap(idx_a, idx_b);
      continue;
    }
    if (onesweep) {
      uint32_t* lb = lookback + lb_words * a;
      const uint32_t* db = pass_db[a];
      auto* pass = last ? (arank ? k_radix_onesweep<KeyT, true, true> : k_radix_onesweep<KeyT, true, false>)
                        : (arank ? k_radix_onesweep<KeyT, false, true> : k_radix_onesweep<KeyT, false, false>);
      hipLaunchKernelGGL(pass, dim3(static_cast<unsigned>((n0 + kOTile - 1) / kOTile)), dim3(kOThreads), 0, ctx->stream, keys_a, idx_a, n0, shift, db, lb,
                         ctl + 15, keys_b, idx_b, out, 0, KeyT(0), KeyT(0), XcdRunFor(ctx, reinterpret_cast<const void*>(pass)),
                         RawcArgs{nullptr, 0, nullptr, 0});
      AA_HIP(hipGetLastError());
      std::swap(keys_a, keys_b);
      std::swap(idx_a, idx_b);
      continue;
    }
    hipLaunchKernelGGL((k_radix_block_hist<KeyT>), dim3(static_cast<unsigned>(nb0)), dim3(kThreads), 0, ctx->stream, keys_a, n0,
                       shift, nb0, bhist);
    AA_HIP(hipGetLastError());
    AA_TRY(ScanU32InPlace(ctx, bhist, hn0, sums, sumoff));
    if (last)
      hipLaunchKernelGGL((k_radix_scatter<KeyT, true>), dim3(static_cast<unsigned>(nb0)), dim3(kThreads), 0, ctx->stream, keys_a,
                         idx_a, n0, shift, nb0, bhist, keys_b, idx_b, out);
    else
      hipLaunchKernelGGL((k_radix_scatter<KeyT, false>), dim3(static_cast<unsigned>(nb0)), dim3(kThreads), 0, ctx->stream, keys_a,
                         idx_a, n0, shift, nb0, bhist, keys_b, idx_b, out);
    AA_HIP(hipGetLastError());
    std::swap(keys_a, keys_b);
    std::swap(idx_a, idx_b);
  }
  if (top) {
    // (the window's own order never rests on the atomics' order: the bucket pass ties on the window position, and its
    // LSD fallback for crowded windows ranks with ballots)
    const unsigned nwin = static_cast<unsigned>((n0 + kWTile - 1) / kWTile);
    auto* crowded = reinterpret_cast<unsigned int*>(bhist);  // (the three-kernel passes' block histograms: unused on this path, 256 words per 4096 keys)
    AA_HIP(hipMemsetAsync(crowded, 0, 4 * static_cast<size_t>(nwin), ctx->stream));
    auto* wbounds = reinterpret_cast<ulonglong2*>(b + o_bounds);
    hipLaunchKernelGGL((k_window_bounds<KeyT>), dim3((nwin + 1 + 255) / 256), dim3(256), 0, ctx->stream, keys_a, n0, top_mask, int64_t(nwin), wbounds);
    if (ctx->sort_mark_ties) hipLaunchKernelGGL((k_window_sort<KeyT, true>), dim3(nwin), dim3(kWThreads), 0, ctx->stream, keys_a, idx_a, n0, top_mask, ctl + 14, crowded, out, wbounds);
    else hipLaunchKernelGGL((k_window_sort<KeyT>), dim3(nwin), dim3(kWThreads), 0, ctx->stream, keys_a, idx_a, n0, top_mask, ctl + 14, crowded, out, wbounds);
    AA_HIP(hipGetLastError());
    window_keys = keys_a; window_idx = idx_a; window_n = nwin;
  }
  if (onesweep) {
    unsigned int err[4] = {0, 0, 0, 0};  // [0] a run of equal top digits too long for the window, [1] a one-sweep pass gave up,
                                         // [2] a one-sweep tile whose atomic ranks were not in lane order, [3] crowded windows are waiting
    AA_TRY(ReadBack(ctx, ctl + 14, 16, err));
    if (err[3] && !err[0] && !err[1] && !err[2]) {
      hipLaunchKernelGGL((k_window_lsd<KeyT>), dim3(window_n), dim3(kWThreads), 0, ctx->stream, window_keys, window_idx, n0, top_mask, wd,
                         reinterpret_cast<const unsigned int*>(bhist), out, ctx->sort_mark_ties ? 1 : 0);
      AA_HIP(hipGetLastError());
    }
    if (err[2]) {  // never seen on gfx950; the device then ranks with ballots from here on
      ctx->lds_rank_ordered = 0;
      return SortImpl<KeyT>(ctx, chunks, nchunks, type_id, order, n, out, perm, allow_onesweep, allow_hybrid);
    }
    if (err[1]) return SortImpl<KeyT>(ctx, chunks, nchunks, type_id, order, n, out, perm, /*allow_onesweep=*/false, false);
    if (err[0]) return SortImpl<KeyT>(ctx, chunks, nchunks, type_id, order, n, out, perm, /*allow_onesweep=*/true, /*allow_hybrid=*/false);
  }
  if (top && ctx->sort_mark_ties) ctx->sort_ties_marked = true;  // (the window kernels wrote `out`, and nothing sent the sort another way)
  return AMDARROW_OK;
}

}  // namespace amdarrow

using namespace amdarrow;

#ifdef WS_PHASES
extern "C" int amdarrow_debug_sort_phases(unsigned long long* out16) {  // sums over the workgroups since the last call; [15] = workgroups
  unsigned long long h[64][16];
  AA_HIP(hipDeviceSynchronize());
  AA_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ws_phase), sizeof(h)));
  for (int i = 0; i < 16; ++i) { out16[i] = 0; for (int s = 0; s < 64; ++s) out16[i] += h[s][i]; }
  memset(h, 0, sizeof(h));
  AA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ws_phase), h, sizeof(h)));
  return AMDARROW_OK;
}
extern "C" int amdarrow_debug_onesweep_phases(unsigned long long* out16) {  // the same for k_radix_onesweep (all passes together)
  unsigned long long h[64][16];
  AA_HIP(hipDeviceSynchronize());
  AA_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_os_phase), sizeof(h)));
  for (int i = 0; i < 16; ++i) { out16[i] = 0; for (int s = 0; s < 64; ++s) out16[i] += h[s][i]; }
  memset(h, 0, sizeof(h));
  AA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_os_phase), h, sizeof(h)));
  return AMDARROW_OK;
}
#endif
extern "C" int amdarrow_sort_indices(amdarrow_ctx* ctx, const amdarrow_array* chunks, int nchunks, int order,
                                     uint64_t* out_indices) {
  if (nchunks < 0) return SetError(AMDARROW_INVALID, "sort_indices: negative chunk count");
  if (nchunks == 0) return AMDARROW_OK;
  const int type_id = chunks[0].type_id;
  int64_t n = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (chunks[c].type_id != type_id) return SetError(AMDARROW_TYPE_ERROR, "sort_indices: chunks differ in type");
    n += chunks[c].length;
  }
  const int bw = BitWidth(type_id);
  if (bw < 8) return SetError(AMDARROW_NOT_IMPLEMENTED, "sort_indices: numeric types only");  // vector_sort.cc:581-587
  if (n >= (int64_t(1) << 32)) return SetError(AMDARROW_NOT_IMPLEMENTED, "sort_indices: 2^32 rows or more");
  if (n == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  int rc = bw <= 32 ? SortImpl<uint32_t>(ctx, chunks, nchunks, type_id, order, n, out_indices)
                    : SortImpl<uint64_t>(ctx, chunks, nchunks, type_id, order, n, out_indices);
  AA_TRY(rc);
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}

namespace amdarrow {
// ---- composite keys for multi-key sorts ---------------------------------------------------------------------
// One stable sort per key (below) has to see key k in the order key k + 1's sort left the rows in: a uniformly random
// gather of 100 M keys, 2 ms however wide the key is (Take's bound: the lines requested from the fabric).  Integer
// key columns seldom need their type's whole range: with every column's minimum and maximum known (one streaming
// pass), the tuple (key 0, key 1, ...) of a row packs into ONE unsigned word -- field k = value - min (ascending) or
// max - value (descending), nulls the code behind the largest value in both orders (nulls last, :224-313), key 0 in
// the top bits -- and ONE stable single-key sort of that word is MultipleKeyComparator's order (:1049-1180).  Taken
// when every key is an integer column and the fields fit 64 bits together; anything else sorts key by key.
constexpr int kMaxCompositeKeys = 4;
struct CompositePiece {  // one key's rows of one segment (a run of rows that no key's chunk boundary cuts)
  const void* data; const uint8_t* valid; int64_t off;  // element / bit offset of the segment's first row
  uint64_t lo, hi;     // the column's range over its valid rows, in the orderable form below
  uint32_t width_bytes, is_signed, descending, shift;
  uint32_t rshift;     // leading-bits composites: the field's low bits that do not fit (>= 64: the whole field)
  uint64_t null_code;  // the field of a null row
  uint32_t is_float;   // float32 / float64: -0.0 is +0.0, NaNs are the word ~0 (no value's) and take nan_code, behind the values in both orders
  uint64_t nan_code;
};
struct CompositeArgs { CompositePiece key[kMaxCompositeKeys]; int nkeys; int64_t len; int64_t step; };  // (step > 1: the range kernel looks at every step-th row only)
// A wave's kU runs of 64 consecutive rows of one column as unsigned words whose order is the column's (signed:
// sign-extended, sign bit flipped).  The switch on the width stands OUTSIDE the loads: inside, every load would sit in a
// basic block of its own and be waited for before the next is issued.
__device__ __forceinline__ uint64_t CompositeFloatWord(uint64_t bits, uint32_t width_bytes) {
  if (width_bytes == 4) {
    uint32_t b = static_cast<uint32_t>(bits);
    if ((b << 1) == 0) b = 0;                                  // -0.0 == +0.0 under `<`
    if ((b & 0x7FFFFFFFu) > 0x7F800000u) return ~0ull;         // NaN
    return (b & 0x80000000u) ? static_cast<uint32_t>(~b) : (b | 0x80000000u);
  }
  if ((bits << 1) == 0) bits = 0;
  if ((bits & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull) return ~0ull;
  return (bits >> 63) ? ~bits : (bits | (1ull << 63));
}
template <int kU>
__device__ __forceinline__ void CompositeLoadRuns(const CompositePiece& k, const int64_t (&idx)[kU], uint64_t (&u)[kU], uint32_t (&vb)[kU]) {
  switch (k.width_bytes) {
    case 1:
#pragma unroll
      for (int j = 0; j < kU; ++j) u[j] = k.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int8_t*>(k.data)[k.off + idx[j]])) : static_cast<const uint8_t*>(k.data)[k.off + idx[j]];
      break;
    case 2:
#pragma unroll
      for (int j = 0; j < kU; ++j) u[j] = k.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int16_t*>(k.data)[k.off + idx[j]])) : static_cast<const uint16_t*>(k.data)[k.off + idx[j]];
      break;
    case 4:
#pragma unroll
      for (int j = 0; j < kU; ++j) u[j] = k.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int32_t*>(k.data)[k.off + idx[j]])) : static_cast<const uint32_t*>(k.data)[k.off + idx[j]];
      break;
    default:
#pragma unroll
      for (int j = 0; j < kU; ++j) u[j] = static_cast<const uint64_t*>(k.data)[k.off + idx[j]];
      break;
  }
  if (k.valid) {
#pragma unroll
    for (int j = 0; j < kU; ++j) vb[j] = k.valid[(k.off + idx[j]) >> 3];
  } else {
#pragma unroll
    for (int j = 0; j < kU; ++j) vb[j] = 0xFFu;
  }
  if (k.is_float) {  // (uniform) IEEE bits -> a word in the order of `<`: negatives complemented, the others above them
#pragma unroll
    for (int j = 0; j < kU; ++j) u[j] = CompositeFloatWord(u[j], k.width_bytes);
    return;
  }
  const uint64_t flip = k.is_signed ? 1ull << 63 : 0ull;
#pragma unroll
  for (int j = 0; j < kU; ++j) u[j] ^= flip;
}
// ranges[3 k ..]: {min, max, nulls} of key k over the segments seen so far (min starts at ~0, the others at 0).
__global__ __launch_bounds__(256) void k_composite_range(CompositeArgs a, unsigned long long* ranges) {
  constexpr int kU = 8;
  const int lane = threadIdx.x & 63;
  const int64_t stride = int64_t(gridDim.x) * 256 * kU;
  for (int k = 0; k < a.nkeys; ++k) {
    const CompositePiece key = a.key[k];
    uint64_t mn = ~0ull, mx = 0;
    uint32_t nulls = 0, nans = 0;
    for (int64_t base = (int64_t(blockIdx.x) * 256 + (threadIdx.x & ~63)) * kU; base < a.len; base += stride) {
      uint64_t u[kU]; uint32_t vb[kU]; int64_t idx[kU];
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const int64_t i = base + 64 * j + lane;
        idx[j] = (i < a.len ? i : a.len - 1) * a.step;
      }
      CompositeLoadRuns<kU>(key, idx, u, vb);
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const bool in = base + 64 * j + lane < a.len;
        const bool ok = (vb[j] >> ((key.off + idx[j]) & 7)) & 1u;
        const bool nan = key.is_float && u[j] == ~0ull;
        if (in && ok && !nan) { mn = u[j] < mn ? u[j] : mn; mx = u[j] > mx ? u[j] : mx; }
        nulls += in && !ok;
        nans += in && ok && nan;
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint64_t omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
      mn = omn < mn ? omn : mn; mx = omx > mx ? omx : mx;
      nulls += __shfl_xor(nulls, d);
      nans += __shfl_xor(nans, d);
    }
    // one set of atomics per workgroup: they all meet at the same three words, where the memory serves them one by one
    // (a set per wave: 40 K atomics, 0.3 ms of the kernel's 0.56)
    __shared__ uint64_t s_mn[4], s_mx[4];
    __shared__ uint32_t s_nulls[4], s_nans[4];
    __syncthreads();  // (the previous key's words have been read)
    if (lane == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; s_nulls[threadIdx.x >> 6] = nulls; s_nans[threadIdx.x >> 6] = nans; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) { mn = s_mn[w] < mn ? s_mn[w] : mn; mx = s_mx[w] > mx ? s_mx[w] : mx; nulls += s_nulls[w]; nans += s_nans[w]; }
      if (mn <= mx) { atomicMin(&ranges[3 * k], mn); atomicMax(&ranges[3 * k + 1], mx); }
      if (nulls | nans) atomicAdd(&ranges[3 * k + 2], static_cast<unsigned long long>(nulls) | static_cast<unsigned long long>(nans) << 32);  // (rows < 2^32)
    }
  }
}
template <typename OutT>
__global__ __launch_bounds__(256) void k_composite_build(CompositeArgs a, OutT* __restrict__ out) {
  constexpr int kU = 8;
  const int lane = threadIdx.x & 63;
  const int64_t stride = int64_t(gridDim.x) * 256 * kU;
  for (int64_t base = (int64_t(blockIdx.x) * 256 + (threadIdx.x & ~63)) * kU; base < a.len; base += stride) {
    int64_t idx[kU];
    uint64_t word[kU];
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t i = base + 64 * j + lane;
      idx[j] = i < a.len ? i : a.len - 1;
      word[j] = 0;
    }
    for (int k = 0; k < a.nkeys; ++k) {
      const CompositePiece& key = a.key[k];
      uint64_t u[kU]; uint32_t vb[kU];
      CompositeLoadRuns<kU>(key, idx, u, vb);
#pragma unroll
      for (int j = 0; j < kU; ++j) {
        const bool ok = (vb[j] >> ((key.off + idx[j]) & 7)) & 1u;
        const uint64_t field = !ok ? key.null_code : key.is_float && u[j] == ~0ull ? key.nan_code : key.descending ? key.hi - u[j] : u[j] - key.lo;
        if (key.rshift < 64) word[j] |= (field >> key.rshift) << key.shift;  // (uniform)
      }
    }
#pragma unroll
    for (int j = 0; j < kU; ++j) {
      const int64_t i = base + 64 * j + lane;
      if (i < a.len) out[i] = static_cast<OutT>(word[j]);
    }
  }
}
// ---- tuples wider than 64 bits: the leading 64 bits sorted, the ties on them put right -------------------------
// The composite of (int32 over 1 M values, int64 over 63 bits) needs 83 bits.  Sorting key by key gathers the earlier key
// through the later key's permutation (2 ms per 100 M rows).  Instead the LEADING 64 bits of the tuple are sorted as one key
// -- the hybrid sort's window kernels mark every row whose 64 bits equal another row's (bit 63 of its index) -- and
// k_fix_ties sorts every run of consecutive marked rows by the whole tuple (read from the columns: a few rows), then by
// row number.  The marked rows of a run are in the order of their leading bits already, which the whole tuple's order
// refines; a run longer than kMaxTieRun (heavily duplicated tuples) gives the call back to the key-by-key sort.
constexpr int kMaxTieRun = 64;
__device__ __forceinline__ uint64_t CompositeField(const CompositePiece& k, uint32_t row) {
  const int64_t i = k.off + row;
  if (k.valid && !((k.valid[i >> 3] >> (i & 7)) & 1u)) return k.null_code;
  uint64_t u;
  switch (k.width_bytes) {
    case 1: u = k.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int8_t*>(k.data)[i])) : static_cast<const uint8_t*>(k.data)[i]; break;
    case 2: u = k.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int16_t*>(k.data)[i])) : static_cast<const uint16_t*>(k.data)[i]; break;
    case 4: u = k.is_signed ? static_cast<uint64_t>(static_cast<int64_t>(static_cast<const int32_t*>(k.data)[i])) : static_cast<const uint32_t*>(k.data)[i]; break;
    default: u = static_cast<const uint64_t*>(k.data)[i]; break;
  }
  if (k.is_float) {
    u = CompositeFloatWord(u, k.width_bytes);
    if (u == ~0ull) return k.nan_code;
  } else if (k.is_signed) u ^= 1ull << 63;
  return k.descending ? k.hi - u : u - k.lo;
}
__global__ __launch_bounds__(256) void k_fix_ties(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int64_t n, CompositeArgs a,
                                                  unsigned int* fail) {
  for (int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const uint64_t v = in[i];
    if (!(v >> 63)) { out[i] = v; continue; }
    if (i > 0 && (in[i - 1] >> 63)) continue;  // inside a run: its first row's thread writes it
    uint32_t rows[kMaxTieRun];
    int len = 0;
    bool too_long = false;
    for (int64_t j = i; j < n; ++j) {
      const uint64_t w = in[j];
      if (!(w >> 63)) break;
      if (len == kMaxTieRun) { too_long = true; break; }
      rows[len++] = static_cast<uint32_t>(w);
    }
    if (too_long) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
    auto less = [&](uint32_t x, uint32_t y) {
      for (int k = 0; k < a.nkeys; ++k) {
        const uint64_t fx = CompositeField(a.key[k], x), fy = CompositeField(a.key[k], y);
        if (fx != fy) return fx < fy;
      }
      return x < y;
    };
    for (int t = 1; t < len; ++t) {  // insertion sort: runs are a handful of rows
      const uint32_t x = rows[t];
      int q = t;
      while (q > 0 && less(x, rows[q - 1])) { rows[q] = rows[q - 1]; --q; }
      rows[q] = x;
    }
    for (int t = 0; t < len; ++t) out[i + t] = rows[t];
  }
}
}  // namespace amdarrow

// Multi-key sort_indices (RadixRecordBatchSorter vector_sort.cc:955-1045 for batches,
// MultipleKeyTableSorter :1381-1549 for tables -- the same order, MultipleKeyComparator
// :1049-1180): one stable sort per key from the LAST key to the first.  Each stage visits the
// rows in the order the previous stage left them, so rows that tie on the earlier keys keep the
// order of the later ones; per key the classes are values (by that key's order), NaNs, nulls.
// `chunks` lists key 0's chunks, then key 1's, ...; every key spans the same number of rows.
extern "C" int amdarrow_sort_indices_multi(amdarrow_ctx* ctx, const amdarrow_array* chunks, const int* nchunks,
                                           const int* orders, int nkeys, uint64_t* out_indices) {
  if (nkeys < 1) return SetError(AMDARROW_INVALID, "Must specify one or more sort keys");  // vector_sort.cc:1634-1636
  std::vector<const amdarrow_array*> first(nkeys);
  int64_t n = -1;
  const amdarrow_array* p = chunks;
  for (int k = 0; k < nkeys; ++k) {
    if (nchunks[k] < 1) return SetError(AMDARROW_INVALID, "sort_indices: a sort key without chunks");
    first[k] = p;
    int64_t len = 0;
    for (int c = 0; c < nchunks[k]; ++c) {
      if (p[c].type_id != p[0].type_id) return SetError(AMDARROW_TYPE_ERROR, "sort_indices: chunks differ in type");
      len += p[c].length;
    }
    if (BitWidth(p[0].type_id) < 8) return SetError(AMDARROW_NOT_IMPLEMENTED, "sort_indices: numeric types only");
    if (n >= 0 && len != n) return SetError(AMDARROW_INVALID, "sort_indices: sort keys differ in length");
    n = len;
    p += nchunks[k];
  }
  if (n >= (int64_t(1) << 32)) return SetError(AMDARROW_NOT_IMPLEMENTED, "sort_indices: 2^32 rows or more");
  if (n == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  // ---- every key an integer column and the tuple fits one word: one sort of the composite key (see above)
  bool all_int = nkeys >= 2 && nkeys <= kMaxCompositeKeys && !getenv("AMDARROW_SORT_NO_COMPOSITE");
  for (int k = 0; k < nkeys && all_int; ++k) all_int = IsSignedInt(first[k][0].type_id) || IsUnsignedInt(first[k][0].type_id) || IsFloat(first[k][0].type_id);
  if (all_int) {
    // segments: runs of rows inside one chunk of every key
    std::vector<int64_t> cuts{0, n};
    for (int k = 0; k < nkeys; ++k) {
      int64_t at = 0;
      for (int c = 0; c < nchunks[k]; ++c) { at += first[k][c].length; cuts.push_back(at); }
    }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    std::vector<CompositeArgs> segs;
    std::vector<int> cur(nkeys, 0);
    std::vector<int64_t> cur_start(nkeys, 0);
    for (size_t si = 0; si + 1 < cuts.size(); ++si) {
      CompositeArgs a{};
      a.nkeys = nkeys; a.len = cuts[si + 1] - cuts[si]; a.step = 1;
      for (int k = 0; k < nkeys; ++k) {
        while (cuts[si] >= cur_start[k] + first[k][cur[k]].length) { cur_start[k] += first[k][cur[k]].length; ++cur[k]; }
        const amdarrow_array& ch = first[k][cur[k]];
        CompositePiece& pc = a.key[k];
        pc.data = ch.values; pc.valid = MayHaveNulls(&ch) ? ch.validity : nullptr; pc.off = ch.offset + (cuts[si] - cur_start[k]);
        pc.width_bytes = BitWidth(ch.type_id) / 8; pc.is_signed = IsSignedInt(ch.type_id); pc.descending = orders[k] == AMDARROW_DESCENDING;
        pc.is_float = IsFloat(ch.type_id);
      }
      segs.push_back(a);
    }
    void* base;
    AA_TRY(Scratch(ctx, 256, &base));
    auto* ranges = static_cast<unsigned long long*>(base);
    unsigned long long init[3 * kMaxCompositeKeys];
    for (int k = 0; k < kMaxCompositeKeys; ++k) { init[3 * k] = ~0ull; init[3 * k + 1] = 0; init[3 * k + 2] = 0; }
    memcpy(static_cast<char*>(ctx->pinned) + kStatusOffset, init, sizeof(init));
    AA_HIP(hipMemcpyAsync(ranges, static_cast<char*>(ctx->pinned) + kStatusOffset, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    unsigned long long got[3 * kMaxCompositeKeys];
    // A sample first (every step-th row, 64 K rows of the longest segment): its ranges are lower bounds, and columns
    // that are plainly too wide together are found without a pass over them.
    bool fits = true, lead = false;
    if (n >= (int64_t(1) << 22)) {
      const CompositeArgs* longest = &segs[0];
      for (const CompositeArgs& a : segs) if (a.len > longest->len) longest = &a;
      CompositeArgs sm = *longest;
      sm.step = sm.len >> 16;
      sm.len = sm.len / sm.step;
      hipLaunchKernelGGL(k_composite_range, dim3(GridFor(sm.len, 256 * 8, ctx->num_cus * 4)), dim3(256), 0, ctx->stream, sm, ranges);
      AA_HIP(hipGetLastError());
      AA_TRY(ReadBack(ctx, ranges, sizeof(got), got));
      int bits = 0;
      for (int k = 0; k < nkeys; ++k)
        if (got[3 * k] < got[3 * k + 1]) bits += 64 - __builtin_clzll(got[3 * k + 1] - got[3 * k]);
      fits = bits <= 64;  // (the sample's minima and maxima stay in `ranges`: they are the column's own values)
      // wider tuples: their leading 64 bits as one key, the ties put right afterwards (k_fix_ties); one chunk per key
      lead = !fits && bits <= 128 && segs.size() == 1 && n < (int64_t(1) << 31) && !getenv("AMDARROW_SORT_NO_LEADING_BITS");
    }
    if (fits || lead) {
      for (const CompositeArgs& a : segs)
        hipLaunchKernelGGL(k_composite_range, dim3(GridFor(a.len, 256 * 8, ctx->num_cus * 4)), dim3(256), 0, ctx->stream, a, ranges);
      AA_HIP(hipGetLastError());
      AA_TRY(ReadBack(ctx, ranges, sizeof(got), got));
    }
    int total_bits = 0;
    int width[kMaxCompositeKeys];
    for (int k = 0; k < nkeys && (fits || lead); ++k) {
      const bool any_valid = got[3 * k] <= got[3 * k + 1];
      const uint64_t span = any_valid ? got[3 * k + 1] - got[3 * k] : 0;  // the largest value field
      const bool nulls = (got[3 * k + 2] & 0xFFFFFFFFull) != 0, nans = (got[3 * k + 2] >> 32) != 0;
      if (span > ~0ull - 2) { fits = lead = false; break; }
      const uint64_t top = span + (nans ? 1 : 0) + (nulls ? 1 : 0);     // the largest field: values, then NaNs, then nulls
      width[k] = top == 0 ? 0 : 64 - __builtin_clzll(top);
      total_bits += width[k];
      if (total_bits > 64) { if (fits) lead = segs.size() == 1 && n >= (int64_t(1) << 22) && n < (int64_t(1) << 31) && !getenv("AMDARROW_SORT_NO_LEADING_BITS"); fits = false; }
      if (total_bits > 128) lead = false;
      for (CompositeArgs& a : segs) {
        a.key[k].rshift = 0;
        a.key[k].lo = any_valid ? got[3 * k] : 0; a.key[k].hi = any_valid ? got[3 * k + 1] : 0;
        a.key[k].nan_code = span + 1;                    // (only ever used when NaNs exist)
        a.key[k].null_code = span + 1 + (nans ? 1 : 0);  // (only ever used when nulls exist)
      }
    }
    if (fits) {
      int shift = total_bits;
      for (int k = 0; k < nkeys; ++k) {
        shift -= width[k];
        for (CompositeArgs& a : segs) a.key[k].shift = static_cast<uint32_t>(shift & 63);  // (a field of width 0 is always 0)
      }
      const bool narrow = total_bits <= 32;
      void* comp = nullptr;
      AA_TRY(amdarrow_malloc(ctx, (narrow ? 4 : 8) * n, &comp));
      int64_t row0 = 0;
      for (const CompositeArgs& a : segs) {
        const dim3 grid(GridFor(a.len, 256 * 8, ctx->num_cus * 8));
        if (narrow) hipLaunchKernelGGL(k_composite_build<uint32_t>, grid, dim3(256), 0, ctx->stream, a, static_cast<uint32_t*>(comp) + row0);
        else hipLaunchKernelGGL(k_composite_build<uint64_t>, grid, dim3(256), 0, ctx->stream, a, static_cast<uint64_t*>(comp) + row0);
        row0 += a.len;
      }
      amdarrow_array arr{};
      arr.values = comp; arr.validity = nullptr; arr.length = n; arr.offset = 0; arr.null_count = 0;
      arr.type_id = narrow ? AMDARROW_UINT32 : AMDARROW_UINT64;
      int rc = hipGetLastError() == hipSuccess ? AMDARROW_OK : SetError(AMDARROW_UNKNOWN_ERROR, "sort_indices: composite key launch failed");
      if (rc == AMDARROW_OK)
        rc = narrow ? SortImpl<uint32_t>(ctx, &arr, 1, arr.type_id, AMDARROW_ASCENDING, n, out_indices)
                    : SortImpl<uint64_t>(ctx, &arr, 1, arr.type_id, AMDARROW_ASCENDING, n, out_indices);
      const hipError_t e = hipStreamSynchronize(ctx->stream);
      (void)amdarrow_free(ctx, comp);
      AA_TRY(rc);
      AA_HIP(e);
      return AMDARROW_OK;
    }
    if (lead && total_bits > 64 && total_bits <= 128) {
      const int drop = total_bits - 64;  // the tuple's low bits the sorted key leaves out
      CompositeArgs cut = segs[0];       // the fields cut to the leading 64 bits; segs[0] keeps the whole fields for k_fix_ties
      int below = total_bits;
      for (int k = 0; k < nkeys; ++k) {
        below -= width[k];               // bits of the tuple below field k
        segs[0].key[k].shift = 0;
        if (below >= drop) { cut.key[k].shift = static_cast<uint32_t>(below - drop); cut.key[k].rshift = 0; }
        else if (below + width[k] > drop) { cut.key[k].shift = 0; cut.key[k].rshift = static_cast<uint32_t>(drop - below); }
        else { cut.key[k].shift = 0; cut.key[k].rshift = 64; }
      }
      void* comp = nullptr;
      uint64_t* marked = nullptr;
      AA_TRY(amdarrow_malloc(ctx, 8 * n + 64, &comp));
      int rc = amdarrow_malloc(ctx, 8 * n, reinterpret_cast<void**>(&marked));
      auto* fail = reinterpret_cast<unsigned int*>(static_cast<char*>(comp) + 8 * n);
      if (rc == AMDARROW_OK) {
        if (hipMemsetAsync(fail, 0, 64, ctx->stream) != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, "sort_indices: composite key setup failed");
        hipLaunchKernelGGL(k_composite_build<uint64_t>, dim3(GridFor(n, 256 * 8, ctx->num_cus * 8)), dim3(256), 0, ctx->stream, cut, static_cast<uint64_t*>(comp));
        if (hipGetLastError() != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, "sort_indices: composite key launch failed");
      }
      bool done = false;
      if (rc == AMDARROW_OK) {
        amdarrow_array arr{};
        arr.values = comp; arr.validity = nullptr; arr.length = n; arr.offset = 0; arr.null_count = 0; arr.type_id = AMDARROW_UINT64;
        ctx->sort_mark_ties = true; ctx->sort_ties_marked = false;
        rc = SortImpl<uint64_t>(ctx, &arr, 1, arr.type_id, AMDARROW_ASCENDING, n, marked);
        ctx->sort_mark_ties = false;
        if (rc == AMDARROW_OK && ctx->sort_ties_marked) {
          hipLaunchKernelGGL(k_fix_ties, dim3(GridFor(n, 256, ctx->num_cus * 16)), dim3(256), 0, ctx->stream, marked, out_indices, n, segs[0], fail);
          unsigned int failed = 1;
          if (hipGetLastError() != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, "sort_indices: tie fix-up launch failed");
          else rc = ReadBack(ctx, fail, 4, &failed);
          done = rc == AMDARROW_OK && failed == 0;
        }
      }
      const hipError_t e = hipStreamSynchronize(ctx->stream);
      if (marked) (void)amdarrow_free(ctx, marked);
      (void)amdarrow_free(ctx, comp);
      AA_TRY(rc);
      AA_HIP(e);
      if (getenv("AMDARROW_SORT_TRACE"))  // (diagnostics)
        fprintf(stderr, "sort_indices_multi: %d bits, leading 64 sorted, ties marked %d, fixed up %d\n", total_bits, int(ctx->sort_ties_marked), int(done));
      if (done) return AMDARROW_OK;  // (else: no hybrid sort for these keys, or a run of ties too long -- key by key)
    }
  }
  // the stages ping-pong between `out_indices` and one temporary so that the last lands in out_indices
  // (from the context's pool, which keeps freed blocks: a hipMalloc / hipFree of 0.8 GB per call cost milliseconds, now and
  // then a hundred -- one step in ten of the two-key bench took 170 ms)
  uint64_t* tmp = nullptr;
  if (nkeys > 1) AA_TRY(amdarrow_malloc(ctx, 8 * n, reinterpret_cast<void**>(&tmp)));
  int rc = AMDARROW_OK;
  const uint64_t* perm = nullptr;
  for (int k = nkeys - 1; k >= 0 && rc == AMDARROW_OK; --k) {
    uint64_t* dst = (k % 2 == 0) ? out_indices : tmp;
    const int type_id = first[k][0].type_id;
    rc = BitWidth(type_id) <= 32 ? SortImpl<uint32_t>(ctx, first[k], nchunks[k], type_id, orders[k], n, dst, perm)
                                 : SortImpl<uint64_t>(ctx, first[k], nchunks[k], type_id, orders[k], n, dst, perm);
    perm = dst;
  }
  const hipError_t e = hipStreamSynchronize(ctx->stream);
  if (tmp) (void)amdarrow_free(ctx, tmp);
  AA_TRY(rc);
  AA_HIP(e);
  return AMDARROW_OK;
}

// selection.hip -- Filter / GetTakeIndices / Take for primitive (1/8/16/32/64-bit) arrays.
//
// Reference semantics: compute/kernels/vector_selection.cc
//   GetFilterOutputSize :61-88, GetTakeIndices :92-237, PrimitiveTakeImpl :273-369,
//   BooleanTakeImpl :371-468, PrimitiveFilterImpl :569-784, PrimitiveFilter :786-835.
//
// Filter is a three-kernel stream compaction, HBM-bound on the value column:
//   (1) k_filter_tile_counts : one wave per 4096-row tile pops 64 filter words
//   (2) k_scan_tiles         : one workgroup, exclusive scan of the tile counts
//   (3) k_filter_compact     : per tile, ranks come from the 64 emit words (wave scan of
//       their popcounts + mbcnt inside the word), selected values are staged in LDS at
//       their rank and leave as one coalesced run; validity (and boolean values) are
//       compacted through LDS byte flags -> wave ballot -> funnel-shift onto the output
//       word grid; only the two edge words of a tile use a global atomicOr.
// Values are loaded only for selected rows, so a 10 % filter does not read 100 % of HBM.
#include "common.h"

#include <cstring>

namespace amdarrow {

constexpr int kTile = 4096;     // rows per tile = 64 bitmap words
constexpr int kThreads = 256;   // 4 waves
constexpr int kCThreads = 256;  // compaction workgroup
constexpr int kCWaves = kCThreads / 64;

// A tile's two edge words sit in a 128-byte line of their own: neighbouring tiles run on different XCDs, and a line
// shared by several of them would leave each L2 as a partial write.
constexpr int kEdgeStride = 16;

struct FilterArgs {
  // filter
  const uint8_t* f_data; const uint8_t* f_valid; int64_t f_off; int emit_null;
  // values
  const void* v_data; const uint8_t* v_valid; int64_t v_off;
  int64_t length; int64_t ntiles;
  // outputs
  void* out_values; uint8_t* out_validity; int64_t out_capacity; int has_validity;
  const int64_t* tile_offsets;  // exclusive scan, ntiles + 1 entries
  unsigned long long* edge_words;  // [ntiles][kEdgeStride], two used: what a tile of >= 64 output rows adds to the validity words it shares with its neighbours
};

// emit = rows that produce an output slot; sel = rows whose value is copied.
//   DROP      : emit = sel = data & valid                       (:540-552, NextAndWord)
//   EMIT_NULL : emit = data | ~valid, sel = data & valid        (:70-75, :700-709)
__device__ __forceinline__ void FilterWords(const uint8_t* f_data, const uint8_t* f_valid, int64_t f_off,
                                            int emit_null, int64_t row, int nbits, uint64_t* emit, uint64_t* sel) {
  const uint64_t d = LoadBits64(f_data, f_off + row, nbits);
  const uint64_t v = LoadBitsOrOnes(f_valid, f_off + row, nbits);
  const uint64_t live = nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0);
  *sel = d & v;
  *emit = emit_null ? ((d | ~v) & live) : (d & v);
}

__global__ __launch_bounds__(kThreads) void k_filter_tile_counts(const uint8_t* f_data, const uint8_t* f_valid,
                                                                 int64_t f_off, int emit_null, int64_t length,
                                                                 int64_t ntiles, uint32_t* counts) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t tile = int64_t(blockIdx.x) * 4 + wave; tile < ntiles; tile += int64_t(gridDim.x) * 4) {
    const int64_t row = tile * kTile + int64_t(lane) * 64;
    int64_t left = length - row;
    const int nbits = left >= 64 ? 64 : (left > 0 ? static_cast<int>(left) : 0);
    uint64_t emit = 0, sel = 0;
    if (nbits > 0) FilterWords(f_data, f_valid, f_off, emit_null, row, nbits, &emit, &sel);
    uint64_t c = WaveReduceSum64(static_cast<uint64_t>(__popcll(emit)));
    if (lane == 0) counts[tile] = static_cast<uint32_t>(c);
  }
}

// Single-workgroup exclusive scan: offsets[i] = sum(counts[0..i)), offsets[n] = total.
// Rounds of 8192 entries: a thread owns 8 consecutive ones (two 16-byte loads, one 64-byte run
// of int64 stores), the 1024 partial sums are scanned across the block.  `host_total`
// (optional, mapped host memory) receives the total as well, see StatusWords in common.h.
// With up to kScanParallelChunks rounds the rounds run as that many workgroups instead of one after the other: workgroup
// b first adds up everything before its round (at most 24 k counts, coalesced) -- a round's barriers and dependent
// shuffles cost ~5 us, three of them in a row are half the scan of a 100 M-row filter's 24 k tile counts.
constexpr int kScanParallelChunks = 4;
__global__ __launch_bounds__(1024) void k_scan_tiles(const uint32_t* counts, int64_t n, int64_t* offsets,
                                                     long long* host_total) {
  constexpr int kPer = 8;
  __shared__ uint64_t s_wave[16];
  __shared__ uint64_t s_before[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint64_t carry = 0;  // identical in every thread
  const int64_t first = int64_t(blockIdx.x) * 1024 * kPer;  // (gridDim.x == 1: the whole input, round by round)
  const int64_t last = gridDim.x == 1 ? n : (first + 1024 * kPer < n ? first + 1024 * kPer : n);
  if (blockIdx.x > 0) {
    uint64_t mine = 0;
    for (int64_t i = tid; i < first; i += 1024) mine += counts[i];
    mine = WaveReduceSum64(mine);
    if (lane == 0) s_before[wave] = mine;
    __syncthreads();
    for (int w = 0; w < 16; ++w) carry += s_before[w];
  }
  for (int64_t base = first; base < last; base += 1024 * kPer) {
    const int64_t i0 = base + int64_t(tid) * kPer;
    uint32_t v[kPer];
    if (i0 + kPer <= n && (reinterpret_cast<uintptr_t>(counts + i0) & 15) == 0) {
      const uint4 a = *reinterpret_cast<const uint4*>(counts + i0), b = *reinterpret_cast<const uint4*>(counts + i0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < kPer; ++k) v[k] = i0 + k < n ? counts[i0 + k] : 0u;
    }
    uint64_t sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) sum += v[k];
    uint64_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint64_t wave_off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const uint64_t x = s_wave[w]; if (w < wave) wave_off += x; total += x; }
    uint64_t run = carry + wave_off + incl - sum;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      if (i0 + k < n) offsets[i0 + k] = static_cast<int64_t>(run);
      run += v[k];
    }
    carry += total;
    __syncthreads();
  }
  if (tid == 0 && blockIdx.x == gridDim.x - 1) {
    offsets[n] = static_cast<int64_t>(carry);
    if (host_total) *host_total = static_cast<long long>(carry);
  }
}
static unsigned ScanTilesGrid(int64_t n) {
  const int64_t chunks = (n + 8191) / 8192;
  return chunks >= 2 && chunks <= kScanParallelChunks ? static_cast<unsigned>(chunks) : 1u;
}

int LaunchScanU32(amdarrow_ctx* ctx, const uint32_t* counts, int64_t n, int64_t* offsets) {
  hipLaunchKernelGGL(k_scan_tiles, dim3(ScanTilesGrid(n)), dim3(1024), 0, ctx->stream, counts, n, offsets,
                     static_cast<long long*>(nullptr));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

__device__ __forceinline__ void DepositPacked(const uint64_t* s_packed, uint32_t count, uint8_t* bm, int64_t bit_base,
                                              unsigned long long* edge);
// The bits of `x` at the positions of the set bits of `m`, packed to the low end (a software PEXT).  Loops over the
// MINORITY value among the selected bits -- a validity word is mostly ones -- clearing (or setting) the bit at the
// rank of each: a handful of iterations per word for the usual few per cent of nulls, 32 at worst.
__device__ __forceinline__ uint64_t CompressBits(uint64_t x, uint64_t m) {
  const int n = __popcll(m);
  const uint64_t ones = x & m, zeros = ~x & m;
  const bool few_zeros = __popcll(zeros) <= __popcll(ones);
  uint64_t rest = few_zeros ? zeros : ones, flip = 0;
  while (rest) {
    const uint64_t low = rest & (0 - rest);
    flip |= uint64_t(1) << __popcll(m & (low - 1));
    rest ^= low;
  }
  const uint64_t all = n >= 64 ? ~uint64_t(0) : (uint64_t(1) << n) - 1;
  return few_zeros ? all & ~flip : flip;
}

// Pack `count` LDS byte flags into the global bitmap `bm` at bit position `bit_base`.
// `bm` has been zeroed; interior words are plain stores, the (at most two) words shared
// with neighbouring tiles are merged with atomicOr.  Called by the whole workgroup.
// `edge` (optional, two words): a tile of at least 64 bits shares its first and / or last word with exactly one
// neighbour each; instead of merging them with memory-side atomics (an atomic drops its line from the L2: every
// tile's first and last bitmap line then leaves early and partial, 33 us per 100 M rows) it hands its share of
// them over in edge[0] / edge[1] and k_filter_merge_edges stores the words whole.
__device__ __forceinline__ void DepositFlags(const uint8_t* s_flags, uint64_t* s_packed, uint32_t count,
                                             uint8_t* bm, int64_t bit_base, unsigned long long* edge = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t nlw = (count + 63) >> 6;
  for (uint32_t lw = wave; lw < nlw; lw += blockDim.x / 64) {
    const uint32_t i = lw * 64 + lane;
    const uint64_t b = __ballot(i < count && s_flags[i] != 0);
    if (lane == 0) s_packed[lw] = b;
  }
  __syncthreads();
  DepositPacked(s_packed, count, bm, bit_base, edge);
}

// The `count` bits packed in s_packed[0 .. (count + 63) / 64) -> the global bitmap at bit position `bit_base` (see
// DepositFlags).  Called by the whole workgroup, the packed words complete (a barrier behind their last write).
__device__ __forceinline__ void DepositPacked(const uint64_t* s_packed, uint32_t count, uint8_t* bm, int64_t bit_base,
                                              unsigned long long* edge) {
  const int tid = threadIdx.x;
  const uint32_t nlw = (count + 63) >> 6;
  const int s = static_cast<int>(bit_base & 63);
  const int64_t g0 = bit_base >> 6;
  const uint32_t ngw = (s + count + 63) >> 6;
  unsigned long long* words = reinterpret_cast<unsigned long long*>(bm);
  for (uint32_t i = tid; i < ngw; i += blockDim.x) {
    uint64_t w = (i < nlw) ? (s_packed[i] << s) : 0;
    if (s > 0 && i > 0) w |= s_packed[i - 1] >> (64 - s);
    const bool owned = (i > 0 || s == 0) && (uint64_t(i + 1) * 64 <= uint64_t(s) + count);
    if (owned) words[g0 + i] = w;
    else if (edge && count >= 64) edge[i == 0 ? 0 : 1] = w;
    else if (w) atomicOr(&words[g0 + i], static_cast<unsigned long long>(w));
  }
  if (edge && tid < 2) {  // a side without a shared word adds nothing
    const bool has_head = s > 0, has_tail = ((s + count) & 63) != 0 && ngw > 1;
    if (count < 64 || (tid == 0 ? !has_head : !has_tail)) edge[tid] = 0;
  }
}

// One thread per tile boundary: the validity word in which tile t ends and tile t + 1 begins.  Two tiles of >= 64
// output rows: nobody else touches the word, it is stored whole.  A smaller tile ORs its own bits in with an
// atomic (k_filter_compact), so the big neighbour's share is ORed in as well.
__global__ __launch_bounds__(256) void k_filter_merge_edges(const int64_t* tile_offsets, int64_t ntiles,
                                                            const unsigned long long* edge_words, uint8_t* out_validity,
                                                            int64_t out_capacity) {
  const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (t >= ntiles) return;
  const int64_t b = tile_offsets[t], e = tile_offsets[t + 1];
  if ((e & 63) == 0 || e > out_capacity) return;
  const int64_t e2 = t + 1 < ntiles ? tile_offsets[t + 2] : e;
  const bool big = e - b >= 64, big_next = e2 - e >= 64 && e2 <= out_capacity;
  unsigned long long w = 0;
  if (big) w |= edge_words[kEdgeStride * t + 1];
  if (big_next) w |= edge_words[kEdgeStride * (t + 1)];
  unsigned long long* words = reinterpret_cast<unsigned long long*>(out_validity);
  if (big && (big_next || t + 1 == ntiles || e2 == e)) {
    // (the last tile with output: nothing follows in the word)
    bool alone = big_next || t + 1 == ntiles;
    if (!alone) { alone = true; for (int64_t u = t + 1; u < ntiles; ++u) if (tile_offsets[u + 1] != e) { alone = false; break; } }
    if (alone) { words[e >> 6] = w; return; }
  }
  if (w) atomicOr(&words[e >> 6], w);
}

template <int W> struct UIntOf;
template <> struct UIntOf<1> { using T = uint8_t; };
template <> struct UIntOf<2> { using T = uint16_t; };
template <> struct UIntOf<4> { using T = uint32_t; };
template <> struct UIntOf<8> { using T = uint64_t; };

// W = value bytes (1,2,4,8); W == 0 selects bit-packed (boolean) values.
// IOTA: the "value" of row r is r itself (GetTakeIndices).
template <int W, bool IOTA>
__global__ __launch_bounds__(kCThreads) void k_filter_compact(FilterArgs a) {
  using T = typename UIntOf<(W == 0 ? 1 : W)>::T;
  constexpr bool kBits = (W == 0);
  __shared__ uint64_t s_emit[64], s_sel[64], s_vvalid[64], s_bvals[64];
  __shared__ uint32_t s_base[64];
  __shared__ uint32_t s_count;
  __shared__ uint64_t s_packed[65];
  __shared__ __attribute__((aligned(16))) T s_stage[kBits ? kTile : 1];  // value-bit flags by rank
  __shared__ uint8_t s_vflag[kBits ? kTile : 1];                // validity flags by rank (bit-packed values only: fixed-width values pack their validity at word level)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* values = reinterpret_cast<const T*>(a.v_data);

  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * kTile;
    if (tid < 64) {
      const int64_t row = row0 + int64_t(tid) * 64;
      const int64_t left = a.length - row;
      const int nbits = left >= 64 ? 64 : (left > 0 ? static_cast<int>(left) : 0);
      uint64_t emit = 0, sel = 0, vv = 0, bv = 0;
      if (nbits > 0) {
        FilterWords(a.f_data, a.f_valid, a.f_off, a.emit_null, row, nbits, &emit, &sel);
        if (a.has_validity) vv = LoadBitsOrOnes(a.v_valid, a.v_off + row, nbits);
        if (kBits && !IOTA) bv = LoadBits64(reinterpret_cast<const uint8_t*>(a.v_data), a.v_off + row, nbits);
      }
      s_emit[tid] = emit; s_sel[tid] = sel; s_vvalid[tid] = vv; s_bvals[tid] = bv;
      const uint32_t pc = __popcll(emit);
      const uint32_t incl = WaveInclusiveScan(pc);
      s_base[tid] = incl - pc;
      if (tid == 63) s_count = incl;
      if (!kBits && a.has_validity) {
        // the output validity of this word's emitted rows (WriteNotNull / WriteMaybeNull / WriteNull, :646-657, :706-709),
        // packed at word level and ORed into the tile's packed bits at the word's rank: no per-row flags, no ballot
        // pass, two barriers per tile less
        s_packed[tid] = 0;  // (LDS operations of one wave execute in order: cleared before any lane's OR below)
        if (tid == 0) s_packed[64] = 0;
        const uint64_t cv = CompressBits(sel & vv, emit);
        const uint32_t at = incl - pc, sh = at & 63u;
        if (pc) {
          atomicOr(reinterpret_cast<unsigned long long*>(&s_packed[at >> 6]), static_cast<unsigned long long>(cv << sh));
          if (sh + pc > 64) atomicOr(reinterpret_cast<unsigned long long*>(&s_packed[(at >> 6) + 1]), static_cast<unsigned long long>(cv >> (64 - sh)));
        }
      }
    }
    __syncthreads();
    const uint32_t count = s_count;
    const int64_t out_base = a.tile_offsets[tile];
    const bool fits = out_base + count <= a.out_capacity;  // never write past the caller's buffers

    // ---- gather selected values (loads only where the value is copied) ----
    constexpr int kIters = kTile / kCThreads;
    T vals[kIters];
    if (!IOTA && !kBits && count * 8 >= kTile) {
      // dense tile: every 64-byte sector holds a selected row anyway, so load all rows --
      // 16 independent loads in flight instead of 16 branches around one load each
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        const int64_t row = row0 + (j * kCWaves + wave) * 64 + lane;
        vals[j] = values[a.v_off + (row < a.length ? row : a.length - 1)];
      }
#pragma unroll
      for (int j = 0; j < kIters; ++j)
        if (!((s_sel[j * kCWaves + wave] >> lane) & 1)) vals[j] = 0;  // emitted-null slots carry 0
    } else {
#pragma unroll
      for (int j = 0; j < kIters; ++j) {
        const int word = j * kCWaves + wave;
        const bool s = (s_sel[word] >> lane) & 1;
        const int64_t row = row0 + word * 64 + lane;
        T v = 0;
        if (s) {
          if (IOTA) v = static_cast<T>(row);
          else if (kBits) v = static_cast<T>((s_bvals[word] >> lane) & 1);
          else v = values[a.v_off + row];
        }
        vals[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const int word = j * kCWaves + wave;
      const uint64_t emit = s_emit[word];
      if ((emit >> lane) & 1) {
        const uint32_t rank = s_base[word] + RankBelowLane(emit);
        // emitted-null slots carry 0 (:743-746).  Fixed-width values go straight to their output
        // slot: a wave's selected lanes hold consecutive ranks, so its stores cover one contiguous
        // run, and without a value stage the workgroup needs 3 KB of LDS instead of 35.
        if (!kBits) { if (fits) reinterpret_cast<T*>(a.out_values)[out_base + rank] = vals[j]; }
        else s_stage[rank] = vals[j];
        if (kBits && a.has_validity) {
          // WriteNotNull / WriteMaybeNull / WriteNull (:646-657, :706-709)
          s_vflag[rank] = static_cast<uint8_t>(((s_sel[word] & s_vvalid[word]) >> lane) & 1);
        }
      }
    }
    if (kBits) __syncthreads();
    if (!fits && a.edge_words && tid < 2) a.edge_words[kEdgeStride * tile + tid] = 0;
    if (fits) {
      if (kBits) {
        DepositFlags(reinterpret_cast<const uint8_t*>(s_stage), s_packed, count,
                     reinterpret_cast<uint8_t*>(a.out_values), out_base);
        __syncthreads();
        if (a.has_validity) DepositFlags(s_vflag, s_packed, count, a.out_validity, out_base, a.edge_words ? a.edge_words + kEdgeStride * tile : nullptr);
      } else if (a.has_validity) {
        DepositPacked(s_packed, count, a.out_validity, out_base, a.edge_words ? a.edge_words + kEdgeStride * tile : nullptr);  // (packed by the first wave, before the barrier above)
      }
    }
    __syncthreads();
  }
}

// ---- Concatenate (array/concatenate.cc:70-92, 171-201) ---------------------------------------
// A bitmap made of the bit ranges [off, off+length) of several sources laid end to end
// (ConcatenateBitmaps); a source without a bitmap contributes set bits (Bitmap::AllSet, :67).
struct BitPiece { const uint8_t* bits; int64_t off; int64_t length; int64_t out_start; };

// One thread builds one output word: whole-word stores only, so the destination needs no
// zero fill and no atomics however the pieces fall.  Bits past `total` in the last word are 0.
__global__ __launch_bounds__(kThreads) void k_concat_bits(const BitPiece* pieces, int npieces, int64_t total, uint64_t* out) {
  const int64_t nwords = (total + 63) >> 6;
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) {
    int64_t pos = w << 6;
    const int64_t end = (pos + 64 < total) ? pos + 64 : total;
    int lo = 0, hi = npieces - 1;
    while (lo < hi) {  // the last piece that starts at or before pos
      const int mid = (lo + hi + 1) >> 1;
      if (pieces[mid].out_start <= pos) lo = mid; else hi = mid - 1;
    }
    uint64_t word = 0;
    for (int c = lo; c < npieces && pos < end; ++c) {
      const BitPiece p = pieces[c];
      const int64_t in_piece = pos - p.out_start;
      int64_t take = p.length - in_piece;
      if (take <= 0) continue;  // an empty piece
      if (take > end - pos) take = end - pos;
      const uint64_t bits = LoadBitsOrOnes(p.bits, p.off + in_piece, static_cast<int>(take));
      word |= bits << (pos & 63);
      pos += take;
    }
    out[w] = word;
  }
}

// Offsets of concatenated variable-width arrays (ConcatenateOffsets, array/concatenate.cc:100-160): piece c's
// offsets are rebased so that its first value starts where piece c - 1's bytes end.
struct OffsetPiece { const int32_t* offsets; int64_t off; int64_t length; int64_t out_start; int64_t rebase; /* out = in + rebase */ };
__global__ __launch_bounds__(kThreads) void k_concat_offsets(const OffsetPiece* pieces, int npieces, int64_t total, int64_t total_bytes,
                                                            int32_t* out) {
  for (int64_t i = int64_t(blockIdx.x) * kThreads + threadIdx.x; i <= total; i += int64_t(gridDim.x) * kThreads) {
    if (i == total) { out[i] = static_cast<int32_t>(total_bytes); continue; }
    int lo = 0, hi = npieces - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (pieces[mid].out_start <= i) lo = mid; else hi = mid - 1;
    }
    const OffsetPiece p = pieces[lo];
    out[i] = static_cast<int32_t>(p.offsets[p.off + (i - p.out_start)] + p.rebase);
  }
}

// ---- Take ---------------------------------------------------------------------------------
struct TakeArgs {
  const void* v_data; const uint8_t* v_valid; int64_t v_off; int64_t v_len; int v_may_null;
  const void* i_data; const uint8_t* i_valid; int64_t i_off; int64_t n;
  int boundscheck;
  void* out_values; uint8_t* out_validity;
  unsigned long long* valid_count;  // device, slotted counters (common.h), counter 0
  unsigned long long* err_pos;      // device, initialised to ~0
  const uint8_t* v_summary;         // nullable: bit B set = byte (v_off >> 3) + B of v_valid is 0xFF (k_validity_summary)
};

// One bit per validity BYTE: is it 0xFF?  A uniformly random take is bound by the number of lines it asks the fabric
// for, not by bytes (100 M rows: 50 M value gathers 1.0 ms; the same plus 50 M validity-byte gathers 1.72 ms -- the
// 12.5 MB bitmap does not fit an XCD's L2).  The summary of the same bitmap is 1.5 MB and does: a row whose
// validity byte is full (92 % of them at 1 % nulls) is served from L2 and only the rest ask for the byte itself.
__global__ __launch_bounds__(kThreads) void k_validity_summary(const uint8_t* valid, int64_t nbytes, uint8_t* summary) {
  const int64_t s = int64_t(blockIdx.x) * kThreads + threadIdx.x;  // summary byte s <- validity bytes [8 s, 8 s + 8)
  if (8 * s >= nbytes) return;
  uint32_t m = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b)
    if (8 * s + b < nbytes && valid[8 * s + b] == 0xFFu) m |= 1u << b;
  summary[s] = static_cast<uint8_t>(m);
}

template <typename IdxT> struct IndexLoad {
  __device__ static bool Load(const void* p, int64_t k, int64_t upper, int64_t* out) {
    const IdxT v = reinterpret_cast<const IdxT*>(p)[k];
    *out = static_cast<int64_t>(v);
    // IsOutOfBounds, util/int_util.cc:548-551
    if (static_cast<IdxT>(-1) < 0) return !(v < 0 || static_cast<uint64_t>(v) >= static_cast<uint64_t>(upper));
    return static_cast<uint64_t>(v) < static_cast<uint64_t>(upper);
  }
};

// One output element per lane; a wave produces one validity word with a ballot.
template <int W, typename IdxT>
__global__ __launch_bounds__(kThreads) void k_take(TakeArgs a) {
  using T = typename UIntOf<(W == 0 ? 1 : W)>::T;
  constexpr bool kBits = (W == 0);
  __shared__ unsigned long long s_valid;
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid == 0) s_valid = 0;
  __syncthreads();
  unsigned long long my_valid = 0;
  // A wave takes 4 x 64 consecutive output rows per step.  The three dependent accesses of a row
  // (index, the value's validity bit, the value) are issued for all four rows before any is used,
  // with clamped addresses instead of branches: a gather is latency-bound and with one row per lane
  // and step a random take ran at 28 G rows/s.
  constexpr int kU = 4;
  const int64_t wave_id = (int64_t(blockIdx.x) * kThreads + tid) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kThreads) >> 6;
  const bool have_values = a.v_len > 0;
  for (int64_t base = wave_id * (64 * kU); base < a.n; base += nwaves * (64 * kU)) {
    // Every stage's loads are stored RAW for all four rows before one of them is looked at: a bit test or a bounds
    // check right behind its load makes the compiler wait for that load (and, the counter being in order, for every
    // load before it) ahead of the next row's -- the "four gathers in flight" were four dependent round trips.
    bool ok[kU];
    int64_t j[kU];
    IdxT iraw[kU];
    uint32_t ivb[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {  // indices (coalesced) and their validity bytes
      const int64_t i = base + 64 * u + lane;
      const int64_t ic = a.i_off + (i < a.n ? i : a.n - 1);
      iraw[u] = reinterpret_cast<const IdxT*>(a.i_data)[ic];
      ivb[u] = a.i_valid ? a.i_valid[ic >> 3] : 0xFFu;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      const int64_t ic = a.i_off + (i < a.n ? i : a.n - 1);
      ok[u] = i < a.n && ((ivb[u] >> (ic & 7)) & 1u);
      j[u] = static_cast<int64_t>(iraw[u]);
      // IsOutOfBounds, util/int_util.cc:548-551
      const bool inb = (static_cast<IdxT>(-1) < 0 && iraw[u] < 0) ? false : static_cast<uint64_t>(iraw[u]) < static_cast<uint64_t>(a.v_len);
      if (ok[u] && !inb) {
        ok[u] = false;
        if (a.boundscheck) atomicMin(a.err_pos, static_cast<unsigned long long>(i));
      }
      if (!ok[u]) j[u] = 0;  // a valid position for the unconditional gathers below
    }
    bool vbit[kU];
    T v[kU];
    uint32_t vvb[kU], bvb[kU];
    const bool need_vbits = a.v_may_null && have_values;
    // (the summary is for scattered indices: where every lane's four rows lie close together -- monotonic indices, a Filter's
    // -- the validity bytes themselves are neighbours in the cache and the summary only adds a dependent stage, +10 %)
    bool summarised = need_vbits && a.v_summary != nullptr;
    if (summarised) {
      int64_t far = 0;
#pragma unroll
      for (int u = 1; u < kU; ++u) { const int64_t d = j[u] - j[0]; far |= (d < 0 ? -d : d) >> 16; }
      summarised = __ballot(far != 0) != 0;  // (wave-uniform)
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {  // gathers, all in flight together
      // values' validity is consulted only when values.null_count != 0 (:306, :338)
      if (summarised) vvb[u] = a.v_summary[(((a.v_off + j[u]) >> 3) - (a.v_off >> 3)) >> 3];
      else vvb[u] = need_vbits ? a.v_valid[(a.v_off + j[u]) >> 3] : 0xFFu;
      v[u] = 0; bvb[u] = 0;
      if (have_values) {
        if (kBits) bvb[u] = reinterpret_cast<const uint8_t*>(a.v_data)[(a.v_off + j[u]) >> 3];
        else v[u] = reinterpret_cast<const T*>(a.v_data)[a.v_off + j[u]];
      }
    }
    if (summarised) {  // (wave-uniform) the validity byte itself only where the summary does not call it full
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int64_t vbyte = (a.v_off + j[u]) >> 3;
        const bool full = (vvb[u] >> ((vbyte - (a.v_off >> 3)) & 7)) & 1u;
        vvb[u] = 0xFFu;
        if (!full) vvb[u] = a.v_valid[vbyte];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      vbit[u] = (vvb[u] >> ((a.v_off + j[u]) & 7)) & 1u;
      if (kBits) v[u] = static_cast<T>((bvb[u] >> ((a.v_off + j[u]) & 7)) & 1u);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      if (base + 64 * u >= a.n) break;  // wave-uniform
      const bool good = ok[u] && vbit[u];
      const T val = good ? v[u] : T(0);  // null slots hold ValueCType{} (:318, :343)
      const uint64_t okmask = __ballot(good);
      if (kBits) {
        const uint64_t vmask = __ballot(val != 0);
        if (lane == 0) reinterpret_cast<uint64_t*>(a.out_values)[i >> 6] = vmask;
      } else if (i < a.n) {
        reinterpret_cast<T*>(a.out_values)[i] = val;
      }
      if (lane == 0) {
        reinterpret_cast<uint64_t*>(a.out_validity)[i >> 6] = okmask;
        my_valid += __popcll(okmask);
      }
    }
  }
  if (lane == 0 && my_valid) atomicAdd(&s_valid, my_valid);
  __syncthreads();
  if (tid == 0 && s_valid) atomicAdd(CounterSlot(a.valid_count), s_valid);
}

// ---- CountSetBits / GetFilterOutputSize -------------------------------------------------------
// mode 0: popcount(a); 1: popcount(a & b); 2: popcount(a | ~b)
__global__ __launch_bounds__(kThreads) void k_popcount(const uint8_t* a, int64_t a_off, const uint8_t* b, int64_t b_off,
                                                       int mode, int64_t length, unsigned long long* total) {
  __shared__ unsigned long long s_sum;
  if (threadIdx.x == 0) s_sum = 0;
  __syncthreads();
  const int64_t nwords = (length + 63) >> 6;
  unsigned long long acc = 0;
  for (int64_t w = int64_t(blockIdx.x) * kThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kThreads) {
    const int64_t row = w << 6;
    const int64_t left = length - row;
    const int nbits = left >= 64 ? 64 : static_cast<int>(left);
    const uint64_t live = nbits < 64 ? (uint64_t(1) << nbits) - 1 : ~uint64_t(0);
    uint64_t x = LoadBits64(a, a_off + row, nbits);
    if (mode == 1) x &= LoadBits64(b, b_off + row, nbits);
    else if (mode == 2) x = (x | ~LoadBits64(b, b_off + row, nbits)) & live;
    acc += __popcll(x);
  }
  acc = WaveReduceSum64(acc);
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&s_sum, acc);
  __syncthreads();
  if (threadIdx.x == 0 && s_sum) atomicAdd(CounterSlot(total), s_sum);
}

static int PopcountImpl(amdarrow_ctx* ctx, const uint8_t* a, int64_t a_off, const uint8_t* b, int64_t b_off, int mode,
                        int64_t length, int64_t* out) {
  if (length <= 0) { *out = 0; return AMDARROW_OK; }
  void* base;
  AA_TRY(Scratch(ctx, kCounterBytes, &base));
  auto* total = static_cast<unsigned long long*>(base);
  AA_TRY(ClearCounters(ctx, total));
  const int grid = GridFor((length + 63) / 64, kThreads, ctx->num_cus * 8);
  hipLaunchKernelGGL(k_popcount, dim3(grid), dim3(kThreads), 0, ctx->stream, a, a_off, b, b_off, mode, length, total);
  AA_HIP(hipGetLastError());
  unsigned long long host = 0;
  AA_TRY(ReadCounters(ctx, total, 1, &host));
  *out = static_cast<int64_t>(host);
  return AMDARROW_OK;
}

static int CheckFilter(const amdarrow_array* filter) {
  if (!filter || filter->type_id != AMDARROW_BOOL)
    return SetError(AMDARROW_NOT_IMPLEMENTED, "Filter argument must be boolean type");  // :1885-1887
  if (filter->length < 0 || filter->offset < 0) return SetError(AMDARROW_INVALID, "negative length/offset");
  return AMDARROW_OK;
}

template <int W, bool IOTA>
static int LaunchCompact(amdarrow_ctx* ctx, const FilterArgs& a) {
  const int grid = GridFor(a.ntiles, 1, ctx->num_cus * 8);
  hipLaunchKernelGGL((k_filter_compact<W, IOTA>), dim3(grid), dim3(kCThreads), 0, ctx->stream, a);
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

// Shared driver of Filter and GetTakeIndices.
static int RunCompaction(amdarrow_ctx* ctx, FilterArgs a, int width_bytes /*0 = bits*/, bool iota, int64_t* total) {
  a.ntiles = (a.length + kTile - 1) / kTile;
  Carver c;
  const size_t off_counts = c.Take(sizeof(uint32_t) * a.ntiles);
  const size_t off_offsets = c.Take(sizeof(int64_t) * (a.ntiles + 1));
  const bool merge_edges = a.has_validity && !getenv("AMDARROW_FILTER_EDGE_ATOMICS");
  const size_t off_edges = c.Take(merge_edges ? 8 * kEdgeStride * static_cast<size_t>(a.ntiles) : 0);
  void* base;
  AA_TRY(Scratch(ctx, c.off, &base));
  auto* counts = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + off_counts);
  auto* offsets = reinterpret_cast<int64_t*>(static_cast<char*>(base) + off_offsets);
  a.tile_offsets = offsets;
  a.edge_words = merge_edges ? reinterpret_cast<unsigned long long*>(static_cast<char*>(base) + off_edges) : nullptr;

  hipLaunchKernelGGL(k_filter_tile_counts, dim3(GridFor(a.ntiles, 4, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     a.f_data, a.f_valid, a.f_off, a.emit_null, a.length, a.ntiles, counts);
  AA_HIP(hipGetLastError());
  // the total goes to mapped host memory; an event behind the scan lets the host pick it up
  // while the compaction (queued right behind) is still running
  hipLaunchKernelGGL(k_scan_tiles, dim3(ScanTilesGrid(a.ntiles)), dim3(1024), 0, ctx->stream, counts, a.ntiles, offsets,
                     reinterpret_cast<long long*>(static_cast<char*>(ctx->pinned_dev) + kStatusOffset));
  AA_HIP(hipGetLastError());
  AA_HIP(hipEventRecord(ctx->status_event, ctx->stream));
  // bitmaps are merged with atomicOr at tile edges: start from zero (also zeroes the tails)
  const size_t bm_bytes = static_cast<size_t>((BytesForBits(a.out_capacity) + 7) & ~int64_t(7));
  if (a.has_validity) AA_HIP(hipMemsetAsync(a.out_validity, 0, bm_bytes, ctx->stream));
  if (width_bytes == 0) AA_HIP(hipMemsetAsync(a.out_values, 0, bm_bytes, ctx->stream));
  int rc;
  if (iota) {
    rc = width_bytes == 2 ? LaunchCompact<2, true>(ctx, a) : LaunchCompact<4, true>(ctx, a);
  } else {
    switch (width_bytes) {
      case 0: rc = LaunchCompact<0, false>(ctx, a); break;
      case 1: rc = LaunchCompact<1, false>(ctx, a); break;
      case 2: rc = LaunchCompact<2, false>(ctx, a); break;
      case 4: rc = LaunchCompact<4, false>(ctx, a); break;
      default: rc = LaunchCompact<8, false>(ctx, a); break;
    }
  }
  AA_TRY(rc);
  if (merge_edges) {
    hipLaunchKernelGGL(k_filter_merge_edges, dim3(static_cast<unsigned>((a.ntiles + 255) / 256)), dim3(256), 0, ctx->stream, offsets,
                       a.ntiles, a.edge_words, a.out_validity, a.out_capacity);
    AA_HIP(hipGetLastError());
  }
  AA_HIP(hipEventSynchronize(ctx->status_event));
  memcpy(total, static_cast<const char*>(ctx->pinned) + kStatusOffset, 8);
  if (*total > a.out_capacity)
    return SetError(AMDARROW_CAPACITY_ERROR, "filter: selection of " + std::to_string(*total) +
                                                 " rows exceeds out_capacity " + std::to_string(a.out_capacity));
  return AMDARROW_OK;
}

}  // namespace amdarrow

namespace amdarrow {

// ---- Take of variable-width binary values (32-bit offsets) -------------------------------------------
// VarBinaryImpl over Selection::VisitTake (vector_selection.cc:1203-1320, 1493-1573): out[i] =
// values[indices[i]], null -- and zero bytes long -- if the index or the value is null; the output
// always carries a validity bitmap (FinishCommon :1250-1255).  Two passes: the lengths (with the
// bounds check and the validity words) scanned into the output offsets, then the bytes.
struct BinaryTakeArgs {
  const int32_t* v_offsets; const uint8_t* v_data; const uint8_t* v_valid; int64_t v_off; int64_t v_len; int v_may_null;
  const void* i_data; const uint8_t* i_valid; int64_t i_off; int64_t n;
  int boundscheck;
  uint32_t* out_lengths;     // n + 1 words: lengths here, offsets after the scan
  uint8_t* out_validity;
  unsigned long long* valid_count;  // slotted counters, counter 0
  unsigned long long* err_pos;
};

template <typename IdxT>
__global__ __launch_bounds__(kThreads) void k_binary_take_lengths(BinaryTakeArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long my_valid = 0;
  const int64_t wave_id = (int64_t(blockIdx.x) * kThreads + tid) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kThreads) >> 6;
  // Four rows per lane and step, every stage's loads stored raw before one is looked at (see k_take).
  constexpr int kU = 4;
  for (int64_t base = wave_id * (64 * kU); base < a.n; base += nwaves * (64 * kU)) {
    IdxT iraw[kU];
    uint32_t ivb[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      const int64_t ic = a.i_off + (i < a.n ? i : a.n - 1);
      iraw[u] = reinterpret_cast<const IdxT*>(a.i_data)[ic];
      ivb[u] = a.i_valid ? a.i_valid[ic >> 3] : 0xFFu;
    }
    bool ok[kU];
    int64_t j[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      const int64_t ic = a.i_off + (i < a.n ? i : a.n - 1);
      ok[u] = i < a.n && ((ivb[u] >> (ic & 7)) & 1u);
      j[u] = static_cast<int64_t>(iraw[u]);
      // IsOutOfBounds, util/int_util.cc:548-551
      const bool inb = (static_cast<IdxT>(-1) < 0 && iraw[u] < 0) ? false : static_cast<uint64_t>(iraw[u]) < static_cast<uint64_t>(a.v_len);
      if (ok[u] && !inb) {
        ok[u] = false;
        if (a.boundscheck) atomicMin(a.err_pos, static_cast<unsigned long long>(i));
      }
      if (!ok[u]) j[u] = 0;
    }
    int32_t o0[kU], o1[kU];
    uint32_t vvb[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      o0[u] = 0; o1[u] = 0; vvb[u] = 0xFFu;
      if (a.v_len > 0) {  // (uniform)
        o0[u] = a.v_offsets[a.v_off + j[u]];
        o1[u] = a.v_offsets[a.v_off + j[u] + 1];
        if (a.v_may_null) vvb[u] = a.v_valid[(a.v_off + j[u]) >> 3];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      if (base + 64 * u >= a.n) break;  // wave-uniform
      bool good = false;
      uint32_t len = 0;
      if (a.v_len > 0) {
        good = ok[u] && ((vvb[u] >> ((a.v_off + j[u]) & 7)) & 1u);
        len = good ? static_cast<uint32_t>(o1[u] - o0[u]) : 0u;
      }
      const uint64_t okmask = __ballot(good);
      if (i < a.n) a.out_lengths[i] = len;
      if (lane == 0) {
        reinterpret_cast<uint64_t*>(a.out_validity)[i >> 6] = okmask;
        my_valid += __popcll(okmask);
      }
    }
  }
  if (lane == 0 && my_valid) atomicAdd(CounterSlot(a.valid_count), my_valid);
}

// out_offsets[n] = total, or the overflow flag when it does not fit an int32 offset
// ("Take operation overflowed binary array capacity", :1540-1544; kOffsetLimit = INT32_MAX - 1).
__global__ void k_binary_close_offsets(uint32_t* offsets, int64_t n, const int64_t* total, unsigned long long* err_pos) {
  const int64_t t = *total;
  offsets[n] = static_cast<uint32_t>(t);
  if (t > 2147483646) err_pos[1] = 1;
}

// A wave copies the bytes of 64 consecutive output rows: short rows lane by lane, long rows one at a
// time with all its lanes.
// (the staged path of k_binary_take_data, per wave and step of 256 rows)
#ifndef BT_KU
#define BT_KU 4
#define BT_SRC 6144
#define BT_DST 3072
#endif
constexpr uint32_t kStageSrc = BT_SRC, kStageDst = BT_DST, kStageRow = 256;
using BinU32x4v = __attribute__((ext_vector_type(4))) unsigned int;
template <typename IdxT>
__global__ __launch_bounds__(kThreads) void k_binary_take_data(BinaryTakeArgs a, const uint32_t* out_offsets, uint8_t* out_data, int staged) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[(kThreads / 64) * (kStageSrc + kStageDst) + 16];  // (+16: the funnel read's second dword)
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t wave_id = (int64_t(blockIdx.x) * kThreads + tid) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kThreads) >> 6;
  // A lane takes kU rows per step, every stage's loads (output offsets and index; source offset; bytes) issued for
  // all of them before one is used: a row is a chain of three dependent round trips plus its bytes, and with one
  // row per lane and step the kernel ran at the speed of that chain (1.0 ms for 50 M eight-byte strings).
  constexpr int kU = BT_KU;
  constexpr uint32_t kShort = 48;
  // (the next step's output offsets and indices are requested while this step's bytes move: a step is a chain of
  // dependent round trips -- offsets and index, source offset, bytes -- and few waves fit beside the staging buffers)
  uint32_t n0[kU], n1[kU];
  IdxT nraw[kU];
  auto load_meta = [&](int64_t base) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      const int64_t ic = i < a.n ? i : a.n - 1;
      n0[u] = out_offsets[ic];
      n1[u] = out_offsets[ic + 1];
      nraw[u] = reinterpret_cast<const IdxT*>(a.i_data)[a.i_off + ic];
    }
  };
  load_meta(wave_id * (64 * kU) < a.n ? wave_id * (64 * kU) : 0);
  for (int64_t base = wave_id * (64 * kU); base < a.n; base += nwaves * (64 * kU)) {
    uint32_t o0[kU], o1[kU], dst[kU], len[kU], mine[kU];
    IdxT iraw[kU];
    int64_t src[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) { o0[u] = n0[u]; o1[u] = n1[u]; iraw[u] = nraw[u]; }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      dst[u] = o0[u];
      len[u] = i < a.n ? o1[u] - o0[u] : 0u;
      // a row with bytes has a valid, in-range index and a valid value; any other row reads source offset 0
      const int64_t j = len[u] ? static_cast<int64_t>(iraw[u]) : 0;
      src[u] = a.v_offsets[a.v_off + j];
    }
    {
      const int64_t nb = base + nwaves * (64 * kU);
      load_meta(nb < a.n ? nb : base);
    }
    // ---- staged path: the step's rows come from a narrow range of the source (a Filter's indices are monotonic; so are
    // most takes of sorted positions).  Lane-per-row copies read and write global memory in pieces of 1-8 bytes per lane
    // -- 64 scattered requests per instruction, 0.9 ms for 50 M eight-byte strings.  Here the wave loads the whole source
    // range into LDS in 16-byte pieces, the rows are copied LDS -> LDS to where they stand in the output range, and that
    // range leaves in 16-byte pieces: global memory only sees whole, aligned vectors.
    // All kU runs of 64 rows together if their ranges fit the buffers, else each half, else each run by itself (a sparser
    // selection spreads the same number of rows over a longer source range); what is left takes the paths below.
    auto stage = [&](auto all_tag, int lo, int size) -> int {
      constexpr bool kAll = decltype(all_tag)::value;  // (the usual, first attempt: no run tests in the unrolled loops)
      auto in = [&](int u) { return kAll || (u >= lo && u < lo + size); };
      uint32_t smin = ~0u, smax = 0, dmin = ~0u, dmax = 0, lmax = 0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (in(u) && len[u]) {
          const uint32_t sb = static_cast<uint32_t>(src[u]);
          smin = sb < smin ? sb : smin; smax = sb + len[u] > smax ? sb + len[u] : smax;
          dmin = dst[u] < dmin ? dst[u] : dmin; dmax = dst[u] + len[u] > dmax ? dst[u] + len[u] : dmax;
          lmax = len[u] > lmax ? len[u] : lmax;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        uint32_t t;
        t = __shfl_xor(smin, o); smin = t < smin ? t : smin;
        t = __shfl_xor(smax, o); smax = t > smax ? t : smax;
        t = __shfl_xor(dmin, o); dmin = t < dmin ? t : dmin;
        t = __shfl_xor(dmax, o); dmax = t > dmax ? t : dmax;
        t = __shfl_xor(lmax, o); lmax = t > lmax ? t : lmax;
      }
      if (smax <= smin) return 0;  // no row of these runs has bytes left (wave-uniform)
      const uint32_t sbase = smin & ~15u, dbase = dmin & ~15u;
      if (smax - sbase <= kStageSrc && dmax - dbase <= kStageDst && lmax <= kStageRow) {
        uint8_t* s_src = s_stage + (tid >> 6) * (kStageSrc + kStageDst);
        uint8_t* s_dst = s_src + kStageSrc;
        const uint32_t spieces = (smax - sbase + 15) >> 4;
        for (uint32_t c = lane; c < spieces; c += 64) {
          const uint8_t* gp = a.v_data + sbase + 16 * c;
          BinU32x4v q;
          if (sbase + 16 * c + 16 <= smax) q = *reinterpret_cast<const BinU32x4v*>(gp);
          else {  // the range's last piece: not a byte past its end
            uint8_t tmp[16];
            for (int b = 0; b < 16; ++b) tmp[b] = sbase + 16 * c + b < smax ? gp[b] : uint8_t(0);
            __builtin_memcpy(&q, tmp, 16);
          }
          *reinterpret_cast<BinU32x4v*>(s_src + 16 * c) = q;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // four source bytes at any alignment as ONE LDS read (two neighbouring dwords, funnel-shifted), then byte stores:
        // five LDS instructions per four bytes where byte-by-byte took eight (the copy is bound by their issue)
        for (uint32_t b0 = 0; b0 < lmax; b0 += 4) {
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            const uint32_t rem = in(u) && len[u] > b0 ? len[u] - b0 : 0u;
            const uint32_t sp = rem ? (static_cast<uint32_t>(src[u]) - sbase) + b0 : 0u;
            const uint32_t* s32 = reinterpret_cast<const uint32_t*>(s_src) + (sp >> 2);
            const uint32_t w = __builtin_amdgcn_alignbyte(s32[1], s32[0], sp & 3u);
            uint8_t* dp = s_dst + (dst[u] - dbase) + b0;
            if (rem > 0) dp[0] = static_cast<uint8_t>(w);
            if (rem > 1) dp[1] = static_cast<uint8_t>(w >> 8);
            if (rem > 2) dp[2] = static_cast<uint8_t>(w >> 16);
            if (rem > 3) dp[3] = static_cast<uint8_t>(w >> 24);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t dpieces = (dmax - dbase + 15) >> 4;
        for (uint32_t c = lane; c < dpieces; c += 64) {
          const uint32_t g0 = dbase + 16 * c;
          uint8_t* gp = out_data + g0;
          if (g0 >= dmin && g0 + 16 <= dmax) *reinterpret_cast<BinU32x4v*>(gp) = *reinterpret_cast<const BinU32x4v*>(s_dst + 16 * c);
          else {  // the range's first and last piece: the bytes outside it belong to the neighbouring steps
            for (int b = 0; b < 16; ++b)
              if (g0 + b >= dmin && g0 + b < dmax) gp[b] = s_dst[16 * c + b];
          }
        }
        __builtin_amdgcn_wave_barrier();  // (the next loads into s_src wait for these reads)
#pragma unroll
        for (int u = 0; u < kU; ++u) if (in(u)) len[u] = 0;  // moved
        return 0;
      }
      // too long a range for the buffers.  Fewer rows at a time pay while most of the source range is wanted: the staged
      // copy reads all of it, the lane-per-row copy below only the rows' own bytes (selecting 30 % of eight-byte strings the
      // two take the same time, below that the rows' own bytes win)
      return lmax <= kStageRow && smax - smin <= 3 * (dmax - dmin) ? 1 : 2;
    };
    if (staged && stage(std::true_type{}, 0, kU) == 1) {
      for (int size = kU / 2; size >= 1; size >>= 1)
        for (int lo = 0; lo < kU; lo += size) stage(std::false_type{}, lo, size);
    }
    // short rows (the usual string): every lane moves its own rows, eight bytes per row in flight per step
    uint32_t longest = 0;
#pragma unroll
    for (int u = 0; u < kU; ++u) { mine[u] = len[u] <= kShort ? len[u] : 0; longest = mine[u] > longest ? mine[u] : longest; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t other = __shfl_xor(longest, o); longest = other > longest ? other : longest; }
    // eight bytes of a row per step as ONE unaligned 8-byte load and store (4 + 2 + 1 for the row's tail): byte by byte
    // the kernel ran at the rate the address unit takes scattered lanes, one per clock -- 64 bytes per instruction
    for (uint32_t b0 = 0; b0 < longest; b0 += 8) {
      unsigned long long q8[kU]; unsigned int w4[kU]; unsigned short h2[kU]; uint8_t b1[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t rem = mine[u] > b0 ? mine[u] - b0 : 0u;
        const uint8_t* sp = a.v_data + src[u] + b0;
        q8[u] = 0; w4[u] = 0; h2[u] = 0; b1[u] = 0;
        if (rem >= 8) __builtin_memcpy(&q8[u], sp, 8);
        else {
          if (rem & 4u) __builtin_memcpy(&w4[u], sp, 4);
          if (rem & 2u) __builtin_memcpy(&h2[u], sp + (rem & 4u), 2);
          if (rem & 1u) b1[u] = sp[rem & 6u];
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const uint32_t rem = mine[u] > b0 ? mine[u] - b0 : 0u;
        uint8_t* dp = out_data + dst[u] + b0;
        if (rem >= 8) __builtin_memcpy(dp, &q8[u], 8);
        else {
          if (rem & 4u) __builtin_memcpy(dp, &w4[u], 4);
          if (rem & 2u) __builtin_memcpy(dp + (rem & 4u), &h2[u], 2);
          if (rem & 1u) dp[rem & 6u] = b1[u];
        }
      }
    }
    // long rows: one at a time with all lanes of the wave
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      uint64_t todo = __ballot(len[u] > kShort);
      while (todo) {
        const int r = __ffsll(static_cast<long long>(todo)) - 1;
        todo &= todo - 1;
        const uint32_t rlen = __shfl(len[u], r), rdst = __shfl(dst[u], r);
        const int64_t rsrc = __shfl(src[u], r);
        for (uint32_t b = lane; b < rlen; b += 64) out_data[rdst + b] = a.v_data[rsrc + b];
      }
    }
  }
}


}  // namespace amdarrow

using namespace amdarrow;

extern "C" {

int amdarrow_count_set_bits(amdarrow_ctx* ctx, const uint8_t* bitmap, int64_t offset, int64_t length,
                            int64_t* out_count) {
  std::lock_guard<std::mutex> lock(ctx->mu);
  return PopcountImpl(ctx, bitmap, offset, nullptr, 0, 0, length, out_count);
}

int amdarrow_filter_output_size(amdarrow_ctx* ctx, const amdarrow_array* filter, int null_selection,
                                int64_t* out_length) {
  AA_TRY(CheckFilter(filter));
  std::lock_guard<std::mutex> lock(ctx->mu);
  const uint8_t* data = static_cast<const uint8_t*>(filter->values);
  if (MayHaveNulls(filter)) {  // :64-82
    return PopcountImpl(ctx, data, filter->offset, filter->validity, filter->offset,
                        null_selection == AMDARROW_EMIT_NULL ? 2 : 1, filter->length, out_length);
  }
  return PopcountImpl(ctx, data, filter->offset, nullptr, 0, 0, filter->length, out_length);  // :85
}

int amdarrow_filter(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* filter,
                    int null_selection, void* out_values, uint8_t* out_validity, int64_t out_capacity,
                    amdarrow_out* out) {
  AA_TRY(CheckFilter(filter));
  const int bw = values ? BitWidth(values->type_id) : -1;
  if (bw < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "filter: values must be a primitive fixed-width type");
  if (values->length != filter->length)
    return SetError(AMDARROW_INVALID, "Array arguments must all be the same length");  // compute/exec.cc:190-192
  const int64_t v_nc = ArgNullCount(values), f_nc = ArgNullCount(filter);
  out->null_count = (v_nc == 0 && (null_selection == AMDARROW_DROP || f_nc == 0)) ? 0 : -1;  // :799-804
  out->has_validity = (v_nc != 0 || f_nc != 0);                                              // :809
  out->length = 0;
  if (out->has_validity && !out_validity) return SetError(AMDARROW_INVALID, "filter: out_validity required");
  if (values->length == 0) return AMDARROW_OK;

  FilterArgs a{};
  const bool non_null_path = (f_nc == 0 && v_nc == 0);  // ExecNonNull :625-630 looks at data bits only
  a.f_data = static_cast<const uint8_t*>(filter->values);
  a.f_valid = non_null_path ? nullptr : filter->validity;
  a.f_off = filter->offset;
  a.emit_null = (null_selection == AMDARROW_EMIT_NULL && a.f_valid != nullptr);
  a.v_data = values->values;
  a.v_valid = values->validity;
  a.v_off = values->offset;
  a.length = values->length;
  a.out_values = out_values;
  a.out_validity = out_validity;
  a.out_capacity = out_capacity;
  a.has_validity = out->has_validity;
  std::lock_guard<std::mutex> lock(ctx->mu);
  int64_t total = 0;
  AA_TRY(RunCompaction(ctx, a, bw == 1 ? 0 : bw / 8, /*iota=*/false, &total));
  out->length = total;
  return AMDARROW_OK;
}

int amdarrow_take_indices(amdarrow_ctx* ctx, const amdarrow_array* filter, int null_selection, void* out_indices,
                          uint8_t* out_validity, int64_t out_capacity, int* out_index_width, amdarrow_out* out) {
  AA_TRY(CheckFilter(filter));
  if (filter->length > 0xFFFFFFFFLL)  // :230-235
    return SetError(AMDARROW_NOT_IMPLEMENTED,
                    "Filter length exceeds UINT32_MAX, consider a different strategy for selecting elements");
  const int w = filter->length <= 65535 ? 2 : 4;  // :224-229
  *out_index_width = w * 8;
  const bool have_nulls = MayHaveNulls(filter);
  const bool emit_nulls = have_nulls && null_selection == AMDARROW_EMIT_NULL;  // :103
  out->has_validity = emit_nulls;
  out->null_count = 0;
  out->length = 0;
  if (emit_nulls && !out_validity) return SetError(AMDARROW_INVALID, "take_indices: out_validity required");
  if (filter->length == 0) return AMDARROW_OK;
  FilterArgs a{};
  a.f_data = static_cast<const uint8_t*>(filter->values);
  a.f_valid = have_nulls ? filter->validity : nullptr;
  a.f_off = filter->offset;
  a.emit_null = emit_nulls;
  a.length = filter->length;
  a.out_values = out_indices;
  a.out_validity = out_validity;
  a.out_capacity = out_capacity;
  a.has_validity = emit_nulls;  // validity of an emitted slot = the filter slot was valid-and-true
  std::lock_guard<std::mutex> lock(ctx->mu);
  int64_t total = 0;
  AA_TRY(RunCompaction(ctx, a, w, /*iota=*/true, &total));
  out->length = total;
  if (emit_nulls) {
    // null_count of the builder's result = number of null filter slots
    int64_t valid = 0;
    AA_TRY(PopcountImpl(ctx, out_validity, 0, nullptr, 0, 0, total, &valid));
    out->null_count = total - valid;
  }
  return AMDARROW_OK;
}

int amdarrow_take(amdarrow_ctx* ctx, const amdarrow_array* values, const amdarrow_array* indices, int boundscheck,
                  void* out_values, uint8_t* out_validity, amdarrow_out* out) {
  const int bw = values ? BitWidth(values->type_id) : -1;
  if (bw < 0) return SetError(AMDARROW_NOT_IMPLEMENTED, "take: values must be a primitive fixed-width type");
  const int it = indices->type_id;
  if (!IsSignedInt(it) && !IsUnsignedInt(it))
    return SetError(AMDARROW_NOT_IMPLEMENTED, "take: indices must be an integer type");  // :2067-2070
  out->length = indices->length;
  out->has_validity = 1;  // always allocated (:505-510)
  out->null_count = 0;
  if (indices->length == 0) return AMDARROW_OK;
  if (!out_validity || !out_values) return SetError(AMDARROW_INVALID, "take: output buffers required");

  std::lock_guard<std::mutex> lock(ctx->mu);
  // The summary of the values' validity (k_validity_summary) pays where the bitmap is larger than an XCD's L2, the take is
  // long enough to repay a pass over the bitmap, and most validity bytes are full.
  const int64_t v_nulls = ArgNullCount(values);
  const int64_t v_first_byte = values->offset >> 3;
  const int64_t v_nbytes = ((values->offset + values->length + 7) >> 3) - v_first_byte;
  const bool summarise = v_nulls != 0 && values->length > 0 && v_nbytes > (int64_t(2) << 20) && indices->length >= (int64_t(1) << 20) &&
                         (v_nulls < 0 || v_nulls * 16 <= values->length) && !getenv("AMDARROW_TAKE_NO_SUMMARY");
  const int64_t summary_bytes = summarise ? (v_nbytes + 7) / 8 : 0;
  void* base;
  AA_TRY(Scratch(ctx, kCounterBytes + 256 + summary_bytes, &base));
  auto* counters = static_cast<unsigned long long*>(base);                    // valid rows, slotted
  auto* err_pos = counters + kCounterSlots * kCounterStride;                  // first out-of-bounds position (atomicMin: rare)
  AA_TRY(ClearCounters(ctx, counters));
  AA_HIP(hipMemsetAsync(err_pos, 0xFF, 8, ctx->stream));
  TakeArgs a{};
  a.v_data = values->values; a.v_valid = values->validity; a.v_off = values->offset; a.v_len = values->length;
  a.v_may_null = v_nulls != 0;
  if (summarise) {
    auto* summary = reinterpret_cast<uint8_t*>(base) + kCounterBytes + 256;
    hipLaunchKernelGGL(k_validity_summary, dim3(static_cast<unsigned>((summary_bytes + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       ctx->stream, values->validity + v_first_byte, v_nbytes, summary);
    a.v_summary = summary;
  }
  a.i_data = indices->values; a.i_valid = indices->validity; a.i_off = indices->offset; a.n = indices->length;
  a.boundscheck = boundscheck;
  a.out_values = out_values; a.out_validity = out_validity;
  a.valid_count = counters; a.err_pos = err_pos;
  const int grid = GridFor(a.n, kThreads, ctx->num_cus * 8);
  const int iw = BitWidth(it) / 8;
  const bool sgn = IsSignedInt(it);
#define AA_TAKE(W)                                                                                              \
  do {                                                                                                          \
    if (iw == 1) { if (sgn) hipLaunchKernelGGL((k_take<W, int8_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); \
                   else hipLaunchKernelGGL((k_take<W, uint8_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); } \
    else if (iw == 2) { if (sgn) hipLaunchKernelGGL((k_take<W, int16_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); \
                        else hipLaunchKernelGGL((k_take<W, uint16_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); } \
    else if (iw == 4) { if (sgn) hipLaunchKernelGGL((k_take<W, int32_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); \
                        else hipLaunchKernelGGL((k_take<W, uint32_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); } \
    else { if (sgn) hipLaunchKernelGGL((k_take<W, int64_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a);    \
           else hipLaunchKernelGGL((k_take<W, uint64_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, a); }     \
  } while (0)
  switch (bw) {
    case 1: AA_TAKE(0); break;
    case 8: AA_TAKE(1); break;
    case 16: AA_TAKE(2); break;
    case 32: AA_TAKE(4); break;
    default: AA_TAKE(8); break;
  }
#undef AA_TAKE
  AA_HIP(hipGetLastError());
  unsigned long long host[2];
  AA_HIP(hipMemcpyAsync(static_cast<char*>(ctx->pinned) + kStatusOffset, err_pos, 8, hipMemcpyDeviceToHost, ctx->stream));
  AA_TRY(ReadCounters(ctx, counters, 1, host));  // synchronises the stream: the err_pos copy above has landed too
  memcpy(&host[1], static_cast<const char*>(ctx->pinned) + kStatusOffset, 8);
  if (boundscheck && host[1] != ~0ull) {
    // Report the first offending index, as CheckIndexBounds does (util/int_util.cc:553-561)
    unsigned char raw[8] = {0};
    AA_HIP(hipMemcpyAsync(raw, static_cast<const char*>(indices->values) + (indices->offset + host[1]) * iw, iw,
                          hipMemcpyDeviceToHost, ctx->stream));
    AA_HIP(hipStreamSynchronize(ctx->stream));
    std::string s;
    if (sgn) {
      long long v = iw == 1 ? *reinterpret_cast<int8_t*>(raw) : iw == 2 ? *reinterpret_cast<int16_t*>(raw)
                  : iw == 4 ? *reinterpret_cast<int32_t*>(raw) : *reinterpret_cast<int64_t*>(raw);
      s = std::to_string(v);
    } else {
      unsigned long long v = iw == 1 ? *reinterpret_cast<uint8_t*>(raw) : iw == 2 ? *reinterpret_cast<uint16_t*>(raw)
                           : iw == 4 ? *reinterpret_cast<uint32_t*>(raw) : *reinterpret_cast<uint64_t*>(raw);
      s = std::to_string(v);
    }
    return SetError(AMDARROW_INDEX_ERROR, "Index " + s + " out of bounds");
  }
  out->null_count = indices->length - static_cast<int64_t>(host[0]);  // :367
  return AMDARROW_OK;
}


/* Concatenate, array/concatenate.cc:171-201 (fixed-width and boolean arrays). */
int amdarrow_concatenate(amdarrow_ctx* ctx, const amdarrow_array* chunks, int nchunks, void* out_values,
                         uint8_t* out_validity, amdarrow_out* out) {
  if (nchunks < 1) return SetError(AMDARROW_INVALID, "Must pass at least one array");  // :428-430
  const int type_id = chunks[0].type_id;
  const int bw = BitWidth(type_id);
  if (bw < 0) return SetError(AMDARROW_TYPE_ERROR, "concatenate: unsupported type");
  int64_t total = 0, nulls = 0;
  for (int c = 0; c < nchunks; ++c) {
    if (chunks[c].type_id != type_id) return SetError(AMDARROW_INVALID, "arrays to be concatenated must be identically typed");
    total += chunks[c].length;
    // :172-180: the null count adds up unless some input's is unknown
    const int64_t nc = chunks[c].validity ? chunks[c].null_count : 0;
    nulls = (nulls < 0 || nc < 0) ? -1 : nulls + nc;
  }
  out->length = total;
  out->null_count = nulls;
  out->has_validity = (nulls != 0 && out_validity != nullptr) ? 1 : 0;  // :190-192
  if (total == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const bool bit_values = bw == 1;
  const int ntables = (out->has_validity ? 1 : 0) + (bit_values ? 1 : 0);
  if (ntables > 0) {
    std::vector<BitPiece> host(static_cast<size_t>(ntables) * nchunks);
    int64_t start = 0;
    for (int c = 0; c < nchunks; ++c) {
      int t = 0;
      if (out->has_validity)
        host[t++ * nchunks + c] = BitPiece{MayHaveNulls(&chunks[c]) ? chunks[c].validity : nullptr, chunks[c].offset,
                                           chunks[c].length, start};
      if (bit_values)
        host[t * nchunks + c] = BitPiece{static_cast<const uint8_t*>(chunks[c].values), chunks[c].offset, chunks[c].length, start};
      start += chunks[c].length;
    }
    void* base;
    AA_TRY(Scratch(ctx, sizeof(BitPiece) * host.size(), &base));
    AA_HIP(hipMemcpyAsync(base, host.data(), sizeof(BitPiece) * host.size(), hipMemcpyHostToDevice, ctx->stream));
    AA_HIP(hipStreamSynchronize(ctx->stream));  // `host` is pageable
    const BitPiece* d = static_cast<const BitPiece*>(base);
    const int grid = GridFor((total + 63) / 64, kThreads, ctx->num_cus * 8);
    int t = 0;
    if (out->has_validity)
      hipLaunchKernelGGL(k_concat_bits, dim3(grid), dim3(kThreads), 0, ctx->stream, d + (t++) * nchunks, nchunks, total,
                         reinterpret_cast<uint64_t*>(out_validity));
    if (bit_values)
      hipLaunchKernelGGL(k_concat_bits, dim3(grid), dim3(kThreads), 0, ctx->stream, d + t * nchunks, nchunks, total,
                         reinterpret_cast<uint64_t*>(out_values));
    AA_HIP(hipGetLastError());
  }
  if (!bit_values) {  // ConcatenateBuffers (:199-203): byte ranges laid end to end
    const int w = bw / 8;
    char* dst = static_cast<char*>(out_values);
    for (int c = 0; c < nchunks; ++c) {
      if (chunks[c].length == 0) continue;
      AA_HIP(hipMemcpyAsync(dst, static_cast<const char*>(chunks[c].values) + chunks[c].offset * w, chunks[c].length * w,
                            hipMemcpyDeviceToDevice, ctx->stream));
      dst += chunks[c].length * w;
    }
  }
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}


/* Take of BinaryArray / StringArray values (32-bit offsets): VarBinaryImpl, vector_selection.cc:1493-1573.
 * Pass 1 -- offsets, validity, null count, bounds check; *out_data_bytes = bytes pass 2 will write. */
int amdarrow_binary_take_offsets(amdarrow_ctx* ctx, const amdarrow_binary_array* values, const amdarrow_array* indices, int boundscheck,
                                 int32_t* out_offsets, uint8_t* out_validity, int64_t* out_data_bytes, amdarrow_out* out) {
  const int iw = BitWidth(indices->type_id) / 8;
  const bool sgn = indices->type_id == AMDARROW_INT8 || indices->type_id == AMDARROW_INT16 || indices->type_id == AMDARROW_INT32 ||
                   indices->type_id == AMDARROW_INT64;
  if (iw < 1 || indices->type_id == AMDARROW_FLOAT || indices->type_id == AMDARROW_DOUBLE || indices->type_id == AMDARROW_BOOL)
    return SetError(AMDARROW_TYPE_ERROR, "take: indices must be integers");
  out->length = indices->length;
  out->null_count = 0;
  out->has_validity = 1;
  *out_data_bytes = 0;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const int64_t n = indices->length;
  if (n == 0) {
    AA_HIP(hipMemsetAsync(out_offsets, 0, 4, ctx->stream));
    AA_HIP(hipStreamSynchronize(ctx->stream));
    return AMDARROW_OK;
  }
  const int64_t nb = (n + 2047) / 2048;
  Carver c;
  const size_t o_cnt = c.Take(kCounterBytes);
  const size_t o_err = c.Take(256);
  const size_t o_sums = c.Take(4 * static_cast<size_t>(nb));
  const size_t o_sumoff = c.Take(8 * static_cast<size_t>(nb + 1));
  void* base;
  AA_TRY(Scratch(ctx, c.off, &base));
  char* b = static_cast<char*>(base);
  auto* counters = reinterpret_cast<unsigned long long*>(b + o_cnt);
  auto* err = reinterpret_cast<unsigned long long*>(b + o_err);
  auto* sums = reinterpret_cast<uint32_t*>(b + o_sums);
  auto* sumoff = reinterpret_cast<int64_t*>(b + o_sumoff);
  AA_TRY(ClearCounters(ctx, counters));
  const unsigned long long init[2] = {~0ull, 0ull};
  AA_HIP(hipMemcpyAsync(err, init, 16, hipMemcpyHostToDevice, ctx->stream));
  BinaryTakeArgs a{};
  a.v_offsets = values->offsets; a.v_data = values->data; a.v_valid = values->validity; a.v_off = values->offset;
  a.v_len = values->length; a.v_may_null = values->validity && values->null_count != 0;
  a.i_data = indices->values; a.i_valid = MayHaveNulls(indices) ? indices->validity : nullptr; a.i_off = indices->offset; a.n = n;
  a.boundscheck = boundscheck;
  a.out_lengths = reinterpret_cast<uint32_t*>(out_offsets); a.out_validity = out_validity; a.valid_count = counters; a.err_pos = err;
  const int grid = GridFor(n, kThreads, ctx->num_cus * 8);
#define AA_BIN(K, ...)                                                                                               \
  do {                                                                                                               \
    if (iw == 1) { if (sgn) hipLaunchKernelGGL((K<int8_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__);  \
                   else hipLaunchKernelGGL((K<uint8_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__); }   \
    else if (iw == 2) { if (sgn) hipLaunchKernelGGL((K<int16_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__); \
                        else hipLaunchKernelGGL((K<uint16_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__); } \
    else if (iw == 4) { if (sgn) hipLaunchKernelGGL((K<int32_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__); \
                        else hipLaunchKernelGGL((K<uint32_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__); } \
    else { if (sgn) hipLaunchKernelGGL((K<int64_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__);       \
           else hipLaunchKernelGGL((K<uint64_t>), dim3(grid), dim3(kThreads), 0, ctx->stream, __VA_ARGS__); }        \
  } while (0)
  AA_BIN(k_binary_take_lengths, a);
  AA_HIP(hipGetLastError());
  AA_TRY(ScanU32InPlace(ctx, a.out_lengths, n, sums, sumoff));
  hipLaunchKernelGGL(k_binary_close_offsets, dim3(1), dim3(1), 0, ctx->stream, a.out_lengths, n, sumoff + nb, err);
  AA_HIP(hipGetLastError());
  unsigned long long host_valid = 0, host_err[2] = {0, 0};
  int64_t total = 0;
  AA_TRY(ReadCounters(ctx, counters, 1, &host_valid));
  AA_TRY(ReadBack(ctx, err, 16, host_err));
  AA_TRY(ReadBack(ctx, sumoff + nb, 8, &total));
  if (boundscheck && host_err[0] != ~0ull) {
    unsigned char raw[8] = {0};
    AA_HIP(hipMemcpyAsync(raw, static_cast<const char*>(indices->values) + (indices->offset + host_err[0]) * iw, iw, hipMemcpyDeviceToHost,
                          ctx->stream));
    AA_HIP(hipStreamSynchronize(ctx->stream));
    std::string s;
    if (sgn) {
      long long v = iw == 1 ? *reinterpret_cast<int8_t*>(raw) : iw == 2 ? *reinterpret_cast<int16_t*>(raw)
                  : iw == 4 ? *reinterpret_cast<int32_t*>(raw) : *reinterpret_cast<int64_t*>(raw);
      s = std::to_string(v);
    } else {
      unsigned long long v = iw == 1 ? *reinterpret_cast<uint8_t*>(raw) : iw == 2 ? *reinterpret_cast<uint16_t*>(raw)
                           : iw == 4 ? *reinterpret_cast<uint32_t*>(raw) : *reinterpret_cast<uint64_t*>(raw);
      s = std::to_string(v);
    }
    return SetError(AMDARROW_INDEX_ERROR, "Index " + s + " out of bounds");
  }
  if (host_err[1]) return SetError(AMDARROW_INVALID, "Take operation overflowed binary array capacity");
  *out_data_bytes = total;
  out->null_count = n - static_cast<int64_t>(host_valid);
  return AMDARROW_OK;
}

/* Pass 2 -- the bytes, into out_data (>= the size pass 1 reported). */
int amdarrow_binary_take_data(amdarrow_ctx* ctx, const amdarrow_binary_array* values, const amdarrow_array* indices,
                              const int32_t* out_offsets, uint8_t* out_data) {
  const int iw = BitWidth(indices->type_id) / 8;
  const bool sgn = indices->type_id == AMDARROW_INT8 || indices->type_id == AMDARROW_INT16 || indices->type_id == AMDARROW_INT32 ||
                   indices->type_id == AMDARROW_INT64;
  const int64_t n = indices->length;
  if (n == 0) return AMDARROW_OK;
  // an empty values array may come without an offsets buffer (length 0: every index is null): no row has bytes, and the
  // copy kernel reads source offset 0 for rows without bytes
  if (values->length == 0 || values->offsets == nullptr) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  BinaryTakeArgs a{};
  a.v_offsets = values->offsets; a.v_data = values->data; a.v_off = values->offset; a.v_len = values->length;
  a.i_data = indices->values; a.i_off = indices->offset; a.n = n;
  const int grid = GridFor(n, kThreads, ctx->num_cus * 8);
  // (the staged path moves 16-byte vectors: both byte buffers on 16-byte boundaries, which Arrow's 64-byte aligned buffers are)
  const int staged = (reinterpret_cast<uintptr_t>(values->data) & 15) == 0 && (reinterpret_cast<uintptr_t>(out_data) & 15) == 0 &&
                     !getenv("AMDARROW_BINARY_TAKE_NO_STAGING");
  AA_BIN(k_binary_take_data, a, reinterpret_cast<const uint32_t*>(out_offsets), out_data, staged);
#undef AA_BIN
  AA_HIP(hipGetLastError());
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}


/* Concatenate of BinaryArrays / StringArrays (32-bit offsets): ConcatenateImpl::Visit(BinaryType), array/concatenate.cc:
 * 203-214 -- offsets rebased piece by piece, only the referenced byte range of every piece copied. */
int amdarrow_binary_concatenate_size(amdarrow_ctx* ctx, const amdarrow_binary_array* chunks, int nchunks, int64_t* out_rows,
                                     int64_t* out_data_bytes) {
  if (nchunks < 1) return SetError(AMDARROW_INVALID, "Must pass at least one array");
  std::lock_guard<std::mutex> lock(ctx->mu);
  int64_t rows = 0, bytes = 0;
  for (int c = 0; c < nchunks; ++c) {
    rows += chunks[c].length;
    if (chunks[c].length == 0) continue;
    int32_t first = 0, last = 0;
    AA_TRY(ReadBack(ctx, chunks[c].offsets + chunks[c].offset, 4, &first));
    AA_TRY(ReadBack(ctx, chunks[c].offsets + chunks[c].offset + chunks[c].length, 4, &last));
    bytes += last - first;
  }
  if (bytes > 2147483647LL) return SetError(AMDARROW_INVALID, "offset overflow while concatenating arrays");  // :138-141
  *out_rows = rows;
  *out_data_bytes = bytes;
  return AMDARROW_OK;
}

int amdarrow_binary_concatenate(amdarrow_ctx* ctx, const amdarrow_binary_array* chunks, int nchunks, int32_t* out_offsets,
                                uint8_t* out_data, uint8_t* out_validity, amdarrow_out* out) {
  if (nchunks < 1) return SetError(AMDARROW_INVALID, "Must pass at least one array");
  std::lock_guard<std::mutex> lock(ctx->mu);
  std::vector<OffsetPiece> op(nchunks);
  std::vector<BitPiece> bp(nchunks);
  int64_t rows = 0, bytes = 0, nulls = 0;
  for (int c = 0; c < nchunks; ++c) {
    const amdarrow_binary_array& a = chunks[c];
    int32_t first = 0, last = 0;
    if (a.length > 0) {
      AA_TRY(ReadBack(ctx, a.offsets + a.offset, 4, &first));
      AA_TRY(ReadBack(ctx, a.offsets + a.offset + a.length, 4, &last));
      AA_HIP(hipMemcpyAsync(out_data + bytes, a.data + first, static_cast<size_t>(last - first), hipMemcpyDeviceToDevice, ctx->stream));
    }
    op[c] = OffsetPiece{a.offsets, a.offset, a.length, rows, bytes - first};
    bp[c] = BitPiece{(a.validity && a.null_count != 0) ? a.validity : nullptr, a.offset, a.length, rows};
    const int64_t nc = a.validity ? a.null_count : 0;
    nulls = (nulls < 0 || nc < 0) ? -1 : nulls + nc;
    rows += a.length;
    bytes += last - first;
  }
  out->length = rows;
  out->null_count = nulls;
  out->has_validity = (nulls != 0 && out_validity != nullptr) ? 1 : 0;
  void* base;
  Carver cv;
  const size_t o_op = cv.Take(sizeof(OffsetPiece) * nchunks);
  const size_t o_bp = cv.Take(sizeof(BitPiece) * nchunks);
  AA_TRY(Scratch(ctx, cv.off, &base));
  char* b = static_cast<char*>(base);
  AA_HIP(hipMemcpyAsync(b + o_op, op.data(), sizeof(OffsetPiece) * nchunks, hipMemcpyHostToDevice, ctx->stream));
  AA_HIP(hipMemcpyAsync(b + o_bp, bp.data(), sizeof(BitPiece) * nchunks, hipMemcpyHostToDevice, ctx->stream));
  AA_HIP(hipStreamSynchronize(ctx->stream));  // pageable host tables
  hipLaunchKernelGGL(k_concat_offsets, dim3(GridFor(rows + 1, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                     reinterpret_cast<const OffsetPiece*>(b + o_op), nchunks, rows, bytes, out_offsets);
  if (out->has_validity && rows > 0)
    hipLaunchKernelGGL(k_concat_bits, dim3(GridFor((rows + 63) / 64, kThreads, ctx->num_cus * 8)), dim3(kThreads), 0, ctx->stream,
                       reinterpret_cast<const BitPiece*>(b + o_bp), nchunks, rows, reinterpret_cast<uint64_t*>(out_validity));
  AA_HIP(hipGetLastError());
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}

}  // extern "C"

// dist_merge.hip -- the device half of the multi-GPU group-by merge (the dense protocol of
// arrow-1_amd/dist.py): each rank scatters its partial group table into accumulators indexed by
// key - kmin, RCCL adds them up (and takes the minimum of a packed "owner" word per key), and the
// merged table is read back out in GLOBAL first-appearance order.
//
// The order needs no sort: shards are row ranges in rank order, so the global order is rank 0's
// groups in their local order, then the groups rank 1 saw first in ITS local order, ...; the owner
// word (rank << 32 | local group id) names that position.  Every key drops its dense index into slot
// rank * capacity + local id, and a stream compaction of the slots (tile counts, single-workgroup
// scan, emit) is the ordered table.  Five short kernels, no host round trip; the torch version of
// the same protocol (dist.merge_dense: ~25 launches and an argsort, 0.5 ms for 1 M groups) stays
// as the CPU / gloo implementation the world_size-2 tests drive.
#include "common.h"

namespace amdarrow {

constexpr int kDmThreads = 256;
constexpr unsigned long long kNoOwner = 0x7FFFFFFFFFFFFFFFull;  // all-reduce(MIN) over int64: larger than any owner word
constexpr int kDmTile = 4096;

int LaunchScanU32(amdarrow_ctx* ctx, const uint32_t* counts, int64_t n, int64_t* offsets);  // selection.hip

// acc: [R + 2][2] 8-byte words ({sum, count} of a key side by side -- one 16-byte access instead of two scattered
// 8-byte ones; column R = the null key, the sum word of column R + 1 = number of keys outside the planned range);
// owner: [R + 2], the last word carries
// -(local group count) through the MIN all-reduce: afterwards it is -(the largest local table), the
// stride of the slot array.
__global__ __launch_bounds__(kDmThreads) void k_dense_clear(unsigned long long* acc, unsigned long long* owner, int64_t R) {
  for (int64_t i = int64_t(blockIdx.x) * kDmThreads + threadIdx.x; i < R + 2; i += int64_t(gridDim.x) * kDmThreads) {
    reinterpret_cast<ulonglong2*>(acc)[i] = make_ulonglong2(0, 0);
    if (i <= R) owner[i] = kNoOwner;
    if (i == R + 1) owner[i] = 0;
  }
}
__device__ __forceinline__ int64_t SlotStride(const unsigned long long* owner, int64_t R, int64_t capacity) {
  const int64_t g = -static_cast<int64_t>(owner[R + 1]);
  return g < 1 ? 1 : (g > capacity ? capacity : g);
}

__device__ __forceinline__ int64_t LoadKey(const void* keys, int key_type, int64_t g) {
  switch (key_type) {
    case AMDARROW_INT8: return static_cast<const int8_t*>(keys)[g];
    case AMDARROW_UINT8: return static_cast<const uint8_t*>(keys)[g];
    case AMDARROW_INT16: return static_cast<const int16_t*>(keys)[g];
    case AMDARROW_UINT16: return static_cast<const uint16_t*>(keys)[g];
    case AMDARROW_INT32: return static_cast<const int32_t*>(keys)[g];
    case AMDARROW_UINT32: return static_cast<const uint32_t*>(keys)[g];
    default: return static_cast<const int64_t*>(keys)[g];
  }
}

// One local group per thread: its sum / count / owner word go to the key's dense column.
__global__ __launch_bounds__(kDmThreads) void k_dense_scatter(const void* keys, int key_type, const uint8_t* key_valid, int valid_is_bitmap,
                                                              const unsigned long long* sums, const long long* counts,
                                                              int64_t num_groups, int64_t kmin, int64_t R, int rank, int counts_as_double,
                                                              unsigned long long* acc, unsigned long long* owner) {
  unsigned long long outside = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) owner[R + 1] = static_cast<unsigned long long>(-num_groups);
  for (int64_t g = int64_t(blockIdx.x) * kDmThreads + threadIdx.x; g < num_groups; g += int64_t(gridDim.x) * kDmThreads) {
    int64_t col = R;  // the null key
    if (!key_valid || (valid_is_bitmap ? GetBit(key_valid, g) : key_valid[g] != 0)) {
      col = LoadKey(keys, key_type, g) - kmin;
      if (col < 0 || col >= R) { ++outside; continue; }
    }
    const long long c = counts[g];
    reinterpret_cast<ulonglong2*>(acc)[col] = make_ulonglong2(sums[g], counts_as_double ? static_cast<unsigned long long>(__double_as_longlong(static_cast<double>(c)))
                                                                                         : static_cast<unsigned long long>(c));
    owner[col] = (static_cast<unsigned long long>(rank) << 32) | static_cast<unsigned long long>(g);
  }
  outside = WaveReduceSum64(outside);
  if ((threadIdx.x & 63) == 0 && outside) {
    if (counts_as_double) atomicAdd(reinterpret_cast<double*>(&acc[2 * (R + 1)]), static_cast<double>(outside));
    else atomicAdd(&acc[2 * (R + 1)], outside);
  }
}

// After the all-reduces: every present key drops (column + 1) into the slot its owner word names.
__global__ __launch_bounds__(kDmThreads) void k_dense_clear_slots(const unsigned long long* owner, int64_t R, int64_t capacity,
                                                                  int world, uint32_t* slots) {
  const int64_t nslots = world * SlotStride(owner, R, capacity);
  for (int64_t i = int64_t(blockIdx.x) * kDmThreads + threadIdx.x; i < nslots; i += int64_t(gridDim.x) * kDmThreads) slots[i] = 0;
}
__global__ __launch_bounds__(kDmThreads) void k_dense_slots(const unsigned long long* owner, int64_t R, int64_t capacity, int world,
                                                            uint32_t* slots) {
  const int64_t stride = SlotStride(owner, R, capacity);
  for (int64_t i = int64_t(blockIdx.x) * kDmThreads + threadIdx.x; i <= R; i += int64_t(gridDim.x) * kDmThreads) {
    const unsigned long long o = owner[i];
    if (o == kNoOwner) continue;
    const int64_t r = static_cast<int64_t>(o >> 32), g = static_cast<int64_t>(o & 0xFFFFFFFFull);
    if (r < world && g < stride) slots[r * stride + g] = static_cast<uint32_t>(i + 1);
  }
}

// counts[tile] for every tile of the host's upper bound; tiles past the live slots count 0
__global__ __launch_bounds__(kDmThreads) void k_dense_tile_counts(const uint32_t* slots, const unsigned long long* owner, int64_t R,
                                                                  int64_t capacity, int world, int64_t ntiles, uint32_t* counts) {
  __shared__ uint32_t s_c;
  const int64_t nslots = world * SlotStride(owner, R, capacity);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x == 0) s_c = 0;
    __syncthreads();
    uint32_t c = 0;
    for (int k = 0; k < kDmTile / kDmThreads; ++k) {
      const int64_t i = tile * kDmTile + k * kDmThreads + threadIdx.x;
      c += i < nslots && slots[i] != 0;
    }
    c = static_cast<uint32_t>(WaveReduceSum64(c));
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&s_c, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[tile] = s_c;
    __syncthreads();
  }
}

// Emit the merged table in slot order = global first-appearance order.
__global__ __launch_bounds__(kDmThreads) void k_dense_emit(const uint32_t* slots, const unsigned long long* owner, int64_t capacity,
                                                           int world, const int64_t* offsets, int64_t ntiles,
                                                           const unsigned long long* acc, int64_t R, int64_t kmin, int key_type,
                                                           int counts_as_double, int64_t out_capacity, void* out_keys,
                                                           uint8_t* out_key_valid /* bytes */, unsigned long long* out_sums,
                                                           long long* out_counts, long long* out_num_groups) {
  __shared__ uint32_t s_wave[kDmThreads / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nslots = world * SlotStride(owner, R, capacity);
  if (blockIdx.x == 0 && threadIdx.x == 0) *out_num_groups = offsets[ntiles];
  for (int64_t tile = blockIdx.x; tile < ntiles && tile * kDmTile < nslots; tile += gridDim.x) {
    int64_t base = offsets[tile];
    if (offsets[tile + 1] == base) continue;  // (uniform) an empty tile: with W ranks the slot array is W tables long and mostly the first rank's is filled
    for (int k = 0; k < kDmTile / kDmThreads; ++k) {  // 16 rounds of 256 slots, in order
      const int64_t i = tile * kDmTile + k * kDmThreads + threadIdx.x;
      const uint32_t v = i < nslots ? slots[i] : 0u;
      const uint64_t m = __ballot(v != 0);
      if (lane == 0) s_wave[wave] = __popcll(m);
      __syncthreads();
      uint32_t before = 0, total = 0;
      for (int w = 0; w < kDmThreads / 64; ++w) { if (w < wave) before += s_wave[w]; total += s_wave[w]; }
      if (v != 0) {
        const int64_t pos = base + before + RankBelowLane(m);
        if (pos < out_capacity) {
          const int64_t col = static_cast<int64_t>(v) - 1;
          const int64_t key = col == R ? 0 : kmin + col;
          switch (key_type) {
            case AMDARROW_INT8: case AMDARROW_UINT8: static_cast<int8_t*>(out_keys)[pos] = static_cast<int8_t>(key); break;
            case AMDARROW_INT16: case AMDARROW_UINT16: static_cast<int16_t*>(out_keys)[pos] = static_cast<int16_t>(key); break;
            case AMDARROW_INT32: case AMDARROW_UINT32: static_cast<int32_t*>(out_keys)[pos] = static_cast<int32_t>(key); break;
            default: static_cast<int64_t*>(out_keys)[pos] = key; break;
          }
          out_key_valid[pos] = col != R;
          const ulonglong2 sc = reinterpret_cast<const ulonglong2*>(acc)[col];
          out_sums[pos] = sc.x;
          const unsigned long long c = sc.y;
          out_counts[pos] = counts_as_double ? static_cast<long long>(__longlong_as_double(static_cast<long long>(c)))
                                             : static_cast<long long>(c);
        }
      }
      base += total;
      __syncthreads();
    }
  }
}

}  // namespace amdarrow

using namespace amdarrow;

extern "C" {

int amdarrow_dense_merge_scatter(amdarrow_ctx* ctx, const void* keys, int key_type, const uint8_t* key_valid, int valid_is_bitmap, const void* sums,
                                 const int64_t* counts, int64_t num_groups, int64_t kmin, int64_t R, int rank, int counts_as_double,
                                 void* acc, void* owner) {
  if (R < 0 || num_groups < 0 || BitWidth(key_type) < 8) return SetError(AMDARROW_INVALID, "dense merge: bad plan");
  std::lock_guard<std::mutex> lock(ctx->mu);
  hipLaunchKernelGGL(k_dense_clear, dim3(GridFor(R + 2, kDmThreads, ctx->num_cus * 8)), dim3(kDmThreads), 0, ctx->stream,
                     static_cast<unsigned long long*>(acc), static_cast<unsigned long long*>(owner), R);
  if (num_groups > 0)
    hipLaunchKernelGGL(k_dense_scatter, dim3(GridFor(num_groups, kDmThreads, ctx->num_cus * 8)), dim3(kDmThreads), 0, ctx->stream, keys,
                       key_type, key_valid, valid_is_bitmap, static_cast<const unsigned long long*>(sums), reinterpret_cast<const long long*>(counts),
                       num_groups, kmin, R, rank, counts_as_double, static_cast<unsigned long long*>(acc),
                       static_cast<unsigned long long*>(owner));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_dense_merge_order(amdarrow_ctx* ctx, const void* acc, const void* owner, int64_t kmin, int64_t R, int world,
                               int64_t capacity, int key_type, int counts_as_double, int64_t out_capacity, void* out_keys,
                               uint8_t* out_key_valid_bytes, void* out_sums, int64_t* out_counts, int64_t* out_num_groups_dev) {
  if (world < 1 || capacity < 1 || R < 0) return SetError(AMDARROW_INVALID, "dense merge: bad plan");
  const int64_t nslots = int64_t(world) * capacity;
  if (nslots >= (int64_t(1) << 32)) return SetError(AMDARROW_NOT_IMPLEMENTED, "dense merge: 2^32 slots or more");
  const int64_t ntiles = (nslots + kDmTile - 1) / kDmTile;
  std::lock_guard<std::mutex> lock(ctx->mu);
  Carver c;
  const size_t o_slots = c.Take(4 * static_cast<size_t>(nslots));
  const size_t o_counts = c.Take(4 * static_cast<size_t>(ntiles));
  const size_t o_offsets = c.Take(8 * static_cast<size_t>(ntiles + 1));
  void* base;
  AA_TRY(Scratch(ctx, c.off, &base));
  char* b = static_cast<char*>(base);
  auto* slots = reinterpret_cast<uint32_t*>(b + o_slots);
  auto* counts = reinterpret_cast<uint32_t*>(b + o_counts);
  auto* offsets = reinterpret_cast<int64_t*>(b + o_offsets);
  const auto* own = static_cast<const unsigned long long*>(owner);
  hipLaunchKernelGGL(k_dense_clear_slots, dim3(ctx->num_cus * 8), dim3(kDmThreads), 0, ctx->stream, own, R, capacity, world, slots);
  hipLaunchKernelGGL(k_dense_slots, dim3(GridFor(R + 1, kDmThreads, ctx->num_cus * 8)), dim3(kDmThreads), 0, ctx->stream, own, R,
                     capacity, world, slots);
  hipLaunchKernelGGL(k_dense_tile_counts, dim3(GridFor(ntiles, 1, ctx->num_cus * 8)), dim3(kDmThreads), 0, ctx->stream, slots, own, R,
                     capacity, world, ntiles, counts);
  AA_HIP(hipGetLastError());
  AA_TRY(LaunchScanU32(ctx, counts, ntiles, offsets));
  hipLaunchKernelGGL(k_dense_emit, dim3(GridFor(ntiles, 1, ctx->num_cus * 8)), dim3(kDmThreads), 0, ctx->stream, slots, own, capacity,
                     world, offsets, ntiles, static_cast<const unsigned long long*>(acc), R, kmin, key_type, counts_as_double, out_capacity, out_keys,
                     out_key_valid_bytes, static_cast<unsigned long long*>(out_sums), reinterpret_cast<long long*>(out_counts),
                     reinterpret_cast<long long*>(out_num_groups_dev));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

}  // extern "C"

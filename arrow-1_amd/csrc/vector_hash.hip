// vector_hash.hip -- the device step dictionary_encode adds to the group-by machinery
// (compute/kernels/vector_hash.cc:165-225, 617-627).  unique / value_counts /
// dictionary_encode are the Grouper (first-appearance ids + uniques, groupby.hip) seen through the
// reference's three hash actions: unique = the uniques, value_counts = uniques + rows per id,
// dictionary_encode = the ids as int32 indices.  Under DictionaryEncodeOptions::MASK (the default)
// a null is not a dictionary entry (DictEncodeAction::ShouldEncodeNulls, :210-212): its rows get a
// null index and the ids after the null group move down by one -- that renumbering is this file.
#include "common.h"

namespace amdarrow {

constexpr int kVThreads = 256;

// The null key's group = the one zero bit of the uniques' validity (at most one group is null).
__global__ __launch_bounds__(kVThreads) void k_find_null_group(const uint64_t* valid, int64_t num_groups, uint32_t* null_group) {
  const int64_t nwords = (num_groups + 63) >> 6;
  for (int64_t w = int64_t(blockIdx.x) * kVThreads + threadIdx.x; w < nwords; w += int64_t(gridDim.x) * kVThreads) {
    const int64_t left = num_groups - (w << 6);
    const uint64_t live = left >= 64 ? ~uint64_t(0) : (uint64_t(1) << left) - 1;
    const uint64_t zeros = ~valid[w] & live;
    if (zeros) atomicMin(null_group, static_cast<uint32_t>((w << 6) + __ffsll(static_cast<long long>(zeros)) - 1));
  }
}

// One wave writes one validity word.  A masked row's index value is 0 (Int32Builder::UnsafeAppendNull).
__global__ __launch_bounds__(kVThreads) void k_dictionary_indices(const uint32_t* ids, int64_t n, const uint32_t* null_group,
                                                                  int32_t* out_indices, uint64_t* out_validity,
                                                                  unsigned long long* counters) {
  const uint32_t ng = *null_group;
  const int lane = threadIdx.x & 63;
  unsigned long long nulls = 0;
  const int64_t nwords = (n + 63) >> 6;
  for (int64_t w = (int64_t(blockIdx.x) * kVThreads + threadIdx.x) >> 6; w < nwords; w += (int64_t(gridDim.x) * kVThreads) >> 6) {
    const int64_t i = (w << 6) + lane;
    bool valid = false;
    if (i < n) {
      const uint32_t id = ids[i];
      valid = id != ng;
      out_indices[i] = valid ? static_cast<int32_t>(id - (id > ng ? 1u : 0u)) : 0;
    }
    const uint64_t bits = __ballot(valid);
    if (lane == 0) {
      out_validity[w] = bits;
      const int64_t left = n - (w << 6);
      nulls += (left >= 64 ? 64 : left) - __popcll(bits);
    }
  }
  if (lane == 0 && nulls) atomicAdd(CounterSlot(counters), nulls);
}

// Grouper::MakeGroupings (hash_aggregate.cc:979-1018) from the stable sort of the ids: position i of the sorted
// order holds row sorted[i] -- that IS the grouping's value i -- and group g starts where the sorted ids first
// reach g.  The thread at a boundary writes the offsets of every group that starts there (the empty ones between
// the previous id and this one included); the last position also closes the groups behind the largest id.
__global__ __launch_bounds__(kVThreads) void k_groupings_from_sorted(const uint32_t* ids, const uint64_t* sorted, int64_t n,
                                                                    uint32_t num_groups, int32_t* out_offsets,
                                                                    int32_t* out_indices, unsigned int* bad) {
  for (int64_t i = int64_t(blockIdx.x) * kVThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kVThreads) {
    const uint64_t row = sorted[i];
    out_indices[i] = static_cast<int32_t>(row);
    const uint32_t g = ids[row];
    if (g >= num_groups) { *bad = 1; continue; }
    const int64_t first = i == 0 ? 0 : int64_t(ids[sorted[i - 1]]) + 1;
    for (int64_t k = first; k <= int64_t(g); ++k) out_offsets[k] = static_cast<int32_t>(i);
    if (i == n - 1)
      for (int64_t k = int64_t(g) + 1; k <= int64_t(num_groups); ++k) out_offsets[k] = static_cast<int32_t>(n);
  }
}

}  // namespace amdarrow

using namespace amdarrow;

extern "C" int amdarrow_dictionary_indices(amdarrow_ctx* ctx, const uint32_t* ids, int64_t n, const uint8_t* uniques_validity,
                                           int64_t num_groups, int32_t* out_indices, uint8_t* out_validity, amdarrow_out* out) {
  out->length = n;
  out->null_count = 0;
  out->has_validity = 0;
  if (n == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  void* base;
  AA_TRY(Scratch(ctx, kCounterBytes + 256, &base));
  auto* counters = static_cast<unsigned long long*>(base);
  auto* null_group = reinterpret_cast<uint32_t*>(static_cast<char*>(base) + kCounterBytes);
  AA_TRY(ClearCounters(ctx, counters));
  AA_HIP(hipMemsetAsync(null_group, 0xFF, 4, ctx->stream));
  if (uniques_validity && num_groups > 0) {
    hipLaunchKernelGGL(k_find_null_group, dim3(GridFor((num_groups + 63) / 64, kVThreads, ctx->num_cus * 4)), dim3(kVThreads), 0,
                       ctx->stream, reinterpret_cast<const uint64_t*>(uniques_validity), num_groups, null_group);
    AA_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_dictionary_indices, dim3(GridFor(n, kVThreads, ctx->num_cus * 8)), dim3(kVThreads), 0, ctx->stream, ids, n,
                     null_group, out_indices, reinterpret_cast<uint64_t*>(out_validity), counters);
  AA_HIP(hipGetLastError());
  unsigned long long nn = 0;
  AA_TRY(ReadCounters(ctx, counters, 1, &nn));
  out->null_count = static_cast<int64_t>(nn);
  out->has_validity = nn > 0;  // the builder materialises a bitmap only once a null was appended
  return AMDARROW_OK;
}

extern "C" int amdarrow_make_groupings(amdarrow_ctx* ctx, const uint32_t* ids, int64_t n, uint32_t num_groups,
                                       int32_t* out_offsets, int32_t* out_indices) {
  using namespace amdarrow;
  if (n < 0 || n > INT32_MAX) return SetError(AMDARROW_INVALID, "MakeGroupings: more rows than int32 offsets address");
  if (n == 0) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    AA_HIP(hipMemsetAsync(out_offsets, 0, 4 * (static_cast<size_t>(num_groups) + 1), ctx->stream));
    return AMDARROW_OK;
  }
  uint64_t* sorted = nullptr;
  AA_TRY(DeviceMalloc(ctx, reinterpret_cast<void**>(&sorted), 8 * static_cast<size_t>(n)));
  amdarrow_array chunk;
  chunk.values = ids; chunk.validity = nullptr; chunk.length = n; chunk.offset = 0;
  chunk.type_id = AMDARROW_UINT32; chunk.null_count = 0;
  int rc = amdarrow_sort_indices(ctx, &chunk, 1, AMDARROW_ASCENDING, sorted);  // stable: rows of a group stay in row order
  if (rc == AMDARROW_OK) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    void* base;
    rc = Scratch(ctx, 256, &base);
    if (rc == AMDARROW_OK) {
      auto* bad = static_cast<unsigned int*>(base);
      hipError_t e = hipMemsetAsync(bad, 0, 4, ctx->stream);
      hipLaunchKernelGGL(k_groupings_from_sorted, dim3(GridFor(n, kVThreads, ctx->num_cus * 8)), dim3(kVThreads), 0, ctx->stream,
                         ids, sorted, n, num_groups, out_offsets, out_indices, bad);
      if (e == hipSuccess) e = hipGetLastError();
      unsigned int flag = 0;
      if (e != hipSuccess) rc = SetError(AMDARROW_UNKNOWN_ERROR, hipGetErrorString(e));
      else rc = ReadBack(ctx, bad, 4, &flag);  // also orders the free below after the kernel
      if (rc == AMDARROW_OK && flag) rc = SetError(AMDARROW_INVALID, "MakeGroupings: an id is not below num_groups");
    }
  }
  (void)hipFree(sorted);
  return rc;
}

// selection_nested.hip -- Take / Filter for the value types that are not plain primitives:
// fixed_size_binary / decimal (FSBImpl), large_binary / large_utf8 (VarBinaryImpl<LargeBinaryType>),
// list / large_list / map (ListImpl), fixed_size_list (FSLImpl), struct (StructImpl).
//
// Reference semantics: compute/kernels/vector_selection.cc
//   Selection::VisitTake :1214-1262 (which output rows are valid: index valid AND value valid; the output ALWAYS
//       carries a validity bitmap and an exact null count, FinishCommon :1205-1210)
//   FSBImpl :1575-1606 (null slots hold zero bytes), VarBinaryImpl :1493-1573, ListImpl :1608-1680 (a null slot is an
//       empty list; child = Take(values.values(), child indices)), FSLImpl :1682-1724 (list_size NULL child indices
//       per null slot), StructImpl :1736-1763 (validity only; every field is taken with the same indices),
//   TakeExec :1330-1345 (CheckIndexBounds first, IndexError), Filter of these types = Take over GetTakeIndices(filter)
//       (FilterExec visits the same rows in the same order; StructFilter :1765-1781 does literally that).
//
// One decomposition serves all of them: amdarrow_take_rows resolves every output row to the SOURCE ROW it copies
// (-1 = null) -- that is VisitTake, and StructImpl's whole job -- and the per-layout calls consume the resolved rows:
// fixed-width bytes are gathered, offsets-based layouts go through lengths -> 64-bit scan -> offsets and then either
// the bytes (binary) or the child indices (lists) are written run by run.  All of it is HBM-bound gather work.
#include "common.h"

#include <cstring>
#include <string>

namespace amdarrow {

constexpr int kNThreads = 256;

struct ResolveArgs {
  const void* i_data; const uint8_t* i_valid; int64_t i_off; int64_t n;
  const uint8_t* v_valid; int64_t v_off; int64_t v_len; int v_may_null;
  int boundscheck;
  int64_t* src; uint8_t* out_validity;
  unsigned long long* valid_count;  // slotted counters, counter 0
  unsigned long long* err_pos;      // initialised to ~0
};

// Four rows per lane and step; every stage's loads are stored raw before one of them is looked at (see k_take).
template <typename IdxT>
__global__ __launch_bounds__(kNThreads) void k_resolve_rows(ResolveArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long my_valid = 0;
  const int64_t wave_id = (int64_t(blockIdx.x) * kNThreads + tid) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kNThreads) >> 6;
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
    uint32_t vvb[kU];
    const bool need_vbits = a.v_may_null && a.v_len > 0;
#pragma unroll
    for (int u = 0; u < kU; ++u) vvb[u] = need_vbits ? a.v_valid[(a.v_off + j[u]) >> 3] : 0xFFu;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t i = base + 64 * u + lane;
      if (base + 64 * u >= a.n) break;  // wave-uniform
      const bool good = ok[u] && ((vvb[u] >> ((a.v_off + j[u]) & 7)) & 1u);
      const uint64_t okmask = __ballot(good);
      if (i < a.n) a.src[i] = good ? j[u] : int64_t(-1);
      if (lane == 0) {
        reinterpret_cast<uint64_t*>(a.out_validity)[i >> 6] = okmask;
        my_valid += __popcll(okmask);
      }
    }
  }
  if (lane == 0 && my_valid) atomicAdd(CounterSlot(a.valid_count), my_valid);
}

// FSBImpl: out row i = the W bytes of source row src[i], or zeros.  A thread moves one C-byte piece (C = the largest
// of 16 / 8 / 4 / 2 / 1 that divides W and the base addresses): writes are coalesced, reads are the gather.
template <typename Piece>
__global__ __launch_bounds__(kNThreads) void k_gather_rows(const Piece* values, int64_t v_off_pieces, int pieces_per_row,
                                                           const int64_t* src, int64_t npieces, Piece* out) {
  for (int64_t g = int64_t(blockIdx.x) * kNThreads + threadIdx.x; g < npieces; g += int64_t(gridDim.x) * kNThreads) {
    const int64_t row = g / pieces_per_row;
    const int c = static_cast<int>(g - row * pieces_per_row);
    const int64_t s = src[row];
    Piece v{};
    if (s >= 0) v = values[v_off_pieces + s * pieces_per_row + c];
    out[g] = v;
  }
}

// Lengths of the taken variable-length slots (0 for a null slot), as int64 for the scan.
template <typename OffT>
__global__ __launch_bounds__(kNThreads) void k_var_lengths(const OffT* offsets, int64_t v_off, const int64_t* src, int64_t n,
                                                           long long* lengths) {
  for (int64_t i = int64_t(blockIdx.x) * kNThreads + threadIdx.x; i < n; i += int64_t(gridDim.x) * kNThreads) {
    const int64_t s = src[i];
    long long len = 0;
    if (s >= 0) len = static_cast<long long>(offsets[v_off + s + 1]) - static_cast<long long>(offsets[v_off + s]);
    lengths[i] = len;
  }
}

// ---- exclusive scan of int64, device-wide (the totals of large_binary / large_list do not fit 32 bits) --------------
constexpr int kScanChunk = 2048;  // elements per block
__global__ __launch_bounds__(kNThreads) void k_scan64_block_sums(const long long* data, int64_t n, long long* sums) {
  __shared__ long long s_part[kNThreads / 64];
  const int64_t base = int64_t(blockIdx.x) * kScanChunk;
  long long mine = 0;
  for (int k = threadIdx.x; k < kScanChunk; k += kNThreads) if (base + k < n) mine += data[base + k];
  mine = static_cast<long long>(WaveReduceSum64(static_cast<uint64_t>(mine)));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) { long long t = 0; for (int w = 0; w < kNThreads / 64; ++w) t += s_part[w]; sums[blockIdx.x] = t; }
}
// one workgroup: sums -> exclusive offsets in place; total -> sums[nb]
__global__ __launch_bounds__(1024) void k_scan64_sums(long long* sums, int64_t nb) {
  __shared__ long long s_wave[16];
  __shared__ long long s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + tid;
    const long long v = i < nb ? sums[i] : 0;
    long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long t = static_cast<long long>(__shfl_up(static_cast<unsigned long long>(incl), d, 64));
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    long long woff = s_carry;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    if (i < nb) sums[i] = woff + incl - v;
    __syncthreads();
    if (tid == 1023) s_carry = woff + incl;
    __syncthreads();
  }
  if (tid == 0) sums[nb] = s_carry;
}
// lengths -> offsets of type OffT (out[n] = total); a 32-bit total past INT32_MAX raises *overflow
template <typename OffT>
__global__ __launch_bounds__(kNThreads) void k_scan64_apply(const long long* lengths, int64_t n, const long long* block_offsets,
                                                            int64_t nb, OffT* out, unsigned int* overflow) {
  __shared__ long long s_wave[kNThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t base = int64_t(blockIdx.x) * kScanChunk;
  long long carry = block_offsets[blockIdx.x];
  for (int k = 0; k < kScanChunk; k += kNThreads) {
    const int64_t i = base + k + tid;
    const long long v = i < n ? lengths[i] : 0;
    long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const long long t = static_cast<long long>(__shfl_up(static_cast<unsigned long long>(incl), d, 64));
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    long long woff = 0, total = 0;
    for (int w = 0; w < kNThreads / 64; ++w) { if (w < wave) woff += s_wave[w]; total += s_wave[w]; }
    if (i < n) out[i] = static_cast<OffT>(carry + woff + incl - v);
    carry += total;
    __syncthreads();
  }
  if (blockIdx.x == 0 && tid == 0) {
    const long long total = block_offsets[nb];
    out[n] = static_cast<OffT>(total);
    if (sizeof(OffT) == 4 && total > 2147483646LL) *overflow = 1;  // kOffsetLimit = INT32_MAX - 1 (:1540-1544)
  }
}

// The runs behind the offsets.  BYTES: out_data[out_off[i] + t] = data[src_off + t]; else (lists) the child indices
// out_idx[out_off[i] + t] = src_off + t.  A lane owns a row: runs of up to kShortRun elements are written by their lane
// (eight rows of a wave in flight per step keep the loads independent), longer runs by the whole wave.
template <typename OffT, bool BYTES>
__global__ __launch_bounds__(kNThreads) void k_var_runs(const OffT* offsets, int64_t v_off, const uint8_t* data, const int64_t* src,
                                                        int64_t n, const OffT* out_offsets, void* out) {
  constexpr int64_t kShortRun = 64;
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t(blockIdx.x) * kNThreads + threadIdx.x) >> 6;
  const int64_t nwaves = (int64_t(gridDim.x) * kNThreads) >> 6;
  for (int64_t base = wave_id * 64; base < n; base += nwaves * 64) {
    const int64_t i = base + lane;
    int64_t dst = 0, len = 0, from = 0;
    if (i < n) {
      const int64_t s = src[i];
      dst = static_cast<int64_t>(out_offsets[i]);
      len = static_cast<int64_t>(out_offsets[i + 1]) - dst;
      from = s >= 0 ? static_cast<int64_t>(offsets[v_off + s]) : 0;
    }
    if (len <= kShortRun) {
      if (BYTES) {
        uint8_t* d = static_cast<uint8_t*>(out) + dst;
        const uint8_t* sp = data + from;
        int64_t t = 0;
        for (; t + 8 <= len; t += 8) { unsigned long long q; __builtin_memcpy(&q, sp + t, 8); __builtin_memcpy(d + t, &q, 8); }
        for (; t < len; ++t) d[t] = sp[t];
      } else {
        OffT* d = static_cast<OffT*>(out) + dst;
        for (int64_t t = 0; t < len; ++t) d[t] = static_cast<OffT>(from + t);
      }
    }
    uint64_t todo = __ballot(len > kShortRun);
    while (todo) {
      const int r = __ffsll(static_cast<long long>(todo)) - 1;
      todo &= todo - 1;
      const int64_t rlen = __shfl(len, r), rdst = __shfl(dst, r), rfrom = __shfl(from, r);
      if (BYTES) for (int64_t t = lane; t < rlen; t += 64) static_cast<uint8_t*>(out)[rdst + t] = data[rfrom + t];
      else for (int64_t t = lane; t < rlen; t += 64) static_cast<OffT*>(out)[rdst + t] = static_cast<OffT>(rfrom + t);
    }
  }
}

// FSLImpl: list_size child indices per output slot, null (value 0, Int64Builder::AppendNulls) for a null slot.
__global__ __launch_bounds__(kNThreads) void k_fsl_child_indices(const int64_t* src, int64_t n, int list_size, int64_t v_off,
                                                                 int64_t* out_idx, uint64_t* out_validity) {
  const int lane = threadIdx.x & 63;
  const int64_t total = n * list_size;
  const int64_t ground = (total + 63) & ~int64_t(63);
  for (int64_t g = int64_t(blockIdx.x) * kNThreads + threadIdx.x; g < ground; g += int64_t(gridDim.x) * kNThreads) {
    bool valid = false;
    if (g < total) {
      const int64_t row = g / list_size;
      const int64_t s = src[row];
      valid = s >= 0;
      out_idx[g] = valid ? (v_off + s) * list_size + (g - row * list_size) : 0;
    }
    const uint64_t vm = __ballot(valid);
    if (lane == 0) out_validity[g >> 6] = vm;
  }
}

int LaunchScan64(amdarrow_ctx* ctx, const long long* lengths, int64_t n, long long* sums /* Scan64Blocks(n) + 1 */, int offset_bits, void* out_offsets,
                 unsigned int* overflow) {
  const int64_t nb = (n + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL(k_scan64_block_sums, dim3(static_cast<unsigned>(nb)), dim3(kNThreads), 0, ctx->stream, lengths, n, sums);
  hipLaunchKernelGGL(k_scan64_sums, dim3(1), dim3(1024), 0, ctx->stream, sums, nb);
  if (offset_bits == 32)
    hipLaunchKernelGGL((k_scan64_apply<int32_t>), dim3(static_cast<unsigned>(nb)), dim3(kNThreads), 0, ctx->stream, lengths, n, sums, nb,
                       static_cast<int32_t*>(out_offsets), overflow);
  else
    hipLaunchKernelGGL((k_scan64_apply<long long>), dim3(static_cast<unsigned>(nb)), dim3(kNThreads), 0, ctx->stream, lengths, n, sums, nb,
                       static_cast<long long*>(out_offsets), overflow);
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

}  // namespace amdarrow

using namespace amdarrow;

extern "C" {

int amdarrow_take_rows(amdarrow_ctx* ctx, const uint8_t* values_validity, int64_t values_length, int64_t values_offset,
                       int64_t values_null_count, const amdarrow_array* indices, int boundscheck, int64_t* out_src_rows,
                       uint8_t* out_validity, amdarrow_out* out) {
  const int it = indices->type_id;
  if (!IsSignedInt(it) && !IsUnsignedInt(it))
    return SetError(AMDARROW_NOT_IMPLEMENTED, "take: indices must be an integer type");  // vector_selection.cc:2067-2070
  out->length = indices->length;
  out->has_validity = 1;  // FinishCommon :1205-1210
  out->null_count = 0;
  if (indices->length == 0) return AMDARROW_OK;
  if (!out_validity || !out_src_rows) return SetError(AMDARROW_INVALID, "take: output buffers required");
  std::lock_guard<std::mutex> lock(ctx->mu);
  void* base;
  AA_TRY(Scratch(ctx, kCounterBytes + 256, &base));
  auto* counters = static_cast<unsigned long long*>(base);
  auto* err_pos = counters + kCounterSlots * kCounterStride;
  AA_TRY(ClearCounters(ctx, counters));
  AA_HIP(hipMemsetAsync(err_pos, 0xFF, 8, ctx->stream));
  ResolveArgs a{};
  a.i_data = indices->values; a.i_valid = indices->validity; a.i_off = indices->offset; a.n = indices->length;
  a.v_valid = values_validity; a.v_off = values_offset; a.v_len = values_length;
  a.v_may_null = values_validity != nullptr && values_null_count != 0;  // ArrayData::MayHaveNulls (:1222)
  a.boundscheck = boundscheck;
  a.src = out_src_rows; a.out_validity = out_validity;
  a.valid_count = counters; a.err_pos = err_pos;
  const int grid = GridFor(a.n, kNThreads, ctx->num_cus * 8);
  const int iw = BitWidth(it) / 8;
  const bool sgn = IsSignedInt(it);
#define AA_RES(T) hipLaunchKernelGGL((k_resolve_rows<T>), dim3(grid), dim3(kNThreads), 0, ctx->stream, a)
  if (iw == 1) { if (sgn) AA_RES(int8_t); else AA_RES(uint8_t); }
  else if (iw == 2) { if (sgn) AA_RES(int16_t); else AA_RES(uint16_t); }
  else if (iw == 4) { if (sgn) AA_RES(int32_t); else AA_RES(uint32_t); }
  else { if (sgn) AA_RES(int64_t); else AA_RES(uint64_t); }
#undef AA_RES
  AA_HIP(hipGetLastError());
  unsigned long long host[2];
  AA_HIP(hipMemcpyAsync(static_cast<char*>(ctx->pinned) + kStatusOffset, err_pos, 8, hipMemcpyDeviceToHost, ctx->stream));
  AA_TRY(ReadCounters(ctx, counters, 1, host));  // synchronises the stream
  memcpy(&host[1], static_cast<const char*>(ctx->pinned) + kStatusOffset, 8);
  if (boundscheck && host[1] != ~0ull) {  // CheckIndexBounds, util/int_util.cc:553-561
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
  out->null_count = indices->length - static_cast<int64_t>(host[0]);
  return AMDARROW_OK;
}

int amdarrow_gather_fixed_width(amdarrow_ctx* ctx, const void* values, int64_t values_offset, int byte_width,
                                const int64_t* src_rows, int64_t n, void* out_values) {
  if (byte_width < 0) return SetError(AMDARROW_INVALID, "fixed_size_binary: negative byte width");
  if (n == 0 || byte_width == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const uintptr_t align = reinterpret_cast<uintptr_t>(values) | reinterpret_cast<uintptr_t>(out_values) | static_cast<uintptr_t>(byte_width);
  const int c = (align % 16 == 0) ? 16 : (align % 8 == 0) ? 8 : (align % 4 == 0) ? 4 : (align % 2 == 0) ? 2 : 1;
  const int ppr = byte_width / c;
  const int64_t npieces = n * ppr;
  const int grid = GridFor(npieces, kNThreads, ctx->num_cus * 16);
#define AA_GATHER(T) hipLaunchKernelGGL((k_gather_rows<T>), dim3(grid), dim3(kNThreads), 0, ctx->stream, static_cast<const T*>(values), \
                                        values_offset * ppr, ppr, src_rows, npieces, static_cast<T*>(out_values))
  switch (c) {
    case 16: AA_GATHER(uint4); break;
    case 8: AA_GATHER(unsigned long long); break;
    case 4: AA_GATHER(uint32_t); break;
    case 2: AA_GATHER(uint16_t); break;
    default: AA_GATHER(uint8_t); break;
  }
#undef AA_GATHER
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_var_take_offsets(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n,
                              void* out_offsets, int64_t* out_total) {
  if (values->offset_bits != 32 && values->offset_bits != 64) return SetError(AMDARROW_INVALID, "offsets must be 32 or 64 bits wide");
  *out_total = 0;
  std::lock_guard<std::mutex> lock(ctx->mu);
  if (n == 0) {
    AA_HIP(hipMemsetAsync(out_offsets, 0, values->offset_bits / 8, ctx->stream));
    return AMDARROW_OK;
  }
  const int64_t nb = (n + kScanChunk - 1) / kScanChunk;
  Carver cv;
  const size_t o_len = cv.Take(8 * static_cast<size_t>(n));
  const size_t o_sums = cv.Take(8 * static_cast<size_t>(nb + 1));
  const size_t o_flag = cv.Take(256);
  void* base;
  AA_TRY(Scratch(ctx, cv.off, &base));
  char* b = static_cast<char*>(base);
  auto* lengths = reinterpret_cast<long long*>(b + o_len);
  auto* sums = reinterpret_cast<long long*>(b + o_sums);
  auto* flag = reinterpret_cast<unsigned int*>(b + o_flag);
  AA_HIP(hipMemsetAsync(flag, 0, 4, ctx->stream));
  const int grid = GridFor(n, kNThreads, ctx->num_cus * 8);
  if (values->offset_bits == 32)
    hipLaunchKernelGGL((k_var_lengths<int32_t>), dim3(grid), dim3(kNThreads), 0, ctx->stream, static_cast<const int32_t*>(values->offsets),
                       values->offset, src_rows, n, lengths);
  else
    hipLaunchKernelGGL((k_var_lengths<long long>), dim3(grid), dim3(kNThreads), 0, ctx->stream, static_cast<const long long*>(values->offsets),
                       values->offset, src_rows, n, lengths);
  AA_HIP(hipGetLastError());
  AA_TRY(LaunchScan64(ctx, lengths, n, sums, values->offset_bits, out_offsets, flag));
  long long total = 0;
  unsigned int over = 0;
  AA_TRY(ReadBack(ctx, sums + nb, 8, &total));
  AA_TRY(ReadBack(ctx, flag, 4, &over));
  if (over) return SetError(AMDARROW_INVALID, "Take operation overflowed binary array capacity");  // :1540-1544
  *out_total = total;
  return AMDARROW_OK;
}

static int VarRuns(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n, const void* out_offsets,
                   void* out, bool bytes) {
  if (values->offset_bits != 32 && values->offset_bits != 64) return SetError(AMDARROW_INVALID, "offsets must be 32 or 64 bits wide");
  if (n == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const int grid = GridFor(n, kNThreads / 4, ctx->num_cus * 16);  // 64 rows per wave and step
#define AA_RUNS(T, B) hipLaunchKernelGGL((k_var_runs<T, B>), dim3(grid), dim3(kNThreads), 0, ctx->stream, static_cast<const T*>(values->offsets), \
                                         values->offset, values->data, src_rows, n, static_cast<const T*>(out_offsets), out)
  if (values->offset_bits == 32) { if (bytes) AA_RUNS(int32_t, true); else AA_RUNS(int32_t, false); }
  else { if (bytes) AA_RUNS(long long, true); else AA_RUNS(long long, false); }
#undef AA_RUNS
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

int amdarrow_var_take_data(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n,
                           const void* out_offsets, uint8_t* out_data) {
  return VarRuns(ctx, values, src_rows, n, out_offsets, out_data, true);
}

int amdarrow_list_take_child_indices(amdarrow_ctx* ctx, const amdarrow_var_array* values, const int64_t* src_rows, int64_t n,
                                     const void* out_offsets, void* out_child_indices) {
  return VarRuns(ctx, values, src_rows, n, out_offsets, out_child_indices, false);
}

int amdarrow_fixed_size_list_child_indices(amdarrow_ctx* ctx, const int64_t* src_rows, int64_t n, int list_size,
                                           int64_t values_offset, int64_t* out_indices, uint8_t* out_index_validity) {
  if (list_size < 0) return SetError(AMDARROW_INVALID, "fixed_size_list: negative list size");
  if (n == 0 || list_size == 0) return AMDARROW_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const int64_t total = n * list_size;
  hipLaunchKernelGGL(k_fsl_child_indices, dim3(GridFor(total, kNThreads, ctx->num_cus * 16)), dim3(kNThreads), 0, ctx->stream, src_rows, n,
                     list_size, values_offset, out_indices, reinterpret_cast<uint64_t*>(out_index_validity));
  AA_HIP(hipGetLastError());
  return AMDARROW_OK;
}

}  // extern "C"

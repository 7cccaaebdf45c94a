// runtime.hip -- context, device memory pool, error plumbing for the C ABI.
#include "common.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace amdarrow {

static thread_local std::string g_last_error;

int SetError(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
int HipError(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", static_cast<int>(e), hipGetErrorString(e), what,
           file, line);
  return SetError(e == hipErrorOutOfMemory ? AMDARROW_OUT_OF_MEMORY : AMDARROW_UNKNOWN_ERROR, buf);
}

int Scratch(amdarrow_ctx* ctx, size_t bytes, void** out) {
  if (bytes > ctx->scratch_cap) {
    // Growing the arena is a (rare) synchronising event: earlier kernels may still use it.
    AA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) AA_HIP(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_cap = 0;
    size_t cap = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    AA_TRY(DeviceMalloc(ctx, &ctx->scratch, cap));
    ctx->scratch_cap = cap;
  }
  *out = ctx->scratch;
  return AMDARROW_OK;
}

int64_t DrainPool(amdarrow_ctx* ctx) {
  std::vector<void*> drop;
  int64_t bytes = 0;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    for (auto& kv : ctx->free_blocks) { for (void* q : kv.second) drop.push_back(q); kv.second.clear(); }
    bytes = ctx->cached_bytes;
    ctx->cached_bytes = 0;
  }
  if (drop.empty()) return 0;
  (void)hipStreamSynchronize(ctx->stream);  // a cached block may still be read by work queued before its free
  for (void* q : drop) (void)hipFree(q);
  return bytes;
}

int DeviceMalloc(amdarrow_ctx* ctx, void** out, size_t bytes) {
  hipError_t e = hipMalloc(out, bytes);
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    if (DrainPool(ctx) > 0) e = hipMalloc(out, bytes);
  }
  if (e != hipSuccess) { *out = nullptr; return HipError(e, "hipMalloc", __FILE__, __LINE__); }
  return AMDARROW_OK;
}

int ReadBack(amdarrow_ctx* ctx, const void* dev, size_t nbytes, void* host_out) {
  if (nbytes > 4096) return SetError(AMDARROW_INVALID, "ReadBack: too large");
  AA_HIP(hipMemcpyAsync(ctx->pinned, dev, nbytes, hipMemcpyDeviceToHost, ctx->stream));
  AA_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(host_out, ctx->pinned, nbytes);
  return AMDARROW_OK;
}

int ClearCounters(amdarrow_ctx* ctx, unsigned long long* base) {
  AA_HIP(hipMemsetAsync(base, 0, kCounterBytes, ctx->stream));
  return AMDARROW_OK;
}
int ReadCounters(amdarrow_ctx* ctx, const unsigned long long* base, int ncounters, unsigned long long* out) {
  unsigned long long host[kCounterSlots * kCounterStride];
  AA_TRY(ReadBack(ctx, base, kCounterBytes, host));
  for (int k = 0; k < ncounters; ++k) {
    out[k] = 0;
    for (int s = 0; s < kCounterSlots; ++s) out[k] += host[s * kCounterStride + k];
  }
  return AMDARROW_OK;
}

}  // namespace amdarrow

using namespace amdarrow;

extern "C" {

int amdarrow_abi_version(void) { return AMDARROW_ABI_VERSION; }

const char* amdarrow_last_error(void) { return g_last_error.c_str(); }

int amdarrow_device_count(int* out_count) {
  AA_HIP(hipGetDeviceCount(out_count));
  return AMDARROW_OK;
}

int amdarrow_ctx_create(int device, void* stream, int private_stream, amdarrow_ctx** out_ctx) {
  int n = 0;
  AA_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return SetError(AMDARROW_INVALID, "amdarrow_ctx_create: no such device");
  AA_HIP(hipSetDevice(device));
  auto* ctx = new amdarrow_ctx;
  ctx->device = device;
  if (!private_stream) {
    ctx->stream = static_cast<hipStream_t>(stream);  // nullptr = the default stream
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return HipError(e, "hipStreamCreate", __FILE__, __LINE__); }
    ctx->own_stream = true;
  }
  hipError_t e = hipHostMalloc(&ctx->pinned, 8192, hipHostMallocMapped);
  if (e != hipSuccess) { delete ctx; return HipError(e, "hipHostMalloc", __FILE__, __LINE__); }
  e = hipHostGetDevicePointer(&ctx->pinned_dev, ctx->pinned, 0);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->status_event, hipEventDisableTiming);
  if (e != hipSuccess) { (void)hipHostFree(ctx->pinned); delete ctx; return HipError(e, "hipEventCreate", __FILE__, __LINE__); }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  // the pool keeps at most an eighth of the device's memory in freed blocks (36 GB on a 288 GB part); AMDARROW_POOL_CACHE_LIMIT
  // (bytes) overrides it, 0 turns the cache off
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) ctx->cache_limit = static_cast<int64_t>(total_b / 8);
  else (void)hipGetLastError();
  if (const char* env = getenv("AMDARROW_POOL_CACHE_LIMIT")) ctx->cache_limit = atoll(env);
  *out_ctx = ctx;
  return AMDARROW_OK;
}

int amdarrow_ctx_destroy(amdarrow_ctx* ctx) {
  if (!ctx) return AMDARROW_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->pack_buf) (void)hipFree(ctx->pack_buf);
  if (ctx->side_buf) (void)hipFree(ctx->side_buf);
  if (ctx->status_event) (void)hipEventDestroy(ctx->status_event);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  for (auto& kv : ctx->allocs) (void)hipFree(kv.first);
  for (auto& kv : ctx->free_blocks) for (void* p : kv.second) (void)hipFree(p);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return AMDARROW_OK;
}

int amdarrow_ctx_synchronize(amdarrow_ctx* ctx) {
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}

void* amdarrow_ctx_stream(amdarrow_ctx* ctx) { return ctx->stream; }

// Size class of a block: capacities up to 4 KB as they are (64-byte multiples), above that rounded up to the next
// multiple of an eighth of their leading power of two (<= 12.5 % over).
static int64_t SizeClass(int64_t cap) {
  if (cap <= 4096) return cap;
  int lg = 63 - __builtin_clzll(static_cast<unsigned long long>(cap));
  const int64_t step = int64_t(1) << (lg - 3);
  return (cap + step - 1) / step * step;
}
int amdarrow_malloc(amdarrow_ctx* ctx, int64_t nbytes, void** out_ptr) {
  if (nbytes < 0) return SetError(AMDARROW_INVALID, "amdarrow_malloc: negative size");
  // capacity rounded up to 64 bytes like arrow::MemoryPool (memory_pool.cc:101, buffer.cc:196-216)
  const int64_t cap = ((nbytes + 63) / 64) * 64 + (nbytes == 0 ? 64 : 0);
  const int64_t klass = SizeClass(cap);
  AA_HIP(hipSetDevice(ctx->device));
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto it = ctx->free_blocks.find(klass);
    if (it != ctx->free_blocks.end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
      ctx->cached_bytes -= klass;
    }
  }
  if (!p) AA_TRY(DeviceMalloc(ctx, &p, static_cast<size_t>(klass)));  // (out of memory: the cache goes back first)
  if (cap > nbytes) {
    // zero the padding so word-at-a-time bitmap readers see defined bytes
    AA_HIP(hipMemsetAsync(static_cast<char*>(p) + nbytes, 0, static_cast<size_t>(cap - nbytes), ctx->stream));
  }
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  ctx->allocs[p] = cap;
  ctx->alloc_class[p] = klass;
  ctx->bytes_allocated += cap;
  if (ctx->bytes_allocated > ctx->max_memory) ctx->max_memory = ctx->bytes_allocated;
  *out_ptr = p;
  return AMDARROW_OK;
}

int amdarrow_free(amdarrow_ctx* ctx, void* ptr) {
  if (!ptr) return AMDARROW_OK;
  int64_t klass = 0;
  bool keep = false;
  {
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto it = ctx->allocs.find(ptr);
    if (it == ctx->allocs.end()) return SetError(AMDARROW_INVALID, "amdarrow_free: pointer not from this pool");
    ctx->bytes_allocated -= it->second;
    ctx->allocs.erase(it);
    klass = ctx->alloc_class[ptr];
    ctx->alloc_class.erase(ptr);
    keep = ctx->cached_bytes + klass <= ctx->cache_limit;
    if (keep) { ctx->free_blocks[klass].push_back(ptr); ctx->cached_bytes += klass; }
  }
  if (!keep) {
    AA_HIP(hipStreamSynchronize(ctx->stream));
    AA_HIP(hipFree(ptr));
  }
  return AMDARROW_OK;
}

// bytes_allocated / max_memory count LIVE buffers at their 64-byte-rounded capacity (MemoryPool::bytes_allocated,
// memory_pool.h:79-90); what the pool holds on the device beyond that -- size-class rounding of live blocks is not
// included, freed blocks kept for reuse are -- is amdarrow_pool_cached_bytes, and amdarrow_pool_trim returns it.
int64_t amdarrow_pool_cached_bytes(amdarrow_ctx* ctx) {
  std::lock_guard<std::mutex> lock(ctx->pool_mu);
  return ctx->cached_bytes;
}
int amdarrow_pool_trim(amdarrow_ctx* ctx, int64_t* out_freed) {
  if (!ctx) return SetError(AMDARROW_INVALID, "null context");
  AA_HIP(hipSetDevice(ctx->device));
  const int64_t freed = DrainPool(ctx);
  if (out_freed) *out_freed = freed;
  return AMDARROW_OK;
}
int amdarrow_pool_set_cache_limit(amdarrow_ctx* ctx, int64_t bytes) {
  if (!ctx || bytes < 0) return SetError(AMDARROW_INVALID, "amdarrow_pool_set_cache_limit: bad argument");
  { std::lock_guard<std::mutex> lock(ctx->pool_mu); ctx->cache_limit = bytes; }
  if (amdarrow_pool_cached_bytes(ctx) > bytes) (void)DrainPool(ctx);
  return AMDARROW_OK;
}

int64_t amdarrow_bytes_allocated(amdarrow_ctx* ctx) { return ctx->bytes_allocated; }
int64_t amdarrow_max_memory(amdarrow_ctx* ctx) { return ctx->max_memory; }

int amdarrow_memcpy_h2d(amdarrow_ctx* ctx, void* dst, const void* src, int64_t n) {
  AA_HIP(hipMemcpyAsync(dst, src, static_cast<size_t>(n), hipMemcpyHostToDevice, ctx->stream));
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}
int amdarrow_memcpy_d2h(amdarrow_ctx* ctx, void* dst, const void* src, int64_t n) {
  AA_HIP(hipMemcpyAsync(dst, src, static_cast<size_t>(n), hipMemcpyDeviceToHost, ctx->stream));
  AA_HIP(hipStreamSynchronize(ctx->stream));
  return AMDARROW_OK;
}
int amdarrow_memcpy_d2d(amdarrow_ctx* ctx, void* dst, const void* src, int64_t n) {
  AA_HIP(hipMemcpyAsync(dst, src, static_cast<size_t>(n), hipMemcpyDeviceToDevice, ctx->stream));
  return AMDARROW_OK;
}
int amdarrow_memset(amdarrow_ctx* ctx, void* dst, int value, int64_t n) {
  AA_HIP(hipMemsetAsync(dst, value, static_cast<size_t>(n), ctx->stream));
  return AMDARROW_OK;
}

}  // extern "C"

// runtime.hip -- context, device memory pool, error plumbing for the C ABI.
#include "common.h"

#include <cstdio>
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
    AA_HIP(hipMalloc(&ctx->scratch, cap));
    ctx->scratch_cap = cap;
  }
  *out = ctx->scratch;
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
constexpr int64_t kPoolCacheLimit = int64_t(16) << 30;  // cached (free) bytes beyond which blocks go back to the device

int amdarrow_malloc(amdarrow_ctx* ctx, int64_t nbytes, void** out_ptr) {
  if (nbytes < 0) return SetError(AMDARROW_INVALID, "amdarrow_malloc: negative size");
  // capacity rounded up to 64 bytes like arrow::MemoryPool (memory_pool.cc:101, buffer.cc:196-216)
  const int64_t cap = ((nbytes + 63) / 64) * 64 + (nbytes == 0 ? 64 : 0);
  const int64_t klass = SizeClass(cap);
  AA_HIP(hipSetDevice(ctx->device));
  void* p = nullptr;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    auto it = ctx->free_blocks.find(klass);
    if (it != ctx->free_blocks.end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
      ctx->cached_bytes -= klass;
    }
  }
  if (!p) {
    hipError_t e = hipMalloc(&p, static_cast<size_t>(klass));
    if (e != hipSuccess) {  // out of memory: give the cache back and try once more
      (void)hipGetLastError();
      std::vector<void*> drop;
      {
        std::lock_guard<std::mutex> lock(ctx->mu);
        for (auto& kv : ctx->free_blocks) { for (void* q : kv.second) drop.push_back(q); kv.second.clear(); }
        ctx->cached_bytes = 0;
      }
      AA_HIP(hipStreamSynchronize(ctx->stream));
      for (void* q : drop) (void)hipFree(q);
      AA_HIP(hipMalloc(&p, static_cast<size_t>(klass)));
    }
  }
  if (cap > nbytes) {
    // zero the padding so word-at-a-time bitmap readers see defined bytes
    AA_HIP(hipMemsetAsync(static_cast<char*>(p) + nbytes, 0, static_cast<size_t>(cap - nbytes), ctx->stream));
  }
  std::lock_guard<std::mutex> lock(ctx->mu);
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
    std::lock_guard<std::mutex> lock(ctx->mu);
    auto it = ctx->allocs.find(ptr);
    if (it == ctx->allocs.end()) return SetError(AMDARROW_INVALID, "amdarrow_free: pointer not from this pool");
    ctx->bytes_allocated -= it->second;
    ctx->allocs.erase(it);
    klass = ctx->alloc_class[ptr];
    ctx->alloc_class.erase(ptr);
    keep = ctx->cached_bytes + klass <= kPoolCacheLimit;
    if (keep) { ctx->free_blocks[klass].push_back(ptr); ctx->cached_bytes += klass; }
  }
  if (!keep) {
    AA_HIP(hipStreamSynchronize(ctx->stream));
    AA_HIP(hipFree(ptr));
  }
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

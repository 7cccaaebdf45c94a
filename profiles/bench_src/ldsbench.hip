// ldsbench: LDS throughput of the partition aggregate's fold (random slots in a 4096-slot table).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int kSlots = 4096;
constexpr int kThreads = 1024;

__device__ __forceinline__ uint32_t Fmix32(uint32_t h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }

// MODE bits: 1 = two bucket reads (b128), 2 = add u64, 4 = min u32, 8 = add u32, 16 = add f64 instead of u64
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_lds(int iters, unsigned long long* sink, const uint4* stream) {
  __shared__ __attribute__((aligned(16))) unsigned int s_tag[kSlots];
  __shared__ unsigned long long s_sum[kSlots];
  __shared__ uint32_t s_first[kSlots];
  __shared__ unsigned int s_cnt[kSlots];
  const int tid = threadIdx.x;
  for (int s = tid; s < kSlots; s += kThreads) { s_tag[s] = s; s_sum[s] = 0; s_first[s] = 0xFFFFFFFFu; s_cnt[s] = 0; }
  __syncthreads();
  uint32_t x = tid * 2654435761u + blockIdx.x;
  unsigned long long acc = 0;
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
  // MODE & 32: the keys come from a streamed buffer (16 bytes per record, one step of 4 records per lane prefetched)
  const u32x4* sp = reinterpret_cast<const u32x4*>(stream) + (size_t(blockIdx.x) * iters) * 4 * kThreads + tid;
  u32x4 cur[4], nxt[4];
  if (MODE & 32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) cur[u] = __builtin_nontemporal_load(sp + u * kThreads);
  }
  for (int it = 0; it < iters; ++it) {
    uint32_t h[4];
    if (MODE & 32) {
      const int itn = it + 1 < iters ? it + 1 : it;
#pragma unroll
      for (int u = 0; u < 4; ++u) nxt[u] = __builtin_nontemporal_load(sp + (size_t(itn) * 4 + u) * kThreads);
#pragma unroll
      for (int u = 0; u < 4; ++u) { h[u] = Fmix32(cur[u].x + cur[u].y + cur[u].z + cur[u].w + it); acc += h[u]; }
    } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) { x = x * 1664525u + 1013904223u; h[u] = Fmix32(x); }
    }
    uint4 a[4], b[4];
    if (MODE & 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = *reinterpret_cast<const uint4*>(&s_tag[(h[u] & 1023) * 4]);
        b[u] = *reinterpret_cast<const uint4*>(&s_tag[((h[u] >> 10) & 1023) * 4]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += a[u].x + b[u].y;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t slot = (h[u] >> 20) & (kSlots - 1);
      if (MODE & 4) atomicMin(&s_first[slot], h[u]);
      if (MODE & 16) atomicAdd(reinterpret_cast<double*>(&s_sum[slot]), 1.0);
      else if (MODE & 2) atomicAdd(&s_sum[slot], (unsigned long long)h[u]);
      if (MODE & 8) atomicAdd(&s_cnt[slot], 1u);
    }
    if (MODE & 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) cur[u] = nxt[u];
    }
  }
  __syncthreads();
  if (acc == 0x1234567 || s_sum[tid] == 0x7654321 || s_cnt[tid] == 0xFFFFFFF0u || s_first[tid] == 1234567) *sink = acc;
}

__global__ void k_fill(uint4* p, size_t n) {
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
    const uint32_t a = Fmix32(uint32_t(i) * 2654435761u + 12345u);
    p[i] = make_uint4(a, Fmix32(a), Fmix32(a + 1), Fmix32(a + 2));
  }
}

template <typename F>
static float TimeIt(F&& launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3 * 1000.f;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  unsigned long long* sink; CK(hipMalloc(&sink, 8));
  uint4* stream; CK(hipMalloc(&stream, size_t(16) * 256 * 48 * 4 * kThreads)); hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, stream, size_t(256) * 48 * 4 * kThreads); CK(hipDeviceSynchronize());
  const int iters = 48;  // 1024 threads x 48 x 4 = 196608 records per workgroup = one CU's share of 50 M records
#define RUN(NAME, MODE) do { float us = TimeIt([&] { hipLaunchKernelGGL((k_lds<MODE>), dim3(256), dim3(kThreads), 0, 0, iters, sink, stream); }); \
    printf("%-40s %8.1f us per 196608 records per CU\n", NAME, us); } while (0)
  RUN("nothing (hash only)", 0);
  RUN("2 bucket reads", 1);
  RUN("add u64", 2);
  RUN("min u32", 4);
  RUN("add u32", 8);
  RUN("add u64 + min u32 + add u32", 14);
  RUN("reads + add u64 + min + add u32 (fold)", 15);
  RUN("reads + add f64 + min + add u32", 31);
  RUN("reads + add u64 + add u32 (no min)", 11);
  RUN("stream 0.8 GB only", 32);
  RUN("stream + 2 bucket reads", 33);
  RUN("stream + atomics", 32 + 14);
  RUN("stream + fold", 32 + 15);
  return 0;
}

// membench: what the memory system gives the scatter's access pattern, without any LDS work.
//   hipcc --offload-arch=gfx950 -O3 -o exp/membench exp/membench.hip && ./exp/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u64x2 = __attribute__((ext_vector_type(2))) unsigned long long;

constexpr int kThreads = 1024;
constexpr int kRows = 4096;  // rows per tile (4 per thread)
constexpr int kP = 512;

struct Args {
  const uint32_t* keys; const uint64_t* vals; const uint8_t* mask;
  uint4* regions; int64_t cap; int64_t n; int64_t rows_per_block; int nblocks;
  unsigned long long* sink;
};

// MODE bits: 1 = loads, 2 = stores; CYCLIC: tile t of workgroup w is global tile t * nblocks + w;
// SECT: bytes per partition write (64 or 128); NT: nontemporal loads / stores
template <int MODE, bool CYCLIC, int SECT, bool NTL, bool NTS, int DEPTH>
__global__ __launch_bounds__(kThreads) void k_mem(Args a) {
  const int tid = threadIdx.x;
  const int64_t ntiles_total = a.n / kRows;
  const int64_t tiles_per_block = a.rows_per_block / kRows;
  uint4* my = a.regions + int64_t(blockIdx.x) * kP * a.cap;
  unsigned long long acc = 0;
  u32x4 kq[DEPTH]; u64x2 va[DEPTH], vb[DEPTH]; uint32_t mb[DEPTH];
  auto tile_row0 = [&](int64_t t) -> int64_t {
    int64_t g = CYCLIC ? t * a.nblocks + blockIdx.x : int64_t(blockIdx.x) * tiles_per_block + t;
    if (g >= ntiles_total) g = ntiles_total - 1;
    return g * kRows;
  };
  auto load = [&](int64_t t, int s) {
    const int64_t r0 = tile_row0(t) + int64_t(tid) * 4;
    if (NTL) {
      kq[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(a.keys + r0));
      va[s] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(a.vals + r0));
      vb[s] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(a.vals + r0 + 2));
    } else {
      kq[s] = *reinterpret_cast<const u32x4*>(a.keys + r0);
      va[s] = *reinterpret_cast<const u64x2*>(a.vals + r0);
      vb[s] = *reinterpret_cast<const u64x2*>(a.vals + r0 + 2);
    }
    mb[s] = a.mask[r0 >> 3];
  };
  if (MODE & 1) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load(s, s);
  }
  for (int64_t t = 0; t < tiles_per_block; t += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      const int64_t tt = t + s;
      if (MODE & 1) {
        acc += kq[s].x + kq[s].y + kq[s].z + kq[s].w + va[s].x + va[s].y + vb[s].x + vb[s].y + mb[s];
      }
      if (MODE & 2) {
        const u32x4 v = {uint32_t(tid), uint32_t(tt), uint32_t(acc), 4u};
        // L = SECT / 16 lanes per partition; a store instruction round covers 1024 / L partitions; every
        // partition gets SECT bytes each `period` tiles so that the bytes per tile stay 32 KB
        constexpr int L = SECT / 16;
        constexpr int rounds = kP / (kThreads / L);      // rounds to cover all partitions once
        constexpr int period = SECT / 64;                // tiles between two writes of a partition
        if ((tt % period) == 0) {
#pragma unroll
          for (int h = 0; h < rounds; ++h) {
            uint4* d = my + int64_t((tid / L) + (kThreads / L) * h) * a.cap + L * (tt / period) + (tid % L);
            if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(d)); else *reinterpret_cast<u32x4*>(d) = v;
          }
        }
      }
      if (MODE & 1) load(tt + DEPTH, s);
    }
  }
  if (acc == 0x123456789abcull) *a.sink = acc;
}

// consumer: workgroup p streams partition p's segment of every block's region (fill records each)
__global__ __launch_bounds__(kThreads) void k_consume(const uint4* regions, int64_t cap, int nblocks, int fill, unsigned long long* sink) {
  const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned long long acc = 0;
  for (int b = wave; b < nblocks; b += kThreads / 64) {
    const uint4* seg = regions + (int64_t(b) * kP + p) * cap;
    for (int i = lane; i < fill; i += 64) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(seg + i));
      acc += v.x + v.z;
    }
  }
  if (acc == 0x123456789abcull) *sink = acc;
}

template <typename F>
static float TimeIt(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1000.f;
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int nblocks = 256;
  const int64_t tiles_per_block = 96;
  const int64_t rows_per_block = tiles_per_block * kRows;
  const int64_t n = rows_per_block * nblocks;  // 100.7 M rows
  Args a{};
  uint32_t* keys; uint64_t* vals; uint8_t* mask; uint4* regions; unsigned long long* sink;
  a.cap = 1784;  // as the library sizes it for this shape
  CK(hipMalloc(&keys, 4 * n)); CK(hipMalloc(&vals, 8 * n)); CK(hipMalloc(&mask, n / 8 + 64));
  CK(hipMalloc(&regions, size_t(16) * a.cap * kP * nblocks)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(keys, 1, 4 * n)); CK(hipMemset(vals, 2, 8 * n)); CK(hipMemset(mask, 0x55, n / 8 + 64));
  a.keys = keys; a.vals = vals; a.mask = mask; a.regions = regions; a.n = n; a.rows_per_block = rows_per_block;
  a.nblocks = nblocks; a.sink = sink;
  const double rd = 12.125 * n, wr = 16.0 * 2048 * tiles_per_block * nblocks;
  printf("rows %ld, read %.2f GB, write %.2f GB\n", (long)n, rd / 1e9, wr / 1e9);
#define RUN(NAME, MODE, CYC, SECT, NTL, NTS, DEPTH)                                                                  \
  do {                                                                                                                 \
    float us = TimeIt([&] { hipLaunchKernelGGL((k_mem<MODE, CYC, SECT, NTL, NTS, DEPTH>), dim3(nblocks), dim3(kThreads), 0, 0, a); }, 5); \
    const double bytes = ((MODE & 1) ? rd : 0) + ((MODE & 2) ? wr : 0);                                                \
    printf("%-44s %8.1f us  %6.2f TB/s\n", NAME, us, bytes / us / 1e6);                                                \
  } while (0)
  RUN("read  blocked nt depth3", 1, false, 64, true, true, 3);
  RUN("read  cyclic  nt depth3", 1, true, 64, true, true, 3);
  RUN("read  cyclic  plain depth3", 1, true, 64, false, true, 3);
  RUN("read  blocked plain depth3", 1, false, 64, false, true, 3);
  RUN("read  cyclic  nt depth2", 1, true, 64, true, true, 2);
  RUN("read  cyclic  nt depth4", 1, true, 64, true, true, 4);
  RUN("write 64B nt", 2, false, 64, true, true, 3);
  RUN("write 64B plain", 2, false, 64, true, false, 3);
  RUN("write 128B nt", 2, false, 128, true, true, 3);
  RUN("write 128B plain", 2, false, 128, true, false, 3);
  RUN("copy  blocked nt / 64B nt (= MEMONLY)", 3, false, 64, true, true, 3);
  RUN("copy  cyclic nt / 64B nt", 3, true, 64, true, true, 3);
  RUN("copy  cyclic nt / 64B plain", 3, true, 64, true, false, 3);
  RUN("copy  cyclic nt / 128B nt", 3, true, 128, true, true, 3);
  RUN("copy  cyclic nt / 128B plain", 3, true, 128, true, false, 3);
  RUN("copy  cyclic plain / 128B plain", 3, true, 128, false, false, 3);
  RUN("copy  cyclic nt / 128B plain depth4", 3, true, 128, true, false, 4);
  RUN("copy  cyclic nt / 256B plain", 3, true, 256, true, false, 3);
  RUN("copy  cyclic nt / 512B plain", 3, true, 512, true, false, 3);
  RUN("copy  cyclic nt / 256B nt", 3, true, 256, true, true, 3);
  RUN("write 256B plain", 2, false, 256, true, false, 3);
  RUN("write 512B plain", 2, false, 512, true, false, 3);
  // chunked: K launches over 1/K of the rows each, all writing the same (small) regions, each followed by a consumer
  for (int K : {1, 4, 8, 16}) {
    Args c = a;
    c.rows_per_block = rows_per_block / K;
    c.n = n;
    const int fill = int(4 * (tiles_per_block / K));  // records per (workgroup, partition) region: 2048 per tile over 512 partitions
    c.cap = K == 1 ? a.cap : ((fill * 5 / 4 + 64) & ~7);
    for (int with_consumer = 0; with_consumer < 2; ++with_consumer) {
      float us = TimeIt([&] {
        for (int k = 0; k < K; ++k) {
          Args ck = c;
          ck.keys = a.keys + int64_t(k) * (n / K); ck.vals = a.vals + int64_t(k) * (n / K); ck.mask = a.mask + int64_t(k) * (n / K / 8);
          ck.n = n / K;
          hipLaunchKernelGGL((k_mem<3, false, 128, true, false, 3>), dim3(nblocks), dim3(kThreads), 0, 0, ck);
          if (with_consumer) hipLaunchKernelGGL(k_consume, dim3(kP), dim3(kThreads), 0, 0, c.regions, c.cap, nblocks, fill, sink);
        }
      }, 5);
      printf("chunks %2d (region span %.0f MB) %s: %8.1f us\n", K, 16.0 * c.cap * kP * nblocks / 1e6, with_consumer ? "scatter+consume" : "scatter only   ", us);
    }
  }
  return 0;
}

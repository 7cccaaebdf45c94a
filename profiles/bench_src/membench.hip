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
template <int MODE, bool CYCLIC, int SECT, bool NTL, bool NTS, int DEPTH, int PMUL = 1>
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
        constexpr int period = SECT / 64 * PMUL;         // tiles between two writes of a partition (PMUL = 2: half the bytes, the 8-byte records)
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

// the filter kernel's geometry: 256-thread workgroups, grid-stride over 4096-row tiles (8 x 16-byte value loads
// per thread and tile, next tile prefetched), a barrier pair per tile; optional contiguous 8-byte stores of half the rows
template <bool STORE, bool BARRIER>
__global__ __launch_bounds__(256) void k_tiles(const uint64_t* vals, uint64_t* out, int64_t ntiles, unsigned long long* sink) {
  const int tid = threadIdx.x;
  u64x2 cur[8], nxt[8];
  unsigned long long acc = 0;
  int64_t tile = blockIdx.x;
  auto issue = [&](int64_t t, u64x2 (&r)[8]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint64_t* src = vals + t * 4096 + (g * 256 + tid) * 4;
      r[2 * g] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(src));
      r[2 * g + 1] = __builtin_nontemporal_load(reinterpret_cast<const u64x2*>(src + 2));
    }
  };
  if (tile < ntiles) issue(tile, cur);
  for (; tile < ntiles; tile += gridDim.x) {
    if (tile + gridDim.x < ntiles) issue(tile + gridDim.x, nxt);
    if (BARRIER) __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += cur[i].x + cur[i].y;
    if (STORE) {
      for (int i = tid; i < 2048; i += 256) out[tile * 2048 + i] = acc + i;
    }
    if (BARRIER) __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
  }
  if (acc == 0x123456789abcull) *sink = acc;
}

// the filter compaction, rebuilt feature by feature on top of k_tiles (FEAT bits):
//   1 = the tile's 64 filter words loaded by the first wave, popcount + wave scan -> LDS, barrier (stores then write `count` values)
//   2 = row-per-lane ranks: every lane reads its word / base from LDS, ranks with mbcnt, stores its value at out[base + rank]
//   4 = output validity: LDS byte flag per rank, ballot-packed and stored after a barrier
template <int FEAT>
__global__ __launch_bounds__(256) void k_filt(const uint64_t* vals, const uint64_t* fwords, uint64_t* out, uint64_t* out_valid,
                                              int64_t ntiles, unsigned long long* sink, const int64_t* tile_off) {
  __shared__ uint64_t s_sel[64];
  __shared__ uint32_t s_base[64];
  __shared__ uint32_t s_count;
  __shared__ uint8_t s_vflag[4096];
  __shared__ uint64_t s_packed[65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long acc = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * 4096;
    uint32_t count = 2048;
    if (FEAT & 1) {
      if (tid < 64) {
        const uint64_t w = fwords[tile * 64 + tid];
        s_sel[tid] = w;
        const uint32_t pc = __popcll(w);
        uint32_t incl = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        s_base[tid] = incl - pc;
        if (tid == 63) s_count = incl;
      }
      __syncthreads();
      count = s_count;
    }
    const int64_t out_base = tile_off ? tile_off[tile] : tile * 2048;
    uint64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = vals[row0 + (j * 4 + wave) * 64 + lane];
    if (FEAT & 2) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int word = j * 4 + wave;
        const uint64_t sel = s_sel[word];
        if ((sel >> lane) & 1) {
          const uint32_t rank = s_base[word] + __builtin_amdgcn_mbcnt_hi(uint32_t(sel >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(sel), 0));
          out[out_base + rank] = v[j];
          if (FEAT & 4) s_vflag[rank & 4095] = uint8_t(v[j] & 1);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += v[j];
      for (int i = tid; i < int(count) && i < 2048; i += 256) out[out_base + i] = acc + i;
    }
    __syncthreads();
    if (FEAT & 4) {
      const uint32_t nlw = (count + 63) >> 6;
      for (uint32_t lw = wave; lw < nlw; lw += 4) {
        const uint32_t i = lw * 64 + lane;
        const uint64_t b = __ballot(i < count && s_vflag[i & 4095] != 0);
        if (lane == 0) s_packed[lw] = b;
      }
      __syncthreads();
      for (uint32_t i = tid; i < nlw && i < 32; i += 256) out_valid[tile * 32 + i] = s_packed[i];
      __syncthreads();
    }
  }
  if (acc == 0x123456789abcull) *sink = acc;
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
  {
    uint64_t* fout; CK(hipMalloc(&fout, 8 * (n / 2 + 4096)));
    const int64_t ntiles = n / 4096;
    for (int grid : {512, 1024, 2048, 4096, 24576}) {
      float t0 = TimeIt([&] { hipLaunchKernelGGL((k_tiles<false, false>), dim3(grid), dim3(256), 0, 0, vals, fout, ntiles, sink); }, 5);
      float t1 = TimeIt([&] { hipLaunchKernelGGL((k_tiles<false, true>), dim3(grid), dim3(256), 0, 0, vals, fout, ntiles, sink); }, 5);
      float t2 = TimeIt([&] { hipLaunchKernelGGL((k_tiles<true, true>), dim3(grid), dim3(256), 0, 0, vals, fout, ntiles, sink); }, 5);
      printf("tiles grid %5d: read %.1f us (%.2f TB/s), + barriers %.1f us, + stores of half the rows %.1f us (%.2f TB/s)\n", grid, t0,
             8.0 * n / t0 / 1e6, t1, t2, 12.0 * n / t2 / 1e6);
    }
  }
  {
    uint64_t* fout; CK(hipMalloc(&fout, 8 * (n / 2 + (1 << 20))));
    uint64_t* fvalid; CK(hipMalloc(&fvalid, n / 8 + 4096));
    uint64_t* fwords; CK(hipMalloc(&fwords, n / 8 + 4096));
    {  // every 64-bit word with exactly 32 set bits (so every tile selects exactly 2048 rows), in a few patterns
      std::vector<uint64_t> h(n / 64);
      const uint64_t pats[4] = {0x5555555555555555ull, 0x33CC33CC33CC33CCull, 0x0F0FF0F00F0FF0F0ull, 0xA5A55A5AC3C33C3Cull};
      for (size_t i = 0; i < h.size(); ++i) h[i] = pats[(i * 2654435761u >> 7) & 3];
      CK(hipMemcpy(fwords, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    }
    const int64_t ntiles = n / 4096;
    const int64_t* toff = nullptr;
#define RUNF(FEAT) do { float t = TimeIt([&] { hipLaunchKernelGGL((k_filt<FEAT>), dim3(2048), dim3(256), 0, 0, vals, fwords, fout, fvalid, ntiles, sink, toff); }, 5); \
      printf("filter features %d%s: %.1f us (%.2f TB/s)\n", FEAT, toff ? " random mask" : "", t, 12.1 * n / t / 1e6); } while (0)
    RUNF(0); RUNF(1); RUNF(3); RUNF(7);
    {  // Bernoulli(0.5) mask words, tile offsets from their popcounts
      std::vector<uint64_t> h(n / 64);
      uint64_t x = 88172645463325252ull;
      for (auto& w : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; w = x; }
      std::vector<int64_t> off(ntiles + 1);
      int64_t run = 0;
      for (int64_t t = 0; t < ntiles; ++t) { off[t] = run; for (int k = 0; k < 64; ++k) run += __builtin_popcountll(h[t * 64 + k]); }
      off[ntiles] = run;
      CK(hipMemcpy(fwords, h.data(), h.size() * 8, hipMemcpyHostToDevice));
      int64_t* d_off; CK(hipMalloc(&d_off, 8 * (ntiles + 1)));
      CK(hipMemcpy(d_off, off.data(), 8 * (ntiles + 1), hipMemcpyHostToDevice));
      toff = d_off;
      RUNF(3); RUNF(7);
    }
  }
  RUN("read  blocked nt depth3", 1, false, 64, true, true, 3);
#define RUNH(NAME, CYC, NTS, PMUL)                                                                                     \
  do {                                                                                                                 \
    float us = TimeIt([&] { hipLaunchKernelGGL((k_mem<3, CYC, 128, true, NTS, 3, PMUL>), dim3(nblocks), dim3(kThreads), 0, 0, a); }, 5); \
    printf("%-44s %8.1f us  %6.2f TB/s\n", NAME, us, (rd + wr / PMUL) / us / 1e6);                                   \
  } while (0)
  // the compact records: the same input next to half (8-byte records) / a quarter of the region bytes
  RUNH("copy  blocked nt / 128B nt, write x 1", false, true, 1);
  RUNH("copy  blocked nt / 128B nt, write x 1/2", false, true, 2);
  RUNH("copy  blocked nt / 128B plain, write x 1/2", false, false, 2);
  RUNH("copy  cyclic nt / 128B nt, write x 1/2", true, true, 2);
  RUNH("copy  blocked nt / 128B nt, write x 1/4", false, true, 4);
  if (getenv("MEMBENCH_SHORT")) return 0;
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

// cu_stream_probe.hip -- how many bytes per second ONE compute unit of gfx950 pulls from HBM, by load path (round 5 probe behind the
// 33..80-row decode projections).   hipcc --offload-arch=gfx950 -O3 -o /tmp/cu_stream_probe tools/cu_stream_probe.hip && /tmp/cu_stream_probe
//   vgpr   : global_load_dwordx4 into registers, U independent loads per wave and iteration (U KB in flight per wave)
//   dma    : global_load_lds_dwordx4 into an LDS ring of U slots per wave (no LDS reads)
// grid = G workgroups of W waves; every wave streams its own contiguous slice of a 2 GB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int U>
__global__ __launch_bounds__(1024) void vgpr_kernel(const uint4* __restrict__ src, uint32_t* out, size_t per_wave_vec, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const uint4* p = src + ((size_t)blockIdx.x * nw + wave) * per_wave_vec + lane;
  uint4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 t = __builtin_nontemporal_load((const u32x4*)(p + (size_t)(it * U + u) * 64));
      v[u] = uint4{t.x, t.y, t.z, t.w};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int U>
__global__ __launch_bounds__(1024) void dma_kernel(const uint4* __restrict__ src, uint32_t* out, size_t per_wave_vec, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* p = (const char*)(src + ((size_t)blockIdx.x * nw + wave) * per_wave_vec + lane);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * U * 1024u;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(lds0 + u * 1024u), "v"(p + (size_t)(it * U + u) * 1024) : "memory", "scc");
    }
    // keep U..2U loads in flight: wait for the older half only
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x * 16] == 0x7f && lane == 77) out[0] = 1;
}

// rows: the access pattern of a row-major [N, K] weight: a wave owns 16 rows (one MFMA tile) and walks K in chunks of C bytes per row;
// one instruction covers 1024 / C rows x C bytes (C = 128 is the decode GEMMs' 64-element chunk).  ROWB = bytes per row.
template <int C, int U, int SWZ = 0>
__global__ __launch_bounds__(1024) void dma_rows_kernel(const char* __restrict__ src, uint32_t* out, int n_tiles, int rowb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int IPC = C / 64, LPR = C / 16, RPI = 1024 / C;      // instructions per chunk, lanes per row, rows per instruction
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * U * 1024u;
  const int n_chunks = rowb / C;
  for (int t = blockIdx.x * nw + wave; t < n_tiles; t += gridDim.x * nw) {
    const int rw = lane / LPR;
    const int key = SWZ ? ((((rw >> 1) & 1) << 2) | ((rw >> 2) & 3)) : 0;      // the XOR swizzle of the decode GEMMs' LDS tiles (C = 128)
    const char* base = src + (size_t)t * 16 * rowb + (size_t)rw * rowb + (((lane % LPR) ^ key) * 16);
    for (int c0 = 0; c0 < n_chunks; c0 += U / IPC) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int chunk = c0 + u / IPC, part = u % IPC;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds0 + u * 1024u), "v"(base + (size_t)part * RPI * rowb + (size_t)chunk * C) : "memory", "scc");
      }
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U / 2) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x * 16] == 0x7f && lane == 77) out[0] = 1;
}

// short: ONE pass like a decode projection -- every wave walks the K = rowb / 2 elements of its TPW tiles once, all workgroups starting
// together; STAG = 0: everybody starts at chunk 0 (all requests in flight share address bits 7..12), 1: workgroup b starts at chunk
// (b * 5) % n_chunks and wraps, 2: per wave.
template <int U, int STAG>
__global__ __launch_bounds__(1024) void dma_short_kernel(const char* __restrict__ src, uint32_t* out, int tpw, int rowb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * U * 1024u;
  const int n_chunks = rowb / 128;
  const int start = STAG == 0 ? 0 : (STAG == 1 ? (blockIdx.x * 5) % n_chunks : ((blockIdx.x * nw + wave) * 5) % n_chunks);
  const int t0 = (blockIdx.x * nw + wave) * tpw;
  const char* base = src + (size_t)t0 * 16 * rowb + (size_t)(lane >> 3) * rowb + (lane & 7) * 16;
  // instruction j of a group: tile (j / 2) % tpw ... keep it simple: walk tile by tile inside a chunk group of U / 2 chunks
  for (int c0 = 0; c0 < n_chunks; c0 += U / 2 / tpw) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int part = u & 1, tile = (u >> 1) % tpw;
      int chunk = c0 + (u >> 1) / tpw + start;
      chunk = chunk >= n_chunks ? chunk - n_chunks : chunk;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(lds0 + u * 1024u), "v"(base + ((size_t)tile * 16 + part * 8) * rowb + (size_t)chunk * 128) : "memory", "scc");
    }
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U / 2) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[threadIdx.x * 16] == 0x7f && lane == 77) out[0] = 1;
}

// vrows: the same row-major walk with the weights going straight into registers in MFMA A-operand layout (lane (l16, q) takes 16 bytes of
// row l16 at byte q * 16 of a 64-byte half chunk: one instruction = 16 rows x 64 bytes), a rolling window of U loads per wave.
template <int U>
__global__ __launch_bounds__(256) void vgpr_rows_kernel(const char* __restrict__ src, uint32_t* out, int n_tiles, int rowb) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, q = lane >> 4;
  const int loads_per_tile = rowb / 64;
  u32x4 acc = {0, 0, 0, 0};
  for (int t = blockIdx.x * nw + wave; t < n_tiles; t += gridDim.x * nw) {
    const char* base = src + (size_t)t * 16 * rowb + (size_t)l16 * rowb + q * 16;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)(base + (size_t)u * 64));
    for (int j0 = U; j0 < loads_per_tile; j0 += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc ^= v[u];
        v[u] = __builtin_nontemporal_load((const u32x4*)(base + (size_t)(j0 + u) * 64));
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  uint4* src; uint32_t* out;
  (void)hipMalloc(&src, bytes); (void)hipMalloc(&out, 64);
  (void)hipMemset(src, 1, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, int G, int W, int U, auto launch) {
    const size_t per_wave = bytes / ((size_t)G * W) / 16;          // uint4 per wave
    const int iters = (int)(per_wave / ((size_t)U * 64));
    const size_t moved = (size_t)G * W * iters * U * 1024;
    launch(G, W, per_wave, iters);                                   // warm
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(G, W, per_wave, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-5s G=%4d W=%2d U=%2d : %7.1f GB/s total, %6.1f GB/s per workgroup  (%s)\n", name, G, W, U, moved / ms / 1e6, moved / ms / 1e6 / G,
           hipGetLastError() == hipSuccess ? "ok" : "ERR");
  };
  auto run_rows = [&](const char* name, int G, int W, int rowb, auto launch) {
    const int n_tiles = (int)(bytes / ((size_t)16 * rowb));
    launch(G, W, n_tiles, rowb);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    launch(G, W, n_tiles, rowb);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-8s G=%4d W=%2d row=%6d B : %7.1f GB/s total, %6.1f GB/s per workgroup  (%s)\n", name, G, W, rowb, bytes / ms / 1e6, bytes / ms / 1e6 / G,
           hipGetLastError() == hipSuccess ? "ok" : "ERR");
  };
#define ROWS(CC) run_rows("rows" #CC, G, W, rowb, [&](int g, int w, int nt, int rb) { \
        (void)hipFuncSetAttribute((const void*)dma_rows_kernel<CC, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((dma_rows_kernel<CC, 16>), dim3(g), dim3(w * 64), (size_t)w * 16 * 1024, 0, (const char*)src, out, nt, rb); })
  if (getenv("MALL_ONLY")) {
    // a matrix that was read a moment ago (by another kernel, from other XCDs): what the 256 MB memory-side cache gives back.
    // cold = 8 launches over 8 different regions of the 2 GB buffer, warm = 8 launches over the SAME region (the launch before it warmed it)
    const int G = 256, W = 4;
    (void)hipFuncSetAttribute((const void*)dma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (size_t mb : {16, 32, 64, 100, 128, 180, 224, 256}) {
      const size_t region = mb << 20;
      const size_t per_wave = region / ((size_t)G * W) / 16;
      const int iters = (int)(per_wave / (16 * 64));
      const size_t moved = (size_t)G * W * iters * 16 * 1024;
      for (int warm = 0; warm < 2; ++warm) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          hipLaunchKernelGGL(dma_kernel<16>, dim3(G), dim3(W * 64), (size_t)W * 16 * 1024, 0, src, out, per_wave, iters);
          (void)hipDeviceSynchronize();
          (void)hipEventRecord(e0);
          for (int r = 0; r < 8; ++r)
            hipLaunchKernelGGL(dma_kernel<16>, dim3(G), dim3(W * 64), (size_t)W * 16 * 1024, 0, src + (warm ? 0 : (size_t)(r % 8) * (region / 16)), out, per_wave, iters);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
          float ms; (void)hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        printf("%-5s region %4zu MB: %7.1f us per pass = %7.1f GB/s\n", warm ? "warm" : "cold", mb, best * 1e3 / 8, moved * 8 / best / 1e6);
      }
    }
    return 0;
  }
  if (getenv("SHORT_ONLY")) {
    const int W = 4;
    for (int rowb : {8192, 22016}) for (int tpw : {1, 2, 4}) for (int G : {48, 96, 192, 256, 512}) {
      const size_t mat = (size_t)G * W * tpw * 16 * rowb;          // bytes of one "projection"
      const int reps = (int)(bytes / mat) < 16 ? (int)(bytes / mat) : 16;
      auto go = [&](const char* name, auto kern) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int pass = 0; pass < 2; ++pass) {
          (void)hipDeviceSynchronize();
          (void)hipEventRecord(e0);
          for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(G), dim3(W * 64), 130 * 1024, 0, (const char*)src + r * mat, out, tpw, rowb);
          (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-9s row=%5d tiles/wave=%d G=%4d : %6.1f MB in %6.1f us = %7.1f GB/s, %5.1f GB/s per workgroup\n", name, rowb, tpw, G, mat / 1e6, ms * 1e3 / reps,
               mat * reps / ms / 1e6, mat * reps / ms / 1e6 / G);
      };
      go("lockstep", dma_short_kernel<16, 0>); go("stag-wg", dma_short_kernel<16, 1>); go("stag-wave", dma_short_kernel<16, 2>);
    }
    return 0;
  }
  if (getenv("VROWS_ONLY")) {
    for (int G : {48, 96, 128, 192, 256, 512}) { const int rowb = 8192, W = 4;
#define VR(UU) run_rows("vrows" #UU, G, W, rowb, [&](int g, int w, int nt, int rb) { \
        hipLaunchKernelGGL((vgpr_rows_kernel<UU>), dim3(g), dim3(w * 64), 130 * 1024, 0, (const char*)src, out, nt, rb); })
      (void)hipFuncSetAttribute((const void*)vgpr_rows_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)vgpr_rows_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)vgpr_rows_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)vgpr_rows_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      VR(8); VR(16); VR(32); VR(64);
    }
    return 0;
  }
  if (getenv("SWZ_ONLY")) {
    const size_t lds = 130 * 1024;          // one workgroup per CU for sure
    for (int G : {48, 64, 96, 128, 192, 256}) for (int W : {4}) { const int rowb = 8192;
      run_rows("plain", G, W, rowb, [&](int g, int w, int nt, int rb) {
        (void)hipFuncSetAttribute((const void*)dma_rows_kernel<128, 16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((dma_rows_kernel<128, 16, 0>), dim3(g), dim3(w * 64), lds, 0, (const char*)src, out, nt, rb); });
      run_rows("swizzle", G, W, rowb, [&](int g, int w, int nt, int rb) {
        (void)hipFuncSetAttribute((const void*)dma_rows_kernel<128, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL((dma_rows_kernel<128, 16, 1>), dim3(g), dim3(w * 64), lds, 0, (const char*)src, out, nt, rb); });
    }
    return 0;
  }
  for (int rowb : {8192, 22016}) for (int G : {64, 128, 192, 256, 512}) for (int W : {4, 8}) {
    ROWS(128); ROWS(256); ROWS(512); ROWS(1024);
  }
  if (getenv("ROWS_ONLY")) return 0;
  for (int G : {32, 128, 256, 512, 1024}) {
    for (int W : {4, 8, 16}) {
      run("vgpr", G, W, 8, [&](int g, int w, size_t pw, int it) { hipLaunchKernelGGL(vgpr_kernel<8>, dim3(g), dim3(w * 64), 0, 0, src, out, pw, it); });
      run("vgpr", G, W, 16, [&](int g, int w, size_t pw, int it) { hipLaunchKernelGGL(vgpr_kernel<16>, dim3(g), dim3(w * 64), 0, 0, src, out, pw, it); });
      run("dma", G, W, 8, [&](int g, int w, size_t pw, int it) {
        (void)hipFuncSetAttribute((const void*)dma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dma_kernel<8>, dim3(g), dim3(w * 64), (size_t)w * 8 * 1024, 0, src, out, pw, it); });
      if (W <= 8) run("dma", G, W, 16, [&](int g, int w, size_t pw, int it) {
        (void)hipFuncSetAttribute((const void*)dma_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(dma_kernel<16>, dim3(g), dim3(w * 64), (size_t)w * 16 * 1024, 0, src, out, pw, it); });
    }
  }
  return 0;
}

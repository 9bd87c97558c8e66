// Streaming x [M][K] fp32 through LDS per CU: LDS-DMA (buffer_load_dwordx4 ... lds) against register loads + ds_write_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int TILE_BYTES = 32768;   // per tile
template <int MODE, int RING>
__global__ void __launch_bounds__(1024) k(const float* __restrict__ x, float* __restrict__ out, int n_tiles, int threads_used) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nthreads = blockDim.x;
  const int R = TILE_BYTES / 16 / nthreads;   // pieces per thread per tile
  const u32x4 rx = {(unsigned)(uintptr_t)x, (unsigned)((uintptr_t)x >> 32) & 0xffffu, 0xFFFFFFF0u, 0x00020000u};
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  float acc = 0.f;
  int tile = blockIdx.x;
  auto dma = [&](int t, int slot) {
    for (int i = 0; i < R; ++i) {
      const unsigned voff = (unsigned)(((int64_t)t * TILE_BYTES) + (tid + nthreads * i) * 16);
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + slot * TILE_BYTES + (wave * 64 + nthreads * i) * 16);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(t < n_tiles ? voff : 0xFFFFFFF0u), "s"(dst), "s"(rx) : "memory");
    }
  };
  if (MODE == 0) {
    for (int r = 0; r < RING; ++r) dma(tile + r * gridDim.x, r);
    int slot = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
      // wait for the oldest tile: (RING-1)*R younger requests may stay in flight
      if (RING == 4 && R == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (RING == 2 && R == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      acc += *reinterpret_cast<float*>(smem + slot * TILE_BYTES + tid * 4);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      dma(tile + RING * gridDim.x, slot);
      slot = slot + 1 == RING ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // register path: R pieces per thread per tile, RING tiles in flight in registers
    f4 v[RING][2];
    auto ld = [&](int t, int r) {
      for (int i = 0; i < 2; ++i) {
        const int64_t off = ((int64_t)t * TILE_BYTES) + (tid + nthreads * i) * 16;
        v[r][i] = t < n_tiles ? *reinterpret_cast<const f4*>(reinterpret_cast<const char*>(x) + off) : f4{0, 0, 0, 0};
      }
    };
#pragma unroll
    for (int r = 0; r < RING; ++r) ld(tile + r * gridDim.x, r);
    for (; tile < n_tiles; tile += RING * gridDim.x) {
#pragma unroll
      for (int r = 0; r < RING; ++r) {
        if (tile + r * (int)gridDim.x < n_tiles) {
          for (int i = 0; i < 2; ++i) *reinterpret_cast<f4*>(smem + (r & 1) * TILE_BYTES + (tid + nthreads * i) * 16) = v[r][i];
          ld(tile + (r + RING) * gridDim.x, r);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          acc += *reinterpret_cast<float*>(smem + (r & 1) * TILE_BYTES + tid * 4);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      }
    }
  }
  if (acc == 123.456f) out[0] = acc;
}
int main() {
  const int64_t bytes = (int64_t)1 << 30;
  float* x; float* out;
  hipMalloc(&x, bytes); hipMalloc(&out, 64);
  hipMemset(x, 0, bytes);
  const int n_tiles = bytes / TILE_BYTES;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char* name, int lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int it = 0; it < 3; ++it) kern<<<256, 1024, lds>>>(x, out, n_tiles, 1024);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) kern<<<256, 1024, lds>>>(x, out, n_tiles, 1024);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.1f us per GiB  %.2f TB/s  (%s)\n", name, ms / 10 * 1e3, bytes / (ms / 10 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
  };
  run(k<0, 2>, "LDS-DMA ring 2", 2 * TILE_BYTES);
  run(k<0, 4>, "LDS-DMA ring 4", 4 * TILE_BYTES);
  run(k<1, 2>, "registers, 2 tiles in flight", 2 * TILE_BYTES);
  run(k<1, 4>, "registers, 4 tiles in flight", 2 * TILE_BYTES);
  return 0;
}

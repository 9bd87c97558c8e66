// Cost of LDS float atomics (ds_add_f32) per wave instruction on gfx950, against ds_add_u32 and plain ds_write_b32, for the address
// patterns of the tile-local backward (egc_fused_tile.hip, MODE 1): 16 lanes x 4 components of a 256-byte row, four rows per wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, int iters, int pattern) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  // pattern 0: lane -> consecutive floats (conflict free); 1: 16 lanes x 16 B apart, 4 row groups 256 B apart (4-way bank conflict);
  // 2: all lanes of a 16-lane group to the same row as in (1) but rows differ per wave
  int idx;
  if (pattern == 0) idx = (wave * 64 + lane) % 16384;
  else idx = (((wave * 4 + (lane >> 4)) * 64) + (lane & 15) * 4) % 16384;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) atomicAdd(&lds[idx], 1.0f);
    else if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(lds) + idx, 1u);
    else if (MODE == 3) { atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (idx >> 1), 1ull); if (pattern != 0) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (idx >> 1) + 1, 1ull); }
    else lds[idx] = (float)i;
    if (pattern != 0) { if (MODE == 0) { atomicAdd(&lds[idx + 1], 1.0f); atomicAdd(&lds[idx + 2], 1.0f); atomicAdd(&lds[idx + 3], 1.0f); }
                        else if (MODE == 1) { atomicAdd(reinterpret_cast<unsigned*>(lds) + idx + 1, 1u); atomicAdd(reinterpret_cast<unsigned*>(lds) + idx + 2, 1u); atomicAdd(reinterpret_cast<unsigned*>(lds) + idx + 3, 1u); }
                        else if (MODE == 3) { }
                        else { lds[idx + 1] = (float)i; lds[idx + 2] = (float)i; lds[idx + 3] = (float)i; } }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 8);
  const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_add_u64"};
  for (int pattern = 0; pattern < 2; ++pattern)
    for (int mode = 0; mode < 4; ++mode)
      for (int threads : {64, 768}) {
        for (int rep = 0; rep < 2; ++rep) {
          if (mode == 0) k<0><<<256, threads, 65536>>>(d, 1000, pattern);
          else if (mode == 1) k<1><<<256, threads, 65536>>>(d, 1000, pattern);
          else if (mode == 3) k<3><<<256, threads, 65536>>>(d, 1000, pattern);
          else k<2><<<256, threads, 65536>>>(d, 1000, pattern);
          hipDeviceSynchronize();
        }
        unsigned long long h[256]; hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        const int per = pattern == 0 ? 1 : (mode == 3 ? 2 : 4);
        printf("pattern %d %-12s %4d threads: %.1f cycles per wave instruction (wave 0's view, %d per iteration)\n", pattern, names[mode], threads, s / 256 / 1000.0 / per, per);
      }
  return 0;
}

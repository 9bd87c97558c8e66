#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int iters) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 256 * 8);
  for (int threads : {64, 256, 512, 768, 1024}) {
    for (int grid : {1, 256}) {
      k<<<grid, threads>>>(d, 1000); hipDeviceSynchronize();
      k<<<grid, threads>>>(d, 1000); hipDeviceSynchronize();
      unsigned long long h[256]; hipMemcpy(h, d, grid * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
      printf("threads %d grid %d: %.1f cycles per barrier\n", threads, grid, s / grid / 1000.0);
    }
  }
  return 0;
}

// In which order does ONE returning LDS atomic instruction (ds_add_rtn_u32) serve the lanes of a wavefront that hit the same
// word?  The deterministic CSR build of the one-launch batch kernels (egc_fused_tile_dev.h, csr_s1) takes an entry's rank
// inside its row from the value such an add returns: the rows come out in input order if -- and only if -- the lanes are
// served in ascending order, and successive instructions of one wavefront in program order.  This program checks both, under
// contention from the other wavefronts of the workgroup (which add to other halves of the same words, as the kernel does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(1024) k(const int* keys, int n_keys, int rounds, int* bad, int* ret_out) {
  __shared__ unsigned cnt[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  // wavefront w adds 1 << (8 * (w & 3)) to word key: its own byte of the word is its running count (rows of < 256 entries here)
  int expect_fail = 0;
  for (int r = 0; r < rounds; ++r) {
    const int key = keys[((blockIdx.x * 16 + wave) * rounds + r) * 64 + lane] % n_keys;
    const unsigned old = __hip_atomic_fetch_add(&cnt[key + 64 * (wave >> 2)], 1u << (8 * (wave & 3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int mine = (old >> (8 * (wave & 3))) & 0xff;
    // what input order demands: the entries of this wavefront with the same key in earlier rounds + in lower lanes of this round
    int want = 0;
    for (int rr = 0; rr <= r; ++rr)
      for (int l = 0; l < (rr == r ? lane : 64); ++l)
        want += (keys[((blockIdx.x * 16 + wave) * rounds + rr) * 64 + l] % n_keys) == key;
    if (mine != want) expect_fail = 1;
    if (blockIdx.x == 0 && wave == 0 && r == 0) ret_out[lane] = mine;
  }
  if (expect_fail) atomicAdd(bad, 1);
}
int main() {
  const int blocks = 256, rounds = 3;
  const int n = blocks * 16 * rounds * 64;
  int* h = (int*)malloc(n * 4);
  int *d, *bad, *ret;
  hipMalloc(&d, n * 4); hipMalloc(&bad, 4); hipMalloc(&ret, 256);
  int total_bad = 0;
  for (int n_keys : {1, 2, 5, 17, 64}) {
    srand(n_keys);
    for (int i = 0; i < n; ++i) h[i] = rand();
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    k<<<blocks, 1024>>>(d, n_keys, rounds, bad, ret);
    int hb = 0, hr[64];
    hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    hipMemcpy(hr, ret, 256, hipMemcpyDeviceToHost);
    printf("keys=%2d: lanes out of input order: %d of %d", n_keys, hb, blocks * 1024);
    if (n_keys == 1) { printf("   returned by lanes 0..7 of one instruction, all on one word:"); for (int i = 0; i < 8; ++i) printf(" %d", hr[i]); }
    printf("\n");
    total_bad += hb;
  }
  printf(total_bad == 0 ? "ORDER OK: ds_add_rtn serves equal addresses in ascending lane order, instructions in program order\n" : "ORDER VIOLATED\n");
  return total_bad != 0;
}

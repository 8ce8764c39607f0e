// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds a [R][64] b16 matrix with value = row*256 + col
// (exact in 16 bits for R<=255).  Every lane supplies its own address; the result shows which (row, col) each
// lane/element received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(unsigned short* out, int variant) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int l = threadIdx.x;
  // hypothesis: within a 16-lane group, lane p supplies the address of row (p>>2), cols 4*(p&3)..+3 of a [4][16] block
  const int grp = l >> 4, p = l & 15;
  int row, col;
  if (variant == 0) { row = (p >> 2) + 4 * grp; col = 4 * (p & 3); }          // group g -> rows 4g..4g+3, cols 0..15
  else { row = p >> 2; col = 4 * (p & 3) + 16 * grp; }                          // group g -> rows 0..3, cols 16g..16g+15
  unsigned addr = (unsigned)(unsigned long long)(const void*)&lds[row * 64 + col];   // low 32 bits of a shared pointer = LDS offset
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int variant = 0; variant < 2; ++variant) {
    hipMemset(d, 0xff, 512);
    probe<<<1, 64>>>(d, variant);
    hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("launch error %s\n", hipGetErrorString(e));
    std::vector<unsigned short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l * 4 + j] >> 8, h[l * 4 + j] & 255);
      printf("\n");
    }
  }
  return 0;
}

// Measurement aid: per-CU LDS-DMA (global_load_lds, 16 B per lane) throughput as a function of the bytes kept in flight,
// for (a) a weight stream SHARED by all blocks (L2-resident after the first block: the fused transformer tail's pattern) and
// (b) a private stream per block (HBM / MALL).  No compute; 256 blocks x 512 threads, one block per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_lds_probe scripts/l2_lds_probe.hip && /tmp/l2_lds_probe > profiles/rNN_l2_lds_probe.txt   (the "shared" rows are the L2 -> LDS ceiling bench.py prices roofline.secondary_bound against)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int L, int S>   // L loads of 16 B per thread per tile (tile = L * 8 KB), S ring stages
__global__ void __launch_bounds__(512) probe(const char* src, size_t block_stride, int ntiles, long long* out) {
  extern __shared__ char smem[];
  constexpr int TILE = L * 512 * 16;
  const int t = threadIdx.x;
  const char* g = src + (size_t)blockIdx.x * block_stride + (size_t)t * 16;
  long long t0 = __builtin_amdgcn_s_memrealtime();
  int issued = 0;
  const bool gather = block_stride == 1;      // pattern of xf_chain.hip: tile = [64 L rows][64 k] of a [320][320] bf16 matrix
  if (gather) g = src;
  auto issue = [&](int slot) {
    if (gather) {
      const int tile = issued, mat = tile / 5, kt = tile - mat * 5;
      const char* b = src + (size_t)mat * 204800 + kt * 128 + (size_t)(t >> 3) * 640 + (((t & 7) ^ ((t >> 4) & 7)) << 4);
#pragma unroll
      for (int i = 0; i < L; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(b + (size_t)i * 64 * 640), (lptr_t)(smem + slot * TILE + i * 8192 + (t / 64) * 1024), 16, 0, 0);
      ++issued; return;
    }
#pragma unroll
    for (int i = 0; i < L; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(g + (size_t)i * 8192), (lptr_t)(smem + slot * TILE + i * 8192 + (t / 64) * 1024), 16, 0, 0);
    g += TILE; ++issued;
  };
#pragma unroll
  for (int s = 0; s < S - 1; ++s) issue(s);
  for (int j = 0; j < ntiles; ++j) {
    if constexpr (S >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * L) : "memory");
    __builtin_amdgcn_s_barrier();
    if (issued < ntiles) issue((j + S - 1) % S);
    else {   // keep the counted wait uniform: dummy re-issue of the last tile
#pragma unroll
      for (int i = 0; i < L; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)((gather ? src : g - TILE) + (size_t)i * 8192), (lptr_t)(smem + ((j + S - 1) % S) * TILE + i * 8192 + (t / 64) * 1024), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  long long t1 = __builtin_amdgcn_s_memrealtime();
  if (t == 0) { out[2 * blockIdx.x] = t0; out[2 * blockIdx.x + 1] = t1 + (long long)(smem[t * 4] == 77); }
}

template <int L, int S>
void run(const char* buf, size_t shared_bytes, long long* dout, const char* tag) {
  constexpr int TILE = L * 8192;
  const int ntiles = (int)(shared_bytes / TILE);
  hipFuncSetAttribute((const void*)probe<L, S>, hipFuncAttributeMaxDynamicSharedMemorySize, S * TILE);
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 2 && L != 5) continue;
    const size_t stride = mode == 2 ? 1 : mode ? shared_bytes : 0;
    float best = 1e9;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      probe<L, S><<<256, 512, S * TILE>>>(buf, stride, ntiles, dout);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<long long> h(512); hipMemcpy(h.data(), dout, 512 * 8, hipMemcpyDeviceToHost);
    double life = 0; for (int b = 0; b < 256; ++b) life += (h[2 * b + 1] - h[2 * b]) * 0.01; life /= 256;
    printf("%s tile %3d KB x %d stages (%3d KB in flight): %s stream: kernel %7.1f us, block mean %7.1f us -> %6.1f GB/s per CU, %5.2f TB/s aggregate\n",
           tag, TILE / 1024, S, (S - 1) * TILE / 1024, mode == 2 ? "gather " : mode ? "private" : "shared ", best * 1e3, life, ntiles * (double)TILE / life * 1e-3, 256.0 * ntiles * TILE / life * 1e-6);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

int main() {
  const size_t shared_bytes = 3u << 20;                    // 3 MB per block (the tail's weight stream is 3.2 MB)
  char* buf; hipMalloc(&buf, 256 * shared_bytes + (1 << 20)); hipMemset(buf, 1, 256 * shared_bytes + (1 << 20));
  long long* dout; hipMalloc(&dout, 512 * 8);
  run<2, 2>(buf, shared_bytes, dout, "a"); run<2, 3>(buf, shared_bytes, dout, "a"); run<2, 4>(buf, shared_bytes, dout, "a");
  run<2, 6>(buf, shared_bytes, dout, "a"); run<2, 8>(buf, shared_bytes, dout, "a");
  run<4, 2>(buf, shared_bytes, dout, "b"); run<4, 3>(buf, shared_bytes, dout, "b"); run<4, 4>(buf, shared_bytes, dout, "b");
  run<6, 2>(buf, shared_bytes, dout, "c"); run<6, 3>(buf, shared_bytes, dout, "c");
  run<5, 2>(buf, shared_bytes, dout, "d"); run<5, 3>(buf, shared_bytes, dout, "d");
  return 0;
}

// Does a memset / D2D-memcpy NODE of a captured hipGraph do its work, in order, when two instantiated graphs that touch the same
// buffer are replayed alternately?  (round-5 root cause of the non-finite latents of BENCH_r04: the per-forward statistics / flag
// pools of the UNet executor were zeroed with hipMemsetAsync inside the captured step.)
//   hipcc --offload-arch=gfx950 -O2 -o graph_memset_probe graph_memset_probe.hip && ./graph_memset_probe
// Every graph is: [pre kernel] -> zero(buf) -> inc(buf) -> copy(buf -> out_g).  After a launch out_g must be all 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
__global__ void k_pre(int* scratch, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) scratch[i] = scratch[i] * 3 + 1; }
__global__ void k_inc(int* buf, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(buf + i, 1); }
__global__ void k_copy(const int* buf, int* out, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = buf[i]; }
__global__ void k_zero(int* buf, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = 0; }

enum Mode { MEMSET = 0, MEMCPY = 1, KERNEL = 2 };
static int build(hipStream_t s, Mode m, int* scratch, int* buf, const int* zeros, int* out, size_t n, hipGraphExec_t* ex) {
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  hipLaunchKernelGGL(k_pre, dim3(256), dim3(256), 0, s, scratch, n);
  if (m == MEMSET) CK(hipMemsetAsync(buf, 0, n * 4, s));
  else if (m == MEMCPY) CK(hipMemcpyAsync(buf, zeros, n * 4, hipMemcpyDeviceToDevice, s));
  else hipLaunchKernelGGL(k_zero, dim3(256), dim3(256), 0, s, buf, n);
  hipLaunchKernelGGL(k_inc, dim3(256), dim3(256), 0, s, buf, n);
  hipLaunchKernelGGL(k_copy, dim3(256), dim3(256), 0, s, buf, out, n);
  hipGraph_t g; CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(ex, g, nullptr, nullptr, 0));
  CK(hipGraphDestroy(g));
  return 0;
}
static long bad(const int* out, size_t n, hipStream_t s, int* first) {
  std::vector<int> h(n);
  hipMemcpyAsync(h.data(), out, n * 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
  long b = 0; *first = 1;
  for (size_t i = 0; i < n; ++i) if (h[i] != 1) { if (!b) *first = h[i]; ++b; }
  return b;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int rcall = 0;
  for (size_t n : {(size_t)65536, (size_t)2 << 20}) {
    int *scratch, *buf, *zeros, *out[2], *small;
    CK(hipMalloc(&small, 4096)); CK(hipMemset(small, 0, 4096));
    CK(hipMalloc(&scratch, n * 4)); CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&zeros, n * 4)); CK(hipMalloc(&out[0], n * 4)); CK(hipMalloc(&out[1], n * 4));
    CK(hipMemset(scratch, 0, n * 4)); CK(hipMemset(zeros, 0, n * 4));
    for (int m = 0; m < 3; ++m) {
      hipGraphExec_t ex[2];
      for (int g = 0; g < 2; ++g) if (build(s, (Mode)m, scratch, buf, zeros, out[g], n, &ex[g])) return 2;
      for (int traffic = 0; traffic < 3; ++traffic)
      for (int pattern = 0; pattern < 3; ++pattern) {          // 0: A A A A, 1: A B A B, 2: A x10 B x10 A x10 ...
        for (int sync = 0; sync < 2; ++sync) {
          CK(hipMemset(buf, 0x7f, n * 4)); CK(hipDeviceSynchronize());
          long nbad = 0; int first_val = 1, first_it = -1;
          const int iters = 40;
          std::vector<int*> outs_seen;
          for (int it = 0; it < iters; ++it) {
            const int g = pattern == 0 ? 0 : pattern == 1 ? (it & 1) : ((it / 10) & 1);
            if (traffic) {          // what the denoise loop does between two replays: a 4-byte D2D copy (the step index) and small eager fills
              CK(hipMemcpyAsync(small + 16, small + (it & 7), 4, hipMemcpyDeviceToDevice, s));
              if (traffic > 1) CK(hipMemsetAsync(small + 64, 0, 1024, s));
            }
            CK(hipGraphLaunch(ex[g], s));
            if (sync || it == iters - 1 || true) {            // always check (the check itself synchronises only in `sync` mode below)
              if (sync) CK(hipStreamSynchronize(s));
              int fv; const long b = bad(out[g], n, s, &fv);   // (copy back on the same stream: ordered after the launch)
              if (b && first_it < 0) { first_it = it; first_val = fv; }
              nbad += b ? 1 : 0;
            }
          }
          printf("n=%zu zero-by=%s traffic=%d pattern=%s host-sync-before-check=%d : %ld / %d launches wrong (first at launch %d, value %d)\n", n,
                 m == 0 ? "memset-node" : m == 1 ? "memcpy-node" : "kernel-node", traffic, pattern == 0 ? "AAAA" : pattern == 1 ? "ABAB" : "A10B10", sync, nbad, iters, first_it, first_val);
          if (nbad) rcall = 1;
        }
      }
      for (int g = 0; g < 2; ++g) CK(hipGraphExecDestroy(ex[g]));
    }
    hipFree(scratch); hipFree(buf); hipFree(zeros); hipFree(out[0]); hipFree(out[1]);
  }
  return rcall;
}

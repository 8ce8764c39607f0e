// Second form of graph_memset_probe.hip, shaped like the denoise loop that failed (BENCH_r04): a graph is captured in the MIDDLE of a pass
// (after one eager step), replayed for the rest of the pass, the next pass captures another graph (other in / out addresses, SAME pool address),
// and a later pass replays an OLD graph from its first step.  Each graph: pre kernel -> memset node (256 KB "flags") -> kernel -> memset node
// (8 MB "statistics") -> inc -> copy.   ./graph_memset_probe2 [host_sync_between_passes=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
__global__ void k_pre(int* scratch, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) scratch[i] = scratch[i] * 3 + 1; }
__global__ void k_inc(int* buf, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(buf + i, 1); }
__global__ void k_fill(int* buf, size_t n, int v) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = v; }
__global__ void k_check(const int* buf, size_t n, int want, unsigned long long* nbad, int* sample) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) if (buf[i] != want) { if (atomicAdd(nbad, 1ull) == 0) *sample = buf[i]; }
}
struct Ctx { hipStream_t s; int *scratch, *flags, *stats; size_t nf, ns; unsigned long long* nbad; int* sample; };
static int walk(const Ctx& c, unsigned long long* nbad_slot) {           // one "UNet step"
  hipLaunchKernelGGL(k_pre, dim3(256), dim3(256), 0, c.s, c.scratch, c.nf);
  CK(hipMemsetAsync(c.flags, 0, c.nf * 4, c.s));
  hipLaunchKernelGGL(k_inc, dim3(256), dim3(256), 0, c.s, c.flags, c.nf);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, c.s, c.stats, c.ns, 0x5a5a5a5a);     // an earlier tensor of the walk lived here: the memset node must come AFTER this kernel node
  CK(hipMemsetAsync(c.stats, 0, c.ns * 4, c.s));
  hipLaunchKernelGGL(k_check, dim3(256), dim3(256), 0, c.s, c.stats, c.ns, 0, nbad_slot, c.sample);      // what the memset node left behind
  hipLaunchKernelGGL(k_inc, dim3(256), dim3(256), 0, c.s, c.stats, c.ns);
  hipLaunchKernelGGL(k_inc, dim3(256), dim3(256), 0, c.s, c.stats, c.ns);
  return 0;
}
int main(int argc, char** argv) {
  const int host_sync = argc > 1 ? atoi(argv[1]) : 1;
  Ctx c; CK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking));
  c.nf = 65536; c.ns = 2 << 20;
  CK(hipMalloc(&c.scratch, c.nf * 4)); CK(hipMalloc(&c.flags, c.nf * 4)); CK(hipMalloc(&c.stats, c.ns * 4)); CK(hipMalloc(&c.sample, 4));
  const int PASSES = 8, STEPS = 50;
  CK(hipMalloc(&c.nbad, PASSES * 8)); CK(hipMemset(c.nbad, 0, PASSES * 8)); CK(hipMemset(c.scratch, 0, c.nf * 4));
  hipGraphExec_t ex[3] = {nullptr, nullptr, nullptr}; int seen[3] = {0, 0, 0};
  int* small; CK(hipMalloc(&small, 4096)); CK(hipMemset(small, 0, 4096));
  for (int p = 0; p < PASSES; ++p) {
    const int key = p == 0 ? 0 : 1 + ((p + 1) & 1);          // keys 0 1 2 1 2 1 2 ... (the address pattern of the bench loop)
    // per-graph distinct bad counter slot is baked at capture: use slot = key so a replay of an old graph reports into its slot; read + reset per pass
    for (int st = 0; st < STEPS; ++st) {
      CK(hipMemcpyAsync(small + 16, small + (st & 7), 4, hipMemcpyDeviceToDevice, c.s));        // the step index copy
      if (ex[key]) { CK(hipGraphLaunch(ex[key], c.s)); continue; }
      if (seen[key]++ == 0) { if (walk(c, c.nbad + key)) return 2; continue; }
      CK(hipStreamBeginCapture(c.s, hipStreamCaptureModeThreadLocal));
      if (walk(c, c.nbad + key)) return 2;
      hipGraph_t g; CK(hipStreamEndCapture(c.s, &g));
      CK(hipGraphInstantiate(&ex[key], g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
      CK(hipGraphLaunch(ex[key], c.s));
    }
    if (host_sync) CK(hipStreamSynchronize(c.s));
    unsigned long long h[3]; int smp = 0;
    CK(hipMemcpyAsync(h, c.nbad, 24, hipMemcpyDeviceToHost, c.s)); CK(hipMemcpyAsync(&smp, c.sample, 4, hipMemcpyDeviceToHost, c.s)); CK(hipStreamSynchronize(c.s));
    CK(hipMemsetAsync(c.nbad, 0, 24, c.s));
    printf("pass %d (graph key %d): words of the statistics pool that were NOT zero right after the memset node, summed over %d steps: %llu (sample value 0x%08x)\n", p, key, STEPS, h[key], smp);
  }
  return 0;
}

// What does ONE launch cost on this box, whatever the kernel does?  Back-to-back launches on one stream, HIP events around 2000 of them:
//   k_null        nothing
//   k_prologue    16 B per thread from a 4 MB buffer into LDS, one barrier (the shape of k_roll7's prologue)
//   k_stores      k_prologue + 10.6 MB of 16 B stores (the bytes of one Empty-8x8 x 65 536 step)
// each as 1024 workgroups x 256 threads x 20 KB LDS (k_roll7's one-step launch at 65 536 envs) and as 1024 x 64.
// Build: hipcc --offload-arch=gfx950 -O3 profiles/tools/launch_floor.hip -o gpurun_out/launch_floor   (profiles/r3_gpu19.sh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__global__ void k_null(const uint4* in, uint4* out, int n) {}
__global__ void k_prologue(const uint4* in, uint4* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 v = in[i];
  ((uint4*)smem)[threadIdx.x] = v;
  __syncthreads();
  if (((uint4*)smem)[(threadIdx.x + 1) % blockDim.x].x == 0xdeadbeefu) out[i] = v;     // never true: keeps the load alive
}
__global__ void k_stores(const uint4* in, uint4* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 v = in[i % 262144];
  ((uint4*)smem)[threadIdx.x] = v;
  __syncthreads();
  const uint4 w = ((uint4*)smem)[(threadIdx.x + 1) % blockDim.x];
  const int total = 663552;                         // 10.6 MB / 16
  for (int c = i; c < total; c += gridDim.x * blockDim.x) out[c] = w;
}
template <class K> float run(K k, int blocks, int threads, size_t lds, const uint4* in, uint4* out, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int r = 0; r < 200; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, in, out, 0);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), lds, 0, in, out, 0);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / reps;
}
int main() {
  uint4 *in, *out;
  CK(hipMalloc(&in, 4u << 20)); CK(hipMalloc(&out, 16u << 20));
  CK(hipMemset(in, 1, 4u << 20)); CK(hipMemset(out, 0, 16u << 20));
  const int reps = 2000;
  for (int threads : {256, 64}) {
    const size_t lds = threads == 256 ? 20480 : 20480;
    printf("1024 x %3d threads, %zu B LDS: null %.2f us | prologue (4 MB in) %.2f us | + 10.6 MB out %.2f us per launch\n", threads, lds,
           run(k_null, 1024, threads, lds, in, out, reps), run(k_prologue, 1024, threads, lds, in, out, reps), run(k_stores, 1024, threads, lds, in, out, reps));
  }
  printf("   1 x  64 threads, 0 B LDS: null %.2f us per launch\n", run(k_null, 1, 64, 0, in, out, reps));
  return 0;
}

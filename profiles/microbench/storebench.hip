// storebench.hip — what separates k_render's store stream (4.6-5.5 TB/s) from rocclr's fill kernel (6.9 TB/s)?
// Stand-alone sweep over the launch-geometry variables of a pure 16 B/lane store loop on an 805 MB buffer
// (= one empty8x8_rgb step).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/storebench.hip -o /tmp/storebench && /tmp/storebench
// Tuning aid only (never linked into the product).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: persistent, workgroup b sweeps pieces b, b + B, ... (fill order)      MODE 1: persistent, contiguous 1/B per workgroup
// MODE 2: short-lived, one piece of `per_wg` chunks per workgroup (grid = total / per_wg)
// MODE 0 / 1 with at most K + 1 stores in flight per wave (s_waitcnt vmcnt(K) after every store): does the fill rate of the
// short-lived geometry come from each of its waves having ONE store in flight, i.e. from a tight chip-wide address window?
template <int ORDER, int K>
__global__ void k_store_throttled(u32x4* out, size_t total) {
  const size_t T = blockDim.x, B = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  u32x4 v; v.x = (unsigned)t; v.y = (unsigned)b; v.z = 0; v.w = 1;
  if (ORDER == 0) for (size_t c = b * T + t; c < total; c += B * T) { out[c] = v; asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory"); }
  else { const size_t per = total / B; for (size_t c = t; c < per; c += T) { out[b * per + c] = v; asm volatile("s_waitcnt vmcnt(%0)" :: "n"(K) : "memory"); } }
}
template <int ORDER, int K>
static float run_throttled(u32x4* buf, size_t total, int blocks, int threads) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_store_throttled<ORDER, K>), dim3(blocks), dim3(threads), 0, 0, buf, total);
  (void)hipEventRecord(a, 0);
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_store_throttled<ORDER, K>), dim3(blocks), dim3(threads), 0, 0, buf, total);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms * 1e3f / reps;
}

template <int MODE, bool NT, int WIDTH>
__global__ void k_store(u32x4* out, size_t total, int per_wg) {
  extern __shared__ unsigned char lds[];        // only to limit occupancy like k_render's atlas does
  const size_t T = blockDim.x, B = gridDim.x, b = blockIdx.x, t = threadIdx.x;
  u32x4 v; v.x = (unsigned)t; v.y = (unsigned)b; v.z = 0; v.w = 1;
  if (t == 0 && total == 1) lds[0] = 1;         // keep the allocation alive
  auto st = [&](size_t c) {
    if (WIDTH == 16) { if (NT) __builtin_nontemporal_store(v, out + c); else out[c] = v; }
    else {
      u32x2 h; h.x = v.x; h.y = v.y;
      u32x2* o2 = (u32x2*)out;
      if (NT) { __builtin_nontemporal_store(h, o2 + 2 * c); __builtin_nontemporal_store(h, o2 + 2 * c + 1); }
      else { o2[2 * c] = h; o2[2 * c + 1] = h; }
    }
  };
  if (MODE == 0) for (size_t c = b * T + t; c < total; c += B * T) st(c);
  if (MODE == 1) { const size_t per = total / B; for (size_t c = t; c < per; c += T) st(b * per + c); }
  if (MODE == 2) { const size_t base = b * (size_t)per_wg; for (size_t c = t; c < (size_t)per_wg && base + c < total; c += T) st(base + c); }
}

template <int MODE, bool NT, int WIDTH>
static float run(u32x4* buf, size_t total, int blocks, int threads, int lds, int per_wg) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_store<MODE, NT, WIDTH>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_store<MODE, NT, WIDTH>), dim3(blocks), dim3(threads), lds, 0, buf, total, per_wg);
  (void)hipEventRecord(a, 0);
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_store<MODE, NT, WIDTH>), dim3(blocks), dim3(threads), lds, 0, buf, total, per_wg);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms * 1e3f / reps;
}

int main() {
  const size_t bytes = 805306368ull, total = bytes / 16;
  u32x4* buf = nullptr;
  if (hipMalloc((void**)&buf, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; i++) (void)hipMemsetAsync(buf, 1, bytes, 0);
    (void)hipEventRecord(a, 0); for (int i = 0; i < 20; i++) (void)hipMemsetAsync(buf, 1, bytes, 0); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b); printf("hipMemsetAsync                                   %7.1f us  %5.2f TB/s\n", ms * 50.f, bytes / (ms * 50.f) / 1e6); }
  struct { const char* name; int mode, blocks, threads, lds, per_wg; } cfg[] = {
    { "fill order persistent  1024 x 256, no LDS", 0, 1024, 256, 0, 0 },
    { "fill order persistent  1024 x 256, 40 KB LDS", 0, 1024, 256, 40 * 1024, 0 },
    { "fill order persistent  2048 x 256, no LDS", 0, 2048, 256, 0, 0 },
    { "fill order persistent  4096 x 256, no LDS", 0, 4096, 256, 0, 0 },
    { "fill order persistent   512 x 1024, no LDS", 0, 512, 1024, 0, 0 },
    { "fill order persistent  2048 x 64,  no LDS", 0, 2048, 64, 0, 0 },
    { "contiguous persistent  1024 x 256, no LDS", 1, 1024, 256, 0, 0 },
    { "contiguous persistent  1024 x 256, 40 KB LDS", 1, 1024, 256, 40 * 1024, 0 },
    { "short-lived 4 KB per wg  (196608 x 256)", 2, 196608, 256, 0, 256 },
    { "short-lived 16 KB per wg  (49152 x 256)", 2, 49152, 256, 0, 1024 },
    { "short-lived 64 KB per wg  (12288 x 256)", 2, 12288, 256, 0, 4096 },
    { "short-lived 192 KB per wg  (4096 x 256), 25 KB LDS", 2, 4096, 256, 25 * 1024, 12288 },
  };
  for (auto& c : cfg) {
    float us[4];
    if (c.mode == 0) { us[0] = run<0, false, 16>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); us[1] = run<0, true, 16>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); us[2] = run<0, false, 8>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); }
    else if (c.mode == 1) { us[0] = run<1, false, 16>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); us[1] = run<1, true, 16>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); us[2] = run<1, false, 8>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); }
    else { us[0] = run<2, false, 16>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); us[1] = run<2, true, 16>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); us[2] = run<2, false, 8>(buf, total, c.blocks, c.threads, c.lds, c.per_wg); }
    printf("%-50s x4 %7.1f us %5.2f TB/s | x4 nt %7.1f us | 2 x x2 %7.1f us\n", c.name, us[0], bytes / us[0] / 1e6, us[1], us[2]);
  }
  {
    const int geos[][2] = { { 1024, 256 }, { 2048, 256 }, { 4096, 256 }, { 2048, 64 }, { 8192, 64 } };
    for (auto& g : geos) {
      const float f0 = run_throttled<0, 0>(buf, total, g[0], g[1]), f1 = run_throttled<0, 1>(buf, total, g[0], g[1]), f3 = run_throttled<0, 3>(buf, total, g[0], g[1]),
                  f7 = run_throttled<0, 7>(buf, total, g[0], g[1]);
      const float c0 = run_throttled<1, 0>(buf, total, g[0], g[1]), c1 = run_throttled<1, 1>(buf, total, g[0], g[1]), c3 = run_throttled<1, 3>(buf, total, g[0], g[1]);
      printf("throttled persistent %5d x %4d: fill order vmcnt(0/1/3/7) %6.1f %6.1f %6.1f %6.1f us (best %4.2f TB/s) | contiguous vmcnt(0/1/3) %6.1f %6.1f %6.1f us\n",
             g[0], g[1], f0, f1, f3, f7, bytes / fminf(fminf(f0, f1), fminf(f3, f7)) / 1e6, c0, c1, c3);
    }
  }
  (void)hipFree(buf);
  return 0;
}

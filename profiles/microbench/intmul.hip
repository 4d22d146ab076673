// Integer-multiply cost on gfx950, one wave per SIMD and four: cycles per instruction of a dependent chain (latency) and of eight independent
// chains (issue rate) for v_mul_lo_u32, v_mul_hi_u32, v_mad_u64_u32, v_mul_u32_u24, v_mad_u32_u24, v_add_u32 -- what a PCG64 step
// (a 128 x 128 -> 128 bit multiply, mg_rng.h) is made of.   hipcc --offload-arch=gfx950 -O3 intmul.hip -o intmul && ./intmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int OP, int CHAINS>
__global__ void k(uint32_t* out, uint64_t* cyc, int iters) {
  uint32_t a[8]; uint64_t w[8];
  for (int c = 0; c < 8; c++) { a[c] = threadIdx.x * 2654435761u + c * 40503u + 1u; w[c] = a[c]; }
  const uint32_t m = 0x9E3779B1u + blockIdx.x;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
        if (OP == 1) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
        if (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[c]) : "v"(a[c]), "v"(m) : "vcc");
        if (OP == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(m));
        if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[c]) : "v"(m));
        if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(m));
        if (OP == 6) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(m));
        if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[c]) : "v"(w[(c + 1) & 7]));
      }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint32_t s = 0; for (int c = 0; c < 8; c++) s += a[c] + (uint32_t)w[c] + (uint32_t)(w[c] >> 32);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int CHAINS>
void run(const char* name, int wps) {
  uint32_t* out; uint64_t* cyc;
  const int blocks = 256;                       // one workgroup per CU (approximately), wps * 4 waves each
  hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256 * wps), 0, 0, out, cyc, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(256 * wps), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; i++) avg += (double)h[i]; avg /= blocks;
  const double n = (double)iters * 16 * CHAINS;
  // the cycle counter ticks at a fixed 100 MHz on this family: report time per instruction instead, and cycles at the shader clock from the wall time
  printf("%-18s chains=%d waves/SIMD=%d : %.2f ns per wave-instruction (wall %.3f ms) = %.1f cycles @2.4 GHz ; per SIMD issue interval %.1f cycles\n", name, CHAINS, wps,
         ms * 1e6 / n, ms, ms * 1e6 / n * 2.4, ms * 1e6 / n * 2.4 / wps);
  hipFree(out); hipFree(cyc);
}
int main() {
#define ALL(OP, NAME) run<OP, 1>(NAME, 1); run<OP, 8>(NAME, 1); run<OP, 8>(NAME, 4);
  ALL(5, "v_add_u32") ALL(0, "v_mul_lo_u32") ALL(1, "v_mul_hi_u32") ALL(2, "v_mad_u64_u32") ALL(3, "v_mul_u32_u24") ALL(4, "v_mad_u32_u24") ALL(6, "v_mul_hi_u32_u24") ALL(7, "v_lshl_add_u64")
  return 0;
}

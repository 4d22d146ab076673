// simd_placement.hip — round 6: WHERE do the wavefronts of a k_roll7-shaped launch run?  k_roll7 gives the waves of a workgroup different roles (wave 0 = the
// dynamics wave with the serial chain, waves 1..3 = encode waves); four workgroups share a CU.  If the dispatcher always puts wave w of a workgroup on SIMD
// w, a CU's four dynamics waves share ONE SIMD while the other three SIMDs hold only encode waves.  This probe launches the same shape (1 024 or more
// workgroups x 256 threads, LDS sized so that four workgroups fit a CU), keeps every wave resident for ~30 us, and records HW_REG_HW_ID / HW_REG_XCC_ID per
// wave: the histogram wave index -> SIMD id, and per CU the number of wave-0s per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/simd_placement.hip -o /tmp/simd_placement && /tmp/simd_placement [workgroups] [lds KB] [threads]
// Tuning aid only (never linked into the product).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

extern __shared__ unsigned char smem[];

__global__ void __launch_bounds__(256) k_probe(unsigned* out, int spin_us) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // gfx9 s_getreg: simm16 = (size - 1) << 11 | offset << 6 | id;  HW_ID = 4, XCC_ID = 20
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  smem[threadIdx.x] = (unsigned char)hw;                 // (keeps the LDS allocation alive)
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long ticks = (unsigned long long)spin_us * 100ull;    // s_memtime: 100 MHz
  while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + ((size_t)blockIdx.x * nw + wave) * 4;
    o[0] = hw; o[1] = xcc; o[2] = (unsigned)t0; o[3] = smem[threadIdx.x ^ 1];
  }
}

int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 1024, lds_kb = argc > 2 ? atoi(argv[2]) : 36, threads = argc > 3 ? atoi(argv[3]) : 256;
  const int nw = threads / 64;
  unsigned* d; hipMalloc(&d, (size_t)wgs * nw * 16);
  hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_probe, dim3(wgs), dim3(threads), lds_kb * 1024, 0, d, 30);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h((size_t)wgs * nw * 4);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave [3:0], simd [5:4], pipe [7:6], cu [11:8], sh [12], se [15:13]
  long hist[4][4] = {};
  std::map<unsigned, std::vector<int>> cu_w0;            // (xcc, se, sh, cu) -> SIMD of every resident wave 0
  std::map<unsigned, int> cu_wgs;
  for (int g = 0; g < wgs; g++) {
    for (int w = 0; w < nw; w++) {
      const unsigned hw = h[((size_t)g * nw + w) * 4], xcc = h[((size_t)g * nw + w) * 4 + 1] & 15u;
      const int simd = (hw >> 4) & 3;
      hist[w][simd]++;
      const unsigned key = (xcc << 16) | (((hw >> 13) & 7u) << 8) | (((hw >> 12) & 1u) << 4) | ((hw >> 8) & 15u);
      if (w == 0) { cu_w0[key].push_back(simd); cu_wgs[key]++; }
    }
  }
  printf("workgroups %d x %d threads, %d KB LDS each; distinct CUs seen %zu\n", wgs, threads, lds_kb, cu_w0.size());
  printf("wave index -> SIMD id histogram (rows: wave in workgroup, columns: SIMD 0..3)\n");
  for (int w = 0; w < nw; w++) printf("  wave %d: %6ld %6ld %6ld %6ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  // per CU: how unevenly are the wave-0s spread over its SIMDs
  long maxhist[9] = {}, wgs_hist[9] = {};
  for (auto& kv : cu_w0) {
    int c[4] = {};
    for (int s : kv.second) c[s]++;
    int mx = 0; for (int s = 0; s < 4; s++) mx = c[s] > mx ? c[s] : mx;
    maxhist[mx > 8 ? 8 : mx]++;
    const int n = (int)kv.second.size(); wgs_hist[n > 8 ? 8 : n]++;
  }
  printf("CUs by resident workgroups (1..8+):"); for (int i = 1; i <= 8; i++) printf(" %ld", wgs_hist[i]); printf("\n");
  printf("CUs by the MAX number of wave-0s on one of their SIMDs (1..8+):"); for (int i = 1; i <= 8; i++) printf(" %ld", maxhist[i]); printf("\n");
  // the first few workgroups in full
  for (int g = 0; g < 12 && g < wgs; g++) {
    printf("  wg %4d:", g);
    for (int w = 0; w < nw; w++) { const unsigned hw = h[((size_t)g * nw + w) * 4], xcc = h[((size_t)g * nw + w) * 4 + 1] & 15u; printf("  [xcc %u se %u cu %2u simd %u slot %2u]", xcc, (hw >> 13) & 7u, (hw >> 8) & 15u, (hw >> 4) & 3u, hw & 15u); }
    printf("\n");
  }
  return 0;
}

// rollstore2.hip — round 6: the write streams of a fused k_roll7 launch in the LOG-SPLIT shape (one dynamics wave that stores the 16-byte scalar record of
// every step, three encode waves that store the observations, step j by encode wave j mod 3: 13 rounds of 12 B per lane = 9 408 contiguous bytes per
// workgroup and step, nontemporal), and what the SCALAR stream costs beside the observation stream.  profiles/r4/rollstore.txt measured (time-split shape,
// plain stores) 6.43 TB/s for the observations alone and 5.09 TB/s with the scalars: 10 % more bytes, 27 % more time.  Variants here move / reshape the
// scalar stores only; the observation stream and the slot-major observation layout (one contiguous [N][147] tensor per step: the API) stay.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rollstore2.hip -o /tmp/rollstore2 && /tmp/rollstore2
// Tuning aid only (never linked into the product).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));

struct Cfg {
  size_t slot_bytes, off_scal;   // slot-major record {obs [N][147] | scalars [N][16]}
  size_t env_major_base;         // scalars as [N][S][16] (mode 4)
  int N, T, S;
  int obs;                       // 0 = no observation stores
  int obs_nt;                    // observation stores nontemporal
  int scal;                      // 0 none, 1 every step (plain), 2 all T at the end of the launch, 3 bursts of four steps, 4 env-major [N][S][16], 5 every step nontemporal,
                                 // 6 every step by the ENCODE wave of the step (behind its observation rounds), 7 as 6 but before them
  int d_dyn, d_enc;              // dependent VALU per step of the dynamics wave / per produced step of an encode wave (x 4 cycles)
};

__global__ void __launch_bounds__(256) k_rs2(unsigned char* out, Cfg C) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = blockIdx.x;
  const int e = wg * 64 + lane;
  unsigned acc = (unsigned)e;
  auto scal_store = [&](int j, bool nt) {
    const int slot = (C.T - 1 - j) % C.S;
    u32x4 v; v.x = acc; v.y = acc >> 3; v.z = (unsigned)j; v.w = 7;
    u32x4* p = C.scal == 4 ? (u32x4*)(out + C.env_major_base + ((size_t)e * C.S + slot) * 16) : (u32x4*)(out + (size_t)slot * C.slot_bytes + C.off_scal + (size_t)e * 16);
    if (nt) __builtin_nontemporal_store(v, p); else *p = v;
  };
  if (wave == 0) {
    for (int j = 0; j < C.T; j++) {
      for (int k = 0; k < C.d_dyn; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(k));
      if (C.scal == 1 || C.scal == 4) scal_store(j, false);
      if (C.scal == 5) scal_store(j, true);
      if (C.scal == 3 && (j & 3) == 3) { scal_store(j - 3, false); scal_store(j - 2, false); scal_store(j - 1, false); scal_store(j, false); }
    }
    if (C.scal == 2) for (int j = 0; j < C.T; j++) scal_store(j, false);
    return;
  }
  for (int j = wave - 1; j < C.T; j += 3) {
    for (int k = 0; k < C.d_enc; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(k));
    const int slot = (C.T - 1 - j) % C.S;
    unsigned char* obase = out + (size_t)slot * C.slot_bytes + (size_t)wg * 9408;
    if (C.scal == 7) scal_store(j, false);
    if (C.obs) {
#pragma unroll
      for (int it = 0; it < 13; it++) {
        const int u = lane + 64 * it;
        u32x3 v; v.x = acc; v.y = (unsigned)j; v.z = (unsigned)u;
        if (it < 12 || u < 784) {
          if (C.obs_nt) __builtin_nontemporal_store(v, (u32x3*)(obase + (size_t)u * 12));      // (sizeof(u32x3) is 16: byte offsets)
          else { struct O12 { unsigned x, y, z; }; O12 w{ v.x, v.y, v.z }; *(O12*)(obase + (size_t)u * 12) = w; }
        }
      }
    }
    if (C.scal == 6) scal_store(j, false);
  }
}

static float run(unsigned char* buf, const Cfg& C) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_rs2, dim3(C.N / 64), dim3(256), 0, 0, buf, C);
  (void)hipEventRecord(a, 0);
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k_rs2, dim3(C.N / 64), dim3(256), 0, 0, buf, C);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 65536, T = 32, S = 32;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  Cfg C{};
  C.N = N; C.T = T; C.S = S;
  C.off_scal = up((size_t)N * 147 + 16);
  C.slot_bytes = up(C.off_scal + (size_t)N * 16 + 256);
  C.env_major_base = C.slot_bytes * S + 4096;
  unsigned char* buf = nullptr;
  const size_t bytes = C.env_major_base + (size_t)N * S * 16 + (1 << 20);
  if (hipMalloc((void**)&buf, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  (void)hipMemset(buf, 0, bytes);
  printf("N = %d envs, T = %d steps per launch: observations %.1f MB + scalars %.1f MB per launch\n", N, T, (double)N * T * 147 / 1e6, (double)N * T * 16 / 1e6);
  struct V { const char* name; int obs, obs_nt, scal; };
  const V vs[] = {
    { "obs nt + scalars every step, plain (the product)", 1, 1, 1 },
    { "obs nt, no scalars", 1, 1, 0 },
    { "obs plain, no scalars", 1, 0, 0 },
    { "obs plain + scalars every step, plain", 1, 0, 1 },
    { "obs nt + scalars every step, nontemporal", 1, 1, 5 },
    { "obs nt + all scalars at the END of the launch", 1, 1, 2 },
    { "obs nt + scalars in bursts of four steps", 1, 1, 3 },
    { "obs nt + scalars env-major [N][S][16]", 1, 1, 4 },
    { "obs nt + scalars by the step's ENCODE wave, behind its rounds", 1, 1, 6 },
    { "obs nt + scalars by the step's ENCODE wave, before its rounds", 1, 1, 7 },
    { "obs plain + scalars every step, nontemporal", 1, 0, 5 },
    { "obs plain + all scalars at the END of the launch", 1, 0, 2 },
    { "obs plain + scalars in bursts of four steps", 1, 0, 3 },
    { "obs plain + scalars by the step's ENCODE wave, behind its rounds", 1, 0, 6 },
    { "obs plain + scalars env-major [N][S][16]", 1, 0, 4 },
    { "scalars alone, every step", 0, 0, 1 },
    { "scalars alone, env-major", 0, 0, 4 },
  };
  const int delays[][2] = { { 0, 0 }, { 60, 120 }, { 100, 200 } };
  for (auto& d : delays) {
    printf("-- dependent VALU per step: dynamics wave %d, encode wave %d per produced step (x 4 cycles)\n", d[0], d[1]);
    for (auto& v : vs) {
      C.obs = v.obs; C.obs_nt = v.obs_nt; C.scal = v.scal; C.d_dyn = d[0]; C.d_enc = d[1];
      const float us = run(buf, C);
      const double b = (double)N * T * ((v.obs ? 147.0 : 0.0) + (v.scal ? 16.0 : 0.0));
      printf("%-66s %7.1f us  %5.2f us/step  %5.2f TB/s\n", v.name, us, us / T, b / us / 1e6);
    }
  }
  (void)hipFree(buf);
  return 0;
}

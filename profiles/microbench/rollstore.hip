// rollstore.hip — the WRITE stream of a fused k_roll7 launch and nothing else: what does the memory system give this order?
// 65 536 envs = 1024 workgroups x 4 waves, T = 32 steps into 32 trajectory slots (slot-major records, 256-byte aligned fields, exactly
// the library's layout); wave w of a workgroup writes steps [split[w], split[w+1]) (the time split), per step 13 rounds of 12 B per lane
// (9 408 contiguous bytes per workgroup and step) + the scalar outputs.  Variants change ONE thing each: scalars off, env-major blocks
// ([workgroup][step][64 envs]: 301 KB contiguous per workgroup and launch), XCD-contiguous workgroup mapping, 16 B stores, a wait after
// every store, one wave per workgroup, and a compute delay between steps (the real kernel issues ~1 000 instructions per step).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 profiles/microbench/rollstore.hip -o /tmp/rollstore && /tmp/rollstore
// Tuning aid only (never linked into the product).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

struct Lay { size_t slot_bytes, off_reward, off_term, off_trunc, off_dir, off_mission, off_action; };
struct Out12 { unsigned x, y, z; };
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { V_BASE = 0, V_NOSCALARS = 1, V_BLOCKS = 2, V_XCD = 3, V_X4 = 4, V_WAIT0 = 5, V_WAIT2 = 6, V_NW1 = 7, V_BLOCKS_XCD = 8, V_NT = 9,
       V_PACKED = 10, V_FAR = 11, V_PACKED_FAR = 12, V_REWARD_ONLY = 13, V_BYTES_ONLY = 14, V_PACKED_FIRST = 15 };
constexpr int NV = 16;

template <int V>
__global__ void __launch_bounds__(256) k_rollstore(unsigned char* out, Lay L, int N, int T, int S, int delay, int s1, int s2, int s3) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;
  int wg = blockIdx.x;
  if (V == V_XCD || V == V_BLOCKS_XCD) wg = (int)(blockIdx.x & 7u) * (nwg >> 3) + (int)(blockIdx.x >> 3);
  const int e = wg * 64 + lane;
  int j0, j1;
  if (V == V_NW1) { if (wave) return; j0 = 0; j1 = T; }
  else { j0 = wave == 0 ? 0 : wave == 1 ? s1 : wave == 2 ? s2 : s3; j1 = wave == 0 ? s1 : wave == 1 ? s2 : wave == 2 ? s3 : T; }
  unsigned acc = (unsigned)e;
  // (the real kernel replays steps 0 .. j0-1 silently before its own: a wave's first store comes after j0 * ~240 instructions)
  for (int k = 0; k < j0 * delay / 4; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(k));
  for (int j = j0; j < j1; j++) {
    const int slot = (T - 1 - j) % S;
    unsigned char* ob = out + (size_t)slot * L.slot_bytes;
    // packed: the 13 scalar bytes of an env as ONE 16-byte record {reward f64, terminated, truncated, direction, action, mission u16, pad} in an
    // array [N][16] behind the observations; far: the scalar arrays live in a separate region (another allocation) instead of behind the slot's obs
    unsigned char* sb = (V == V_FAR || V == V_PACKED_FAR) ? out + (size_t)S * L.slot_bytes + (size_t)N * 9408 / 64 * T + (size_t)slot * ((size_t)N * 16 + 4096) - L.off_reward : ob;
    if (V == V_PACKED || V == V_PACKED_FAR || V == V_PACKED_FIRST) {
      u32x4 v; v.x = acc; v.y = acc >> 3; v.z = (unsigned)j; v.w = 7;
      *(u32x4*)(sb + L.off_reward + (size_t)e * 16) = v;
    } else if (V == V_REWARD_ONLY) {
      *(double*)(ob + L.off_reward + (size_t)e * 8) = (double)acc;
    } else if (V == V_BYTES_ONLY) {
      ob[L.off_term + e] = (unsigned char)acc; ob[L.off_trunc + e] = (unsigned char)(acc >> 8); ob[L.off_dir + e] = (unsigned char)(acc >> 16);
      *(unsigned short*)(ob + L.off_mission + (size_t)e * 2) = (unsigned short)acc; ob[L.off_action + e] = (unsigned char)j;
    } else if (V != V_NOSCALARS) {
      ob = sb;
      *(double*)(ob + L.off_reward + (size_t)e * 8) = (double)acc;
      ob[L.off_term + e] = (unsigned char)acc; ob[L.off_trunc + e] = (unsigned char)(acc >> 8); ob[L.off_dir + e] = (unsigned char)(acc >> 16);
      *(unsigned short*)(ob + L.off_mission + (size_t)e * 2) = (unsigned short)acc; ob[L.off_action + e] = (unsigned char)j;
      ob = out + (size_t)slot * L.slot_bytes;
    }
    unsigned char* obase = (V == V_BLOCKS || V == V_BLOCKS_XCD) ? out + ((size_t)wg * T + (size_t)j) * 9408 : ob + (size_t)wg * 9408;
    if (V == V_X4) {
      for (int c = lane; c < 588; c += 64) { u32x4 v; v.x = acc; v.y = j; v.z = c; v.w = 1; ((u32x4*)obase)[c] = v; }
    } else {
#pragma unroll
      for (int it = 0; it < 13; it++) {
        const int u = lane + 64 * it;
        Out12 v; v.x = acc; v.y = (unsigned)j; v.z = (unsigned)u;
        if (it < 12 || u < 784) {
          if (V == V_NT) { __builtin_nontemporal_store(v.x, (unsigned*)obase + 3 * u); __builtin_nontemporal_store(v.y, (unsigned*)obase + 3 * u + 1); __builtin_nontemporal_store(v.z, (unsigned*)obase + 3 * u + 2); }
          else ((Out12*)obase)[u] = v;
        }
        if (V == V_WAIT0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V == V_WAIT2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
    }
    for (int k = 0; k < delay; k++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(k));
  }
}

template <int V>
static float run(unsigned char* buf, const Lay& L, int N, int T, int S, int delay) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  const int s1 = 10, s2 = 18, s3 = 25;          // the library's split for T = 32, ratio 0.12
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k_rollstore<V>), dim3(N / 64), dim3(256), 0, 0, buf, L, N, T, S, delay, s1, s2, s3);
  (void)hipEventRecord(a, 0);
  const int reps = 20;
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_rollstore<V>), dim3(N / 64), dim3(256), 0, 0, buf, L, N, T, S, delay, s1, s2, s3);
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 65536, T = 32, S = 32;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  Lay L;
  L.off_reward = up((size_t)N * 147 + 16); L.off_term = L.off_reward + up((size_t)N * 8); L.off_trunc = L.off_term + up(N);
  L.off_dir = L.off_trunc + up(N); L.off_mission = L.off_dir + up(N); L.off_action = L.off_mission + up(2 * (size_t)N);
  L.slot_bytes = up(L.off_action + N);
  if (L.slot_bytes < L.off_reward + (size_t)N * 16 + 256) L.slot_bytes = up(L.off_reward + (size_t)N * 16 + 256);
  unsigned char* buf = nullptr;
  const size_t bytes = L.slot_bytes * S + (size_t)N * 9408 / 64 * T + (size_t)S * ((size_t)N * 16 + 4096) + (1 << 20);
  if (hipMalloc((void**)&buf, bytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  (void)hipMemset(buf, 0, bytes);
  const double per_launch = (double)N * T * 160.0, obs_only = (double)N * T * 147.0;
  printf("N = %d envs, T = %d steps per launch: %.1f MB per launch (obs only %.1f MB)\n", N, T, per_launch / 1e6, obs_only / 1e6);
  const char* names[] = { "as k_roll7 (slot-major, 4 waves, time split)", "  - scalars off", "  env-major blocks [wg][step][64 envs]", "  XCD-contiguous workgroup mapping",
                          "  16-byte stores", "  wait vmcnt(0) after every store", "  wait vmcnt(2) after every store", "  one wave per workgroup, steps in order",
                          "  env-major blocks + XCD mapping", "  nontemporal dword stores", "  scalars packed 16 B/env, one store", "  scalar arrays in a far region",
                          "  packed scalars in a far region", "  reward only (no byte arrays)", "  byte arrays only (no reward)", "  (packed scalars again)" };
  const int delays[] = { 0, 100, 200 };
  for (int d : delays) {
    printf("-- %d dependent VALU between a wave's steps (x 4 cycles)\n", d);
    float us[NV];
    us[0] = run<0>(buf, L, N, T, S, d); us[1] = run<1>(buf, L, N, T, S, d); us[2] = run<2>(buf, L, N, T, S, d); us[3] = run<3>(buf, L, N, T, S, d);
    us[4] = run<4>(buf, L, N, T, S, d); us[5] = run<5>(buf, L, N, T, S, d); us[6] = run<6>(buf, L, N, T, S, d); us[7] = run<7>(buf, L, N, T, S, d);
    us[8] = run<8>(buf, L, N, T, S, d); us[9] = run<9>(buf, L, N, T, S, d); us[10] = run<10>(buf, L, N, T, S, d); us[11] = run<11>(buf, L, N, T, S, d);
    us[12] = run<12>(buf, L, N, T, S, d); us[13] = run<13>(buf, L, N, T, S, d); us[14] = run<14>(buf, L, N, T, S, d); us[15] = run<15>(buf, L, N, T, S, d);
    for (int v = 0; v < NV; v++) {
      const double b = v == 1 ? obs_only : v == 13 ? (double)N * T * 155.0 : v == 14 ? (double)N * T * 152.0 : (v == 10 || v == 12 || v == 15) ? (double)N * T * 163.0 : per_launch;
      printf("%-48s %7.1f us  %5.2f us/step  %5.2f TB/s\n", names[v], us[v], us[v] / T, b / us[v] / 1e6);
    }
  }
  (void)hipFree(buf);
  return 0;
}

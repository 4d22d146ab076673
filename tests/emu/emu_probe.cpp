// tests/emu/emu_probe.cpp -- TEST INFRASTRUCTURE: negative controls for the sanitizer builds of the emulated library (build_emu.py --sanitize=...).
#include <hip/hip_runtime.h>

namespace mg { extern uint8_t smem[160 * 1024]; }

// (tests/test_emu_sanitizers_cpu.py) Tiny kernels that each do ONE thing on purpose, so that "no report" from the real kernels means something.
//   expected to be REPORTED: 1 a global load one byte past a device buffer; 2 an LDS store one byte past the launch's dynamic LDS size;
//   3 a misaligned 4-byte load and a shift by the operand's width; 4 two waves store to one LDS word with nothing between them; 6 two workgroups
//   store to one global word; 7 lane 1 reads the LDS word lane 0 of its wave wrote with no wave-order marker between them (a lockstep assumption);
//   8 a step counter published before the data it announces, read by a polling wave
//   expected to be CLEAN: 5 the hand-offs the kernels use -- __syncthreads between a write and another wave's read, a SyncWord counter published
//   behind the data and polled by another wave, a wave barrier between two lanes of a wave, atomics from two workgroups
struct ProbeArgs { int what; uint8_t* buf; int lds; };
static void probe_body(void* c) {
  const ProbeArgs& a = *(const ProbeArgs*)c;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  volatile uint32_t sink = 0;
  uint32_t* lds = (uint32_t*)mg::smem;
  switch (a.what) {
    case 1: if (tid == 0) sink = a.buf[100]; break;
    case 2: if (tid == 0) mg::smem[a.lds] = 1; break;
    case 3: if (tid == 0) { const uint32_t* mis = (const uint32_t*)(a.buf + 1 + (a.lds & 1)); sink = mis[0]; volatile int sh = 32; sink = (uint32_t)(1 << sh); } break;   // (UBSan does not check volatile accesses: a plain load, like the kernels')
    case 4: if (lane == 0) lds[0] = (uint32_t)tid; break;
    case 5: {
      if (tid == 0) lds[0] = 7u;
      __syncthreads();
      if (tid == 64) sink = lds[0];
      ::emu::SyncWord* sync = (::emu::SyncWord*)(mg::smem + 64);
      if (tid == 0) *sync = 0u;
      __syncthreads();
      if (wave == 0) { if (lane == 0) lds[1] = 9u; emu_wave_barrier(); if (lane == 0) *sync = 1u; }
      else { while ((uint32_t)emu_readfirstlane((int)(uint32_t)*sync) < 1u) emu_yield(); sink = lds[1]; }
      if (wave == 0) { if (lane == 0) lds[2] = 3u; emu_wave_barrier(); if (lane == 1) sink = lds[2]; }
      atomicAdd((uint32_t*)a.buf, 1u);
      break;
    }
    case 6: if (tid == 0) ((uint32_t*)a.buf)[1] = blockIdx.x; break;
    case 7: if (tid == 0) lds[3] = 1u; if (tid == 1) sink = lds[3]; break;
    case 8: {                              // a counter published BEFORE the data it announces
      ::emu::SyncWord* sync = (::emu::SyncWord*)(mg::smem + 64);
      if (tid == 0) *sync = 0u;
      __syncthreads();
      if (wave == 0) { if (lane == 0) { *sync = 1u; lds[1] = 9u; } }
      else { while ((uint32_t)emu_readfirstlane((int)(uint32_t)*sync) < 1u) emu_yield(); if (lane == 0) sink = lds[1]; }
      break;
    }
    default: break;
  }
  (void)sink;
}
extern "C" int emu_san_probe(int what) {
  ProbeArgs a; a.what = what; a.lds = 256; a.buf = nullptr;
  if (hipMalloc((void**)&a.buf, 100) != hipSuccess) return -1;
  memset(a.buf, 0, 100);
  emu::launch(probe_body, &a, dim3(what == 6 || what == 5 ? 2 : 1), dim3(what == 4 || what == 5 || what == 8 ? 128 : 64), (size_t)a.lds);
  hipFree(a.buf);
  return 0;
}

// tests/emu/shim/hip/hip_runtime.h — TEST INFRASTRUCTURE (never part of the product): a stand-in for <hip/hip_runtime.h> that lets the product's
// own HIP sources (minigrid_amd/csrc/*.hip, unchanged) be compiled as plain C++ for the HOST SIMT EMULATOR of tests/emu/emu_runtime.cpp:
// a workgroup runs as one fiber per lane on one OS thread, cross-lane operations (__ballot, __shfl, readfirstlane, wave / workgroup barriers)
// are resolved when every runnable lane of the wave has arrived, LDS is one host array, "device memory" is host memory, streams run in
// enqueue order.  What it is for: `pytest -m "not gpu"` can execute the real kernels -- wave-level plumbing included -- against the oracle in
// this GPU-less container (tests/test_emu_cpu.py).  The product never loads it: minigrid_amd/_binding.py opens libminigrid_hip.so unless a
// test sets MINIGRID_AMD_LIB, and the emulated library answers mg_build_info() with "emulator=1" (bench.py and smoke() refuse it).
#pragma once
#define MG_EMU 1
#ifndef __HIPCC__
#define __HIPCC__ 1          // (the sources keep their wave-cooperative classes under it)
#endif
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <algorithm>

// ---- qualifiers ----
#define __host__
#define __device__
#define __global__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...)

// ---- vector types ----
struct dim3 { unsigned x, y, z; constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) ulonglong1 { unsigned long long x; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 v; v.x = x; v.y = y; return v; }

// ---- the lane's coordinates: set by the scheduler whenever a fiber is resumed ----
namespace emu {
extern dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
enum Op { OP_BALLOT, OP_SHFL, OP_SHFL_DOWN, OP_SHFL_UP, OP_SHFL_XOR, OP_FIRST, OP_WAVE_BARRIER, OP_BLOCK_BARRIER, OP_YIELD };
unsigned long long xlane(Op op, unsigned long long v, int arg, const void* site);      // blocks the calling lane until its wave (block) has arrived
void launch(void (*tramp)(void*), void* closure, dim3 grid, dim3 block, size_t lds);
}
#define threadIdx (::emu::g_threadIdx)
#define blockIdx (::emu::g_blockIdx)
#define blockDim (::emu::g_blockDim)
#define gridDim (::emu::g_gridDim)
#define warpSize 64

// ---- cross-lane operations ----
#define EMU_SITE() __builtin_return_address(0)
// A cross-lane operation is CONVERGENT: the device compiler never duplicates or sinks one across control flow (the lanes that reach a call
// site are the lanes that take part).  The host compiler has to be told: without the attributes, jump threading may clone a call into both arms of
// a lane-dependent branch (seen with the ASan build of k_roll7's spare staging loop: lanes 0-15 and 16-63 of a wave at two copies of one
// __shfl), and the emulator -- which groups lanes by call site -- would resolve the two halves separately.
#if defined(__clang__)
#define EMU_XLANE __attribute__((noinline, convergent, noduplicate))
#else
#define EMU_XLANE __attribute__((noinline))
#endif
static inline EMU_XLANE unsigned long long __ballot(int pred) { return ::emu::xlane(::emu::OP_BALLOT, pred ? 1ull : 0ull, 0, EMU_SITE()); }
static inline EMU_XLANE int __shfl(int v, int src, int = 64) { return (int)(unsigned)::emu::xlane(::emu::OP_SHFL, (unsigned)v, src, EMU_SITE()); }
static inline EMU_XLANE int __shfl_down(int v, unsigned d, int = 64) { return (int)(unsigned)::emu::xlane(::emu::OP_SHFL_DOWN, (unsigned)v, (int)d, EMU_SITE()); }
static inline EMU_XLANE int __shfl_up(int v, unsigned d, int = 64) { return (int)(unsigned)::emu::xlane(::emu::OP_SHFL_UP, (unsigned)v, (int)d, EMU_SITE()); }
static inline EMU_XLANE int __shfl_xor(int v, int m, int = 64) { return (int)(unsigned)::emu::xlane(::emu::OP_SHFL_XOR, (unsigned)v, m, EMU_SITE()); }
static inline EMU_XLANE int emu_readfirstlane(int v) { return (int)(unsigned)::emu::xlane(::emu::OP_FIRST, (unsigned)v, 0, EMU_SITE()); }
static inline EMU_XLANE void emu_wave_barrier() { ::emu::xlane(::emu::OP_WAVE_BARRIER, 0, 0, EMU_SITE()); }
static inline EMU_XLANE void __syncthreads() { ::emu::xlane(::emu::OP_BLOCK_BARRIER, 0, 0, EMU_SITE()); }
static inline EMU_XLANE void emu_yield() { ::emu::xlane(::emu::OP_YIELD, 0, 0, EMU_SITE()); }
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_readlane(v, l) __shfl((v), (l))
#define __builtin_amdgcn_s_sleep(n) emu_yield()
#define __builtin_amdgcn_s_setprio(n) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_ds_bpermute(idx, v) __shfl((v), (idx) >> 2)
static inline unsigned long long __builtin_readcyclecounter_emu() { return 0; }

// ---- scalar device functions ----
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __brev(unsigned x) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i); return r; }
static inline unsigned long long __brevll(unsigned long long x) { unsigned long long r = 0; for (int i = 0; i < 64; i++) r |= ((x >> i) & 1ull) << (63 - i); return r; }
static inline int __double2loint(double d) { unsigned long long u; memcpy(&u, &d, 8); return (int)(unsigned)u; }
static inline int __double2hiint(double d) { unsigned long long u; memcpy(&u, &d, 8); return (int)(unsigned)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) { unsigned long long u = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double d; memcpy(&d, &u, 8); return d; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
using std::min;
using std::max;
static inline unsigned min(unsigned a, int b) { return a < (unsigned)b ? a : (unsigned)b; }
static inline unsigned min(int a, unsigned b) { return (unsigned)a < b ? (unsigned)a : b; }
static inline unsigned max(unsigned a, int b) { return a > (unsigned)b ? a : (unsigned)b; }
static inline unsigned max(int a, unsigned b) { return (unsigned)a > b ? (unsigned)a : b; }

// ---- atomics: one OS thread, lanes interleave only at cross-lane operations; real atomic builtins so that the thread-sanitizer build sees them as such ----
template <class T, class U> static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicMax(T* p, U v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while ((T)v > o && !__atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { } return o; }
template <class T, class U> static inline T atomicMin(T* p, U v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while ((T)v < o && !__atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { } return o; }
template <class T, class U> static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicExch(T* p, U v) { return __atomic_exchange_n(p, (T)v, __ATOMIC_RELAXED); }
template <class T, class U> static inline T atomicCAS(T* p, U c, U v) { T o = (T)c; __atomic_compare_exchange_n(p, &o, (T)v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return o; }
// An LDS word that one wave publishes and another polls (k_roll7's step counters, mg_roll.h MG_LDS_VU32).  On the device: a volatile LDS word; the DS
// operations of a wave execute in order, so what the wave wrote before the counter is visible to whoever has seen the counter.  For the host
// compiler and its thread sanitizer that contract is a release store / an acquire load.
namespace emu {
struct SyncWord {
  uint32_t v;
  operator uint32_t() const { return __atomic_load_n(&v, __ATOMIC_ACQUIRE); }
  uint32_t operator=(uint32_t x) { __atomic_store_n(&v, x, __ATOMIC_RELEASE); return x; }
};
}
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- runtime API (tests/emu/emu_runtime.cpp): device memory = host memory, streams run in enqueue order ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; size_t sharedMemPerBlock; int maxSharedMemoryPerMultiProcessor; int clockRate; int major, minor; };
extern "C" {
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
static inline hipError_t hipMemGetInfo(size_t* fr, size_t* tot) { *fr = (size_t)64 << 30; *tot = (size_t)64 << 30; return hipSuccess; }   // (the emulated device: host memory)
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned flags);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int prio);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError(void);
const char* hipGetErrorString(hipError_t e);
hipError_t hipDeviceSynchronize(void);
}
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned flags = 0) { return hipHostMalloc((void**)p, n, flags); }
template <class T> static inline hipError_t hipHostGetDevicePointer(T** d, void* h, unsigned flags) { return hipHostGetDevicePointer((void**)d, h, flags); }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// kernel launch: the whole grid runs before the call returns
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                                   \
  do {                                                                                              \
    auto emu_closure = [=]() { kernel(__VA_ARGS__); };                                              \
    ::emu::launch([](void* c) { (*(decltype(emu_closure)*)c)(); }, (void*)&emu_closure, dim3(grid), dim3(block), (size_t)(lds)); \
  } while (0)

// tests/emu/emu_runtime.cpp — TEST INFRASTRUCTURE (never part of the product): the host SIMT emulator behind tests/emu/shim/hip/hip_runtime.h.
//
// A kernel launch runs its grid one workgroup at a time on the calling OS thread.  Every lane of the workgroup is a fiber (ucontext) that executes
// the kernel's C++ body; a lane runs until it reaches a cross-lane operation (__ballot, __shfl*, readfirstlane / readlane, a wave barrier =
// the sources' LDS hand-off macros, __syncthreads, s_sleep = "let the other waves run") and parks there.  When no lane of a wave can run, the
// parked lanes are resolved GROUP BY GROUP -- a group = the lanes parked at the same call site with the same operation, i.e. the lanes a real
// wavefront would have active there; of several groups (divergent control flow) the one at the lowest code address goes first, which lets lanes
// that are behind in program order catch up with the ones waiting at a reconvergence point.  __syncthreads releases when every live lane of
// the workgroup has arrived.  The waves of a workgroup take turns, so LDS polling protocols between waves (k_roll7's dynamics / encode rings) make
// progress.  No lane can run and nothing can be resolved = a deadlock the real kernel would have too (or a lockstep assumption this emulator
// does not model): reported and aborted, never spun on.
//
// What it does NOT model: timing, memory ordering weaker than program order, lockstep execution BETWEEN cross-lane operations (a lane runs ahead
// of its neighbours until the next one: code that relies on "all lanes have executed the previous statement" without one of the macros would
// misbehave here -- the sources mark every such place), hardware limits (LDS size is checked, registers are not).
#include <hip/hip_runtime.h>

#include <sys/mman.h>
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mg { alignas(256) uint8_t smem[160 * 1024]; }        // the workgroup's LDS (`extern __shared__ uint8_t smem[]` in the kernels)

namespace emu {

dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

enum State { RUNNABLE = 0, PARKED = 1, DONE = 2 };
struct Fiber {
  ucontext_t ctx;
  void* stack = nullptr;
  State state = DONE;
  Op op = OP_YIELD;
  unsigned long long v = 0, result = 0;
  int arg = 0;
  const void* site = nullptr;
};
constexpr size_t STACK_BYTES = 512 * 1024;
static std::vector<Fiber> F;
static int g_cur = -1, g_n = 0;
static ucontext_t g_sched;
static void (*g_tramp)(void*) = nullptr;
static void* g_closure = nullptr;
static bool g_in_kernel = false;

static void fiber_main() {
  g_tramp(g_closure);
  F[g_cur].state = DONE;
  swapcontext(&F[g_cur].ctx, &g_sched);
}

unsigned long long xlane(Op op, unsigned long long v, int arg, const void* site) {
  if (!g_in_kernel) { fprintf(stderr, "emu: cross-lane operation outside a kernel\n"); abort(); }
  Fiber& f = F[g_cur];
  f.op = op; f.v = v; f.arg = arg; f.site = site; f.state = PARKED;
  swapcontext(&f.ctx, &g_sched);
  return f.result;
}

static void resume(int i) {
  g_cur = i;
  g_threadIdx = dim3((unsigned)i, 0, 0);
  swapcontext(&g_sched, &F[i].ctx);
  g_cur = -1;
}

// resolve ONE group of wave w's parked lanes (block barriers excluded); false if there is none
static bool resolve_wave(int w, bool& yielded) {
  const int lo = w * 64, hi = std::min(g_n, lo + 64);
  int lead = -1;
  for (int i = lo; i < hi; i++)
    if (F[i].state == PARKED && F[i].op != OP_BLOCK_BARRIER && (lead < 0 || (uintptr_t)F[i].site < (uintptr_t)F[lead].site)) lead = i;
  if (lead < 0) return false;
  const Op op = F[lead].op;
  const void* site = F[lead].site;
  bool in[64] = { false };
  unsigned long long ballot = 0;
  int first = -1;
  for (int i = lo; i < hi; i++)
    if (F[i].state == PARKED && F[i].op == op && F[i].site == site) {
      in[i - lo] = true;
      if (first < 0) first = i;
      if (op == OP_BALLOT && F[i].v) ballot |= 1ull << (i - lo);
    }
  unsigned long long res[64];
  for (int l = 0; l < hi - lo; l++) {
    if (!in[l]) continue;
    const Fiber& f = F[lo + l];
    int src = l;
    switch (op) {
      case OP_BALLOT: res[l] = ballot; continue;
      case OP_FIRST: res[l] = F[first].v; continue;
      case OP_SHFL: src = f.arg & 63; break;
      case OP_SHFL_DOWN: src = l + f.arg; break;
      case OP_SHFL_UP: src = l - f.arg; break;
      case OP_SHFL_XOR: src = l ^ f.arg; break;
      default: res[l] = 0; continue;
    }
    // (a source lane outside the wave keeps the lane's own value; an inactive source lane reads as 0 -- what ds_bpermute gives)
    if (src < 0 || src >= 64 || lo + src >= hi) res[l] = f.v;
    else res[l] = in[src] ? F[lo + src].v : 0ull;
  }
  for (int l = 0; l < hi - lo; l++)
    if (in[l]) { F[lo + l].result = res[l]; F[lo + l].state = RUNNABLE; }
  yielded = op == OP_YIELD;
  return true;
}

static void run_block() {
  const int nw = (g_n + 63) / 64;
  for (;;) {
    bool progress = false, alive = false;
    for (int w = 0; w < nw; w++) {
      const int lo = w * 64, hi = std::min(g_n, lo + 64);
      // the wave runs until it yields (s_sleep: a polling loop) or nothing in it can move
      for (int rounds = 0;; rounds++) {
        bool ran = false;
        for (int i = lo; i < hi; i++)
          while (F[i].state == RUNNABLE) { resume(i); ran = true; }
        bool yielded = false;
        if (!resolve_wave(w, yielded)) { progress |= ran; break; }
        progress = true;
        if (yielded) break;
      }
    }
    // __syncthreads: every lane that is not finished has arrived
    int parked_bar = 0, live = 0;
    for (int i = 0; i < g_n; i++) {
      if (F[i].state != DONE) { live++; alive = true; }
      if (F[i].state == PARKED && F[i].op == OP_BLOCK_BARRIER) parked_bar++;
    }
    if (!alive) return;
    if (live == parked_bar) {
      for (int i = 0; i < g_n; i++) if (F[i].state == PARKED) { F[i].result = 0; F[i].state = RUNNABLE; }
      progress = true;
    }
    if (!progress) {
      fprintf(stderr, "emu: DEADLOCK in workgroup %u (%d lanes): ", g_blockIdx.x, g_n);
      for (int i = 0; i < g_n; i++)
        if (F[i].state == PARKED) { fprintf(stderr, "lane %d parked at %p op %d; ", i, F[i].site, (int)F[i].op); if (i % 64 > 2) i = (i / 64 + 1) * 64 - 1; }
      fprintf(stderr, "\n");
      abort();
    }
  }
}

void launch(void (*tramp)(void*), void* closure, dim3 grid, dim3 block, size_t lds) {
  if (g_in_kernel) { fprintf(stderr, "emu: nested launch\n"); abort(); }
  if (lds > sizeof(mg::smem)) { fprintf(stderr, "emu: %zu bytes of LDS requested (160 KB per CU)\n", lds); abort(); }
  const int n = (int)(block.x * block.y * block.z);
  if (block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1 || n < 1 || n > 1024) { fprintf(stderr, "emu: 1-D launches of up to 1024 lanes only\n"); abort(); }
  if ((int)F.size() < n) F.resize(n);
  for (int i = 0; i < n; i++)
    if (!F[i].stack) {
      F[i].stack = mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (F[i].stack == MAP_FAILED) { perror("emu: mmap"); abort(); }
    }
  g_tramp = tramp; g_closure = closure; g_n = n;
  g_blockDim = block; g_gridDim = grid;
  g_in_kernel = true;
  for (unsigned b = 0; b < grid.x; b++) {
    g_blockIdx = dim3(b, 0, 0);
    memset(mg::smem, 0xCD, lds);                     // (LDS contents are undefined at workgroup start: not zero)
    for (int i = 0; i < n; i++) {
      getcontext(&F[i].ctx);
      F[i].ctx.uc_stack.ss_sp = F[i].stack; F[i].ctx.uc_stack.ss_size = STACK_BYTES; F[i].ctx.uc_link = nullptr;
      makecontext(&F[i].ctx, fiber_main, 0);
      F[i].state = RUNNABLE;
    }
    run_block();
  }
  g_in_kernel = false;
}

}  // namespace emu

// ---- the runtime API: device memory is host memory, every stream runs its work at enqueue time ----
struct emuStream { int dummy; };
struct emuEvent { int dummy; };
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "host SIMT emulator (tests/emu)"); strcpy(p->gcnArchName, "emu");
  p->multiProcessorCount = 4; p->totalGlobalMem = (size_t)8 << 30; p->sharedMemPerBlock = 64 * 1024; p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); if (*p) memset(*p, 0xA5, n); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emuStream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = new emuStream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "emulator error"; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
}

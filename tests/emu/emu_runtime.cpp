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

// (thread sanitizer: THIS file is compiled without -fsanitize=thread but with -DEMU_TSAN=1 -- the scheduler's own state is shared by every fiber by
// design; the kernels, the library's host code and emu_probe.cpp are instrumented)
// ---- sanitizer builds (build_emu.py --sanitize=...): the fibers announced to the runtime, exact-sized device buffers, the unused LDS poisoned ----
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define EMU_ASAN 1
#endif
#if __has_feature(thread_sanitizer) && !defined(EMU_TSAN)
#define EMU_TSAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__) && !defined(EMU_ASAN)
#define EMU_ASAN 1
#endif
#if defined(__SANITIZE_THREAD__) && !defined(EMU_TSAN)
#define EMU_TSAN 1
#endif
#ifdef EMU_ASAN
extern "C" {
void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
void __asan_poison_memory_region(void const volatile* addr, size_t size);
void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
}
#endif
#ifdef EMU_TSAN
extern "C" {
void* __tsan_get_current_fiber(void);
void* __tsan_create_fiber(unsigned flags);
void __tsan_destroy_fiber(void* fiber);
void __tsan_switch_to_fiber(void* fiber, unsigned flags);
void __tsan_acquire(void* addr);
void __tsan_release(void* addr);
}
// (every switch is "no sync": the scheduler must not become a happens-before relay between the lanes -- the edges that exist on the device are
// drawn explicitly: launch start / end, cross-lane operations within a wave, __syncthreads, the LDS counters' release / acquire in the sources)
static constexpr unsigned TSAN_NO_SYNC = 1u;
#endif

namespace mg { alignas(4096) uint8_t smem[160 * 1024]; }        // the workgroup's LDS (`extern __shared__ uint8_t smem[]` in the kernels)

namespace emu {

// What uninitialised memory holds: EMU_FILL=<byte> XORs the fill patterns below (device buffers 0xA5, pinned host memory 0xBE, LDS at workgroup
// start 0xCD).  A kernel whose results depend on memory nobody wrote gives different results under two fills -- tests/test_emu_cpu.py runs its cases
// under a second fill (the poor man's MemorySanitizer: the real one needs an instrumented python).
static unsigned fill_xor() { static const unsigned v = getenv("EMU_FILL") ? (unsigned)strtoul(getenv("EMU_FILL"), nullptr, 0) & 0xFFu : 0u; return v; }

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
  void* fake = nullptr;                     // (ASan: the fiber's fake-stack handle while it is switched out)
  void* tsan = nullptr;                     // (TSan: the fiber's identity)
};
constexpr size_t STACK_BYTES = 512 * 1024;
static std::vector<Fiber> F;
static int g_cur = -1, g_n = 0;
static ucontext_t g_sched;
static void (*g_tramp)(void*) = nullptr;
static void* g_closure = nullptr;
static bool g_in_kernel = false;
#ifdef EMU_ASAN
static void* g_sched_fake = nullptr;
static const void* g_sched_bottom = nullptr;
static size_t g_sched_size = 0;
#endif
#ifdef EMU_TSAN
static void* g_sched_tsan = nullptr;
static char g_sync_launch, g_sync_done;          // happens-before carriers: launch start, launch end
static char* g_carriers = nullptr;               // ... and, allocated per workgroup (nothing carries over to the next one): [0..15] per wave, [16] the workgroup
#endif

// lane -> scheduler (the lane parks or ends)
static inline void to_sched(Fiber& f, bool last) {
#ifdef EMU_ASAN
  __sanitizer_start_switch_fiber(last ? nullptr : &f.fake, g_sched_bottom, g_sched_size);
#endif
#ifdef EMU_TSAN
  if (last) __tsan_release(&g_sync_done);                  // what the lane did happens-before whatever follows the launch (stream order)
  __tsan_switch_to_fiber(g_sched_tsan, TSAN_NO_SYNC);
#endif
  swapcontext(&f.ctx, &g_sched);
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(f.fake, nullptr, nullptr);
#endif
}

static void fiber_main() {
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_sched_bottom, &g_sched_size);
#endif
#ifdef EMU_TSAN
  __tsan_acquire(&g_sync_launch);                          // everything the host (and earlier launches) did happens-before the kernel
#endif
  g_tramp(g_closure);
  F[g_cur].state = DONE;
  to_sched(F[g_cur], true);
}

unsigned long long xlane(Op op, unsigned long long v, int arg, const void* site) {
  if (!g_in_kernel) { fprintf(stderr, "emu: cross-lane operation outside a kernel\n"); abort(); }
  Fiber& f = F[g_cur];
  f.op = op; f.v = v; f.arg = arg; f.site = site; f.state = PARKED;
#ifdef EMU_TSAN
  // a cross-lane operation orders the lanes that take part in it: everything a lane did before it happens-before what any lane of the wave
  // (the workgroup, for __syncthreads) does after it.  s_sleep (OP_YIELD) orders nothing.
  char* carrier = g_carriers + (op == OP_BLOCK_BARRIER ? 16 : (g_cur / 64) & 15);
  if (op != OP_YIELD) __tsan_release(carrier);
#endif
  to_sched(f, false);
#ifdef EMU_TSAN
  if (op != OP_YIELD) __tsan_acquire(carrier);
#endif
  return f.result;
}

static void resume(int i) {
  g_cur = i;
  g_threadIdx = dim3((unsigned)i, 0, 0);
#ifdef EMU_ASAN
  __sanitizer_start_switch_fiber(&g_sched_fake, F[i].stack, STACK_BYTES);
#endif
#ifdef EMU_TSAN
  __tsan_switch_to_fiber(F[i].tsan, TSAN_NO_SYNC);
#endif
  swapcontext(&g_sched, &F[i].ctx);
#ifdef EMU_ASAN
  __sanitizer_finish_switch_fiber(g_sched_fake, nullptr, nullptr);
#endif
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
  static const bool trace = getenv("EMU_TRACE_DIVERGE") != nullptr;
  if (trace) {
    // a group resolved while other lanes of the wave wait at ANOTHER cross-lane site: real divergence, or this emulator's reconvergence order
    for (int i = lo; i < hi; i++)
      if (F[i].state == PARKED && F[i].op != OP_BLOCK_BARRIER && !in[i - lo]) {
        static std::vector<std::pair<const void*, const void*>> seen;
        std::pair<const void*, const void*> key(site, F[i].site);
        bool dup = false;
        for (auto& k : seen) dup |= k == key;
        if (!dup) {
          seen.push_back(key);
          fprintf(stderr, "emu: diverged: resolving op %d at %p (lane %d) while lane %d waits at %p (op %d), workgroup %u\n", (int)op, site, first, i, F[i].site, (int)F[i].op, g_blockIdx.x);
        }
        break;
      }
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

// EMU_SCHED_SEED=<n != 0>: another legal schedule.  The default one is fixed (waves in index order, each until it polls or blocks; lanes in index
// order; workgroups in index order).  The device promises none of that, so a seeded run shuffles what it is free to shuffle: the order of the
// workgroups of a grid, the order in which the waves of a workgroup get their turn, where a wave is preempted (after any cross-lane operation, with
// probability 1/4), and whether the lanes of a wave run in ascending or descending order between two cross-lane operations.  Parity under several
// seeds (tests/test_emu_cpu.py) = the inter-wave protocols and ring hand-offs do not lean on the one schedule the default run happens to take.
static uint64_t g_sched_state = 0;
static bool sched_fuzz() {
  static const bool on = [] { const char* s = getenv("EMU_SCHED_SEED"); const uint64_t v = s ? strtoull(s, nullptr, 0) : 0ull; g_sched_state = v * 0x9E3779B97F4A7C15ull + 1ull; if (v) fprintf(stderr, "emu: seeded schedule %llu\n", (unsigned long long)v); return v != 0ull; }();
  return on;
}
static uint32_t sched_rand() {                       // xorshift64*
  g_sched_state ^= g_sched_state >> 12; g_sched_state ^= g_sched_state << 25; g_sched_state ^= g_sched_state >> 27;
  return (uint32_t)((g_sched_state * 0x2545F4914F6CDD1Dull) >> 32);
}

static void run_block() {
  const int nw = (g_n + 63) / 64;
  const bool fuzz = sched_fuzz();
  for (;;) {
    bool progress = false, alive = false;
    int order[16];
    for (int w = 0; w < nw; w++) order[w] = w;
    if (fuzz) for (int w = nw - 1; w > 0; w--) std::swap(order[w], order[sched_rand() % (uint32_t)(w + 1)]);
    for (int wi = 0; wi < nw; wi++) {
      const int w = order[wi];
      const int lo = w * 64, hi = std::min(g_n, lo + 64);
      // the wave runs until it yields (s_sleep: a polling loop) or nothing in it can move
      for (int rounds = 0;; rounds++) {
        bool ran = false;
        if (fuzz && (sched_rand() & 1u)) {
          for (int i = hi - 1; i >= lo; i--)
            while (F[i].state == RUNNABLE) { resume(i); ran = true; }
        } else {
          for (int i = lo; i < hi; i++)
            while (F[i].state == RUNNABLE) { resume(i); ran = true; }
        }
        bool yielded = false;
        if (!resolve_wave(w, yielded)) { progress |= ran; break; }
        progress = true;
        if (yielded) break;
        if (fuzz && (sched_rand() & 3u) == 0u) break;          // preempted: the other waves get a turn first
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
#ifdef EMU_ASAN
  __asan_poison_memory_region(mg::smem + lds, sizeof(mg::smem) - lds);      // an access past the launch's dynamic LDS size is an error
#endif
#ifdef EMU_TSAN
  g_sched_tsan = __tsan_get_current_fiber();
#endif
  // (seeded schedules: the workgroups of a grid in a rotated, possibly reversed order -- the device runs them in no particular one)
  const unsigned rot = sched_fuzz() ? sched_rand() % grid.x : 0u;
  const bool rev = sched_fuzz() && (sched_rand() & 1u);
  for (unsigned bi = 0; bi < grid.x; bi++) {
    const unsigned bb = (bi + rot) % grid.x, b = rev ? grid.x - 1u - bb : bb;
    g_blockIdx = dim3(b, 0, 0);
#ifdef EMU_TSAN
    // The workgroups of a grid are unordered on the device and stay unordered here (two workgroups touching one global word without an atomic is
    // reported).  What the emulator REUSES from workgroup to workgroup -- the LDS array, the fibers' stacks -- is mapped afresh (the sanitizer
    // forgets a remapped range and books it as written by the mapping thread = this scheduler, whose release below the lanes acquire when
    // they start); every workgroup gets fresh fiber identities and fresh happens-before carriers.
    if (mmap(mg::smem, sizeof(mg::smem), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0) != (void*)mg::smem) { perror("emu: mmap LDS"); abort(); }
    g_carriers = (char*)malloc(32);
    for (int i = 0; i < n; i++) {
      if (mmap(F[i].stack, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0) != F[i].stack) { perror("emu: mmap stack"); abort(); }
      if (F[i].tsan) __tsan_destroy_fiber(F[i].tsan);
      F[i].tsan = __tsan_create_fiber(0);
    }
#endif
    memset(mg::smem, (int)(0xCDu ^ fill_xor()), lds);  // (LDS contents are undefined at workgroup start: not zero)
#ifdef EMU_TSAN
    __tsan_release(&g_sync_launch);
#endif
    for (int i = 0; i < n; i++) {
      getcontext(&F[i].ctx);
      F[i].ctx.uc_stack.ss_sp = F[i].stack; F[i].ctx.uc_stack.ss_size = STACK_BYTES; F[i].ctx.uc_link = nullptr;
      makecontext(&F[i].ctx, fiber_main, 0);
      F[i].state = RUNNABLE;
    }
    run_block();
#ifdef EMU_TSAN
    free(g_carriers); g_carriers = nullptr;
#endif
  }
#ifdef EMU_TSAN
  __tsan_acquire(&g_sync_done);
#endif
#ifdef EMU_ASAN
  __asan_unpoison_memory_region(mg::smem, sizeof(mg::smem));
#endif
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
#ifdef MG_EMU_SANITIZE
// exact size: the sanitizer's red zone starts at the first byte the library did not ask for
hipError_t hipMalloc(void** p, size_t n) { *p = nullptr; if (posix_memalign(p, 256, n ? n : 1)) *p = nullptr; if (*p) memset(*p, (int)(0xA5u ^ emu::fill_xor()), n); return *p ? hipSuccess : hipErrorInvalidValue; }
#else
hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); if (*p) memset(*p, (int)(0xA5u ^ emu::fill_xor()), (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorInvalidValue; }
#endif
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); if (*p) memset(*p, (int)(0xBEu ^ emu::fill_xor()), n); return *p ? hipSuccess : hipErrorInvalidValue; }   // (pinned host memory is not zeroed either)
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

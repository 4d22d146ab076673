// mg_rng.h — device random streams for the map generators.
//   Pcg64Stream : bit-exact numpy Generator(PCG64(SeedSequence(seed))) as reached through gymnasium.Env.reset
//                 (minigrid_env.py:125) and MiniGridEnv._rand_* (minigrid_env.py:247-311).  numpy is third-party to
//                 the reference; the algorithm below is written from numpy's published one (bit_generator.pyx
//                 SeedSequence, src/pcg64/pcg64.h, src/distributions/distributions.c) and is pinned by
//                 tests/golden/rng_kat.npz + every generator golden.
//   PhiloxStream: Philox4x32-10 counter-based stream keyed by (seed, episode); same draw interface.
// Both expose next32(); bounded integers (Lemire, 32-bit) and the shuffle interval are built on top exactly the
// way numpy builds them, so the two modes share the generator code.
#pragma once
#include "mg_device.h"

namespace mg {

typedef unsigned __int128 u128;

// ---------------- SeedSequence(seed).generate_state(4, uint64) ----------------
MG_HD uint32_t ss_hashmix(uint32_t v, uint32_t& hc) {
  v ^= hc; hc *= 0x931e8875u; v *= hc; v ^= v >> 16; return v;
}
MG_HD uint32_t ss_mix(uint32_t x, uint32_t y) {
  uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; r ^= r >> 16; return r;
}
MG_HD void seedseq_words(uint64_t seed, uint64_t out[4]) {
  uint32_t e0 = (uint32_t)seed, e1 = (uint32_t)(seed >> 32);   // little-endian 32-bit words; e1 present iff non-zero
  uint32_t pool[4];
  uint32_t hc = 0x43b0d7e5u;
  pool[0] = ss_hashmix(e0, hc);
  pool[1] = ss_hashmix(e1, hc);      // absent word hashes as 0, present word as itself: identical when e1 == 0
  pool[2] = ss_hashmix(0u, hc);
  pool[3] = ss_hashmix(0u, hc);
#pragma unroll
  for (int s = 0; s < 4; s++) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
      if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
    }
  }
  uint32_t hb = 0x8b51f9ddu;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t v = pool[i & 3];
    v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; w[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}

// ---------------- PCG64 (XSL-RR 128/64) with numpy's 32-bit half cache ----------------
struct Pcg64Stream {
  static constexpr bool kEpisodic = false;   // carried state; nothing to do at an episode boundary
  u128 state, inc;
  uint32_t has32, cache32;

  MG_HD void step() {
    const u128 mult = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
    state = state * mult + inc;
  }
  MG_HD void seed(uint64_t s) {
    uint64_t w[4];
    seedseq_words(s, w);
    u128 initstate = ((u128)w[0] << 64) | w[1];
    u128 initseq = ((u128)w[2] << 64) | w[3];
    state = 0; inc = (initseq << 1) | 1;
    step(); state += initstate; step();
    has32 = 0; cache32 = 0;
  }
  MG_HD uint64_t next64() {
    step();
    uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    uint64_t x = hi ^ lo;
    uint32_t rot = (uint32_t)(hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
  }
  MG_HD uint32_t next32() {
    if (has32) { has32 = 0; return cache32; }
    uint64_t n = next64();
    has32 = 1; cache32 = (uint32_t)(n >> 32);
    return (uint32_t)n;
  }
  // SoA words: {state_hi, state_lo, inc_hi, inc_lo, has<<32|cache}
  MG_HD void load(const uint64_t* base, size_t n, size_t i) {
    state = ((u128)base[i] << 64) | base[n + i];
    inc = ((u128)base[2 * n + i] << 64) | base[3 * n + i];
    uint64_t c = base[4 * n + i];
    has32 = (uint32_t)(c >> 32) & 1u; cache32 = (uint32_t)c;
  }
  MG_HD void store(uint64_t* base, size_t n, size_t i) const {
    base[i] = (uint64_t)(state >> 64); base[n + i] = (uint64_t)state;
    base[2 * n + i] = (uint64_t)(inc >> 64); base[3 * n + i] = (uint64_t)inc;
    base[4 * n + i] = ((uint64_t)has32 << 32) | cache32;
  }
};

// ---------------- Philox4x32-10 ----------------
MG_HD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Counter-based stream: key = 64-bit seed, counter = (block, episode_lo, episode_hi, 0x4D47 "MG").
// SoA words reuse the PCG layout: {seed, episode, block<<8|pos, buf01, buf23}.
struct PhiloxStream {
  static constexpr bool kEpisodic = true;    // counter restarts per episode (begin_episode)
  uint64_t key, episode;
  uint32_t block, pos;       // pos = next unread word of buf (4 = empty)
  uint32_t buf[4];

  MG_HD void seed(uint64_t s) { key = s; episode = 0; block = 0; pos = 4; buf[0] = buf[1] = buf[2] = buf[3] = 0; }
  MG_HD void begin_episode() { episode++; block = 0; pos = 4; }
  MG_HD uint32_t next32() {
    if (pos >= 4) {
      buf[0] = block++; buf[1] = (uint32_t)episode; buf[2] = (uint32_t)(episode >> 32); buf[3] = 0x4D47u;
      philox4x32_10(buf, (uint32_t)key, (uint32_t)(key >> 32));
      pos = 0;
    }
    return buf[pos++];
  }
  MG_HD void load(const uint64_t* base, size_t n, size_t i) {
    key = base[i]; episode = base[n + i];
    uint64_t bp = base[2 * n + i]; block = (uint32_t)(bp >> 8); pos = (uint32_t)(bp & 0xFF);
    uint64_t a = base[3 * n + i], b = base[4 * n + i];
    buf[0] = (uint32_t)a; buf[1] = (uint32_t)(a >> 32); buf[2] = (uint32_t)b; buf[3] = (uint32_t)(b >> 32);
  }
  MG_HD void store(uint64_t* base, size_t n, size_t i) const {
    base[i] = key; base[n + i] = episode; base[2 * n + i] = ((uint64_t)block << 8) | pos;
    base[3 * n + i] = (uint64_t)buf[0] | ((uint64_t)buf[1] << 32);
    base[4 * n + i] = (uint64_t)buf[2] | ((uint64_t)buf[3] << 32);
  }
};

// ======================================================================================================
// Wave-cooperative streams: ONE wavefront draws for ONE environment.
// A stream is sequential by definition, but both generators can be jumped: the 64 lanes compute the next 64
// raw outputs in parallel, and the (wave-uniform) consumer picks draw k with v_readlane.  The generator code
// that consumes the draws (mg_gen.h) is then uniform control flow: no lane ever waits on another env's
// rejection loop, which is what made one-lane-per-env generation latency-bound.
// ======================================================================================================
MG_D uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
MG_D uint64_t uni64(uint64_t v) { return (uint64_t)uni32((uint32_t)v) | ((uint64_t)uni32((uint32_t)(v >> 32)) << 32); }
MG_D uint32_t lane32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
MG_D uint64_t lane64(uint64_t v, uint32_t l) { return (uint64_t)lane32((uint32_t)v, l) | ((uint64_t)lane32((uint32_t)(v >> 32), l) << 32); }

// LCG jump table: state after k steps = A_k * state + S_k * inc (mod 2^128), A_k = mult^k, S_k = 1 + mult + ... + mult^(k-1)
struct PcgJump { uint64_t a_hi[65], a_lo[65], s_hi[65], s_lo[65]; };
constexpr PcgJump make_pcg_jump() {
  PcgJump t{};
  const u128 mult = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
  u128 a = 1, s = 0;
  for (int k = 0; k <= 64; k++) {
    t.a_hi[k] = (uint64_t)(a >> 64); t.a_lo[k] = (uint64_t)a; t.s_hi[k] = (uint64_t)(s >> 64); t.s_lo[k] = (uint64_t)s;
    s = s * mult + 1; a = a * mult;
  }
  return t;
}
#if defined(__HIPCC__)
__device__ const PcgJump kPcgJump = make_pcg_jump();

// numpy PCG64 stream position = (state, inc, has_uint32, uinteger); same SoA words as Pcg64Stream.
// Buffer: lane l holds the state after l+1 steps from `base` and that step's 64-bit output, i.e. 128 32-bit draws
// (numpy hands out the low half first and caches the high half).
struct WavePcg64 {
  static constexpr bool kEpisodic = false;
  u128 base, inc;              // wave-uniform
  u128 st;                     // per lane
  uint32_t out_lo, out_hi;     // per lane
  uint32_t wpos;               // uniform: 32-bit words consumed from the current buffer, 0..128
  uint32_t pending, cache_in;  // a cached high half carried in from the previous episode comes first
  uint32_t lane;
  uint64_t w_in[5];            // the words as loaded (the caller snapshots them)

  MG_D void refill() {
    const uint32_t k = lane + 1u;
    const u128 A = ((u128)kPcgJump.a_hi[k] << 64) | kPcgJump.a_lo[k];
    const u128 S = ((u128)kPcgJump.s_hi[k] << 64) | kPcgJump.s_lo[k];
    st = A * base + S * inc;
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const uint64_t x = hi ^ lo;
    const uint32_t rot = (uint32_t)(hi >> 58);
    const uint64_t o = (x >> rot) | (x << ((64u - rot) & 63u));
    out_lo = (uint32_t)o; out_hi = (uint32_t)(o >> 32);
    wpos = 0;
  }
  MG_D void load(const uint64_t* b, size_t n, size_t i, uint32_t lane_) {
    lane = lane_;
#pragma unroll
    for (int k = 0; k < 5; k++) w_in[k] = uni64(b[k * n + i]);
    base = ((u128)w_in[0] << 64) | w_in[1];
    inc = ((u128)w_in[2] << 64) | w_in[3];
    pending = (uint32_t)(w_in[4] >> 32) & 1u; cache_in = (uint32_t)w_in[4];
    refill();
  }
  MG_D uint32_t next32() {
    if (pending) { pending = 0; return cache_in; }
    if (wpos == 128u) {
      cache_in = lane32(out_hi, 63);
      base = ((u128)lane64((uint64_t)(st >> 64), 63) << 64) | lane64((uint64_t)st, 63);
      refill();
    }
    const uint32_t j = wpos >> 1;
    const uint32_t v = (wpos & 1u) ? lane32(out_hi, j) : lane32(out_lo, j);
    wpos++;
    return v;
  }
  // every lane holds the same final words; the caller lets one lane write them
  MG_D void final_words(uint64_t w[5]) const {
    const uint32_t nout = (wpos + 1u) >> 1;
    const uint32_t l = nout ? nout - 1u : 0u;
    const uint64_t sh = lane64((uint64_t)(st >> 64), l), sl = lane64((uint64_t)st, l);
    const uint32_t ch = lane32(out_hi, l);
    w[0] = nout ? sh : (uint64_t)(base >> 64); w[1] = nout ? sl : (uint64_t)base;
    w[2] = (uint64_t)(inc >> 64); w[3] = (uint64_t)inc;
    const uint32_t has = pending ? 1u : (wpos & 1u);
    const uint32_t cache = nout ? ch : cache_in;
    w[4] = ((uint64_t)has << 32) | cache;
  }
};

// Philox4x32-10: lane l computes counter block (bbase + l) of the episode = 256 draws per buffer.
struct WavePhilox {
  static constexpr bool kEpisodic = true;
  uint64_t key, episode;       // uniform
  uint32_t bbase, dpos;        // uniform: first block of the buffer, draws consumed from it (0..256)
  uint32_t buf[4];             // per lane
  uint32_t lane;
  uint64_t w_in[5];

  MG_D void refill() {
    buf[0] = bbase + lane; buf[1] = (uint32_t)episode; buf[2] = (uint32_t)(episode >> 32); buf[3] = 0x4D47u;
    philox4x32_10(buf, (uint32_t)key, (uint32_t)(key >> 32));
    dpos = 0;
  }
  MG_D void load(const uint64_t* b, size_t n, size_t i, uint32_t lane_) {
    lane = lane_;
#pragma unroll
    for (int k = 0; k < 5; k++) w_in[k] = uni64(b[k * n + i]);
    key = w_in[0]; episode = w_in[1]; bbase = 0; dpos = 0;
    buf[0] = buf[1] = buf[2] = buf[3] = 0;
  }
  MG_D void begin_episode() { episode++; bbase = 0; refill(); }
  MG_D uint32_t next32() {
    if (dpos == 256u) { bbase += 64u; refill(); }
    const uint32_t l = dpos >> 2, k = dpos & 3u;
    const uint32_t a = lane32(buf[0], l), b = lane32(buf[1], l), c = lane32(buf[2], l), d = lane32(buf[3], l);
    dpos++;
    return k == 0 ? a : k == 1 ? b : k == 2 ? c : d;
  }
  MG_D void final_words(uint64_t w[5]) const {
    const uint32_t nblk = (dpos + 3u) >> 2;
    const uint32_t l = nblk ? nblk - 1u : 0u;
    w[0] = key; w[1] = episode;
    w[2] = ((uint64_t)(bbase + nblk) << 8) | (nblk ? dpos - 4u * (nblk - 1u) : 4u);
    w[3] = (uint64_t)lane32(buf[0], l) | ((uint64_t)lane32(buf[1], l) << 32);
    w[4] = (uint64_t)lane32(buf[2], l) | ((uint64_t)lane32(buf[3], l) << 32);
  }
};
#endif  // __HIPCC__

// ---------------- numpy draw primitives on top of next32() ----------------
// Generator.integers(low, high) for a range that fits 32 bits: range 1 draws nothing; otherwise Lemire's
// nearly-divisionless method with rejection (buffered_bounded_lemire_uint32).  MiniGridEnv._rand_int (247-252).
template <class R>
MG_HD int rand_int(R& r, int low, int high) {
  uint32_t rng = (uint32_t)(high - 1 - low);
  if (rng == 0) return low;
  uint32_t rng_excl = rng + 1u;
  uint64_t m = (uint64_t)r.next32() * rng_excl;
  uint32_t leftover = (uint32_t)m;
  if (leftover < rng_excl) {
    uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
    while (leftover < threshold) { m = (uint64_t)r.next32() * rng_excl; leftover = (uint32_t)m; }
  }
  return low + (int)(m >> 32);
}
// random_interval(max) as used by Generator.shuffle on a list: masked rejection on 32-bit draws
template <class R>
MG_HD uint32_t rand_interval(R& r, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  uint32_t v;
  do { v = r.next32() & mask; } while (v > max);
  return v;
}

}  // namespace mg

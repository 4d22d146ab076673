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

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
  static constexpr bool kWave = false;       // one lane's own stream (k_move_obstacles, k_refill_lane)
  MG_HD void checkpoint() {}                 // (generator restart points matter to the wave-cooperative draw buffers only)
  u128 state, inc;
  uint32_t has32, cache32;

  MG_HD void step() {
    const u128 mult = (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
    state = state * mult + inc;
  }
  MG_HD bool dead() const { return false; }
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
  static constexpr bool kWave = false;
  MG_HD void checkpoint() {}
  uint64_t key, episode;
  uint32_t block, pos;       // pos = next unread word of buf (4 = empty)
  uint32_t buf[4];

  MG_HD bool dead() const { return false; }
  MG_HD void seed(uint64_t s) { key = s; episode = 0; block = 0; pos = 4; buf[0] = buf[1] = buf[2] = buf[3] = 0; }
  MG_HD void begin_episode() { episode++; block = 0; pos = 4; }
  MG_HD uint32_t next32() {
    if (pos >= 4) {
      buf[0] = block++; buf[1] = (uint32_t)episode; buf[2] = (uint32_t)(episode >> 32); buf[3] = 0x4D47u;
      philox4x32_10(buf, (uint32_t)key, (uint32_t)(key >> 32));
      pos = 0;
    }
    // (selects, not buf[pos]: a register array indexed at run time goes to scratch memory)
    const uint32_t v = pos == 0 ? buf[0] : pos == 1 ? buf[1] : pos == 2 ? buf[2] : buf[3];
    pos++;
    return v;
  }
  MG_HD uint64_t next64() { const uint32_t lo = next32(), hi = next32(); return (uint64_t)lo | ((uint64_t)hi << 32); }   // (WavePhilox::next64's order)
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
// A stream is sequential by definition, but both generators can be jumped: the 64 lanes compute the next 64 raw
// outputs in parallel and park them, in consumption order, in a per-wave LDS buffer; the (wave-uniform) consumer in
// mg_gen.h then takes draw k with one broadcast LDS read.  next32() is ~5 instructions and never refills: a pass
// that runs past the buffered draws goes "dead" (returns draw 0, every draw-dependent loop exits), and the caller
// replays the episode from its start with twice the budget.  A replay sees the same draws, so it follows the same
// path and continues.  This keeps the generator code small enough to stay in the instruction cache -- with the
// refill inlined at every draw site the kernel was instruction-fetch bound (12 us for one DoorKey episode).
// ======================================================================================================
MG_D uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
MG_D uint64_t uni64(uint64_t v) { return (uint64_t)uni32((uint32_t)v) | ((uint64_t)uni32((uint32_t)(v >> 32)) << 32); }
MG_D uint32_t lane32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// LDS hand-off inside ONE wave: lanes wrote different addresses, the wave reads them next (DS ops of a wave execute
// in order; the fences only stop the compiler from moving the accesses across).
#define MG_WAVE_LDS_SYNC()                                    \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

constexpr uint32_t GEN_SBASE_ENTRIES = 40;                    // PCG64 stream bases after r refills (16 B each)

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

MG_D u128 pcg_jump(u128 state, u128 inc, uint32_t k) {
  const u128 A = ((u128)kPcgJump.a_hi[k] << 64) | kPcgJump.a_lo[k];
  const u128 S = ((u128)kPcgJump.s_hi[k] << 64) | kPcgJump.s_lo[k];
  return A * state + S * inc;
}

// numpy PCG64 stream position = (state, inc, has_uint32, uinteger); same SoA words as Pcg64Stream.
// One refill = the next 64 raw outputs = 128 32-bit draws (numpy hands out the low half first, then the cached
// high half).  A cached half carried in from the previous episode is simply the first word of the buffer.
struct WavePcg64 {
  static constexpr bool kEpisodic = false;
  static constexpr bool kWave = true;
  static constexpr uint32_t kRefillWords = 128;
  uint32_t* buf;               // LDS: [carried-in half][stream words ...]
  uint64_t* sbase;             // LDS: stream state after r refills (hi, lo), r < GEN_SBASE_ENTRIES
  u128 inc;                    // wave-uniform
  uint32_t off, limit, wpos, refills, cache_in, lane, ck;
  uint32_t swap_at;            // next64() with a half cached: draws swap_at .. swap_at + 2 come from words swap_at + 1, + 2, + 0
  uint32_t reg_even, reg_odd;  // per lane: logical draws 2*lane and 2*lane+1 (the first 128 draws live in registers:
                               // v_readlane is ~10x quicker than the LDS round trip, and most episodes need < 128)
  u128 reg_st;                 // per lane: stream state after lane+1 outputs of the first refill
  u128 si, c64;                // per episode (round 6): S_(lane+1) * inc and S_64 * inc -- the increment's share of a jump does not depend on the state, so a
                               // refill is two 128-bit multiplies (A_(lane+1) * base, A_64 * base) instead of four; the maze levels refill 4-8 times per episode
  uint64_t w_in[5];            // the words as loaded (the caller snapshots them)

  MG_D void prefetch(uint32_t lane_) { lane = lane_; }

  MG_D void load(const uint64_t* b, size_t n, size_t i, uint8_t* lds) {
    sbase = (uint64_t*)lds; buf = (uint32_t*)(lds + GEN_SBASE_ENTRIES * 16);
#pragma unroll
    for (int k = 0; k < 5; k++) w_in[k] = uni64(b[k * n + i]);
    inc = ((u128)w_in[2] << 64) | w_in[3];
    si = (((u128)kPcgJump.s_hi[lane + 1u] << 64) | kPcgJump.s_lo[lane + 1u]) * inc;
    c64 = (((u128)kPcgJump.s_hi[64] << 64) | kPcgJump.s_lo[64]) * inc;
    off = (uint32_t)(w_in[4] >> 32) & 1u; cache_in = (uint32_t)w_in[4];
    sbase[0] = w_in[0]; sbase[1] = w_in[1];
    buf[0] = cache_in;                       // overwritten by stream word 0 when nothing was carried in
    refills = 0; limit = off; wpos = 0; swap_at = 0x80000000u;
    MG_WAVE_LDS_SYNC();
  }
  MG_D void refill() {
    const u128 base = ((u128)uni64(sbase[2 * refills]) << 64) | uni64(sbase[2 * refills + 1]);
    const u128 st = (((u128)kPcgJump.a_hi[lane + 1u] << 64) | kPcgJump.a_lo[lane + 1u]) * base + si;
    const uint64_t hi = (uint64_t)(st >> 64), lo = (uint64_t)st;
    const uint64_t x = hi ^ lo;
    const uint32_t rot = (uint32_t)(hi >> 58);
    const uint64_t o = (x >> rot) | (x << ((64u - rot) & 63u));
    uint32_t* dst = buf + off + kRefillWords * refills + 2u * lane;
    dst[0] = (uint32_t)o; dst[1] = (uint32_t)(o >> 32);
    if (refills == 0) {
      // logical draw order = [carried-in half] lo0 hi0 lo1 hi1 ...; with a carried-in half everything shifts by one
      const uint32_t prev_hi = (uint32_t)__shfl_up((int)(uint32_t)(o >> 32), 1);
      reg_even = off ? (lane == 0 ? cache_in : prev_hi) : (uint32_t)o;
      reg_odd = off ? (uint32_t)o : (uint32_t)(o >> 32);
      reg_st = st;
    }
    const u128 nb = (((u128)kPcgJump.a_hi[64] << 64) | kPcgJump.a_lo[64]) * base + c64;
    sbase[2 * refills + 2] = (uint64_t)(nb >> 64); sbase[2 * refills + 3] = (uint64_t)nb;
    refills++; limit = off + kRefillWords * refills;
    MG_WAVE_LDS_SYNC();
  }
  MG_D void begin_pass() { wpos = 0; ck = 0; swap_at = 0x80000000u; }
  MG_D bool dead() const { return wpos > limit; }
  MG_D void checkpoint() { ck = wpos; }     // a restart point of the generator: nothing before it is needed again
  MG_D uint32_t word_of(uint32_t p) const { const uint32_t d = p - swap_at; return d < 3u ? swap_at + (d == 2u ? 0u : d + 1u) : p; }
  MG_D uint32_t next32() {
    uint32_t v;
    const uint32_t q = word_of(wpos);
    if (__builtin_expect(q < 128u, 1)) v = (q & 1u) ? lane32(reg_odd, q >> 1) : lane32(reg_even, q >> 1);
    else v = uni32(buf[wpos < limit ? q : 0u]);
    wpos++;
    return v;
  }
  // numpy's next_uint64 (Generator.uniform / random): a whole raw output, low half first.  It does not look at the cached 32-bit
  // half (pcg64_next64 leaves has_uint32 alone): with a half cached, the NEXT output is taken and the cached half stays for the
  // 32-bit draw after it.  In the linear draw order kept here that is a rotation of three words.  The generator must make at least
  // one more 32-bit draw before it ends or sets a checkpoint (LevelGen always does).
  MG_D uint64_t next64() {
    const bool cached = wpos < off || (((wpos - off) & 1u) != 0u);
    if (cached) swap_at = wpos;
    const uint32_t lo = next32(), hi = next32();
    return (uint64_t)lo | ((uint64_t)hi << 32);
  }
  // lane-parallel look-ahead for speculative rejection sampling: logical draw p (a different p per lane); only meaningful for p < window().
  // Round 6: straight out of the draw buffer in LDS (one gather per peek, every buffered draw: logical draw q sits at buf[word_of(q)]) -- the
  // register window of round 2 (two ds_bpermutes per peek) ended at draw 128, so the later attempts of a maze episode (5 whole-level attempts of
  // ~250 draws each) placed their objects one scalar try at a time.
  MG_D uint32_t window() const { return limit; }
  MG_D uint32_t peek_lane(uint32_t p_) const { return buf[min(word_of(p_), limit ? limit - 1u : 0u)]; }
  // stream position after `pos` draws, in numpy's terms.  Every lane computes the same words.
  MG_D void final_words(uint64_t w[5]) const { words_at(wpos, w); }
  // make the checkpoint the new origin of the draw buffer (the draws before it are never replayed)
  MG_D void rebase_to_checkpoint() {
    uint64_t w[5];
    words_at(ck, w);
    MG_WAVE_LDS_SYNC();
    off = (uint32_t)(w[4] >> 32) & 1u; cache_in = (uint32_t)w[4];
    sbase[0] = w[0]; sbase[1] = w[1];
    buf[0] = cache_in;
    refills = 0; limit = off; wpos = 0; ck = 0; swap_at = 0x80000000u;
    MG_WAVE_LDS_SYNC();
  }
  MG_D void words_at(uint32_t pos, uint64_t w[5]) const {
    const uint32_t wpos = pos;
    const uint32_t used = wpos > off ? wpos - off : 0u;      // stream words consumed (the carried-in half is not one)
    const uint32_t nout = (used + 1u) >> 1;                  // raw 64-bit outputs consumed
    const uint32_t r = nout >> 6, k = nout & 63u;
    if (r == 0 && k != 0) {                                  // the usual case: a lane of the first refill holds it
      w[0] = (uint64_t)lane32((uint32_t)(reg_st >> 64), k - 1u) | ((uint64_t)lane32((uint32_t)(reg_st >> 96), k - 1u) << 32);
      w[1] = (uint64_t)lane32((uint32_t)reg_st, k - 1u) | ((uint64_t)lane32((uint32_t)(reg_st >> 32), k - 1u) << 32);
    } else {
      const u128 base = ((u128)uni64(sbase[2 * r]) << 64) | uni64(sbase[2 * r + 1]);
      const u128 st = pcg_jump(base, inc, k);                // k == 0: A = 1, S = 0
      w[0] = (uint64_t)(st >> 64); w[1] = (uint64_t)st;
    }
    w[2] = (uint64_t)(inc >> 64); w[3] = (uint64_t)inc;
    const uint32_t has = wpos < off ? 1u : (used & 1u);      // carried-in half still unread, or a fresh half cached
    const uint32_t cache = nout ? uni32(buf[off + 2u * nout - 1u]) : cache_in;
    w[4] = ((uint64_t)has << 32) | cache;
  }
};

// Philox4x32-10: lane l computes counter block (64 r + l) of the episode; one refill = 256 draws.
struct WavePhilox {
  static constexpr bool kEpisodic = true;
  static constexpr bool kWave = true;
  static constexpr uint32_t kRefillWords = 256;
  uint32_t* buf;
  uint64_t key, episode;       // uniform
  uint32_t cbase, skip;        // the buffer starts at counter block `cbase`; its first `skip` words are already used
  uint32_t off, limit, wpos, refills, lane, ck;    // off/limit/wpos count draws from the buffer origin (off == 0)
  uint32_t reg[4];             // per lane: buffer words 4*lane .. 4*lane+3 of the first refill
  uint64_t w_in[5];

  MG_D void prefetch(uint32_t lane_) { lane = lane_; }
  MG_D void load(const uint64_t* b, size_t n, size_t i, uint8_t* lds) {
    buf = (uint32_t*)(lds + GEN_SBASE_ENTRIES * 16);
#pragma unroll
    for (int k = 0; k < 5; k++) w_in[k] = uni64(b[k * n + i]);
    key = w_in[0]; episode = w_in[1] + 1u;      // the counter restarts with every episode
    cbase = 0; skip = 0; off = 0; refills = 0; limit = 0; wpos = 0; ck = 0;
    buf[0] = 0;
    MG_WAVE_LDS_SYNC();
  }
  MG_D void refill() {
    uint32_t c[4] = { cbase + 64u * refills + lane, (uint32_t)episode, (uint32_t)(episode >> 32), 0x4D47u };
    philox4x32_10(c, (uint32_t)key, (uint32_t)(key >> 32));
    uint32_t* dst = buf + kRefillWords * refills + 4u * lane;
    dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2]; dst[3] = c[3];
    if (refills == 0) { reg[0] = c[0]; reg[1] = c[1]; reg[2] = c[2]; reg[3] = c[3]; }
    refills++; limit = kRefillWords * refills - skip;
    MG_WAVE_LDS_SYNC();
  }
  MG_D void begin_pass() { wpos = 0; ck = 0; }
  MG_D bool dead() const { return wpos > limit; }
  MG_D void checkpoint() { ck = wpos; }
  // counter-based: the checkpoint becomes the origin of a fresh buffer (block-aligned, `skip` words into its block)
  MG_D void rebase_to_checkpoint() {
    MG_WAVE_LDS_SYNC();
    const uint32_t q = ck + skip;
    cbase += q >> 2; skip = q & 3u;
    refills = 0; limit = 0; wpos = 0; ck = 0;
  }
  MG_D uint32_t next32() {
    uint32_t v;
    const uint32_t q = wpos + skip;
    if (__builtin_expect(q < 256u, 1)) {
      const uint32_t l = q >> 2, k = q & 3u;
      const uint32_t a = lane32(reg[0], l), b = lane32(reg[1], l), c = lane32(reg[2], l), d = lane32(reg[3], l);
      v = k == 0 ? a : k == 1 ? b : k == 2 ? c : d;
    } else v = uni32(buf[wpos < limit ? q : 0u]);
    wpos++;
    return v;
  }
  MG_D uint64_t next64() { const uint32_t lo = next32(), hi = next32(); return (uint64_t)lo | ((uint64_t)hi << 32); }
  MG_D uint32_t window() const { return limit; }
  MG_D uint32_t peek_lane(uint32_t p) const { return buf[min(p, limit ? limit - 1u : 0u) + skip]; }       // (logical draw p = buffer word p + skip)
  MG_D void final_words(uint64_t w[5]) const {
    const uint32_t q = wpos + skip;                  // words consumed from the buffer origin
    const uint32_t nblk = (q + 3u) >> 2;
    const uint32_t last = nblk ? 4u * (nblk - 1u) : 0u;
    w[0] = key; w[1] = episode;
    w[2] = ((uint64_t)(cbase + nblk) << 8) | (nblk ? q - last : 4u);
    w[3] = (uint64_t)uni32(buf[last]) | ((uint64_t)uni32(buf[last + 1]) << 32);
    w[4] = (uint64_t)uni32(buf[last + 2]) | ((uint64_t)uni32(buf[last + 3]) << 32);
  }
};
#endif  // __HIPCC__

// ---------------- numpy draw primitives on top of next32() ----------------
// Generator.integers(low, high) for a range that fits 32 bits: range 1 draws nothing; otherwise Lemire's
// nearly-divisionless method with rejection (buffered_bounded_lemire_uint32).  MiniGridEnv._rand_int (247-252).
__host__ __device__ __attribute__((noinline)) inline uint32_t lemire_threshold(uint32_t rng, uint32_t rng_excl) {
  return (0xFFFFFFFFu - rng) % rng_excl;          // taken with probability ~range / 2^32: keep the division out of line
}
template <class R>
MG_HD int rand_int(R& r, int low, int high) {
  uint32_t rng = (uint32_t)(high - 1 - low);
  if (rng == 0) return low;
  uint32_t rng_excl = rng + 1u;
  uint64_t m = (uint64_t)r.next32() * rng_excl;
  uint32_t leftover = (uint32_t)m;
  if (__builtin_expect(leftover < rng_excl, 0)) {
    uint32_t threshold = lemire_threshold(rng, rng_excl);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (R::kWave) threshold = uni32(threshold);     // a call result is not known to be wave-uniform; keep the draw position scalar
#endif
    while (leftover < threshold && !r.dead()) { m = (uint64_t)r.next32() * rng_excl; leftover = (uint32_t)m; }
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
  do { v = r.next32() & mask; } while (v > max && !r.dead());
  return v;
}

}  // namespace mg

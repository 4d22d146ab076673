// mg_verify.h -- RoomGridLevel.step's second half for the sentence levels (roomgrid_level.py:87-104): update_objs_poss after a drop,
// instrs.verify(action) over the instruction record (mg_device.h INSTR_WORDS), and -- max_steps is per episode there (:71-85) -- the
// truncation limit and the reward.  One lane per env on the env's instruction record `I` (global memory) and its grid `g` (wherever
// it lives).  Used by k_verify (after a k_step launch; mg_kernels_aux.h) and inside the step loop of k_roll7<GG_SENTENCE> (mg_roll.h).
// Host-callable (MG_HD): mg_selftest_verify runs it on the CPU against a literal restatement of ActionInstr / And / Before / After over
// randomly built records and states (tests/test_verifier_cpu.py).
#pragma once
#include "mg_step.h"

namespace mg {

// Round 4: the verifier was 66 of the 90 us of a BossLevel step at 131 072 envs (profiles/r4/bosslevel_attr.txt) -- not for its memory accesses
// (staging the records in LDS alone changed nothing) but for its INSTRUCTIONS under divergence: verify(action) is a tree walk whose leaf
// check was inlined at nine call sites, each with up to two 63-entry scans of the position table, and with 64 envs per wavefront some lane
// takes every path in every step.  Restated so that a wave executes each piece once:
//   * the position table is scanned ONCE per step (id_at of the cell in front of the agent: every identity question of a step is about that
//     cell); what the leaves see after this step's own bookkeeping follows from it (a picked-up / opened-away object has left the cell, a
//     dropped one is the object that was carried: objects never share a cell);
//   * the four leaves' verify_action results are computed up front in ONE rolled loop, side-effect free; the tree walk (And / Before / After,
//     verifier.py:464-571) then only looks results up and notes which leaves it LOOKED AT; the side effects of looking at a leaf
//     (preCarrying, lastStepMatch: ActionInstr.verify / PickupInstr / PutNextInstr) are applied to exactly those afterwards;
// ... and (third cut) the record's hot words -- header, the four leaf words, the eight object sets, the eight stale-cell words: 21 of its 40 --
// are LOADED ONCE per step into registers (InstrWords; k_roll7 issues the loads before MiniGridEnv.step's own work, so that their latency -- the
// record lives in global memory, every lane on its own 320 bytes -- is hidden behind it), worked on there, and the changed ones stored back.
// The position table (16 words) stays in memory: one scan, one or two writes.
struct InstrWords {
  uint64_t hd, leaf[4], set[8], stale[8];
  MG_HD void load(const uint64_t* I) {
    hd = I[0];
#pragma unroll
    for (int k = 0; k < 4; k++) leaf[k] = I[IW_LEAF + k];
#pragma unroll
    for (int k = 0; k < 8; k++) { set[k] = I[IW_SET + k]; stale[k] = I[IW_STALE + k]; }
  }
};

struct InstrRef {
  uint64_t* I; const uint8_t* g; int W, H;
  uint32_t w_magic;                  // ceil(2^16 / W): cell index -> row without a division
  uint32_t act, carry_id;            // carry_id: id + 1 of what the agent holds after the action
  int fidx; bool inb;                // the cell in front of the agent after the action
  int fid;                           // id of the object in that cell after this step's bookkeeping, -1 = none tracked
  uint32_t errbits;
  MG_HD uint16_t* pos() const { return (uint16_t*)(I + IW_POS); }
  // first id whose position is `cell` (ids 0 .. 62), two positions per word, from the top down so that the lowest index wins
  MG_HD int id_at(int cell) const {
    const uint32_t* p32 = (const uint32_t*)(I + IW_POS);
    int id = -1;
#pragma unroll 8
    for (int w = 31; w >= 0; w--) {
      const uint32_t v = p32[w];
      if ((int)(v >> 16) == cell && w != 31) id = 2 * w + 1;              // (index 63 is not an id)
      if ((int)(v & 0xFFFFu) == cell) id = 2 * w;
    }
    return id;
  }
  static MG_HD bool in_stale(uint64_t s, int cell) {
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 4; k++) hit |= (int)((s >> (16 * k)) & 0xFFFFull) == cell;
    return hit;
  }
  MG_HD bool adjacent(uint32_t p, uint32_t q) const {                     // Manhattan distance 1 between two cell indices
    const int py = (int)((p * w_magic) >> 16), px = (int)p - py * W, qy = (int)((q * w_magic) >> 16), qx = (int)q - qy * W;
    return abs(px - qx) + abs(py - qy) == 1;
  }
  // an object left `cell` without a refresh of obj_poss (picked up, or a box toggled away): every description tracking it keeps the cell
  MG_HD void left(InstrWords& R, int id, int cell) {
#pragma unroll
    for (int j = 0; j < 8; j++)
      if ((R.set[j] >> id) & 1ull) {
        const uint64_t s = R.stale[j];
        int slot = -1;
#pragma unroll
        for (int k = 3; k >= 0; k--) if (((s >> (16 * k)) & 0xFFFFull) == 0xFFFFull) slot = k;
        if (slot < 0) errbits |= ERR_TRACKED;
        else R.stale[j] = (s & ~(0xFFFFull << (16 * slot))) | ((uint64_t)cell << (16 * slot));
      }
  }
  // verify_action of a leaf on the state after this step's bookkeeping, WITHOUT its side effect (the preCarrying update of the pick-up and
  // put-next instructions): verifier.py GoToInstr :309-316, OpenInstr :270-287, PickupInstr :343-363, PutNextInstr :406-431.
  // L = the leaf word, dset / fset = its description's and fixed description's objects, sd / sf = their stale cells.
  MG_HD uint32_t leaf_result(uint64_t L, uint64_t dset, uint64_t fset, uint64_t sd, uint64_t sf) const {
    const uint32_t verb = (uint32_t)L & 3u, strict = (uint32_t)(L >> 20) & 1u;
    const uint32_t pre = (uint32_t)(L >> 21) & 127u;                      // preCarrying as the leaf last saw it
    if (verb == V_GOTO) {
      if (!inb) return R_CONTINUE;
      const uint32_t c = g[fidx];
      bool hit = in_stale(sd, fidx);
      if (!hit && c != CELL_EMPTY && cell_type(c) != T_WALL) hit = fid >= 0 && ((dset >> fid) & 1ull);
      return hit ? R_SUCCESS : R_CONTINUE;
    }
    if (verb == V_OPEN) {
      if (act != A_TOGGLE || !inb) return R_CONTINUE;
      const uint32_t c = g[fidx];
      if (cell_ref_type(c) != T_DOOR || cell_type(c) == T_BOX_KEY) return R_CONTINUE;
      if (fid >= 0 && ((dset >> fid) & 1ull) && cell_type(c) == T_DOOR) return R_SUCCESS;
      return strict ? R_FAILURE : R_CONTINUE;
    }
    if (verb == V_PICKUP) {
      if (act != A_PICKUP) return R_CONTINUE;
      if (pre == 0u && carry_id != 0u && ((dset >> (carry_id - 1u)) & 1ull)) return R_SUCCESS;
      return (strict && carry_id != 0u) ? R_FAILURE : R_CONTINUE;
    }
    if (strict && act == A_PICKUP && carry_id != 0u) return R_FAILURE;
    if (act != A_DROP) return R_CONTINUE;
    if (pre == 0u || !((dset >> (pre - 1u)) & 1ull)) return R_CONTINUE;
    const uint32_t cur = pos()[pre - 1u];                                 // obj_a.cur_pos: where it was just dropped, or (-1, -1)
    if (cur >= POS_GONE) return R_CONTINUE;
    bool next = false;
    uint64_t fs = fset & 0x7FFFFFFFFFFFFFFFull;                           // the fixed description's objects (ids 0 .. 62)
    while (fs) {
      const int m = __builtin_ffsll((long long)fs) - 1;
      fs &= fs - 1ull;
      const uint32_t q = pos()[m];
      if (q < POS_GONE) next |= adjacent(cur, q);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { const uint32_t q = (uint32_t)(sf >> (16 * j)) & 0xFFFFu; if (q != 0xFFFFu) next |= adjacent(cur, q); }
    return next ? R_SUCCESS : R_CONTINUE;
  }
};

// One step's verification of one env AFTER the action was applied (agent `a`, grid `g`); Wd = the record's hot words as loaded before the
// step (InstrWords::load).  Returns the instruction's status (R_CONTINUE / R_SUCCESS / R_FAILURE) and the episode's max_steps; OR-s tracking
// errors into errbits; stores the words it changed.
MG_HD uint32_t verify_action(uint64_t* I, InstrWords& Wd, const uint8_t* g, int W, int H, const Agent& a, uint32_t act, uint32_t& max_steps_out,
                            uint32_t& errbits, int done_actions = 0) {
  InstrRef R;
  R.I = I; R.g = g; R.W = W; R.H = H; R.errbits = 0;
  R.w_magic = (65536u + (uint32_t)W - 1u) / (uint32_t)W;                  // (W is uniform: scalar arithmetic)
  R.act = act;
  const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
  R.inb = (unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H;
  R.fidx = R.inb ? fy * W + fx : 0;
  uint64_t Hd = Wd.hd;
  uint32_t carry_id = (uint32_t)(Hd >> 55) & 127u;
  const InstrWords W0 = Wd;                                               // (what to compare against when storing back)
  // object identity through the action (minigrid_env.py:556-577): a pickup / drop shows as a change of `carrying`; a box that was opened
  // is gone (Box.toggle replaces it by its -- empty -- content).  At most one of the three happened, in the cell in front of the agent.
  const bool picked = a.carry != 0u && carry_id == 0u && R.inb;
  const bool dropped = a.carry == 0u && carry_id != 0u && R.inb;
  const bool box_gone = !picked && !dropped && R.act == A_TOGGLE && R.inb && R.g[R.fidx] == CELL_EMPTY;
  const int pid = R.inb ? R.id_at(R.fidx) : -1;                           // the step's one scan of the position table
  R.fid = pid;
  if (picked) {
    if (pid >= 0) { carry_id = (uint32_t)pid + 1u; R.pos()[pid] = (uint16_t)POS_CARRIED; }
    else R.errbits |= ERR_TRACKED;
    R.fid = -1;
  } else if (dropped) {
    R.fid = (int)carry_id - 1;
    R.pos()[carry_id - 1u] = (uint16_t)R.fidx; carry_id = 0u;
  } else if (box_gone) {
    if (pid >= 0) R.pos()[pid] = (uint16_t)POS_GONE;
    R.fid = -1;
  }
  if ((picked || box_gone) && pid >= 0) R.left(Wd, pid, R.fidx);
  R.carry_id = carry_id;
  if (R.act == A_DROP) {                                                  // update_objs_poss (roomgrid_level.py:92-93, 106-117)
#pragma unroll
    for (int j = 0; j < 8; j++) Wd.stale[j] = ~0ull;
  }
  // the four leaves' results, their lastStepMatch bits, which of them carry a preCarrying (pick up = 1, put next = 3: the odd verbs)
  uint32_t res = 0, lastm = 0, side = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t L = Wd.leaf[k];
    res |= R.leaf_result(L, Wd.set[2 * k], Wd.set[2 * k + 1], Wd.stale[2 * k], Wd.stale[2 * k + 1]) << (2 * k);
    lastm |= ((uint32_t)(L >> 28) & 1u) << k;
    side |= ((uint32_t)L & 1u) << k;
  }
  // ActionInstr.verify (verifier.py:228-242): with use_done_actions only `done` reports -- success iff the previous action completed the
  // instruction (lastStepMatch), failure otherwise; any other action runs verify_action, remembers whether it matched and returns None,
  // which every caller treats like "continue"
  uint32_t looked = 0;
  auto leaf = [&](uint32_t k) -> uint32_t {
    looked |= 1u << k;
    if (!done_actions) return (res >> (2u * k)) & 3u;
    if (act == A_DONE) return ((lastm >> k) & 1u) ? (uint32_t)R_SUCCESS : (uint32_t)R_FAILURE;
    return R_CONTINUE;
  };
  // instrs.verify(action): leaf | And (verifier.py:556-571) | Before / After (:464-486, :507-529) over leaves or And nodes.
  // (the node fields [3:27) and the done states [27:39) of the header as two 32-bit words: the walk shifts by run-time amounts)
  const uint32_t root = (uint32_t)Hd & 7u, nodes = (uint32_t)(Hd >> 3) & 0xFFFFFFu;
  uint32_t dn = (uint32_t)(Hd >> 27) & 0xFFFu;
  auto nodef = [&](uint32_t n) -> uint32_t { return (nodes >> (8u * n)) & 255u; };
  auto done_get = [&](uint32_t n, int which) -> uint32_t { return (dn >> (4u * n + 2u * (uint32_t)which)) & 3u; };
  auto done_set = [&](uint32_t n, int which, uint32_t v) { const uint32_t sh = 4u * n + 2u * (uint32_t)which; dn = (dn & ~(3u << sh)) | (v << sh); };
  // (AndInstr.verify's `use_done_actions and action is self.env.actions.done` branch, verifier.py:561-563, is an IDENTITY test against the enum member:
  // integer actions -- the only kind a vector of actions holds -- never take it: done_actions = 1 is that integer behaviour, what env.step(6) and
  // gymnasium.vector.SyncVectorEnv give the reference; done_actions = 2 is env.step(env.actions.done), the branch taken.  INTEGRATION.md section 4.)
  auto and_verify = [&](uint32_t n) -> uint32_t {
    const uint32_t nd = nodef(n), ia = (nd >> 2) & 7u, ib = (nd >> 5) & 7u;
    if (done_get(n, 0) != R_SUCCESS) done_set(n, 0, leaf(ia));
    if (done_get(n, 1) != R_SUCCESS) done_set(n, 1, leaf(ib));
    if (done_actions == 2 && act == A_DONE && done_get(n, 0) == R_FAILURE && done_get(n, 1) == R_FAILURE) return (uint32_t)R_FAILURE;
    return (done_get(n, 0) == R_SUCCESS && done_get(n, 1) == R_SUCCESS) ? (uint32_t)R_SUCCESS : (uint32_t)R_CONTINUE;
  };
  auto sub_verify = [&](uint32_t idx) -> uint32_t { return idx < 4u ? leaf(idx) : and_verify(idx - 4u); };
  uint32_t status;
  if (root < 4u) status = leaf(root);
  else {
    const uint32_t n = root - 4u, nd = nodef(n), kind = nd & 3u, ia = (nd >> 2) & 7u, ib = (nd >> 5) & 7u;
    if (kind == N_AND) status = and_verify(n);
    else {
      const uint32_t first = kind == N_BEFORE ? ia : ib, second = kind == N_BEFORE ? ib : ia;
      const int wf = kind == N_BEFORE ? 0 : 1, ws = 1 - wf;
      status = R_CONTINUE;
      bool look_at_second = done_get(n, wf) == R_SUCCESS;
      if (!look_at_second) {
        const uint32_t r = sub_verify(first);
        done_set(n, wf, r);
        if (r == R_FAILURE) status = R_FAILURE;
        look_at_second = r == R_SUCCESS;                                  // "return self.verify(action)": the second one sees this action too
      }
      if (look_at_second) {
        const uint32_t r = sub_verify(second);
        done_set(n, ws, r);
        if (r != R_CONTINUE) status = r;
      }
    }
  }
  // what looking at a leaf did to it: verify_action ran (unless use_done_actions answered a `done` from lastStepMatch alone) -- the pick-up and
  // put-next instructions remember what the agent carries NOW (preCarrying is updated only when the leaf is looked at), use_done_actions
  // remembers whether the action matched
  if (!done_actions || act != A_DONE) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if ((looked >> k) & 1u) {
        uint64_t Ln = Wd.leaf[k];
        if ((side >> k) & 1u) Ln = (Ln & ~(127ull << 21)) | ((uint64_t)carry_id << 21);
        if (done_actions) Ln = (Ln & ~(1ull << 28)) | ((uint64_t)(((res >> (2 * k)) & 3u) == R_SUCCESS) << 28);
        Wd.leaf[k] = Ln;
      }
  }
  Hd = (Hd & ~((0xFFFull << 27) | (127ull << 55))) | ((uint64_t)dn << 27) | ((uint64_t)carry_id << 55);
  Wd.hd = Hd;
  // store what changed
  I[0] = Hd;
#pragma unroll
  for (int k = 0; k < 4; k++) if (Wd.leaf[k] != W0.leaf[k]) I[IW_LEAF + k] = Wd.leaf[k];
#pragma unroll
  for (int k = 0; k < 8; k++) if (Wd.stale[k] != W0.stale[k]) I[IW_STALE + k] = Wd.stale[k];
  max_steps_out = (uint32_t)(Hd >> 39) & 0xFFFFu;
  errbits |= R.errbits;
  return status;
}
// (the form that loads the words itself: k_verify, after a step kernel)
MG_HD uint32_t verify_action(uint64_t* I, const uint8_t* g, int W, int H, const Agent& a, uint32_t act, uint32_t& max_steps_out, uint32_t& errbits,
                            int done_actions = 0) {
  InstrWords Wd;
  Wd.load(I);
  return verify_action(I, Wd, g, W, H, a, act, max_steps_out, errbits, done_actions);
}

}  // namespace mg

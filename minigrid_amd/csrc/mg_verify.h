// mg_verify.h -- RoomGridLevel.step's second half for the sentence levels (roomgrid_level.py:87-104): update_objs_poss after a drop,
// instrs.verify(action) over the instruction record (mg_device.h INSTR_WORDS), and -- max_steps is per episode there (:71-85) -- the
// truncation limit and the reward.  One lane per env on the env's instruction record `I` (global memory) and its grid `g` (wherever
// it lives).  Used by k_verify (after a k_step launch; mg_kernels_aux.h) and inside the step loop of k_roll7<GG_SENTENCE> (mg_roll.h).
#pragma once
#include "mg_step.h"

namespace mg {

struct InstrRef {
  uint64_t* I; const uint8_t* g; int W, H;
  bool done_actions;                 // verifier.py:26 use_done_actions
  uint32_t act, carry_id;            // carry_id: id + 1 of what the agent holds after the action
  int fidx; bool inb;                // the cell in front of the agent after the action
  uint32_t errbits;
  MG_D uint16_t* pos() const { return (uint16_t*)(I + IW_POS); }
  MG_D int id_at(int cell) const { const uint16_t* p = pos(); for (int i = 0; i < 63; i++) if ((int)p[i] == cell) return i; return -1; }
  MG_D bool in_stale(int j, int cell) const {
    const uint64_t s = I[IW_STALE + j];
    bool hit = false;
    for (int k = 0; k < 4; k++) hit |= (int)((s >> (16 * k)) & 0xFFFFull) == cell;
    return hit;
  }
  // an object left `cell` without a refresh of obj_poss (picked up, or a box toggled away): every description tracking it keeps the cell
  MG_D void left(int id, int cell) {
    for (int j = 0; j < 8; j++)
      if ((I[IW_SET + j] >> id) & 1ull) {
        uint64_t s = I[IW_STALE + j];
        int slot = -1;
        for (int k = 3; k >= 0; k--) if (((s >> (16 * k)) & 0xFFFFull) == 0xFFFFull) slot = k;
        if (slot < 0) errbits |= ERR_TRACKED;
        else I[IW_STALE + j] = (s & ~(0xFFFFull << (16 * slot))) | ((uint64_t)cell << (16 * slot));
      }
  }
  // ActionInstr.verify (verifier.py:228-242): with use_done_actions only `done` reports -- success iff the previous action completed this
  // instruction (lastStepMatch: bit 28 of the leaf word), failure otherwise; any other action runs verify_action, remembers whether it
  // matched and returns None, which every caller treats like "continue"
  MG_D uint32_t leaf(int k) {
    if (!done_actions) return leaf_action(k);
    if (act == A_DONE) return ((I[IW_LEAF + k] >> 28) & 1ull) ? (uint32_t)R_SUCCESS : (uint32_t)R_FAILURE;
    const uint32_t r = leaf_action(k);
    I[IW_LEAF + k] = (I[IW_LEAF + k] & ~(1ull << 28)) | ((uint64_t)(r == R_SUCCESS) << 28);
    return R_CONTINUE;
  }
  // verifier.py: GoToInstr :309-316, OpenInstr :270-287, PickupInstr :343-363, PutNextInstr :406-431
  MG_D uint32_t leaf_action(int k) {
    const uint64_t L = I[IW_LEAF + k];
    const uint32_t verb = (uint32_t)L & 3u, strict = (uint32_t)(L >> 20) & 1u;
    const uint64_t dset = I[IW_SET + 2 * k], fset = I[IW_SET + 2 * k + 1];
    if (verb == V_GOTO) {
      if (!inb) return R_CONTINUE;
      const uint32_t c = g[fidx];
      bool hit = in_stale(2 * k, fidx);
      if (!hit && c != CELL_EMPTY && cell_type(c) != T_WALL) { const int id = id_at(fidx); hit = id >= 0 && ((dset >> id) & 1ull); }
      return hit ? R_SUCCESS : R_CONTINUE;
    }
    if (verb == V_OPEN) {
      if (act != A_TOGGLE || !inb) return R_CONTINUE;
      const uint32_t c = g[fidx];
      if (cell_ref_type(c) != T_DOOR || cell_type(c) == T_BOX_KEY) return R_CONTINUE;
      const int id = id_at(fidx);
      if (id >= 0 && ((dset >> id) & 1ull) && cell_type(c) == T_DOOR) return R_SUCCESS;
      return strict ? R_FAILURE : R_CONTINUE;
    }
    const uint32_t pre = (uint32_t)(L >> 21) & 127u;                      // preCarrying: updated only when this leaf is looked at
    I[IW_LEAF + k] = (L & ~(127ull << 21)) | ((uint64_t)carry_id << 21);
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
    const int cx = (int)cur % W, cy = (int)cur / W;
    bool next = false;
    for (int m = 0; m < 63; m++)
      if ((fset >> m) & 1ull) { const uint32_t q = pos()[m]; if (q < POS_GONE) next |= abs(cx - (int)q % W) + abs(cy - (int)q / W) == 1; }
    const uint64_t sf = I[IW_STALE + 2 * k + 1];
    for (int j = 0; j < 4; j++) { const uint32_t q = (uint32_t)(sf >> (16 * j)) & 0xFFFFu; if (q != 0xFFFFu) next |= abs(cx - (int)q % W) + abs(cy - (int)q / W) == 1; }
    return next ? R_SUCCESS : R_CONTINUE;
  }
};

// One step's verification of one env AFTER the action was applied (agent `a`, grid `g`).  Returns the instruction's status
// (R_CONTINUE / R_SUCCESS / R_FAILURE) and the episode's max_steps; OR-s tracking errors into errbits.
MG_D uint32_t verify_action(uint64_t* I, const uint8_t* g, int W, int H, const Agent& a, uint32_t act, uint32_t& max_steps_out, uint32_t& errbits,
                            bool done_actions = false) {
  InstrRef R;
  R.I = I; R.g = g; R.W = W; R.H = H; R.errbits = 0; R.done_actions = done_actions;
  R.act = act;
  const int fx = (int)a.x + dir_dx(a.dir), fy = (int)a.y + dir_dy(a.dir);
  R.inb = (unsigned)fx < (unsigned)W && (unsigned)fy < (unsigned)H;
  R.fidx = R.inb ? fy * W + fx : 0;
  uint64_t Hd = I[0];
  uint32_t carry_id = (uint32_t)(Hd >> 55) & 127u;
  // object identity through the action (minigrid_env.py:556-577): a pickup / drop shows as a change of `carrying`
  if (a.carry != 0u && carry_id == 0u && R.inb) {
    const int id = R.id_at(R.fidx);
    if (id >= 0) { carry_id = (uint32_t)id + 1u; R.pos()[id] = (uint16_t)POS_CARRIED; R.left(id, R.fidx); }
    else R.errbits |= ERR_TRACKED;
  } else if (a.carry == 0u && carry_id != 0u && R.inb) {
    R.pos()[carry_id - 1u] = (uint16_t)R.fidx; carry_id = 0u;
  } else if (R.act == A_TOGGLE && R.inb && R.g[R.fidx] == CELL_EMPTY) {
    const int id = R.id_at(R.fidx);                                       // a box was opened: Box.toggle replaces it by its (empty) content
    if (id >= 0) { R.pos()[id] = (uint16_t)POS_GONE; R.left(id, R.fidx); }
  }
  R.carry_id = carry_id;
  if (R.act == A_DROP) for (int j = 0; j < 8; j++) I[IW_STALE + j] = ~0ull;          // update_objs_poss (roomgrid_level.py:92-93, 106-117)
  // instrs.verify(action): leaf | And (verifier.py:556-571) | Before / After (:464-486, :507-529) over leaves or And nodes
  const uint32_t root = (uint32_t)Hd & 7u;
  auto nodef = [&](uint32_t n) -> uint32_t { return (uint32_t)(Hd >> (3 + 8 * n)) & 255u; };
  auto done_get = [&](uint32_t n, int which) -> uint32_t { return (uint32_t)(Hd >> (27 + 4 * n + 2 * which)) & 3u; };
  auto done_set = [&](uint32_t n, int which, uint32_t v) { Hd = (Hd & ~(3ull << (27 + 4 * n + 2 * which))) | ((uint64_t)v << (27 + 4 * n + 2 * which)); };
  auto and_verify = [&](uint32_t n) -> uint32_t {
    const uint32_t nd = nodef(n), ia = (nd >> 2) & 7u, ib = (nd >> 5) & 7u;
    if (done_get(n, 0) != R_SUCCESS) done_set(n, 0, R.leaf((int)ia));
    if (done_get(n, 1) != R_SUCCESS) done_set(n, 1, R.leaf((int)ib));
    return (done_get(n, 0) == R_SUCCESS && done_get(n, 1) == R_SUCCESS) ? (uint32_t)R_SUCCESS : (uint32_t)R_CONTINUE;
  };
  auto sub_verify = [&](uint32_t idx) -> uint32_t { return idx < 4u ? R.leaf((int)idx) : and_verify(idx - 4u); };
  uint32_t status;
  if (root < 4u) status = R.leaf((int)root);
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
  Hd = (Hd & ~(127ull << 55)) | ((uint64_t)carry_id << 55);
  I[0] = Hd;
  max_steps_out = (uint32_t)(Hd >> 39) & 0xFFFFu;
  errbits |= R.errbits;
  return status;
}

}  // namespace mg

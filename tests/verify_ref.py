"""Test helper: a literal, sequential restatement of what one step does to a sentence level's instruction record -- RoomGridLevel.step's
second half (envs/babyai/core/roomgrid_level.py:87-104) with ActionInstr.verify (verifier.py:228-242), the four verify_action bodies
(GoToInstr :309-316, OpenInstr :270-287, PickupInstr :343-363, PutNextInstr :406-431), AndInstr (:556-571), BeforeInstr / AfterInstr
(:464-486, :507-529) and the object-identity bookkeeping the record stands in for (minigrid_env.py:556-577) -- written leaf by leaf, call
by call, the way the reference walks the tree: every leaf check scans the position table itself and updates the record in place.  The product's
verify_action (minigrid_amd/csrc/mg_verify.h) reorganises all of this for SIMT execution (one scan per step, leaf results up front, side
effects applied afterwards); tests/test_verifier_cpu.py runs both over random records and requires identical records and statuses.

Record layout: minigrid_amd/csrc/mg_device.h (INSTR_WORDS = 40 u64)."""

IW_LEAF, IW_SET, IW_STALE, IW_POS, IW_MISSION = 1, 5, 13, 21, 37
POS_CARRIED, POS_GONE = 0xFFFF, 0xFFFE
V_GOTO, V_PICKUP, V_OPEN, V_PUTNEXT = 0, 1, 2, 3
N_BEFORE, N_AFTER, N_AND = 1, 2, 3
CONT, SUCCESS, FAILURE = 0, 1, 2
A_PICKUP, A_DROP, A_TOGGLE, A_DONE = 3, 4, 5, 6
T_EMPTY, T_WALL, T_DOOR, T_BOX = 1, 2, 4, 7
ERR_TRACKED = 8
M64 = (1 << 64) - 1
DX, DY = (1, 0, -1, 0), (0, 1, 0, -1)


class Rec:
    """One env's record as a list of 40 Python ints, with the position table as a view."""
    def __init__(self, words):
        self.I = [int(w) for w in words]

    def pos(self, i):
        return (self.I[IW_POS + i // 4] >> (16 * (i % 4))) & 0xFFFF

    def set_pos(self, i, v):
        w, sh = IW_POS + i // 4, 16 * (i % 4)
        self.I[w] = (self.I[w] & ~(0xFFFF << sh) & M64) | (v << sh)

    def id_at(self, cell):
        for i in range(63):
            if self.pos(i) == cell:
                return i
        return -1


def verify_step(words, grid, W, H, agent, act, done_actions):
    """grid[x][y] = (type, colour, state) AFTER the action; agent = (x, y, dir, carried type, carried colour).
    Returns (status, max_steps, errbits, new words)."""
    R = Rec(words)
    I = R.I
    err = 0
    ax, ay, d, ctype, _ccol = agent
    fx, fy = ax + DX[d], ay + DY[d]
    inb = 0 <= fx < W and 0 <= fy < H
    fidx = fy * W + fx if inb else 0
    front = tuple(grid[fx][fy]) if inb else (T_WALL, 0, 0)
    carrying = ctype != 0
    hd = I[0]
    carry_id = (hd >> 55) & 127

    def left(obj, cell):
        nonlocal err
        for j in range(8):
            if (I[IW_SET + j] >> obj) & 1:
                s = I[IW_STALE + j]
                slot = -1
                for k in (3, 2, 1, 0):
                    if (s >> (16 * k)) & 0xFFFF == 0xFFFF:
                        slot = k
                if slot < 0:
                    err |= ERR_TRACKED
                else:
                    I[IW_STALE + j] = (s & ~(0xFFFF << (16 * slot)) & M64) | (cell << (16 * slot))

    # object identity through the action: `carrying` changed, or a box was opened (Box.toggle replaces it by its empty content)
    if carrying and carry_id == 0 and inb:
        obj = R.id_at(fidx)
        if obj >= 0:
            carry_id = obj + 1
            R.set_pos(obj, POS_CARRIED)
            left(obj, fidx)
        else:
            err |= ERR_TRACKED
    elif not carrying and carry_id != 0 and inb:
        R.set_pos(carry_id - 1, fidx)
        carry_id = 0
    elif act == A_TOGGLE and inb and front[0] == T_EMPTY:
        obj = R.id_at(fidx)
        if obj >= 0:
            R.set_pos(obj, POS_GONE)
            left(obj, fidx)
    if act == A_DROP:                                        # update_objs_poss
        for j in range(8):
            I[IW_STALE + j] = M64

    def in_stale(j, cell):
        s = I[IW_STALE + j]
        return any((s >> (16 * k)) & 0xFFFF == cell for k in range(4))

    def adjacent(p, q):
        return abs(p % W - q % W) + abs(p // W - q // W) == 1

    def leaf_action(k):
        L = I[IW_LEAF + k]
        verb, strict = L & 3, (L >> 20) & 1
        dset, fset = I[IW_SET + 2 * k], I[IW_SET + 2 * k + 1]
        if verb == V_GOTO:
            if not inb:
                return CONT
            hit = in_stale(2 * k, fidx)
            if not hit and front[0] not in (T_EMPTY, T_WALL):
                obj = R.id_at(fidx)
                hit = obj >= 0 and (dset >> obj) & 1 == 1
            return SUCCESS if hit else CONT
        if verb == V_OPEN:
            if act != A_TOGGLE or not inb:
                return CONT
            if front[0] != T_DOOR:
                return CONT
            obj = R.id_at(fidx)
            if obj >= 0 and (dset >> obj) & 1 and front[2] == 0:          # the described door, open after the toggle
                return SUCCESS
            return FAILURE if strict else CONT
        pre = (L >> 21) & 127                                             # preCarrying: updated only when this leaf is looked at
        I[IW_LEAF + k] = (L & ~(127 << 21) & M64) | (carry_id << 21)
        if verb == V_PICKUP:
            if act != A_PICKUP:
                return CONT
            if pre == 0 and carry_id != 0 and (dset >> (carry_id - 1)) & 1:
                return SUCCESS
            return FAILURE if (strict and carry_id != 0) else CONT
        if strict and act == A_PICKUP and carry_id != 0:
            return FAILURE
        if act != A_DROP:
            return CONT
        if pre == 0 or not (dset >> (pre - 1)) & 1:
            return CONT
        cur = R.pos(pre - 1)
        if cur >= POS_GONE:
            return CONT
        nxt = False
        for m in range(63):
            if (fset >> m) & 1:
                q = R.pos(m)
                if q < POS_GONE and adjacent(cur, q):
                    nxt = True
        sf = I[IW_STALE + 2 * k + 1]
        for j in range(4):
            q = (sf >> (16 * j)) & 0xFFFF
            if q != 0xFFFF and adjacent(cur, q):
                nxt = True
        return SUCCESS if nxt else CONT

    def leaf(k):                                              # ActionInstr.verify
        if not done_actions:
            return leaf_action(k)
        if act == A_DONE:
            return SUCCESS if (I[IW_LEAF + k] >> 28) & 1 else FAILURE
        r = leaf_action(k)
        I[IW_LEAF + k] = (I[IW_LEAF + k] & ~(1 << 28) & M64) | ((1 if r == SUCCESS else 0) << 28)
        return CONT

    def node(n):
        return (hd >> (3 + 8 * n)) & 255

    def done_get(n, which):
        return (hd >> (27 + 4 * n + 2 * which)) & 3

    def done_set(n, which, v):
        nonlocal hd
        sh = 27 + 4 * n + 2 * which
        hd = (hd & ~(3 << sh) & M64) | (v << sh)

    def and_verify(n):
        nd = node(n)
        ia, ib = (nd >> 2) & 7, (nd >> 5) & 7
        if done_get(n, 0) != SUCCESS:
            done_set(n, 0, leaf(ia))
        if done_get(n, 1) != SUCCESS:
            done_set(n, 1, leaf(ib))
        if done_actions == 2 and act == A_DONE and done_get(n, 0) == FAILURE and done_get(n, 1) == FAILURE:
            return FAILURE                                    # `action is self.env.actions.done` (verifier.py:561): stepped with the enum member
        return SUCCESS if done_get(n, 0) == SUCCESS and done_get(n, 1) == SUCCESS else CONT

    def sub_verify(idx):
        return leaf(idx) if idx < 4 else and_verify(idx - 4)

    root = hd & 7
    if root < 4:
        status = leaf(root)
    else:
        n = root - 4
        nd = node(n)
        kind, ia, ib = nd & 3, (nd >> 2) & 7, (nd >> 5) & 7
        if kind == N_AND:
            status = and_verify(n)
        else:
            first, second = (ia, ib) if kind == N_BEFORE else (ib, ia)
            wf = 0 if kind == N_BEFORE else 1
            ws = 1 - wf
            status = CONT
            look_at_second = done_get(n, wf) == SUCCESS
            if not look_at_second:
                r = sub_verify(first)
                done_set(n, wf, r)
                if r == FAILURE:
                    status = FAILURE
                look_at_second = r == SUCCESS
            if look_at_second:
                r = sub_verify(second)
                done_set(n, ws, r)
                if r != CONT:
                    status = r
    hd = (hd & ~(127 << 55) & M64) | (carry_id << 55)
    I[0] = hd
    return status, (hd >> 39) & 0xFFFF, err, I

"""oracle/render.py — CPU restatement (numpy) of the reference's RGB observation path.  TEST INFRASTRUCTURE ONLY:
imported by tests/, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg, never by the product.

Parity status: PINNED — `tests/test_oracle_golden.py` checks `tile_atlas()` against EVERY tile `Grid.render_tile`
produces for tile sizes 4/8/12/16 (tests/golden/rgb_atlas.npz) and `render_full`/`render_pov` against
`RGBImgObsWrapper` / `RGBImgPartialObsWrapper` frames recorded from the unmodified reference along rollouts
(tests/golden/rgb*_<env>.npz, written by oracle/make_golden.py rgb).

What it restates (reference file:line):
  utils/rendering.py:8-22     downsample        mean over the sub-sample columns, then over the rows, in float64
  utils/rendering.py:25-37    fill_coords       sample point ((x+0.5)/S, (y+0.5)/S), colour stored into a uint8 image
  utils/rendering.py:40-52    rotate_fn         utils/rendering.py:55-84 point_in_line (float32 end points)
  utils/rendering.py:87-99    point_in_circle / point_in_rect        :102-127 point_in_triangle (float32 corners)
  utils/rendering.py:130-137  highlight_img     img + 0.3 * (255 - img), truncated
  core/world_object.py:116-288  the render() of Goal, Floor, Lava, Wall, Door, Key, Ball, Box
  core/grid.py:145-198        Grid.render_tile  (grid lines, object, agent triangle, highlight, 3x supersampling)
  core/grid.py:200-242        Grid.render       (float tile stored into the uint8 frame = truncation)
  minigrid_env.py:652-666     get_pov_render    :668-714 get_full_render     wrappers.py:287-380 the two wrappers
  core/grid.py:291-328        process_vis blanks the invisible cells of the POV grid
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

COLORS = {0: (255, 0, 0), 1: (0, 255, 0), 2: (0, 0, 255), 3: (112, 39, 195), 4: (255, 255, 0), 5: (100, 100, 100)}  # by COLOR_TO_IDX
T_EMPTY, T_WALL, T_FLOOR, T_DOOR, T_KEY, T_BALL, T_BOX, T_GOAL, T_LAVA = 1, 2, 3, 4, 5, 6, 7, 8, 9
SUBDIVS = 3


def tile_keys():
    """(type, colour, state) of every drawable cell; (1, 0, 0) = empty.  Same order as make_golden.rgb_tile_keys."""
    keys = [(T_EMPTY, 0, 0)]
    for t in (T_WALL, T_FLOOR, T_KEY, T_BALL, T_BOX):
        keys += [(t, c, 0) for c in range(6)]
    keys += [(T_DOOR, c, st) for c in range(6) for st in range(3)]
    keys += [(T_GOAL, 1, 0), (T_LAVA, 0, 0)]
    return keys


class _Canvas:
    """A (S, S, 3) uint8 image with fill_coords over vectorised predicates."""

    def __init__(self, S):
        self.S = S
        self.img = np.zeros((S, S, 3), np.uint8)
        f = (np.arange(S, dtype=np.float64) + 0.5) / S            # rendering.py:33-34
        self.x = np.broadcast_to(f[None, :], (S, S))
        self.y = np.broadcast_to(f[:, None], (S, S))

    def fill(self, mask, color):
        self.img[mask] = np.asarray(color)                         # float colours truncate in the uint8 store

    def rect(self, xmin, xmax, ymin, ymax):
        x, y = self.x, self.y
        return (x >= xmin) & (x <= xmax) & (y >= ymin) & (y <= ymax)

    def circle(self, cx, cy, r):
        x, y = self.x, self.y
        return (x - cx) * (x - cx) + (y - cy) * (y - cy) <= r * r

    def line(self, x0, y0, x1, y1, r):
        p0 = np.array([x0, y0], dtype=np.float32)
        p1 = np.array([x1, y1], dtype=np.float32)
        d = p1 - p0
        dist = np.linalg.norm(d)
        d = d / dist
        xmin, xmax = min(x0, x1) - r, max(x0, x1) + r
        ymin, ymax = min(y0, y1) - r, max(y0, y1) + r
        x, y = self.x, self.y
        box = ~((x < xmin) | (x > xmax) | (y < ymin) | (y > ymax))
        pqx, pqy = x - np.float64(p0[0]), y - np.float64(p0[1])
        a = pqx * np.float64(d[0]) + pqy * np.float64(d[1])
        a = np.clip(a, 0, np.float64(dist))
        px, py = np.float64(p0[0]) + a * np.float64(d[0]), np.float64(p0[1]) + a * np.float64(d[1])
        dd = np.sqrt((x - px) * (x - px) + (y - py) * (y - py))
        return box & (dd <= r)

    def triangle(self, a, b, c, cx, cy, theta):
        """point_in_triangle composed with rotate_fn(cx, cy, theta)."""
        x, y = self.x - cx, self.y - cy
        ct, st = math.cos(-theta), math.sin(-theta)
        x2 = cx + x * ct - y * st
        y2 = cy + y * ct + x * st
        a = np.array(a, dtype=np.float32)
        b = np.array(b, dtype=np.float32)
        c = np.array(c, dtype=np.float32)
        v0, v1 = c - a, b - a
        v2x, v2y = x2 - np.float64(a[0]), y2 - np.float64(a[1])
        dot00, dot01, dot11 = np.dot(v0, v0), np.dot(v0, v1), np.dot(v1, v1)         # float32
        dot02 = np.float64(v0[0]) * v2x + np.float64(v0[1]) * v2y
        dot12 = np.float64(v1[0]) * v2x + np.float64(v1[1]) * v2y
        inv = 1 / (dot00 * dot11 - dot01 * dot01)                                    # float32
        u = (np.float64(dot11) * dot02 - np.float64(dot01) * dot12) * np.float64(inv)
        v = (np.float64(dot00) * dot12 - np.float64(dot01) * dot02) * np.float64(inv)
        return (u >= 0) & (v >= 0) & ((u + v) < 1)


def _draw_object(cv: _Canvas, t, c, st):
    col = np.array(COLORS[c])
    if t == T_GOAL or t == T_WALL:
        cv.fill(cv.rect(0, 1, 0, 1), col)
    elif t == T_FLOOR:
        cv.fill(cv.rect(0.031, 1, 0.031, 1), col / 2)
    elif t == T_LAVA:
        cv.fill(cv.rect(0, 1, 0, 1), (255, 128, 0))
        for i in range(3):
            ylo, yhi = 0.3 + 0.2 * i, 0.4 + 0.2 * i
            cv.fill(cv.line(0.1, ylo, 0.3, yhi, 0.03), (0, 0, 0))
            cv.fill(cv.line(0.3, yhi, 0.5, ylo, 0.03), (0, 0, 0))
            cv.fill(cv.line(0.5, ylo, 0.7, yhi, 0.03), (0, 0, 0))
            cv.fill(cv.line(0.7, yhi, 0.9, ylo, 0.03), (0, 0, 0))
    elif t == T_DOOR:
        if st == 0:                                        # open
            cv.fill(cv.rect(0.88, 1.00, 0.00, 1.00), col)
            cv.fill(cv.rect(0.92, 0.96, 0.04, 0.96), (0, 0, 0))
        elif st == 2:                                      # locked
            cv.fill(cv.rect(0.00, 1.00, 0.00, 1.00), col)
            cv.fill(cv.rect(0.06, 0.94, 0.06, 0.94), 0.45 * col)
            cv.fill(cv.rect(0.52, 0.75, 0.50, 0.56), col)
        else:
            cv.fill(cv.rect(0.00, 1.00, 0.00, 1.00), col)
            cv.fill(cv.rect(0.04, 0.96, 0.04, 0.96), (0, 0, 0))
            cv.fill(cv.rect(0.08, 0.92, 0.08, 0.92), col)
            cv.fill(cv.rect(0.12, 0.88, 0.12, 0.88), (0, 0, 0))
            cv.fill(cv.circle(0.75, 0.50, 0.08), col)
    elif t == T_KEY:
        cv.fill(cv.rect(0.50, 0.63, 0.31, 0.88), col)
        cv.fill(cv.rect(0.38, 0.50, 0.59, 0.66), col)
        cv.fill(cv.rect(0.38, 0.50, 0.81, 0.88), col)
        cv.fill(cv.circle(0.56, 0.28, 0.190), col)
        cv.fill(cv.circle(0.56, 0.28, 0.064), (0, 0, 0))
    elif t == T_BALL:
        cv.fill(cv.circle(0.5, 0.5, 0.31), col)
    elif t == T_BOX:
        cv.fill(cv.rect(0.12, 0.88, 0.12, 0.88), col)
        cv.fill(cv.rect(0.18, 0.82, 0.18, 0.82), (0, 0, 0))
        cv.fill(cv.rect(0.16, 0.84, 0.47, 0.53), col)


def render_tile(t, c, st, agent_dir, highlight, tile_size):
    """Grid.render_tile (core/grid.py:145-198) + the uint8 store of Grid.render (grid.py:236)."""
    cv = _Canvas(tile_size * SUBDIVS)
    cv.fill(cv.rect(0, 0.031, 0, 1), (100, 100, 100))
    cv.fill(cv.rect(0, 1, 0, 0.031), (100, 100, 100))
    if t != T_EMPTY:
        _draw_object(cv, t, c, st)
    if agent_dir is not None:
        cv.fill(cv.triangle((0.12, 0.19), (0.87, 0.50), (0.12, 0.81), 0.5, 0.5, 0.5 * math.pi * agent_dir), (255, 0, 0))
    img = cv.img
    if highlight:
        blend = img + 0.30 * (np.array((255, 255, 255), dtype=np.uint8) - img)
        img = blend.clip(0, 255).astype(np.uint8)
    f = SUBDIVS
    d = img.reshape([tile_size, f, tile_size, f, 3]).mean(axis=3).mean(axis=1)
    out = np.zeros((tile_size, tile_size, 3), np.uint8)
    out[:, :, :] = d
    return out


@lru_cache(maxsize=None)
def tile_atlas(tile_size: int):
    """(atlas, lut): atlas[key][agent: 0 none, 1..4 = dir 0..3][highlight] uint8 tiles; lut[type, colour, state] -> key."""
    keys = tile_keys()
    atlas = np.zeros((len(keys), 5, 2, tile_size, tile_size, 3), np.uint8)
    lut = np.full((16, 8, 4), -1, np.int32)
    for k, (t, c, st) in enumerate(keys):
        lut[t, c if t != T_EMPTY else 0, st] = k
        for ad in range(5):
            for hl in range(2):
                atlas[k, ad, hl] = render_tile(t, c, st, None if ad == 0 else ad - 1, bool(hl), tile_size)
    lut[T_EMPTY, :, :] = 0
    return atlas, lut


DIR_TO_VEC = np.array([(1, 0), (0, 1), (-1, 0), (0, -1)], np.int64)


def _compose(keys, agent_state, hl, tile_size):
    """keys/agent_state/hl: (N, Wt, Ht) indexed [x][y] like Grid.get(i, j) -> (N, Ht*ts, Wt*ts, 3) frames."""
    atlas, _ = tile_atlas(tile_size)
    if (keys < 0).any():
        raise ValueError("cell the reference cannot draw")
    tiles = atlas[keys.transpose(0, 2, 1), agent_state.transpose(0, 2, 1), hl.transpose(0, 2, 1).astype(np.int64)]   # (N, Ht, Wt, ts, ts, 3)
    n, ht, wt = tiles.shape[:3]
    return np.ascontiguousarray(tiles.transpose(0, 1, 3, 2, 4, 5)).reshape(n, ht * tile_size, wt * tile_size, 3)


def vis_from_partial(partial_obs):
    """The vis_mask gen_obs applied: a visible cell encodes type >= 1, an invisible one (0, 0, 0) (grid.py:258-262)."""
    return np.asarray(partial_obs)[..., 0] != 0


def render_full(grid, agent, partial_obs, tile_size=8, highlight=True):
    """RGBImgObsWrapper.observation -> get_full_render (minigrid_env.py:668-714).
    grid (N, W, H, 3) Grid.encode() triples; agent (N, >=3) = x, y, dir; partial_obs (N, V, V, 3) the env's own obs."""
    grid = np.asarray(grid)
    agent = np.asarray(agent)
    n, W, H = grid.shape[:3]
    _, lut = tile_atlas(tile_size)
    keys = lut[grid[..., 0], grid[..., 1], grid[..., 2]]
    vis = vis_from_partial(partial_obs)
    V = vis.shape[1]
    ax, ay, ad = agent[:, 0].astype(np.int64), agent[:, 1].astype(np.int64), agent[:, 2].astype(np.int64)
    f = DIR_TO_VEC[ad]
    r = np.stack([-f[:, 1], f[:, 0]], 1)
    top_left = np.stack([ax, ay], 1) + f * (V - 1) - r * (V // 2)
    hl = np.zeros((n, W, H), bool)
    idx = np.arange(n)
    for vj in range(V):
        for vi in range(V):
            p = top_left - f * vj + r * vi
            ok = vis[:, vi, vj] & (p[:, 0] >= 0) & (p[:, 0] < W) & (p[:, 1] >= 0) & (p[:, 1] < H)
            hl[idx[ok], p[ok, 0], p[ok, 1]] = True
    if not highlight:
        hl[:] = False
    agent_state = np.zeros((n, W, H), np.int64)
    agent_state[idx, ax, ay] = 1 + ad
    return _compose(keys, agent_state, hl, tile_size)


def render_pov(partial_obs, tile_size=8):
    """RGBImgPartialObsWrapper.observation -> get_pov_render (minigrid_env.py:652-666): gen_obs_grid's grid is drawn with
    vis_mask as the highlight.  process_vis has already blanked every invisible cell (grid.py:324-327), so the frame
    follows from the env's own observation: (0, 0, 0) -> an empty, un-highlighted tile; anything else -> that object,
    highlighted; the agent at (V//2, V-1) facing up (dir 3) over what it carries (minigrid_env.py:623-630)."""
    obs = np.asarray(partial_obs)
    vis = vis_from_partial(obs)
    n, V = vis.shape[:2]
    _, lut = tile_atlas(tile_size)
    keys = np.where(vis, lut[obs[..., 0], obs[..., 1], obs[..., 2]], 0)
    agent_state = np.zeros((n, V, V), np.int64)
    agent_state[:, V // 2, V - 1] = 1 + 3
    return _compose(keys, agent_state, vis, tile_size)

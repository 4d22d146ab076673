"""Mission sentences of the BabyAI levels whose instruction is a tree (`Instr.surface`, envs/babyai/core/verifier.py:73-103,
254-268, 319-321, 364-366, 432-434, 487-489, 530-532, 572-574): the device carries the instruction as data (two u64 per env,
layout in include/minigrid_hip.h `mg_outputs.sentence`), the host turns it into the string the reference puts into obs["mission"].
"""
from __future__ import annotations

import numpy as np

_VERB = ("go to ", "pick up ", "open ", "put ")
_TYPE = ("door", "key", "ball", "box")
_COLOR = ("", "red ", "green ", "blue ", "purple ", "yellow ", "grey ")          # 0 = none, COLOR_TO_IDX + 1
_LOC = ("", " on your left", " on your right", " in front of you", " behind you")
_JOIN = {1: ", then ", 2: " after you ", 3: " and "}


def _desc(d: int) -> str:
    return ("a " if (d >> 8) & 1 else "the ") + _COLOR[(d >> 2) & 7] + _TYPE[d & 3] + _LOC[(d >> 5) & 7]


def _leaf(l: int) -> str:
    verb = l & 3
    s = _VERB[verb] + _desc((l >> 2) & 511)
    if verb == 3:
        s += " next to " + _desc((l >> 11) & 511)
    return s


def decode(w0: int, w1: int) -> str:
    leaves = [(w0 >> (20 * k)) & 0xFFFFF for k in range(3)] + [w1 & 0xFFFFF]
    nodes = [(w1 >> (20 + 8 * n)) & 0xFF for n in range(3)]

    def surf(idx: int) -> str:
        if idx < 4:
            return _leaf(leaves[idx])
        nd = nodes[idx - 4]
        return surf((nd >> 2) & 7) + _JOIN[nd & 3] + surf((nd >> 5) & 7)
    return surf((w0 >> 60) & 7)


class SentenceDecoder:
    """(N, 2) u64 mission words -> numpy array of N strings; the distinct sentences of a batch are decoded once and kept."""

    def __init__(self):
        self._cache = {}

    def __call__(self, words: np.ndarray) -> np.ndarray:
        words = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, 2)
        uniq, inv = np.unique(words, axis=0, return_inverse=True)
        out = []
        for a, b in uniq:
            key = (int(a), int(b))
            s = self._cache.get(key)
            if s is None:
                s = self._cache[key] = decode(*key)
            out.append(s)
        return np.asarray(out, dtype=object)[inv.reshape(-1)].astype(str)

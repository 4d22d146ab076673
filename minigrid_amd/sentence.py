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


def _parse_desc(text: str) -> int:
    art, rest = text.split(" ", 1)
    loc = 0
    for k in range(1, 5):
        if rest.endswith(_LOC[k]):
            loc, rest = k, rest[: -len(_LOC[k])]
            break
    words = rest.split(" ")
    typ = _TYPE.index(words[-1])
    col = _COLOR.index(words[0] + " ") if len(words) == 2 else 0
    return typ | (col << 2) | (loc << 5) | ((1 if art == "a" else 0) << 8)


def _parse_leaf(text: str) -> int:
    for verb, prefix in enumerate(_VERB):
        if text.startswith(prefix):
            body = text[len(prefix):]
            if verb == 3:
                d, f = body.split(" next to ")
                return verb | (_parse_desc(d) << 2) | (_parse_desc(f) << 11)
            return verb | (_parse_desc(body) << 2)
    raise ValueError(f"not an instruction: {text!r}")


def encode(sentence: str):
    """The inverse of decode (the tree shape LevelGen can produce: leaf | And | Before / After over leaves or Ands): used by the
    CPU stand-in of the multi-GPU tests and to check decode; the device builds the words from the instruction itself."""
    leaves, nodes = [], []

    def sub(text: str) -> int:
        if " and " in text:
            a, b = text.split(" and ")
            ia = len(leaves); leaves.append(_parse_leaf(a))
            ib = len(leaves); leaves.append(_parse_leaf(b))
            nodes.append(3 | (ia << 2) | (ib << 5))
            return 3 + len(nodes)
        leaves.append(_parse_leaf(text))
        return len(leaves) - 1
    for kind, sep in ((1, ", then "), (2, " after you ")):
        if sep in sentence:
            a, b = sentence.split(sep)
            ia = sub(a)
            ib = sub(b)
            nodes.append(kind | (ia << 2) | (ib << 5))
            root = 3 + len(nodes)
            break
    else:
        root = sub(sentence)
    if len(leaves) > 4 or len(nodes) > 3:
        raise ValueError(f"more than four instructions: {sentence!r}")
    leaves += [0] * (4 - len(leaves))
    nodes += [0] * (3 - len(nodes))
    w0 = leaves[0] | (leaves[1] << 20) | (leaves[2] << 40) | (root << 60)
    w1 = leaves[3] | (nodes[0] << 20) | (nodes[1] << 28) | (nodes[2] << 36)
    return w0, w1


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

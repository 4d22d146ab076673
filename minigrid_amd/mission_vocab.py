"""Mission strings -> fixed-vocabulary word indices: the host-side part of DictObservationSpaceWrapper
(minigrid/wrappers.py:429-554).  The missions of an env id are a small fixed set, so the device only carries a
mission id per env and the token vectors are a per-id table (built once in MiniGridVecEnv).

The vocabulary ORDER is the wrapper's contract (index = position in colours + objects + verbs + extra words,
wrappers.py:474-537); it is restated here as data.  Pinned by the reference doctest wrappers.py:442-447.
"""
from __future__ import annotations

from typing import Dict, List

_COLORS = "red green blue yellow purple grey"
_OBJECTS = "unseen empty wall floor box key ball door goal agent lava"
_VERBS = "pick avoid get find put use open go fetch reach unlock traverse"
_EXTRA = "up the a at , square and then to of rooms near opening must you matching end hallway object from room maze"

MAX_WORDS_IN_MISSION = 50


def minigrid_words() -> Dict[str, int]:
    words = (_COLORS + " " + _OBJECTS + " " + _VERBS + " " + _EXTRA).split()
    assert len(words) == len(set(words))
    return {w: i for i, w in enumerate(words)}


_WORDS = minigrid_words()


def string_to_indices(mission: str, max_words: int = MAX_WORDS_IN_MISSION, offset: int = 1) -> List[int]:
    """Word indices (+offset, 0 = padding) of `mission`, padded to `max_words`; unknown word -> ValueError."""
    out = []
    for word in mission.replace(",", " , ").split():
        if word not in _WORDS:
            raise ValueError(f"Unknown word: {word}")
        out.append(_WORDS[word] + offset)
    assert len(out) < max_words
    return out + [0] * (max_words - len(out))

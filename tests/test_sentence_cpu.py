"""Host side of the sentence levels: the mission words the device carries (include/minigrid_hip.h `mg_outputs.sentence`)
-> `Instr.surface()` strings (envs/babyai/core/verifier.py:73-103, 254-268 and the surface() of each instruction class)."""
import numpy as np

from minigrid_amd.sentence import SentenceDecoder, decode


def d9(t, c=0, loc=0, art=0):
    """type: 0 door 1 key 2 ball 3 box; c: 0 none, COLOR_TO_IDX + 1; loc: 0 none, 1..4 left right front behind"""
    return t | (c << 2) | (loc << 5) | (art << 8)


def leaf(verb, d, f=0):
    return verb | (d << 2) | (f << 11)


def node(kind, a, b):
    return kind | (a << 2) | (b << 5)


def words(leaves, nodes, root):
    leaves = list(leaves) + [0] * (4 - len(leaves))
    nodes = list(nodes) + [0] * (3 - len(nodes))
    w0 = leaves[0] | (leaves[1] << 20) | (leaves[2] << 40) | (root << 60)
    w1 = leaves[3] | (nodes[0] << 20) | (nodes[1] << 28) | (nodes[2] << 36)
    return w0, w1


def test_single_instructions_surface_like_the_reference():
    # strings recorded from the reference (tests/golden/gen_BabyAI-*.npz mission_str)
    assert decode(*words([leaf(0, d9(3, 2))], [], 0)) == "go to the green box"
    assert decode(*words([leaf(0, d9(2, 5, 0, 1))], [], 0)) == "go to a yellow ball"
    assert decode(*words([leaf(1, d9(2, 0, 0, 1))], [], 0)) == "pick up a ball"
    assert decode(*words([leaf(2, d9(0, 6, 3))], [], 0)) == "open the grey door in front of you"
    assert decode(*words([leaf(2, d9(0, 0, 4))], [], 0)) == "open the door behind you"
    assert decode(*words([leaf(3, d9(1, 6, 0, 1), d9(3, 1))], [], 0)) == "put a grey key next to the red box"
    assert decode(*words([leaf(0, d9(1, 6, 2))], [], 0)) == "go to the grey key on your right"


def test_sequences_and_conjunctions():
    # "go to a key and go to a grey door after you go to a yellow key and go to the green box" (gen_BabyAI-GoToSeq-v0, env 0)
    leaves = [leaf(0, d9(1, 0, 0, 1)), leaf(0, d9(0, 6, 0, 1)), leaf(0, d9(1, 5, 0, 1)), leaf(0, d9(3, 2))]
    nodes = [node(3, 0, 1), node(3, 2, 3), node(2, 4, 5)]
    assert decode(*words(leaves, nodes, 6)) == "go to a key and go to a grey door after you go to a yellow key and go to the green box"
    # Before: "open the red door, then open the blue door" (OpenRedBlueDoors)
    assert decode(*words([leaf(2, d9(0, 1)), leaf(2, d9(0, 3))], [node(1, 0, 1)], 4)) == "open the red door, then open the blue door"
    # a leaf on one side, a conjunction on the other
    w = words([leaf(1, d9(2, 5)), leaf(2, d9(0, 6, 0, 1)), leaf(0, d9(3, 4))], [node(3, 1, 2), node(1, 0, 4)], 5)
    assert decode(*w) == "pick up the yellow ball, then open a grey door and go to the purple box"


def test_batch_decoder_keeps_order_and_caches():
    dec = SentenceDecoder()
    a = words([leaf(0, d9(3, 2))], [], 0)
    b = words([leaf(1, d9(2, 0, 0, 1))], [], 0)
    arr = np.asarray([a, b, a, a, b], dtype=np.uint64)
    out = dec(arr)
    assert list(out) == ["go to the green box", "pick up a ball", "go to the green box", "go to the green box", "pick up a ball"]
    assert len(dec._cache) == 2
    assert list(dec(arr[::-1])) == list(out[::-1])


def test_encode_is_the_inverse_of_decode_on_every_golden_mission():
    """Every mission string the reference produced for the BabyAI goldens (sentence levels and one-instruction levels alike) survives
    string -> words -> string; the words are what the multi-GPU record carries for the sentence levels (minigrid_amd/sharded.py)."""
    import glob
    import os

    from conftest import GOLDEN
    from minigrid_amd.sentence import encode
    n = 0
    for f in sorted(glob.glob(os.path.join(GOLDEN, "gen_BabyAI-*.npz"))):
        g = np.load(f)
        if "mission_str" not in g.files:
            continue
        for s in np.unique(g["mission_str"]):
            s = str(s)
            assert decode(*encode(s)) == s, (os.path.basename(f), s)
            n += 1
    assert n > 500

"""The tree must stay shippable: gpurun / the driver snapshot /root/repo minus .git/ and gpurun_out/ onto the GPU box and refuse above 512 MiB
(round 4 lost its whole driver measurement to 700 MB of generated emulator objects).  This walks the repo the same way WITHOUT applying
.gpurunignore (so that a broken ignore pattern cannot hide a problem) and fails far below the limit."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIMIT = 128 << 20


def _walk():
    total, big = 0, []
    for d, dirs, files in os.walk(ROOT):
        rel = os.path.relpath(d, ROOT)
        dirs[:] = [x for x in dirs if not (rel == "." and x in (".git", "gpurun_out"))]
        for f in files:
            p = os.path.join(d, f)
            if os.path.islink(p):
                continue
            try:
                s = os.path.getsize(p)
            except OSError:
                continue
            total += s
            big.append((s, os.path.relpath(p, ROOT)))
    return total, sorted(big, reverse=True)[:8]


def test_repo_snapshot_stays_small():
    total, big = _walk()
    assert total < LIMIT, f"repo snapshot is {total >> 20} MiB (limit {LIMIT >> 20}); largest: {big}"


def test_no_generated_build_trees_in_repo():
    for junk in ("tests/emu/_build", ".scratch"):
        assert not os.path.exists(os.path.join(ROOT, junk)), f"{junk} must live out of the tree (MINIGRID_AMD_EMU_BUILD / MINIGRID_AMD_SCRATCH)"


def test_build_entry_does_not_build_the_emulator():
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    body = src[src.index("def build()"):src.index("def smoke()")]
    assert "build_emu" not in body

"""The REAL kernels under the host compiler's sanitizers (no GPU needed).  tests/emu compiles the product's HIP sources for a host SIMT emulator
(tests/test_emu_cpu.py); built with -fsanitize=... the same run becomes a memory / undefined-behaviour / data-race check of the kernels themselves:

* address,undefined -- device buffers are exact-sized heap blocks and the LDS past the launch's dynamic size is poisoned, so a kernel load or store
  one byte outside a buffer (or outside the LDS it asked for) is an AddressSanitizer error; misaligned accesses, out-of-range shifts, signed overflow
  are UBSan reports.  (The device-side ASan runtime of this image cannot start -- profiles/r3/asan_attempt_*.log; MG_GUARD red zones on the GPU see
  out-of-bounds WRITES only.)
* thread -- every lane is a ThreadSanitizer fiber and the only happens-before edges are the ones the device has: launch order, cross-lane
  operations inside a wave (incl. the sources' MG_WAVE_ORDER / MG_LDS_SYNC markers), __syncthreads, the release / acquire of k_roll7's LDS step
  counters, atomics.  Two waves (or two workgroups) touching one word with none of these between them is a reported race; so is a lane reading
  what its neighbour wrote without a marker (an unmarked lockstep assumption).  One deliberate pattern is annotated in the sources
  (MG_MASKED_READS, mg_roll.h: obs7_view's aligned line reads past the grid edge, masked to walls before use).

Negative controls (tests/emu/emu_probe.cpp) prove each detector fires on a one-line wrong kernel and stays silent on the hand-offs the kernels use.

What runs by default: the probes and a subset of the parity cases under the thread sanitizer (its build takes a minute).  With
MINIGRID_AMD_SANITIZER_TESTS=1: every case of test_emu_cpu.py
-- the product kernels and the MG_LANE_WIDE variant -- under both sanitizer builds (the address,undefined builds take ~5 minutes each).  The logs of
that full run on the committed tree are under profiles/r4/sanitizer_*.txt."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

FULL = os.environ.get("MINIGRID_AMD_SANITIZER_TESTS", "0") == "1"
REPORT = re.compile(r"ThreadSanitizer|AddressSanitizer|runtime error:|LeakSanitizer|DEADLOCK")


_USABLE = {}


def _require_runtime(sanitize):
    """Skip (not fail) where the sanitizer's runtime cannot even start under this kernel / container (ThreadSanitizer is particular about the address-space
    layout): the check is about the KERNELS, and it needs a working detector."""
    if sanitize not in _USABLE:
        import build_emu
        try:
            env = dict(os.environ, **build_emu.sanitizer_env(sanitize))
            out = subprocess.run([sys.executable, "-c", "import numpy, ctypes; print('alive')"], env=env, capture_output=True, text=True, timeout=300)
            _USABLE[sanitize] = None if (out.returncode == 0 and "alive" in out.stdout) else (out.stderr or out.stdout)[-300:]
        except Exception as ex:  # noqa: BLE001
            _USABLE[sanitize] = repr(ex)[:300]
    if _USABLE[sanitize] is not None:
        pytest.skip(f"the {sanitize} sanitizer runtime does not start in this environment: {_USABLE[sanitize]}")


def run_cases(defines, sanitize, cases, timeout=3000):
    """(result lines, sanitizer reports found in stderr, stderr) of tests/emu/run_cases.py on the sanitizer build."""
    import build_emu
    lib = build_emu.build(defines, sanitize=sanitize)
    env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1", **build_emu.sanitizer_env(sanitize))
    for k in [k for k in env if k.startswith("MG_")]:
        del env[k]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_cases.py"), json.dumps(cases)], env=env, capture_output=True, text=True,
                         timeout=timeout)
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    reports = [l for l in out.stderr.splitlines() if REPORT.search(l) and "doesn't fully support makecontext" not in l]
    return lines, reports, out


def check(defines, sanitize, cases):
    _require_runtime(sanitize)
    lines, reports, out = run_cases(defines, sanitize, cases)
    assert len(lines) == len(cases), (out.returncode, out.stdout[-2000:], out.stderr[-4000:])
    bad = [(r["case"], r.get("error"), r.get("where")) for r in lines if not r["ok"]]
    assert not bad, bad
    assert not reports, (len(reports), reports[:6], out.stderr[-6000:])
    return lines


def probe(sanitize, what):
    import build_emu
    lib = build_emu.build([], sanitize=sanitize)
    env = dict(os.environ, **build_emu.sanitizer_env(sanitize))
    code = f"import ctypes; L = ctypes.CDLL({lib!r}); print('rc', L.emu_san_probe({int(what)}))"
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)


def _subset():
    import test_emu_cpu as T
    want = [("MiniGrid-Empty-8x8-v0", None), ("MiniGrid-DoorKey-8x8-v0", "4"), ("BabyAI-GoToRedBall-v0", None), ("MiniGrid-LavaCrossingS9N1-v0", None),
            ("MiniGrid-Dynamic-Obstacles-Random-6x6-v0", None), ("MiniGrid-KeyCorridorS3R3-v0", None), ("BabyAI-MiniBossLevel-v0", None)]
    out = []
    for env_id, nw in want:
        for c in T.PRODUCT_CASES:
            if c["env"] == env_id and (nw is None or c.get("knobs", {}).get("MG_ROLL_NW") == nw):
                out.append(c)
                break
    assert len(out) == len(want)
    return out


# ---- the detectors detect (negative controls) ----

def test_thread_sanitizer_probes():
    _require_runtime("thread")
    for what in (4, 6, 7, 8):           # two waves / two workgroups / two lanes of a wave on one word with nothing between them; a counter published before its data
        out = probe("thread", what)
        assert out.returncode == 0 and "rc 0" in out.stdout, (what, out.stderr[-2000:])
        assert "ThreadSanitizer: data race" in out.stderr and "emu_probe.cpp" in out.stderr, (what, out.stderr[-2000:])
    out = probe("thread", 5)            # the hand-offs the kernels use: __syncthreads, a published LDS counter, a wave barrier, atomics
    assert out.returncode == 0 and "rc 0" in out.stdout and "ThreadSanitizer" not in out.stderr, out.stderr[-3000:]


@pytest.mark.skipif(not FULL, reason="address,undefined build (~5 min): MINIGRID_AMD_SANITIZER_TESTS=1; logs: profiles/r4/sanitizer_*.txt")
def test_address_and_undefined_sanitizer_probes():
    _require_runtime("address,undefined")
    out = probe("address,undefined", 1)
    assert "AddressSanitizer: heap-buffer-overflow" in out.stderr and out.returncode != 0, out.stderr[-2000:]
    out = probe("address,undefined", 2)
    assert "AddressSanitizer: use-after-poison" in out.stderr and out.returncode != 0, out.stderr[-2000:]
    out = probe("address,undefined", 3)
    assert "runtime error: load of misaligned address" in out.stderr and "runtime error: shift exponent 32" in out.stderr, out.stderr[-2000:]
    out = probe("address,undefined", 5)
    assert out.returncode == 0 and not [l for l in out.stderr.splitlines() if REPORT.search(l) and "makecontext" not in l], out.stderr[-2000:]


# ---- the kernels are clean ----

def test_product_kernels_under_the_thread_sanitizer():
    """Parity AND no data race: the four BASELINE levels (fused launches of the driver's lengths, the LOG split with four waves, FullyObs' staged split),
    DynamicObstacles in the loop with SAME_STEP autoreset, a wavefront-per-episode generator with its ring refills, a sentence level with its verifier."""
    import test_emu_cpu as T
    check([], "thread", T.PRODUCT_CASES if FULL else _subset())


@pytest.mark.skipif(not FULL, reason="address,undefined build (~5 min): MINIGRID_AMD_SANITIZER_TESTS=1; logs: profiles/r4/sanitizer_*.txt")
def test_product_kernels_under_the_address_and_undefined_sanitizers():
    import test_emu_cpu as T
    check([], "address,undefined", T.PRODUCT_CASES)


@pytest.mark.skipif(not FULL, reason="MG_LANE_WIDE variant under the thread sanitizer: MINIGRID_AMD_SANITIZER_TESTS=1; logs: profiles/r4/sanitizer_*.txt")
def test_lane_wide_variant_under_the_thread_sanitizer():
    import test_emu_cpu as T
    check(["-DMG_LANE_WIDE=1"], "thread", T.WIDE_CASES)


@pytest.mark.skipif(not FULL, reason="MG_LANE_WIDE variant, address,undefined build (~5 min): MINIGRID_AMD_SANITIZER_TESTS=1; logs: profiles/r4/sanitizer_*.txt")
def test_lane_wide_variant_under_the_address_and_undefined_sanitizers():
    import test_emu_cpu as T
    check(["-DMG_LANE_WIDE=1"], "address,undefined", T.WIDE_CASES)

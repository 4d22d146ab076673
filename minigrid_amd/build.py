"""Build libminigrid_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot).

    python -m minigrid_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is linked against whatever `libamdhip64.so` the process already
has (PyTorch-ROCm ships its own copy; importing torch first makes both share ONE HIP runtime) and falls back to
/opt/rocm/lib through RUNPATH.

The kernels are spread over several translation units (one per k_step rule group, one per generator group and stream
kind: csrc/mg_launch.h) that compile in parallel; an object is rebuilt only when the hash of its own sources changes.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libminigrid_hip.so")
OBJDIR = os.path.join(CSRC, ".obj")
ABI_HEADER = os.path.join("..", "..", "include", "minigrid_hip.h")

_COMMON = ["mg_device.h", "mg_rng.h", "mg_tiles.h", "mg_launch.h"]
_STEP = _COMMON + ["mg_step.h", "mg_roll.h", "mg_verify.h", "mg_dynobs.h", "mg_step_tu.inc"]
_GEN = _COMMON + ["mg_gen.h", "mg_genk.h", "mg_gen_tu.inc"]
# translation unit -> the headers it is built from (its own file included)
UNITS = {
    "mg_api.hip": _COMMON + ["mg_step.h", "mg_roll.h", "mg_verify.h", "mg_dynobs.h", "mg_gen.h", "mg_genk.h", "mg_kernels.h", "mg_kernels_aux.h", "mg_genlane.h", "mg_knobs.h", "mg_host.h", ABI_HEADER],
    # the host self-test entry points (mg_selftest_*): the per-env device code on the CPU, no handle
    "mg_selftest.hip": _COMMON + ["mg_step.h", "mg_roll.h", "mg_verify.h", "mg_dynobs.h", "mg_gen.h", "mg_genk.h", "mg_kernels.h", "mg_genlane.h", "mg_host.h", ABI_HEADER],
    "mg_step_none.hip": _STEP, "mg_step_light.hip": _STEP, "mg_step_roomgrid.hip": _STEP, "mg_step_rooms.hip": _STEP,
    "mg_step_sentence.hip": _STEP, "mg_step_dynobs.hip": _STEP, 
}
# the one-rule units (mg_launch.h MG_ONE_RULE_UNITS): k_roll7 for ONE rule of a rule group
for _n in ("goto", "pickup", "unlock", "gotoobj", "putnear", "gotobig", "pickupdesc", "openfront", "putnext", "opendoor", "fetch", "gotodoor", "redblue", "memory"):
    UNITS[f"mg_step_{_n}.hip"] = _STEP
for _u in ("", "_a", "_b", "_c", "_d"):          # the lane-per-episode generator kernels, by generator function (mg_gen_lane_tu.inc)
    UNITS[f"mg_gen_lane{_u}.hip"] = _GEN + ["mg_genlane.h", "mg_genmr.h", "mg_gen_lane_tu.inc"]
for _g in ("rooms", "sentence", "roomgrid", "light"):
    for _r in ("pcg", "philox"):
        for _k in ("refill", "generate"):
            UNITS[f"mg_gen_{_g}_{_r}_{_k}.hip"] = _GEN
# the slowest units (minutes) first, so that the pool does not end on one of them
SOURCES = sorted(UNITS, key=lambda n: (not n.startswith(("mg_gen_rooms", "mg_gen_lane_")), not n.startswith("mg_gen_sentence")))
HEADERS = sorted({h for deps in UNITS.values() for h in deps})

STAMP = LIB + ".srchash"      # sha256 of the sources the library was built from (travels with the .so; mtimes do not)
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
          "-fvisibility=hidden"]


def _hash_files(names, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for n in names:
        with open(os.path.join(CSRC, n), "rb") as f:
            h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()


def _source_hash() -> str:
    with open(os.path.abspath(__file__), "rb") as f:
        me = hashlib.sha256(f.read()).hexdigest()
    return _hash_files(SOURCES + HEADERS, me)


def step_kernel_hash() -> str:
    """sha256 (first 16 hex digits) of what the step kernels (k_roll7 / k_step, the kernels bench.py's roofline prices) are compiled from:
    the step translation units, their headers and the compiler flags -- NOT mg_api.hip's host code and NOT the generator units.  A committed
    rocprofv3 pass carries the hash of the build it measured (profiles/r*/meta_*.json "step_kernel_srchash"); bench.py refuses to quote a
    pass whose hash differs from the tree it runs on."""
    units = sorted(u for u in UNITS if u.startswith("mg_step_"))
    # (mg_launch.h: declarations of the HOST launch functions of every unit -- the part a step unit sees, under MG_STEP_TU_ONLY, names the functions
    # mg_step_tu.inc defines, which is hashed; the rest belongs to the generators)
    return _hash_files(units + sorted(set(_STEP) - {"mg_launch.h"}), " ".join(CFLAGS) + " gfx950")[:16]


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def build(force: bool = False, verbose: bool = False, missing_hipcc_ok: bool = False, lib: str = LIB, extra_flags=(), extra_link=(),
          tag: str = "", arch: str = "gfx950", flag_units=None) -> str:
    """`lib` / `extra_flags` / `extra_link` / `tag`: an alternative build of the same sources next to the product library (the
    sanitizer build, profiles/asan_build.py; the attribution / A-B variants, profiles/variant_build.py) -- selected at run time with
    MINIGRID_AMD_LIB.  `flag_units`: the translation units the extra flags apply to (the others are shared with the product build)."""
    if lib == LIB and not force and not _stale():
        return LIB
    # one builder at a time per tree (N ranks of a multi-GPU job importing the package at once must not compile into the same
    # object directory): the others wait here and find the library up to date when they get the lock
    import fcntl
    os.makedirs(OBJDIR, exist_ok=True)
    with open(os.path.join(OBJDIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if lib == LIB and not force and not _stale():
            return LIB
        return _build_locked(force, verbose, missing_hipcc_ok, lib, extra_flags, extra_link, tag, arch, flag_units)


def _build_locked(force, verbose, missing_hipcc_ok, lib, extra_flags, extra_link, tag, arch, flag_units=None) -> str:
    # the hash the finished library is stamped with is taken BEFORE anything is compiled: if a source changes while the objects are being built (an
    # editor, or a second process that found the tree stale), the stamp describes the older state and the next build() sees a stale library -- stamping
    # with the hash at the END once marked a library "fresh" whose mg_api.o predated an edit (round 6: every GPU test of a call failed at load)
    stamp_hash = _source_hash() if lib == LIB else None
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if missing_hipcc_ok and os.path.exists(lib):
            return lib                      # a deployed tree: use the library it was shipped with
        raise RuntimeError("hipcc not found: cannot build libminigrid_hip.so (no CPU fallback exists)")
    rocm_lib = os.environ.get("ROCM_PATH", "/opt/rocm") + "/lib"
    # DT_NEEDED must read "libamdhip64.so" (no version suffix): that is the name PyTorch-ROCm's bundled runtime is
    # loaded under, so a process that imported torch first shares ONE HIP runtime with this library; without torch
    # the same name resolves to /opt/rocm/lib/libamdhip64.so through RUNPATH.  A SONAME-less stub gives that name.
    stub_dir = os.path.join(HERE, "csrc", ".stub")
    os.makedirs(stub_dir, exist_ok=True)
    stub = os.path.join(stub_dir, "libamdhip64.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-x", "c", "/dev/null", "-o", stub])
    base_cflags = ["--offload-arch=" + arch] + CFLAGS
    os.makedirs(OBJDIR, exist_ok=True)

    def compile_one(src):
        mine = flag_units is None or src in flag_units
        cflags = base_cflags + (list(extra_flags) if mine else [])
        base = os.path.splitext(src)[0] + (tag if mine else "")
        obj, stamp = os.path.join(OBJDIR, base + ".o"), os.path.join(OBJDIR, base + ".hash")
        want = _hash_files([src] + UNITS[src], " ".join(cflags))
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            return obj
        cmd = [hipcc] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj]
        t0 = time.time()
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(want)
        if verbose:
            print(f"[build] {src}{' ' + tag if tag else ''}: {time.time() - t0:.0f}s", flush=True)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    jobs = int(os.environ.get("MINIGRID_AMD_BUILD_JOBS", "0")) or min(len(SOURCES), os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=" + arch, "-shared", "-fPIC", "-no-hip-rt", "-L" + stub_dir, "-Wl,--no-as-needed", "-lamdhip64",
           "-Wl,--as-needed", "-Wl,-rpath," + rocm_lib, "-Wl,--enable-new-dtags"] + list(extra_link) + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if lib == LIB:
        with open(STAMP, "w") as f:
            f.write(stamp_hash)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

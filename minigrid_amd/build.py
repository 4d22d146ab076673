"""Build libminigrid_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo snapshot).

    python -m minigrid_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is linked against whatever `libamdhip64.so` the process already
has (PyTorch-ROCm ships its own copy; importing torch first makes both share ONE HIP runtime) and falls back to
/opt/rocm/lib through RUNPATH.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libminigrid_hip.so")
SOURCES = ["mg_api.hip"]
HEADERS = ["mg_device.h", "mg_rng.h", "mg_gen.h", "mg_kernels.h", "mg_tiles.h", os.path.join("..", "..", "include", "minigrid_hip.h")]


STAMP = LIB + ".srchash"      # sha256 of the sources the library was built from (travels with the .so; mtimes do not)


def _source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def build(force: bool = False, verbose: bool = False, missing_hipcc_ok: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if missing_hipcc_ok and os.path.exists(LIB):
            return LIB                      # a deployed tree: use the library it was shipped with
        raise RuntimeError("hipcc not found: cannot build libminigrid_hip.so (no CPU fallback exists)")
    rocm_lib = os.environ.get("ROCM_PATH", "/opt/rocm") + "/lib"
    # DT_NEEDED must read "libamdhip64.so" (no version suffix): that is the name PyTorch-ROCm's bundled runtime is
    # loaded under, so a process that imported torch first shares ONE HIP runtime with this library; without torch
    # the same name resolves to /opt/rocm/lib/libamdhip64.so through RUNPATH.  A SONAME-less stub gives that name.
    stub_dir = os.path.join(HERE, "csrc", ".stub")
    os.makedirs(stub_dir, exist_ok=True)
    stub = os.path.join(stub_dir, "libamdhip64.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-x", "c", "/dev/null", "-o", stub])
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-function", "-fvisibility=hidden",
           "-no-hip-rt", "-L" + stub_dir, "-Wl,--no-as-needed", "-lamdhip64", "-Wl,--as-needed",
           "-Wl,-rpath," + rocm_lib, "-Wl,--enable-new-dtags"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_source_hash())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds tests/emu/_build/libminigrid_emu.so = the product's HIP sources (minigrid_amd/csrc/*.hip, unchanged) compiled as plain
C++ against tests/emu/shim/hip/hip_runtime.h + the host SIMT emulator tests/emu/emu_runtime.cpp.  Used only by tests/test_emu_cpu.py (through
MINIGRID_AMD_LIB in a subprocess); the product never loads it.

    python tests/emu/build_emu.py [-DMG_LANE_WIDE=1 ...]      # extra -D switches: the variant builds' kernels under the emulator (tag in the name)
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "minigrid_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CXX_CANDIDATES = ["/opt/rocm/lib/llvm/bin/clang++", "clang++", "g++"]
FLAGS = ["-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-fno-strict-aliasing", "-Wno-unknown-attributes", "-Wno-unused-value", "-Wno-attributes", "-w",
         "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "include")]


def _cxx():
    for c in CXX_CANDIDATES:
        try:
            subprocess.check_output([c, "--version"], stderr=subprocess.STDOUT)
            return c
        except (OSError, subprocess.CalledProcessError):
            continue
    raise RuntimeError("no host C++ compiler found")


def _digest(paths, extra):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(p.encode() + b"\0" + f.read())
    return h.hexdigest()


def build(defines=(), verbose=False):
    tag = "".join("_" + d.lstrip("-D").replace("=", "") for d in defines)
    lib = os.path.join(OUT, f"libminigrid_emu{tag}.so")
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc", ".hip"))]
    deps += [os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "shim", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "minigrid_hip.h"),
             os.path.abspath(__file__)]
    want = _digest(deps, " ".join(defines))
    stamp = lib + ".srchash"
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return lib
    cxx = _cxx()
    objs = []

    def one(src):
        obj = os.path.join(OUT, os.path.basename(src).rsplit(".", 1)[0] + tag + ".o")
        cmd = [cxx] + FLAGS + list(defines) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(2, (os.cpu_count() or 4))) as ex:
        objs = list(ex.map(one, srcs + [os.path.join(HERE, "emu_runtime.cpp")]))
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"])
    with open(stamp, "w") as f:
        f.write(want)
    return lib


if __name__ == "__main__":
    print(build([a for a in sys.argv[1:] if a.startswith("-D")], verbose="-v" in sys.argv))

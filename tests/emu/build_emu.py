#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: builds $MINIGRID_AMD_EMU_BUILD/libminigrid_emu.so (default /tmp/minigrid_emu_build, out of the tree) = the product's HIP sources (minigrid_amd/csrc/*.hip, unchanged) compiled as plain
C++ against tests/emu/shim/hip/hip_runtime.h + the host SIMT emulator tests/emu/emu_runtime.cpp.  Used only by tests/test_emu_cpu.py (through
MINIGRID_AMD_LIB in a subprocess); the product never loads it.

    python tests/emu/build_emu.py [-DMG_LANE_WIDE=1 ...]      # extra -D switches: the variant builds' kernels under the emulator (tag in the name)
    python tests/emu/build_emu.py --sanitize=address,undefined   # the same sources under the host compiler's sanitizers (tag _san_<list>): the kernels'
                                                               # loads and stores checked against exact-sized device buffers and the launch's LDS size,
                                                               # their arithmetic against UBSan; sanitizer_env() = what the loading process needs
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "minigrid_amd", "csrc")
# generated objects NEVER live inside the repo (a 600 MB tests/emu/_build once made the tree too big for the GPU box snapshot)
OUT = os.environ.get("MINIGRID_AMD_EMU_BUILD") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "minigrid_emu_build")
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


def _runtime_lib(cxx, name):
    """Path of a shared sanitizer runtime of the host compiler (to LD_PRELOAD into the python process that dlopens the instrumented library)."""
    out = subprocess.check_output([cxx, "-print-file-name=" + name], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def sanitizer_env(sanitize):
    """Environment additions for a process that loads a build(..., sanitize=...) library: the shared runtime preloaded (python itself is not
    instrumented), leak checking off (the interpreter's own allocations), UBSan reports with stacks and fatal."""
    cxx = _cxx()
    kinds = sanitize.split(",")
    pre = []
    if "address" in kinds:
        rt = _runtime_lib(cxx, "libclang_rt.asan-x86_64.so") or _runtime_lib(cxx, "libasan.so")
        if not rt:
            raise RuntimeError("no shared ASan runtime for " + cxx)
        pre.append(rt)
    elif "undefined" in kinds:
        rt = _runtime_lib(cxx, "libclang_rt.ubsan_standalone-x86_64.so") or _runtime_lib(cxx, "libubsan.so")
        if rt:
            pre.append(rt)
    if "thread" in kinds:
        rt = _runtime_lib(cxx, "libclang_rt.tsan-x86_64.so") or _runtime_lib(cxx, "libtsan.so")
        if not rt:
            raise RuntimeError("no shared TSan runtime for " + cxx)
        pre = [rt]
    stdcxx = _runtime_lib(cxx, "libstdc++.so.6") or _runtime_lib("g++", "libstdc++.so.6")
    if stdcxx:
        pre.append(stdcxx)                    # (the runtimes' C++ interceptors want it resolved before python's extension modules load theirs)
    return {"LD_PRELOAD": ":".join(pre),
            "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=66:allocator_may_return_null=1:detect_stack_use_after_return=0",
            "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=0:report_error_type=1",
            "TSAN_OPTIONS": "report_bugs=1:halt_on_error=0:exitcode=0:report_thread_leaks=0:report_signal_unsafe=0:history_size=4"}


def _kernel_deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc", ".hip"))] + \
           [os.path.join(HERE, "shim", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "minigrid_hip.h")]


def _switches(defines, sanitize):
    return " ".join(FLAGS) + " | " + " ".join(defines) + " | sanitize=" + str(sanitize) + " | v2"


def _plan(defines, sanitize):
    """(library path, stamp path, digest of everything the build depends on: sources, flags, switches)"""
    tag = "".join("_" + d.lstrip("-D").replace("=", "") for d in defines)
    if sanitize:
        tag += "_san_" + sanitize.replace(",", "_")
    lib = os.path.join(OUT, f"libminigrid_emu{tag}.so")
    deps = _kernel_deps() + [os.path.join(HERE, "emu_runtime.cpp")] + ([os.path.join(HERE, "emu_probe.cpp")] if sanitize else [])
    want = _digest(deps, _switches(defines, sanitize) + " | v3")
    return tag, lib, lib + ".srchash", want


def _object_digest(src, defines, sanitize):
    """What ONE object depends on: the kernel translation units on every kernel header, the emulator's own two files on themselves and the shim."""
    own = os.path.basename(src) in ("emu_runtime.cpp", "emu_probe.cpp")
    deps = [src, os.path.join(HERE, "shim", "hip", "hip_runtime.h")] if own else _kernel_deps()
    return _digest(deps, _switches(defines, sanitize) + " | " + os.path.basename(src))


def up_to_date(defines=(), sanitize=None):
    """The library's path if it is built for the current sources, else None (never builds)."""
    _, lib, stamp, want = _plan(list(defines), sanitize)
    return lib if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == want else None


def build(defines=(), verbose=False, sanitize=None):
    defines = list(defines)
    tag, lib, stamp, want = _plan(defines, sanitize)
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    if up_to_date(defines, sanitize):
        return lib
    cxx = _cxx()
    objs = []
    san = []
    if sanitize:
        san = ["-fsanitize=" + sanitize, "-fno-omit-frame-pointer", "-DMG_EMU_SANITIZE=1"] + (["-DMG_EMU_TSAN=1"] if "thread" in sanitize else [])
        if "clang" in os.path.basename(cxx):
            san.append("-shared-libsan")

    def one(src):
        obj = os.path.join(OUT, os.path.basename(src).rsplit(".", 1)[0] + tag + ".o")
        odig = _object_digest(src, defines, sanitize)
        if os.path.exists(obj) and os.path.exists(obj + ".srchash") and open(obj + ".srchash").read().strip() == odig:
            return obj                                  # (an edit of the emulator's runtime does not recompile the 24 kernel translation units)
        flags = san
        if "thread" in (sanitize or "") and os.path.basename(src) == "emu_runtime.cpp":
            # the scheduler's own state is shared by every fiber by design: the runtime is not instrumented, it only talks to the sanitizer's fiber API
            flags = ["-DMG_EMU_SANITIZE=1", "-DEMU_TSAN=1", "-fno-omit-frame-pointer"]
        cmd = [cxx] + FLAGS + flags + list(defines) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(obj + ".srchash", "w") as f:
            f.write(odig)
        return obj
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(2, (os.cpu_count() or 4))) as ex:
        objs = list(ex.map(one, srcs + [os.path.join(HERE, "emu_runtime.cpp")] + ([os.path.join(HERE, "emu_probe.cpp")] if sanitize else [])))
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", lib] + san + objs + ["-lpthread"])
    with open(stamp, "w") as f:
        f.write(want)
    return lib


if __name__ == "__main__":
    san = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--sanitize=")]
    print(build([a for a in sys.argv[1:] if a.startswith("-D")], verbose="-v" in sys.argv, sanitize=san[0] if san else None))

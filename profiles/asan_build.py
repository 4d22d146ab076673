#!/usr/bin/env python3
"""AddressSanitizer build of the product library (host AND device code instrumented: hipcc -fsanitize=address with the xnack+
target, SURVEY.md section 5 "kernels under address-sanitizer").  Same sources, same ABI, next to the product library:

    python profiles/asan_build.py                      # -> minigrid_amd/libminigrid_hip_asan.so  (cross-compiles without a GPU)
    HSA_XNACK=1 LD_PRELOAD=$(python profiles/asan_build.py --runtime) ASAN_OPTIONS=detect_leaks=0 \
        MINIGRID_AMD_LIB=minigrid_amd/libminigrid_hip_asan.so python <anything that uses minigrid_amd>

`--host-only` instruments the host side only (plain gfx950 target: runs on a box without XNACK)."""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def runtime() -> str:
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else ""


if __name__ == "__main__":
    if "--runtime" in sys.argv:
        print(runtime())
        sys.exit(0)
    from minigrid_amd import build as B
    host_only = "--host-only" in sys.argv
    lib = os.path.join(B.HERE, "libminigrid_hip_asan_host.so" if host_only else "libminigrid_hip_asan.so")
    flags = ["-fsanitize=address", "-shared-libasan", "-g", "-fno-omit-frame-pointer"]
    if host_only:
        flags = ["-Xarch_host", "-fsanitize=address", "-shared-libasan", "-g", "-fno-omit-frame-pointer"]
    print(B.build(force="--force" in sys.argv, verbose=True, lib=lib, extra_flags=flags,
                  extra_link=["-fsanitize=address", "-shared-libasan"], tag="_asanh" if host_only else "_asan",
                  arch="gfx950" if host_only else "gfx950:xnack+"))

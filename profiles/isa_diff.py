#!/usr/bin/env python3
"""Device ISA of every translation unit of the library at two source states, compared (no GPU needed: hipcc --cuda-device-only -S for gfx950).

    python profiles/isa_diff.py <git-ref> [<git-ref-2> | WORKTREE]      # default second state: the working tree

What it is for: after the round's GPU minutes are spent, source edits that are meant to leave the product untouched (host-callable self-tests,
emulator / sanitizer annotations under MG_EMU, comments) are checked to produce the SAME device code as the tree the GPU suite last ran on.
Prints one line per translation unit (identical / DIFFERENT + the number of differing lines) and exits 1 on any difference.
Compared: the assembly text without comments and .ident / .file / .loc lines, the compilation unit's id symbol normalised.
    --reuse: compare the assembly already under $MINIGRID_AMD_SCRATCH/isa_diff/ (default /tmp/minigrid_scratch) (no recompile)."""
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(os.environ.get("MINIGRID_AMD_SCRATCH") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "minigrid_scratch"), "isa_diff")   # out of the tree
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-S"]


def checkout(ref):
    d = os.path.join(SCRATCH, "src_" + ref.replace("/", "_"))
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    if ref == "WORKTREE":
        for sub in ("minigrid_amd/csrc", "include"):
            shutil.copytree(os.path.join(ROOT, sub), os.path.join(d, sub))
    else:
        tar = subprocess.run(["git", "-C", ROOT, "archive", ref, "minigrid_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", d], input=tar, check=True)
    return d


def normalise(text):
    out = []
    for l in text.splitlines():
        l = re.sub(r";.*$", "", l).rstrip()
        l = re.sub(r"__hip_cuid_[0-9a-f]+", "__hip_cuid", l)                 # (the compilation unit's id: a hash of the file's path and content)
        l = re.sub(r"(?<=StreamE)Li0E(?=EEvNS_7GenArgsE)", "", l)           # (k_refill_lane<R> became k_refill_lane<R, FN = 0>: the same kernel under a longer name)
        if not l or re.match(r"\s*\.(ident|file|loc|section\s+\.debug|asciz|string)", l):
            continue
        out.append(l)
    return out


def compile_all(d, tag):
    csrc = os.path.join(d, "minigrid_amd", "csrc")
    units = sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
    outdir = os.path.join(SCRATCH, "asm_" + tag)
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir)

    def one(u):
        o = os.path.join(outdir, u[:-4] + ".s")
        subprocess.run(["hipcc"] + FLAGS + ["-I" + os.path.join(d, "include"), "-o", o, os.path.join(csrc, u)], check=True, stderr=subprocess.DEVNULL)
        return u, normalise(open(o).read())
    with concurrent.futures.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        return dict(ex.map(one, units))


def main():
    a = sys.argv[1]
    b = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "WORKTREE"
    if "--reuse" in sys.argv:
        load = lambda tag: {f[:-2] + ".hip": normalise(open(os.path.join(SCRATCH, "asm_" + tag, f)).read()) for f in sorted(os.listdir(os.path.join(SCRATCH, "asm_" + tag)))}
        A, B = load("a"), load("b")
    else:
        A, B = compile_all(checkout(a), "a"), compile_all(checkout(b), "b")
    bad = 0
    for u in sorted(set(A) | set(B)):
        if u not in A or u not in B:
            print(f"{u:44s} only in {'second' if u in B else 'first'}"); bad += 1; continue
        if A[u] == B[u]:
            print(f"{u:44s} identical ({len(A[u])} lines)")
        else:
            import difflib
            n = sum(1 for l in difflib.unified_diff(A[u], B[u], lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---"))
            print(f"{u:44s} DIFFERENT ({n} differing lines of {len(A[u])})"); bad += 1
    print(f"{a} vs {b}: {'device ISA identical in all translation units' if not bad else str(bad) + ' translation unit(s) differ'}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Registers / occupancy / static instruction mix of the kernels of one translation unit (device-only compile, no GPU needed):
    python profiles/isa_stats.py mg_step_none.hip [name-filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "mg_step_none.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
marks = "--marks" in sys.argv
out = f"/tmp/isa_{os.path.splitext(src)[0]}.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "--cuda-device-only", "-S", "-o", out] + (["-DMG_ISA_MARKS"] if marks else []) + [
                       os.path.join(ROOT, "minigrid_amd", "csrc", src)], stderr=subprocess.DEVNULL)
t = open(out).read()
meta = {m.group(1): m.group(2) for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", t, re.S)}
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", t, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name or name not in meta:
        continue
    b = meta[name]
    g = lambda k: int(re.search(k + r":\s+(\d+)", b).group(1))
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = lambda p: sum(1 for i in ins if i.startswith(p))
    vg = g(".vgpr_count")
    occ = min(8, 512 // max(vg, 1)) if vg else 8
    print(f"{name[:60]:60s} VGPR {vg:3d} SGPR {g('.sgpr_count'):3d} scratch {g('.private_segment_fixed_size'):4d} occ {occ} | instr {len(ins):5d} "
          f"v_ {c('v_'):5d} s_ {c('s_'):5d} ds_ {c('ds_'):4d} global_ {c('global_'):3d} waitcnt {c('s_waitcnt'):3d} readlane {c('v_readlane'):3d} "
          f"writelane {c('v_writelane'):3d} perm {c('v_perm'):3d} mul_lo {c('v_mul_lo'):2d}")
    if marks:
        # instructions between consecutive ##MARK comments, in program order (static counts: a section may hold several paths)
        sec, counts, order = "start", {}, []
        for l in body.split("\n"):
            mm = re.search(r"##MARK (\w+)", l)
            if mm:
                sec = mm.group(1)
                continue
            if l.startswith("\t") and not l.strip().startswith((".", ";")):
                op = l.split()[0]
                k = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "ds" if op.startswith("ds_") else "mem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "o"
                if sec not in counts:
                    counts[sec] = {"v": 0, "s": 0, "ds": 0, "mem": 0, "o": 0}; order.append(sec)
                counts[sec][k] += 1
        for k in order:
            print(f"    {k:18s} {counts[k]}")

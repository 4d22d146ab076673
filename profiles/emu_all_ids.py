#!/usr/bin/env python3
"""Every registered env id on the host SIMT emulator of tests/emu (no GPU): the product's kernels -- generator, fused k_roll7 launch, single steps -- against
the oracle, one small case per id (tests/emu/run_cases.py), in parallel subprocesses.  Writes profiles/<round>/emu_all_ids.txt.

    python profiles/emu_all_ids.py [r4] [--sanitize=thread] [--wide]      # --wide: the MG_LANE_WIDE variant of the generator kernels (mg_genlane.h)"""
import concurrent.futures
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu  # noqa: E402
import test_emu_cpu as T  # noqa: E402


def main():
    rnd = next((a for a in sys.argv[1:] if not a.startswith("--")), "r4")
    san = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--sanitize=")), None)
    wide = "--wide" in sys.argv
    lib = build_emu.build(["-DMG_LANE_WIDE=1"] if wide else [], sanitize=san)
    env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1", **(build_emu.sanitizer_env(san) if san else {}))
    cases = T.all_id_cases()
    chunks = [cases[k::7] for k in range(7)]
    t0 = time.time()

    def run(chunk):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_cases.py"), json.dumps(chunk)], env=env, capture_output=True, text=True)
        lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        reports = [l for l in out.stderr.splitlines() if ("Sanitizer" in l or "runtime error:" in l or "DEADLOCK" in l) and "makecontext" not in l]
        return lines, reports, len(chunk)
    with concurrent.futures.ThreadPoolExecutor(max_workers=7) as ex:
        res = list(ex.map(run, chunks))
    lines = [r for ls, _, _ in res for r in ls]
    reports = [r for _, rs, _ in res for r in rs]
    ok = sum(1 for r in lines if r["ok"])
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    name = os.path.join(ROOT, "profiles", rnd, "emu_all_ids" + ("_lanewide" if wide else "") + ("_" + san.replace(",", "_") if san else "") + ".txt")
    with open(name, "w") as f:
        f.write(f"# {os.path.basename(lib)} (tree {head}): every registered id but BabyAI-SynthS5R2-v0, 40 envs, reset(seed=0), fused launches of 16 and 5 steps under the\n"
                f"# device policy, 2 single steps, ring of 4 spares, max_steps 6 (the sentence levels keep their own): every slot's image, reward bytes, flags,\n"
                f"# direction, mission, the final state and stream positions against the oracle.  {ok} of {len(cases)} ok, {len(reports)} sanitizer reports, {time.time() - t0:.0f} s\n")
        for r in sorted(lines, key=lambda r: r["case"]["env"]):
            f.write(f"{'ok  ' if r['ok'] else 'FAIL'} {r['case']['env']:40s} episodes={r.get('episodes')} {'' if r['ok'] else r.get('error')}\n")
        for l in reports[:40]:
            f.write(l + "\n")
    print(f"{name}: {ok}/{len(cases)} ok ({len(lines)} reported back), {len(reports)} reports, {time.time() - t0:.0f} s")
    sys.exit(0 if ok == len(cases) and not reports else 1)


if __name__ == "__main__":
    main()

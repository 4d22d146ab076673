#!/usr/bin/env python3
"""The kernels under the host compiler's sanitizers, on the host SIMT emulator of tests/emu (no GPU): builds the four sanitizer libraries
(product / MG_LANE_WIDE variant x thread / address,undefined), runs the negative controls and every parity case of tests/test_emu_cpu.py under each,
and writes one log per run to profiles/<round>/sanitizer_<variant>_<sanitizers>.txt.

    python profiles/sanitize_emu.py [r4]

tests/test_emu_sanitizers_cpu.py is the pytest form of the same runs (a subset by default, everything with MINIGRID_AMD_SANITIZER_TESTS=1)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu  # noqa: E402
import test_emu_cpu as T  # noqa: E402
import test_emu_sanitizers_cpu as S  # noqa: E402

EXPECT = {"thread": {4: "report", 5: "clean", 6: "report", 7: "report", 8: "report"}, "address,undefined": {1: "report", 2: "report", 3: "report", 5: "clean"}}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r4"
    outdir = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(outdir, exist_ok=True)
    head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "minigrid_amd", "include", "tests/emu"], text=True).strip()
    cxx = subprocess.check_output([build_emu._cxx(), "--version"], text=True).splitlines()[0]
    rc = 0
    for variant, defines, cases in (("product", [], T.PRODUCT_CASES), ("lanewide", ["-DMG_LANE_WIDE=1"], T.WIDE_CASES)):
        for san in ("thread", "address,undefined"):
            t0 = time.time()
            lib = build_emu.build(defines, sanitize=san)
            t_build = time.time() - t0
            log = [f"# {os.path.basename(lib)}: minigrid_amd/csrc/*.hip compiled for the host SIMT emulator with -fsanitize={san}",
                   f"# tree: {head}{' + uncommitted changes' if dirty else ''}; compiler: {cxx}; build {t_build:.0f} s (0 = already built)",
                   f"# command: python profiles/sanitize_emu.py {rnd}", ""]
            if variant == "product":
                log.append("## negative controls (tests/emu/emu_probe.cpp): the detector fires on a one-line wrong kernel, stays silent on the kernels' hand-offs")
                for what, want in EXPECT[san].items():
                    out = S.probe(san, what)
                    reps = [l.strip() for l in out.stderr.splitlines() if S.REPORT.search(l) and "makecontext" not in l]
                    got = "report" if reps else "clean"
                    log.append(f"probe {what}: expected {want}, got {got}{' -- ' + reps[0][:160] if reps else ''}")
                    if got != want:
                        rc = 1
                log.append("")
            t0 = time.time()
            lines, reports, out = S.run_cases(defines, san, cases)
            log.append(f"## {len(cases)} parity cases of tests/test_emu_cpu.py ({'PRODUCT_CASES' if variant == 'product' else 'WIDE_CASES'}), {time.time() - t0:.0f} s")
            for r in lines:
                c = r["case"]
                extra = {k: v for k, v in c.items() if k not in ("env", "n", "launches")}
                log.append(f"{'ok  ' if r['ok'] else 'FAIL'} {c['env']:44s} n={c['n']:<4d} launches={c['launches']} {extra if extra else ''} episodes={r.get('episodes')}")
                if not r["ok"]:
                    log.append("     " + str(r.get("error"))[:400])
                    rc = 1
            if len(lines) != len(cases):
                log.append(f"ONLY {len(lines)} OF {len(cases)} CASES REPORTED BACK (exit code {out.returncode})")
                log.append(out.stderr[-3000:])
                rc = 1
            log.append("")
            log.append(f"## sanitizer reports: {len(reports)}")
            log += reports[:40]
            if reports:
                rc = 1
                log += ["", "## stderr (tail)", out.stderr[-8000:]]
            name = os.path.join(outdir, f"sanitizer_{variant}_{san.replace(',', '_')}.txt")
            with open(name, "w") as f:
                f.write("\n".join(log) + "\n")
            print(f"{name}: {sum(1 for r in lines if r['ok'])}/{len(cases)} cases ok, {len(reports)} reports", flush=True)
    # uninitialised locals: every automatic variable the code does not initialise starts as 0xAA.. (-ftrivial-auto-var-init=pattern) -- same parity?
    import json
    for variant, defines, cases in (("product", [], T.PRODUCT_CASES), ("lanewide", ["-DMG_LANE_WIDE=1"], T.WIDE_CASES)):
        lib = build_emu.build(defines + ["-ftrivial-auto-var-init=pattern"])
        env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "run_cases.py"), json.dumps(cases)], env=env, capture_output=True, text=True)
        lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
        ok = sum(1 for r in lines if r["ok"])
        name = os.path.join(outdir, f"uninit_locals_{variant}.txt")
        with open(name, "w") as f:
            f.write(f"# {os.path.basename(lib)}: the emulated library built with -ftrivial-auto-var-init=pattern (uninitialised locals read 0xAA..), tree {head}\n")
            f.write(f"# {len(cases)} parity cases of tests/test_emu_cpu.py: {ok} ok\n")
            for r in lines:
                f.write(f"{'ok  ' if r['ok'] else 'FAIL'} {r['case']['env']} {'' if r['ok'] else r.get('error')}\n")
        print(f"{name}: {ok}/{len(cases)} cases ok", flush=True)
        if ok != len(cases):
            rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()

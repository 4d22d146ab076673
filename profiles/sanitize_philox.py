import os, subprocess, sys, re
sys.path.insert(0, "/root/repo/tests/emu")
import build_emu
out_lines = []
for san in ("thread", "address,undefined"):
    lib = build_emu.build([], sanitize=san)
    env = dict(os.environ, MINIGRID_AMD_LIB=lib, MINIGRID_AMD_NO_TORCH="1", MINIGRID_AMD_EMU_RERUN="1", MG_SPARE_RING="4", **build_emu.sanitizer_env(san))
    out = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_philox.py", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-k", "state_injection or deterministic"],
                         env=env, capture_output=True, text=True, cwd="/root/repo")
    reps = [l for l in (out.stderr + out.stdout).splitlines() if re.search(r"ThreadSanitizer|AddressSanitizer:|runtime error:", l) and "makecontext" not in l]
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else ""
    out_lines.append(f"-fsanitize={san}: pytest tests/test_gpu_philox.py -k 'state_injection or deterministic' (200 envs, MINIGRID_AMD_EMU_RERUN=1): {tail}; sanitizer reports: {len(reps)}")
    out_lines += reps[:10]
print("\n".join(out_lines))
open("/root/repo/profiles/r4/sanitizer_philox.txt", "w").write("# the Philox-keyed generator kernels (MG_RNG_PHILOX) on the emulator under the host compiler's sanitizers: python profiles/sanitize_philox.py\n" + "\n".join(out_lines) + "\n")

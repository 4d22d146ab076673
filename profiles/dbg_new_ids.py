import subprocess, sys
ids = ["BabyAI-PutNextLocal-v0", "BabyAI-PutNextLocalS5N3-v0", "BabyAI-PutNextS4N1-v0", "BabyAI-PutNextS5N2Carrying-v0", "BabyAI-ActionObjDoor-v0", "BabyAI-OpenDoor-v0", "BabyAI-OpenDoorLoc-v0"]
code = """
import sys, numpy as np
import minigrid_amd as mg
env = mg.make_vec(sys.argv[1], int(sys.argv[2]))
print('created', flush=True)
obs, _ = env.reset(seed=0)
print('reset ok', obs['image'].shape, flush=True)
for t in range(20):
    env.step(np.random.default_rng(t).integers(0, 7, env.num_envs, dtype=np.uint8))
print('steps ok', flush=True)
env.close()
"""
for i in ids:
    for n in (64, 2048):
        r = subprocess.run([sys.executable, "-c", code, i, str(n)], capture_output=True, text=True)
        print(i, n, "rc", r.returncode, "|", r.stdout.replace("\n", " ; ")[-200:], "|", r.stderr[-600:].replace("\n", " ; "), flush=True)

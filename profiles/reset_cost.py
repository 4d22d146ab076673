import time, sys
sys.path.insert(0, "/root/repo")
import torch
import minigrid_amd as mg
def run(env_id, n, **kw):
    env = mg.make_vec(env_id, n, output="torch", **kw)
    env.reset(seed=0)
    env.rollout(512, action_seed=1, fused=True); env.sync()
    c0 = env.counters()["episodes"]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    env.rollout(2048, action_seed=2, fused=True); env.sync()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    c1 = env.counters()["episodes"]
    print("%-28s n %6d %s: %.3f us/step, %.2f G env-steps/s, episode ends per env-step %.5f" % (env_id, n, kw, dt / 2048 * 1e6, n * 2048 / dt / 1e9, (c1 - c0) / (n * 2048.0)), flush=True)
    env.close()
for n in (32768, 65536):
    run("BabyAI-GoToRedBall-v0", n)
    run("BabyAI-GoToRedBall-v0", n, max_steps=512)
    run("BabyAI-GoToRedBall-v0", n, max_steps=4096)
    run("MiniGrid-Empty-8x8-v0", n)
    run("MiniGrid-Empty-8x8-v0", n, max_steps=32)

import sys, numpy as np
sys.path.insert(0, '.')
import minigrid_amd as mg
ids = ["BabyAI-BossLevel-v0", "BabyAI-SynthSeq-v0", "BabyAI-MiniBossLevel-v0", "BabyAI-OpenDoorsOrderN4-v0", "BabyAI-MoveTwoAcrossS8N9-v0",
       "BabyAI-PutNextS7N4Carrying-v0", "BabyAI-KeyInBox-v0", "BabyAI-GoToObjDoor-v0", "BabyAI-UnblockPickup-v0", "MiniGrid-ObstructedMaze-Full-v1"]
for i in ids:
    for rng in ("philox", "pcg64"):
        env = mg.make_vec(i, 1024, rng=rng)
        obs, _ = env.reset(seed=7)
        r = np.random.default_rng(0)
        nd = 0
        for t in range(120):
            obs, rew, term, trunc, _ = env.step(r.integers(0, 7, 1024, dtype=np.uint8))
            nd += int(term.sum() + trunc.sum())
        a = mg.make_vec(i, 1024, rng=rng); o2, _ = a.reset(seed=7)
        same = (o2["image"] == env.reset(seed=7)[0]["image"]).all()
        print(i, rng, "ok, finished", nd, "deterministic reset:", bool(same), "| mission[0]:", str(obs["mission"][0])[:60], flush=True)
        env.close(); a.close()
import __graft_entry__ as g
g.smoke(); print("smoke ok")

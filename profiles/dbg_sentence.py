import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import minigrid_amd as mg
from conftest import golden
for env_id in sys.argv[1:]:
    g = golden(f"gen_{env_id}.npz")
    n, episodes = g["grid"].shape[:2]
    try:
        env = mg.make_vec(env_id, n)
    except Exception as ex:
        print(env_id, "create failed:", ex); continue
    for ep in range(episodes):
        try:
            obs, info = env.reset(seed=list(range(n))) if ep == 0 else env.reset()
        except Exception as ex:
            print(env_id, ep, "reset failed:", ex); break
        grid, agent = env.get_state()
        badg = np.argwhere((grid != g["grid"][:, ep]).reshape(n, -1).any(1)).ravel()
        bada = np.argwhere((agent[:, :6] != g["agent"][:, ep, :6]).any(1)).ravel()
        badm = np.argwhere(obs["mission"] != g["mission_str"][:, ep]).ravel()
        print(env_id, "ep", ep, "grid mismatches", badg[:10], "agent", bada[:10], "mission", badm[:10])
        for i in badm[:6]:
            print("   env", i, "| dev:", obs["mission"][i], "| ref:", g["mission_str"][i, ep])
    env.close()

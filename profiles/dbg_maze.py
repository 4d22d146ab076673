import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minigrid_amd as mg
def show(grid, agent):
    W, H = grid.shape[:2]
    ch = {1: ".", 2: "#", 4: "D", 5: "k", 6: "o", 7: "b"}
    rows = []
    for y in range(H):
        r = ""
        for x in range(W):
            t, c, s = grid[x, y]
            r += ("A" if (x, y) == (agent[0], agent[1]) else ch.get(int(t), "?")) + (str(int(c)) if t in (4, 5, 6, 7) else " ")
        rows.append(r)
    return "\n".join(rows)
for env_id in ("BabyAI-GoToObjMazeS4R2-v0", "BabyAI-GoToObjMazeS4-v0", "BabyAI-GoTo-v0"):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"gen_{env_id}.npz"))
    n = g["grid"].shape[0]
    env = mg.make_vec(env_id, n)
    env.reset(seed=list(range(n)))
    grid, agent = env.get_state()
    bad = np.argwhere((grid != g["grid"][:, 0]).reshape(n, -1).any(1)).ravel()
    print(env_id, "mismatching seeds", bad[:20], "of", n, "agent mismatch", np.argwhere((agent[:, :3] != g["agent"][:, 0, :3]).any(1)).ravel()[:20])
    for s in list(bad[:2]):
        print("seed", s, "device agent", agent[s][:3], "golden agent", g["agent"][s, 0, :3])
        print(show(grid[s], agent[s])); print("--- golden"); print(show(g["grid"][s, 0], g["agent"][s, 0]))
    env.close()

"""How long reset(seed) + the ring fill behind it takes (direct generation: k_generate_lane, one lane per env, against k_generate, one wavefront per episode --
MG_LANE_DIRECT=0).  Usage: python profiles/reset_latency_lanes.py <env id> <n>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import minigrid_amd as mg
env_id, n = sys.argv[1], int(sys.argv[2])
env = mg.make_vec(env_id, n)
env.reset(seed=0); env.sync()
t0 = time.perf_counter(); env.reset(seed=1); t1 = time.perf_counter(); env.sync(); t2 = time.perf_counter()
print(f"{env_id} x {n} MG_LANE_DIRECT={os.environ.get('MG_LANE_DIRECT', 'default')}: reset(seed) returns after {1e3 * (t1 - t0):.1f} ms, ring full after {1e3 * (t2 - t0):.1f} ms")
env.close()

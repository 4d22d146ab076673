"""tuning aid: k_generate latency for a small batch (every wave busy with exactly one episode)
   rocprofv3 --kernel-trace --stats --output-format csv -- python profiles/gen_latency.py <env_id> <n>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import minigrid_amd as mg
env_id, n = sys.argv[1], int(sys.argv[2])
env = mg.make_vec(env_id, n)
for k in range(30):
    env.reset(seed=list(range(k * n, (k + 1) * n)))
env.close()

"""N > 1 ranks on real hardware: ShardedVecEnv over `nccl` (RCCL).  Skipped unless the box shows >= 2 GPUs -- the
1-GPU boxes of the round cannot run it; the driver's multi-GPU node can (`pytest -m gpu`).  The same sharding / record
gather logic runs under gloo on CPU in tests/test_sharded_gloo.py."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, env_id, n, full):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from minigrid_amd.sharded import ShardedVecEnv, record_layout
        from oracle import oracle as O
        env = ShardedVecEnv(env_id, n, gather=True, obs_mode="full" if full else "partial", device=rank)
        assert dist.get_world_size() == world and env.local.device == rank
        lay = env.local.record_layout()
        want = record_layout(env.local_num_envs, int(np.prod(env.local.image_shape)))
        assert all(lay[k] == want[k] for k in want), (lay, want)
        orc = O.OracleVec(env_id, n, full_obs=full)              # every rank checks the WHOLE gathered batch
        obs, _ = env.reset(seed=5)
        o_obs, _, _ = orc.reset(seeds=np.arange(5, 5 + n, dtype=np.uint64))
        assert (obs["image"].cpu().numpy() == o_obs).all()
        rng = np.random.default_rng(3)
        c0 = env.collectives
        for t in range(60):
            a = rng.integers(0, 7, n, dtype=np.uint8)
            obs, rew, term, trunc, _ = env.step(a)
            oo, orew, oterm, otrunc, od, om = orc.step(a)
            assert (obs["image"].cpu().numpy() == oo).all(), (rank, t)
            assert rew.cpu().numpy().tobytes() == orew.tobytes() and (term.cpu().numpy() == oterm).all()
            assert (trunc.cpu().numpy() == otrunc).all() and (obs["direction"].cpu().numpy() == od).all()
        assert env.collectives - c0 == 60                         # exactly one all-gather per step
        env.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("env_id,n,full", [("MiniGrid-DoorKey-8x8-v0", 4096, False), ("MiniGrid-LavaCrossingS9N1-v0", 3001, True),
                                            ("BabyAI-GoToRedBall-v0", 2048, False)])
def test_sharded_env_over_rccl_matches_the_oracle(env_id, n, full):
    import torch
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    if world < 2:
        pytest.skip("needs >= 2 visible GPUs (RCCL all-gather of the step record)")
    mp.spawn(_worker, args=(world, _free_port(), env_id, n, full), nprocs=world, join=True)

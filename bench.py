#!/usr/bin/env python3
"""bench.py — env-steps/s of the MI355X-native MiniGrid hot path under a uniform-random policy.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one lockstep pass of MiniGridEnv.step()+gen_obs() over the whole batch (BASELINE.json configs[1]:
MiniGrid-Empty-8x8-v0, 65 536 envs per GPU, 7x7x3 partial obs), actions drawn on the device (Philox4x32-10), every
step writing its full outputs (obs u8 (N,7,7,3), reward f64, terminated, truncated, direction, mission id) to HBM,
NEXT_STEP autoreset inside the timed region.  Env state and all buffers are resident in HBM before timing starts.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     — dominant kernel (k_roll7; k_step for the other observation modes): ALGORITHMIC bytes per launch (324 B/env-step partial obs,
                 W*H*3*2+30 for FullyObs) / average launch period measured with HIP events on the launch stream.
  cpu_baseline — the oracle's C port (oracle/minigrid_oracle.c) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (env id, envs per GPU, obs mode)
    "empty8x8": ("MiniGrid-Empty-8x8-v0", 65536, "partial"),            # BASELINE.json configs[1]  (headline)
    "doorkey8x8": ("MiniGrid-DoorKey-8x8-v0", 262144, "partial"),       # configs[2]
    "lavacrossing_full": ("MiniGrid-LavaCrossingS9N1-v0", 131072, "full"),  # configs[3], per-GPU shard of 1 048 576
    "gotoredball": ("BabyAI-GoToRedBall-v0", 32768, "partial"),         # configs[4], per-GPU shard of 262 144
    # SURVEY.md §8(f) rank 4 -- what the stock benchmark.py times: RGBImgObsWrapper / RGBImgPartialObsWrapper frames
    "empty8x8_rgb": ("MiniGrid-Empty-8x8-v0", 65536, "rgb"),            # 64x64x3 frame per env-step (805 MB per step)
    "doorkey8x8_rgb_partial": ("MiniGrid-DoorKey-8x8-v0", 65536, "rgb_partial"),   # 56x56x3 agent-POV frame
    # step() itself draws (the obstacles move on the env's stream): live generation + k_move_obstacles + k_step per step
    "dynobs16x16": ("MiniGrid-Dynamic-Obstacles-16x16-v0", 65536, "partial"),
    # the sentence levels (instruction trees; the verifier runs inside the fused step loop since round 3)
    "bosslevel": ("BabyAI-BossLevel-v0", 131072, "partial"),
}


def algorithmic_bytes_per_env_step(env_id: str, obs_mode: str, W: int, H: int, view: int = 7) -> int:
    """action 1 + grid read (49 view cells or W*H cells) x 3 B + agent record r/w 8+8 +
    cell write-back 3 + image out + reward 8 + terminated 1 + truncated 1 (+ direction 1 + mission id 1 for BabyAI)."""
    cells = W * H if obs_mode in ("full", "symbolic", "rgb") else view * view
    out_per_cell = 20 if obs_mode == "onehot" else 3
    if obs_mode in ("rgb", "rgb_partial"):
        # k_step reads the view (and, for the full frame, the whole grid) and writes a 1 B/cell tile map; k_render reads
        # it back (+ the agent record for the full frame) and writes tile_size^2 x 3 bytes per cell (tile_size = 8)
        read_cells = view * view + (W * H if obs_mode == "rgb" else 0)
        out_per_cell = 1 + 1 + 8 * 8 * 3
        return 1 + read_cells * 3 + 8 + 8 + 3 + cells * out_per_cell + (8 if obs_mode == "rgb" else 0) + 8 + 1 + 1
    b = 1 + cells * 3 + 8 + 8 + 3 + cells * out_per_cell + 8 + 1 + 1
    if env_id.startswith("BabyAI"):
        b += 2
    return b


def _profile_dirs(workload: str, n_per_gpu: int, spl: int):
    """profiles/r*/ directories whose committed rocprofv3 passes of `workload` were taken at THIS batch size and THIS steps-per-
    launch (profiles/collect.sh writes the run's parameters to meta_<workload>.json next to the counters)."""
    import glob
    out = []
    for d in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*"))):
        try:
            meta = json.load(open(os.path.join(d, f"meta_{workload}.json")))
        except Exception:
            continue
        if int(meta.get("envs_per_gpu", -1)) == n_per_gpu and int(meta.get("steps_per_launch", -1)) == spl:
            out.append(d)
    return out


def pmc_traffic_bytes(workload: str, n_per_gpu: int, spl: int):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes of this same command
    (profiles/<round>/pmc_{FETCH,WRITE}_SIZE_<workload>.txt, separate --pmc runs, written by profiles/collect.sh).
    Units and gfx950 correction as MI355X_MICROARCH.md prescribes: the counters are in KiB (x1024); FETCH_SIZE reads
    exactly half of a wide (16 B/lane) coalesced read stream on gfx950, so it is doubled; WRITE_SIZE is taken as is.
    Counters cannot be read from inside the timed process, so this is the committed measurement -- of a run with the same
    batch size and steps per launch (the per-call maximum = the full launches) -- or None."""
    import re
    best = None
    for d in _profile_dirs(workload, n_per_gpu, spl):
        vals = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            f = os.path.join(d, f"pmc_{c}_{workload}.txt")
            if not os.path.exists(f):
                break
            for line in open(f):
                m = re.match(rf"{c},(?:void )?mg::(k_\w+<[^>]*>|k_render),calls=\d+,mean=([0-9.]+)(?:,total=[0-9.]+)?(?:,max=([0-9.]+))?", line)
                if m and (m.group(1).startswith("k_step") or m.group(1).startswith("k_roll") or m.group(1) == "k_render"):
                    # RGB workloads: k_step + k_render make one step.  A run mixes full launches with one-step reset observations:
                    # the per-call maximum is the full launch when the summary has it
                    vals[c] = vals.get(c, 0.0) + float(m.group(3) or m.group(2))
        if len(vals) == 2:
            best = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    return best


def rocprof_kernel_us_per_step(workload: str, n_per_gpu: int, spl: int):
    """Average duration of a FULL step launch / steps per launch from the committed `rocprofv3 --kernel-trace --stats` summary of
    this command (profiles/<round>/kernel_stats_<workload>.csv; meta_<workload>.json holds the full-launch average computed
    from the trace by profiles/collect.sh), or None."""
    best = None
    for d in _profile_dirs(workload, n_per_gpu, spl):
        try:
            meta = json.load(open(os.path.join(d, f"meta_{workload}.json")))
            best = float(meta["full_launch_avg_us"]) / spl
        except Exception:
            pass
    return best


def reference_python_baseline(workload: str):
    """The reference's own CPU path (unmodified /root/reference, minigrid/benchmark.py:32-43 plumbing with random actions),
    timed by profiles/ref_python_baseline.py in the build container -- /root/reference does not exist on the GPU box, so
    this is the committed measurement of that script, quoted with its hardware and source file."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "reference_python_baseline.json"))):
        try:
            d = json.load(open(f))
            w = d["workloads"].get(workload)
            if w:
                best = {"one_core": w["one_core"]["value"], "all_cores": w["all_cores"]["value"], "cores": w["all_cores"]["cores"],
                        "unit": "env-steps/s", "wrapper": w["wrapper"], "hardware": d["hardware"],
                        "source": os.path.relpath(f, ROOT) + "@" + d.get("git_head", "?"),
                        "how": "profiles/ref_python_baseline.py: unmodified reference via oracle/gym_shim, BASELINE.md section 3 loop"}
        except Exception:
            pass
    return best


def pmc_traffic_source(workload: str, n_per_gpu: int, spl: int):
    src = None
    for d in _profile_dirs(workload, n_per_gpu, spl):
        if all(os.path.exists(os.path.join(d, f"pmc_{c}_{workload}.txt")) for c in ("FETCH_SIZE", "WRITE_SIZE")):
            src = os.path.relpath(d, ROOT) + f"/pmc_{{FETCH,WRITE}}_SIZE_{workload}.txt"
    return src


def cpu_baseline_rgb(env_id: str, obs_mode: str, budget_s: float = 10.0):
    """RGB workloads: the oracle's C port steps, oracle/render.py (numpy) draws every frame; one batch per thread."""
    from concurrent.futures import ThreadPoolExecutor

    import numpy as np

    from oracle import oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    n_per = 256
    vecs = [O.OracleVec(env_id, n_per, obs=obs_mode) for _ in range(cores)]
    for i, v in enumerate(vecs):
        v.reset(seeds=range(i * n_per, (i + 1) * n_per))

    def work(args):
        v, T, seed = args
        rng = np.random.default_rng(seed)
        for _ in range(T):
            v.step(rng.integers(0, 7, n_per, dtype=np.uint8))

    def timed(T):
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.perf_counter()
            list(ex.map(work, [(v, T, i) for i, v in enumerate(vecs)]))
            return time.perf_counter() - t0

    timed(2)
    T, dt, chunk = 0, 0.0, 20
    while dt < budget_s and T < 100_000:
        dt += timed(chunk)
        T += chunk
    return {"value": cores * n_per * T / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {n_per} envs x {T} steps of {env_id} ({obs_mode} frames), oracle C port + "
                      f"numpy tile mosaic (oracle/render.py), random actions, NEXT_STEP autoreset, {dt:.1f}s"}


def cpu_baseline(env_id: str, obs_mode: str, budget_s: float = 10.0):
    """Time the oracle's C port on the host cores this process may use (one independent batch per thread; ctypes
    drops the GIL).  Bounded: a short all-thread calibration sizes the sample to ~budget_s seconds."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    n_per = 512
    vecs = [O.OracleVec(env_id, n_per, full_obs=(obs_mode == "full")) for _ in range(cores)]
    for i, v in enumerate(vecs):
        v.reset(seeds=range(i * n_per, (i + 1) * n_per))

    def timed(T):
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda v: v.rollout(T, 7), vecs))
            return time.perf_counter() - t0

    timed(50)                                     # warm the thread pool / page in the library
    T, dt, chunk = 0, 0.0, 1000                   # fixed-size chunks until the sample is ~budget_s long (bounded)
    while dt < budget_s and T < 5_000_000:
        dt += timed(chunk)
        T += chunk
    return {"value": cores * n_per * T / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {n_per} envs x {T} steps of {env_id} ({obs_mode} obs), oracle C port "
                      f"(oracle/minigrid_oracle.c), xorshift random actions, NEXT_STEP autoreset, {dt:.1f}s"}


def np_prod(shape):
    p = 1
    for v in shape:
        p *= int(v)
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--workload", default="empty8x8", choices=sorted(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1, help="1: the fused rollout kernel (up to max_fused_steps steps per k_step launch, "
                    "grids resident in LDS, every step's outputs to its own trajectory slot); 0: one k_step launch per step")
    ap.add_argument("--gather-obs", type=int, default=0, help="RCCL all-gather the obs tensor every step")
    ap.add_argument("--obs-mode", default="", help="override the workload's obs mode: partial|full|onehot|symbolic|rgb|rgb_partial")
    ap.add_argument("--view", type=int, default=7, help="agent_view_size (ViewSizeWrapper) for partial/onehot")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets the multi-process path be exercised "
                         "on a box with fewer GPUs than ranks: ranks then share devices round-robin)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import minigrid_amd as mg

    env_id, n_per_gpu, obs_mode = WORKLOADS[args.workload]
    if args.envs_per_gpu:
        n_per_gpu = args.envs_per_gpu
    if args.obs_mode:
        obs_mode = args.obs_mode
    gather = bool(args.gather_obs and world > 1)
    if world > 1:
        # weak scaling: the global batch is world x n_per_gpu envs; rank g owns the contiguous block g (seed = global
        # env index), no data-path collective unless --gather-obs asks for the optional all-gather of the obs tensor
        from minigrid_amd.sharded import ShardedVecEnv
        senv = ShardedVecEnv(env_id, n_per_gpu * world, gather=gather, obs_mode=obs_mode, device=local_rank,
                             agent_view_size=args.view)
        env = senv.local
        assert env.env_index_base == rank * n_per_gpu and env.num_envs == n_per_gpu
    else:
        senv = None
        env = mg.make_vec(env_id, n_per_gpu, obs_mode=obs_mode, device=local_rank, output="torch", agent_view_size=args.view)
    fused = bool(args.fused)
    spl = min(env.max_fused_steps, args.steps) if fused else 1          # steps per k_step launch in the timed region
    env.reset(seed=0)
    env.sync()

    def run(k, seed):
        if not gather:
            env.rollout(k, action_seed=seed, fused=fused)
        elif fused:
            # ONE all_gather_into_tensor per fused launch (its max_fused_steps step records = one contiguous block of the trajectory
            # ring), on a communication stream ordered by events: launch k + 1 runs under the gather of launch k
            senv.rollout_gather(k, action_seed=seed)
        else:
            with senv._on_step_stream():
                for _ in range(k):
                    env.rollout(1, action_seed=seed)
                    senv.gather_record()         # ONE all_gather_into_tensor of the step record, stream-ordered (no host sync)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    env.timer_start()                    # (the warm-up also warms the event pair the timed region uses)
    run(args.warmup, 1)
    env.timer_stop()
    env.sync()
    barrier()
    env.timer_start()                    # HIP event on the step stream ...
    t0 = time.perf_counter()             # ... and the host clock, both opened before the first launch is enqueued
    run(args.steps, 2)
    ev_ms = env.timer_stop()             # event after the last launch on the same stream (waits for it): the launches' own time
    env.sync()                           # + the generator stream: every episode consumed in the region is drawn again
    if gather:
        senv.finish()                    # + the communication stream: every collective issued in the region has completed
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0        # THIS rank's clock (>= the event time): its K steps are complete on the device
    if world > 1:
        dist.barrier()                   # the closing barrier of the bracket; the job's time is the MAX over the ranks' clocks (below) --
                                         # the ranks started together, so that is when the slowest one finished; the latency of the
                                         # barrier collective itself (tens of us over 8 GPUs) is not part of the K steps

    per_rank_us = [dt / args.steps * 1e6]
    if world > 1:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        allt = torch.zeros(world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, t)                       # every rank's own clock ...
        per_rank_us = [float(x) / args.steps * 1e6 for x in allt.cpu()]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                   # ... and the job's: the slowest rank
        dt = float(t.item())
    counters = env.counters()
    # the step kernel this configuration runs (mg_api.hip: k_roll7 for the default 7x7 view and for FullyObs of grids up to 341 cells)
    kname = ("k_roll7" if (obs_mode == "partial" and args.view == 7) else
             "k_roll7<., FullyObs>" if (obs_mode == "full" and env.width * env.height <= 341) else "k_step")

    if rank == 0:
        total_envs = n_per_gpu * world
        value = total_envs * args.steps / dt
        bpe = algorithmic_bytes_per_env_step(env_id, obs_mode, env.width, env.height, args.view)
        n_launch = -(-args.steps // spl)
        launch_s = (ev_ms / 1e3) / n_launch             # average k_step launch period on its stream (HIP events)
        step_s = (ev_ms / 1e3) / args.steps
        bytes_per_launch = bpe * n_per_gpu * args.steps / n_launch      # (the last launch of the region may be shorter)
        achieved = bytes_per_launch / launch_s / 1e9    # algorithmic bytes per launch / average launch duration
        obe = int(np_prod(env.image_shape))
        # what a fused launch has to move per env-step: the outputs (obs + reward 8 + 5 flag/id bytes); the grid and the
        # agent record are read and written once per launch, not per step
        hbm_min = obe + 13 + (2 * (env.width * env.height) + 16) / spl
        out = {
            "metric": "env-steps/s (random policy)", "value": value, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "host_ms": dt * 1e3, "event_ms": ev_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{env_id}, {n_per_gpu} envs/GPU x {world} GPU, {obs_mode} obs "
                                   f"{'x'.join(map(str, env.image_shape))}, device Philox random actions, NEXT_STEP autoreset",
                       "env_id": env_id, "envs_per_gpu": n_per_gpu, "obs_mode": obs_mode,
                       "launch": (f"fused: {spl} steps per {kname} launch, state resident in LDS, each step's outputs to its own trajectory slot"
                                  if spl > 1 else f"one {kname} launch per step") + (" + one k_render" if obs_mode.startswith("rgb") else ""),
                       "steps_per_launch": spl,
                       "gather_obs": gather, "episodes_finished_rank0": counters["episodes"],
                       "distributed": {"world_size": (dist.get_world_size() if world > 1 else 1),
                                       "backend": (dist.get_backend() if world > 1 else None),
                                       "per_rank_us_per_step": per_rank_us,
                                       "collective": (("one all_gather_into_tensor per fused launch (its %d step records, one contiguous block), on a "
                                                       "communication stream overlapped with the next launch" % spl) if gather and fused else
                                                      "one all_gather_into_tensor of the step record per step" if gather else "none on the data path"),
                                       "collectives_rank0": (senv.collectives if senv is not None else 0)}},
            "roofline": {"bound": "hbm", "kernel": "k_step + k_render (one step)" if obs_mode.startswith("rgb") else kname, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic_bytes(args.workload, n_per_gpu, spl) if not args.obs_mode and args.view == 7 else None,
                         "traffic_unit": "bytes per step-kernel launch (rocprofv3 PMC of the same command, committed under profiles/; not measured in this run)",
                         "traffic_source": pmc_traffic_source(args.workload, n_per_gpu, spl),
                         "launches": n_launch, "algorithmic_bytes_per_launch": bytes_per_launch,
                         "algorithmic_bytes_per_env_step": bpe, "avg_launch_us": launch_s * 1e6, "avg_step_us": step_s * 1e6,
                         "kernel_us_per_step": rocprof_kernel_us_per_step(args.workload, n_per_gpu, spl) if not args.obs_mode and args.view == 7 else None,
                         "hbm_bytes_per_env_step_this_kernel": hbm_min,
                         "frac_of_peak_on_actual_bytes": n_per_gpu * hbm_min / step_s / 1e9 / HBM_PEAK_GBPS,
                         "note": "achieved/frac price the SURVEY 8(d) algorithmic bytes (the reference's 3 B/cell grid re-read "
                                 "every step); the fused kernel keeps the grid in LDS, so its real HBM traffic per env-step is "
                                 "hbm_bytes_per_env_step_this_kernel and the bound it actually runs against is the HBM WRITE stream"},
        }
        if not args.no_cpu_baseline and world == 1:
            if obs_mode.startswith("rgb"):
                out["cpu_baseline"] = cpu_baseline_rgb(env_id, obs_mode)
            else:
                out["cpu_baseline"] = cpu_baseline(env_id, obs_mode if obs_mode in ("partial", "full") else "partial")
            out["cpu_baseline"]["reference_python"] = reference_python_baseline(args.workload)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

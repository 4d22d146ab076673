#!/bin/bash
# two ranks on ONE MI355X over gloo (the box has one GPU: RCCL needs one device per rank): bench.py spawns its ranks itself
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r4ad; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1]); c=d['config']['distributed']; print('$1 n_gpus %d world %s backend %s  %.3f G env-steps/s  %.2f us/step  per-rank %s  collective: %s (%d on rank 0)' % (d['n_gpus'], c['world_size'], c['backend'], d['value']/1e9, d['ms_per_step']*1e3, ['%.2f' % x for x in c['per_rank_us_per_step']], c['collective'][:60], c['collectives_rank0']))"; }
timeout 300 python bench.py --gpus 2 --backend gloo --envs-per-gpu 32768 --steps 512 --warmup 64 2>/dev/null | line "no collective      "
timeout 300 python bench.py --gpus 2 --backend gloo --envs-per-gpu 8192 --steps 128 --warmup 32 --gather-obs 1 2>/dev/null | line "fused block gather "
timeout 300 python bench.py --gpus 2 --backend gloo --envs-per-gpu 8192 --steps 64 --warmup 16 --gather-obs 1 --fused 0 2>/dev/null | line "per-step gather    "
timeout 300 python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 2>/dev/null | line "driver-sized x 2    "
timeout 600 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -3

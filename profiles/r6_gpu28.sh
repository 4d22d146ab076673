#!/bin/bash
# Round 6, call 28: occupancy scan -- resident workgroups per CU and rounds per launch of the step kernel, per workload (profiles/occupancy_scan.py)
export TMPDIR=/tmp
ROOT=$PWD; OUT=$ROOT/gpurun_out/r6; mkdir -p $OUT
cd /tmp
: > $OUT/occupancy_scan.txt
scan() {   # label, bench args
  local label=$1; shift
  rm -rf /tmp/occ; timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/occ -o t -- python $ROOT/bench.py "$@" --steps 128 --warmup 32 --no-cpu-baseline > /tmp/occ.log 2>&1
  python $ROOT/profiles/occupancy_scan.py $(find /tmp/occ -name '*kernel_trace.csv' | head -1) "$label" | tee -a $OUT/occupancy_scan.txt
}
for w in empty8x8 doorkey8x8 lavacrossing_full gotoredball dynobs16x16 dynobs8x8 dynobs6x6 keycorridor multiroom babyai_goto unlockpickup unlock blockedunlockpickup bosslevel; do scan $w --workload $w; done
for id in MiniGrid-Fetch-8x8-N3-v0 MiniGrid-GoToDoor-8x8-v0 MiniGrid-RedBlueDoors-8x8-v0 MiniGrid-MemoryS11-v0 MiniGrid-MemoryS17Random-v0 BabyAI-PickupDist-v0 BabyAI-PutNextLocal-v0 BabyAI-OpenRedDoor-v0 MiniGrid-LockedRoom-v0 \
          MiniGrid-FourRooms-v0 MiniGrid-LavaGapS7-v0 MiniGrid-DistShift1-v0 MiniGrid-SimpleCrossingS11N5-v0 MiniGrid-Empty-16x16-v0 MiniGrid-DoorKey-16x16-v0 MiniGrid-ObstructedMaze-Full-v0 MiniGrid-Playground-v0 \
          MiniGrid-PutNear-8x8-N3-v0 MiniGrid-GoToObject-8x8-N2-v0 BabyAI-GoToLocal-v0 BabyAI-UnlockLocal-v0 BabyAI-Pickup-v0 BabyAI-Synth-v0 BabyAI-GoToSeq-v0 BabyAI-MiniBossLevel-v0 BabyAI-KeyCorridorS6R3-v0; do
  scan "$id x 131072" --workload keycorridor --env-id $id
done
scan "lavacrossing partial x 131072" --workload lavacrossing_full --obs-mode partial
scan "doorkey8x8 full x 262144" --workload doorkey8x8 --obs-mode full
scan "empty8x8 full x 65536" --workload empty8x8 --obs-mode full
scan "keycorridor full x 131072" --workload keycorridor --obs-mode full
scan "empty8x8 unfused" --workload empty8x8 --fused 0

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2m; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "UnlockPickup or BlockedUnlockPickup or UnlockToUnlock or BabyAI-Unlock-v0 or GoToDoor or GoToObjDoor or GoToImpUnlock or UnblockPickup or PickupAbove" > $O/t_new.log 2>&1; echo "new rc=$?" | tee -a $O/summary.txt; tail -40 $O/t_new.log | cut -c1-400

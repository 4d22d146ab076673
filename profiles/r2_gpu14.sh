#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -k "PutNext or ActionObjDoor or BabyAI-OpenDoor" > $O/t_new.log 2>&1; echo "new rc=$?" | tee -a $O/summary.txt; tail -40 $O/t_new.log | cut -c1-600

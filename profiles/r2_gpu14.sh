#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2n; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -u -m pytest tests -m gpu -v -k "PutNext or ActionObjDoor or BabyAI-OpenDoor" > $O/t_new.log 2> $O/t_new.err; echo "new rc=$?" | tee -a $O/summary.txt; grep -v PASSED $O/t_new.log | tail -40 | cut -c1-600; tail -5 $O/t_new.err | cut -c1-300

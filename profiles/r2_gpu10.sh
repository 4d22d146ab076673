#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused.py -x -q -s -k "320" > $O/t_dbg.log 2>&1; echo "dbg rc=$?" | tee -a $O/summary.txt; grep -v "^  File" $O/t_dbg.log | head -30
timeout 1500 python -m pytest tests -m gpu -q -x > $O/t_all.log 2>&1; echo "all rc=$?" | tee -a $O/summary.txt; tail -4 $O/t_all.log

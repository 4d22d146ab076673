#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2s; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_generators_match_reference_goldens and (OpenTwoDoors or OpenRedBlue or OpenDoorsOrder or MoveTwoAcross or PickupLoc or GoToSeq or Synth or BossLevel)" > $O/t_gen.log 2>&1; echo "gen rc=$?" | tee -a $O/summary.txt; tail -30 $O/t_gen.log | cut -c1-500
timeout 1200 python -u -m pytest tests/test_gpu_parity.py -m gpu -q -k "OpenTwoDoors or OpenRedBlue or OpenDoorsOrder or MoveTwoAcross or PickupLoc or GoToSeq or Synth or BossLevel" > $O/t_new.log 2>&1; echo "new rc=$?" | tee -a $O/summary.txt; tail -45 $O/t_new.log | cut -c1-300

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2o; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -u -m pytest tests -m gpu -q -k "PutNext or ActionObjDoor or BabyAI-OpenDoor" > $O/t_new.log 2>&1; echo "new rc=$?" | tee -a $O/summary.txt; tail -5 $O/t_new.log | cut -c1-300
# hunt for the intermittent SIGABRT seen twice inside the library: many short create / reset / step / get_state / close cycles
# under the debugger; a normal exit prints nothing, an abort leaves the native backtrace of every thread
for i in 1 2 3 4 5 6; do
  timeout 170 rocgdb -batch -ex "handle SIGSEGV nostop noprint pass" -ex run -ex "thread apply all bt 25" --args python -u -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_generators_match_reference_goldens" > $O/gdb_$i.log 2>&1
  echo "gdb loop $i rc=$? $(grep -c 'SIGABRT\|Aborted' $O/gdb_$i.log) $(tail -1 $O/gdb_$i.log | cut -c1-120)" | tee -a $O/summary.txt
  if grep -q "SIGABRT" $O/gdb_$i.log; then break; fi
done

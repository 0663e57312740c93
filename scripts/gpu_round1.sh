#!/bin/bash
# One GPU call: parity tests, stream tuning sweep, bench.  Stops after any abnormal exit.
mkdir -p gpurun_out
run() {  # run <seconds> <log> <cmd...>
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "[$log] exit $rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal exit, stopping"; tail -30 "gpurun_out/$log"; exit $rc; fi
  return 0
}
run 600 test1.log python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread -x -k "not full_size"
tail -40 gpurun_out/test1.log
run 300 tune1.log ./tools/stream_tune
run 300 bench1.log python bench.py --steps 10 --warmup 2
tail -5 gpurun_out/bench1.log

#!/bin/bash
# One GPU call: parity tests, bench, rocprofv3 kernel trace + HBM counters.  Stops after any
# abnormal exit (anything but 0 / pytest's 1).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
run() {  # run <seconds> <log> <cmd...>
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$O/$log" 2>&1
  local rc=$?
  echo "[$log] exit $rc"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "abnormal exit, stopping"; tail -30 "$O/$log"; exit $rc; fi
  return 0
}
cd $R
run 900 test.log python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread -x
grep -E "PASSED|FAILED|ERROR|passed|failed" $O/test.log | tail -15
run 300 bench.log python bench.py --steps 20 --warmup 3
tail -1 $O/bench.log
if [ -n "$PROFILE" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof_trace $O/prof_fetch $O/prof_write
  run 300 prof_trace.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline
  run 200 prof_fetch.log timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline
  run 200 prof_write.log timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline
  find $O/prof_trace $O/prof_fetch $O/prof_write -type f | head -30
fi

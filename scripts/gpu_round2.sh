#!/bin/bash
# One GPU call: full parity suite, bench.py, rocprofv3 kernel trace + the two HBM counter passes of bench.py.
# Stops after any abnormal exit.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
run() {  # run <seconds> <log> <cmd...>
  local secs=$1 log=$2; shift 2
  timeout -k 10 "$secs" "$@" > "$O/$log" 2>&1
  local rc=$?
  echo "[$log] exit $rc"
  if [ $rc -ne 0 ]; then echo "stopping"; tail -30 "$O/$log"; exit $rc; fi
  return 0
}
cd $R
run 900 test.log python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread
tail -2 $O/test.log
run 300 bench.log python bench.py --steps 20 --warmup 3
tail -1 $O/bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_trace $O/prof_fetch $O/prof_write
run 300 prof_trace.log rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline
run 200 prof_fetch.log timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --shard-elems 100000000 --sub-elems 1000000
run 200 prof_write.log timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --shard-elems 100000000 --sub-elems 1000000
find $O/prof_trace $O/prof_fetch $O/prof_write -name "*.csv" | head -20

#!/bin/bash
# Full GPU check: all -m gpu tests, then the secondary config benches (2, 3, 4).  Stops at the first abnormal exit.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread -x > $O/test.log 2>&1
rc=$?; echo "[test] exit $rc"; tail -5 $O/test.log
if [ $rc -ne 0 ]; then grep -E "FAILED|Error|assert" $O/test.log | head -20; exit $rc; fi
for c in ${CONFIGS:-2 3 4}; do
  timeout -k 10 600 python bench_configs.py --config $c > $O/config$c.jsonl 2> $O/config$c.err
  rc=$?; echo "[config$c] exit $rc"; cut -c1-170 $O/config$c.jsonl
  if [ $rc -ne 0 ]; then tail -5 $O/config$c.err; exit $rc; fi
done

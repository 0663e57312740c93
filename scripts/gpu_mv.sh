#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_gpu_mv.py -m gpu -v --timeout 300 --timeout-method thread > $O/test_mv.log 2>&1
rc=$?; echo "[test_mv] exit $rc"; grep -E "PASSED|FAILED|ERROR|passed|failed|assert|Error|error" $O/test_mv.log | tail -40
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then tail -30 $O/test_mv.log; exit $rc; fi
timeout -k 10 300 python bench_configs.py --config 4 > $O/config4.jsonl 2> $O/config4.err
rc=$?; echo "[config4] exit $rc"; cat $O/config4.jsonl; tail -5 $O/config4.err

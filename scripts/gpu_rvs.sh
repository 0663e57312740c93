#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 600 python -u -m pytest tests/test_rvs.py -m gpu -v --timeout 300 --timeout-method thread -x > $O/test_rvs.log 2>&1
rc=$?; echo "[test_rvs] exit $rc"; grep -E "PASSED|FAILED|ERROR|passed|failed|assert|Error" $O/test_rvs.log | tail -30
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 600 python bench_configs.py --config 3 > $O/config3.jsonl 2> $O/config3.err
rc=$?; echo "[config3] exit $rc"; cat $O/config3.jsonl; tail -5 $O/config3.err
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python bench_configs.py --config 2 > $O/config2.jsonl 2> $O/config2.err
rc=$?; echo "[config2] exit $rc"; cat $O/config2.jsonl; tail -5 $O/config2.err

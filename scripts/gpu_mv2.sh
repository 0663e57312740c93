#!/bin/bash
# MV parity tests, config-4 timings and a rocprofv3 kernel trace of config 4 (kernel-level durations).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_mv.py tests/test_golden_mv.py -m gpu -q -x --timeout 120 --timeout-method thread > $O/r02_mv_test.log 2>&1 || { echo "tests failed"; tail -30 $O/r02_mv_test.log; exit 1; }
tail -2 $O/r02_mv_test.log
timeout -k 10 300 python bench_configs.py --config 4 > $O/r02_config4.jsonl 2> $O/r02_config4.err || { echo "bench failed"; tail -20 $O/r02_config4.err; exit 1; }
cut -c1-330 $O/r02_config4.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_mv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mv -- python3 $R/bench_configs.py --config 4 > $O/r02_prof_mv.log 2>&1 || { echo "prof failed"; tail -20 $O/r02_prof_mv.log; exit 1; }
f=$(find $O/prof_mv -name "*kernel_stats.csv" | head -1)
grep -E "Name|k_cholesky|k_solve" $f | cut -c1-260

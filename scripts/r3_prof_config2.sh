#!/bin/bash
# rocprofv3 kernel trace of bench_configs.py --config <cfg> (round 3): per-kernel durations -> gpurun_out/prof_c<cfg>_<tag>_kernel_stats.csv
CFG=${1:-2}; TAG=${2:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_c${CFG}_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c${CFG}_$TAG -- python3 $R/bench_configs.py --config $CFG > $O/prof_c${CFG}_$TAG.jsonl 2> $O/prof_c${CFG}_$TAG.err || { tail -5 $O/prof_c${CFG}_$TAG.err; exit 1; }
f=$(find $O/prof_c${CFG}_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $O/prof_c${CFG}_${TAG}_kernel_stats.csv

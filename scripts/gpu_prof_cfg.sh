#!/bin/bash
# rocprofv3 kernel trace of one bench_configs.py config: usage gpu_prof_cfg.sh <config> <tag> [n]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
CFG=$1; TAG=$2; N=${3:-0}
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$TAG -- python3 $R/bench_configs.py --config $CFG --n $N > $O/prof_$TAG.log 2>&1
rc=$?; echo "[prof_$TAG] exit $rc"; tail -5 $O/prof_$TAG.log
f=$(find $O/prof_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -30
exit $rc

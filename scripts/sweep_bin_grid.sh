#!/bin/bash
# A/B of the binned kernel's grid cap (M355_BIN_GRID) on config 2's special-function rows.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for g in 32768 16384 8192 4096 2048; do
  echo "grid $g"
  M355_BIN_GRID=$g timeout -k 10 200 python $R/bench_configs.py --config 2 --n 100000000 2>/dev/null > $R/gpurun_out/sweep_$g.log || exit 1
  grep "logcdf\|ppf" $R/gpurun_out/sweep_$g.log | grep -v normal | sed 's/.*"fn": "\([a-z.]*\)".*"ms": \([0-9.]*\),.*/\1=\2/' | tr '\n' ' '
  echo
done

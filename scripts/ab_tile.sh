#!/bin/bash
# A/B of two builds of the library on config 2 (same box, back to back): copies each over the in-tree library of this
# scratch copy of the repository.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  cp $R/tools/ab/$v $R/distributions_amd/libm355dist.so || exit 1
  echo "== $v"
  timeout -k 10 200 python $R/bench_configs.py --config 2 --n 100000000 2>/dev/null > $R/gpurun_out/ab_$v.log || exit 1
  grep "scalar params\"" $R/gpurun_out/ab_$v.log | grep "logcdf\|logsf\|ppf" | sed 's/.*"fn": "\([a-z.]*\)".*"ms": \([0-9.]*\),.*/\1=\2/' | awk -F= '{printf "%s=%.2f ", $1, $2}'
  echo
done

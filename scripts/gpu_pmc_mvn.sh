#!/bin/bash
# PMC passes for the multivariate kernels of bench_configs.py --config 4 (one rocprofv3 --pmc run per pass).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rm -rf $O/pmc_$tag
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench_configs.py --config 4 > $O/pmc_$tag.log 2>&1
  local rc=$?; echo "[pmc_$tag] exit $rc"
  if [ $rc -ne 0 ]; then tail -5 $O/pmc_$tag.log; exit $rc; fi
}
pass mvn_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE
pass mvn_b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS
python3 - <<'PY'
import csv, glob, collections, os
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out'
for tag in ('mvn_a', 'mvn_b'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f'{O}/pmc_{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:60]
            agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        if 'mvn' in k or 'chol' in k or 'solve' in k:
            print(tag, k, {c: x / max(1, cnt[(k, c)]) for c, x in v.items()})
PY

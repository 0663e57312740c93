#!/bin/bash
# Round 3: PMC passes over tools/shared_probe (k_mvn_shared alone: d = 64, n = $1), one rocprofv3 --pmc run per pass.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
N=${1:-1000000}
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rm -rf $O/pmcs_$tag
  timeout -s KILL 60 rocprofv3 --pmc "$@" --output-format csv -d $O/pmcs_$tag -- $R/tools/shared_probe 64 $N > $O/pmcs_$tag.log 2>&1
  local rc=$?; echo "[pmcs_$tag] exit $rc"
  if [ $rc -ne 0 ]; then tail -5 $O/pmcs_$tag.log; exit $rc; fi
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE &&
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS &&
pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS TA_BUSY_avr TA_BUSY_max
python3 - "$N" <<'PY'
import csv, glob, collections, os, json, sys
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out'
out = {}
for tag in ('a', 'b', 'c'):
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{O}/pmcs_{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_mvn_shared' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in agg.items():
        out[c] = sum(v) / len(v)
gui = out.get('GRBM_GUI_ACTIVE', 0) / 8.0   # summed over 8 XCDs
d = {'n': int(sys.argv[1]), 'd': 64, 'counters_avg_per_launch': out}
if gui:
    d['mfma_busy'] = out.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui) * 4   # per SIMD: counter sums cycles over SIMDs in units of 4
    d['gui_cycles'] = gui
json.dump(d, open(f'{O}/r03_shared_pmc_{sys.argv[1]}.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(d, indent=1, sort_keys=True))
PY

#!/bin/bash
# Round 3: PMC passes for the multivariate kernels of bench_configs.py --config 4 (one rocprofv3 --pmc run per pass);
# summary -> gpurun_out/r03_config4_pmc.json (copied to profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {
  local tag=$1; shift
  rm -rf $O/pmc4_$tag
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc4_$tag -- python3 $R/bench_configs.py --config 4 > $O/pmc4_$tag.log 2>&1
  local rc=$?; echo "[pmc4_$tag] exit $rc"
  if [ $rc -ne 0 ]; then tail -5 $O/pmc4_$tag.log; exit $rc; fi
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE &&
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE &&
pass c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections, os, json, re
O = os.environ.get('GRAFT_REPO_ROOT', '.') + '/gpurun_out'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('a', 'b', 'c'):
    for f in glob.glob(f'{O}/pmc4_{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if re.search(r'k_mvn|k_chol|k_solve', k):
                # launches of one kernel differ in grid (n = 1e5 / 1e6): keep them apart
                key = re.sub(r'\(.*', '', k).replace('void m355::', '') + ' grid=' + r.get('Grid_Size', '?')
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    gui = m.get('GRBM_GUI_ACTIVE', 0) / 8.0   # the counter sums the 8 XCDs
    e = {'launches': max(len(v) for v in c.values()), 'counters_avg_per_launch': m}
    if gui:
        e['gui_cycles'] = gui
        e['mfma_busy'] = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * gui)     # busy cycles summed over 1024 SIMDs
        e['valu_busy'] = 4 * m.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * gui)      # (quad-cycle units)
        e['wave_wait_frac'] = m.get('SQ_WAIT_INST_ANY', 0) / max(1.0, m.get('SQ_WAVE_CYCLES', 0))
    out[k] = e
json.dump(out, open(f'{O}/r03_config4_pmc.json', 'w'), indent=1, sort_keys=True)
for k, e in out.items():
    print(k[:70], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a != 'counters_avg_per_launch'})
PY

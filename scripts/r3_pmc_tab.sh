#!/bin/bash
# Round 3: PMC passes over bench_configs.py --config 2 for the table kernels (k_tab / k_coltab and their build launches).
# One rocprofv3 --pmc run per pass (<= 8 SQ counters, <= 2 GRBM each); summary -> gpurun_out/r03_tab_pmc.json (copied to profiles/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
TAG=${1:-a}
run() { tag=$1; shift; rm -rf $O/pmc3_$tag; timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc3_$tag -- python3 $R/bench_configs.py --config 2 > $O/pmc3_$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $O/pmc3_$tag.log; exit 1; }; }
run lds SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS &&
run valu GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVES &&
python3 - "$TAG" <<'PY'
import csv, glob, collections, os, json, sys, re
R = os.environ.get('GRAFT_REPO_ROOT', '.'); O = R + '/gpurun_out'
N = 99999744.0  # elements per launch of the row-periodic cases; the scalar ones run 1e8
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ('lds', 'valu'):
    for f in glob.glob(f'{O}/pmc3_{tag}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if re.search(r'k_tab|k_coltab', k):
                agg[re.sub(r'\(.*', '', k).replace('void m355::', '')][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n = 1e8 if k.startswith('k_tab<') else N
    e = {'launches': len(next(iter(c.values()))), 'counters_avg_per_launch': m}
    if not k.startswith(('k_tab_build', 'k_coltab_build', 'k_coltab_declined')):
        e['valu_insts_per_element'] = 64 * m.get('SQ_INSTS_VALU', 0) / n      # wave instructions x 64 lanes / elements
        e['lds_insts_per_element'] = 64 * m.get('SQ_INSTS_LDS', 0) / n
        e['lds_bank_conflict_cycles_per_lds_active_cycle'] = m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_ACTIVE_INST_LDS', 0))
    gui = m.get('GRBM_GUI_ACTIVE', 0) / 8.0
    if gui > 0:
        e['valu_busy'] = 4.0 * m.get('SQ_ACTIVE_INST_VALU', 0) / (1024 * gui)
        e['lane_util'] = m.get('SQ_THREAD_CYCLES_VALU', 0) / max(1.0, 64.0 * m.get('SQ_ACTIVE_INST_VALU', 0))
    out[k] = e
json.dump(out, open(f'{O}/r03_tab_pmc_{sys.argv[1]}.json', 'w'), indent=1, sort_keys=True)
for k, e in out.items():
    print(k[:34], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in e.items() if a != 'counters_avg_per_launch'})
PY

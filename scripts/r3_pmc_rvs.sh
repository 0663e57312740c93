#!/bin/bash
# Round 3: VALU utilisation of the rvs kernels (config 3 at 2e8 draws), one rocprofv3 --pmc pass (8 SQ + 1 GRBM counters).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_r3rvs
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE \
  --output-format csv -d $O/pmc_r3rvs -- python3 $R/bench_configs.py --config 3 --n 200000000 > $O/pmc_r3rvs.log 2>&1
rc=$?; echo "[pmc_r3rvs] exit $rc"; if [ $rc -ne 0 ]; then tail -5 $O/pmc_r3rvs.log; exit $rc; fi
cd $R && python3 tools/summarize_pmc.py r03_rvs pmc_r3rvs

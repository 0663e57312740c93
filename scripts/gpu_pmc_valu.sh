#!/bin/bash
# PMC passes for the VALU-bound (special function / rvs) kernels and the MFMA kernel:
# separate rocprofv3 --pmc runs (<= 8 SQ + 2 GRBM counters each), kernel trace in its own run.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <tag> <config> <n> <counters...>
  local tag=$1 cfg=$2 n=$3; shift 3
  rm -rf $O/pmc_$tag
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$tag -- python3 $R/bench_configs.py --config $cfg --n $n > $O/pmc_$tag.log 2>&1
  local rc=$?; echo "[pmc_$tag] exit $rc"
  if [ $rc -ne 0 ]; then tail -5 $O/pmc_$tag.log; exit $rc; fi
}
pass valu2 2 20000000 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass valu3 3 200000000 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE
pass mfma4 4 100000 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE

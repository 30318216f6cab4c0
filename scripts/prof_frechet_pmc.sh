#!/bin/bash
# PMC passes over scripts/bench_frechet_large.py (one rocprofv3 run per counter group, no tracing)
# usage: prof_frechet_pmc.sh <tag>  -> gpurun_out/<tag>/pmc_<group>/
set -e
TAG=${1:-frechet}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    g=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$g -- python3 $R/scripts/bench_frechet_large.py 512 512 513 1e-2 > $OUT/pmc_$g.log 2>&1
    python3 $R/scripts/pmc_summary.py $OUT/pmc_$g k_ > $OUT/pmc_$g.txt
    rm -rf $OUT/pmc_$g
}
run sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY GRBM_GUI_ACTIVE
run tcc1 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
cat $OUT/pmc_sq1.txt $OUT/pmc_sq2.txt

#!/bin/bash
# Full profile of the default bench workload for profiles/<tag>/: kernel-trace stats, then one
# rocprofv3 --pmc pass per counter group (never combined with tracing).  usage: scripts/prof_all.sh r1
set -o pipefail
TAG=${1:-r1}
bash scripts/prof_stats.sh $TAG || exit 1
bash scripts/prof_pmc.sh $TAG fetch FETCH_SIZE GRBM_GUI_ACTIVE || exit 1
bash scripts/prof_pmc.sh $TAG write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
bash scripts/prof_pmc.sh $TAG cache TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum || exit 1
bash scripts/prof_pmc.sh $TAG sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS || exit 1
bash scripts/prof_pmc.sh $TAG sq2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA || exit 1
python3 scripts/pmc_summary.py gpurun_out/prof_$TAG k_ > gpurun_out/prof_$TAG/pmc_summary.txt
cat gpurun_out/prof_$TAG/pmc_summary.txt | head -40

#!/bin/bash
# rocprofv3: kernel trace, then issue / wait / LDS / cache counter passes of one BASELINE config's shard (scripts/prof_config.py);
# AOENV_DEBUG_OPTION in the environment selects a diagnostic path.   usage (via gpurun, repo root): bash scripts/prof_counters.sh C3 tag
set -o pipefail
C=$1; TAG=${2:-cnt}
# KERNEL_RE in the environment keeps the counter files small (a config with a long calibration writes > 64 MiB otherwise)
INC=""; if [ -n "$KERNEL_RE" ]; then INC="--kernel-include-regex $KERNEL_RE"; fi
OUT=$PWD/gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats $INC --output-format csv -d $OUT/${TAG}_trace -- python3 $REPO/scripts/prof_config.py $C > $OUT/${TAG}_trace.log 2>&1 || echo "trace failed"
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $P | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $P $INC --output-format csv -d $OUT/${TAG}_pmc_$N -- python3 $REPO/scripts/prof_config.py $C > $OUT/${TAG}_pmc_$N.log 2>&1 || echo "pass $P failed"
done
echo done

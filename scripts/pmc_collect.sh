#!/bin/bash
# rocprofv3 passes of the headline benchmark command on the GPU box (run from the repo root through gpurun):
#   kernel trace + stats, then one counter pass per line of PASSES (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are
#   never combined with the trace domains).  Outputs under gpurun_out/<tag>_*; scripts/pmc_to_json.py turns them into the
#   tracked profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc.json.
#   usage: scripts/pmc_collect.sh TAG [bench flags...]
set -o pipefail
TAG=${1:-r03}; shift
BENCH="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --min-seconds 0.2 $*"
OUT=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $REPO/$BENCH > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.err || exit 1
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  N=$(echo $P | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $P --output-format csv -d $OUT/${TAG}_pmc_$N -- python3 $REPO/$BENCH > $OUT/${TAG}_pmc_$N.json 2> $OUT/${TAG}_pmc_$N.err || echo "pass $P failed (see $OUT/${TAG}_pmc_$N.err)"
done
echo done

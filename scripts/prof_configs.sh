#!/bin/bash
# rocprofv3 kernel trace + HBM traffic counters of the BASELINE configs[2..4] shards (scripts/prof_config.py); from the repo root via gpurun
set -o pipefail
OUT=$PWD/gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_${C}_trace -- python3 $REPO/scripts/prof_config.py $C > $OUT/r03_${C}_trace.log 2>&1 || echo "trace $C failed"
  for P in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $P --output-format csv -d $OUT/r03_${C}_pmc_$P -- python3 $REPO/scripts/prof_config.py $C > $OUT/r03_${C}_pmc_$P.log 2>&1 || echo "pmc $P $C failed"
  done
done
echo done

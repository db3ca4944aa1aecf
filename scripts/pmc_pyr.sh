#!/bin/bash
# PMC pass (VALU / LDS counters) of the Pyramid passes at the C3 geometry: scripts/pmc_pyr.sh TAG   (from the repo root via gpurun)
TAG=${1:-pyr}
OUT=$PWD/gpurun_out
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/${TAG}_pmc -- python3 $REPO/scripts/time_pyr.py 256 > $OUT/${TAG}_pmc.log 2>&1
f=$(find $OUT/${TAG}_pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "pyr" not in k or "double" in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); cnt[k] += 1
for k in agg:
    print(k, cnt[k], {c: round(v / cnt[k] / 1e6, 2) for c, v in agg[k].items()})
PY

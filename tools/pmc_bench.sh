#!/bin/bash
# PMC passes (separate runs, counters only) over the default K/V-projection GEMM (gemm_bench shape 0,
# variant 5 = warp-specialised): FETCH_SIZE, WRITE_SIZE, L2 hit/miss, SQ busy, GRBM clock.
set -o pipefail
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -o c -- /root/repo/tests/native/gemm_bench 1 0 5 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections,json
res=collections.defaultdict(list)
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_ws_kernel" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:sum(v)/len(v) for k,v in res.items()}
out["launches_per_counter"]={k:len(v) for k,v in res.items()}
json.dump(out,open("$OUT/summary.json","w"),indent=1)
print(json.dumps(out))
PY

#!/bin/bash
# PMC passes (separate runs, counters only) over the folded cross-attention kernels of the video Q-Former at the
# headline shape: FETCH_SIZE and WRITE_SIZE per kernel -> gpurun_out/pmc_fold_<tag>/summary.json
set -o pipefail
TAG=${1:-r01g}
OUT=/root/repo/gpurun_out/pmc_fold_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $grp --output-format csv -d $OUT/$grp -o c -- python3 /root/repo/tools/pmc_fold_run.py > $OUT/$grp.log 2>&1 || { tail -5 $OUT/$grp.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json
keys = {"Li5ELb0ELi176ELi384ELi8E": "scores_gemm", "Li0ELb0ELi176ELi384ELi8E": "p_enc_gemm", "softmax_rescale": "softmax_rescale", "fold_rowfactor": "row_factors",
        "transpose_pad64": "enc_transpose", "Li64ELi64ELi2ELi2ELi0E": "small_gemm_epi_op"}
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for k, name in keys.items():
            if k in r["Kernel_Name"]:
                res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, c in res.items():
    if name == "enc_transpose":      # the big one is the per-forward enc transpose; the 6 small ones regroup W_k once
        c = {k: sorted(v)[-3:] for k, v in c.items()}
    fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    out[name] = {"FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "launches": len(c["FETCH_SIZE"])}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out))
PY

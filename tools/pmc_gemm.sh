#!/bin/bash
# PMC passes over the K/V-projection GEMM (gemm_bench shape 0), each counter group in its own run.
# usage: bash tools/pmc_gemm.sh <tag> <variant 0=ring 1=v1>
set -o pipefail
OUT=/root/repo/gpurun_out/pmc_$1
V=${2:-1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/l2 -o l2 -- /root/repo/tests/native/gemm_bench 1 0 $V > $OUT/l2.log 2>&1 || { tail -5 $OUT/l2.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- /root/repo/tests/native/gemm_bench 1 0 $V > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- /root/repo/tests/native/gemm_bench 1 0 $V > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq -o sq -- /root/repo/tests/native/gemm_bench 1 0 $V > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/grbm -o grbm -- /root/repo/tests/native/gemm_bench 1 0 $V > $OUT/grbm.log 2>&1 || { tail -5 $OUT/grbm.log; exit 1; }
find $OUT -name "*.csv" | head -20

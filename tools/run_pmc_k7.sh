#!/bin/bash
# HBM traffic of the K7 matrix-mode block call (ViT-g block, 4 fp16 matrices): rocprofv3 --pmc in
# SEPARATE passes (no tracing flags) over tools/wanda_launches.py --only matrixblock, summed per call
# and set against the algorithmic 2*s*numel + 4*cols.
#   gpurun --timeout 600 -- 'bash tools/run_pmc_k7.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k7_pmc
mkdir -p $OUT; rm -rf /tmp/pmc_k7f /tmp/pmc_k7w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_k7f -- python3 $R/tools/wanda_launches.py --only matrixblock > $OUT/launches.log 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_k7w -- python3 $R/tools/wanda_launches.py --only matrixblock > /dev/null 2> $OUT/write.err
f=$(find /tmp/pmc_k7f -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmc_k7w -name "*counter_collection.csv" | head -1)
grep -E "Counter_Name|wanda_matrix|sqrt_cols" $f > $OUT/fetch_k7.csv; grep -E "Counter_Name|wanda_matrix|sqrt_cols" $w > $OUT/write_k7.csv
python3 $R/tools/k7_pmc_summary.py $OUT/fetch_k7.csv $OUT/write_k7.csv > $OUT/k7_pmc_traffic.json
cat $OUT/k7_pmc_traffic.json; tail -3 $OUT/launches.log

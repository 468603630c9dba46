# HBM traffic of K1 with torch's stream in registers (zo_torch_layers_kernel): rocprofv3 --pmc in
# SEPARATE passes (no tracing flags), merged into the committed summary's layout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k1_pmc_torch
mkdir -p $OUT; rm -rf /tmp/pmc_tf /tmp/pmc_tw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_tf -- python3 $R/tools/k1_launches.py --reps 2 --form block --z torch > $OUT/launches_torch_block.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_tw -- python3 $R/tools/k1_launches.py --reps 2 --form block --z torch > /dev/null 2> $OUT/write.err
f=$(find /tmp/pmc_tf -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmc_tw -name "*counter_collection.csv" | head -1)
grep -E "Counter_Name|zo_torch_layers" $f > $OUT/fetch_torch_block.csv; grep -E "Counter_Name|zo_torch_layers" $w > $OUT/write_torch_block.csv
python3 $R/tools/summarize_pmc.py $OUT/fetch_torch_block.csv $OUT/write_torch_block.csv $OUT/launches_torch_block.json $R/profiles/k1_pmc_traffic.json > $OUT/k1_pmc_traffic_all.json
tail -30 $OUT/k1_pmc_traffic_all.json

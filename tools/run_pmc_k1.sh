# HBM traffic of the layer-batched K1 kernel: rocprofv3 --pmc in SEPARATE passes (no tracing
# flags), reduced by tools/summarize_pmc.py -> gpurun_out/k1_pmc/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k1_pmc
mkdir -p $OUT; rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/tools/k1_launches.py --reps 2 > $OUT/launches_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/tools/k1_launches.py --reps 2 > $OUT/launches_write.json 2> $OUT/write.err
f=$(find /tmp/pmc_f -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmc_w -name "*counter_collection.csv" | head -1)
grep -E "Counter_Name|zo_perturb_units" $f > $OUT/fetch.csv; grep -E "Counter_Name|zo_perturb_units" $w > $OUT/write.csv
tail -1 $OUT/launches_fetch.json > $OUT/launches.json
python3 $R/tools/summarize_pmc.py $OUT/fetch.csv $OUT/write.csv $OUT/launches.json > $OUT/k1_pmc_traffic.json
tail -12 $OUT/k1_pmc_traffic.json
# the block-batched form (zo_perturb_layers_kernel), merged into the same summary
rm -rf /tmp/pmc_bf /tmp/pmc_bw
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_bf -- python3 $R/tools/k1_launches.py --reps 2 --form block > $OUT/launches_block.json 2> $OUT/fetch_block.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_bw -- python3 $R/tools/k1_launches.py --reps 2 --form block > /dev/null 2> $OUT/write_block.err
f=$(find /tmp/pmc_bf -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmc_bw -name "*counter_collection.csv" | head -1)
grep -E "Counter_Name|zo_perturb_layers" $f > $OUT/fetch_block.csv; grep -E "Counter_Name|zo_perturb_layers" $w > $OUT/write_block.csv
python3 $R/tools/summarize_pmc.py $OUT/fetch_block.csv $OUT/write_block.csv $OUT/launches_block.json $OUT/k1_pmc_traffic.json > $OUT/k1_pmc_traffic_all.json
tail -14 $OUT/k1_pmc_traffic_all.json

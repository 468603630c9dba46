# rocprofv3 kernel trace + stats of tools/wanda_launches.py -> gpurun_out/<dir>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_wanda}
mkdir -p $OUT; rm -rf /tmp/prof_w
python3 $R/tools/wanda_launches.py > $OUT/wanda_launches.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_w -- python3 $R/tools/wanda_launches.py > /dev/null 2>&1
cp $(find /tmp/prof_w -name "*kernel_stats.csv" | head -1) $OUT/wanda_kernel_stats.csv
cat $OUT/wanda_launches.log | grep -v amdgpu.ids
grep -i "wanda\|colsq\|sqrt_cols" $OUT/wanda_kernel_stats.csv | cut -c1-170
# kernel time per call (sum of the call's kernels) of the block-level operations
tr=$(find /tmp/prof_w -name "*kernel_trace.csv" | head -1)
echo "--- K7 matrix, three-histogram path (ECOFLAP_WANDA_SAMPLED=0): kernels per call" >> $OUT/wanda_launches.log
python3 $R/tools/diag/kernel_groups.py $tr --first sqrt_cols --match wanda_matrix_hist,wanda_matrix_apply_kernel,sqrt_cols >> $OUT/wanda_launches.log
echo "--- K7 matrix, sampled-bracket path (default): kernels per call" >> $OUT/wanda_launches.log
python3 $R/tools/diag/kernel_groups.py $tr --first wanda_matrix_sbracket --match wanda_matrix_sbracket,wanda_matrix_apply2 >> $OUT/wanda_launches.log
echo "--- K6, one launch per block" >> $OUT/wanda_launches.log
python3 $R/tools/diag/kernel_groups.py $tr --first colsq_multi --match colsq_multi >> $OUT/wanda_launches.log
echo "--- K7 rows, fused grid per block" >> $OUT/wanda_launches.log
python3 $R/tools/diag/kernel_groups.py $tr --first wanda_rows_fused --match wanda_rows_fused >> $OUT/wanda_launches.log
tail -12 $OUT/wanda_launches.log

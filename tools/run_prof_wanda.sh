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

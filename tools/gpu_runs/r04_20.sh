cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04
rm -rf /tmp/pmc_g
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_g -- python3 $R/tools/gemm_f32_launches.py > $R/gpurun_out/r04/gemm_f32_pmc_run.log 2>&1
f=$(find /tmp/pmc_g -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_by_kernel.py $f gemm_f32 Cijk | tee $R/gpurun_out/r04/gemm_f32_pmc.txt | cut -c1-330

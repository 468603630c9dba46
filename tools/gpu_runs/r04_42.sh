cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
rm -f gpurun_out/verified_runs.jsonl
bash tools/diag/verified_runs.sh 3
cp gpurun_out/verified_runs.jsonl gpurun_out/r04/verified_runs_final_tree.jsonl

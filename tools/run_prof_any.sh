# rocprofv3 --kernel-trace --stats of an arbitrary python command -> gpurun_out/<dir>/kernel_stats.csv
# usage: bash tools/run_prof_any.sh <outdir> <script> [args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; rm -rf /tmp/prof_any
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_any -- python3 $R/"$@" > $OUT/stdout.txt 2> $OUT/stderr.txt
cp $(find /tmp/prof_any -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv

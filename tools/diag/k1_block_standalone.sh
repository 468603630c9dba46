# block-batched K1 alone under rocprofv3 (tools/k1_launches.py --form block), then in the bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/k1s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1s -- python3 $R/tools/k1_launches.py --reps 3 --form block > /dev/null 2>&1
python3 - $(find /tmp/k1s -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "zo_perturb_layers" in r["Kernel_Name"]]
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r["Grid_Size"]) if "Grid_Size" in r else 0
    print(f"standalone block launch grid {g:9d}: {d:8.1f} us")
PY

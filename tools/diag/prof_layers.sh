# rocprofv3 kernel trace of bench.py on a chosen set of matrices -> per-kernel totals
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_layers
mkdir -p $OUT; rm -rf /tmp/prof_l
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -- python3 $R/bench.py --no-cpu-baseline --no-parity-leg --warmup 2 --layers $1 > $OUT/bench.json 2> $OUT/bench.err
cp $(find /tmp/prof_l -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
tr=$(find /tmp/prof_l -name "*kernel_trace.csv" | head -1)
python3 - "$tr" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# timed region = after the last big gap? take the last 60 % of the trace by time as steady state
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
# find the K1 block kernel launches (zo_perturb_layers) : the timed region starts at the last one
k1 = [int(r["Start_Timestamp"]) for r in rows if "zo_perturb_layers" in r["Kernel_Name"]]
start = k1[-1] if k1 else t0
sel = [r for r in rows if int(r["Start_Timestamp"]) >= start]
wall = (t1 - start) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e6
# union of busy intervals
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel)
u = 0; cs, ce = iv[0]
for a, b in iv[1:]:
    if a > ce: u += ce - cs; cs, ce = a, b
    else: ce = max(ce, b)
u += ce - cs
print(f"region {wall:.1f} ms, kernel time {busy:.1f} ms ({busy/wall:.2f}x), device non-idle {u/1e6:.1f} ms ({u/1e6/wall:.2f}), kernels {len(sel)}")
agg = collections.Counter(); cnt = collections.Counter()
for r in sel:
    n = r["Kernel_Name"]
    key = ("fp32 GEMM" if "Cijk" in n and "_S_B_" in n else "fp16 GEMM" if "Cijk" in n and "HHS" in n else
           "bf16 GEMM" if "Cijk" in n and "BBS" in n else "attention" if "attn_fwd" in n else n[:60])
    agg[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[key] += 1
for k, v in agg.most_common(18):
    print(f"  {k:62s} {cnt[k]:7d} {v/1e6:8.1f} ms {100*v/1e6/busy:5.1f}%")
PY

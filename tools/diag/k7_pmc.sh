# SQ counters of the K7 rows kernels (separate rocprofv3 --pmc passes, no tracing flags)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k7_pmc
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf /tmp/pmc_k7
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_k7 -- python3 $R/tools/wanda_launches.py --only rows > /dev/null 2> $OUT/err_$i.txt
  f=$(find /tmp/pmc_k7 -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then grep -E "Counter_Name|wanda_rows" $f > $OUT/grp_$i.csv; else echo "group $i failed"; tail -3 $OUT/err_$i.txt; fi
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/grp_*.csv")):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:70], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    if "fused" not in k[0]:
        continue
    print(k)
    m = {c: sum(v) / len(v) for c, v in d.items()}
    for c, v in m.items():
        print(f"   {c:24s} {v:16.1f}")
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        # SQ counters are in quad-cycles summed over the chip; GRBM_GUI_ACTIVE in cycles summed over 8 XCDs
        quads = m["GRBM_GUI_ACTIVE"] / 8 / 4 * 1024
        print(f"   VALU busy = {m['SQ_ACTIVE_INST_VALU'] / quads:.2f} of 1024 SIMDs x kernel time; "
              f"VALU instructions per wave = {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}")
PY

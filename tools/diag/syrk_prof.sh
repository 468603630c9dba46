# kernel trace + two PMC passes of the SYRK launches (tools/wanda_launches.py --only syrk)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/syrk_prof
mkdir -p $OUT; rm -rf /tmp/sp_*
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp_t -- python3 $R/tools/wanda_launches.py --only syrk > $OUT/trace_run.log 2>&1
cp $(find /tmp/sp_t -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
grep -i "syrk" $OUT/kernel_stats.csv | cut -c1-160
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/sp_p1 -- python3 $R/tools/wanda_launches.py --only syrk > $OUT/pmc1_run.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/sp_p2 -- python3 $R/tools/wanda_launches.py --only syrk > $OUT/pmc2_run.log 2>&1
for d in sp_p1 sp_p2; do f=$(find /tmp/$d -name "*counter_collection.csv" | head -1); cp $f $OUT/$d.csv; done
python3 - $OUT/sp_p1.csv $OUT/sp_p2.csv <<'PY'
import csv, sys, collections
# per dispatch: the x8 launches are the upper half of each kernel's dispatches by SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE
for path in sys.argv[1:]:
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if "syrk256_kernel<1" in r["Kernel_Name"]:
            disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    rows = list(disp.values())
    key = "SQ_WAVE_CYCLES" if "SQ_WAVE_CYCLES" in rows[0] else "GRBM_GUI_ACTIVE"
    rows.sort(key=lambda d: d[key])
    big = rows[len(rows) // 2:]
    print("syrk256<1> x8 launches:", {c: round(sum(d[c] for d in big) / len(big)) for c in big[0]}, "n", len(big))
PY
grep syrk256 $OUT/kernel_stats.csv | cut -c1-200

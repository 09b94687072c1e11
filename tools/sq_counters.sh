set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/plm; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-stage-pass --no-clip"
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -o r -- $BENCH > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/sq2 -o r -- $BENCH > $OUT/sq2.log 2>&1
cd $ROOT
python - <<'PY'
import csv,glob,collections
for d in ("sq","sq2"):
    f=glob.glob(f"gpurun_out/plm/{d}/**/*counter_collection.csv",recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in f:
        for r in csv.DictReader(open(fn)):
            agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "blend" in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY

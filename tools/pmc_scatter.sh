#!/bin/bash
# dev: SQ counters of the two partition passes separately (full template names).  usage: pmc_scatter.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_scatter; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-other > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-other > $OUT/sq2.log 2>&1
cd $R
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(sys.argv[1], sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "Scatter" not in n: continue
            acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    print("   " + "  ".join("%s %.4g" % (c, sum(v[len(v)//2:]) / len(v[len(v)//2:])) for c, v in sorted(cs.items())))
PY
rm -rf $OUT/sq $OUT/sq2

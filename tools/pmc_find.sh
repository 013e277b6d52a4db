#!/bin/bash
# dev: L2 hit / miss and HBM fetch of the lookup kernels of config 3 (direct probes against the partitioned path)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "direct 9" "part 8" "part 9"; do
  set -- $cfg
  export MODGPU_FIND_PATH=$1 MODGPU_FIND_BITS=$2
  for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    rm -rf /tmp/pf; mkdir -p /tmp/pf
    (cd /tmp && rocprofv3 --pmc $c --output-format csv -d /tmp/pf -- python3 $R/bench.py --only c3 --steps 3 > /tmp/pf/log 2>&1)
    f=$(find /tmp/pf -name "*counter_collection.csv" | head -1)
    echo "== $cfg : $c"
    python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
    if n in ("mgBinFindKernel", "mgTableFindSegKernel", "mgUnpartKernel", "mgPartScatterKernel"):
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print("  ", n, {c: round(sum(v[len(v)//2:]) / len(v[len(v)//2:])) for c, v in cs.items()})
PY
  done
done

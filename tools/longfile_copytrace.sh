#!/bin/bash
# dev: memory-copy + kernel trace of tools/longfile_probe.py -> per-copy durations and rates of the 128 MiB window copies (gpurun_out/longfile_copies.txt)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/lfc
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lfc -- python3 $R/tools/longfile_probe.py > /tmp/lfc.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/longfile_copies.txt
import csv, glob
f = sorted(glob.glob('/tmp/lfc/**/*memory_copy_trace.csv', recursive=True), key=lambda p: -__import__('os').path.getsize(p))[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
big = [r for r in rows if 'HOST_TO_DEVICE' in r.get('Direction', r.get('Kind', '')).upper() or 'H2D' in str(r)]
ws = []
for r in rows:
    try:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    except Exception:
        continue
    ws.append((s, e, r))
ws.sort()
win = [(s, e, r) for s, e, r in ws if (e - s) > 1_000_000]           # copies of more than 1 ms
print(len(ws), "copies,", len(win), "of more than 1 ms")
t0 = win[0][0] if win else 0
for s, e, r in win[-60:]:
    print("start %10.3f ms  dur %7.3f ms  %s" % ((s - t0) / 1e6, (e - s) / 1e6, {k: r[k] for k in r if k in ('Direction', 'Kind', 'Bytes', 'Size')}))
PY
tail -70 gpurun_out/longfile_copies.txt

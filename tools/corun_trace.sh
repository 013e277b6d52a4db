#!/bin/bash
# dev: kernel trace of tools/corun_probe.py (scan alone, build alone, both at once from two streams): per-kernel durations
# in the three phases -> gpurun_out/corun_trace_<tag>.txt.  usage: tools/corun_trace.sh <tag> [ENV=...]
TAG=$1; shift
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in "$@"; do export "$a"; done
cd /tmp
rm -rf /tmp/corun_$TAG
rocprofv3 --kernel-trace --output-format csv -d /tmp/corun_$TAG -- python3 $R/tools/corun_probe.py > /tmp/corun_$TAG.log 2>&1
cd $R
tail -1 /tmp/corun_$TAG.log > gpurun_out/corun_trace_$TAG.txt
python3 tools/corun_trace_summary.py /tmp/corun_$TAG >> gpurun_out/corun_trace_$TAG.txt 2>&1
cat gpurun_out/corun_trace_$TAG.txt

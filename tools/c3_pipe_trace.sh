#!/bin/bash
# dev: kernel trace of bench.py --only c3 with the pipelined query (scan of batch i + 1 beside the lookups of batch i): per kernel, mean duration
# alone (the synchronous warm-up batch, the first scan, the last lookups) and overlapped -> gpurun_out/c3_pipe_trace.txt
export TMPDIR=/tmp MODGPU_BENCH_C3_PIPE=1 MODGPU_BENCH_C3_REFREAD=0
R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/c3pipe
rocprofv3 --kernel-trace --output-format csv -d /tmp/c3pipe -- python3 $R/bench.py --only c3 > /tmp/c3pipe.log 2>&1
cd $R
python3 -c "
import json;j=json.loads(open('/tmp/c3pipe.log').read().strip().splitlines()[-1])['result'];print('pipelined', j['pipelined'], j['value'], 'Gbp/s', j['ms_per_batch'], 'ms per batch', j['ms_each_batch'])" > gpurun_out/c3_pipe_trace.txt
python3 tools/corun_trace_summary.py /tmp/c3pipe >> gpurun_out/c3_pipe_trace.txt 2>&1
cat gpurun_out/c3_pipe_trace.txt

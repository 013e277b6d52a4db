#!/usr/bin/env python3
"""dev: BASELINE config 3 alone (bench.bench_c3), for profiling the query path"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import modimizer_amd as mg
from modimizer_amd import synth
cx = bench.Ctx()
cx.torch, cx.dist, cx.mg, cx.synth = torch, None, mg, synth
cx.dev = torch.device("cuda", 0); cx.L = mg.lib()
mg.check(cx.L.mgSetDevice(0))
cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
class A: steps = 5
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
print(json.dumps(getattr(bench, "bench_" + which)(cx, A), indent=1))

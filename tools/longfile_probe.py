"""dev: bench.py's end_to_end.modmap_query_file_long alone, with the text parser's and the callers' lap timing on stderr
(MODGPU_TEXT_TIMING=1 MODGPU_SEED_TIMING=1): where a 3 Gbp reference file and a 5 Gbp query file spend their time.
usage: longfile_probe.py [query Gbp]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1:
    os.environ["MODGPU_BENCH_LONG_QUERY_GBP"] = sys.argv[1]
os.environ["MODGPU_TEXT_TIMING"] = "1"; os.environ["MODGPU_SEED_TIMING"] = "1"
import torch
import modimizer_amd as mg
import bench
L = mg.lib(); mg.check(L.mgSetDevice(0))
cx = bench.Ctx(); cx.torch, cx.mg, cx.L = torch, mg, L
from modimizer_amd import synth
cx.synth = synth; cx.dev = torch.device("cuda", 0); cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print(json.dumps(bench.modmap_query_file_long(cx), indent=1))

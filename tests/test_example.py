"""examples/sketch_file.c: the C ABI used from plain C (compiled with gcc against include/modgpu.h)."""
import os
import subprocess

import pytest

from tests import util

SRC = os.path.join(util.ROOT, "examples", "sketch_file.c")
SRC_MULTI = os.path.join(util.ROOT, "examples", "multi_gpu.c")
SRC_MAP = os.path.join(util.ROOT, "examples", "map_file.c")


@pytest.mark.parametrize("src", [SRC, SRC_MULTI, SRC_MAP])
def test_example_is_plain_c99(src, tmp_path):
    """the header and the examples compile as C99 with warnings on (no C++-isms, no HIP types in the ABI)"""
    r = subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=c99", "-I", os.path.join(util.ROOT, "include"),
                        "-c", src, "-o", str(tmp_path / "x.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def build_multi_gpu(tmp_path):
    exe = str(tmp_path / "multi_gpu")
    libdir = os.path.join(util.ROOT, "modimizer_amd")
    r = subprocess.run(["gcc", "-O2", "-pthread", "-I", os.path.join(util.ROOT, "include"), SRC_MULTI, "-o", exe, "-L", libdir, "-lmodgpu", "-lm",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.gpu
def test_multi_gpu_example_on_one_gpu(tmp_path):
    """examples/multi_gpu.c = BASELINE config 4 in C (mgCommInitAll on librccl, a host thread per GPU, mgHistogramAllReduce,
    mgModsetMergeRankOrder, mgDepthAllReduce) at N = 1: the all-reduced histogram is the GPU's own (modsetDepthHistogramDevice), the merged set is the
    single-stream set.  N > 1 runs where the box has more GPUs (tests/test_dist.py::test_bench_on_every_gpu_of_the_box)."""
    exe = build_multi_gpu(tmp_path)
    r = subprocess.run([exe, "1", "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
    assert "MULTI_GPU_OK" in r.stdout and "histogram: all-reduced == sum of the ranks' own on every rank: yes" in r.stdout
    assert "identical to the single-stream build over all blocks (value[], depth[]): yes" in r.stdout
    assert "sum over the GPUs == one stream, on every GPU: yes" in r.stdout          # mgDepthAllReduce (reads counted against a fixed set)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["k21d16", "k17d31"])
def test_ingest_example_runs_like_modasm(tag, golden_dir, tmp_path):
    """examples/ingest_file.c = `modasm -m src.mod -f reads2.fa -S -w stem` on the library: the reference program's own stdout
    (tests/golden/asm_*.stdout.txt) and, but for the addresses a file holds, its .mod and .readset bytes"""
    import gzip
    from tests import test_readset as trs
    exe = str(tmp_path / "ingest_file")
    libdir = os.path.join(util.ROOT, "modimizer_amd")
    r = subprocess.run(["gcc", "-O2", "-I", os.path.join(util.ROOT, "include"), os.path.join(util.ROOT, "examples", "ingest_file.c"), "-o", exe,
                        "-L", libdir, "-lmodgpu", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    stem = os.path.join(golden_dir, "asm_%s" % tag)
    out = str(tmp_path / "out")
    r = subprocess.run([exe, stem + "_src.mod", os.path.join(golden_dir, "reads2.fa"), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.splitlines() == util.golden_text("asm_%s.stdout.txt" % tag).splitlines()[:len(r.stdout.splitlines())] and r.stdout.count("RS ") == 6
    assert trs.mod_mask(gzip.open(out + ".mod").read()) == trs.mod_mask(gzip.open(stem + ".mod").read())
    assert trs.readset_mask(gzip.open(out + ".readset").read()) == trs.readset_mask(gzip.open(stem + ".readset").read())


@pytest.mark.gpu
def test_example_runs_like_modutils(golden_dir, tmp_path):
    exe = str(tmp_path / "sketch_file")
    libdir = os.path.join(util.ROOT, "modimizer_amd")
    r = subprocess.run(["gcc", "-O2", "-I", os.path.join(util.ROOT, "include"), SRC, "-o", exe, "-L", libdir, "-lmodgpu",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    hist, dump = str(tmp_path / "h.txt"), str(tmp_path / "d.txt")
    r = subprocess.run([exe, "20", "21", "64", "17", hist, dump, os.path.join(golden_dir, "reads.fa"),
                        os.path.join(golden_dir, "reads2.fa")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == util.golden_text("modutils_k21d64.stdout.txt").splitlines()[:9]
    assert open(hist).read() == util.golden_text("modutils_k21d64.hist.txt")
    assert open(dump).read() == util.golden_text("modutils_k21d64.dump.txt")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(util.MODMAP_TAGS))
@pytest.mark.parametrize("through_files", [False, True])
def test_map_example_runs_like_modmap(tag, through_files, golden_dir, tmp_path):
    """examples/map_file.c = `modmap -f ref.fa [-w stem -r stem] -q queries.fa` from plain C: the reference program's output, line for line;
    with a stem the reference is written (multi-member gzip .mod + .ref), read back (the parallel reader) and queried from the copy"""
    exe = str(tmp_path / "map_file")
    libdir = os.path.join(util.ROOT, "modimizer_amd")
    r = subprocess.run(["gcc", "-O2", "-I", os.path.join(util.ROOT, "include"), SRC_MAP, "-o", exe, "-L", libdir, "-lmodgpu",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    k, w = util.MODMAP_TAGS[tag]
    args = [exe, "20", str(k), str(w), "17", os.path.join(golden_dir, "ref.fa"), os.path.join(golden_dir, "queries.fa")]
    if through_files:
        args.append(str(tmp_path / "stem"))
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    assert r.stdout.splitlines() == util.golden_text("modmap_%s.stdout.txt" % tag).splitlines()

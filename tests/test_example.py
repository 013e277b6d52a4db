"""examples/sketch_file.c: the C ABI used from plain C (compiled with gcc against include/modgpu.h)."""
import os
import subprocess

import pytest

from tests import util

SRC = os.path.join(util.ROOT, "examples", "sketch_file.c")


def test_example_is_plain_c99(tmp_path):
    """the header and the example compile as C99 with warnings on (no C++-isms, no HIP types in the ABI)"""
    r = subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-std=c99", "-I", os.path.join(util.ROOT, "include"),
                        "-c", SRC, "-o", str(tmp_path / "x.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


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

"""The C-ABI boundary: libmodgpu.so loads, exports every symbol include/modgpu.h and modgpu_compat.h declare, keeps the
reference's struct layouts, and fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import modimizer_amd as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "modgpu.h")).read() + open(os.path.join(ROOT, "include", "modgpu_compat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w \t\*]*?[\s\*]([A-Za-z_]\w*)\s*\([^;{]*\)\s*;", src, flags=re.M):
        names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol():
    L = mg.lib()
    decl = header_functions()
    assert len(decl) > 50
    missing = [n for n in sorted(decl) if not hasattr(L, n)]
    assert not missing, missing
    assert set(mg.EXPORTS) <= decl | {"seqhash"}
    # and the python binding lists them all
    assert decl <= set(mg.EXPORTS), sorted(decl - set(mg.EXPORTS))


def test_struct_layouts_match_reference():
    # seqhash.h:15-23 is 80 bytes with mask at 16 and factor1 at 32 (measured on the reference)
    assert C.sizeof(mg.Seqhash) == 80
    assert mg.Seqhash.mask.offset == 16 and mg.Seqhash.shift1.offset == 24
    assert mg.Seqhash.factor1.offset == 32 and mg.Seqhash.patternRC.offset == 48
    assert C.sizeof(mg.SeqhashRCiterator) == 72
    assert mg.SeqhashRCiterator.hashBuf.offset == 40 and mg.SeqhashRCiterator.isDone.offset == 68
    assert mg.Modset.index.offset == 32 and mg.Modset.max.offset == 64 and C.sizeof(mg.Modset) == 72


def test_product_never_touches_the_oracle():
    """modimizer_amd/ (python + C/HIP sources) must not import, link or call anything under oracle/"""
    pkg = os.path.join(ROOT, "modimizer_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(d, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt, (d, f)
    out = subprocess.run(["ldd", mg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "modref" not in out


no_gpu = pytest.mark.skipif(mg.lib().mgDeviceCount() > 0, reason="checks the no-device behaviour")


@no_gpu
def test_batch_path_fails_loudly_without_device():
    L = mg.lib()
    sh = mg.seqhashCreate(21, 64, 17)
    bases = np.zeros(100, np.uint8)
    offs = np.array([0, 100], np.int64)
    with pytest.raises(mg.ModgpuError, match="no HIP device"):
        mg.scan_batch(sh, bases, offs)
    ms = mg.modsetCreate(sh, 20)
    assert L.modsetAddBatchDevice(ms, None, 10, None, 1, None) == 1      # MG_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.mgLastError()
    assert L.mgAddSequenceBatch(ms, bases.ctypes.data, offs.ctypes.data, 1) == -1
    p = C.c_void_p()
    assert L.mgDeviceAlloc(C.byref(p), 64) == 1
    L.modsetDestroy(ms)


@no_gpu
@pytest.mark.parametrize("n", [100, 10000])
def test_iterator_dies_without_device(n):
    """modRCiterator needs a HIP device whichever leg a read's length selects (the scalar loop below the launch-latency
    crossover, one kernel launch above it): with no device it die()s like every reference error (utils.c:19-30)"""
    code = ("import numpy as np, modimizer_amd as mg\n"
            "sh = mg.seqhashCreate(21, 64, 17)\n"
            "mg.iterate(sh, np.zeros(%d, np.uint8))\n" % n)
    env = dict(os.environ, MODGPU_NO_TORCH="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "FATAL ERROR" in r.stderr and "modRCiterator" in r.stderr and "no HIP device" in r.stderr


def test_binary_carries_the_hash_of_its_sources(tmp_path, monkeypatch):
    """libmodgpu.so is git-ignored yet rides along to the GPU box, so the binary says what it was built from --
    csrc/Makefile bakes a hash of every source into it (mg_version.c), the binding recomputes it from the tree, and a library whose
    hash differs is rebuilt before it is loaded (modimizer_amd.build); smoke() asserts equality."""
    import shutil
    L = mg.lib()
    h = mg.source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h)
    assert L.mgSourceHash().decode() == h == mg.binary_hash()
    assert L.mgVersion().decode().endswith("src=" + h)
    # any change to any source changes the tree's hash: the unchanged binary is then seen as stale
    csrc = tmp_path / "csrc"
    shutil.copytree(mg.CSRC, csrc, ignore=shutil.ignore_patterns("*.o"))
    monkeypatch.setattr(mg, "CSRC", str(csrc))
    assert mg.source_hash() == h
    for name in ("mg_scan.hip", "mg_host.c", "mg_common.h", "Makefile"):
        p = os.path.join(str(csrc), name)
        old = open(p, "rb").read()
        open(p, "ab").write(b"\n")
        assert mg.source_hash() != h, name
        open(p, "wb").write(old)
        assert mg.source_hash() == h
    # a library file without the marker (or none at all) has no hash
    assert mg.binary_hash(str(tmp_path / "nothing.so")) is None

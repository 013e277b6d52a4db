"""dev: rate of mgCopyD2HBig / mgCopyH2DBig (mg_xfer.hip) by piece size and thread count; destination pages warm and cold"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import modimizer_amd as mg
L = mg.lib(); mg.check(L.mgSetDevice(0))
n = 1 << 30
d = mg.DeviceBuffer(n)
mg.check(L.mgMemsetD(d.ptr, 7, n, None)); mg.check(L.mgStreamSynchronize(None))
for kb, th, own in [(4096, 4, "0"), (4096, 4, "1"), (4096, 2, "0"), (4096, 6, "0"), (2048, 4, "0"), (8192, 4, "0"), (4096, 4, "0"), (4096, 4, "1")]:
    if True:
        with mg.knobs(XFER_PIECE_KB=str(kb), XFER_THREADS=str(th), XFER_STREAMS=own):
            L.mgReleaseBuffers()
            h = np.empty(n, np.uint8)
            t0 = time.perf_counter(); mg.check(L.mgCopyD2HBig(h.ctypes.data, d.ptr, n)); t_cold = time.perf_counter() - t0
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); mg.check(L.mgCopyD2HBig(h.ctypes.data, d.ptr, n)); ts.append(time.perf_counter() - t0)
            assert h[0] == 7 and h[-1] == 7 and int(h[::4097].sum()) == 7 * len(h[::4097])
            tu = []
            for _ in range(3):
                t0 = time.perf_counter(); mg.check(L.mgCopyH2DBig(d.ptr, h.ctypes.data, n)); tu.append(time.perf_counter() - t0)
            print("own streams %s piece %5d KiB threads %2d: D2H first (lanes + cold pages) %6.1f ms, warm %6.1f ms = %5.1f GB/s; H2D %6.1f ms = %5.1f GB/s"
                  % (own, kb, th, t_cold * 1e3, min(ts) * 1e3, n / min(ts) / 1e9, min(tu) * 1e3, n / min(tu) / 1e9), flush=True)
            del h

#!/usr/bin/env python3
"""bench.py — Gbp/s hashed+sketched (k=21, d=64) on MI355X, with HBM-roofline and CPU baseline.

One "step" = one pass of the hot path over one batch of synthetic reads already resident in HBM:
clear the modset (what modsetCreate's calloc is to the reference), scan every read for modimizers
(seqhash.c:154-196) and insert them with depth counting (modset.c:45-62 + modutils.c:19-31).
With N > 1 each rank owns its own contiguous block of reads (weak scaling) and builds its own
modset; the 65536-bin depth histograms are summed with an RCCL all-reduce (BASELINE.json config 4).

Workload at N=1: BASELINE.json configs[1] — 10 Gbp ONT-like reads (log-normal lengths, N50 20 kb,
5 % substitutions, 30x of a 333 Mbp genome), k=21 d=64 seed 17, table bits 30.
Environment overrides (for quick runs): MODGPU_BENCH_GBP, MODGPU_BENCH_BITS, MODGPU_CPU_SAMPLE_MBP.

Launch: python bench.py [--gpus N --steps K --warmup W]   (N>1 via torch.distributed.run)
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only does dmabuf IPC (RCCL needs it)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    import torch.distributed as dist
    import modimizer_amd as mg
    from modimizer_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    L = mg.lib()
    mg.check(L.mgSetDevice(local_rank))
    # MODGPU_BENCH_FORCE_DIST=1: run the multi-rank code path (process group, histogram all-reduce, barriers)
    # with whatever WORLD_SIZE the launcher gave, even 1 — for checking that path on a one-GPU box
    multi = world > 1 or os.environ.get("MODGPU_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # RCCL writes its version banner to stdout (C stdio, block-buffered on a pipe) when the communicator
        # comes up: bring it up and push the banner out now, so that rank 0's JSON line is the last line of stdout
        dist.barrier()
        torch.cuda.synchronize()
        C.CDLL(None).fflush(None)

    k, d, seed, bits = 21, 64, 17, int(os.environ.get("MODGPU_BENCH_BITS", "30"))
    gbp = float(os.environ.get("MODGPU_BENCH_GBP", "10"))
    total = int(gbp * 1e9)
    genome_bases = max(int(total / 30), 1_000_000)
    err = 0.05
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- synthetic reads, generated in HBM -------------------------------------------------
    t_gen = time.time()
    starts, offsets, strands = synth.ont_read_plan(total, genome_bases, seed=1000 + rank)
    n_reads = len(starts)
    genome = torch.empty(L.mgPackedWords(genome_bases), dtype=torch.int32, device=dev)
    mg.check(L.mgSynthGenome(genome.data_ptr(), genome_bases, 12345, stream))
    d_starts = torch.from_numpy(starts.view(np.int64)).to(dev)
    d_offsets = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_strands = torch.from_numpy(strands).to(dev)
    reads = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=dev)
    mg.check(L.mgSynthReads(genome.data_ptr(), genome_bases, d_starts.data_ptr(), d_offsets.data_ptr(),
                            d_strands.data_ptr(), n_reads, total, err, 777 + rank, reads.data_ptr(), stream))
    torch.cuda.synchronize()
    del genome
    t_gen = time.time() - t_gen

    sh = mg.seqhashCreate(k, d, seed)
    ms = mg.modsetCreate(sh, bits)
    hist = torch.zeros(65536, dtype=torch.int64, device=dev)
    n_hash = C.c_uint64(0)

    def step():
        mg.check(L.mgModsetClear(ms, stream))
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads,
                                    C.byref(n_hash), stream))
        if multi:
            hist.zero_()
            mg.check(L.modsetDepthHistogramDevice(ms, hist.data_ptr(), stream))
            dist.all_reduce(hist)

    def read_profile():
        out = {}
        for i in range(L.mgProfileKernels()):
            name = C.c_char_p(); ms_tot = C.c_double(); n = C.c_uint64()
            mg.check(L.mgProfileGet(i, C.byref(name), C.byref(ms_tot), C.byref(n)))
            if n.value:
                out[name.value.decode()] = (ms_tot.value, n.value, i)
        return out

    # Warm-up steps run with every kernel launch bracketed by HIP events (on the launch stream): that gives the
    # per-kernel table and says which kernel dominates.  An event pair costs a few microseconds of stream time per
    # launch (25 launches a step), so in the timed region only the dominant kernel is bracketed.
    L.mgProfileOnly(-1)
    L.mgProfileEnable(1)
    L.mgProfileReset()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    warm = read_profile()
    warm_steps = args.warmup
    dom_id = max(warm.values(), key=lambda v: v[0])[2] if warm else -1

    L.mgProfileReset()
    L.mgProfileOnly(dom_id)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- HIP-event timings over the timed region -> roofline of the dominant kernel ---------
    kern = read_profile()
    # one more step, outside the timed region, with every launch bracketed: the per-kernel table
    L.mgProfileOnly(-1)
    L.mgProfileReset()
    step()
    torch.cuda.synchronize()
    warm = read_profile()
    warm_steps = 1
    L.mgProfileEnable(0)
    S = n_hash.value
    entries = ms.contents.max
    alg_bytes = {                                   # algorithmic bytes per launch
        "mgScanKernel": (0.25 + 12.0 / d) * total,  # 2-bit read + (kmer 8 + pos 4) per modimizer
        "mgSegCompactKernel": 16.0 * S,             # kmer read + written
        "mgPartHistKernel": 8.0 * S,
        "mgPartScatterKernel": 24.0 * S,            # (kmer 8 + ordinal 4) read and written, per pass
        "mgBucketDedupKernel": 12.0 * S + 16.0 * entries,
        "mgRankAssignKernel": 9.0 * S + 8.0 * entries,
        "mgBucketMergeKernel": 16.0 * entries + 16.0 * entries / 0.6,   # uniques in, table buckets out
        "mgTableInsertKernel": 16.0 * S,
    }
    dom = max(kern.items(), key=lambda kv: kv[1][0])[0] if kern else None
    roofline = None
    if dom:
        avg_ms = kern[dom][0] / kern[dom][1]
        ach = alg_bytes.get(dom, 0.0) / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(HERE, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("%g" % gbp)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4),
                    "algorithmic_bytes_per_launch": alg_bytes.get(dom),
                    "kernels_ms_per_step": ({kname: round(v[0] / warm_steps, 4) for kname, v in sorted(warm.items())}
                                            if warm_steps and dom_id >= 0 else
                                            {kname: round(v[0] / args.steps, 4) for kname, v in sorted(kern.items())}),
                    "kernels_ms_per_step_from": "one extra step after the timed region, every launch bracketed" if warm_steps and dom_id >= 0 else "timed steps"}

    value = world * total * args.steps / dt / 1e9
    out = {
        "metric": "Gbp/s hashed+sketched (k=21,d=64)", "value": round(value, 3), "unit": "Gbp/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "%g Gbp per GPU synthetic ONT-like reads (log-normal N50 20 kb, 5%% subs, 30x of a %d Mbp genome), "
                               "k=21 d=64 seed=17, seqhash scan + modset build (table bits %d)%s"
                               % (gbp, genome_bases // 1_000_000, bits,
                                  ", per-GPU build + RCCL all-reduce of the depth histogram" if world > 1 else ""),
                   "reads_per_gpu": n_reads, "bases_per_gpu": total, "modimizers_per_gpu": S,
                   "modset_entries": entries, "k": k, "d": d, "table_bits": bits,
                   "parallelism": "reads sharded x%d, modset per GPU" % world},
        "roofline": roofline,
        "setup_s": round(t_gen, 2),
    }

    # ---- CPU baseline (rank 0, N=1 only): the compiled reference on a bounded sample ---------
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(L, mg, torch, dev, reads, offsets, k, d, seed, stream)

    if multi:
        dist.barrier()                    # every rank is done (and silent) before the line goes out
        C.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def cpu_baseline(L, mg, torch, dev, reads, offsets, k, d, seed, stream):
    """The reference's own C path (oracle/_ref/ref_bench, built from the unmodified sources) timed
    single-threaded — its real execution model — on the first reads of the same workload."""
    sample_mbp = float(os.environ.get("MODGPU_CPU_SAMPLE_MBP", "1200"))
    want = int(sample_mbp * 1e6)
    n = int(np.searchsorted(offsets, want, side="right")) - 1
    n = max(1, min(n, len(offsets) - 1))
    nb = int(offsets[n])
    d_bytes = torch.empty(nb, dtype=torch.uint8, device=dev)
    mg.check(L.mgUnpackDevice(reads.data_ptr(), nb, d_bytes.data_ptr(), stream))
    torch.cuda.synchronize()
    h_bytes = d_bytes.cpu().numpy()
    del d_bytes
    off = offsets[:n + 1].astype(np.int64)
    sample_desc = "first %d reads (%.0f Mbp) of the same workload, single thread, table bits 28" % (n, nb / 1e6)
    ref_bench = os.path.join(HERE, "oracle", "_ref", "ref_bench")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, "modgpu_cpu_sample_%d.bin" % os.getpid())
    try:
        if os.path.exists(ref_bench):
            with open(path, "wb") as f:
                np.array([n, nb], np.uint64).tofile(f); off.tofile(f); h_bytes.tofile(f)
            try:
                r = subprocess.run([ref_bench, path, str(k), str(d), str(seed), "28"],
                                   capture_output=True, text=True, timeout=900)
                if r.returncode == 0:
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    res = {"value": round(j["sketch_mbps"] / 1e3, 5), "unit": "Gbp/s", "cores": 1,
                           "kind": "reference", "sample": sample_desc,
                           "scan_only_gbps": round(j["scan_mbps"] / 1e3, 5),
                           "host_cores_online": os.cpu_count()}
                    try:
                        res["all_cores_port"] = all_cores_port(h_bytes, off, k, d, seed)
                    except Exception as e:      # informational only
                        res["all_cores_port"] = {"error": str(e)[:200]}
                    return res
            except Exception:
                pass
    finally:
        if os.path.exists(path):
            os.remove(path)
    # fall back to this repo's C restatement of the same path
    from oracle import pyoracle as po
    oh = po.Hasher(k, d, seed)
    oms = po.Modset(oh, 28)
    t0 = time.perf_counter()
    po.lib().orcScanMany(C.byref(oh.c), h_bytes.ctypes.data, off.ctypes.data, n, oms.p)
    dt = time.perf_counter() - t0
    return {"value": round(nb / dt / 1e9, 5), "unit": "Gbp/s", "cores": 1, "kind": "port",
            "sample": sample_desc, "host_cores_online": os.cpu_count()}


def all_cores_port(h_bytes, off, k, d, seed):
    """Informational, so the GPU is not flattered by a single-thread baseline: this repo's C restatement
    (oracle/, kind "port") with the sample's reads sharded over every host core, each thread building a
    PRIVATE modset (no merge step, which favours the CPU).  The reference itself has no threading."""
    import threading
    from oracle import pyoracle as po
    n = len(off) - 1
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                        # the box may cap CPU time below the core count (cgroup v2 cpu.max)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            avail = max(1, min(avail, int(quota) // int(period)))
    except Exception:
        pass
    T = max(1, min(avail, 64, n))
    hashers = [po.Hasher(k, d, seed) for _ in range(T)]
    sets = [po.Modset(hashers[t], 24) for t in range(T)]
    bounds = [n * t // T for t in range(T + 1)]
    lib = po.lib()

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        sub = np.ascontiguousarray(off[lo:hi + 1] - off[lo])
        base = h_bytes[int(off[lo]):int(off[hi])]
        lib.orcScanMany(C.byref(hashers[t].c), base.ctypes.data, sub.ctypes.data, hi - lo, sets[t].p)
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    return {"value": round(int(off[-1]) / dt / 1e9, 4), "unit": "Gbp/s", "threads": T, "kind": "port",
            "note": "private per-thread modsets (table bits 24), no merge"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — Gbp/s hashed+sketched (k=21, d=64) on MI355X, with HBM-roofline and CPU baseline.

One "step" = one pass of the hot path over one batch of synthetic reads already resident in HBM:
clear the modset (what modsetCreate's calloc is to the reference), scan every read for modimizers
(seqhash.c:154-196) and insert them with depth counting (modset.c:45-62 + modutils.c:19-31).

Workloads (BASELINE.json configs; SURVEY §8(d) sizes):
  N = 1   configs[1]: 10 Gbp ONT-like reads (log-normal lengths, N50 20 kb, 5 % substitutions, 30x of a 333 Mbp
          genome), k=21 d=64 seed 17, table bits 30.
  N > 1   configs[3]: the 100 Gbp ONT set cut into contiguous blocks of 12.5 Gbp of reads, one block per GPU (rank r
          owns block r; with N < 8 the first N blocks), all drawn from the same 3.33 Gbp genome; per-GPU modset build
          + one RCCL all-reduce of the 65536-bin depth histogram per step.  Weak scaling: 12.5 Gbp per GPU at every N.
  other_configs (N = 1, after the headline): configs[2] (modmap: 3 Gbp reference modset, 90 Gbp of reads queried in
          10 Gbp batches) and configs[4] (depth histogram of 50x 150 b reads, k=31 d=4), each with its own roofline.

Launch: python bench.py [--gpus N --steps K --warmup W]
  --gpus N > 1 without WORLD_SIZE in the environment: this process starts N ranks (torch.distributed.run, one per GPU)
  BEFORE anything touches the GPU and passes their output through; under torch.distributed.run it is a rank itself.
Environment overrides (for quick runs): MODGPU_BENCH_GBP, MODGPU_BENCH_BITS, MODGPU_CPU_SAMPLE_MBP,
MODGPU_BENCH_FORCE_DIST=1 (run the N>1 code path at whatever world size, even 1); MODGPU_BENCH_ONE_GPU=1 (all ranks on cuda:0,
gloo: the world > 1 code path on a one-GPU box, rates meaningless).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this pool's driver only does dmabuf IPC (RCCL needs it)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
# integer issue: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every 2 cycles (MI355X_MICROARCH.md,
# "Wave scheduling"), at the 2.4 GHz peak clock
VALU_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-other", action="store_true", help="skip configs 3 and 5 (other_configs)")
    ap.add_argument("--only", default="", help="run ONE of the other configs (c3, c5, c4_block, ref_default, realistic) without the headline workload and print its JSON: for profiler passes")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch path only: gloo ranks, a tiny host-side modset each, the histogram all-reduce (no GPU)")
    return ap.parse_args()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(args):
    """--gpus N from a plain `python bench.py`: N ranks as children of this process, which has not imported torch
    or touched HIP (a process that has initialised the GPU must never be replaced or forked into ranks)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.dry_launch:
        return dry_rank(args)
    return gpu_rank(args)


# ------------------------------------------------------------------------------------------------
# --dry-launch: what a rank does around the GPU work, on CPU (gloo)

def dry_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MODGPU_NO_TORCH"] = "0"
    import modimizer_amd as mg
    from modimizer_amd import synth
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = mg.lib()
    sh = mg.seqhashCreate(21, 64, 17)
    ms = mg.modsetCreate(sh, 20)
    # the rank's "block": 3000 + 100*rank pseudo k-mers with repeats, through the reference's scalar insert
    # (modset.c:45-62 + modutils.c:26 on the host arrays: nothing here needs a device)
    n = 3000 + 100 * rank
    kmers = synth.splitmix64(np.arange(n, dtype=np.uint64) % np.uint64(1000 + 10 * rank) + np.uint64(rank << 20)) & np.uint64((1 << 42) - 1)
    for km in kmers:
        ix = L.modsetIndexFind(ms, int(km), 1)
        dd = (int(ms.contents.depth[ix]) + 1) & 0xffff
        ms.contents.depth[ix] = dd if dd else 0xffff
    depth = np.ctypeslib.as_array(ms.contents.depth, (ms.contents.max + 1,))[1:]
    local = torch.from_numpy(np.bincount(depth, minlength=65536).astype(np.int64))
    hist = local.clone()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hist.copy_(local); dist.all_reduce(hist)
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = torch.tensor([int(local.sum()), n], dtype=torch.int64); dist.all_reduce(tot)
    if rank == 0:
        print(json.dumps({"metric": "Gbp/s hashed+sketched (k=21,d=64)", "dry_launch": True, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "backend": "gloo",
                          "histogram_entries": int(hist.sum()), "entries_all_ranks": int(tot[0]),
                          "kmers_all_ranks": int(tot[1]), "ms_per_step": round(float(tmax) / max(args.steps, 1) * 1e3, 3)}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# the real thing

class Ctx:
    pass


def read_profile(L, mg):
    out = {}
    for i in range(L.mgProfileKernels()):
        name = C.c_char_p(); ms_tot = C.c_double(); n = C.c_uint64()
        mg.check(L.mgProfileGet(i, C.byref(name), C.byref(ms_tot), C.byref(n)))
        if n.value:
            out[name.value.decode()] = (ms_tot.value, n.value, i)
    return out


def make_reads(cx, total, genome, genome_bases, plan_seed, err, err_seed, plan=None):
    """device-resident packed reads drawn from `genome` (device, packed): returns (reads, d_offsets, offsets, n_reads)"""
    import numpy as np
    torch, L, mg, synth = cx.torch, cx.L, cx.mg, cx.synth
    starts, offsets, strands = plan if plan is not None else synth.ont_read_plan(total, genome_bases, seed=plan_seed)
    n_reads = len(starts)
    d_starts = torch.from_numpy(starts.view(np.int64)).to(cx.dev)
    d_offsets = torch.from_numpy(offsets.view(np.int64)).to(cx.dev)
    d_strands = torch.from_numpy(strands).to(cx.dev)
    reads = torch.empty(L.mgPackedWords(total), dtype=torch.int32, device=cx.dev)
    mg.check(L.mgSynthReads(genome.data_ptr(), genome_bases, d_starts.data_ptr(), d_offsets.data_ptr(),
                            d_strands.data_ptr(), n_reads, total, err, err_seed, reads.data_ptr(), cx.stream))
    torch.cuda.synchronize()
    return reads, d_offsets, offsets, n_reads


def make_genome(cx, genome_bases, seed):
    genome = cx.torch.empty(cx.L.mgPackedWords(genome_bases), dtype=cx.torch.int32, device=cx.dev)
    cx.mg.check(cx.L.mgSynthGenome(genome.data_ptr(), genome_bases, seed, cx.stream))
    return genome


def alg_bytes_table(total, S, entries, d, slots, k=21):
    """algorithmic bytes per launch: SURVEY §8(d)'s per-unit figures x the units a launch processes"""
    import math
    packed = 2 * k - 8 + max(1, math.ceil(math.log2(max(S, 2)))) <= 64      # the partition's one-word element format (mg_table.hip)
    return {
        "mgScanKernel": (0.25 + 8.0 / d) * total,    # 2-bit read + the 8-byte k-mer per modimizer (this path needs no pos)
        "mgSegCompactKernel": 16.0 * S,              # kmer read + written
        "mgPartHistKernel": 1.0 * S,               # (round 4) the second pass counts its digits from the bytes the first pass leaves: 1 byte per modimizer (8 before)
        "mgPartScatterKernel": (16.5 if packed else 24.5) * S,      # one 8-byte word (or k-mer 8 + ordinal 4) read and written, per pass; the first of the two also writes a digit byte
        "mgBucketDedupKernel": (8.0 if packed else 12.0) * S + 16.0 * entries,
        "mgRankAssignKernel": 9.0 * S + 8.0 * entries,
        "mgRankLookupKernel": 8.0 * entries,         # a unique's ordinal read, its index written
        "mgBucketMergeKernel": 16.0 * entries + 16.0 * slots,   # uniques in, table buckets out
        "mgTableInsertKernel": 16.0 * S,
        "mgTableFindKernel": 24.0 * S,               # kmer 8 read + one 16-byte slot probed + (index) 4 written ~ SURVEY's 24*S
        "mgTableFindSegKernel": 24.0 * S,            # the same lookups, k-mers read from the scan's segments
        "mgBucketFindKernel": 16.0 * S + 16.0 * slots,   # partitioned lookups: an element read and rewritten, the table's buckets once
        "mgUnpartKernel": 28.0 * S,                  # both pulls: (position, index) 8 in + 4 out, then element 8 + index 4 in, index 4 out
    }


# What really bounds a kernel, where it is not HBM bytes, with the ceiling measured by the microbenchmarks
# under tools/ (profiles/r02_ubench*): the achieved fraction OF THAT ceiling goes into roofline.bound_actual.
RANDOM_LOADS_PER_S_BIG = 55e9        # tools/ubench_rand: random 16-byte loads from a footprint above 64 MB (Infinity Cache or HBM alike)
RANDOM_LOADS_PER_S_L2 = 250e9        # the same from a footprint inside one XCD's L2 (<= 4 MB): 235-260 G/s


def bound_actual(kernel, avg_ms, units, alu=None):
    """units: what the kernel does per launch in the unit of its real bound (probes, lookups)"""
    if kernel == "mgScanKernel":
        if not alu or not alu.get("issue_frac"):
            return {"bound": "valu_issue"}
        return {"bound": "valu_issue", "ceiling": alu["issue_peak_wave_insts_per_s"], "unit": "wave VALU instructions/s (measured, tools/ubench.hip)",
                "achieved": round(alu["valu_per_start"] * units / 64 / (avg_ms * 1e-3), 1), "frac_of_ceiling": alu["issue_frac"]}
    if kernel in ("mgTableFindKernel", "mgTableFindSegKernel"):
        ach = units / (avg_ms * 1e-3)
        return {"bound": "random_access", "ceiling": RANDOM_LOADS_PER_S_BIG, "unit": "random 16-byte loads/s, footprint > 64 MB (measured, tools/ubench_rand.hip)",
                "achieved": round(ach, 1), "frac_of_ceiling": round(ach / RANDOM_LOADS_PER_S_BIG, 3),
                "note": "achieved counts one load per lookup; a lookup that collides probes again (about 1.3 probes per lookup at load 0.6)"}
    if kernel == "mgRankLookupKernel":
        ach = units / (avg_ms * 1e-3)
        return {"bound": "random_access", "ceiling": RANDOM_LOADS_PER_S_L2, "unit": "random 16-byte loads/s, footprint inside one XCD's L2 (measured, tools/ubench_rand.hip)",
                "achieved": round(ach, 1), "frac_of_ceiling": round(ach / RANDOM_LOADS_PER_S_L2, 3)}
    return None


def time_steps(cx, step, steps, warmup, multi):
    """W untimed warm-up steps (every launch bracketed: the per-kernel table and the dominant kernel), then exactly
    K steps between barrier + synchronize on both sides with only the dominant kernel bracketed (an event pair costs
    a few microseconds of stream time per launch).  Returns (seconds, events of the dominant kernel, per-kernel table)."""
    torch, dist, L, mg = cx.torch, cx.dist, cx.L, cx.mg
    L.mgProfileOnly(-1); L.mgProfileEnable(1); L.mgProfileReset()
    # the dominant kernel is taken from the LAST warm-up step alone (one more untimed step when W < 2), not from the first ones: those hold
    # what happens once (first allocations, the transfer team's streams being made on their own thread: a launch that waits behind one of
    # these looks like a 40 ms kernel)
    for i in range(max(warmup, 2)):
        if i == max(warmup, 2) - 1:
            torch.cuda.synchronize(); L.mgProfileReset()
        step()
    torch.cuda.synchronize()
    warm = read_profile(L, mg)
    dom_id = max(warm.values(), key=lambda v: v[0])[2] if warm else -1
    L.mgProfileReset(); L.mgProfileOnly(dom_id)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cx.dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kern = read_profile(L, mg)
    # one more step, outside the timed region, with every launch bracketed: the per-kernel table
    L.mgProfileOnly(-1); L.mgProfileReset()
    step()
    torch.cuda.synchronize()
    table = read_profile(L, mg)
    L.mgProfileEnable(0)
    return dt, kern, table


def scan_alu(starts, scan_ms):
    """the integer-issue bound of mgScanKernel<FAST> (k=21 d=64): VALU instructions per start from the round's --pmc pass
    (profiles/scan_issue.json), the issue peak measured by tools/ubench.hip, the fraction of it this run's scan time means"""
    if not scan_ms:
        return None
    ipath = os.path.join(HERE, "profiles", "scan_issue.json")     # SQ_INSTS_VALU etc. of mgScanKernel from a --pmc pass
    valu_per_start, src, peak = None, None, VALU_WAVE_INSTS_PER_S
    if os.path.exists(ipath):
        try:
            ij = json.load(open(ipath)); valu_per_start = ij.get("valu_per_start"); src = ij.get("_from")
            peak = ij.get("measured_int_valu_peak_wave_insts_per_s", peak)     # tools/ubench.hip: 37.6 T integer lane-ops/s
        except Exception:
            pass
    floor7 = 7.0 * starts / 64 / peak * 1e3                        # the 7-instruction candidate filter alone
    return {"valu_per_start": valu_per_start, "valu_per_start_from": src,
            "filter_floor_valu_per_start": 7, "floor_ms": round(floor7, 3),
            "issue_peak_wave_insts_per_s": peak,
            "issue_peak_from": "measured integer VALU rate (v_mul_lo_u32 / v_alignbit / v_min / v_add chains, tools/ubench.hip); "
                               "the fp32 datasheet rate would be %.3g" % VALU_WAVE_INSTS_PER_S,
            "issue_frac": (round(valu_per_start * starts / 64 / peak / (scan_ms * 1e-3), 3) if valu_per_start else None),
            "scan_ms": round(scan_ms, 4)}


def best_of_two(cx, step, steps):
    """the other configs (never the headline `value`): two timed regions of `steps` steps each, the faster one reported and both
    listed -- after host-heavy legs (CPU baseline, drop-in programs) the first kernel of a call is now and then dispatched tens of
    milliseconds late on some boxes, which a single region of ten steps shows as a 20 % slower config"""
    a = time_steps(cx, step, steps, 1, False)
    b = time_steps(cx, step, steps, 1, False)
    best = a if a[0] <= b[0] else b
    return best[0], best[1], best[2], [round(a[0] / steps * 1e3, 3), round(b[0] / steps * 1e3, 3)]


def roofline_of(kern, table, alg_bytes, tag, extra=None, units=None):
    """units: {kernel: work items per launch in the unit of the kernel's real bound} for bound_actual"""
    if not kern:
        return None
    dom = max(kern.items(), key=lambda kv: kv[1][0])[0]
    avg_ms = kern[dom][0] / kern[dom][1]
    ab = alg_bytes.get(dom, 0.0)
    ach = ab / (avg_ms * 1e-3) / 1e9
    traffic, tfrom = None, None
    tpath = os.path.join(HERE, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch from an EARLIER rocprofv3 run
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(dom, {}).get(tag)
            if traffic is not None:
                tfrom = "profiles/traffic.json (%s; separate rocprofv3 --pmc passes of an earlier run of this command, not this run)" % tj.get("_from", "round profile")
        except Exception:
            traffic = None
    r = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_from": tfrom,
         "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": ab,
         "kernels_ms_per_step": {k: round(v[0], 4) for k, v in sorted(table.items())},
         "kernels_hbm_frac": {k: round(alg_bytes[k] * max(v[1], 1) / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                              for k, v in sorted(table.items()) if alg_bytes.get(k) and v[0] > 0},   # algorithmic bytes of all its launches in a step / its time in the step / 8 TB/s
         "kernels_ms_per_step_from": "one extra step after the timed region, every launch bracketed"}
    if extra:
        r.update(extra)
    ba = bound_actual(dom, avg_ms, (units or {}).get(dom, 0), (extra or {}).get("alu"))
    if ba:
        r["bound_actual"] = ba
    others = {}
    for kname, v in table.items():                    # the kernels of the step that sit on another bound than HBM bytes
        if kname != dom and (units or {}).get(kname) and v[0] > 0:
            o = bound_actual(kname, v[0] / max(v[1], 1), units[kname])
            if o:
                others[kname] = o
    if others:
        r["bound_actual_other_kernels"] = others
    return r


def gpu_rank(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import modimizer_amd as mg
    from modimizer_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # MODGPU_BENCH_ONE_GPU=1: every rank on cuda:0 with the gloo backend (RCCL refuses two ranks on one device) — the
    # world > 1 code path (shares, rank != 0 branches, barriers, collectives) checked on a one-GPU box; its rates mean nothing
    one_gpu = os.environ.get("MODGPU_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    cx = Ctx()
    cx.torch, cx.dist, cx.mg, cx.synth = torch, dist, mg, synth
    cx.dev = torch.device("cuda", local_rank)
    cx.L = L = mg.lib()
    mg.check(L.mgSetDevice(local_rank))
    # MODGPU_BENCH_FORCE_DIST=1: run the multi-rank code path (process group, histogram all-reduce, barriers)
    # with whatever WORLD_SIZE the launcher gave, even 1 — for checking that path on a one-GPU box
    multi = world > 1 or os.environ.get("MODGPU_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=cx.dev)
        # RCCL writes its version banner to stdout (C stdio, block-buffered on a pipe) when the communicator
        # comes up: bring it up and push the banner out now, so that rank 0's JSON line is the last line of stdout
        dist.barrier()
        torch.cuda.synchronize()
        C.CDLL(None).fflush(None)

    cx.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if args.only:                                     # a profiler pass over one of the other configs (N = 1)
        fn = {"c3": bench_c3, "c5": bench_c5, "c4_block": bench_c4_block, "ref_default": bench_ref_default, "realistic": bench_realistic}[args.only]
        print(json.dumps({"only": args.only, "result": fn(cx, args)}), flush=True)
        return
    k, d, seed, bits = 21, 64, 17, int(os.environ.get("MODGPU_BENCH_BITS", "30"))
    # config 2 at N = 1; one 12.5 Gbp block of config 4's 100 Gbp set per GPU otherwise
    gbp = float(os.environ.get("MODGPU_BENCH_GBP", "12.5" if multi else "10"))
    total = int(gbp * 1e9)
    blocks = 8 if multi else 1                                   # the config-4 set is always the 8-block one
    genome_bases = max(int(total * blocks / 30), 1_000_000)
    err = 0.05
    cx.stream = stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- synthetic reads, generated in HBM -------------------------------------------------
    t_gen = time.time()
    genome = make_genome(cx, genome_bases, 12345)
    reads, d_offsets, offsets, n_reads = make_reads(cx, total, genome, genome_bases, 1000 + rank, err, 777 + rank)
    del genome
    t_gen = time.time() - t_gen

    sh = mg.seqhashCreate(k, d, seed)
    ms = mg.modsetCreate(sh, bits)
    hist = torch.zeros(65536, dtype=torch.int64, device=cx.dev)
    local_hist = torch.zeros(65536, dtype=torch.int64, device=cx.dev)
    n_hash = C.c_uint64(0)

    def step():
        mg.check(L.mgModsetClear(ms, stream))
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads,
                                    C.byref(n_hash), stream))
        if multi:
            hist.zero_()
            mg.check(L.modsetDepthHistogramDevice(ms, hist.data_ptr(), stream))
            dist.all_reduce(hist)

    dt, kern, table = time_steps(cx, step, args.steps, args.warmup, multi)
    S = n_hash.value
    entries = ms.contents.max
    slots = float(L.mgModsetDeviceSlots(ms))
    alg = alg_bytes_table(total, S, entries, d, slots)

    # the scan's two bounds (SURVEY §8(d) asks for both): HBM bytes and integer issue
    scan_ms = table.get("mgScanKernel", (0, 1))[0] / max(table.get("mgScanKernel", (0, 1))[1], 1)
    starts = float(total)
    alu = scan_alu(starts, scan_ms)
    extra = {"alu": alu,
             "scan_bytes": {"this_path_8B_per_modimizer": alg["mgScanKernel"],
                            "survey_12B_per_modimizer": (0.25 + 12.0 / d) * total,
                            "read_only_0.25B_per_base": 0.25 * total,
                            "scan_GBps_this_path": round(alg["mgScanKernel"] / (scan_ms * 1e-3) / 1e9, 1) if scan_ms else None,
                            "scan_read_only_frac": round(0.25 * total / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if scan_ms else None},
             "whole_step": {"bytes_per_base": 0.25 + 28.0 / d,
                            "GBps": round((0.25 + 28.0 / d) * total * args.steps / dt / 1e9, 1),
                            "frac": round((0.25 + 28.0 / d) * total * args.steps / dt / 1e9 / HBM_PEAK_GBS, 4)}}
    roofline = roofline_of(kern, table, alg, "%g" % gbp, extra, units={"mgScanKernel": starts, "mgRankLookupKernel": float(entries)})

    value = world * total * args.steps / dt / 1e9
    scan_step_ms = sum(v[0] for kname, v in table.items() if kname in ("mgScanKernel", "mgSegScanKernel", "mgSegCompactKernel", "mgTileInfoKernel"))
    out = {
        "metric": "Gbp/s hashed+sketched (k=21,d=64)", "value": round(value, 3), "unit": "Gbp/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_steps_before_the_timed_region": max(args.warmup, 2),
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": ("BASELINE config 4: block %s of the 100 Gbp synthetic ONT set (8 contiguous blocks of %g Gbp of reads from one "
                                "%d Mbp genome, log-normal N50 20 kb, 5%% subs), one block per GPU, k=21 d=64 seed=17, per-GPU seqhash scan + "
                                "modset build (table bits %d) + RCCL all-reduce of the depth histogram"
                                % ("r on rank r" if world > 1 else "0", gbp, genome_bases // 1_000_000, bits)) if multi else
                               ("BASELINE config 2: %g Gbp synthetic ONT-like reads (log-normal N50 20 kb, 5%% subs, 30x of a %d Mbp genome), "
                                "k=21 d=64 seed=17, seqhash scan + modset build (table bits %d)" % (gbp, genome_bases // 1_000_000, bits)),
                   "reads_per_gpu": n_reads, "bases_per_gpu": total, "modimizers_per_gpu": S,
                   "modset_entries": entries, "k": k, "d": d, "table_bits": bits,
                   "parallelism": "reads sharded x%d, modset per GPU" % world},
        "roofline": roofline,
        "scan_only": {"ms": round(scan_step_ms, 4), "Gbp_per_s": round(total / (scan_step_ms * 1e-3) / 1e9, 1) if scan_step_ms else None,
                      "what": "tile info + scan (which also counts the first partition digit) + segment scan, by HIP events; the modimizers stay in the "
                              "per-worker segments, where the build reads them (a dense (read,pos)-ordered copy, as the query path makes, is mgSegCompactKernel: +0.6 ms)"},
        "setup_s": round(t_gen, 2),
    }

    if multi:
        # the collective by itself, and a check of what it produced
        local_hist.zero_()
        mg.check(L.modsetDepthHistogramDevice(ms, local_hist.data_ptr(), stream))
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            hist.copy_(local_hist); dist.all_reduce(hist)
        torch.cuda.synchronize(); dist.barrier()
        ar_ms = (time.perf_counter() - t0) / reps * 1e3
        sums = torch.tensor([int(local_hist.sum().item()), entries], dtype=torch.int64, device=cx.dev)
        dist.all_reduce(sums)
        ok = int(hist.sum().item()) == int(sums[0].item()) and int(sums[0].item()) == int(sums[1].item())
        if world == 1:
            ok = ok and bool(torch.equal(hist, local_hist))
        out["collective"] = {"what": "all_reduce(SUM) of the 65536 x int64 depth histogram (512 KiB) over RCCL",
                             "allreduce_ms": round(ar_ms, 4), "histogram_entries": int(hist.sum().item()),
                             "entries_all_ranks": int(sums[1].item()), "matches_local_sums": ok}
        # every rank checks its own block's result (outside the timed region): the modset of the first ~20 Mbp of the block,
        # built on its own by mgAddReadsDevice, must be the PREFIX of the block's modset -- same k-mers at the same indices
        # (indices are handed out in order of first occurrence: modset.c:57), depths no larger -- and the block's depths must
        # add up to its modimizer count.  (The bit-exactness of a build against the oracle is what tests/ pins; this ties every
        # rank's full-size result to it on the GPU it ran on.)
        par = rank_parity(cx, ms, reads, d_offsets, offsets, n_reads, S, k, d, seed)
        flags = torch.tensor([1 if par["ok"] else 0], dtype=torch.int64, device=cx.dev)
        dist.all_reduce(flags)
        out["per_rank_parity"] = bool(int(flags.item()) == world)
        out["per_rank_parity_rank0"] = par
        # what ONE GPU does on this workload with the others idle (rank 0's block, no all-reduce in the step): the figure a
        # scaling efficiency is to be read against
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:
            def step1():
                mg.check(L.mgModsetClear(ms, stream))
                mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), stream))
                hist.zero_()
                mg.check(L.modsetDepthHistogramDevice(ms, hist.data_ptr(), stream))
            step1(); torch.cuda.synchronize()
            k1 = max(3, min(args.steps, 10))
            t0 = time.perf_counter()
            for _ in range(k1):
                step1()
            torch.cuda.synchronize()
            out["single_gpu_block_gbps"] = round(total * k1 / (time.perf_counter() - t0) / 1e9, 2)
        dist.barrier()

    # ---- CPU baseline (rank 0, N=1 only): the compiled reference on a bounded sample ---------
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(cx, reads, offsets, k, d, seed)

    if rank == 0 and world == 1 and not multi and not args.no_cpu:
        try:
            out["end_to_end"] = end_to_end(cx, reads, offsets, k, d, seed)
        except Exception as e:
            out["end_to_end"] = {"error": str(e)[:300]}
        try:
            out["end_to_end"]["sync_to_host"] = sync_to_host(cx, ms, step, S, entries)
        except Exception as e:
            out["end_to_end"]["sync_to_host"] = {"error": str(e)[:300]}
        try:
            out["end_to_end"]["write_mod"] = write_mod(cx, ms)
        except Exception as e:
            out["end_to_end"]["write_mod"] = {"error": str(e)[:300]}

    n4 = None
    if rank == 0 and world == 1 and not multi and not args.no_other:
        try:
            n4 = bench_minimizers(cx, reads, d_offsets, offsets, n_reads)
        except Exception as e:
            n4 = {"error": str(e)[:300]}
    L.modsetDestroy(ms)
    del reads, d_offsets
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not multi and not args.no_cpu and os.environ.get("MODGPU_BENCH_LONGFILES", "1") == "1":
        try:
            out["end_to_end"]["modmap_query_file_long"] = modmap_query_file_long(cx)
        except Exception as e:
            out["end_to_end"]["modmap_query_file_long"] = {"error": str(e)[:300]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not multi and not args.no_other:
        other = {}
        for name, fn in (("c4_block", bench_c4_block), ("c5", bench_c5), ("c3", bench_c3), ("ref_default", bench_ref_default), ("realistic", bench_realistic)):
            if name not in os.environ.get("MODGPU_BENCH_OTHER", "c4_block,c5,c3,ref_default,realistic").split(","):      # dev: run a subset
                continue
            try:
                other[name] = fn(cx, args)
            except Exception as e:                                # the headline line must still go out
                other[name] = {"error": str(e)[:300]}
        if n4 is not None:
            other["n4_minimizers"] = n4
        out["other_configs"] = other

    if multi and not args.no_other:
        # north_star's other sharding (query reads over replicated modsets), every rank taking part
        try:
            c3s = bench_c3(cx, args, shard=(rank, world))
        except Exception as e:
            c3s = {"error": str(e)[:300]}
        out["other_configs"] = {"c3_sharded": c3s}
    if multi:
        dist.barrier()                    # every rank is done (and silent) before the line goes out
        C.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if multi:
        dist.destroy_process_group()


def rank_parity(cx, ms, reads, d_offsets, offsets, n_reads, S, k, d, seed):
    """see the call site: prefix property of this rank's block on this rank's GPU"""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    m = max(1, min(int(np.searchsorted(offsets, 20_000_000, side="right")) - 1, n_reads))
    nb = int(offsets[m])
    sh = mg.seqhashCreate(k, d, seed)
    ms2 = mg.modsetCreate(sh, 24)
    n2 = C.c_uint64(0)
    mg.check(L.mgAddReadsDevice(ms2, reads.data_ptr(), nb, d_offsets.data_ptr(), m, C.byref(n2), cx.stream))
    mg.check(L.modsetSyncToHost(ms2, 0)); mg.check(L.modsetSyncToHost(ms, 0))
    u2, u = ms2.contents.max, ms.contents.max
    v2 = np.ctypeslib.as_array(ms2.contents.value, (u2 + 1,)); d2 = np.ctypeslib.as_array(ms2.contents.depth, (u2 + 1,))
    v = np.ctypeslib.as_array(ms.contents.value, (u + 1,)); dd = np.ctypeslib.as_array(ms.contents.depth, (u + 1,))
    ok = bool(0 < u2 <= u and np.array_equal(v[1:u2 + 1], v2[1:]) and np.all(dd[1:u2 + 1] >= d2[1:])
              and int(d2[1:].astype(np.int64).sum()) == n2.value
              and (int(dd[1:].astype(np.int64).sum()) == S or int(dd.max()) == 65535))
    res = {"ok": ok, "prefix_bases": nb, "prefix_entries": int(u2), "block_entries": int(u)}
    L.modsetDestroy(ms2)
    return res


# ------------------------------------------------------------------------------------------------
# BASELINE configs 5 and 3 (N = 1)

def bench_minimizers(cx, reads, d_offsets, offsets, n_reads):
    """minimizerRCiterator / minimizerRCnext (seqhash.c:83-152; no caller in the reference) run to exhaustion on every read of
    the first ~2 Gbp of the headline's batch, device resident, at the reference's default k = 19, w = 31: count pass + scan + write pass."""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    want = int(float(os.environ.get("MODGPU_BENCH_MIN_GBP", "2")) * 1e9)
    n = max(1, min(int(np.searchsorted(offsets, want, side="right")) - 1, n_reads))
    total = int(offsets[n])
    k, w = 19, 31
    sh = mg.seqhashCreate(k, w, 17)
    cap = int(total / (w / 2 + 1) + total / 8 + n + 1024)
    dH = torch.empty(cap, dtype=torch.int64, device=cx.dev); dP = torch.empty(cap, dtype=torch.int32, device=cx.dev)
    dS = torch.empty(n + 2, dtype=torch.int64, device=cx.dev)
    nm = C.c_uint64()
    best = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mg.check(L.seqhashMinimizerBatchDevice(sh, reads.data_ptr(), total, d_offsets.data_ptr(), n, dH.data_ptr(), dP.data_ptr(), dS.data_ptr(), cap, C.byref(nm), cx.stream))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it:
            best = dt if best is None else min(best, dt)
    m = nm.value
    pos = (dP[:m] & 0x7fffffff).to(torch.int64)
    st = dS[:n + 1]
    # every read's positions strictly increase, and no step is longer than w (the next window starts right behind the last minimum)
    d = pos[1:] - pos[:-1]
    first = torch.zeros(m, dtype=torch.bool, device=cx.dev); first[st[:-1][st[:-1] < m]] = True
    inner = ~first[1:]
    ok = bool((d[inner] > 0).all().item()) and bool((d[inner] <= w).all().item()) and int(st[-1].item()) == m
    return {"entry": "seqhashMinimizerBatchDevice", "k": k, "w": w, "bases": total, "reads": n, "minimizers": int(m), "ms": round(best * 1e3, 2),
            "Gbp_per_s": round(total / best / 1e9, 1), "bases_per_minimizer": round(total / max(m, 1), 2), "checks_ok": ok,
            "what": "a wave per read; tiles of 496 positions staged in LDS (hashes by all lanes, prefix / suffix arg-minima per block of w, next[] per position), "
                    "the chain walked one LDS read a link and written by all lanes; two passes (count, write). Round 4: 24 Gbp/s, every link a wave-wide window scan from global memory"}


def bench_c4_block(cx, args):
    """What ONE GPU does at N > 1 (configs[3]): block 0 of the 100 Gbp set — 12.5 Gbp of reads from the 3.33 Gbp genome —
    scan + modset build + depth histogram, without the collective.  N x this is what `--gpus N` can reach; it is not
    config 2 (3.75x coverage per block: 1.66e8 distinct modimizers against 1.03e8), so a scaling efficiency taken against
    the N = 1 line (config 2) starts at this ratio."""
    torch, L, mg = cx.torch, cx.L, cx.mg
    k, d, bits = 21, 64, int(os.environ.get("MODGPU_BENCH_BITS", "30"))
    total = int(12.5e9)
    genome_bases = int(total * 8 / 30)
    genome = make_genome(cx, genome_bases, 12345)
    reads, d_offsets, offsets, n_reads = make_reads(cx, total, genome, genome_bases, 1000, 0.05, 777)
    del genome
    sh = mg.seqhashCreate(k, d, 17)
    ms = mg.modsetCreate(sh, bits)
    hist = torch.zeros(65536, dtype=torch.int64, device=cx.dev)
    n_hash = C.c_uint64(0)

    def step():
        mg.check(L.mgModsetClear(ms, cx.stream))
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
        hist.zero_()
        mg.check(L.modsetDepthHistogramDevice(ms, hist.data_ptr(), cx.stream))

    steps = max(3, min(args.steps, 10))
    dt, kern, table, regions = best_of_two(cx, step, steps)
    S, entries = n_hash.value, ms.contents.max
    alg = alg_bytes_table(total, S, entries, d, float(L.mgModsetDeviceSlots(ms)))
    res = {"timed_regions_ms_per_step": regions,
           "workload": "one GPU's share of BASELINE config 4 on this GPU alone: block 0 of 8 (12.5 Gbp of reads from the %d Mbp genome), "
                       "k=21 d=64, table bits %d: seqhash scan + modset build + depth histogram (no all-reduce)" % (genome_bases // 1_000_000, bits),
           "value": round(total * steps / dt / 1e9, 2), "unit": "Gbp/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "bases": total, "modimizers": S, "modset_entries": entries, "histogram_entries": int(hist.sum().item()),
           "roofline": roofline_of(kern, table, alg, "12.5", {"alu": scan_alu(float(total), table.get("mgScanKernel", (0, 1))[0] / max(table.get("mgScanKernel", (0, 1))[1], 1))},
                                   units={"mgScanKernel": float(total), "mgRankLookupKernel": float(entries)})}
    L.modsetDestroy(ms)
    del reads, d_offsets
    torch.cuda.empty_cache()
    return res


def bench_c5(cx, args):
    """configs[4]: modutils depth histogram on 50x synthetic Illumina 150 b reads, k=31 d=4 (20 Mbp genome,
    6 666 667 reads, 0.5 % substitutions, table bits 28).  Step = clear + scan + build + depth histogram (modutils.c:19-63)."""
    torch, L, mg, synth = cx.torch, cx.L, cx.mg, cx.synth
    k, d, bits = 31, 4, 28
    scale = float(os.environ.get("MODGPU_BENCH_C5_SCALE", "1"))
    genome_bases = int(20_000_000 * scale)
    n_reads = int(6_666_667 * scale)
    total = n_reads * 150
    genome = make_genome(cx, genome_bases, 555)
    plan = synth.fixed_read_plan(n_reads, 150, genome_bases, 556)
    reads, d_offsets, offsets, _ = make_reads(cx, total, genome, genome_bases, 0, 0.005, 557, plan=plan)
    del genome
    sh = mg.seqhashCreate(k, d, 17)
    ms = mg.modsetCreate(sh, bits)
    hist = torch.zeros(65536, dtype=torch.int64, device=cx.dev)
    n_hash = C.c_uint64(0)

    def step():
        mg.check(L.mgModsetClear(ms, cx.stream))
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
        hist.zero_()
        mg.check(L.modsetDepthHistogramDevice(ms, hist.data_ptr(), cx.stream))

    steps = max(3, min(args.steps, 10))
    dt, kern, table, regions = best_of_two(cx, step, steps)
    S, entries = n_hash.value, ms.contents.max
    alg = alg_bytes_table(total, S, entries, d, float(L.mgModsetDeviceSlots(ms)), k)
    h = hist.cpu().numpy()
    res = {"timed_regions_ms_per_step": regions,
           "workload": "BASELINE config 5: %d x 150 b reads (50x of a %d Mbp genome, 0.5%% subs), k=31 d=4 seed=17, table bits %d: "
                       "seqhash scan + modset build + depth histogram" % (n_reads, genome_bases // 1_000_000, bits),
           "value": round(total * steps / dt / 1e9, 2), "unit": "Gbp/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "bases": total, "modimizers": S, "modset_entries": entries,
           "histogram": {"entries": int(h.sum()), "depth_sum_le_modimizers": bool(int((h * range(65536)).sum()) <= S),
                         "mode_depth": int(h[1:].argmax()) + 1},
           "whole_step": {"bytes_per_base": 0.25 + 28.0 / d, "GBps": round((0.25 + 28.0 / d) * total * steps / dt / 1e9, 1),
                          "frac": round((0.25 + 28.0 / d) * total * steps / dt / 1e9 / HBM_PEAK_GBS, 4)},
           "roofline": roofline_of(kern, table, alg, "c5")}
    L.modsetDestroy(ms)
    del reads, d_offsets
    torch.cuda.empty_cache()
    return res


def bench_ref_default(cx, args):
    """The reference's OWN default parameters (modmap.c:314-317, modutils.c:140: k = 19, w = 31, seed 17) on config 2's reads:
    w is not a power of two, so the scan is the exact-mode kernel for an odd modulus (mgScanKernel<MG_MODE_ODD32>: both 64-bit
    hashes at every start; divisibility by 31 without a division, in 32-bit arithmetic because the hash is 38 bits: mgDivisibleOdd32).  Step = clear + scan + build, as the headline."""
    torch, L, mg = cx.torch, cx.L, cx.mg
    k, w, bits = 19, 31, int(os.environ.get("MODGPU_BENCH_BITS", "30"))
    total = int(float(os.environ.get("MODGPU_BENCH_GBP", "10")) * 1e9)
    genome_bases = max(total // 30, 1_000_000)
    genome = make_genome(cx, genome_bases, 12345)
    reads, d_offsets, offsets, n_reads = make_reads(cx, total, genome, genome_bases, 1000, 0.05, 777)     # the headline's reads
    del genome
    sh = mg.seqhashCreate(k, w, 17)
    ms = mg.modsetCreate(sh, bits)
    n_hash = C.c_uint64(0)

    def step():
        mg.check(L.mgModsetClear(ms, cx.stream))
        mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))

    steps = max(3, min(args.steps, 10))
    dt, kern, table, regions = best_of_two(cx, step, steps)
    S, entries = n_hash.value, ms.contents.max
    alg = alg_bytes_table(total, S, entries, w, float(L.mgModsetDeviceSlots(ms)), k)
    scan_ms = table.get("mgScanKernel", (0, 1))[0] / max(table.get("mgScanKernel", (0, 1))[1], 1)
    alu = scan_alu(float(total), scan_ms)
    try:
        pm = json.load(open(os.path.join(HERE, "profiles", "scan_issue.json"))).get("ref_default")
        if pm and scan_ms:
            alu = dict(alu, valu_per_start=pm["valu_per_start"], valu_per_start_from=pm["from"],
                       wave_valu_per_s=round(pm["valu_per_start"] * total / 64 / (scan_ms * 1e-3), 1),
                       issue_frac=round(pm["valu_per_start"] * total / 64 / (scan_ms * 1e-3) / alu["issue_peak_wave_insts_per_s"], 4),
                       filter_floor_valu_per_start=None, floor_ms=None,
                       note="issue_frac is against the rate of the multiply / shift / compare class (tools/ubench.hip); the exact-mode loop also holds adds and "
                            "logic operations, which issue at 1.7x that rate, so it can read a little above 1")
    except Exception:
        pass
    res = {"timed_regions_ms_per_step": regions,
           "workload": "the reference's default parameters (modmap.c:314-317: k=19 w=31 seed=17; exact-mode scan, w not a power of two) on "
                       "BASELINE config 2's reads (%.1f Gbp ONT-like, N50 20 kb, 5%% subs, 30x of a %d Mbp genome), table bits %d: "
                       "seqhash scan + modset build" % (total / 1e9, genome_bases // 1_000_000, bits),
           "value": round(total * steps / dt / 1e9, 2), "unit": "Gbp/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
           "bases": total, "modimizers": S, "modset_entries": entries,
           "whole_step": {"bytes_per_base": 0.25 + 28.0 / w, "GBps": round((0.25 + 28.0 / w) * total * steps / dt / 1e9, 1),
                          "frac": round((0.25 + 28.0 / w) * total * steps / dt / 1e9 / HBM_PEAK_GBS, 4)},
           "roofline": roofline_of(kern, table, alg, "ref_default", {"alu": alu}, units={"mgScanKernel": float(total), "mgRankLookupKernel": float(entries)})}
    L.modsetDestroy(ms)
    del reads, d_offsets
    torch.cuda.empty_cache()
    return res


def bench_realistic(cx, args):
    """What repeats cost: 1 Gbp of ONT-like reads (N50 20 kb, 5 % subs, 10x) from a 100 Mbp genome with the repeat
    structure of a real one (synth.repeat_genome: an Alu-like family with poly-A tails, satellite arrays, (CA)n) beside the same
    from an iid genome, and 0.5 Gbp of nothing but poly-A -- every start a modimizer of ONE k-mer at k=21 d=64 seed 17.  A
    k-mer's occurrences all fall into one table bucket; buckets beyond 16384 occurrences (MG_HOT_SPLIT_DEFAULT) are reduced chunk by chunk by many
    workgroups first (mgHotReduceKernel).  Step = clear + scan + build, k=21 d=64, table bits 28."""
    import numpy as np
    torch, L, mg, synth = cx.torch, cx.L, cx.mg, cx.synth
    k, d, bits = 21, 64, 28
    G = int(float(os.environ.get("MODGPU_BENCH_REALISTIC_GENOME_MBP", "100")) * 1e6)
    res = {"workload": "1 Gbp ONT-like reads (N50 20 kb, 5% subs) from a 100 Mbp genome: iid / with 10% Alu-like + poly-A tails, 3% satellite arrays, "
                       "1% (CA)n; and 0.5 Gbp of poly-A; k=21 d=64 seed=17, table bits 28: seqhash scan + modset build"}
    sh = mg.seqhashCreate(k, d, 17)
    steps = max(3, min(args.steps, 5))
    for name, host_genome, total in (("iid_genome", np.random.default_rng(11).integers(0, 4, G).astype(np.uint8), 1_000_000_000),
                                     ("repeat_genome", synth.repeat_genome(G, 12), 1_000_000_000),
                                     ("poly_a", np.zeros(1_000_000, np.uint8), 500_000_000)):
        gb = len(host_genome)
        words = np.zeros(L.mgPackedWords(gb), np.uint32)
        L.mgPackHost(host_genome.ctypes.data, gb, words.ctypes.data)
        genome = torch.from_numpy(words.view(np.int32)).to(cx.dev)
        reads, d_offsets, offsets, n_reads = make_reads(cx, total, genome, gb, 21, 0.05 if name != "poly_a" else 0.0, 22)
        del genome, host_genome
        ms = mg.modsetCreate(sh, bits)
        n_hash = C.c_uint64(0)

        def step():
            mg.check(L.mgModsetClear(ms, cx.stream))
            mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
        dt, kern, table, regions = best_of_two(cx, step, steps)
        per = {kn.replace("Kernel", "").replace("mg", ""): round(v[0], 3) for kn, v in sorted(table.items(), key=lambda kv: -kv[1][0])[:7]}
        mg.check(L.modsetSyncToHost(ms, 0))
        dep = np.ctypeslib.as_array(ms.contents.depth, (ms.contents.max + 1,))[1:]
        res[name] = {"value": round(total * steps / dt / 1e9, 2), "unit": "Gbp/s", "ms_per_Gbp": round(dt / steps * 1e3 / (total / 1e9), 3),
                     "bases": total, "modimizers": n_hash.value, "modset_entries": ms.contents.max,
                     "saturated_entries": int((dep == 65535).sum()), "kernels_ms_per_step": per}
        L.modsetDestroy(ms)
        del reads, d_offsets
        torch.cuda.empty_cache()
    res["repeats_cost"] = round(res["repeat_genome"]["ms_per_Gbp"] / res["iid_genome"]["ms_per_Gbp"], 3)
    return res


def bench_c3(cx, args, shard=None):
    """shard = (rank, world) at N > 1: north_star's query sharding -- the reference modset replicated (every rank builds it from the
    same genome), the 90 Gbp of reads split into `world` contiguous shares of whole batches, no collective on the data path; the
    timed region is bracketed by barriers and the MAX over ranks is taken.
    configs[2]: modmap — a 3 Gbp synthetic reference (24 sequences of 125 Mbp, table bits 28, modmap.c:93-134 insert
    loop) queried with 30x = 90 Gbp of ONT-like reads in batches of 10 Gbp (modmap.c:197-206 lookup loop: one seed
    (index,pos) per modimizer incl. misses).  Timed: the query batches (scan + lookup), reads resident in HBM."""
    import numpy as np
    torch, L, mg, synth = cx.torch, cx.L, cx.mg, cx.synth
    k, d, bits = 21, 64, 28
    scale = float(os.environ.get("MODGPU_BENCH_C3_SCALE", "1"))
    n_seq = 24
    seq_len = int(125_000_000 * scale)
    genome_bases = n_seq * seq_len
    batch = int(float(os.environ.get("MODGPU_BENCH_C3_BATCH_GBP", "10")) * 1e9 * min(scale * 4, 1.0))
    n_batches = int(os.environ.get("MODGPU_BENCH_C3_BATCHES", "9"))
    rank, world = shard if shard else (0, 1)
    first_batch = 0
    if shard:                                          # 90 Gbp / world per rank, in batches of at most 10 Gbp (the shares are whole numbers of equal batches)
        share = n_batches * batch // world
        per = -(-share // batch)                       # batches per rank
        batch = share // per // 16 * 16
        n_batches = per
        first_batch = rank * per
    genome = make_genome(cx, genome_bases, 333)
    ref_off = torch.arange(0, n_seq + 1, dtype=torch.int64, device=cx.dev) * seq_len
    sh = mg.seqhashCreate(k, d, 17)
    ms = mg.modsetCreate(sh, bits)
    cap = int(genome_bases / d * 1.3) + (1 << 16)
    s_idx = torch.empty(cap, dtype=torch.int32, device=cx.dev)
    s_pos = torch.empty(cap, dtype=torch.int32, device=cx.dev)
    s_rd = torch.empty(cap, dtype=torch.int32, device=cx.dev)
    n_seeds = C.c_uint64(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mg.check(L.mgInsertReadsDevice(ms, genome.data_ptr(), genome_bases, ref_off.data_ptr(), n_seq,
                                   s_idx.data_ptr(), s_pos.data_ptr(), s_rd.data_ptr(), cap, C.byref(n_seeds), cx.stream))
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    ref_occ, ref_entries = n_seeds.value, ms.contents.max
    del s_idx, s_pos, s_rd
    torch.cuda.empty_cache()
    ref_read = None
    if not shard and os.environ.get("MODGPU_BENCH_C3_REFREAD", "1") == "1":
        try:
            ref_read = reference_read_whole(cx, genome, genome_bases, n_seq, seq_len, k, d, bits, ref_occ, ref_entries)
        except Exception as e:
            ref_read = {"error": str(e)[:300]}

    qcap = int(batch / d * 1.3) + (1 << 16)
    q_pos = torch.empty(qcap, dtype=torch.int32, device=cx.dev)
    q_rd = torch.empty(qcap, dtype=torch.int32, device=cx.dev)
    # all query batches are generated first (22 GB of packed reads in HBM) and then queried back to back, as a
    # streaming caller would: nothing else touches the GPU or keeps the host busy between two timed calls
    batches = []
    for b in range(n_batches):
        reads, d_offsets, offsets, n_reads = make_reads(cx, batch, genome, genome_bases, 4000 + first_batch + b, 0.05, 5000 + first_batch + b)
        batches.append((reads, d_offsets, n_reads, torch.empty(qcap, dtype=torch.int32, device=cx.dev)))
    L.mgProfileOnly(-1); L.mgProfileEnable(1); L.mgProfileReset()
    reads, d_offsets, n_reads, q_idx = batches[0]                # warm-up of the arena and the kernels, untimed
    mg.check(L.mgQueryReadsDevice(ms, reads.data_ptr(), batch, d_offsets.data_ptr(), n_reads,
                                  q_idx.data_ptr(), q_pos.data_ptr(), q_rd.data_ptr(), qcap, C.byref(n_seeds), cx.stream))
    torch.cuda.synchronize()
    if not shard and os.environ.get("MODGPU_BENCH_C3_PIPE", "1") != "0":        # ... and of the pipeline's second arena and stream
        tk = [C.c_void_p(), C.c_void_p()]
        for j in (0, 1):
            mg.check(L.mgQueryReadsDeviceAsync(ms, reads.data_ptr(), batch, d_offsets.data_ptr(), n_reads, q_idx.data_ptr(), q_pos.data_ptr(), q_rd.data_ptr(), qcap, C.byref(tk[j]), cx.stream))
        for j in (0, 1):
            mg.check(L.mgQueryReadsDeviceWait(tk[j], C.byref(n_seeds), cx.stream))
        torch.cuda.synchronize()
    L.mgProfileReset()
    # batches that follow one another, two ways: one synchronous call a batch (the headline of this config), and pipelined -- the scan of
    # batch i + 1 started (mgQueryReadsDeviceAsync: the library's own stream, the other of two scratch arenas) before the lookups of batch i
    # are run (mgQueryReadsDeviceWait) -- reported beside it (profiles/r05_c3_pipe_trace.txt says why it gains nothing)
    q_pos2 = torch.empty(qcap, dtype=torch.int32, device=cx.dev)
    q_rd2 = torch.empty(qcap, dtype=torch.int32, device=cx.dev)

    def run_batches(pipe):
        per_, seeds_ = [], []
        torch.cuda.synchronize()
        if shard:
            cx.dist.barrier()
        t_all = time.perf_counter()
        if pipe:
            def start(i):
                reads, d_offsets, n_reads, q_idx = batches[i]
                t = C.c_void_p()
                mg.check(L.mgQueryReadsDeviceAsync(ms, reads.data_ptr(), batch, d_offsets.data_ptr(), n_reads, q_idx.data_ptr(),
                                                   (q_pos2 if i & 1 else q_pos).data_ptr(), (q_rd2 if i & 1 else q_rd).data_ptr(), qcap, C.byref(t), cx.stream))
                return t
            nxt = start(0)
            for i in range(len(batches)):
                t0 = time.perf_counter()
                cur, nxt = nxt, (start(i + 1) if i + 1 < len(batches) else None)
                mg.check(L.mgQueryReadsDeviceWait(cur, C.byref(n_seeds), cx.stream))
                per_.append(round((time.perf_counter() - t0) * 1e3, 2)); seeds_.append(n_seeds.value)
        else:
            for reads, d_offsets, n_reads, q_idx in batches:
                t0 = time.perf_counter()
                mg.check(L.mgQueryReadsDevice(ms, reads.data_ptr(), batch, d_offsets.data_ptr(), n_reads,
                                              q_idx.data_ptr(), q_pos.data_ptr(), q_rd.data_ptr(), qcap, C.byref(n_seeds), cx.stream))
                per_.append(round((time.perf_counter() - t0) * 1e3, 2))      # the call returns when its seeds are complete (it synchronises)
                seeds_.append(n_seeds.value)
        torch.cuda.synchronize()
        if shard:
            cx.dist.barrier()
        return time.perf_counter() - t_all, per_, seeds_
    pipelined = None
    if not shard and os.environ.get("MODGPU_BENCH_C3_PIPE", "1") != "0":
        L.mgProfileEnable(0)
        tp, perp, seedsp = run_batches(True)
        pipelined = {"value": round(n_batches * batch / tp / 1e9, 2), "ms_per_batch": round(tp / n_batches * 1e3, 3), "ms_each_batch": perp}
        L.mgProfileEnable(1); L.mgProfileReset()
    t_query, per, seeds_each = run_batches(False)
    if pipelined is not None:
        pipelined["same_seed_counts"] = seedsp == seeds_each
    tot_bases, tot_seeds, tot_hits = n_batches * batch, sum(seeds_each), 0
    for (reads, d_offsets, n_reads, q_idx), ns in zip(batches, seeds_each):
        tot_hits += int((q_idx[:ns] != 0).sum().item())
    if shard:                                          # whole job: all ranks' bases over the slowest rank's time
        tm = torch.tensor([t_query], dtype=torch.float64, device=cx.dev); cx.dist.all_reduce(tm, op=cx.dist.ReduceOp.MAX)
        tot = torch.tensor([tot_bases, tot_seeds, tot_hits], dtype=torch.int64, device=cx.dev); cx.dist.all_reduce(tot)
        t_job = float(tm.item()); job_bases, job_seeds, job_hits = (int(x) for x in tot.tolist())
    del batches, reads, d_offsets, q_idx, q_pos2, q_rd2
    table = read_profile(L, mg)
    L.mgProfileEnable(0)
    per_batch = {kname: (v[0] / n_batches, 1, v[2]) for kname, v in table.items()}
    S = tot_seeds / n_batches
    alg = alg_bytes_table(batch, S, ref_entries, d, float(L.mgModsetDeviceSlots(ms)))
    alg["mgScanKernel"] = (0.25 + 16.0 / d) * batch              # this path writes kmer 8 + pos 4 + read 4 per modimizer
    alg["mgSegCompactKernel"] = 16.0 * S                         # pos + read, 4 bytes each, read and written (the lookups read the k-mers from the segments)
    kern = {kname: (v[0], v[1], v[2]) for kname, v in table.items()}
    if shard:
        res = {"workload": "BASELINE config 3 sharded over %d GPUs (north_star: reads shard, modset replicated): every rank builds the reference "
                           "%d x %d Mbp (table bits %d, %d modset entries) and queries its %d batches of %.3f Gbp (scan + lookup, seeds out); "
                           "no data-path collective" % (world, n_seq, seq_len // 1_000_000, bits, ref_entries, n_batches, batch / 1e9),
               "value": round(job_bases / t_job / 1e9, 2), "unit": "Gbp/s", "n_gpus": world, "query_bases": job_bases, "seeds": job_seeds,
               "seed_hit_fraction": round(job_hits / max(job_seeds, 1), 4), "ms_each_batch_rank0": per,
               "rank0_Gbp_per_s": round(tot_bases / t_query / 1e9, 2), "reference_insert_device_s": round(t_ref, 3)}
        L.modsetDestroy(ms)
        del genome, q_pos, q_rd
        torch.cuda.empty_cache()
        return res
    res = {"workload": "BASELINE config 3: modmap, reference %d x %d Mbp = %.1f Gbp (table bits %d, %d occurrences, %d modset entries), "
                       "%d query batches of %g Gbp ONT-like reads from it (5%% subs): scan + lookup, seeds (index,pos,read) out"
                       % (n_seq, seq_len // 1_000_000, genome_bases / 1e9, bits, ref_occ, ref_entries, n_batches, batch / 1e9),
           "value": round(tot_bases / t_query / 1e9, 2), "unit": "Gbp/s", "ms_per_batch": round(t_query / n_batches * 1e3, 3),
           "pipelined": pipelined, "ms_each_batch": per, "query_bases": tot_bases, "seeds": tot_seeds, "seed_hit_fraction": round(tot_hits / max(tot_seeds, 1), 4),
           "reference_insert_device_s": round(t_ref, 3), "reference_insert_device_Gbp_per_s": round(genome_bases / t_ref / 1e9, 1),
           "reference_read": ref_read, "reference_read_s": (ref_read or {}).get("whole_call_s"),
           "whole_batch": {"bytes_per_base": 0.25 + 36.0 / d, "GBps": round((0.25 + 36.0 / d) * tot_bases / t_query / 1e9, 1),
                           "frac": round((0.25 + 36.0 / d) * tot_bases / t_query / 1e9 / HBM_PEAK_GBS, 4),
                           "note": "0.25 B/base read + per seed: 12 B written by the scan + 24 B lookup (kmer 8, one 16-byte slot, wait index 4 out)"},
           "lookup_path": "partitioned (two levels): counts + 2 scatter passes + mgBucketFindKernel + 2 pulls" if "mgBucketFindKernel" in per_batch else "direct probes in ordinal order (mgTableFindSegKernel)",
           "lookups_ms_per_batch": round(sum(v[0] for kn, v in per_batch.items() if kn in ("mgBucketFindKernel", "mgUnpartKernel", "mgPartScatterKernel", "mgPartHistKernel", "mgPartChunks+ScanKernel", "mgTableFindSegKernel", "mgTableFindKernel")), 3),
           "roofline": roofline_of(kern, per_batch, alg, "c3", units={"mgTableFindSegKernel": S, "mgTableFindKernel": S})}
    L.modsetDestroy(ms)
    del genome, q_pos, q_rd
    torch.cuda.empty_cache()
    return res


def reference_read_whole(cx, genome, genome_bases, n_seq, seq_len, k, d, bits, want_occ, want_entries):
    """What `modmap -f` does with the reference (modmap.c:93-134 + 74-91), as ONE call from host bytes: mgReferenceRead = upload +
    scan + insert + per-occurrence bookkeeping + copy classes + referencePack + every array back in the caller's Reference /
    Modset (index, offset, id, depth, rev, loc, info, value).  `reference_insert_device_s` beside it is mgInsertReadsDevice alone."""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    d_bytes = torch.empty(genome_bases, dtype=torch.uint8, device=cx.dev)
    mg.check(L.mgUnpackDevice(genome.data_ptr(), genome_bases, d_bytes.data_ptr(), cx.stream))
    torch.cuda.synchronize()
    h = d_bytes.cpu().numpy(); del d_bytes
    torch.cuda.empty_cache()
    off = (np.arange(n_seq + 1, dtype=np.int64) * seq_len)
    names = (C.c_char_p * n_seq)(*[b"chr%d" % (i + 1) for i in range(n_seq)])
    times = []
    res = {}
    for it in range(2):
        sh = mg.seqhashCreate(k, d, 17); ms = mg.modsetCreate(sh, bits)
        ref = L.mgReferenceCreate(ms, 1 << 26)
        with mg.CFile(os.devnull, "w") as fo:
            t0 = time.perf_counter()
            rc = L.mgReferenceRead(ref, h.ctypes.data, off.ctypes.data, n_seq, names, True, fo)
            times.append(time.perf_counter() - t0)
        if rc:
            raise RuntimeError("mgReferenceRead failed")
        r = C.cast(ref, C.POINTER(mg.MgReference)).contents
        n, m = r.max, ms.contents.max + 1
        if it == 0:
            rev = np.ctypeslib.as_array(r.rev, (n,)); loc = np.ctypeslib.as_array(r.loc, (m,)); ix = np.ctypeslib.as_array(r.index, (n,))
            dep = np.ctypeslib.as_array(r.depth, (m,))
            grouped = ix[rev]                                     # rev lists the occurrences index by index ...
            ok = (n == want_occ and m - 1 == want_entries and bool(np.all(np.diff(grouped.astype(np.int64)) >= 0))
                  and int(loc[-1]) + int(dep[-1]) == n and bool(np.array_equal(np.bincount(ix, minlength=m)[:m], dep)))
            same = grouped[1:] == grouped[:-1]                    # ... and inside an index in occurrence order
            ok = ok and bool(np.all(rev[1:][same] > rev[:-1][same]))
            res["checks_ok"] = ok
            del rev, loc, ix, dep, grouped, same
        L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
    res.update({"entry": "mgReferenceRead", "whole_call_s": round(min(times), 4), "first_call_s": round(times[0], 4),
                "Gbp_per_s": round(genome_bases / min(times) / 1e9, 2), "occurrences": want_occ, "modset_entries": want_entries,
                "what": "host bytes (1 per base) -> pack + H2D -> scan + insert -> occurrences appended, copy classes, loc (exclusive scan), rev "
                        "(stable radix sort by index) on the device -> index / offset / id / depth / rev / loc / info / value in the caller's arrays"})
    return res


# ------------------------------------------------------------------------------------------------

def end_to_end(cx, reads, offsets, k, d, seed):
    """PCIe- and parser-inclusive rates of the host entry points on a sample of the same reads (never `value`):
    mgAddSequenceBatch from host bytes (one base per byte, as the reference's iterator takes them: packed to 2 bits on the
    host, pinned staging, H2D, scan, build) and mgAddSequenceFile from an 80-column FASTA file in /dev/shm (parse pool
    -> pack -> H2D -> scan -> build, the next batch parsed while the GPU works on the current one)."""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    res = {}
    want = int(float(os.environ.get("MODGPU_E2E_GBP", "4")) * 1e9)
    n = max(1, min(int(np.searchsorted(offsets, want, side="right")) - 1, len(offsets) - 1))
    nb = int(offsets[n])
    d_bytes = torch.empty(nb, dtype=torch.uint8, device=cx.dev)
    mg.check(L.mgUnpackDevice(reads.data_ptr(), nb, d_bytes.data_ptr(), cx.stream))
    torch.cuda.synchronize()
    h = d_bytes.cpu().numpy(); del d_bytes
    off = offsets[:n + 1].astype(np.int64)
    sh = mg.seqhashCreate(k, d, seed); ms = mg.modsetCreate(sh, 28)
    best = None
    for it in range(3):                                   # the first call sets up the pinned staging and the device buffers
        mg.check(L.mgModsetClear(ms, None))
        t0 = time.perf_counter()
        nh = L.mgAddSequenceBatch(ms, h.ctypes.data, off.ctypes.data, n)
        dt_add = time.perf_counter() - t0
        if nh < 0:
            raise RuntimeError(L.mgLastError().decode())
        mg.check(L.modsetSyncToHost(ms, 0))               # end to end includes the D2H of the results
        dt = time.perf_counter() - t0
        if it and (best is None or dt < best):
            best, best_add = dt, dt_add
    res["host_bytes"] = {"entry": "mgAddSequenceBatch + modsetSyncToHost", "Gbp_per_s": round(nb / best / 1e9, 1), "bases": nb,
                         "Gbp_per_s_without_result_mirror": round(nb / best_add / 1e9, 1), "result_mirror_ms": round((best - best_add) * 1e3, 2),
                         "modset_entries": int(ms.contents.max),
                         "what": "1 byte per base in pageable host memory -> 2-bit pack on host threads -> pinned staging -> H2D -> scan -> build -> "
                                 "value[] / depth[] of the set back in the caller's Modset arrays (modsetSyncToHost)"}
    # FASTA file, 80 columns, of the first ~1 Gbp
    m = max(1, min(int(np.searchsorted(offsets, want // 2, side="right")) - 1, n))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, "modgpu_e2e_%d.fa" % os.getpid())
    letters = np.frombuffer(b"ACGT", np.uint8)
    try:
        with open(path, "wb") as f:
            for r in range(m):
                s_ = letters[h[int(off[r]):int(off[r + 1])]]
                pad = (-len(s_)) % 80
                t_ = np.concatenate([s_, np.zeros(pad, np.uint8)]).reshape(-1, 80)
                t_ = np.concatenate([t_, np.full((len(t_), 1), 10, np.uint8)], axis=1).ravel()
                f.write(b">r%d\n" % r); f.write(t_[t_ != 0].tobytes())
        fb = int(off[m])

        def time_file(p_, host_parser):
            os.environ["MODGPU_TEXT_HOST"] = "1" if host_parser else "0"        # 1 = the host parser (mg_seqio.c), 0 = plain text parsed on the device (mg_textgpu.hip)
            L.mgReloadKnobs()
            best_ = None
            try:
                for it in range(3):
                    mg.check(L.mgModsetClear(ms, None))
                    t0 = time.perf_counter()
                    with mg.CFile(os.devnull, "w") as fo:
                        rc = L.mgAddSequenceFile(ms, p_.encode(), fo)
                    dt = time.perf_counter() - t0
                    if rc:
                        raise RuntimeError("mgAddSequenceFile failed")
                    if it:
                        best_ = dt if best_ is None else min(best_, dt)
            finally:
                del os.environ["MODGPU_TEXT_HOST"]
                L.mgReloadKnobs()
            return best_
        t_dev, t_host = time_file(path, False), time_file(path, True)
        res["fasta_file"] = {"entry": "mgAddSequenceFile", "Gbp_per_s": round(fb / t_dev / 1e9, 2), "Gbp_per_s_host_parser": round(fb / t_host / 1e9, 2),
                             "bases": fb, "file_bytes": os.path.getsize(path),
                             "what": "80-column FASTA in the page cache -> parallel pread into pinned memory -> the text across PCIe -> parsed on the device "
                                     "(record starts, headers, bases) -> 2-bit pack -> scan -> build; host_parser: parser threads -> pack -> H2D -> scan -> build"}
        os.remove(path)
        # 150-base reads as FASTQ (config 5's shape of input; k = 21 d = 64 like the other end-to-end legs)
        nq = min(4_000_000, nb // 150)
        seqs = letters[h[:nq * 150]].reshape(nq, 150)
        blk = np.concatenate([np.tile(np.frombuffer(b"@r\n", np.uint8), (nq, 1)), seqs, np.tile(np.frombuffer(b"\n+\n", np.uint8), (nq, 1)),
                              np.full((nq, 150), ord("I"), np.uint8), np.full((nq, 1), 10, np.uint8)], axis=1)
        path = os.path.join(shm, "modgpu_e2e_%d.fq" % os.getpid())
        blk.tofile(path); del blk, seqs
        t_dev, t_host = time_file(path, False), time_file(path, True)
        res["fastq_file"] = {"entry": "mgAddSequenceFile", "Gbp_per_s": round(nq * 150 / t_dev / 1e9, 2), "Gbp_per_s_host_parser": round(nq * 150 / t_host / 1e9, 2),
                             "bases": nq * 150, "reads": nq, "file_bytes": os.path.getsize(path),
                             "what": "150-base reads, four-line FASTQ in the page cache, parsed on the device (line = newlines before a byte, line mod 4 = what the byte is; "
                                     "'@', '+' and equal lengths checked, any breach goes back to the host parser)"}
        # modmap from files (modmap.c:93-134,188-281): a 20 Mbp reference FASTA, then the 150-base FASTQ file queried against it --
        # parse, scan, lookup, tallies and chaining on the device, one "Q" line per read (and "M" lines) formatted by the host's threads
        rpath = os.path.join(shm, "modgpu_e2e_%d_ref.fa" % os.getpid())
        try:
            rb = min(20_000_000, nb) // 80 * 80
            t_ = np.concatenate([letters[h[:rb]].reshape(-1, 80), np.full((rb // 80, 1), 10, np.uint8)], axis=1)
            with open(rpath, "wb") as f:
                f.write(b">ref1\n"); f.write(t_.tobytes())
            del t_

            def time_query(host_parser):
                os.environ["MODGPU_TEXT_HOST"] = "1" if host_parser else "0"
                L.mgReloadKnobs()
                try:
                    sh2 = mg.seqhashCreate(k, d, seed); ms2 = mg.modsetCreate(sh2, 24)
                    ref = L.mgReferenceCreate(ms2, 1 << 26)
                    best_, lines = None, 0
                    with mg.CFile(os.devnull, "w") as fo:
                        if L.mgReferenceFastaRead(ref, rpath.encode(), True, fo):
                            raise RuntimeError("mgReferenceFastaRead failed")
                    for it in range(3):
                        outp = os.path.join(shm, "modgpu_e2e_%d_q.txt" % os.getpid())
                        t0 = time.perf_counter()
                        with mg.CFile(outp, "w") as fo:
                            rc = L.mgQueryFile(ref, path.encode(), fo)
                        dt = time.perf_counter() - t0
                        if rc:
                            raise RuntimeError("mgQueryFile failed")
                        lines = os.path.getsize(outp)
                        with open(outp, "rb") as fo_:
                            sha = __import__("hashlib").sha1(fo_.read()).hexdigest()
                        os.remove(outp)
                        if it:
                            best_ = dt if best_ is None else min(best_, dt)
                    L.mgReferenceDestroy(ref); L.modsetDestroy(ms2)
                    return best_, lines, sha
                finally:
                    del os.environ["MODGPU_TEXT_HOST"]
                    L.mgReloadKnobs()
            (t_dev, out_bytes, sha_dev), (t_host, out_bytes_h, sha_host) = time_query(False), time_query(True)
            res["modmap_query_file"] = {"entry": "mgReferenceFastaRead + mgQueryFile", "Gbp_per_s": round(nq * 150 / t_dev / 1e9, 2),
                                        "Gbp_per_s_host_parser": round(nq * 150 / t_host / 1e9, 2), "reads": nq, "bases": nq * 150,
                                        "reference_bases": rb, "output_bytes": out_bytes, "same_output": sha_dev == sha_host and out_bytes == out_bytes_h, "output_sha1": sha_dev,
                                        "lines_per_s": round(nq / t_dev / 1e6, 1),
                                        "what": "150-base reads, four-line FASTQ in the page cache -> parsed on the device (record ids copied out of the pinned windows) -> scan + "
                                                "lookup + tallies + chaining on the device, a batch per 128 MiB window -> one Q line per read (M lines where blocks chain) formatted by a "
                                                "team of threads and written into a file in /dev/shm by two more threads while the next window is parsed and queried (mgQueryPipe*); "
                                                "host_parser: the same through mg_seqio.c and the 1-byte-per-base upload; unit of lines_per_s: million"}
        finally:
            if os.path.exists(rpath):
                os.remove(rpath)
    finally:
        if os.path.exists(path):
            os.remove(path)
    # SURVEY §8(f) N3, modasm's read ingest (modasm.c:151-191 + 258-287): the same reads against the modset built from them -- scan + lookups +
    # hit lists with distances on the device per batch, the hits per mod counted there, and at the end depth[], the inverse lists (a stable sort
    # of the hits' read numbers by mod) and the reads' copy-class tallies made on the device and mirrored into the caller's MgReadset
    try:
        mg.check(L.mgModsetClear(ms, None))
        if L.mgAddSequenceBatch(ms, h.ctypes.data, off.ctypes.data, n) < 0:
            raise RuntimeError(L.mgLastError().decode())
        best_rs, info_rs = None, None
        for it in range(3):                                        # best of three (best of a few after warm-up): the result arrays are fresh pages every time
            rs = L.mgReadsetCreate(ms)
            t0 = time.perf_counter()
            rc = L.mgReadsetRead(rs, h.ctypes.data, off.ctypes.data, n)
            dt = time.perf_counter() - t0
            if rc:
                raise RuntimeError("mgReadsetRead failed")
            R = C.cast(rs, C.POINTER(mg.MgReadset)).contents
            mmax = ms.contents.max
            inv_total = int(np.ctypeslib.as_array(R.invStart, (mmax + 2,))[mmax + 1])
            dep = np.ctypeslib.as_array(ms.contents.depth, (mmax + 1,))
            nhit = np.ctypeslib.as_array(R.nHit, (R.nReads + 1,))
            ok = (int(nhit[1:].sum()) == R.totHit and inv_total == int(dep[(dep > 0) & (dep < 65535)].astype(np.int64).sum()) and R.nReads == n)
            info_rs = {"reads": int(R.nReads), "hits": int(R.totHit), "inverse_list_entries": inv_total, "checks_ok": bool(ok)}
            L.mgReadsetDestroy(rs)
            best_rs = dt if best_rs is None else min(best_rs, dt)
        res["readset_ingest"] = dict(info_rs, entry="mgReadsetRead", Gbp_per_s=round(nb / best_rs / 1e9, 1), seconds=round(best_rs, 3), bases=nb,
                                     what="modasm's readsetFileRead + invBuild from host bytes: pack + H2D, scan + lookups + hit lists (index | strand, 16-bit distances) on the device, "
                                          "hits per mod counted on the device across batches, depth[] / invStart[] / invSpace[] / nCopy[] made there and mirrored")
    except Exception as e:
        res["readset_ingest"] = {"error": str(e)[:300]}
    L.modsetDestroy(ms)
    try:
        res["dropin_unmodified"] = dropin_unmodified(h, shm)
    except Exception as e:
        res["dropin_unmodified"] = {"error": str(e)[:300]}
    return res


def sync_to_host(cx, ms, step, S, entries):
    """the reference's Modset is transparent -- callers read ms->value / depth / index themselves
    (modset.h:17-28, modutils.c:26,69,186-198, modset.c:79-88) -- so the path ends when the host arrays hold what the device built.
    modsetSyncToHost at the headline's size (config 2's set, table bits 30): value[] + depth[] (11 bytes an entry), then index[]
    (4 * 2^bits bytes: the reference's slot layout replayed on the device).  `first`: straight after the timed steps, the host arrays
    never written before (page faults included); the steady figures: the set rebuilt (clear + scan + build, untimed) and synced again."""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    def one():
        torch.cuda.synchronize()
        t0 = time.perf_counter(); mg.check(L.modsetSyncToHost(ms, 0)); t1 = time.perf_counter()
        mg.check(L.modsetSyncToHost(ms, 1)); t2 = time.perf_counter()
        return (t1 - t0) * 1e3, (t2 - t1) * 1e3
    first = one()
    best = None
    for _ in range(2):
        step(); torch.cuda.synchronize()
        t = one()
        best = t if best is None or t[0] + t[1] < best[0] + best[1] else best
    m = ms.contents
    n = m.max
    dep = np.ctypeslib.as_array(m.depth, (n + 1,))
    val = np.ctypeslib.as_array(m.value, (n + 1,))
    idx = np.ctypeslib.as_array(m.index, (1 << m.tableBits,))
    vd_bytes = 10 * n
    ix_bytes = 4 << m.tableBits
    nz = int(np.count_nonzero(idx))
    ok = (n == entries and int(dep[1:].astype(np.int64).sum()) == S and nz == n and int(idx.max()) == n
          and len(np.unique(val[1:1 + min(n, 1 << 20)])) == min(n, 1 << 20))
    return {"entry": "modsetSyncToHost", "entries": n, "value_depth_ms": round(best[0], 2), "index_ms": round(best[1], 2),
            "with_index_ms": round(best[0] + best[1], 2),
            "GBps": round(vd_bytes / (best[0] * 1e-3) / 1e9, 2), "GBps_with_index": round((vd_bytes + ix_bytes) / ((best[0] + best[1]) * 1e-3) / 1e9, 2),
            "bytes_value_depth": vd_bytes, "bytes_index": ix_bytes,
            "first_call_ms": {"value_depth": round(first[0], 2), "index": round(first[1], 2)},
            "host_threads": int(L.mgXferThreadCount()),
            "checks_ok": bool(ok),
            "what": "device -> the Modset's own malloc()ed arrays: pending 32-bit counts exported as 16-bit, value[] / counts / replayed index[] in 4 MiB "
                    "pieces through page-locked blocks on one copy stream per host thread, each thread emptying its pieces into the destination "
                    "(memcpy; depth: saturating add, modutils.c:26); checks: depth sum == modimizers, index[] holds every entry once"}


def write_mod(cx, ms):
    """`modutils -a ... -w`: the config-2 set (already mirrored in the host arrays, index[] included: sync_to_host ran) written as a .mod
    (modset.c:79-88) through the library's gzip writer -- independent members deflated by a team of threads (mg_pgzip.c), which gzread,
    i.e. the reference, reads as one stream -- beside the rate of ONE zlib stream at the same level (what the reference's fzopen +
    gzwrite is, utils.c:107-127) on a sample of the same bytes."""
    import zlib
    import numpy as np
    L, mg = cx.L, cx.mg
    m = ms.contents
    n = m.max + 1
    raw_bytes = 104 + (4 << m.tableBits) + 11 * n
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, "modgpu_write_%d.mod" % os.getpid())
    libc = C.CDLL(None); libc.fclose.argtypes = [C.c_void_p]
    try:
        t0 = time.perf_counter()
        f = L.mgGzipOpenWrite(path.encode())
        if not f:
            raise RuntimeError("cannot create " + path)
        L.modsetWrite(ms, C.c_void_p(f))
        if libc.fclose(C.c_void_p(f)):
            raise RuntimeError("write failed")
        dt = time.perf_counter() - t0
        zsize = os.path.getsize(path)
        # the head of the file decompresses to the header + the head of index[]
        idx = np.ctypeslib.as_array(m.index, (1 << m.tableBits,))
        with open(path, "rb") as fh:
            head = zlib.decompressobj(31).decompress(fh.read(64 << 20), 8 << 20)
        ok = head[:8] == b"MSHSTv2\0" and head[104:] == idx[:(len(head) - 104) // 4 + 1].tobytes()[:len(head) - 104]
        # one stream, one thread, level 6: 192 MiB of index[] from the middle + 64 MiB of value[]
        val = np.ctypeslib.as_array(m.value, (n,))
        sample = [idx[len(idx) // 2: len(idx) // 2 + (48 << 20)].tobytes(), val[1:1 + (8 << 20)].tobytes()]
        z = zlib.compressobj(6, zlib.DEFLATED, 31)
        t0 = time.perf_counter()
        zs = sum(len(z.compress(b)) for b in sample) + len(z.flush())
        dt1 = time.perf_counter() - t0
        sb = sum(len(b) for b in sample)
        # and back: modsetRead (modset.c:90-104) through the library's reader -- the members found by the sizes in their extra fields and
        # inflated by the team, whole members straight into index[] / value[] -- beside ONE inflate stream (what gzread is) on the sample
        t0 = time.perf_counter()
        fr = L.mgGzipOpenRead(path.encode())
        if not fr:
            raise RuntimeError("mgGzipOpenRead refused the file")
        L.modsetRead.restype = C.POINTER(mg.Modset)
        ms2 = L.modsetRead(C.c_void_p(fr))
        libc.fclose(C.c_void_p(fr))
        dt_r = time.perf_counter() - t0
        m2 = ms2.contents
        idx2 = np.ctypeslib.as_array(m2.index, (1 << m2.tableBits,)); val2 = np.ctypeslib.as_array(m2.value, (n,))
        dep2 = np.ctypeslib.as_array(m2.depth, (n,)); dep1 = np.ctypeslib.as_array(m.depth, (n,))
        same = bool(m2.max == m.max and np.array_equal(idx2, idx) and np.array_equal(val2[1:], val[1:]) and np.array_equal(dep2, dep1))
        L.modsetDestroy(ms2)
        zc = zlib.compressobj(6, zlib.DEFLATED, 31); comp = b"".join(zc.compress(b) for b in sample) + zc.flush()
        t0 = time.perf_counter(); zlib.decompress(comp, 31); dt_r1 = time.perf_counter() - t0
    finally:
        if os.path.exists(path):
            os.remove(path)
    par, one = raw_bytes / dt / 1e6, sb / dt1 / 1e6
    return {"entry": "modsetWrite through mgGzipOpenWrite", "raw_bytes": raw_bytes, "file_bytes": zsize, "seconds": round(dt, 2),
            "MBps": round(par, 1), "single_stream_MBps": round(one, 1), "speedup_vs_single_stream": round(par / one, 1),
            "single_stream_seconds_estimate": round(raw_bytes / (one * 1e6), 1), "single_stream_sample_bytes": sb,
            "single_stream_sample_ratio": round(zs / sb, 3), "file_ratio": round(zsize / raw_bytes, 3),
            "threads": min(int(os.environ.get("MODGPU_GZIP_THREADS", "0")) or int(L.mgCpuBudget()), 32), "head_decompresses_ok": bool(ok),
            "read_back": {"entry": "modsetRead through mgGzipOpenRead", "seconds": round(dt_r, 2), "MBps": round(raw_bytes / dt_r / 1e6, 1),
                          "single_stream_MBps": round(sb / dt_r1 / 1e6, 1), "speedup_vs_single_stream": round(raw_bytes / dt_r / (sb / dt_r1), 1),
                          "same_arrays": same},
            "what": "config 2's set, table bits 30: 104 + 4 * 2^30 + 11 * (max + 1) bytes -> gzip members of 16 MiB deflated in parallel (level 6; per member Z_RLE where its first 128 KiB "
                    "say that costs no size: the zero runs of index[] and of the k-mers' high bytes), written in order into /dev/shm; single_stream: zlib level 6, default strategy, "
                    "on one thread over a 256 MiB sample of index[] and value[] -- what the reference's fzopen + gzwrite does"}


def modmap_query_file_long(cx):
    """BASELINE config 3 in its own shape, FROM FILES (modmap.c:93-134,188-281): a 24 x 125 Mbp FASTA reference (80 columns) through
    mgReferenceFastaRead, then >= 5 Gbp of ONT-like reads drawn from it (FASTA, one line a read) through mgQueryFile -- text parsed on the
    device, scan, lookups, tallies and chaining there, Q / M lines formatted and written by the host's threads.  The chaining kernels'
    share (mg_chain.hip) is split out per 10 Gbp from the library's event timers."""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    scale = float(os.environ.get("MODGPU_BENCH_C3_SCALE", "1"))
    n_seq, seq_len = 24, int(125_000_000 * scale) // 80 * 80
    genome_bases = n_seq * seq_len
    q_bases = int(float(os.environ.get("MODGPU_BENCH_LONG_QUERY_GBP", "5")) * 1e9 * min(1.0, scale * 4))
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    need = genome_bases * 81 // 80 + q_bases + (1 << 28)
    free = __import__("shutil").disk_usage(shm).free
    if free < need * 1.2:
        return {"skipped": "%s has %.1f GB free, the two files need %.1f GB" % (shm, free / 1e9, need / 1e9)}
    k, d, bits = 21, 64, 28
    rpath = os.path.join(shm, "modgpu_long_%d_ref.fa" % os.getpid())
    qpath = os.path.join(shm, "modgpu_long_%d_reads.fa" % os.getpid())
    opath = os.path.join(shm, "modgpu_long_%d_out.txt" % os.getpid())

    def letters_of(packed, n):                                     # bases 0..3 of a packed stream as ASCII, on the device
        b = torch.empty(n, dtype=torch.uint8, device=cx.dev)
        mg.check(L.mgUnpackDevice(packed.data_ptr(), n, b.data_ptr(), cx.stream))
        torch.cuda.synchronize()
        b += 65; b += (b > 65).to(torch.uint8); b += (b > 67).to(torch.uint8) * 3; b += (b > 71).to(torch.uint8) * 12      # 0 1 2 3 -> 65 67 71 84 = A C G T
        return b
    t_files = time.perf_counter()
    try:
        genome = make_genome(cx, genome_bases, 333)
        nl = torch.full((seq_len // 80, 1), 10, dtype=torch.uint8, device=cx.dev)
        with open(rpath, "wb") as f:
            for i in range(n_seq):
                # (sequence i starts on a word boundary: seq_len is a multiple of 16)
                view = genome[i * seq_len // 16:]
                t_ = torch.cat([letters_of(view, seq_len).view(-1, 80), nl], dim=1).cpu().numpy()
                f.write(b">chr%d\n" % (i + 1)); f.write(t_.tobytes())
        del nl
        reads, d_offsets, offsets, n_reads = make_reads(cx, q_bases, genome, genome_bases, 4242, 0.05, 5252)
        del genome
        q_bases = int(offsets[n_reads])
        h = letters_of(reads, q_bases).cpu().numpy()
        del reads, d_offsets
        torch.cuda.empty_cache()
        mv = memoryview(h)
        with open(qpath, "wb", buffering=1 << 24) as f:
            for r in range(n_reads):
                f.write(b">r%d\n" % r); f.write(mv[int(offsets[r]):int(offsets[r + 1])]); f.write(b"\n")
        del mv, h
        t_files = time.perf_counter() - t_files
        sh = mg.seqhashCreate(k, d, 17); ms = mg.modsetCreate(sh, bits)
        ref = L.mgReferenceCreate(ms, 1 << 26)
        t_refs = []
        for it in range(2):                                        # (the first call makes the parser's page-locked windows and device buffers)
            if it:
                L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
                sh = mg.seqhashCreate(k, d, 17); ms = mg.modsetCreate(sh, bits)
                ref = L.mgReferenceCreate(ms, 1 << 26)
            with mg.CFile(os.devnull, "w") as fo:
                t0 = time.perf_counter()
                if L.mgReferenceFastaRead(ref, rpath.encode(), True, fo):
                    raise RuntimeError("mgReferenceFastaRead failed")
                t_refs.append(time.perf_counter() - t0)
        t_ref = min(t_refs)
        r_ = C.cast(ref, C.POINTER(mg.MgReference)).contents
        best, lines, chain_ms = None, 0, None
        for it in range(3):
            if it == 2:
                L.mgProfileOnly(-1); L.mgProfileEnable(1); L.mgProfileReset()
            t0 = time.perf_counter()
            with mg.CFile(opath, "w") as fo:
                rc = L.mgQueryFile(ref, qpath.encode(), fo)
            dt = time.perf_counter() - t0
            if rc:
                raise RuntimeError("mgQueryFile failed")
            if it == 2:
                table = read_profile(L, mg); L.mgProfileEnable(0)
                chain_ms = table.get("mgChainKernel", (0.0, 0))[0] + table.get("mgChainResolveKernel", (0.0, 0))[0]
                kern_ms = {kn: round(v[0], 2) for kn, v in table.items() if v[0] >= 0.05}
            elif it:
                best = dt
        with open(opath, "rb") as fo:
            txt = fo.read()
        n_q, n_m = txt.count(b"\nQ\t") + txt.startswith(b"Q\t"), txt.count(b"\nM\t")
        res = {"entry": "mgReferenceFastaRead + mgQueryFile",
               "reference": {"sequences": n_seq, "bases": genome_bases, "file_bytes": os.path.getsize(rpath), "read_s": round(t_ref, 3), "first_call_s": round(t_refs[0], 3),
                             "Gbp_per_s": round(genome_bases / t_ref / 1e9, 2), "occurrences": int(r_.max), "modset_entries": int(ms.contents.max)},
               "query": {"reads": n_reads, "bases": q_bases, "file_bytes": os.path.getsize(qpath), "seconds": round(best, 3),
                         "Gbp_per_s": round(q_bases / best / 1e9, 2), "Q_lines": int(n_q), "M_lines": int(n_m), "all_reads_reported": int(n_q) == n_reads,
                         "chain_ms_per_10Gbp": round(chain_ms / q_bases * 1e10, 3) if chain_ms is not None else None,
                         "chain_ms_total": round(chain_ms, 3) if chain_ms is not None else None, "kernel_ms_profiled_run": kern_ms},
               "files_written_in_s": round(t_files, 1),
               "what": "80-column FASTA reference and one-line-per-read FASTA reads in the page cache (/dev/shm); parse on the device, scan + insert + "
                       "reference arrays on the device (mg_refpack.hip); queries: scan + lookups + tallies + chaining on the device a batch at a time"}
        L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
        return res
    finally:
        for p_ in (rpath, qpath, opath):
            if os.path.exists(p_):
                os.remove(p_)


def dropin_unmodified(h, shm):
    """The reference's UNMODIFIED modutils.c (its own main(), seqio and per-read loop modutils.c:19-51: modRCiterator /
    modRCnext / modsetIndexFind per read) linked on libmodgpu.so (oracle/_ref/modutils_dropin) beside the reference program
    itself (oracle/_ref/modutils_ref) and the batch-patched one (oracle/_ref/modutils_batch, examples/modutils_batch.patch)
    on the same FASTA files: 10 kb, 150 b and 24 kb reads cut from the bench's reads (the first two are scanned by modRCiterator's host leg, the
    third by its kernel leg: every row says which).  Wall clock of the whole program, and the
    marginal rate (big file minus a 1/20 file: start-up, HIP initialisation and table allocation cancel)."""
    import numpy as np
    refdir = os.path.join(HERE, "oracle", "_ref")
    progs = {n: os.path.join(refdir, n) for n in ("modutils_ref", "modutils_dropin", "modutils_batch")}
    if not all(os.path.exists(p) for p in progs.values()):
        return {"skipped": "oracle/_ref programs not present (built where the reference tree is)"}
    letters = np.frombuffer(b"ACGT", np.uint8)
    out = {"what": "whole-program wall clock, `modutils -c 26 21 64 17 -a <file>`; Mbp/s = marginal (big file minus small file)",
           "iterator_crossover_bases": int(__import__("modimizer_amd").lib().mgIterHostBelow(-1)),
           "crossover_note": "modRCiterator scans reads shorter than this with the library's own scalar loop (a synchronous call cannot hide "
                             "the 13-15 us of a kernel launch + poll), longer ones with one kernel launch (mg_host.c)"}
    crossover = out["iterator_crossover_bases"]
    for tag, rl, mbp in (("reads_10kb", 10000, 400), ("reads_150b", 150, 400), ("reads_24kb", 24000, 400)):
        paths = []
        for frac in (20, 1):
            nb = min(len(h), int(mbp * 1e6) // frac) // rl * rl
            seq = letters[h[:nb]].reshape(-1, rl)
            path = os.path.join(shm, "modgpu_dropin_%d_%s_%d.fa" % (os.getpid(), tag, frac))
            with open(path, "wb") as f:
                hdr = np.frombuffer(b">r\n", np.uint8)
                rec = np.concatenate([np.tile(hdr, (len(seq), 1)), seq, np.full((len(seq), 1), 10, np.uint8)], axis=1)
                f.write(rec.tobytes())
            paths.append((path, nb))
        # say which leg of modRCiterator a row measures -- below the crossover the drop-in's scan is the library's scalar HOST loop
        # (the GPU is required but idle); the 24 kb row is the one that runs the one-launch-per-read kernel
        row = {"read_length": rl, "bases": paths[1][1], "reads": paths[1][1] // rl,
               "iterator_leg_of_modutils_dropin": "host scalar loop (read shorter than the crossover: no kernel runs)" if rl < crossover else "GPU kernel, one launch per read"}
        try:
            for name, prog in progs.items():
                t = []
                for path, nb in paths:
                    t0 = time.perf_counter()
                    r = subprocess.run([prog, "-c", "26", "21", "64", "17", "-a", path], capture_output=True, text=True, timeout=1200)
                    t.append(time.perf_counter() - t0)
                    if r.returncode != 0:
                        raise RuntimeError("%s failed: %s" % (name, r.stderr[-200:]))
                    line = [l for l in r.stdout.splitlines() if l.startswith("added ")]
                    row.setdefault("stdout_added_line", {})[name] = line[-1] if line else None
                d_b, d_t = paths[1][1] - paths[0][1], t[1] - t[0]
                row[name] = {"wall_s_small": round(t[0], 3), "wall_s": round(t[1], 3),
                             "Mbp_per_s": round(d_b / d_t / 1e6, 1) if d_t > 0 else None,
                             "us_per_read": round(d_t / (d_b / rl) * 1e6, 2) if d_t > 0 else None}
            row["same_result"] = len(set(row["stdout_added_line"].values())) == 1
        finally:
            for path, _ in paths:
                if os.path.exists(path):
                    os.remove(path)
        out[tag] = row
    return out


def cpu_baseline(cx, reads, offsets, k, d, seed):
    """The reference's own C path (oracle/_ref/ref_bench, built from the unmodified sources) timed
    single-threaded — its real execution model — on the first reads of the same workload; and, so that the GPU is not
    flattered, the same sample sharded over every host core with private modsets merged in block order
    (modsetMerge semantics, modset.c:106-128) by oracle/cpu_bench (this repo's C restatement, pthreads)."""
    import numpy as np
    torch, L, mg = cx.torch, cx.L, cx.mg
    sample_mbp = float(os.environ.get("MODGPU_CPU_SAMPLE_MBP", "1200"))
    want = int(sample_mbp * 1e6)
    n = int(np.searchsorted(offsets, want, side="right")) - 1
    n = max(1, min(n, len(offsets) - 1))
    nb = int(offsets[n])
    d_bytes = torch.empty(nb, dtype=torch.uint8, device=cx.dev)
    mg.check(L.mgUnpackDevice(reads.data_ptr(), nb, d_bytes.data_ptr(), cx.stream))
    torch.cuda.synchronize()
    h_bytes = d_bytes.cpu().numpy()
    del d_bytes
    off = offsets[:n + 1].astype(np.int64)
    sample_desc = "first %d reads (%.0f Mbp) of the same workload, single thread, table bits 28" % (n, nb / 1e6)
    ref_bench = os.path.join(HERE, "oracle", "_ref", "ref_bench")
    cpu_bench = os.path.join(HERE, "oracle", "cpu_bench")
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    path = os.path.join(shm, "modgpu_cpu_sample_%d.bin" % os.getpid())
    res = None
    try:
        with open(path, "wb") as f:
            np.array([n, nb], np.uint64).tofile(f); off.tofile(f); h_bytes.tofile(f)
        if os.path.exists(ref_bench):
            try:
                r = subprocess.run([ref_bench, path, str(k), str(d), str(seed), "28"],
                                   capture_output=True, text=True, timeout=900)
                if r.returncode == 0:
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    res = {"value": round(j["sketch_mbps"] / 1e3, 5), "unit": "Gbp/s", "cores": 1,
                           "kind": "reference", "sample": sample_desc,
                           "scan_only_gbps": round(j["scan_mbps"] / 1e3, 5),
                           "host_cores_online": os.cpu_count()}
            except Exception:
                res = None
        if res is None:
            # fall back to this repo's C restatement of the same path
            from oracle import pyoracle as po
            oh = po.Hasher(k, d, seed)
            oms = po.Modset(oh, 28)
            t0 = time.perf_counter()
            po.lib().orcScanMany(C.byref(oh.c), h_bytes.ctypes.data, off.ctypes.data, n, oms.p)
            dt = time.perf_counter() - t0
            res = {"value": round(nb / dt / 1e9, 5), "unit": "Gbp/s", "cores": 1, "kind": "port",
                   "sample": sample_desc, "host_cores_online": os.cpu_count()}
        if os.path.exists(cpu_bench):
            try:
                r = subprocess.run([cpu_bench, path, str(k), str(d), str(seed), "28"], capture_output=True, text=True, timeout=900)
                res["all_cores"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-300:]}
            except Exception as e:                  # informational only
                res["all_cores"] = {"error": str(e)[:200]}
    finally:
        if os.path.exists(path):
            os.remove(path)
    return res


if __name__ == "__main__":
    main()

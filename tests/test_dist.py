"""The N>1 path on CPU: world_size-2 gloo.  Reads are sharded in contiguous blocks, each rank
builds its own modset (here with the oracle, since there is no GPU in this container) and the
depth histograms are summed with one all-reduce — the collective bench.py runs over RCCL."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from modimizer_amd import dist as mdist   # noqa: E402
from modimizer_amd import synth           # noqa: E402
from tests.util import bench_lines        # noqa: E402


def test_shard_bounds_cover_and_are_contiguous():
    for n in (0, 1, 7, 8, 9, 1000, 598835):
        for world in (1, 2, 3, 8):
            blocks = [mdist.shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    offs = np.array([0, 5, 5, 12, 40, 41], np.int64)
    sub, base, lo, hi = mdist.shard_offsets(offs, 2, 1)
    assert (lo, hi) == (3, 5) and base == 12 and list(sub) == [0, 28, 29]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    k, d = 21, 64
    genome = synth.iid_bases(60000, 1)
    starts, offsets, strands = synth.ont_read_plan(1_500_000, len(genome), 2, n50=4000, lo=100, hi=20000)
    bases = synth.reads_from_genome(genome, starts, offsets, strands, 0.03, 3)
    sub, base, lo, hi = mdist.shard_offsets(offsets.astype(np.int64), world, rank)
    h = po.Hasher(k, d, 17)
    ms = po.Modset(h, 22)
    for r in range(len(sub) - 1):
        ms.add_sequence(bases[base + sub[r]:base + sub[r + 1]])
    hist = torch.from_numpy(ms.histogram().astype(np.int64))
    mine = hist.clone()
    mdist.allreduce_histogram(hist)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, mine.numpy(), hist.numpy(), float(t.item()), ms.max))
    dist.barrier()
    dist.destroy_process_group()


def test_histogram_allreduce_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per_rank = [r[1] for r in res]
    total = per_rank[0] + per_rank[1]
    for r in res:
        assert np.array_equal(r[2], total)          # every rank holds the summed histogram
        assert r[3] == float(world)                 # MAX reduction
    assert total.sum() == res[0][4] + res[1][4]     # one bin entry per modset entry per rank
    # the shards are independent: rebuilding them serially gives the same per-rank histograms
    from oracle import pyoracle as po
    genome = synth.iid_bases(60000, 1)
    starts, offsets, strands = synth.ont_read_plan(1_500_000, len(genome), 2, n50=4000, lo=100, hi=20000)
    bases = synth.reads_from_genome(genome, starts, offsets, strands, 0.03, 3)
    for rank in range(world):
        sub, base, lo, hi = mdist.shard_offsets(offsets.astype(np.int64), world, rank)
        ms = po.Modset(po.Hasher(21, 64, 17), 22)
        for r in range(len(sub) - 1):
            ms.add_sequence(bases[base + sub[r]:base + sub[r + 1]])
        assert np.array_equal(ms.histogram().astype(np.int64), per_rank[rank])


def _merge_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ["MODGPU_NO_TORCH"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    import modimizer_amd as mg
    from oracle import pyoracle as po
    L = mg.lib()
    k, d, bits = 21, 16, 20
    genome = synth.iid_bases(50000, 11)
    starts, offsets, strands = synth.ont_read_plan(900_000, len(genome), 12, n50=3000, lo=100, hi=15000)
    bases = synth.reads_from_genome(genome, starts, offsets, strands, 0.02, 13)
    sub, base, lo, hi = mdist.shard_offsets(offsets.astype(np.int64), world, rank)
    sh = mg.seqhashCreate(k, d, 17); oh = po.Hasher(k, d, 17)
    ms = mg.modsetCreate(sh, bits)
    # this rank's shard, built with the reference-style scalar loop on the host arrays (no GPU here)
    for r in range(len(sub) - 1):
        for km in oh.scan(bases[base + sub[r]:base + sub[r + 1]])[0]:
            ix = L.modsetIndexFind(ms, int(km), 1)
            dd = (int(ms.contents.depth[ix]) + 1) & 0xffff
            ms.contents.depth[ix] = dd if dd else 0xffff
    merged = mdist.merge_modsets_in_rank_order(ms, L)
    if rank == 0:
        v, dep, _ = mg.modset_arrays(merged)
        q.put((v, dep))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_order_merge_equals_single_stream_gloo_world2():
    """per-rank modsets over contiguous read blocks, merged in rank order (modsetMerge semantics),
    reproduce the single-stream build bit for bit: same first-occurrence index order, same depths"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    v, dep = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import pyoracle as po
    genome = synth.iid_bases(50000, 11)
    starts, offsets, strands = synth.ont_read_plan(900_000, len(genome), 12, n50=3000, lo=100, hi=15000)
    bases = synth.reads_from_genome(genome, starts, offsets, strands, 0.02, 13)
    oms = po.Modset(po.Hasher(21, 16, 17), 20)
    for r in range(len(starts)):
        oms.add_sequence(bases[int(offsets[r]):int(offsets[r + 1])])
    assert len(v) - 1 == oms.max
    assert np.array_equal(v[1:], oms.values()[1:]) and np.array_equal(dep[1:], oms.depths()[1:])


def _count_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    ms, reads = _fixed_set_and_reads(po)
    lo, hi = mdist.shard_bounds(len(reads), world, rank)
    q.put((rank, mdist.allreduce_depth(_count_hits(po, ms, reads[lo:hi]))))
    dist.barrier()
    dist.destroy_process_group()


def _fixed_set_and_reads(po):
    """a modset built from a genome, and reads to count against it: among them one short repeat unit a few hundred thousand times over, so
    that some entries pass 65535 on the single stream and on neither rank alone"""
    k, d = 15, 4
    genome = synth.iid_bases(30000, 7)
    ms = po.Modset(po.Hasher(k, d, 17), 20)
    ms.add_sequence(genome)
    unit = genome[1000:1040]
    tiled = [np.tile(unit, 6000) for _ in range(14)]                    # 84 000 copies of the unit in all: 42 000 in either half of the reads
    reads = tiled[:7] + [genome[i * 900:i * 900 + 1500] for i in range(30)] + [synth.iid_bases(5000, 99)] + tiled[7:]
    return ms, reads


def _count_hits(po, ms, reads):
    """modasm.c:161-176 on the oracle: depth[] of a fixed set from these reads alone, saturating"""
    dep = np.zeros(ms.max + 1, np.uint32)
    h = po.Hasher(15, 4, 17)
    for r in reads:
        for km in h.scan(r)[0]:
            ix = ms.find(int(km))
            if ix:
                dep[ix] += 1
    return np.minimum(dep, 65535).astype(np.uint16)


def test_depth_allreduce_fixed_modset_gloo_world2():
    """SURVEY 8(e) 'count against a fixed modset': every rank counts its block of reads against the same set; all-reduce (sum, 32-bit) and
    clamp = the saturated counts of one stream over all the reads, also where the sum passes 65535 and no rank's count does"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_count_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import pyoracle as po
    ms, reads = _fixed_set_and_reads(po)
    whole = _count_hits(po, ms, reads)
    halves = [_count_hits(po, ms, reads[slice(*mdist.shard_bounds(len(reads), world, r))]) for r in range(world)]
    assert (whole == 65535).any() and not any((h == 65535).any() for h in halves)
    for _, got in res:
        assert got.dtype == np.uint16 and np.array_equal(got, whole)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_starts_n_ranks(n):
    """`python bench.py --gpus N` (no WORLD_SIZE in the environment) starts N ranks itself, before anything touches
    a GPU; --dry-launch runs what a rank does around the GPU work on CPU: process group (gloo), a small modset per
    rank through the scalar host API, the histogram all-reduce, MAX-over-ranks timing, one JSON line from rank 0.
    N = 8 is the shape of the driver's scaling run (it must not be the first time that launch executes)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1 and r.stdout.rstrip().splitlines()[-1] == line[0]      # one line, from rank 0 only, and it is the last
    assert len(line[0]) < 8192
    j = json.loads(line[0])
    assert j["n_gpus"] == n and j["dry_launch"] is True and j["steps"] == 3
    assert j["kmers_all_ranks"] == sum(3000 + 100 * r_ for r_ in range(n))       # every rank's block took part
    assert j["histogram_entries"] == j["entries_all_ranks"] == sum(1000 + 10 * r_ for r_ in range(n))   # the all-reduced histogram holds every entry of every rank


def test_bench_under_the_drivers_launcher_world8():
    """the driver's own command for N = 8 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 8 ...` -- where bench.py is a RANK (WORLD_SIZE set), not the launcher; dry: gloo, no GPU"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1
    j = json.loads(line[0])
    assert j["n_gpus"] == 8 and j["histogram_entries"] == j["entries_all_ranks"] == sum(1000 + 10 * r_ for r_ in range(8))


@pytest.mark.gpu
def test_bench_multi_rank_path_on_one_gpu():
    """the N > 1 code path of bench.py (RCCL process group, per-step histogram + all-reduce, barriers, collective
    timing) at world size 1: the all-reduced histogram is the rank's own modsetDepthHistogramDevice result"""
    env = dict(os.environ, MODGPU_BENCH_FORCE_DIST="1", MODGPU_BENCH_GBP="0.3", MASTER_PORT=str(_free_port()),
               MODGPU_BENCH_C3_SCALE="0.02", MODGPU_BENCH_C3_BATCHES="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j, full, compact = bench_lines(r.stdout)
    assert len(compact) == 1 and len(compact[0]) < 8192 and r.stdout.rstrip().splitlines()[-1] == compact[0]
    assert j["n_gpus"] == 1 and "config 4" in j["config"]["workload"]
    c = j["collective"]
    assert c["matches_local_sums"] is True and c["histogram_entries"] == j["config"]["modset_entries"] > 0
    assert j["roofline"]["kernel"] and j["roofline"]["frac"] > 0 and full["roofline"]["alu"]["floor_ms"] > 0
    # every rank's block checked on its own GPU (prefix property), the single-GPU figure of the same workload, and the
    # query sharding (north_star: reads shard, modset replicated) at this world size
    assert j["per_rank_parity"] is True and full["per_rank_parity_rank0"]["prefix_entries"] > 0
    assert j["single_gpu_block_gbps"] > 0
    c3 = full["other_configs"]["c3_sharded"]
    assert "error" not in c3 and c3["n_gpus"] == 1 and c3["value"] > 0 and 0.25 < c3["seed_hit_fraction"] < 0.45
    assert j["other_configs"]["c3_sharded"]["value"] == c3["value"]


@pytest.mark.gpu
def test_c_comm_entry_points_world1():
    """the one-process-per-device way into RCCL from the C ABI (mgCommGetUniqueId + mgCommInitRank) at world size 1: the all-reduced
    histogram is modsetDepthHistogramDevice's, mgModsetMergeRankOrder with nobody to merge leaves the set as it is.  Run in a process of
    its own (librccl is dlopen()ed into it)."""
    code = r"""
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib(); mg.check(L.mgSetDevice(0))
ident = (C.c_ubyte * 128)(); mg.check(L.mgCommGetUniqueId(ident))
comm = C.c_void_p(); mg.check(L.mgCommInitRank(C.byref(comm), 1, 0, ident, 0))
assert L.mgCommRank(comm) == 0 and L.mgCommSize(comm) == 1
sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 24)
bases = synth.iid_bases(3_000_000, 5); bases = np.concatenate([bases, bases[:1_000_000]])      # depths 1 and 2
off = np.array([0, len(bases)], np.int64)
n = mg.add_sequence_batch(ms, bases, off)
hist = np.zeros(65536, np.uint64); mg.check(L.mgHistogramAllReduce(ms, hist.ctypes.data, comm))
d = mg.DeviceBuffer(65536 * 8); mg.check(L.mgMemsetD(d.ptr, 0, 65536 * 8, None)); mg.check(L.modsetDepthHistogramDevice(ms, d.ptr, None))
own = d.to_numpy(np.uint64, 65536)
assert np.array_equal(hist, own) and int(hist.sum()) == ms.contents.max and hist[2] > 0 and int((hist * np.arange(65536, dtype=np.uint64)).sum()) == n
before = ms.contents.max
mg.check(L.mgModsetMergeRankOrder(ms, comm, 0)); assert ms.contents.max == before
mg.check(L.modsetSyncToHost(ms, 0)); dep = mg.modset_arrays(ms)[1].copy()
mg.check(L.mgDepthAllReduce(ms, comm))                                        # one rank: the sum is its own counts, on the host and in the device table
assert np.array_equal(mg.modset_arrays(ms)[1], dep)
d2 = mg.DeviceBuffer(65536 * 8); mg.check(L.mgMemsetD(d2.ptr, 0, 65536 * 8, None)); mg.check(L.modsetDepthHistogramDevice(ms, d2.ptr, None))
assert np.array_equal(d2.to_numpy(np.uint64, 65536), own)
n2 = mg.add_sequence_batch(ms, bases[:500_000], np.array([0, 500_000], np.int64))   # and the table still counts on top of it
mg.check(L.modsetSyncToHost(ms, 0)); assert int(mg.modset_arrays(ms)[1][1:].astype(np.int64).sum()) == n + n2
L.mgCommDestroy(comm); L.modsetDestroy(ms)
print("COMM_OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "COMM_OK" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_query_scratch_follows_the_thread_to_another_gpu(tmp_path):
    """mgQueryFile on device 0 leaves the calling thread's query scratch allocated there (it is kept between the batches of a
    file); the same thread then moves to device 1 (mgSetDevice) and queries there -- the scratch is the thread's and its device's, so it
    is dropped and made again on device 1 instead of device-0 blocks being handed to kernels on device 1.  Needs two GPUs."""
    same = os.environ.get("MODGPU_TEST_TWO_GPU_ON_ONE") == "1"        # (dev: both halves on device 0, to check the script itself on a one-GPU box)
    if _gpu_count() < 2 and not same:
        pytest.skip("needs 2 GPUs, this box has %d" % _gpu_count())
    code = r"""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, %r)
import modimizer_amd as mg
from modimizer_amd import synth
L = mg.lib()
genome = synth.iid_bases(2_000_000, 5)
reads = [genome[i * 7000:i * 7000 + 6000] for i in range(200)]
fa = os.path.join(%r, "q.fa")
with open(fa, "w") as f:
    for i, r in enumerate(reads):
        f.write(">r%%d\n%%s\n" %% (i, "".join("ACGT"[b] for b in r)))
outs = []
for half, dev in enumerate(%r):
    mg.check(L.mgSetDevice(dev))
    sh = mg.seqhashCreate(21, 64, 17); ms = mg.modsetCreate(sh, 22)
    ref = L.mgReferenceCreate(ms, 1 << 22)
    off = np.array([0, len(genome)], np.int64); names = (C.c_char_p * 1)(b"g")
    with mg.CFile(os.devnull, "w") as fo:
        assert L.mgReferenceRead(ref, genome.ctypes.data, off.ctypes.data, 1, names, True, fo) == 0
    out = os.path.join(%r, "o%%d.txt" %% half)
    with mg.CFile(out, "w") as fo:
        if half == 0:
            assert L.mgQueryFile(ref, fa.encode(), fo) == 0                  # leaves the scratch of this thread on device 0
        else:
            b = np.concatenate(reads); o = np.arange(len(reads) + 1, dtype=np.int64) * 6000
            nm = (C.c_char_p * len(reads))(*[b"r%%d" %% i for i in range(len(reads))])
            assert L.mgQueryProcess(ref, b.ctypes.data, o.ctypes.data, len(reads), nm, fo) == 0
    outs.append(open(out).read())
    L.mgReferenceDestroy(ref); L.modsetDestroy(ms)
assert outs[0] == outs[1] and outs[0].count("Q\t") == len(reads)
print("TWO_GPU_OK")
""" % (ROOT, str(tmp_path), (0, 0) if same else (0, 1), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "TWO_GPU_OK" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


@pytest.mark.gpu
def test_bench_world2_code_path_on_one_gpu():
    """`python bench.py --gpus 2` with both ranks on cuda:0 (MODGPU_BENCH_ONE_GPU=1: gloo instead of RCCL, which refuses two
    ranks on one device): the world > 1 branches a one-GPU box cannot otherwise reach -- block r on rank r, the summed histogram
    holding both ranks' entries, the parity flag all-reduced over both, rank 1 waiting out rank 0's single-GPU leg, the query
    reads split into two shares over replicated modsets.  The rates of such a run mean nothing and are not looked at."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MODGPU_BENCH_ONE_GPU="1", MODGPU_BENCH_GBP="0.3", MODGPU_BENCH_C3_SCALE="0.02", MODGPU_BENCH_C3_BATCHES="3")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=1200, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j, full, lines = bench_lines(r.stdout)
    assert len(lines) == 1 and len(lines[0]) < 8192        # rank 0 only, and under the size the driver can read
    assert r.stdout.rstrip().splitlines()[-1] == lines[0]  # ... as the LAST line of stdout
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and "block r on rank r" in j["config"]["workload"]
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert key in j, key
    r_ = j["roofline"]                                     # the N > 1 line carries the roofline object too
    assert r_["bound"] == "hbm" and r_["kernel"] and r_["frac"] > 0 and r_["peak"] == 8000.0 and r_["avg_launch_ms"] > 0
    c = j["collective"]
    assert c["matches_local_sums"] is True and c["histogram_entries"] == c["entries_all_ranks"] > j["config"]["modset_entries"]
    assert j["per_rank_parity"] is True and j["single_gpu_block_gbps"] > 0
    c3 = full["other_configs"]["c3_sharded"]
    assert "error" not in c3 and c3["n_gpus"] == 2 and c3["value"] > 0 and 0.25 < c3["seed_hit_fraction"] < 0.45
    assert j["other_configs"]["c3_sharded"]["n_gpus"] == 2


@pytest.mark.gpu
def test_bench_forced_dist_agrees_with_plain_single_gpu():
    """N = 1 through the N > 1 code path (MODGPU_BENCH_FORCE_DIST=1: process group on RCCL, histogram + all-reduce in the step) against the
    same block through the plain N = 1 code (`--only c4_block`: scan + build + histogram, no collective): the all-reduce of 512 KiB and
    the barriers must not cost more than 3 % of a step."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MODGPU_BENCH_GBP="6", MASTER_PORT=str(_free_port()))
    a = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--no-cpu", "--no-other"],
                       capture_output=True, text=True, timeout=900, env=dict(env, MODGPU_BENCH_FORCE_DIST="1"), cwd=ROOT)
    assert a.returncode == 0, a.stderr[-2000:]
    ja, _, _ = bench_lines(a.stdout)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--only", "c4_block"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert b.returncode == 0, b.stderr[-2000:]
    jb = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])["result"]
    assert ja["collective"]["matches_local_sums"] is True and ja["config"]["modset_entries"] == jb["modset_entries"]
    assert abs(ja["value"] / jb["value"] - 1) < 0.03, (ja["value"], jb["value"], ja["single_gpu_block_gbps"])


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", sorted({2, max(2, _gpu_count())}))
def test_bench_on_every_gpu_of_the_box(n):
    """`python bench.py --gpus N` on a box that has N > 1 GPUs (skipped on a one-GPU box, so an 8-GPU run of this suite covers
    the RCCL leg by itself): one rank per GPU, block r of config 4 on rank r, the histogram all-reduce; the summed histogram
    holds every rank's entries, every rank's block passes its prefix check, and N GPUs do at least 0.8 x N x what one does
    alone on the same workload.  Reference: SURVEY §8(e), north_star (reads shard, modset per GPU)."""
    if _gpu_count() < n:
        pytest.skip("needs %d GPUs, this box has %d" % (n, _gpu_count()))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=2400, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j, full, lines = bench_lines(r.stdout)
    assert len(lines[-1]) < 8192 and j["roofline"]["frac"] > 0
    assert j["n_gpus"] == n and j["scaling"] == "weak"
    assert j["collective"]["matches_local_sums"] is True and j["per_rank_parity"] is True
    assert j["value"] >= 0.8 * n * j["single_gpu_block_gbps"], (j["value"], j["single_gpu_block_gbps"])
    c3 = full["other_configs"]["c3_sharded"]
    assert "error" not in c3 and c3["n_gpus"] == n and c3["value"] > 0
    # the same from C: examples/multi_gpu.c (one process, a host thread per GPU, mgCommInitAll / mgHistogramAllReduce /
    # mgModsetMergeRankOrder on librccl): the all-reduced histogram is the sum of the GPUs' own, and the per-GPU sets merged in rank
    # order are, bit for bit, the set one stream over all the blocks builds
    import tempfile, pathlib
    from tests.test_example import build_multi_gpu
    with tempfile.TemporaryDirectory() as td:
        exe = build_multi_gpu(pathlib.Path(td))
        r = subprocess.run([exe, str(n), "200"], capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0 and "MULTI_GPU_OK" in r.stdout, r.stdout[-800:] + r.stderr[-800:]

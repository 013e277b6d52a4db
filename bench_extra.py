#!/usr/bin/env python3
"""bench_extra.py -- the legs of the benchmark that are NOT the driver's line: what a caller waits for around the kernels
(host bytes, files, result mirrors, .mod writing, the reference's unmodified programs on the library) and the secondary
paths (minimizers, repeat-rich genomes).  `python bench.py --full` runs them after the headline and writes them, with
everything else, to profiles/bench_detail_last.json; the driver's command does not set the flag, so its run is the
headline workload, the CPU baseline and one {value, ms, frac} triple per other BASELINE config.

Every function takes the bench's context `cx` (torch, library handle, device, stream; `cx.B` is bench.py's module: its
generators and timers)."""
import ctypes as C
import json
import os
import subprocess
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))

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
        reads, d_offsets, offsets, n_reads = cx.B.make_reads(cx, total, genome, gb, 21, 0.05 if name != "poly_a" else 0.0, 22)
        del genome, host_genome
        ms = mg.modsetCreate(sh, bits)
        n_hash = C.c_uint64(0)

        def step():
            mg.check(L.mgModsetClear(ms, cx.stream))
            mg.check(L.mgAddReadsDevice(ms, reads.data_ptr(), total, d_offsets.data_ptr(), n_reads, C.byref(n_hash), cx.stream))
        dt, kern, table, regions = cx.B.best_of_two(cx, step, steps)
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
        genome = cx.B.make_genome(cx, genome_bases, 333)
        nl = torch.full((seq_len // 80, 1), 10, dtype=torch.uint8, device=cx.dev)
        with open(rpath, "wb") as f:
            for i in range(n_seq):
                # (sequence i starts on a word boundary: seq_len is a multiple of 16)
                view = genome[i * seq_len // 16:]
                t_ = torch.cat([letters_of(view, seq_len).view(-1, 80), nl], dim=1).cpu().numpy()
                f.write(b">chr%d\n" % (i + 1)); f.write(t_.tobytes())
        del nl
        reads, d_offsets, offsets, n_reads = cx.B.make_reads(cx, q_bases, genome, genome_bases, 4242, 0.05, 5252)
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
                table = cx.B.read_profile(L, mg); L.mgProfileEnable(0)
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



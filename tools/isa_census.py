#!/usr/bin/env python3
"""dev: instruction census of one kernel of a .hip file by basic block (hipcc -S, gfx950): VALU instructions by issue class
(the classes tools/ubench.hip / ubench2.hip measured, profiles/r02_ubench_valu_rates.txt), with each block's loop depth.
usage: python tools/isa_census.py mg_scan.hip '_Z12mgScanKernelILi4ELb0EEv10MgScanArgs'"""
import collections, json, os, re, subprocess, sys
# lane-ops/s measured on MI355X, 8 independent chains x 8 waves per SIMD (profiles/r02_ubench_valu_rates.txt, profiles/r06_ubench3_valu_rates.txt):
#   full  v_add / v_sub / v_and / v_or / v_xor / v_mov / v_ashrrev_i32: 62-66 T
#   half  everything else priced -- v_mul_lo/hi, u24 products, v_alignbit, 32- AND 64-bit shifts, v_cmp (32 and 64 bit), v_addc, v_min/max, v_add3,
#         v_lshl_add (32 and 64), v_bfrev, v_perm, v_cndmask with its mask in an SGPR pair or in a VCC a v_cmp wrote: 36-38 T (v_bitop3: 42)
#   mad64 v_mad_u64_u32: 30 T in ubench3's harness (17 T in ubench.hip's form, one accumulator chain per lane); the faster figure is used, so the ceiling
#         is the higher one and a fraction of it the lower
RATE = {"full": 62.0e12, "half": 37.6e12, "mad64": 30.0e12}
FULL = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_ashrrev_i32", "v_not_b32")
def klass(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in ("v_mad_u64_u32", "v_mad_i64_i32"): return "mad64"
    if base in FULL: return "full"
    return "half"
def weighted(counts):
    """(wave instructions/s a SIMD set of 1024 issues this mix at, seconds per lane for the mix)"""
    n = sum(counts.values()); t = sum(c / RATE[k] for k, c in counts.items())
    return (n / t / 64.0 if t else 0.0), t
def census(src, sym, extra=()):
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "modimizer_amd", "csrc")
    out = "/tmp/%s.census.s" % os.path.basename(src)
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(csrc, "..", "..", "include"), "-I" + csrc,
                    "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", out] + list(extra), check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    blocks, cur = [], {"name": "entry", "depth": 0, "ops": collections.Counter(), "valu": collections.Counter()}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            d = re.search(r"Depth=(\d+)", l)
            cur = {"name": m.group(1), "depth": int(d.group(1)) if d else None, "ops": collections.Counter(), "valu": collections.Counter()}
            continue
        d = re.search(r"Depth=(\d+)", l)
        if d and l.strip().startswith(";") and cur["depth"] is None:
            cur["depth"] = int(d.group(1))
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        cur["ops"][op] += 1
        if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_nop")):
            cur["valu"][klass(op)] += 1
    blocks.append(cur)
    return blocks
if __name__ == "__main__":
    blocks = census(sys.argv[1], sys.argv[2], sys.argv[3:])
    tot = collections.Counter()
    for b in blocks:
        n = sum(b["valu"].values())
        if n >= 8:
            print("%-12s depth %s  VALU %4d  %s" % (b["name"], b["depth"], n, dict(b["valu"])))
        tot.update(b["valu"])
    print("whole kernel (static):", dict(tot), sum(tot.values()))
    body = max(blocks, key=lambda b: sum(b["valu"].values()))
    rate, t = weighted(body["valu"])
    print("largest block %s: %s = %d VALU; issued at %.4g wave instructions/s by this mix (uniform half rate: %.4g)" % (body["name"], dict(body["valu"]), sum(body["valu"].values()), rate, RATE["half"] / 64))
    print(json.dumps({"block": body["name"], "valu": dict(body["valu"]), "ops": dict(body["ops"].most_common(40))}))

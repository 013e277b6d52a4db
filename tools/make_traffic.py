#!/usr/bin/env python3
"""profiles/traffic.json and profiles/scan_issue.json from the summaries tools/prof_round.sh leaves under
gpurun_out/prof_<tag>_{c2,c3,c5}/ (also copies the summaries and kernel_stats into profiles/<tag>_<config>_*).
usage: tools/make_traffic.py r03"""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
tags = {"c2": "10", "c3": "c3", "c5": "c5", "c4": "12.5", "odd": "ref_default"}           # ("odd": tools/prof_pmc.sh <tag>_odd --only ref_default)           # the keys bench.py's roofline_of() looks a kernel's traffic up under
out = {"_note": "HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units) of bench.py: key '10' = the headline "
                "workload (config 2, 10 Gbp), 'c3' / 'c5' / '12.5' = bench.py --only c3 / c5 / c4_block.  FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 counts wide "
                "coalesced streaming reads at half) ONLY for the kernels whose reads are such streams ('x2'); for gather-dominated kernels (random rank "
                "records, random flag bytes, bucket probes) the raw counter is used ('raw').  Values are PER-LAUNCH averages: mgPartScatterKernel runs "
                "twice per step (both passes together: twice its value).  _step_bytes: all the library's kernels of a step (config 3: of a query batch), every launch counted.",
       "_from": "profiles/%s_{c2,c3,c5}_pmc_summary.txt" % tag}
for cfg, key in tags.items():
    d = os.path.join(R, "gpurun_out", "prof_%s_%s" % (tag, cfg))
    if not os.path.exists(os.path.join(d, "summary.json")):
        continue
    j = json.load(open(os.path.join(d, "summary.json")))
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(R, "profiles", "%s_%s_pmc_summary.txt" % (tag, cfg)))
    if os.path.exists(os.path.join(d, "kernel_stats.csv")):
        shutil.copy(os.path.join(d, "kernel_stats.csv"), os.path.join(R, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)))
    steps = max(j.get("mgScanKernel", {}).get("calls", 1), 1) / (9.0 if cfg == "c3" else 1.0)      # launches of the scan = steps profiled (config 3: nine batches a "step")
    out.setdefault("_step_bytes", {})[key] = round(sum(v["hbm_bytes"] * v["calls"] for k, v in j.items() if k.startswith("mg") and "Synth" not in k and "hbm_bytes" in v) / steps / (9.0 if cfg == "c3" else 1.0))
    for k, v in j.items():
        if "hbm_bytes" in v:
            if k == "mgPartHistBytesKernel":                # the library's profile slot (and bench.py's table) calls the second pass's counts mgPartHistKernel whichever kernel made them
                k = "mgPartHistKernel"
                if not v.get("fetch_correction", "").startswith("x2"):
                    v = dict(v, hbm_bytes=v.get("hbm_bytes_fetch_x2", v["hbm_bytes"]), fetch_correction="x2 (wide coalesced streaming reads)")
            e = out.setdefault(k, {})
            e[key] = v["hbm_bytes"]
            e["fetch"] = "x2" if v.get("fetch_correction", "").startswith("x2") else "raw"
    if cfg == "c2":
        s = j["mgScanKernel"]
        starts = 1e10
        old = {}
        try: old = json.load(open(os.path.join(R, "profiles", "scan_issue.json")))
        except Exception: pass
        si = {"_from": "profiles/%s_c2_pmc_summary.txt (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... on bench.py, 10 Gbp) and profiles/r02_ubench_valu_rates.txt (tools/ubench.hip)" % tag,
              "kernel": "mgScanKernel<FAST, k-mers only>", "starts_per_launch": starts}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"):
            if c in s: si[c] = s[c]
        si["valu_per_start"] = round(s["SQ_INSTS_VALU"] * 64 / starts, 3)
        si["salu_per_start"] = round(s["SQ_INSTS_SALU"] * 64 / starts, 3)
        si["measured_int_valu_peak_lane_ops_per_s"] = old.get("measured_int_valu_peak_lane_ops_per_s", 37.6e12)
        si["measured_int_valu_peak_wave_insts_per_s"] = old.get("measured_int_valu_peak_wave_insts_per_s", 587.5e9)
        si["avg_ms_under_trace"] = s.get("avg_ms")
        hist = dict(old.get("history", {"round 1": 13.2, "round 2": 11.87, "round 3": 10.9}))
        hist["round %d" % int(tag.lstrip("r"))] = si["valu_per_start"]
        si["history"] = hist
        if "ref_default" in old: si["ref_default"] = old["ref_default"]      # the exact-mode scan's own pass (tools/prof_pmc.sh --only ref_default)
        if "census" in old: si["census"] = old["census"]                     # tools/isa_census.py's class mix and the ceiling it prices (round 6)
        json.dump(si, open(os.path.join(R, "profiles", "scan_issue.json"), "w"), indent=1)
# the exact-mode scan's counters (bench.py --only ref_default): VALU per start into scan_issue.json's ref_default entry
d = os.path.join(R, "gpurun_out", "prof_%s_odd" % tag, "summary.json")
if os.path.exists(d):
    j = json.load(open(d)); sip = os.path.join(R, "profiles", "scan_issue.json"); si = json.load(open(sip))
    s = j["mgScanKernel"]; rd = si.setdefault("ref_default", {})
    rd.update({"from": "profiles/%s_odd_pmc_summary.txt (rocprofv3 --pmc SQ_INSTS_VALU ... on bench.py --only ref_default, 10 Gbp)" % tag,
               "SQ_INSTS_VALU": s["SQ_INSTS_VALU"], "SQ_INSTS_SALU": s.get("SQ_INSTS_SALU"), "valu_per_start": round(s["SQ_INSTS_VALU"] * 64 / 1e10, 2), "avg_ms_under_trace": s.get("avg_ms")})
    json.dump(si, open(sip, "w"), indent=1)
json.dump(out, open(os.path.join(R, "profiles", "traffic.json"), "w"), indent=1)
print("wrote profiles/traffic.json, profiles/scan_issue.json;", {k: v for k, v in out.items() if k in ("mgScanKernel", "mgTableFindSegKernel", "mgPartScatterKernel")})

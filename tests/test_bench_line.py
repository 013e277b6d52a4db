"""The driver's bench line: round 5's one-line JSON had grown to 24 KB and the driver could not parse
it out of its stdout tail.  bench.compact_line makes the LAST stdout line; these tests feed recorded whole results of earlier
rounds (profiles/r0N_bench.json, copies of what the box printed) through it."""
import copy
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

RECORDED = [p for p in ("r05_bench.json", "r04_bench.json", "r03_o_bench.json") if os.path.exists(os.path.join(ROOT, "profiles", p))]


def _load(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


@pytest.mark.parametrize("name", RECORDED)
def test_recorded_results_make_a_small_line_with_the_contract_keys(name):
    full = _load(name)
    line = bench.compact_line(full, "profiles/bench_detail_last.json")
    assert "\n" not in line and len(line) < bench.LINE_LIMIT == 8192
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in j and j[k] == full[k], k
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] == full["roofline"]["frac"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    c = j["cpu_baseline"]
    assert c["value"] == full["cpu_baseline"]["value"] > 0 and c["cores"] == 1 and c["kind"] in ("reference", "port") and c["sample"]
    for cfg, o in j.get("other_configs", {}).items():
        assert len(json.dumps(o)) < 400, cfg                # a triple (+ the dominant kernel), not a table
        if cfg in ("c5", "c3", "c4_block", "ref_default"):
            assert o["value"] == full["other_configs"][cfg]["value"] and o["ms"] > 0


def test_line_stays_small_whatever_the_legs_return():
    """a leg that returns a huge object (or many legs) cannot push the line over the limit: the optional parts are dropped first"""
    full = copy.deepcopy(_load(RECORDED[0]))
    full["other_configs"] = {"leg%d" % i: {"value": float(i), "unit": "Gbp/s", "ms_per_step": 1.0, "whole_step": {"frac": 0.1},
                                             "roofline": {"kernel": "mgScanKernel", "frac": 0.1}, "junk": "x" * 5000} for i in range(200)}
    full["end_to_end"] = {"leg%d" % i: {"Gbp_per_s": 1.0, "what": "y" * 3000} for i in range(300)}
    line = bench.compact_line(full, None)
    j = json.loads(line)
    assert len(line) < bench.LINE_LIMIT and j["roofline"]["frac"] and j["cpu_baseline"]["value"] and j["value"] == full["value"]


def test_multi_rank_line_keeps_the_checks():
    """the N > 1 line (config 4: a block per GPU + the histogram all-reduce) through the same formatter"""
    full = copy.deepcopy(_load(RECORDED[0]))
    full.pop("cpu_baseline"); full.pop("end_to_end", None)
    full.update(n_gpus=8, collective={"what": "w" * 500, "allreduce_ms": 0.05, "histogram_entries": 8, "entries_all_ranks": 8, "matches_local_sums": True},
                per_rank_parity=True, per_rank_parity_rank0={"ok": True}, single_gpu_block_gbps=1100.0,
                other_configs={"c3_sharded": {"value": 9000.0, "unit": "Gbp/s", "n_gpus": 8, "seed_hit_fraction": 0.34, "ms_each_batch_rank0": [1.0] * 9, "workload": "z" * 600}})
    line = bench.compact_line(full, None)
    j = json.loads(line)
    assert len(line) < 4096
    assert j["n_gpus"] == 8 and j["collective"]["matches_local_sums"] is True and j["per_rank_parity"] is True and j["single_gpu_block_gbps"] == 1100.0
    assert j["roofline"]["frac"] > 0 and j["other_configs"]["c3_sharded"] == {"value": 9000.0, "unit": "Gbp/s", "n_gpus": 8}


def test_emit_prints_the_compact_line_last(tmp_path, monkeypatch):
    full = _load(RECORDED[0])
    monkeypatch.setattr(bench, "HERE", str(tmp_path)); os.makedirs(tmp_path / "profiles")
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(full)
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2 and lines[0].startswith("BENCH_DETAIL ") and lines[1].startswith("{")
    assert json.loads(lines[0][len("BENCH_DETAIL "):]) == full
    assert json.loads(lines[1])["detail"] == "profiles/bench_detail_last.json" and len(lines[1]) < 8192
    assert json.load(open(tmp_path / "profiles" / "bench_detail_last.json")) == full

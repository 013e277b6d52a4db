#!/usr/bin/env python3
"""print per-kernel ms/step from a bench.py JSON line on stdin (dev helper)"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if line.startswith("{"):
        j = json.loads(line)
        print(tag, j["value"], "Gbp/s", j["ms_per_step"], "ms/step", j["roofline"]["kernels_ms_per_step"])

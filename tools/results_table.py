#!/usr/bin/env python3
"""Prints the results table of DESIGN.md 6 / README.md from profiles/r03[_configN]_bench.json and the kernel traces beside them."""
import csv, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
for cfg in (3, 2, 4, 5):
    suf = "" if cfg == 3 else "_config%d" % cfg
    j = json.loads(open(os.path.join(P, "r03%s_bench.json" % suf)).read().strip().splitlines()[-1])
    r, c = j["roofline"], j["config"]
    fs = c.get("frame_steps_ms") or {}
    stats = list(csv.DictReader(open(os.path.join(P, "r03%s_kernel_stats.csv" % suf))))
    top = stats[0]
    cb = j.get("cpu_baseline") or {}
    print("config %d: %.4f ms per frame of a run (%s step; other %.4f), %.2f Grays/s, kernel %s %.4f ms by events / %.4f ms traced (%s launches), frac %.4f (frame %.4f), traffic %.1f MB, cpu %.2f Mrays/s on %s of %s threads, x%.0f" % (
        cfg, j["ms_per_step"], fs.get("timed_step"), fs.get("other") or 0.0, j["value"] / 1e3, top["Name"][:40], r["kernel_ms"], float(top["TotalDurationNs"]) / float(top["Calls"]) / 1e6, top["Calls"],
        r["frac"], r["frame"]["frac"], (r["traffic"] or 0) / 1e6, cb.get("value", 0), cb.get("cores"), cb.get("host_cores"), c.get("gpu_over_cpu", 0)))

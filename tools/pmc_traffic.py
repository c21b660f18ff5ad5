#!/usr/bin/env python3
"""HBM traffic of one headline frame over ALL its kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py --steps 2 --warmup 1 --no-cpu-baseline`, written with the git blob hashes of the kernel sources it was measured on
(bench.py reports it only while those hashes match).  Usage: pmc_traffic.py DIR_WITH_THE_TWO_PASSES OUT.json"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> per-dispatch values
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = defaultdict(float)
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "skr_" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                per[(row["Kernel_Name"], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
per_kernel, total = {}, 0.0
for k, cs in sorted(acc.items()):
    fetch = sum(cs["FETCH_SIZE"]) / max(1, len(cs["FETCH_SIZE"]))   # KiB per launch (one launch of each kernel per frame)
    write = sum(cs["WRITE_SIZE"]) / max(1, len(cs["WRITE_SIZE"]))
    b = (2.0 * fetch + write) * 1024.0  # MI355X_MICROARCH.md HBM: FETCH_SIZE reports half the bytes of a coalesced read stream on gfx950
    per_kernel[k.replace("void ", "").replace("(RenderParams)", "")] = {"fetch_kib": fetch, "write_kib": write, "bytes": b}
    total += b
json.dump({"variant": "node_levels_v5", "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (separate passes: tools/pmc_pass.sh)",
           "correction": "MI355X_MICROARCH.md HBM: FETCH_SIZE doubled (gfx950 reports half of a coalesced read stream), WRITE_SIZE as is; KiB -> bytes",
           "per_kernel": per_kernel, "traffic_bytes_per_frame": total, "sources": bench.source_hashes()}, open(out, "w"), indent=1)
print(json.dumps(per_kernel, indent=1)); print("total MB per frame: %.1f" % (total / 1e6))

#!/usr/bin/env python3
"""HBM traffic of one frame over ALL its kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
`bench.py [--config C] --steps 2 --warmup 1 --no-cpu-baseline`, written with the git blob hashes of the kernel sources it was
measured on (bench.py reports it only while those hashes match).  Every dispatch of every skr_ kernel is summed and divided by the
frames that command asked the device for — bench.py counts them (config.frames_enqueued in its JSON line, found in the passes' logs: warm-up,
timed frames, the other frame step's side pass, the kernel-timing pass, a mesh's counting frame) —, so configurations whose frame takes several
launches of a kernel (AA samples, bands) are counted whole.
Usage: pmc_traffic.py DIR_WITH_THE_TWO_PASSES OUT.json [CONFIG [VARIANT]]"""
import csv, glob, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

root, out = sys.argv[1], sys.argv[2]
config = int(sys.argv[3]) if len(sys.argv) > 3 else 3
variant = sys.argv[4] if len(sys.argv) > 4 else ("node_levels_v5" if config in (3, 5) else "direct_v3")
FRAMES = None
for log in sorted(glob.glob(os.path.join(root, "*.log"))):
    for line in open(log, errors="replace"):
        if line.startswith("{") and '"frames_enqueued"' in line:
            try:
                n = int(json.loads(line)["config"]["frames_enqueued"])
            except Exception:
                continue
            assert FRAMES in (None, n), "the passes rendered different numbers of frames: %s and %s" % (FRAMES, n)
            FRAMES = n
if FRAMES is None:
    sys.exit("no bench.py JSON line with config.frames_enqueued in %s/*.log" % root)
acc = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum over dispatches
launches = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "skr_" in row["Kernel_Name"] and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[(row["Kernel_Name"], row["Counter_Name"])].add(row["Dispatch_Id"])
per_kernel, total = {}, 0.0
for k, cs in sorted(acc.items()):
    fetch, write = cs["FETCH_SIZE"] / FRAMES, cs["WRITE_SIZE"] / FRAMES   # KiB per frame
    b = (2.0 * fetch + write) * 1024.0  # MI355X_MICROARCH.md HBM: FETCH_SIZE reports half the bytes of a coalesced read stream on gfx950
    per_kernel[k.replace("void ", "").replace("(RenderParams)", "")] = {"fetch_kib": fetch, "write_kib": write, "bytes": b, "launches_per_frame": len(launches[(k, "FETCH_SIZE")]) / FRAMES}
    total += b
json.dump({"variant": variant, "config": config,
           "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -- python3 bench.py --config %d --steps 2 --warmup 1 --no-cpu-baseline (separate passes: tools/pmc_pass.sh); per frame = all dispatches / %d frames" % (config, FRAMES),
           "correction": "MI355X_MICROARCH.md HBM: FETCH_SIZE doubled (gfx950 reports half of a coalesced read stream), WRITE_SIZE as is; KiB -> bytes",
           "per_kernel": per_kernel, "traffic_bytes_per_frame": total, "sources": bench.source_hashes()}, open(out, "w"), indent=1)
print(json.dumps(per_kernel, indent=1)); print("total MB per frame: %.1f" % (total / 1e6))

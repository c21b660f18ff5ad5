#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark on MI355X.

Metric (BASELINE.json): Mrays/sec + frame ms, scenes/spheres2.scn 1920x1080 --gillum 16
--shadow (depth 3), on 1/2/4/8 MI355X.  A "step" is one whole frame: every rank renders its
interleaved row tiles with the HIP megakernel (C ABI, include/skr.h), the u8 tiles are gathered
to rank 0 over RCCL and de-interleaved there.  Inputs (the SoA scene) are resident in HBM
before the timed region.  `value` = radiance rays actually traced per second, whole job
(rays = shade() calls with depth > 0, counted by the kernel itself; deterministic and
partition-independent).  SURVEY.md §8d's closed form W*H*S*sum N^k is the full-tree upper
bound (every ray hitting a sphere) and is reported beside it as `nominal_rays`.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE = os.path.join(ROOT, "tests", "golden", "scenes", "spheres2.scn")
W, H = 1920, 1080
KW = dict(gillum=16, shadow=True, depth=3, seed=20261004)
TILE_ROWS = 8  # interleaved row tiles (cost is very non-uniform vertically: sky rows vs ground rows)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector


def reference_sample(orc):
    """The reference's own shade()/parseScene() (oracle/_ref/ref_render: its sources compiled in place in the build
    container, serial entry — the only one that can run this configuration) on ONE core, on a 320x180 sample of the
    headline configuration; the ray count comes from the oracle's replay mode, which is bit-identical to it."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_render")
    if not os.path.exists(exe):
        return None
    import subprocess
    import tempfile
    w, h = 320, 180
    try:
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            subprocess.run([exe, "--path", SCENE, "--output", os.path.join(tmp, "ref.ppm"), "--width", str(w), "--height", str(h), "--gillum", str(KW["gillum"]),
                            "--shadow", "--depth", str(KW["depth"]), "--seed", "1"], check=True, capture_output=True, timeout=120, cwd=tmp)
            dt = time.perf_counter() - t0
        _, _, st = orc.render(SCENE, w, h, rng=orc.RNG_GLIBC_REPLAY, math=orc.MATH_LIBM, gillum=KW["gillum"], shadow=KW["shadow"], depth=KW["depth"], seed=1)
        return {"value": int(st[0]) / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/ref_render %dx%d --gillum %d --shadow: %d radiance rays in %.2f s (process start and scene parse included)" % (w, h, KW["gillum"], int(st[0]), dt)}
    except Exception as e:  # the checker binary is optional; the port above is the baseline
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def cpu_baseline():
    """The oracle (CPU restatement, counter RNG, OpenMP over (row, 32-pixel span) items) timed on this
    box's host cores on the same workload: whole frames, repeated until >= 8 s of wall time."""
    from oracle import pyoracle as orc
    cores = orc.host_cores()
    scene = orc.OracleScene(SCENE)
    kw = dict(rng=orc.RNG_COUNTER, math=orc.MATH_SHARED, threads=cores, gillum=KW["gillum"], shadow=KW["shadow"],
              depth=KW["depth"], seed=KW["seed"])
    orc.render(scene, W, H, y0=0, y1=64, **kw)  # warm up the thread pool
    rays, frames = 0, 0
    t0 = time.perf_counter()
    while frames < 2 or time.perf_counter() - t0 < 8.0:
        _, _, st = orc.render(scene, W, H, **kw)
        rays += int(st[0])
        frames += 1
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    reference = reference_sample(orc)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port", "cpu_model": model, "reference_1core": reference,
            "ms_per_frame": dt / frames * 1e3,
            "sample": "oracle/liboracle.so (C restatement of the reference path, gcc -O2, OpenMP %d threads = this box's cgroup CPU "
                      "quota), %d whole frames of the same workload: %d radiance rays in %.2f s" % (cores, frames, rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import skele_raytracer_amd as skr

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path)"
    # SKR_REHEARSE_GLOO=1: every rank on GPU 0 and the collectives over gloo — how the N > 1 path (partition, gather,
    # de-interleave, rank reductions) is rehearsed on a one-GPU box (tests/test_gpu_parity.py); never a measurement
    rehearsal = world > 1 and os.environ.get("SKR_REHEARSE_GLOO") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    from skele_raytracer_amd.distributed import FrameSharder

    scene = skr.parse_scene(SCENE)
    r = skr.Renderer(scene, local_rank)
    opt = skr.Options(W, H, **KW)
    sharder = FrameSharder(W, H, TILE_ROWS, rank, world, dev)  # interleaved row tiles + one RCCL all-gather
    k_max = sharder.k_max
    stream = torch.cuda.current_stream(dev)
    ev_box = [None]

    def render_into(buf):
        ev = ev_box[0]
        if ev:
            ev[0].record(stream)
        r.render_tiles_into(opt, TILE_ROWS, rank, world, buf.data_ptr(), None, stream.cuda_stream)
        if ev:
            ev[1].record(stream)

    def step(ev=None):
        ev_box[0] = ev
        sharder.step(render_into)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    r.counters(reset=True)
    r.kernel_timing(True)
    r.kernel_ms()  # drop the warm-up launches
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    sync()
    dt = time.perf_counter() - t0

    variant = r.kernel_variant()
    queued = r.last_parent_count() if variant in ("parent_queue_v3", "level_queues_v4") else 0  # before the counters are reset
    level1 = r.last_level1_count() if variant == "level_queues_v4" else 0
    cnt = r.counters(reset=True)
    pipeline_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps)  # everything this rank enqueues per frame before the collective
    kernel_ms, timed_launches = r.kernel_ms()                                      # the dominant kernel alone (HIP events on its stream)
    stats = torch.tensor([dt, float(cnt["radiance_rays"]), float(cnt["shadow_rays"]), float(cnt["sphere_hits"]), kernel_ms, pipeline_ms],
                         dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, kernel_ms, pipeline_ms = float(mx[0]), float(mx[4]), float(mx[5])
        rays, shadow, hits = float(sm[1]), float(sm[2]), float(sm[3])
    else:
        rays, shadow, hits = float(stats[1]), float(stats[2]), float(stats[3])

    if rank == 0:
        rays_per_frame = rays / args.steps
        info = scene.info
        # algorithmic HBM bytes of one launch of the dominant kernel on this rank:
        #  single megakernel: its share of the u8 framebuffer + one read of the scene
        #  parent-queue pipeline: the GI kernel reads one 64-byte record per primary hit and writes that pixel (3 B)
        scene_bytes = info.n_spheres * 64 + info.n_point_lights * 32 + info.n_triangles * 48
        frame_bytes = W * min(H, k_max * TILE_ROWS) * 3
        n_parents = queued
        if level1:   # leaf kernel of the level-queue pipeline: one 64-byte record read and one 12-byte slot written per level-1 hit
            launch_bytes = level1 * (64 + 12) + scene_bytes
        else:        # GI kernel of the parent-queue pipeline, or the single megakernel
            launch_bytes = (n_parents * (64 + 3) + scene_bytes) if n_parents else (frame_bytes + scene_bytes)
        achieved_gbs = launch_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from PMC counters cannot be collected from inside this process; the figure of the
        # last committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command is reported (N=1 only)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")) as f:
                tj = json.load(f)
            if world == 1 and tj.get("variant") == r.kernel_variant():
                traffic = tj["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        # algorithmic flops: 34 flop per ray-sphere test; per radiance ray n_sph tests,
        # per shadow ray at most n_sph (early-out ignored => upper bound), ~150 flop shading per hit
        alg_flop = (rays_per_frame * info.n_spheres + shadow / args.steps * info.n_spheres) * 34 + hits / args.steps * 150
        out = {
            "metric": "Mrays/sec + frame ms, 1920x1080 gillum=16 spheres2.scn",
            "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "reference scene file scenes/spheres2.scn (spherical_fog line skipped: UB in the reference)",
            "config": {"workload": "scenes/spheres2.scn 1920x1080 --gillum 16 --shadow --depth 3 (BASELINE.json configs[2])",
                       "rays_per_frame": rays_per_frame, "nominal_rays": skr.radiance_ray_count(opt),
                       "nominal_mrays_per_s": skr.radiance_ray_count(opt) * args.steps / dt / 1e6,
                       "shadow_rays_per_frame": shadow / args.steps, "partition": "interleaved %d-row tiles, rank = tile %% %d" % (TILE_ROWS, world),
                       "gather": ("REHEARSAL on one GPU over gloo - not a measurement" if rehearsal else "RCCL all-gather of the u8 tile buffers, rank 0 de-interleaves") if world > 1 else "none (1 GPU)",
                       "kernel": r.kernel_variant(), "seed": KW["seed"]},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; FETCH doubled per the gfx950 note)" if traffic else None,
                         "kernel": {"level_queues_v4": "skr_leaf_kernel<false>", "parent_queue_v3": "skr_gi_kernel<3, 3, false>", "wave_streaming_v2": "skr_wave_kernel<3, 3>"}.get(r.kernel_variant(), "skr_render_kernel<3>"), "kernel_ms": kernel_ms, "render_ms": pipeline_ms, "algorithmic_bytes_per_launch": launch_bytes, "frame_bytes": frame_bytes, "queued_parents": n_parents, "queued_level1_hits": level1,
                         "note": "algorithmic HBM bytes of the dominant kernel: the level-1 hit records it reads (64 B) and the slots it writes (12 B) + ~1 KB of scene; this path is FP32-VALU bound, see roofline_valu"},
            "roofline_valu": {"bound": "fp32_valu", "achieved": alg_flop / (kernel_ms * 1e-3) / 1e12 / world * 1.0,
                              "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": alg_flop / (kernel_ms * 1e-3) / 1e12 / world / VALU_PEAK_TFLOPS,
                              "note": "algorithmic flops (34/sphere test, 150/shaded hit; shadow early-outs ignored) per GPU / peak FP32 vector"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

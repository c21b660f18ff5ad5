#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark on MI355X, and the other BASELINE.json configurations measured the same way.

Metric (BASELINE.json): Mrays/sec + frame ms, scenes/spheres2.scn 1920x1080 --gillum 16 --shadow (depth 3), on 1/2/4/8
MI355X.  A "step" is one whole frame: every rank renders its interleaved row tiles (C ABI, include/skr.h), the u8 tiles are
gathered with ONE RCCL all-gather over xGMI and rank 0 de-interleaves them on the device — the whole step runs inside libskr
(skr_comm_render_frame); torch.distributed only carries the 128-byte RCCL id and the barriers.  Inputs (the SoA scene) are
resident in HBM before the timed region.  `value` = radiance rays actually traced per second, whole job (rays = shade() calls
with depth > 0, counted by the kernels themselves; deterministic and partition-independent).  SURVEY.md 8d's closed form
W*H*S*sum N^k is the full-tree upper bound and is reported beside it as `nominal_rays`.

  python bench.py [--config 2|3|4|5] [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--config picks the workload (BASELINE.json `configs`, 1-based as SURVEY.md 8d numbers them; default 3 = the headline):
  2  spheres2.scn 1920x1080 --jsample 5 --shadow            3  spheres2.scn 1920x1080 --gillum 16 --shadow
  4  dragon.scn   1920x1080 --gillum 16 (10 002 triangles)  5  spheres2.scn 3840x2160 --gillum 64 --jsample 5 --shadow

Prints ONE JSON line on rank 0.  `roofline` is the bound that binds (FP32 VALU issue) for the dominant kernel: its own counted
work over its own mean launch duration (HIP events on its stream); `roofline_hbm` is the figure north_star asks for (algorithmic
bytes over the frame time, with the PMC-measured traffic beside it); `cpu_baseline` is the oracle on this box's host cores on a
stated sample of the same workload.  A fallback from the frame step that was asked for is an error, not a note.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
SEED = 20261004
# cpu_bands: the rows the CPU baseline renders — None = whole frames; (period, rows) = rows y with y % period < rows (spread over the
# frame, whose cost is very non-uniform vertically), the rate scaled from the rays those rows trace (BASELINE.md 3 allows a subset)
CONFIGS = {
    2: dict(scene="spheres2.scn", w=1920, h=1080, kw=dict(jsample=5, shadow=True, depth=3), steps=300, warmup=20, cpu_bands=None,
            workload="scenes/spheres2.scn 1920x1080 --jsample 5 --shadow (BASELINE.json configs[1])"),
    3: dict(scene="spheres2.scn", w=1920, h=1080, kw=dict(gillum=16, shadow=True, depth=3), steps=600, warmup=30, cpu_bands=None,
            workload="scenes/spheres2.scn 1920x1080 --gillum 16 --shadow --depth 3 (BASELINE.json configs[2])"),
    4: dict(scene="dragon.scn", w=1920, h=1080, kw=dict(gillum=16, depth=3), steps=300, warmup=20, cpu_bands=(120, 4),
            workload="scenes/dragon.scn 1920x1080 --gillum 16 (BASELINE.json configs[3]; 10 002 triangles, no spheres: shade() never recurses)"),
    5: dict(scene="spheres2.scn", w=3840, h=2160, kw=dict(gillum=64, jsample=5, shadow=True, depth=3), steps=2, warmup=1, cpu_bands=(540, 1),
            workload="scenes/spheres2.scn 3840x2160 --gillum 64 --jsample 5 --shadow (BASELINE.json configs[4]; the whole frame on however many GPUs --gpus names)"),
}
TILE_ROWS = 8  # interleaved row tiles (cost is very non-uniform vertically: sky rows vs ground rows)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector
# flops per ray-sphere test, per ray-triangle test (precomputed edges: 46), per shaded hit; a culling-sphere test of the
# triangle walk (shade_common.h line_touches: cross, dot, one product, one compare) is 19
FLOP_SPHERE_TEST, FLOP_TRI_TEST, FLOP_TRI_TEST_REF, FLOP_CULL_TEST, FLOP_SHADED_HIT = 34, 46, 58, 19, 150
# the files whose contents decide what the kernels move: the measured traffic is only reported for the exact sources it was measured on
KERNEL_SOURCES = ["skele_raytracer_amd/csrc/render_nodes.hip", "skele_raytracer_amd/csrc/render_wave.hip", "skele_raytracer_amd/csrc/render_generic.hip", "skele_raytracer_amd/csrc/wave_common.h",
                  "skele_raytracer_amd/csrc/shade_common.h", "skele_raytracer_amd/csrc/device_math.h", "skele_raytracer_amd/csrc/render_params.h"]
DOMINANT = {"node_levels_v5": "skr_leaf_kernel2<false, false>", "node_levels_v5_flat": "skr_trace_kernel<false> (last level) + skr_shade_leaf_kernel<false>",
            "direct_v3": "skr_direct_kernel", "level_pipeline_g1": "skr_gtrace_kernel + skr_gactivate_kernel + skr_gfinalize_kernel (all levels of the last band)"}


def traffic_json(config):
    return os.path.join(ROOT, "profiles", "r03_hbm_traffic.json" if config == 3 else "r03_hbm_traffic_config%d.json" % config)


def git_blob_hash(path):
    """`git hash-object` of a file: sha1 of 'blob <len>\\0' + contents."""
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes():
    return {p: git_blob_hash(os.path.join(ROOT, p)) for p in KERNEL_SOURCES}


def measured_traffic(config, variant):
    """HBM bytes per frame over ALL kernels of the frame from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this command (tools/pmc_traffic.py; FETCH doubled per MI355X_MICROARCH.md) — or None when the kernel sources have changed since."""
    path = traffic_json(config)
    try:
        with open(path) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return None, "no %s" % os.path.relpath(path, ROOT)
    if tj.get("variant") != variant:
        return None, "measured for kernel variant %s" % tj.get("variant")
    if tj.get("sources") != source_hashes():
        return None, "stale: the kernel sources have changed since the PMC passes (git blob hashes differ)"
    return tj, None


def reference_sample(orc, cfg):
    """The reference's own shade()/parseScene() (oracle/_ref/ref_render: its sources compiled in place in the build container,
    serial entry — the only one that honours these options) on ONE core, on a small frame of the same configuration; the ray
    count comes from the oracle's replay mode, which is bit-identical to it."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_render")
    if not os.path.exists(exe):
        return None
    import subprocess
    import tempfile
    kw = cfg["kw"]
    w, h = (320, 180) if "gillum" in kw and cfg["scene"] != "dragon.scn" and not kw.get("jsample") else (160, 90)
    if kw.get("gillum", 0) >= 64:
        w, h = 64, 36
    scene = os.path.join(SCENES, cfg["scene"])
    args = ["--path", scene, "--width", str(w), "--height", str(h), "--depth", str(kw["depth"]), "--seed", "1"]
    if "gillum" in kw:
        args += ["--gillum", str(kw["gillum"])]
    if kw.get("jsample"):
        args += ["--jsample", str(kw["jsample"])]
    if kw.get("shadow"):
        args += ["--shadow"]
    try:
        with tempfile.TemporaryDirectory() as tmp:
            t0 = time.perf_counter()
            subprocess.run([exe] + args + ["--output", os.path.join(tmp, "ref.ppm")], check=True, capture_output=True, timeout=300, cwd=tmp)
            dt = time.perf_counter() - t0
        _, _, st = orc.render(scene, w, h, rng=orc.RNG_GLIBC_REPLAY, math=orc.MATH_LIBM, seed=1, **kw)
        return {"value": int(st[0]) / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": "oracle/_ref/ref_render %dx%d %s: %d radiance rays in %.2f s (process start and scene parse included)" % (w, h, " ".join(args[6:-2]), int(st[0]), dt)}
    except Exception as e:  # the checker binary is optional; the port above is the baseline
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def cpu_baseline(cfg):
    """The oracle (CPU restatement, counter RNG, OpenMP over (row, 32-pixel span) items) timed on this box's host cores on the
    same workload: whole frames, or the stated rows of it, repeated until >= 8 s of wall time."""
    from oracle import pyoracle as orc
    cores = orc.host_cores()
    scene = orc.OracleScene(os.path.join(SCENES, cfg["scene"]))
    W, H = cfg["w"], cfg["h"]
    kw = dict(rng=orc.RNG_COUNTER, math=orc.MATH_SHARED, threads=cores, seed=SEED, **cfg["kw"])
    bands = [(0, H)]
    if cfg["cpu_bands"]:
        period, rows = cfg["cpu_bands"]
        bands = [(y, min(H, y + rows)) for y in range(period // 2, H, period)]
    orc.render(scene, W, H, y0=bands[0][0], y1=bands[0][1], **kw)  # warm up the thread pool
    rays, passes = 0, 0
    t0 = time.perf_counter()
    while passes < (1 if cfg["cpu_bands"] else 2) or time.perf_counter() - t0 < 8.0:
        for y0, y1 in bands:
            _, _, st = orc.render(scene, W, H, y0=y0, y1=y1, **kw)
            rays += int(st[0])
        passes += 1
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    host = os.cpu_count() or cores
    n_rows = sum(b - a for a, b in bands)
    what = ("%d whole frames" % passes) if not cfg["cpu_bands"] else ("%d passes over %d of the frame's %d rows (rows y with y %% %d in [%d, %d): spread over the frame; the rate is rays of those rows over their time)"
                                                                    % (passes, n_rows, H, cfg["cpu_bands"][0], cfg["cpu_bands"][0] // 2, cfg["cpu_bands"][0] // 2 + cfg["cpu_bands"][1]))
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "host_cores": host, "kind": "port", "cpu_model": model, "reference_1core": reference_sample(orc, cfg),
            "ms_per_frame": (dt / passes * 1e3) if not cfg["cpu_bands"] else (dt / passes * 1e3 * H / n_rows),
            "sample": "oracle/liboracle.so (C restatement of the reference path, gcc -O2, OpenMP) on %d of this host's %d hardware threads — the box's cgroup CPU quota; "
                      "scaled to all %d the baseline would be ~%.0fx higher — %s of the same workload: %d radiance rays in %.2f s"
                      % (cores, host, host, host / max(1, cores), what, rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configuration (default 3: the headline)")
    ap.add_argument("--steps", type=int, default=None)   # default per configuration: >= 1 s of frames at one GPU for the headline (the clock the chip settles at, not its first milliseconds)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-frames", action="store_true", help="(the default since round 3: the pipelined frame step; kept so that old command lines run)")
    ap.add_argument("--sync-frames", action="store_true", help="libskr's frame step with the collective on the render stream (skr_comm_render_frame) instead of pipelined behind the next frame")
    ap.add_argument("--no-side-pass", action="store_true", help="do not time the other frame step beside the timed one (profiling runs: a trace of --sync-frames --no-side-pass holds serial frames only)")
    ap.add_argument("--torch-gather", action="store_true", help="the round-1 frame step (torch.distributed all_gather + torch de-interleave) instead of libskr's")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.steps is None:
        args.steps = cfg["steps"]
    if args.warmup is None:
        args.warmup = cfg["warmup"]
    W, H, KW = cfg["w"], cfg["h"], dict(cfg["kw"], seed=SEED)
    SCENE = os.path.join(SCENES, cfg["scene"])

    import numpy as np
    import torch
    import torch.distributed as dist
    import skele_raytracer_amd as skr
    from skele_raytracer_amd import binding

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

    def fail(msg):
        """A frame step other than the one asked for is not a measurement of it: every rank leaves, non-zero."""
        sys.stderr.write("bench.py: %s\n" % msg)
        sys.stderr.flush()
        if world > 1:
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        sys.exit(3)

    scene = skr.parse_scene(SCENE)
    r = skr.Renderer(scene, local_rank)
    opt = skr.Options(W, H, **KW)
    stream = torch.cuda.current_stream(dev)
    k_max = binding.shard_tiles_per_rank(H, TILE_ROWS, world)

    # the frame step: libskr's own (RCCL inside the library); --torch-gather (and the one-GPU rehearsal, which has no RCCL peers) asks for round 1's
    comm = None
    if not args.torch_gather and not rehearsal:
        err = None
        try:
            uid = None
            if world > 1:
                t = torch.zeros(binding.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
                if rank == 0:
                    t.copy_(torch.from_numpy(np.frombuffer(binding.comm_unique_id(), dtype=np.uint8).copy()))
                dist.broadcast(t, 0)
                uid = bytes(t.cpu().numpy().tobytes())
            comm = binding.Comm(r, rank, world, uid)
        except skr.SkrError as e:
            err = str(e)[:200]
        ok = 1 if comm is not None else 0
        if world > 1:
            t_ok = torch.tensor([ok], device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok = int(t_ok.item())
        if not ok:
            fail("libskr's RCCL frame step could not be set up on every rank (%s); run with --torch-gather to measure the torch.distributed step instead" % (err or "another rank failed"))
    sharder = None if comm is not None else FrameSharder(W, H, TILE_ROWS, rank, world, dev)
    if comm is not None and world > 1:
        T_all = (H + TILE_ROWS - 1) // TILE_ROWS
        plan = r.shard_plan(opt, TILE_ROWS, world)  # (what the frame steps use: skr_shard_plan)
        blind = (np.arange(T_all) % world) * k_max + np.arange(T_all) // world
        tile_map_name = "tile t -> rank t mod G (the counted tile costs leave it within 10 % of balance)" if np.array_equal(plan, blind) else "longest-processing-time-first over the counted tile costs"
        if os.environ.get("SKR_SHARD"):
            tile_map_name += " [SKR_SHARD=%s]" % os.environ["SKR_SHARD"]
    else:
        tile_map_name = "tile t -> rank t mod G"

    # The pipelined step is the timed one at every N: frame f's collective and de-interleave on the communicator's stream, and the frames of the run
    # alternating between two renderers on two streams (two frames in flight: the next frame's short kernels fill this frame's tail).  Throughput of a run
    # of frames, which is what `value` is; the serial step (one frame at a time: its latency) is printed beside it from a side pass (--sync-frames swaps them).
    pipelined = comm is not None and not args.sync_frames

    frames_enqueued = [0]  # every frame this process asks the device for, timed or not (tools/pmc_traffic.py divides a profiled run's counters by it)

    def step():
        frames_enqueued[0] += 1
        if pipelined:  # frame f's all-gather + de-interleave on the communicator's stream while this stream renders frame f + 1
            comm.render_frame_async(opt, TILE_ROWS, stream.cuda_stream, want_previous=False)
        elif comm is not None:
            comm.render_frame(opt, TILE_ROWS, stream.cuda_stream)
        else:
            sharder.step(lambda buf: r.render_tiles_into(opt, TILE_ROWS, rank, world, buf.data_ptr(), None, stream.cuda_stream))

    if pipelined:
        # one pipelined frame before anything is timed: it either works on every rank or the run is not the one that was asked for
        ok_async, err = 1, None
        try:
            frames_enqueued[0] += 1
            comm.render_frame_async(opt, TILE_ROWS, stream.cuda_stream, want_previous=False)
            comm.flush(stream.cuda_stream)
            torch.cuda.synchronize(dev)
        except skr.SkrError as e:
            ok_async, err = 0, str(e)[:200]
        if world > 1:
            t_ok = torch.tensor([ok_async], device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok_async = int(t_ok.item())
        if not ok_async:
            fail("the pipelined frame step (skr_comm_render_frame_async) failed (%s); run with --sync-frames to measure the serial step instead" % (err or "on another rank"))

    def sync():
        if pipelined:
            comm.flush(stream.cuda_stream)  # the last frame's collective is part of the timed region
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    r.work(reset=True)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    cnt = r.work(reset=True)

    # the other frame step of the library beside the timed one, over a short pass of its own (outside the timed region): the serial step
    # (collective on the render stream) when the pipelined one was timed and the other way round — so that a run on G GPUs shows what
    # the pipelining buys.  Same barrier + synchronize bracket, MAX over ranks below.
    other_ms, other_name = None, None
    if comm is not None and not args.no_side_pass:
        n_other = min(args.steps, 100)
        if pipelined:
            other_name = "skr_comm_render_frame (serial: collective on the render stream)"
            sync()
            t1 = time.perf_counter()
            for _ in range(n_other):
                frames_enqueued[0] += 1
                comm.render_frame(opt, TILE_ROWS, stream.cuda_stream)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            other_ms = (time.perf_counter() - t1) / n_other * 1e3
        else:
            other_name = "skr_comm_render_frame_async (pipelined: collective of frame f behind the render of frame f + 1)"
            try:
                frames_enqueued[0] += 1
                comm.render_frame_async(opt, TILE_ROWS, stream.cuda_stream, want_previous=False)
                comm.flush(stream.cuda_stream)
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(n_other):
                    frames_enqueued[0] += 1
                    comm.render_frame_async(opt, TILE_ROWS, stream.cuda_stream, want_previous=False)
                comm.flush(stream.cuda_stream)
                if world > 1:
                    dist.barrier()
                torch.cuda.synchronize(dev)
                other_ms = (time.perf_counter() - t1) / n_other * 1e3
            except skr.SkrError as e:
                fail("the pipelined frame step failed in the side pass (%s)" % str(e)[:200])
        r.work(reset=True)

    variant = r.kernel_variant()
    queued = r.last_parent_count()
    level1 = r.last_level1_count()
    # the dominant kernel alone: HIP events on its stream around every launch of a short extra pass (not in the timed region),
    # and the work counters copied in front of and behind it (skr_renderer_kernel_work)
    r.kernel_timing(True)
    r.kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_probe = min(args.steps, 50)
    probe = torch.zeros((k_max * TILE_ROWS, W, 3), dtype=torch.uint8, device=dev)
    frames_enqueued[0] += 1
    r.render_tiles_into(opt, TILE_ROWS, rank, world, probe.data_ptr(), None, stream.cuda_stream)
    torch.cuda.synchronize(dev)
    r.kernel_ms()
    e0.record(stream)
    for _ in range(n_probe):
        frames_enqueued[0] += 1
        r.render_tiles_into(opt, TILE_ROWS, rank, world, probe.data_ptr(), None, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    kernel_ms, _ = r.kernel_ms()
    kwork = r.kernel_work()
    r.kernel_timing(False)
    pipeline_ms = e0.elapsed_time(e1) / n_probe  # everything this rank enqueues per frame before the collective
    r.work(reset=True)
    # what the triangle walks execute, from ONE frame rendered with the walks counting (never a timed frame: counting costs the dragon
    # walk ~19 %); scaled to the timed frames below (every frame of a run is the same frame)
    tri = {"cull_tests": 0, "triangle_tests": 0, "reference_triangle_tests": 0}
    if scene.info.n_triangles > 0:
        r.count_triangle_work(True)
        r.triangle_work(reset=True)
        frames_enqueued[0] += 1
        r.render_tiles_into(opt, TILE_ROWS, rank, world, probe.data_ptr(), None, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        tri = r.triangle_work(reset=True)
        r.count_triangle_work(False)
        r.work(reset=True)
        tri = {k: v * args.steps for k, v in tri.items()}

    stats = torch.tensor([dt, float(cnt["radiance_rays"]), float(cnt["shadow_rays"]), float(cnt["sphere_hits"]), float(cnt["sphere_tests"]), kernel_ms, pipeline_ms,
                          float(tri["cull_tests"]), float(tri["triangle_tests"]), float(tri["reference_triangle_tests"]), float(other_ms or 0.0)],
                         dtype=torch.float64, device=dev if not rehearsal else "cpu")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, kernel_ms, pipeline_ms = float(mx[0]), float(mx[5]), float(mx[6])
        if other_ms is not None:
            other_ms = float(mx[10])
        tot = sm
    else:
        tot = stats
    rays, shadow, hits, tests = float(tot[1]), float(tot[2]), float(tot[3]), float(tot[4])
    cull_tests, tri_tests, tri_tests_ref = float(tot[7]), float(tot[8]), float(tot[9])

    if rank == 0:
        n = args.steps
        rays_per_frame = rays / n
        ms_per_step = dt / n * 1e3
        info = scene.info
        nsamp = max(1, KW.get("jsample", 0) ** 2)
        # ---- HBM, as SURVEY.md 8(d) defines it: the algorithm's compulsory bytes per frame — the u8 framebuffer out and one read of the scene —
        # over the frame time.  Structurally ~4e-4 of peak: this path is not HBM-bound; what the counters saw is `traffic`.
        scene_bytes = info.n_spheres * 64 + info.n_point_lights * 32 + info.n_triangles * 48
        alg_bytes = W * H * 3 + scene_bytes
        achieved_gbs = alg_bytes / (ms_per_step * 1e-3) / 1e9
        # the node pipeline's own tables (one band = the frame): a level-0 node is a 32-byte geometry row (read by the trace kernel and, gathered,
        # by the leaf kernel) and a 32-byte shading row (read by finalize); a level-1 record is 16 bytes written and read, its result 12 bytes
        # written and read; every trace wave (64 sibling pairs) leaves a 48-byte header that finalize reads
        N = KW.get("gillum", 0)
        pipeline_bytes = (queued * (64 + 3 * 32) + level1 * (2 * 16 + 2 * 12) + (queued * ((N + 1) // 2) + 63) // 64 * 48 * 2) if variant == "node_levels_v5" and nsamp == 1 else None
        tj, why = measured_traffic(args.config, variant) if world == 1 else (None, "measured at N = 1 only")
        traffic = tj["traffic_bytes_per_frame"] if tj else None
        # ---- FP32 VALU.  Executed work: every ray-sphere test the reference runs (a shadow walk stops at its first occluder: counted by the
        # kernels, asserted equal to the oracle's count), every shaded hit, and for meshes the tests the culled walk EXECUTED (counted by the
        # walk: lanes that needed the test) — not the 10 002 per ray of the reference's loop, which the exact culling provably never needs.
        frame_flop = (tests * FLOP_SPHERE_TEST + hits * FLOP_SHADED_HIT + tri_tests * FLOP_TRI_TEST + cull_tests * FLOP_CULL_TEST) / n
        frame_flop_ref = (tests * FLOP_SPHERE_TEST + hits * FLOP_SHADED_HIT + tri_tests_ref * FLOP_TRI_TEST_REF) / n
        frame_tflops = frame_flop / (ms_per_step * 1e-3) / 1e12 / world
        # the dominant kernel's own share (skr_renderer_kernel_work: counters copied around it; mesh walks are not split per kernel: whole-frame
        # figure where the wave kernel IS the frame).  The wave kernel traces all the AA samples of a frame in one launch.
        whole_frame_kernel = variant == "direct_v3"
        if whole_frame_kernel:
            kernel_flop = frame_flop
        else:
            kernel_flop = kwork["sphere_tests"] * FLOP_SPHERE_TEST + kwork["sphere_hits"] * FLOP_SHADED_HIT
        kernel_tflops = (kernel_flop / (kernel_ms * 1e-3) / 1e12) if kernel_ms > 0 else 0.0
        out = {
            "metric": "Mrays/sec + frame ms, 1920x1080 gillum=16 spheres2.scn" if args.config == 3 else "Mrays/sec + frame ms, BASELINE.json configs[%d]" % (args.config - 1),
            "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": n, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "reference scene file scenes/%s%s" % (cfg["scene"], " (spherical_fog line skipped: UB in the reference)" if cfg["scene"] == "spheres2.scn" else ""),
            "config": {"workload": cfg["workload"], "baseline_config": args.config,
                       "rays_per_frame": rays_per_frame, "nominal_rays": skr.radiance_ray_count(opt),
                       "nominal_mrays_per_s": skr.radiance_ray_count(opt) * n / dt / 1e6,
                       "shadow_rays_per_frame": shadow / n, "sphere_tests_per_frame": tests / n, "shaded_hits_per_frame": hits / n,
                       "triangle_tests_executed_per_frame": tri_tests / n, "culling_sphere_tests_executed_per_frame": cull_tests / n,
                       "triangle_tests_of_the_reference_loop_per_frame": tri_tests_ref / n,
                       "partition": "%d-row tiles over %d rank(s)" % (TILE_ROWS, world),
                       "frame_step": ("REHEARSAL on one GPU over gloo - not a measurement" if rehearsal else
                                      (("libskr skr_comm_render_frame_async: tiles -> ncclAllGather (RCCL, in the library) -> de-interleave kernel on rank 0, the collective of frame f on its own stream behind the render of frame f + 1, two frames in flight (the frames of the run alternate between the renderer and a clone of it on a second stream); the last frame's collective inside the timed region" if pipelined else "libskr skr_comm_render_frame: tiles -> ncclAllGather (RCCL, in the library) -> de-interleave kernel on rank 0") if comm is not None else
                                       "torch.distributed all_gather_into_tensor of the u8 tile buffers, rank 0 de-interleaves (torch) [--torch-gather]")) if world > 1
                                     else (("libskr skr_comm_render_frame_async (1 GPU: tiles, then the de-interleave kernel on the communicator's stream; no collective; two frames in flight: the frames of the run alternate between the renderer and a clone of it on a second stream — throughput of the run, the serial step's per-frame latency is config.frame_steps_ms.other)" if pipelined else "libskr skr_comm_render_frame (1 GPU: tiles + de-interleave kernel, no collective)") if comm is not None else "skr_render_tiles + torch de-interleave [--torch-gather]"),
                       "frame_steps_ms": ({"timed": ms_per_step, "timed_step": "pipelined" if pipelined else "serial", "other": other_ms, "other_step": other_name,
                                           "render_only": pipeline_ms, "note": "`other` and `render_only` come from short passes outside the timed region (MAX over ranks)"}
                                          if other_ms is not None else None),
                       "tile_map": tile_map_name, "frames_enqueued": frames_enqueued[0],
                       "kernel": variant, "seed": KW["seed"]},
            "roofline": {"bound": "fp32_valu", "achieved": kernel_tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": kernel_tflops / VALU_PEAK_TFLOPS,
                         "traffic": traffic,
                         "kernel": DOMINANT.get(variant, variant), "kernel_ms": kernel_ms, "kernel_gflop_per_launch": kernel_flop / 1e9,
                         "kernel_work_per_launch": (dict(kwork) if not whole_frame_kernel else
                                                    {"sphere_tests": tests / n, "sphere_hits": hits / n, "triangle_tests": tri_tests / n, "cull_tests": cull_tests / n}),
                         "frame": {"achieved": frame_tflops, "frac": frame_tflops / VALU_PEAK_TFLOPS, "gflop_per_frame": frame_flop / 1e9,
                                   "reference_loop_gflop_per_frame": frame_flop_ref / 1e9, "ms": ms_per_step},
                         "note": "the bound that binds: FP32 VALU issue (branchy scalar FP32; no MFMA; HBM at 4e-4 of peak: roofline_hbm).  achieved = the dominant kernel's own counted work "
                                 "(%d flop per ray-sphere test the reference runs — shadow walks stop at their first occluder: counted, not assumed — %d per shaded hit, %d per ray-triangle test and %d per "
                                 "culling-sphere test the walk EXECUTED; SURVEY.md 8d) over its mean launch duration (HIP events on its stream, a separate untimed pass).  `frame` is the same over the whole "
                                 "frame time; `reference_loop_gflop_per_frame` prices the triangle loop as the reference runs it (every triangle for every ray, %d flop each), which the exact culling never "
                                 "executes.  The spec forbids FMA contraction, so half the FMA peak is the ceiling of this instruction stream."
                                 % (FLOP_SPHERE_TEST, FLOP_SHADED_HIT, FLOP_TRI_TEST, FLOP_CULL_TEST, FLOP_TRI_TEST_REF)},
            "roofline_hbm": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                             "traffic": traffic, "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                             "traffic_per_kernel": tj.get("per_kernel") if tj else None,
                             "traffic_source": ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over every kernel of the frame, FETCH doubled per the gfx950 note; keyed by the "
                                                "git blob hashes of the kernel sources" % os.path.relpath(traffic_json(args.config), ROOT)) if traffic else why,
                             "algorithmic_bytes_per_frame": alg_bytes, "pipeline_bytes": pipeline_bytes, "render_ms": pipeline_ms,
                             "level0_nodes": queued, "level1_records": level1,
                             "note": "north_star's requested figure — SURVEY.md 8(d): algorithmic bytes = W*H*3 + scene per frame, over ms_per_step; this path is not HBM-bound"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["config"]["gpu_over_cpu_note"] = "against %d of the host's %d hardware threads (the box's cgroup quota)" % (out["cpu_baseline"]["cores"], out["cpu_baseline"]["host_cores"])
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — the reference's headline benchmark on MI355X.

Metric (BASELINE.json): Mrays/sec + frame ms, scenes/spheres2.scn 1920x1080 --gillum 16
--shadow (depth 3), on 1/2/4/8 MI355X.  A "step" is one whole frame: every rank renders its
interleaved row tiles (C ABI, include/skr.h), the u8 tiles are gathered with ONE RCCL
all-gather over xGMI and rank 0 de-interleaves them on the device — the whole step runs
inside libskr (skr_comm_render_frame); torch.distributed only carries the 128-byte RCCL id
and the barriers.  Inputs (the SoA scene) are resident in HBM before the timed region.
`value` = radiance rays actually traced per second, whole job (rays = shade() calls with
depth > 0, counted by the kernels themselves; deterministic and partition-independent).
SURVEY.md §8d's closed form W*H*S*sum N^k is the full-tree upper bound (every ray hitting a
sphere) and is reported beside it as `nominal_rays`.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
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
FLOP_PER_SPHERE_TEST, FLOP_PER_SHADED_HIT = 34, 150
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")
# the files whose contents decide what the kernels move: the measured traffic is only reported for the exact sources it was measured on
KERNEL_SOURCES = ["skele_raytracer_amd/csrc/render_nodes.hip", "skele_raytracer_amd/csrc/render_wave.hip", "skele_raytracer_amd/csrc/wave_common.h",
                  "skele_raytracer_amd/csrc/shade_common.h", "skele_raytracer_amd/csrc/device_math.h", "skele_raytracer_amd/csrc/render_params.h"]
DOMINANT = {"node_levels_v5": "skr_leaf_kernel2<false, false>", "node_levels_v5_flat": "skr_trace_kernel<false> (last level) + skr_shade_leaf_kernel<false>", "level_queues_v4": "skr_leaf_kernel<false>", "parent_queue_v3": "skr_gi_kernel<3, 3, false>",
            "wave_streaming_v2": "skr_wave_kernel<3, 3>"}


def git_blob_hash(path):
    """`git hash-object` of a file: sha1 of 'blob <len>\\0' + contents."""
    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes():
    return {p: git_blob_hash(os.path.join(ROOT, p)) for p in KERNEL_SOURCES}


def measured_traffic(variant):
    """HBM bytes per frame over ALL kernels of the frame from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this command (tools/pmc_traffic.sh; FETCH doubled per MI355X_MICROARCH.md) — or None when the kernel sources have changed since."""
    try:
        with open(TRAFFIC_JSON) as f:
            tj = json.load(f)
    except (OSError, ValueError):
        return None, "no %s" % os.path.relpath(TRAFFIC_JSON, ROOT)
    if tj.get("variant") != variant:
        return None, "measured for kernel variant %s" % tj.get("variant")
    if tj.get("sources") != source_hashes():
        return None, "stale: the kernel sources have changed since the PMC passes (git blob hashes differ)"
    return tj, None


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
    host = os.cpu_count() or cores
    reference = reference_sample(orc)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "host_cores": host, "kind": "port", "cpu_model": model, "reference_1core": reference,
            "ms_per_frame": dt / frames * 1e3,
            "sample": "oracle/liboracle.so (C restatement of the reference path, gcc -O2, OpenMP) on %d of this host's %d cores — the box's cgroup CPU quota; "
                      "scaled to all %d cores the baseline would be ~%.0fx higher — %d whole frames of the same workload: %d radiance rays in %.2f s"
                      % (cores, host, host, host / max(1, cores), frames, rays, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)   # >= 1 s of frames at one GPU: the clock the chip settles at, not its first milliseconds
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--async-frames", action="store_true", help="the pipelined frame step at N = 1 too (it is the default for N > 1)")
    ap.add_argument("--sync-frames", action="store_true", help="libskr's frame step with the collective on the render stream (skr_comm_render_frame) instead of pipelined behind the next frame")
    ap.add_argument("--torch-gather", action="store_true", help="the round-1 frame step (torch.distributed all_gather + torch de-interleave) instead of libskr's")
    args = ap.parse_args()

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

    scene = skr.parse_scene(SCENE)
    r = skr.Renderer(scene, local_rank)
    opt = skr.Options(W, H, **KW)
    stream = torch.cuda.current_stream(dev)
    k_max = binding.shard_tiles_per_rank(H, TILE_ROWS, world)

    # the frame step: libskr's own (RCCL inside the library) unless it cannot be set up — then, and on request, round 1's
    comm, native_note = None, None
    if not args.torch_gather and not rehearsal:
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
            native_note = "libskr's RCCL step unavailable (%s): torch.distributed all_gather used" % str(e)[:160]
    if world > 1:  # every rank takes the same path
        ok = torch.tensor([1 if comm is not None else 0], device=dev) if not rehearsal else torch.tensor([0])
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            comm = None
    sharder = None if comm is not None else FrameSharder(W, H, TILE_ROWS, rank, world, dev)

    # (at N = 1 there is no collective to hide and the two extra stream waits cost 1 %: measured 1.796 against 1.779 ms)
    pipelined = comm is not None and (world > 1 or args.async_frames) and not args.sync_frames

    def step():
        if pipelined:  # frame f's all-gather + de-interleave on the communicator's stream while this stream renders frame f + 1
            comm.render_frame_async(opt, TILE_ROWS, stream.cuda_stream, want_previous=False)
        elif comm is not None:
            comm.render_frame(opt, TILE_ROWS, stream.cuda_stream)
        else:
            sharder.step(lambda buf: r.render_tiles_into(opt, TILE_ROWS, rank, world, buf.data_ptr(), None, stream.cuda_stream))

    if pipelined:
        # one pipelined frame before anything is timed: a rank on which it cannot be set up takes every rank back to the serial step
        ok_async = 1
        try:
            comm.render_frame_async(opt, TILE_ROWS, stream.cuda_stream, want_previous=False)
            comm.flush(stream.cuda_stream)
            torch.cuda.synchronize(dev)
        except skr.SkrError as e:
            ok_async, native_note = 0, "pipelined frame step unavailable (%s): serial skr_comm_render_frame used" % str(e)[:160]
        if world > 1:
            t_ok = torch.tensor([ok_async], device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            ok_async = int(t_ok.item())
        pipelined = bool(ok_async)

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

    variant = r.kernel_variant()
    queued = r.last_parent_count()
    level1 = r.last_level1_count()
    cnt = r.work(reset=True)
    # the dominant kernel alone: HIP events on its stream around every launch of a short extra pass (not in the timed region)
    r.kernel_timing(True)
    r.kernel_ms()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_probe = min(args.steps, 50)
    r.render_tiles_into(opt, TILE_ROWS, rank, world, r_probe_buf(torch, dev, k_max).data_ptr(), None, stream.cuda_stream)  # (allocates the probe buffer)
    torch.cuda.synchronize(dev)
    r.kernel_ms()
    e0.record(stream)
    for _ in range(n_probe):
        r.render_tiles_into(opt, TILE_ROWS, rank, world, r_probe_buf(torch, dev, k_max).data_ptr(), None, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    kernel_ms, _ = r.kernel_ms()
    r.kernel_timing(False)
    pipeline_ms = e0.elapsed_time(e1) / n_probe  # everything this rank enqueues per frame before the collective
    r.work(reset=True)

    stats = torch.tensor([dt, float(cnt["radiance_rays"]), float(cnt["shadow_rays"]), float(cnt["sphere_hits"]), float(cnt["sphere_tests"]), kernel_ms, pipeline_ms],
                         dtype=torch.float64, device=dev if not rehearsal else "cpu")
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt, kernel_ms, pipeline_ms = float(mx[0]), float(mx[5]), float(mx[6])
        rays, shadow, hits, tests = float(sm[1]), float(sm[2]), float(sm[3]), float(sm[4])
    else:
        rays, shadow, hits, tests = float(stats[1]), float(stats[2]), float(stats[3]), float(stats[4])

    if rank == 0:
        n = args.steps
        rays_per_frame = rays / n
        ms_per_step = dt / n * 1e3
        info = scene.info
        # ---- HBM roofline, as SURVEY.md §8(d) defines it: the algorithm's compulsory bytes per frame — the u8 framebuffer out and one read of
        # the scene — over the frame time.  (0.011 B per nominal ray x the rays of a frame.)  Structurally ~4e-4 of peak: this path is not
        # HBM-bound; what the pipeline itself moves between its kernels is `pipeline_bytes`, what the counters saw is `traffic`.
        scene_bytes = info.n_spheres * 64 + info.n_point_lights * 32 + info.n_triangles * 48
        alg_bytes = W * H * 3 + scene_bytes
        achieved_gbs = alg_bytes / (ms_per_step * 1e-3) / 1e9
        # the node pipeline's own tables (one band = the frame): a level-0 node is a 32-byte geometry row (read by the trace kernel and, gathered,
        # by the leaf kernel) and a 32-byte shading row (read by finalize); a level-1 record is 16 bytes written and read, its result 12 bytes
        # written and read; every trace wave (64 sibling pairs) leaves a 48-byte header that finalize reads
        N = KW["gillum"]
        pipeline_bytes = (queued * (64 + 3 * 32) + level1 * (2 * 16 + 2 * 12) + (queued * ((N + 1) // 2) + 63) // 64 * 48 * 2) if variant == "node_levels_v5" else None
        tj, why = measured_traffic(variant) if world == 1 else (None, "measured at N = 1 only")
        traffic = tj["traffic_bytes_per_frame"] if tj else None
        # ---- FP32-VALU roofline: the flops the reference's algorithm needs for this frame — every ray-sphere test it would run (early-outs of the
        # shadow walks counted by the kernels, asserted equal to the oracle's count) and every shaded hit — over the frame time, per GPU
        alg_flop = tests / n * FLOP_PER_SPHERE_TEST + hits / n * FLOP_PER_SHADED_HIT
        valu_tflops = alg_flop / (ms_per_step * 1e-3) / 1e12 / world
        out = {
            "metric": "Mrays/sec + frame ms, 1920x1080 gillum=16 spheres2.scn",
            "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": n, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "reference scene file scenes/spheres2.scn (spherical_fog line skipped: UB in the reference)",
            "config": {"workload": "scenes/spheres2.scn 1920x1080 --gillum 16 --shadow --depth 3 (BASELINE.json configs[2])",
                       "rays_per_frame": rays_per_frame, "nominal_rays": skr.radiance_ray_count(opt),
                       "nominal_mrays_per_s": skr.radiance_ray_count(opt) * n / dt / 1e6,
                       "shadow_rays_per_frame": shadow / n, "sphere_tests_per_frame": tests / n, "shaded_hits_per_frame": hits / n,
                       "partition": "interleaved %d-row tiles, rank = tile %% %d" % (TILE_ROWS, world),
                       "frame_step": ("REHEARSAL on one GPU over gloo - not a measurement" if rehearsal else
                                      (("libskr skr_comm_render_frame_async: tiles -> ncclAllGather (RCCL, in the library) -> de-interleave kernel on rank 0, the collective of frame f on its own stream behind the render of frame f + 1; the last frame's collective inside the timed region" if pipelined else "libskr skr_comm_render_frame: tiles -> ncclAllGather (RCCL, in the library) -> de-interleave kernel on rank 0") if comm is not None else
                                       "torch.distributed all_gather_into_tensor of the u8 tile buffers, rank 0 de-interleaves (torch)")) if world > 1
                                     else (("libskr skr_comm_render_frame_async (1 GPU: tiles, then the de-interleave kernel on the communicator's stream; no collective)" if pipelined else "libskr skr_comm_render_frame (1 GPU: tiles + de-interleave kernel, no collective)") if comm is not None else "skr_render_tiles"),
                       "frame_step_note": native_note, "kernel": variant, "seed": KW["seed"]},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "traffic_per_kernel": tj.get("per_kernel") if tj else None,
                         "traffic_source": ("profiles/r02_hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over every kernel of the frame, FETCH doubled "
                                            "per the gfx950 note; keyed by the git blob hashes of the kernel sources") if traffic else why,
                         "algorithmic_bytes_per_frame": alg_bytes, "pipeline_bytes": pipeline_bytes,
                         "kernel": DOMINANT.get(variant, "skr_render_kernel<3>"), "kernel_ms": kernel_ms, "render_ms": pipeline_ms,
                         "level0_nodes": queued, "level1_records": level1,
                         "note": "SURVEY.md 8(d): algorithmic bytes = W*H*3 + scene per frame, over ms_per_step; this path is bound by FP32 VALU issue, not by HBM (roofline_valu); "
                                 "kernel_ms = the dominant kernel's mean launch duration (HIP events on its stream, a separate untimed pass)"},
            "roofline_valu": {"bound": "fp32_valu", "achieved": valu_tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": valu_tflops / VALU_PEAK_TFLOPS,
                              "algorithmic_gflop_per_frame": alg_flop / 1e9,
                              "note": "frame flops over frame time per GPU: %d flop per ray-sphere test the reference runs (shadow walks stop at their first occluder: counted, "
                                      "not assumed), %d per shaded hit (SURVEY.md 8d); the spec forbids FMA contraction, so 1/2 of the FMA peak is the ceiling of this instruction stream"
                                      % (FLOP_PER_SPHERE_TEST, FLOP_PER_SHADED_HIT)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["config"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_probe = {}


def r_probe_buf(torch, dev, k_max):
    if "b" not in _probe:
        _probe["b"] = torch.zeros((k_max * TILE_ROWS, W, 3), dtype=torch.uint8, device=dev)
    return _probe["b"]


if __name__ == "__main__":
    main()

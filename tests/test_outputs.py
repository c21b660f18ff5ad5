"""Other outputs and progressive accumulation: `--format png|pfm`, `--progressive K [--progressive-every M]`.

The reference writes P6 and nothing else (main.cpp:199-211) and shows the frame in an SDL window while it forms
(main.cpp:183-197).  PNG carries the PPM's bytes, PFM the unquantised floats, and the progressive mean is defined in
include/skr.h (skr_options.progressive_passes): K frames under the seeds s .. s+K-1, summed in binary32 in pass order,
divided by (float) K once.  Every single pass is a frame the other tests pin; what is checked here is the container
formats (decoded independently) and that the mean is exactly that sum, on every entry point.
"""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
from conftest import ROOT, read_ppm_bytes, scene_path


def read_png(path):
    """Minimal independent PNG reader: 8-bit RGB, filter 0 only, checks every CRC."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", data[at + 8 + n:at + 12 + n])
        assert zlib.crc32(kind + body) == crc, kind
        chunks.append((kind, body))
        at += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)  # (zlib checks the adler32)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 3)


def read_pfm(path):
    data = open(path, "rb").read()
    magic, dims, scale, body = data.split(b"\n", 3)
    assert magic == b"PF" and float(scale) < 0  # colour, little-endian
    w, h = map(int, dims.split())
    return np.frombuffer(body, "<f4", w * h * 3).reshape(h, w, 3)[::-1]  # bottom row first in the file


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (640, 360), (1920, 1080)])  # 1080p: 6.2 MB of rows = 95 stored deflate blocks
def test_png_holds_the_bytes_of_the_ppm(tmp_path, w, h):
    rgb = np.random.default_rng(w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    skr.write_png(str(tmp_path / "a.png"), rgb)
    skr.write_ppm(str(tmp_path / "a.ppm"), rgb)
    assert (read_png(str(tmp_path / "a.png")) == rgb).all()
    assert (read_ppm_bytes(open(str(tmp_path / "a.ppm"), "rb").read()) == rgb).all()


def test_pfm_holds_the_float_frame_bit_for_bit(tmp_path):
    f = np.random.default_rng(3).standard_normal((33, 47, 3)).astype(np.float32)
    f[0, 0] = (np.nan, np.inf, -0.0)
    f[5, 6] = (1e-42, 3.4e38, -1.0)  # a denormal, near-max, negative
    skr.write_pfm(str(tmp_path / "a.pfm"), f)
    assert (read_pfm(str(tmp_path / "a.pfm")).view(np.uint32) == f.view(np.uint32)).all()


def test_writers_fail_loudly(tmp_path):
    rgb = np.zeros((2, 2, 3), np.uint8)
    with pytest.raises(skr.SkrError):
        skr.write_png(str(tmp_path / "no_such_dir" / "a.png"), rgb)
    with pytest.raises(skr.SkrError):
        skr.write_pfm(str(tmp_path / "no_such_dir" / "a.pfm"), rgb.astype(np.float32))


def test_progressive_passes_scale_the_ray_count():
    one = skr.Options(64, 36, gillum=4, jsample=2)
    five = skr.Options(64, 36, gillum=4, jsample=2, progressive=5)
    assert skr.radiance_ray_count(five) == 5 * skr.radiance_ray_count(one)
    assert skr.Options(8, 8).c.progressive_passes == 1 and skr.Options(8, 8, progressive=0).c.progressive_passes == 1


def test_oracle_quantiser_restatement_matches_the_c_one(oracle):
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(100000) * rng.choice([1e-3, 1, 10, 1e6, 1e12], 100000)).astype(np.float32)
    x[:6] = (np.nan, np.inf, -np.inf, 1.0, -0.0, 0.99999994)
    want = np.array([oracle.lib().sko_quantise(float(v)) for v in x], np.uint8)
    assert (oracle.quantise(x) == want).all()


# ------------------------------------------------------------------------------------------------ on the device ----

KW = dict(gillum=4, shadow=True, seed=20)
W, H = 96, 54


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def spheres2():
    sc = skr.parse_scene(scene_path("spheres2.scn"))
    return skr.Renderer(sc)


@pytest.mark.gpu
@pytest.mark.parametrize("name,scn,passes,kw", [
    ("gi4_x3", "spheres2.scn", 3, KW),
    ("gi2_d4_x2", "spheres2.scn", 2, dict(gillum=2, depth=4, shadow=True, seed=7)),
    ("js2_x4", "spheres2.scn", 4, dict(jsample=2, shadow=True, seed=2 ** 64 - 2)),   # the seeds wrap: 2^64-2, 2^64-1, 0, 1
    ("tris_x2", "test.scn", 2, dict(gillum=3, seed=4)),
    ("nogi_x3", "spheres1.scn", 3, dict(shadow=True)),                                  # no random number anywhere: the mean of three equal frames
], ids=["gi4_x3", "gi2_d4_x2", "js2_x4", "tris_x2", "nogi_x3"])
def test_progressive_mean_matches_the_oracle_bit_for_bit(gpu, oracle, name, scn, passes, kw):
    r = skr.Renderer(skr.parse_scene(scene_path(scn)))
    rgb, rgbf = r.render(skr.Options(W, H, progressive=passes, **kw), want_float=True)
    gpu.cuda.synchronize()
    o_rgb, o_f, st, _ = oracle.render_progressive(scene_path(scn), W, H, passes, **kw)
    assert (rgbf.cpu().numpy().view(np.uint32) == o_f.view(np.uint32)).all()
    assert (rgb.cpu().numpy() == o_rgb).all()
    cnt = r.counters()
    assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])


@pytest.mark.gpu
def test_one_pass_is_the_plain_frame_and_the_partition_does_not_matter(gpu, spheres2):
    plain, plain_f = spheres2.render(skr.Options(W, H, **KW), want_float=True)
    one, one_f = spheres2.render(skr.Options(W, H, progressive=1, **KW), want_float=True)
    assert gpu.equal(plain, one) and gpu.equal(plain_f.view(gpu.int32), one_f.view(gpu.int32))
    opt = skr.Options(W, H, progressive=3, **KW)
    whole, whole_f = spheres2.render(opt, want_float=True)
    assert not gpu.equal(whole, plain)
    # interleaved row tiles over 4 "ranks", last tile partial (54 rows in tiles of 8): rows beyond the image stay untouched
    tile_rows, G = 8, 4
    frame = gpu.zeros_like(whole)
    for rank in range(G):
        part, _ = spheres2.render(opt, tile_rows=tile_rows, first_tile=rank, tile_stride=G)
        n = spheres2.tile_count(opt, tile_rows, rank, G)
        for k in range(n):
            y0 = (rank + k * G) * tile_rows
            rows = min(tile_rows, H - y0)
            frame[y0:y0 + rows] = part[k * tile_rows:k * tile_rows + rows]
            assert (part[k * tile_rows + rows:(k + 1) * tile_rows] == 0).all()  # (render() hands out zeroed buffers)
    assert gpu.equal(frame, whole)
    rows, rows_f = spheres2.render_rows(opt, 10, 31, want_float=True)
    assert gpu.equal(rows, whole[10:31]) and gpu.equal(rows_f.view(gpu.int32), whole_f[10:31].view(gpu.int32))


@pytest.mark.gpu
def test_the_mean_can_be_watched_while_it_forms(gpu, oracle, spheres2):
    passes, every = 5, 2
    opt = skr.Options(W, H, progressive=passes, **KW)
    _, _, _, means = oracle.render_progressive(scene_path("spheres2.scn"), W, H, passes, **KW)
    seen = []

    def progress(done, total, rgb, rgbf):  # (an exception raised inside a ctypes callback is swallowed: record, assert afterwards)
        seen.append((done, total, bool((rgbf.view(np.uint32) == means[done - 1].view(np.uint32)).all()), bool((rgb == oracle.quantise(means[done - 1])).all())))
        return False

    rgb, rgbf, ms = spheres2.render_progressive_host(opt, every, want_float=True, progress=progress)
    assert seen == [(2, passes, True, True), (4, passes, True, True), (5, passes, True, True)] and ms > 0
    one_shot, one_shot_f = spheres2.render(opt, want_float=True)
    assert (rgb == one_shot.cpu().numpy()).all() and (rgbf.view(np.uint32) == one_shot_f.cpu().numpy().view(np.uint32)).all()
    # a viewer that closes after the first look keeps the mean it saw
    seen.clear()
    rgb2, rgbf2, _ = spheres2.render_progressive_host(opt, every, want_float=True, progress=lambda d, t, a, b: seen.append(d) or True)
    assert seen == [2] and (rgbf2.view(np.uint32) == means[1].view(np.uint32)).all()
    # and the two steps on their own (include/skr.h skr_accumulate / skr_resolve_accumulated)
    acc = gpu.zeros((H, W, 3), dtype=gpu.float32, device="cuda")
    out = gpu.zeros((H, W, 3), dtype=gpu.uint8, device="cuda")
    for k in range(3):
        _, f = spheres2.render(skr.Options(W, H, **dict(KW, seed=KW["seed"] + k)), want_float=True)
        binding._check(binding.lib().skr_accumulate(acc.data_ptr(), f.data_ptr(), acc.numel(), int(k == 0), None), "skr_accumulate")
    binding._check(binding.lib().skr_resolve_accumulated(acc.data_ptr(), 3, W, H, out.data_ptr(), None, None), "skr_resolve_accumulated")
    gpu.cuda.synchronize()
    assert (out.cpu().numpy() == oracle.quantise(means[2])).all()


@pytest.mark.gpu
def test_both_command_lines_write_the_other_formats(gpu, oracle, tmp_path):
    scn = scene_path("spheres2.scn")
    passes = 3
    o_rgb, o_f, _, means = oracle.render_progressive(scn, W, H, passes, **KW)
    args = ["--path", scn, "--width", str(W), "--height", str(H), "--gillum", "4", "--shadow", "--seed", str(KW["seed"]), "--progressive", str(passes)]
    exe = os.path.join(ROOT, "bin", "raytracer")
    native = lambda extra, out: subprocess.run([exe, *args, *extra, "--output", out, "--quiet"], check=True, cwd=str(tmp_path), capture_output=True, text=True)
    py = lambda extra, out: subprocess.run([sys.executable, "-m", "skele_raytracer_amd.render_cli", *args, *extra, "--output", out], check=True, cwd=ROOT, capture_output=True, text=True)
    for tag, run in (("native", native), ("py", py)):
        p = lambda ext: str(tmp_path / ("%s.%s" % (tag, ext)))
        run([], p("ppm"))
        assert (read_ppm_bytes(open(p("ppm"), "rb").read()) == o_rgb).all(), tag
        run(["--format", "png"], p("png"))
        assert (read_png(p("png")) == o_rgb).all(), tag
        run(["--format", "pfm"], p("pfm"))
        assert (read_pfm(p("pfm")).view(np.uint32) == o_f.view(np.uint32)).all(), tag
        done = run(["--progressive-every", "1", "--format", "png"], p("watch.png"))
        assert (read_png(p("watch.png")) == o_rgb).all(), tag
        if tag == "py":
            assert [ln for ln in done.stdout.splitlines() if ln.startswith("pass ")] == ["pass 1 of 3", "pass 2 of 3", "pass 3 of 3"]
    # the sharded native path: K passes on every device before the one gather
    native(["--gpus", "1", "--tile-rows", "8"], str(tmp_path / "sharded.ppm"))
    assert (read_ppm_bytes(open(str(tmp_path / "sharded.ppm"), "rb").read()) == o_rgb).all()
    bad = subprocess.run([exe, *args, "--gpus", "1", "--format", "pfm", "--output", str(tmp_path / "x.pfm"), "--quiet"], cwd=str(tmp_path), capture_output=True, text=True)
    assert bad.returncode != 0 and "single-device" in bad.stderr

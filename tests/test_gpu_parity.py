"""GPU parity tests proper (pytest -m gpu): the HIP path, called through the C ABI, against
the CPU oracle on the same inputs.  Bar (BASELINE.json north_star): per-channel |delta| <= 1/255
under a fixed seed; what is actually asserted is stronger — the unquantised float image is
compared bit for bit (the kernel and the oracle's counter/shared-math mode implement the same
arithmetic spec) and the u8 image byte for byte."""
import numpy as np
import pytest

import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
from conftest import args_to_kwargs, manifest, read_golden_ppm, scene_path
from scenegen import write_random_mesh_scene

pytestmark = pytest.mark.gpu

TOL_U8 = 1  # north_star tolerance, 1/255 per channel


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


_renderers = {}


def renderer(scn, strict=False):
    key = (scn, bool(strict))
    if key not in _renderers:
        sc = skr.parse_scene(scene_path(scn), strict=strict)  # strict: --strict-scn, the directional lights kept
        _renderers[key] = (sc, skr.Renderer(sc))
    return _renderers[key][1]


def gpu_render(scn, w, h, want_float=True, strict=False, **kw):
    r = renderer(scn, strict)
    r.counters(reset=True)  # renderers are cached across tests: drop what earlier launches accumulated
    rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=want_float)
    import torch
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), (rgbf.cpu().numpy() if want_float else None), r.counters()


def compare(gpu_rgb, gpu_f, orc_rgb, orc_f, what):
    d = np.abs(gpu_rgb.astype(np.int32) - orc_rgb.astype(np.int32))
    assert d.max() <= TOL_U8, "%s: max |delta| = %d/255 at %s" % (what, d.max(), np.argwhere(d > TOL_U8)[:5])
    nb = int((gpu_f.view(np.uint32) != orc_f.view(np.uint32)).sum())
    assert nb == 0, "%s: %d float words differ (u8 differing: %d)" % (what, nb, int((d > 0).sum()))
    assert int((d > 0).sum()) == 0


CASES = [
    # name, scene, w, h, kwargs  (BASELINE.json configs at oracle-sized resolutions + edge cases)
    ("cfg1_spheres1_d1", "spheres1.scn", 640, 360, dict(depth=1)),
    ("spheres1_shadow", "spheres1.scn", 320, 180, dict(shadow=True)),
    ("cfg2_spheres2_js5_shadow", "spheres2.scn", 240, 135, dict(jsample=5, shadow=True, seed=42)),
    ("cfg3_spheres2_gi16_shadow", "spheres2.scn", 240, 135, dict(gillum=16, shadow=True, seed=20261004)),
    ("spheres2_gi16_noshadow", "spheres2.scn", 160, 90, dict(gillum=16, seed=3)),
    ("spheres2_gi4_js2_d2", "spheres2.scn", 160, 90, dict(gillum=4, jsample=2, depth=2, shadow=True, seed=5)),
    ("spheres2_gi3_d4", "spheres2.scn", 96, 54, dict(gillum=3, depth=4, shadow=True, seed=12)),
    ("spheres2_gi2_d6", "spheres2.scn", 64, 36, dict(gillum=2, depth=6, shadow=True, seed=8)),
    ("spheres2_gi2_d8", "spheres2.scn", 48, 27, dict(gillum=2, depth=8, shadow=True, seed=8)),        # beyond the old cap of 6 (main.cpp:318-329 takes any positive depth)
    ("spheres2_gi3_d5_js2", "spheres2.scn", 40, 23, dict(gillum=3, depth=5, jsample=2, shadow=True, seed=4)),
    ("spheres2_nogi_d9", "spheres2.scn", 64, 36, dict(depth=9, shadow=True)),                        # without --gillum shade() never recurses: the depth-1 image
    ("dragon_gi4_d7", "dragon.scn", 64, 48, dict(gillum=4, depth=7)),                                 # no spheres: nothing to recurse under
    ("test_mixed_gi3_d4", "test.scn", 64, 36, dict(gillum=3, depth=4, shadow=True, seed=3)),          # triangles under the node pipeline
    ("spheres2_gi5_odd", "spheres2.scn", 100, 57, dict(gillum=5, shadow=True, seed=77)),   # odd N: half-used Philox pair; ragged tiles
    ("spheres2_gi1", "spheres2.scn", 64, 36, dict(gillum=1, shadow=True, seed=1)),
    ("spheres2_gi0_nan", "spheres2.scn", 64, 36, dict(gillum=0, shadow=True)),           # N=0: 0/0 -> NaN -> 255 (main.cpp:205)
    ("spheres2_fov90_ragged", "spheres2.scn", 203, 151, dict(fov=90.0, shadow=True)),     # W%4 != 0: byte store path
    ("cfg4_dragon", "dragon.scn", 160, 120, dict(gillum=16)),
    ("test_mixed_gi4", "test.scn", 80, 60, dict(gillum=4, shadow=True, seed=3)),
    ("test_mixed_shadow", "test.scn", 160, 120, dict(shadow=True)),
    ("bear_shadow", "bear.scn", 160, 120, dict(shadow=True)),
    ("bear_gi8", "bear.scn", 96, 72, dict(gillum=8, shadow=True, seed=9)),
    ("spheres2_gi64", "spheres2.scn", 48, 27, dict(gillum=64, shadow=True, seed=6)),        # large-gillum LDS budget (Cfg<2>)
    ("spheres2_gi33_d2", "spheres2.scn", 64, 36, dict(gillum=33, depth=2, shadow=True, seed=2)),
    ("spheres2_gi256_d2", "spheres2.scn", 24, 14, dict(gillum=256, depth=2, seed=2)),          # largest gillum the streaming kernel takes
    ("spheres2_gi300_d2", "spheres2.scn", 16, 9, dict(gillum=300, depth=2, seed=2)),           # beyond it: per-pixel kernel
    ("spheres1_gi7_js2", "spheres1.scn", 64, 36, dict(gillum=7, jsample=2, shadow=True, seed=21)),
    ("tiny_1x1", "spheres2.scn", 1, 1, dict(gillum=4, shadow=True)),
    # --strict-scn: spheres2.scn's two directional lights pushed and shaded (blinn_phong.h:77-85,122-131; shadow test utils.h:60-76)
    ("strict_spheres2_shadow", "spheres2.scn", 240, 135, dict(shadow=True, strict=True)),
    ("strict_spheres2_noshadow_js2", "spheres2.scn", 160, 90, dict(jsample=2, seed=3, strict=True)),
    ("strict_spheres2_gi8_shadow", "spheres2.scn", 160, 90, dict(gillum=8, shadow=True, seed=11, strict=True)),
    ("strict_spheres2_gi3_d4", "spheres2.scn", 64, 36, dict(gillum=3, depth=4, shadow=True, seed=5, strict=True)),
    ("strict_spheres1_same_as_default", "spheres1.scn", 96, 54, dict(gillum=4, shadow=True, seed=2, strict=True)),  # no directional light in the file
    ("tall_3x70", "spheres1.scn", 3, 70, dict(jsample=2, shadow=True)),
]


@pytest.mark.parametrize("schedule", ["auto", "persistent"])
@pytest.mark.parametrize("name,scn,w,h,kw", CASES, ids=[c[0] for c in CASES])
def test_gpu_matches_oracle_bit_for_bit(gpu, oracle, monkeypatch, name, scn, w, h, kw, schedule):
    # the node pipeline has two schedules: frames this small take the flat one by themselves; SKR_FLAT=0 puts
    # the same cases through the persistent leaf kernel, which full-size frames use
    if schedule == "persistent":
        if kw.get("gillum") is None:
            pytest.skip("no --gillum tree: the node pipeline is not involved")
        monkeypatch.setenv("SKR_FLAT", "0")
    g_rgb, g_f, cnt = gpu_render(scn, w, h, **kw)
    o_rgb, o_f, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
    compare(g_rgb, g_f, o_rgb, o_f, name)
    if kw.get("strict") and scn == "spheres2.scn":  # and the lights do change the picture
        d_rgb, _, _ = gpu_render(scn, w, h, **dict(kw, strict=False))
        assert (d_rgb != g_rgb).mean() > 0.001  # (both lights point downwards: only surfaces facing down see them)
    # the work counters are part of the metric: they must agree with the oracle's count
    assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])
    if kw.get("shadow"):
        assert cnt["shadow_rays"] == int(st[2])


REF_CASES = [(n, c) for n, c in sorted(manifest()["cases"].items())
             if "--gillum" not in c["args"] and "--jsample" not in c["args"] and n != "dragon_parallel_entry"]


@pytest.mark.parametrize("name,case", REF_CASES, ids=[c[0] for c in REF_CASES])
def test_gpu_matches_reference_output_where_no_rng_is_involved(gpu, name, case):
    """Deterministic configurations: compare straight against PPMs written by the reference's own
    shade() (tests/golden, made with oracle/_ref).  Only powf differs in provenance (spec vs libm)."""
    kw = args_to_kwargs(case["args"])
    w, h = kw.pop("width"), kw.pop("height")
    gold = read_golden_ppm(case["file"]).astype(np.int32)
    g_rgb, _, _ = gpu_render(case["scene"], w, h, want_float=False, **kw)
    d = np.abs(g_rgb.astype(np.int32) - gold)
    assert d.max() <= TOL_U8
    assert (d > 0).mean() < 1e-4


def test_reference_fixture_testcpu_on_gpu(gpu):
    """renders/testcpu.ppm (the reference's own pixel-exact fixture) reproduced by the HIP path."""
    gold = read_golden_ppm("testcpu.ppm.gz")
    g_rgb, _, _ = gpu_render("dragon.scn", 640, 480, want_float=False, depth=1)
    assert np.array_equal(g_rgb, gold)


@pytest.mark.parametrize("tile_rows,G", [(16, 2), (8, 3), (16, 8), (5, 4)])
def test_partition_independence(gpu, tile_rows, G):
    """Row-tile interleaving over G 'ranks' reassembles to the single-launch frame, byte for byte
    (RNG keyed by global pixel index)."""
    w, h = 200, 117
    opt = skr.Options(w, h, gillum=4, shadow=True, seed=11)
    r = renderer("spheres2.scn")
    full, _ = r.render(opt)
    full = full.cpu().numpy()
    out = np.zeros_like(full)
    n_tiles = (h + tile_rows - 1) // tile_rows
    for rank in range(G):
        part, _ = r.render(opt, tile_rows=tile_rows, first_tile=rank, tile_stride=G)
        part = part.cpu().numpy()
        for k, t in enumerate(range(rank, n_tiles, G)):
            y0, y1 = t * tile_rows, min(h, (t + 1) * tile_rows)
            out[y0:y1] = part[k * tile_rows:k * tile_rows + (y1 - y0)]
    assert np.array_equal(out, full)
    rows, _ = r.render_rows(opt, 32, 80)
    assert np.array_equal(rows.cpu().numpy(), full[32:80])


def test_full_size_headline_config_rows_against_oracle(gpu, oracle):
    """BASELINE config 3 at its real size (1920x1080 gillum 16 shadows): the whole frame is rendered
    on the GPU, a band of rows is checked bit for bit against the oracle, and the frame-level
    invariants (ray counters, determinism) are checked on the full frame."""
    w, h = 1920, 1080
    kw = dict(gillum=16, shadow=True, seed=20261004)
    g_rgb, g_f, cnt = gpu_render("spheres2.scn", w, h, **kw)
    for y0, y1 in ((400, 404), (700, 703), (1076, 1080)):
        o_rgb, o_f, _ = oracle.render(scene_path("spheres2.scn"), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED,
                                      want_float=True, y0=y0, y1=y1, **kw)
        compare(g_rgb[y0:y1], g_f[y0:y1], o_rgb, o_f, "rows %d-%d" % (y0, y1))
    assert w * h <= cnt["radiance_rays"] <= skr.radiance_ray_count(skr.Options(w, h, **kw))
    again, _, cnt2 = gpu_render("spheres2.scn", w, h, want_float=False, **kw)
    assert np.array_equal(again, g_rgb) and cnt2 == cnt  # idempotent
    other, _, _ = gpu_render("spheres2.scn", w, h, want_float=False, gillum=16, shadow=True, seed=1)
    assert not np.array_equal(other, g_rgb)  # the seed matters


def test_full_size_jsample_config_rows_against_oracle(gpu, oracle):
    w, h = 1920, 1080
    kw = dict(jsample=5, shadow=True, seed=9)
    g_rgb, g_f, cnt = gpu_render("spheres2.scn", w, h, **kw)
    o_rgb, o_f, _ = oracle.render(scene_path("spheres2.scn"), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED,
                                  want_float=True, y0=560, y1=600, **kw)
    compare(g_rgb[560:600], g_f[560:600], o_rgb, o_f, "js5 rows 560-600")
    assert cnt["radiance_rays"] == 1920 * 1080 * 25


def test_unsupported_configs_fail_loudly(gpu):
    r = renderer("spheres2.scn")
    with pytest.raises(skr.SkrError):
        r.render(skr.Options(64, 36, depth=0))
    with pytest.raises(skr.SkrError, match="2\\^32"):  # tree node ids are 32-bit RNG counter words
        r.render(skr.Options(64, 36, gillum=16, depth=10))
    with pytest.raises(skr.SkrError, match="budget"):   # one 16x16 block of --gillum 16 --depth 7 needs ~12 GB of tables
        r.render(skr.Options(64, 36, gillum=16, depth=7))


def test_cli_drop_in_writes_the_same_ppm(gpu, oracle, tmp_path):
    """bin/raytracer (reference command line, main.cpp:230-413) end to end: flags anywhere in argv,
    unknown tokens ignored (`--shadow on`), P6 output byte-identical to the oracle's frame."""
    import os
    import subprocess
    from conftest import ROOT, read_ppm_bytes
    exe = os.path.join(ROOT, "bin", "raytracer")
    assert os.path.exists(exe), "run `make cli`"
    out = str(tmp_path / "cli.ppm")
    cmd = [exe, "--output", out, "--gillum", "4", "--shadow", "on", "--width", "160", "--height", "90", "--parallel", "true",
           "--path", scene_path("spheres2.scn"), "--seed", "7", "bogus-token"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert "WROTE TO PPM" in res.stdout and "Monte carlo: 1" in res.stdout and "Sphere as position" in res.stdout
    got = read_ppm_bytes(open(out, "rb").read())
    want, _, _ = oracle.render(scene_path("spheres2.scn"), 160, 90, gillum=4, shadow=True, seed=7)
    assert np.array_equal(got, want)
    # usage errors: message on stderr, exit status 0 (main.cpp:381-391)
    res = subprocess.run([exe, "--output", out], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "no scene file was passed" in res.stderr
    res = subprocess.run([exe, "--path", scene_path("spheres2.scn"), "--output", out, "--depth", "0"], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "depth takes a positive int" in res.stderr
    res = subprocess.run([exe, "--path", str(tmp_path / "nope.scn"), "--output", out], capture_output=True, text=True, timeout=60)
    assert res.returncode == 0 and "Can't open file" in res.stdout


def test_cli_strict_scn(gpu, oracle, tmp_path):
    """`raytracer --strict-scn` (both front ends): the directional lights are shaded, and the .scn's own film_resolution and
    max_depth hold for whatever the command line leaves open (spheres2.scn: max_depth 2; test.scn: film_resolution 1024 768,
    max_depth 10) — the reference parses all three and then drops / overrides / never reads them."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, read_ppm_bytes
    exe = os.path.join(ROOT, "bin", "raytracer")
    out = str(tmp_path / "strict.ppm")
    base = ["--output", out, "--path", scene_path("spheres2.scn"), "--width", "160", "--height", "90", "--gillum", "4", "--shadow", "--seed", "7", "--quiet"]
    subprocess.run([exe] + base + ["--strict-scn"], check=True, capture_output=True, timeout=120)
    got = read_ppm_bytes(open(out, "rb").read())
    want, _, _ = oracle.render(scene_path("spheres2.scn"), 160, 90, gillum=4, shadow=True, seed=7, depth=2, strict=True)  # the file's max_depth 2
    assert np.array_equal(got, want)
    subprocess.run([exe] + base + ["--strict-scn", "--depth", "3"], check=True, capture_output=True, timeout=120)  # argv wins
    want3, _, _ = oracle.render(scene_path("spheres2.scn"), 160, 90, gillum=4, shadow=True, seed=7, depth=3, strict=True)
    assert np.array_equal(read_ppm_bytes(open(out, "rb").read()), want3) and not np.array_equal(want3, want)
    env = dict(os.environ, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, "-m", "skele_raytracer_amd.render_cli"] + base[:-1] + ["--strict-scn"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr
    assert np.array_equal(read_ppm_bytes(open(out, "rb").read()), want)
    res = subprocess.run([exe, "--output", out, "--path", scene_path("test.scn"), "--strict-scn", "--shadow", "--quiet"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    got = read_ppm_bytes(open(out, "rb").read())
    assert got.shape == (768, 1024, 3)
    want, _, _ = oracle.render(scene_path("test.scn"), 1024, 768, shadow=True, depth=10, strict=True, y0=300, y1=340)
    assert np.array_equal(got[300:340], want)


def _write_synthetic_scn(path, rng, n_spheres, n_lights, n_tris):
    """A random but well-conditioned scene in the reference's .scn grammar."""
    lines = ["camera 0 1.5 -9 0 -.05 1 0 1 0 30", "background .1 .2 .3", "ambient_light .3 .3 .3"]
    lines += ["material .6 .6 .6 .7 .7 .7 .2 .2 .2 8 0 0 0 1", "sphere 0 -40 0 40"]
    for _ in range(n_spheres - 1):
        ka, kd, ks = rng.random(3), rng.random(3), rng.random(3) * 0.5
        lines.append("material %g %g %g %g %g %g %g %g %g %d 0 0 0 1" % (*ka, *kd, *ks, int(rng.choice([1, 2, 7, 16, 33, 100]))))
        lines.append("sphere %g %g %g %g" % (rng.uniform(-6, 6), rng.uniform(0.2, 4), rng.uniform(-3, 8), rng.uniform(0.3, 1.2)))
    lines.append("material 0 0 0 .5 .5 .5 .5 .5 .5 2.5 0 0 0 1")  # non-integer phong power: general pow branch
    lines.append("sphere 2 1 -2 .8")
    for _ in range(3 * n_tris):
        lines.append("vertex %g %g %g" % (rng.uniform(-5, 5), rng.uniform(0, 5), rng.uniform(2, 9)))
    for i in range(n_tris):
        lines.append("triangle %d %d %d" % (3 * i, 3 * i + 1, 3 * i + 2))
    for _ in range(n_lights):
        lines.append("point_light %g %g %g %g %g %g" % (*rng.uniform(5, 40, 3), rng.uniform(-8, 8), rng.uniform(3, 9), rng.uniform(-8, 8)))
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("n_spheres,n_lights,n_tris,kw", [
    (40, 3, 0, dict(gillum=6, shadow=True, seed=1)),     # odd light count, > 32 spheres (large LDS budget)
    (9, 1, 5, dict(gillum=8, shadow=True, seed=2)),      # one light, a few triangles among GI rays
    (5, 0, 0, dict(gillum=4, shadow=True, seed=3)),      # no lights at all
    (70, 5, 3, dict(jsample=2, shadow=True, seed=4)),    # many lights, no GI
    (3, 2, 0, dict(gillum=9, depth=2, seed=5)),
], ids=["40s3l", "9s1l5t", "5s0l", "70s5l3t", "3s2l_d2"])
def test_synthetic_scenes_match_oracle(gpu, oracle, tmp_path, n_spheres, n_lights, n_tris, kw):
    rng = np.random.default_rng(n_spheres * 1000 + n_lights)
    scn = str(tmp_path / "synthetic.scn")
    _write_synthetic_scn(scn, rng, n_spheres, n_lights, n_tris)
    w, h = 96, 54
    sc = skr.parse_scene(scn)
    r = skr.Renderer(sc)
    rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
    gpu.cuda.synchronize()
    o_rgb, o_f, st = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
    compare(rgb.cpu().numpy(), rgbf.cpu().numpy(), o_rgb, o_f, "synthetic %d/%d/%d" % (n_spheres, n_lights, n_tris))
    cnt = r.counters()
    assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])


@pytest.mark.parametrize("n_spheres", [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 17])
def test_sphere_counts_around_the_trip_size(gpu, oracle, tmp_path, monkeypatch, n_spheres):
    """The sphere loops take the sphere rows four per trip through the scalar cache and ask for the rows of the next trip without a
    bounds test (shade_common.h table_rows: the tables are padded): every count around the trip size, through the direct kernel, both
    schedules of the node pipeline, the general level pipeline and --legacy-reflect, against the oracle bit for bit, counts included."""
    rng = np.random.default_rng(500 + n_spheres)
    scn = str(tmp_path / "count.scn")
    _write_synthetic_scn(scn, rng, n_spheres, 2, 0)
    w, h = 64, 36
    r = skr.Renderer(skr.parse_scene(scn))
    for kw, env in ((dict(shadow=True, jsample=2, seed=1), {}),
                    (dict(gillum=4, shadow=True, seed=2), {"SKR_FLAT": "0"}),
                    (dict(gillum=4, shadow=True, seed=2), {"SKR_FLAT": "1"}),
                    (dict(gillum=3, depth=4, shadow=True, seed=3), {"SKR_PIPELINE": "generic"}),
                    (dict(depth=3, shadow=True, legacy_reflect=True), {})):
        for k in ("SKR_FLAT", "SKR_PIPELINE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r.counters(reset=True)
        rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
        gpu.cuda.synchronize()
        cnt = r.counters(reset=True)
        o_rgb, o_f, st = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
        compare(rgb.cpu().numpy(), rgbf.cpu().numpy(), o_rgb, o_f, "%d spheres %s %s [%s]" % (n_spheres, kw, env, r.kernel_variant()))
        assert (cnt["radiance_rays"], cnt["sphere_hits"], cnt["shadow_rays"]) == tuple(int(v) for v in st[:3]), (n_spheres, kw, env)


def test_full_size_config5_rows_against_oracle(gpu, oracle):
    """BASELINE config 5 at its real size (3840x2160 --gillum 64 --jsample 5 --shadow, ~9e10 radiance rays):
    whole frame on the GPU (25 AA samples through the parent-queue pipeline), two row bands bit for bit
    against the oracle."""
    w, h = 3840, 2160
    kw = dict(gillum=64, jsample=5, shadow=True, seed=5)
    g_rgb, g_f, cnt = gpu_render("spheres2.scn", w, h, **kw)
    for y0, y1 in ((1399, 1400), (2158, 2159)):
        o_rgb, o_f, _ = oracle.render(scene_path("spheres2.scn"), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED,
                                      want_float=True, y0=y0, y1=y1, **kw)
        compare(g_rgb[y0:y1], g_f[y0:y1], o_rgb, o_f, "4K rows %d-%d" % (y0, y1))
    assert cnt["radiance_rays"] > 25 * w * h


def test_all_kernel_variants_agree(gpu, monkeypatch):
    """The product has one arithmetic spec and three schedules of a --gillum tree: the node pipeline with its persistent leaf
    kernel (full-size frames), the node pipeline's flat schedule (small launches) — each in one band or several — and the general
    level pipeline (one lane per ray: meshes, --shade-triangles, --legacy-reflect, here forced onto a sphere scene).  Every one
    of them must produce the same bits and the same ray counts."""
    w, h = 176, 99
    r = renderer("spheres2.scn")
    knobs = ("SKR_PIPELINE", "SKR_LEVELS_BUDGET_MB", "SKR_FLAT")

    def run(opt, env):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r.counters(reset=True)
        rgb, rgbf = r.render(opt, want_float=True)
        gpu.cuda.synchronize()
        return rgb.cpu().numpy(), rgbf.cpu().numpy().view(np.uint32), r.counters(), r.kernel_variant()

    opt = skr.Options(w, h, gillum=8, shadow=True, seed=31)
    base = run(opt, {})
    assert base[3] == "node_levels_v5_flat"  # (a frame this small: every level a plain grid)
    seen = {base[3]}
    for env in ({"SKR_FLAT": "0"},              # the persistent leaf kernel, as on a full-size frame
                {"SKR_LEVELS_BUDGET_MB": "2"},  # 2 MiB of tables: bands of a few 16x16 blocks (the persistent schedule: flat runs in one piece only)
                {"SKR_FLAT": "1", "SKR_LEVELS_BUDGET_MB": "8"},  # flat forced into bands
                {"SKR_PIPELINE": "generic"}, {"SKR_PIPELINE": "generic", "SKR_LEVELS_BUDGET_MB": "8"}):  # the general level pipeline, whole and in bands of rows
        got = run(opt, env)
        seen.add(got[3])
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), env
        assert got[2] == base[2], env
    assert seen == {"node_levels_v5_flat", "node_levels_v5", "level_pipeline_g1"}
    # depth 2 (the leaf kernel works on the primary hits), depth 4 (one activate + trace level in between), depth 1 under --gillum and AA
    for opt2, others in ((skr.Options(w, h, gillum=8, shadow=True, depth=2, seed=31), ({"SKR_PIPELINE": "generic"}, {"SKR_FLAT": "0"})),
                         (skr.Options(96, 54, gillum=3, shadow=True, depth=4, seed=31), ({"SKR_PIPELINE": "generic"}, {"SKR_LEVELS_BUDGET_MB": "8"}, {"SKR_FLAT": "0"})),
                         (skr.Options(96, 54, gillum=5, jsample=2, shadow=True, depth=3, seed=7), ({"SKR_PIPELINE": "generic"}, {"SKR_FLAT": "0"}))):
        b2 = run(opt2, {})
        assert b2[3] == "node_levels_v5_flat"
        for env in others:
            got = run(opt2, env)
            assert np.array_equal(got[0], b2[0]) and np.array_equal(got[1], b2[1]) and got[2] == b2[2], env
    for opt3 in (skr.Options(w, h, gillum=8, shadow=True, depth=1, seed=31), skr.Options(w, h, jsample=3, shadow=True, seed=31)):  # no tree: the direct kernel
        b3 = run(opt3, {})
        assert b3[3] == "direct_v3"
        got = run(opt3, {"SKR_PIPELINE": "generic"})
        assert got[3] == "level_pipeline_g1" and np.array_equal(got[0], b3[0]) and np.array_equal(got[1], b3[1]) and got[2] == b3[2]
    for k in knobs:
        monkeypatch.delenv(k, raising=False)


def test_node_pipeline_in_bands_on_triangles_and_deep(gpu, oracle, monkeypatch):
    """The node pipeline against the oracle where it is not in one piece or not the default: bands of a few 16x16 blocks
    with AA (several bands x several samples), odd N, a triangle scene it is forced onto, N = 255 and 256, depth 5 in bands."""
    for scn, w, h, kw, env in (("test.scn", 96, 54, dict(gillum=4, shadow=True, seed=3), {"SKR_PIPELINE": "nodes"}),
                               ("spheres2.scn", 200, 113, dict(gillum=6, jsample=2, shadow=True, seed=9), {"SKR_LEVELS_BUDGET_MB": "2"}),
                               ("spheres2.scn", 131, 77, dict(gillum=5, shadow=True, seed=4), {"SKR_LEVELS_BUDGET_MB": "1"}),
                               ("spheres2.scn", 131, 77, dict(gillum=3, depth=5, shadow=True, seed=4), {"SKR_LEVELS_BUDGET_MB": "24"}),
                               ("bear.scn", 160, 90, dict(gillum=255, seed=4), {}),
                               ("spheres2.scn", 24, 14, dict(gillum=256, depth=2, seed=2), {})):
        for k in ("SKR_PIPELINE", "SKR_LEVELS_BUDGET_MB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = renderer(scn)
        r.counters(reset=True)
        rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
        gpu.cuda.synchronize()
        assert r.kernel_variant() in ("node_levels_v5", "node_levels_v5_flat"), (scn, kw, env)
        o_rgb, o_f, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
        compare(rgb.cpu().numpy(), rgbf.cpu().numpy(), o_rgb, o_f, "%s %s %s" % (scn, kw, env))
        cnt = r.counters()
        assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1]) and cnt["shadow_rays"] == int(st[2])
    for k in ("SKR_PIPELINE", "SKR_LEVELS_BUDGET_MB"):
        monkeypatch.delenv(k, raising=False)


def test_general_level_pipeline_on_meshes_in_bands_and_wide(gpu, oracle, monkeypatch):
    """The general level pipeline (render_generic.hip) against the oracle where it is the product path — triangle meshes under
    --gillum, more than 256 children per node — and in bands of rows with AA (several bands x several samples), at depth 6."""
    for scn, w, h, kw, env, forced in (("test.scn", 96, 54, dict(gillum=4, shadow=True, seed=3), {}, False),
                                       ("test.scn", 64, 36, dict(gillum=3, depth=6, shadow=True, seed=3), {"SKR_LEVELS_BUDGET_MB": "16"}, False),
                                       ("dragon.scn", 96, 54, dict(gillum=4, depth=5), {}, False),
                                       ("spheres2.scn", 200, 113, dict(gillum=6, jsample=2, shadow=True, seed=9), {"SKR_PIPELINE": "generic", "SKR_LEVELS_BUDGET_MB": "8"}, True),
                                       ("spheres2.scn", 24, 14, dict(gillum=300, depth=2, seed=2), {}, False),
                                       ("bear.scn", 80, 45, dict(gillum=40, depth=3, seed=4), {"SKR_PIPELINE": "generic"}, True)):
        for k in ("SKR_PIPELINE", "SKR_LEVELS_BUDGET_MB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        r = renderer(scn)
        r.counters(reset=True)
        rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
        gpu.cuda.synchronize()
        want = "direct_v3" if scn == "dragon.scn" else "level_pipeline_g1"  # (no spheres: shade() never recurses, api.cpp folds the depth)
        assert r.kernel_variant() == want, (scn, kw, env, r.kernel_variant())
        o_rgb, o_f, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
        compare(rgb.cpu().numpy(), rgbf.cpu().numpy(), o_rgb, o_f, "%s %s %s" % (scn, kw, env))
        cnt = r.counters()
        assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1]) and cnt["shadow_rays"] == int(st[2])
    for k in ("SKR_PIPELINE", "SKR_LEVELS_BUDGET_MB"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("scn,w,h,kw", [("dragon.scn", 1920, 1080, dict(gillum=16)), ("test.scn", 320, 180, dict(gillum=4, shadow=True, seed=2)),
                                        ("dragon.scn", 333, 187, dict(fov=120.0, jsample=2, seed=4))], ids=["dragon_1080p", "test_gi", "dragon_fov120_aa"])
def test_triangle_chunk_culling_changes_nothing(gpu, monkeypatch, scn, w, h, kw):
    """The chunk spheres (scene_host.cpp build_triangle_chunks) may only skip triangles that would have failed
    utils.h:181-213 anyway: the culled walk must reproduce the brute-force walk bit for bit (BASELINE config 4 at full size)."""
    r = renderer(scn)
    opt = skr.Options(w, h, **kw)
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    a, af = r.render(opt, want_float=True)
    monkeypatch.setenv("SKR_NO_CULL", "1")
    b, bf = r.render(opt, want_float=True)
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    gpu.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert np.array_equal(af.cpu().numpy().view(np.uint32), bf.cpu().numpy().view(np.uint32))


def _write_triangle_soup(path, rng, n_tris, with_spheres):
    """Many small triangles of very different sizes (edge 1e-3 .. 2, slivers included) in front of the camera:
    the culling spheres of scene_host.cpp must never hide one from a ray the brute-force oracle lets it stop."""
    lines = ["camera 0 1.5 -9 0 -.05 1 0 1 0 30", "background .2 .3 .4", "ambient_light .3 .3 .3"]
    if with_spheres:
        lines += ["material .6 .6 .6 .7 .7 .7 .2 .2 .2 8 0 0 0 1", "sphere 0 -40 0 40",
                  "material .2 .5 .6 .3 .7 .7 .4 .4 .4 16 0 0 0 1", "sphere -2 1 1 1", "sphere 2.5 1.2 3 1.2",
                  "point_light 30 30 30 6 8 -6"]
    for _ in range(n_tris):
        c = np.array([rng.uniform(-6, 6), rng.uniform(-1, 6), rng.uniform(-2, 12)])
        size = 10.0 ** rng.uniform(-3, 0.3)
        a, b = rng.normal(size=3) * size, rng.normal(size=3) * size
        if rng.random() < 0.2:
            b = a * rng.uniform(0.5, 2) + rng.normal(size=3) * size * 1e-3  # sliver: |det| near the 1e-5 cut
        for v in (c, c + a, c + b):
            lines.append("vertex %.9g %.9g %.9g" % tuple(v))
    for i in range(n_tris):
        lines.append("triangle %d %d %d" % (3 * i, 3 * i + 1, 3 * i + 2))
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("n_tris,with_spheres,kw", [
    (700, True, dict(gillum=4, depth=2, shadow=True, seed=11)),   # GI children start on spheres, incoherent waves
    (1500, False, dict(jsample=2, seed=12)),                     # sphere-free: 8-triangle chunks, jittered camera rays
    (333, True, dict(fov=150.0, seed=13)),                       # long un-normalised primary directions (|d| ~ 7)
], ids=["soup700_gi", "soup1500_aa", "soup333_fov150"])
def test_triangle_soup_matches_oracle(gpu, oracle, tmp_path, n_tris, with_spheres, kw):
    rng = np.random.default_rng(n_tris)
    scn = str(tmp_path / "soup.scn")
    _write_triangle_soup(scn, rng, n_tris, with_spheres)
    w, h = 128, 72
    r = skr.Renderer(skr.parse_scene(scn))
    rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
    gpu.cuda.synchronize()
    o_rgb, o_f, st = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
    compare(rgb.cpu().numpy(), rgbf.cpu().numpy(), o_rgb, o_f, "soup %d" % n_tris)
    black = int((o_rgb.reshape(-1, 3).sum(axis=1) == 0).sum())
    assert black > 10, "the soup must actually cover pixels (%d black)" % black


@pytest.mark.parametrize("name,scn,kw,rows", [
    ("config2_jsample5", "spheres2.scn", dict(jsample=5, shadow=True, seed=9), None),
    ("config3_gillum16", "spheres2.scn", dict(gillum=16, shadow=True, seed=20261004), None),
    ("config4_dragon", "dragon.scn", dict(gillum=16), None),        # 2e10 brute-force triangle tests on the CPU side: ~20 s on the box
], ids=["config2", "config3", "config4"])
def test_whole_frames_of_the_baseline_configs_against_oracle(gpu, oracle, name, scn, kw, rows):
    """BASELINE.json configs[1..3] at 1920x1080: every pixel of the frame bit for bit against the oracle, u8 and
    float, plus the ray counters (config 4: the culled triangle walk against the oracle's brute-force one)."""
    w, h = 1920, 1080
    g_rgb, g_f, cnt = gpu_render(scn, w, h, **kw)
    y0, y1 = rows if rows else (0, h)
    o_rgb, o_f, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, y0=y0, y1=y1, **kw)
    compare(g_rgb[y0:y1], g_f[y0:y1], o_rgb, o_f, name)
    if rows is None:
        assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])
    else:
        black = int((o_rgb.reshape(-1, 3).sum(axis=1) == 0).sum())
        assert black > 20000, "the band must cross the dragon (%d black pixels)" % black


def _write_bumpy_sphere(path, n_lat, n_lon):
    """A tessellated, radially displaced sphere of 2*n_lat*n_lon small triangles in front of the camera."""
    rng = np.random.default_rng(3)
    lines = ["camera 0 0 -6 0 0 1 0 1 0 30", "background .2 .3 .4"]
    th = np.linspace(0.02, np.pi - 0.02, n_lat + 1)
    ph = np.linspace(0, 2 * np.pi, n_lon + 1)
    rad = 1.5 + 0.05 * rng.random((n_lat + 1, n_lon + 1))
    rad[:, -1] = rad[:, 0]
    x = rad * np.sin(th)[:, None] * np.cos(ph)[None, :]
    y = rad * np.cos(th)[:, None] * np.ones_like(ph)[None, :]
    z = rad * np.sin(th)[:, None] * np.sin(ph)[None, :]
    v = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    lines += ["vertex %.7g %.7g %.7g" % tuple(p) for p in v]
    idx = lambda i, j: i * (n_lon + 1) + j
    for i in range(n_lat):
        for j in range(n_lon):
            lines.append("triangle %d %d %d" % (idx(i, j), idx(i + 1, j), idx(i, j + 1)))
            lines.append("triangle %d %d %d" % (idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)))
    open(path, "w").write("\n".join(lines) + "\n")
    return 2 * n_lat * n_lon


def test_large_mesh_walk_scales_and_stays_exact(gpu, oracle, tmp_path, monkeypatch):
    """125 000 triangles (12x the dragon): the tree walk must reproduce the brute-force walk on the whole frame and
    the oracle on a band of rows, and its cost must not follow the triangle count."""
    import time
    scn = str(tmp_path / "bumpy.scn")
    nt = _write_bumpy_sphere(scn, 250, 250)
    w, h = 1280, 720
    r = skr.Renderer(skr.parse_scene(scn))
    opt = skr.Options(w, h)
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    r.render(opt)
    gpu.cuda.synchronize()
    t0 = time.perf_counter()
    a, af = r.render(opt, want_float=True)
    gpu.cuda.synchronize()
    t_tree = time.perf_counter() - t0
    monkeypatch.setenv("SKR_NO_CULL", "1")
    t0 = time.perf_counter()
    b, bf = r.render(opt, want_float=True)
    gpu.cuda.synchronize()
    t_brute = time.perf_counter() - t0
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert np.array_equal(a, b) and np.array_equal(af.cpu().numpy().view(np.uint32), bf.cpu().numpy().view(np.uint32))
    y0, y1 = 356, 364
    o_rgb, o_f, _ = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, y0=y0, y1=y1)
    compare(a[y0:y1], af.cpu().numpy()[y0:y1], o_rgb, o_f, "bumpy sphere rows")
    black = int((a.reshape(-1, 3).sum(axis=1) == 0).sum())
    assert black > 0.05 * w * h, "the mesh must cover a good part of the frame (%d black pixels)" % black
    print("\n%d triangles %dx%d: tree walk %.2f ms, brute force %.1f ms" % (nt, w, h, t_tree * 1e3, t_brute * 1e3))
    assert t_tree * 20 < t_brute


def _write_grazing_floor(path, with_spheres):
    """A tessellated strip of floor (4 x 24 unit quads) seen through a narrow lens from a camera 0.01 above its plane:
    camera rays meet it between 0.02 rad and 0.0004 rad of grazing, across the kappa = 1e-3 of the culling data's cone
    test — where it has to hand over to the general radius."""
    lines = ["camera 0 0.01 -6 0 0 1 0 1 0 30", "background .2 .3 .4", "ambient_light .3 .3 .3"]
    if with_spheres:
        lines += ["material .6 .6 .6 .7 .7 .7 .2 .2 .2 8 0 0 0 1", "sphere -0.4 -0.9 3 1.0", "sphere 0.5 -0.75 9 0.8", "point_light 30 30 30 0 9 1"]
    nx, nz = 4, 24
    for i in range(nx + 1):
        for j in range(nz + 1):
            lines.append("vertex %g 0 %g" % (i - nx / 2.0, j - 5.5))
    idx = lambda i, j: i * (nz + 1) + j
    for i in range(nx):
        for j in range(nz):
            lines.append("triangle %d %d %d" % (idx(i, j), idx(i + 1, j), idx(i, j + 1)))
            lines.append("triangle %d %d %d" % (idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)))
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("with_spheres,kw", [(False, dict(fov=4.0)), (True, dict(fov=6.0, gillum=4, depth=2, shadow=True, seed=21)), (False, dict(jsample=2, fov=3.0, seed=22))],
                         ids=["camera_rays", "gi", "aa_fov3"])
def test_grazing_floor_cones_change_nothing(gpu, oracle, tmp_path, monkeypatch, with_spheres, kw):
    scn = str(tmp_path / "floor.scn")
    _write_grazing_floor(scn, with_spheres)
    w, h = 160, 90
    sc = skr.parse_scene(scn)
    ch = sc.culling(0)[4]
    assert np.any(ch[:, 4:7] != 0, axis=1).mean() > 0.9, "the floor's chunks must carry cones"
    r = skr.Renderer(sc)
    opt = skr.Options(w, h, **kw)
    outs = []
    for env in (None, "SKR_NO_CONES", "SKR_NO_CULL"):
        monkeypatch.delenv("SKR_NO_CONES", raising=False)
        monkeypatch.delenv("SKR_NO_CULL", raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        rgb, rgbf = r.render(opt, want_float=True)
        gpu.cuda.synchronize()
        outs.append((rgb.cpu().numpy(), rgbf.cpu().numpy()))
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    for rgb, rgbf in outs[1:]:
        assert np.array_equal(rgb, outs[0][0]) and np.array_equal(rgbf.view(np.uint32), outs[0][1].view(np.uint32))
    o_rgb, o_f, _ = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
    compare(outs[0][0], outs[0][1], o_rgb, o_f, "grazing floor")
    black = int((o_rgb.reshape(-1, 3).sum(axis=1) == 0).sum())
    assert 0.02 * w * h < black < 0.95 * w * h, "floor and sky must both be visible (%d black pixels)" % black


@pytest.mark.parametrize("seed", range(16))
def test_random_meshes_match_oracle(gpu, oracle, tmp_path, monkeypatch, seed):
    """Random planar patches + triangle soup under random options: the culled walk (cones where they exist) against the
    brute-force oracle, and against the GPU's own brute-force walk."""
    rng = np.random.default_rng(1000 + seed)
    scn = str(tmp_path / "mesh.scn")
    write_random_mesh_scene(scn, rng)
    kw = dict(fov=float(rng.choice([20, 45, 60, 100, 140])), seed=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.6:
        kw.update(gillum=int(rng.choice([2, 3, 5])), depth=int(rng.choice([2, 3])))
    if rng.random() < 0.4:
        kw.update(jsample=2)
    if rng.random() < 0.5:
        kw.update(shadow=True)
    w, h = 64, 40
    r = skr.Renderer(skr.parse_scene(scn))
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    rgb, rgbf = r.render(skr.Options(w, h, **kw), want_float=True)
    monkeypatch.setenv("SKR_NO_CULL", "1")
    rgb2, rgbf2 = r.render(skr.Options(w, h, **kw), want_float=True)
    monkeypatch.delenv("SKR_NO_CULL", raising=False)
    gpu.cuda.synchronize()
    a, af = rgb.cpu().numpy(), rgbf.cpu().numpy()
    assert np.array_equal(a, rgb2.cpu().numpy()) and np.array_equal(af.view(np.uint32), rgbf2.cpu().numpy().view(np.uint32))
    o_rgb, o_f, st = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, **kw)
    compare(a, af, o_rgb, o_f, "random mesh %d %s" % (seed, kw))


def test_distributed_cli_writes_the_same_ppm(gpu, tmp_path):
    """python -m skele_raytracer_amd.render_cli (one process per GPU, here world size 1 directly and under
    torch.distributed.run) must write the PPM the C++ raytracer writes, and keep the reference's usage behaviour."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    exe = os.path.join(ROOT, "bin", "raytracer")
    args = ["--path", scene_path("spheres2.scn"), "--width", "200", "--height", "120", "--gillum", "4", "--shadow", "--seed", "5", "--bogus", "x"]
    want = str(tmp_path / "cpp.ppm")
    subprocess.run([exe] + args + ["--output", want, "--quiet"], check=True, capture_output=True, timeout=120)
    env = dict(os.environ, PYTHONPATH=ROOT)
    got = str(tmp_path / "py.ppm")
    res = subprocess.run([sys.executable, "-m", "skele_raytracer_amd.render_cli"] + args + ["--output", got], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0 and "WROTE TO PPM" in res.stdout, res.stderr
    assert open(got, "rb").read() == open(want, "rb").read()
    got2 = str(tmp_path / "py2.ppm")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29577",
                          "-m", "skele_raytracer_amd.render_cli"] + args + ["--output", got2, "--tile-rows", "16"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr
    assert open(got2, "rb").read() == open(want, "rb").read()
    res = subprocess.run([sys.executable, "-m", "skele_raytracer_amd.render_cli", "--output", got], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert res.returncode == 0 and "no scene file was passed" in res.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_paths_rehearsed_on_one_gpu(gpu, tmp_path, world):
    """bench.py and render_cli under torch.distributed.run with `world` ranks, all on GPU 0 with gloo carrying the
    collectives (SKR_REHEARSE_GLOO=1): partition, gather, de-interleave and the rank reductions of the N > 1 path run
    on real kernels — the PPM must equal the single-process one and the ray count must not depend on the world size."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, PYTHONPATH=ROOT, SKR_REHEARSE_GLOO="1")
    launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
              "--master-port", str(29620 + world)]
    args = ["--path", scene_path("spheres2.scn"), "--width", "333", "--height", "187", "--gillum", "4", "--jsample", "2", "--shadow", "--seed", "5"]
    want = str(tmp_path / "one.ppm")
    subprocess.run([os.path.join(ROOT, "bin", "raytracer")] + args + ["--output", want, "--quiet"], check=True, capture_output=True, timeout=120)
    got = str(tmp_path / "many.ppm")
    res = subprocess.run(launch + ["-m", "skele_raytracer_amd.render_cli"] + args + ["--output", got], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    assert open(got, "rb").read() == open(want, "rb").read()
    res = subprocess.run(launch + [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["steps"] == 3 and out["scaling"] == "strong" and "REHEARSAL" in out["config"]["frame_step"]
    # the ray counts of one rank rendering the whole frame in this process (the headline configuration bench.py runs by default)
    r = renderer("spheres2.scn")
    r.work(reset=True)
    r.render(skr.Options(1920, 1080, gillum=16, shadow=True, depth=3, seed=20261004))
    gpu.cuda.synchronize()
    one = r.work(reset=True)
    assert out["config"]["rays_per_frame"] == float(one["radiance_rays"]) and out["config"]["shadow_rays_per_frame"] == float(one["shadow_rays"])
    assert "cpu_baseline" not in out and out["roofline"]["kernel_ms"] > 0


@pytest.mark.parametrize("inflight", ["1", "2"])
def test_pipelined_frame_step_returns_every_frame(gpu, monkeypatch, inflight):
    """skr_comm_render_frame_async: the collective and the de-interleave of frame f on the communicator's own stream while the
    caller's stream renders frame f + 1.  Six frames with six seeds (and a change of geometry in between): every frame handed back —
    one call late, the last one by skr_comm_flush — must be the frame a plain render of that seed gives.  SKR_INFLIGHT=2: what a
    world of more than one rank does by default — the odd frames of the run on a clone of the renderer (skr_renderer_clone) and a
    second stream, two shares in flight; the work counters of both land in the one set."""
    monkeypatch.setenv("SKR_INFLIGHT", inflight)
    r = renderer("spheres2.scn")
    st = gpu.cuda.current_stream()

    def frame_at(addr, w, h):  # a device address libskr owns -> numpy, in stream order
        buf = gpu.empty((h, w, 3), dtype=gpu.uint8, device="cuda")
        from skele_raytracer_amd.binding import C
        lib = C.CDLL("libamdhip64.so")
        lib.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        assert lib.hipMemcpyAsync(buf.data_ptr(), addr, w * h * 3, 3, st.cuda_stream) == 0  # 3 = device to device
        st.synchronize()
        return buf.cpu().numpy()

    for with_rccl in (False, True):
        c = binding.Comm(r, 0, 1, binding.comm_unique_id() if with_rccl else None)
        shapes = [(333, 187, 8)] * 3 + [(200, 113, 16)] * 3
        got, want = [], []
        gpu.cuda.synchronize()
        r.counters(reset=True)
        for k, (w, h, tile_rows) in enumerate(shapes):
            opt = skr.Options(w, h, gillum=4, shadow=True, seed=100 + k)
            prev = c.render_frame_async(opt, tile_rows, st.cuda_stream)
            if k == 0:
                assert prev is None
            else:
                pw, ph, _ = shapes[k - 1]
                got.append(frame_at(prev, pw, ph))
        last = c.flush(st.cuda_stream)
        got.append(frame_at(last, shapes[-1][0], shapes[-1][1]))
        in_run = r.counters(reset=True)
        c.close()
        for k, (w, h, _) in enumerate(shapes):
            f, _ = r.render(skr.Options(w, h, gillum=4, shadow=True, seed=100 + k))
            want.append(f.cpu().numpy())
        gpu.cuda.synchronize()
        assert in_run == r.counters(reset=True)  # (every ray of the run counted once, whichever renderer traced it)
        assert len(got) == len(want) == 6
        for k in range(6):
            assert np.array_equal(got[k], want[k]), (with_rccl, k)


def test_a_clone_renders_the_same_frames_on_its_own_stream(gpu):
    """skr_renderer_clone: the same uploaded scene, its own tables.  Two different frames enqueued at once on two streams — one on the
    renderer, one on its clone — are each the frame a plain render gives, and the shared work counters hold the rays of both."""
    r = renderer("spheres2.scn")
    a_opt, b_opt = skr.Options(320, 180, gillum=8, shadow=True, seed=5), skr.Options(256, 144, gillum=3, depth=4, shadow=True, seed=6)
    want_a, _ = r.render(a_opt)
    want_b, _ = r.render(b_opt)
    gpu.cuda.synchronize()
    each = r.counters(reset=True)
    c = r.clone()
    sa, sb = gpu.cuda.Stream(), gpu.cuda.Stream()
    buf_a = gpu.zeros((180, 320, 3), dtype=gpu.uint8, device="cuda")
    buf_b = gpu.zeros((144, 256, 3), dtype=gpu.uint8, device="cuda")
    for _ in range(3):
        r.render_tiles_into(a_opt, 180, 0, 1, buf_a.data_ptr(), None, sa.cuda_stream)
        c.render_tiles_into(b_opt, 144, 0, 1, buf_b.data_ptr(), None, sb.cuda_stream)
    gpu.cuda.synchronize()
    assert gpu.equal(buf_a, want_a) and gpu.equal(buf_b, want_b)
    both = c.counters(reset=True)  # (read through either)
    assert both["radiance_rays"] == 3 * each["radiance_rays"] and both["sphere_hits"] == 3 * each["sphere_hits"]
    c.close()
    again, _ = r.render(a_opt)
    assert gpu.equal(again, want_a)  # (the source outlives its clone)


def test_native_frame_step_on_one_gpu(gpu, tmp_path):
    """The multi-GPU frame step that lives inside libskr (include/skr.h "multi-GPU": tiles into the gather buffer, ONE
    ncclAllGather, de-interleave kernel), as far as a one-GPU box can run it: a world of one with a real RCCL communicator
    (skr_comm_*: what bench.py uses under torchrun), the single-process form on one device (skr_multi_*: what
    `raytracer --gpus N` uses) and the CLI flag itself — each must produce the frame of a plain skr_render_tiles call."""
    import os
    import subprocess
    from conftest import ROOT
    w, h = 333, 187
    opt = skr.Options(w, h, gillum=4, jsample=2, shadow=True, seed=5)
    r = renderer("spheres2.scn")
    want, _ = r.render(opt)
    want = want.cpu().numpy()
    assert binding.rccl_available()
    for with_rccl in (False, True):  # without RCCL at all; with a communicator of one rank
        for tile_rows in (8, 16, 5):
            c = binding.Comm(r, 0, 1, binding.comm_unique_id() if with_rccl else None)
            st = gpu.cuda.current_stream().cuda_stream
            assert c.render_frame(opt, tile_rows, st)
            got = c.frame_to_host(opt, st)
            assert np.array_equal(got, want), (with_rccl, tile_rows)
            c.close()
    m = binding.Multi(renderer("spheres2.scn").scene, 1)
    got, ms = m.render_frame_host(opt, 8)
    assert np.array_equal(got, want) and ms > 0
    assert m.counters()["radiance_rays"] > 0
    # the pipelined form: three frames under three seeds, each handed back one call late (the last by skr_multi_flush)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def fetch(addr):
        out = np.zeros((h, w, 3), np.uint8)
        assert hip.hipMemcpy(out.ctypes.data, addr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out
    seeds = (5, 6, 7)
    wants = [r.render(skr.Options(w, h, gillum=4, jsample=2, shadow=True, seed=sd))[0].cpu().numpy() for sd in seeds]
    back = []
    for sd in seeds:
        prev = m.render_frame_async(skr.Options(w, h, gillum=4, jsample=2, shadow=True, seed=sd), 8)
        if prev:
            back.append(fetch(prev))
    back.append(fetch(m.flush()))
    assert len(back) == 3 and all(np.array_equal(a, b) for a, b in zip(back, wants))
    m.close()
    exe = os.path.join(ROOT, "bin", "raytracer")
    args = ["--path", scene_path("spheres2.scn"), "--width", str(w), "--height", str(h), "--gillum", "4", "--jsample", "2", "--shadow", "--seed", "5", "--quiet"]
    a, b = str(tmp_path / "one.ppm"), str(tmp_path / "sharded.ppm")
    subprocess.run([exe] + args + ["--output", a], check=True, capture_output=True, timeout=120)
    res = subprocess.run([exe] + args + ["--output", b, "--gpus", "1", "--tile-rows", "16"], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and '"gpus": 1' in res.stderr, res.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
    res = subprocess.run([exe] + args + ["--output", b, "--gpus", "2"], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "visible" in res.stderr  # one GPU on this box: a loud failure, not a silent fallback


def test_cost_aware_tile_map_renders_the_same_frame(gpu, monkeypatch):
    """The frame steps deal the tiles `t mod G` unless the counted work of the tiles (skr_tile_costs) says that is more than 10 % off
    balance, then longest-processing-time-first (skr_shard_plan), and render each rank's list with skr_render_tile_list.  On one GPU:
    every rank's list of a world of 3, 4 and 8 under the forced LPT map and under the rule's, rendered in turn into its slot of a
    hand-made gather buffer, de-interleaved under the map — the plain frame, byte for byte."""
    w, h, tr = 333, 187, 8
    opt = skr.Options(w, h, gillum=4, shadow=True, seed=11)
    r = renderer("spheres2.scn")
    want, _ = r.render(opt)
    want = want.cpu().numpy()
    T = (h + tr - 1) // tr
    r.work(reset=True)
    cost = r.tile_costs(opt, tr)
    assert cost.min() > 0 and cost.max() > 10 * cost[0]   # sky on top (one ray per pixel, against every sphere), spheres and their trees below
    assert r.work()["radiance_rays"] == 0                 # (the probe leaves the caller's counters alone)
    st = gpu.cuda.current_stream().cuda_stream
    for G in (3, 4, 8):
        k_max = binding.shard_tiles_per_rank(h, tr, G)
        blind = (np.arange(T) % G) * k_max + np.arange(T) // G
        for rule in ("lpt", None, "interleave"):
            if rule: monkeypatch.setenv("SKR_SHARD", rule)
            else: monkeypatch.delenv("SKR_SHARD", raising=False)
            slot = r.shard_plan(opt, tr, G)
            assert len(set(slot.tolist())) == T
            if rule == "lpt": assert np.array_equal(slot, binding.shard_lpt(cost, G)) and not np.array_equal(slot, blind)
            elif rule == "interleave": assert np.array_equal(slot, blind)
            else: assert np.array_equal(slot, binding.shard_by_cost(cost, G))
            gathered = gpu.zeros((G * k_max * tr, w, 3), dtype=gpu.uint8, device="cuda")
            for rank in range(G):
                tiles = np.full(k_max, 0xFFFFFFFF, np.uint32)
                for t in range(T):
                    if slot[t] // k_max == rank:
                        tiles[slot[t] % k_max] = t
                d = gpu.from_numpy(tiles.astype(np.int64)).cuda().to(gpu.int32).contiguous()
                r.render_tile_list_into(opt, tr, d.data_ptr(), k_max, gathered[rank * k_max * tr:].data_ptr(), None, st)
            gpu.cuda.synchronize()
            assert np.array_equal(binding.shard_deinterleave_map_host(gathered.cpu().numpy(), w, h, tr, slot), want), (G, rule)
            load = np.bincount(slot // k_max, weights=cost.astype(np.float64), minlength=G)
            assert load.max() <= np.bincount(np.arange(T) % G, weights=cost.astype(np.float64), minlength=G).max()
    monkeypatch.delenv("SKR_SHARD", raising=False)


def test_sphere_tests_are_counted_like_the_reference_runs_them(gpu, oracle):
    """bench.py's FP32-VALU roofline is priced on the ray-sphere tests the reference's loops execute — every sphere for a
    radiance ray (raytrace.h:152-165), up to and including the first occluder for a shadow ray (utils.h:52-55).  The
    kernels count them (skr_renderer_read_work); the oracle counts them in its own loops."""
    for scn, w, h, kw, env in (("spheres2.scn", 240, 135, dict(gillum=16, shadow=True, seed=20261004), {}),
                               ("spheres2.scn", 160, 90, dict(jsample=3, shadow=True, seed=2), {}),
                               ("bear.scn", 96, 72, dict(gillum=8, shadow=True, depth=4, seed=9), {}),
                               ("spheres1.scn", 200, 113, dict(gillum=5, shadow=True, depth=2, seed=1), {})):
        r = renderer(scn)
        r.work(reset=True)
        r.render(skr.Options(w, h, **kw))
        gpu.cuda.synchronize()
        got = r.work()
        _, _, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, **kw)
        assert (got["radiance_rays"], got["sphere_hits"], got["shadow_rays"], got["sphere_tests"]) == tuple(int(v) for v in st[:4]), (scn, kw, r.kernel_variant())

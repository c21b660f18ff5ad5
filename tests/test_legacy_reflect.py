"""--legacy-reflect: the reflection / refraction / Fresnel code behind the early `return total_colour;` of
raytrace.h:44 (raytrace.h:45-103, blinn_phong.h:137-184).

PARITY UNPINNED against the reference's code: those lines are unreachable at HEAD, so nothing HEAD writes covers them.  What
there is: (1) the oracle restates them literally and the GPU is held to the oracle bit for bit; (2) the reference's README
pictures were made when the code still ran — `tests/golden/readme_bp_jsample5_parallel_shadows_quarter.npy.gz` is one of them
(`renders/shadows/sample_pngs/bp_jsample5_parallel_shadows.png`, a 1919x1003 screenshot of the 1920x1080 window, box-filtered 4x):
the mode has to look like it, and clearly more so than HEAD's output does — SURVEY's "visual check", as a number.
"""
import gzip
import io
import os
import subprocess
import sys

import numpy as np
import pytest

import skele_raytracer_amd as skr
from conftest import GOLD, ROOT, read_ppm_bytes, scene_path


def readme_picture():
    with gzip.open(os.path.join(GOLD, "readme_bp_jsample5_parallel_shadows_quarter.npy.gz"), "rb") as f:
        return np.load(io.BytesIO(f.read())).astype(np.float32)  # [250, 479, 3]


def likeness(rgb_quarter, ref):
    """Best (mean |difference| / 255-scale, correlation) of a 480x270 frame against the screenshot over the crop offsets."""
    best = None
    for dy in range(0, rgb_quarter.shape[0] - ref.shape[0] + 1):
        for dx in (0, 1):
            a = rgb_quarter[dy:dy + ref.shape[0], dx:dx + ref.shape[1]].astype(np.float32)
            mad = float(np.abs(a - ref).mean())
            if best is None or mad < best[0]:
                best = (mad, float(np.corrcoef(a.reshape(-1), ref.reshape(-1))[0, 1]))
    return best


def test_the_mode_is_what_the_readme_pictures_show(oracle):
    ref = readme_picture()
    scn = scene_path("spheres2.scn")
    legacy, _, _ = oracle.render(scn, 480, 270, shadow=True, legacy_reflect=True)
    head, _, _ = oracle.render(scn, 480, 270, shadow=True)
    mad_l, cc_l = likeness(legacy, ref)
    mad_h, cc_h = likeness(head, ref)
    assert cc_l > 0.98 and mad_l < 5.5, (mad_l, cc_l)          # measured: 4.94, 0.987
    assert cc_h < 0.96 and mad_h > mad_l + 0.5, (mad_h, cc_h)  # HEAD's own output: 5.93, 0.952


def test_leaf_functions_equal_the_reference_own(oracle):
    """FUNCTION-LEVEL PIN (round 3).  The code of raytrace.h:45-103 is unreachable, but the functions it calls are ordinary ones:
    oracle/_ref/ref_render --eval-legacy ran the reference's own bp::fresnel (blinn_phong.h:156), bp::refraction (:143) and
    bp::reflect_direction (:137) on 10 000 (direction, normal, ior) triples (tests/golden/ref_legacy_eval.npy.gz); the oracle's
    restatements must return the same bits.  (The composition — which rays are spawned, in which order they are summed — is
    pinned by the four ref_*_legacy_* goldens of tests/test_oracle_golden.py: ref_driver.cpp --legacy restates only the control
    flow and calls the reference for every value.)"""
    import ctypes as C
    import gzip
    import io
    import os
    from conftest import GOLD
    words = np.load(io.BytesIO(gzip.open(os.path.join(GOLD, "ref_legacy_eval.npy.gz")).read()))
    assert words.shape == (10000, 14) and words.dtype == np.uint32
    f = words.view(np.float32)
    L = oracle.lib()
    L.sko_legacy_eval.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]
    L.sko_legacy_eval.restype = None
    out = (C.c_float * 7)()
    got = np.zeros((len(f), 7), np.float32)
    for i, row in enumerate(f):
        L.sko_legacy_eval((C.c_float * 3)(*row[0:3].tolist()), (C.c_float * 3)(*row[3:6].tolist()), float(row[6]), out)
        got[i] = list(out)
    want = words[:, 7:]
    same = (got.view(np.uint32) == want) | (np.isnan(got) & np.isnan(f[:, 7:]))
    assert same.all(), "first mismatch at triple %d" % int(np.argwhere(~same)[0][0])
    # the sample covers the branches: total internal reflection (fresnel == 1 / a zero refraction direction), both signs of cos
    assert (f[:, 7] == 1.0).sum() > 50 and ((f[:, 8:11] == 0).all(axis=1)).sum() > 50 and (f[:, 7] < 0.05).sum() > 50


def test_depth_one_has_nothing_to_add(oracle):
    """shade(depth - 1 = 0) is (0,0,0): fr * 0 and (1 - fr) * ks * 0 add +0 to the direct term."""
    scn = scene_path("spheres2.scn")
    a, af, _ = oracle.render(scn, 96, 54, depth=1, shadow=True, legacy_reflect=True, want_float=True)
    b, bf, _ = oracle.render(scn, 96, 54, depth=1, shadow=True, want_float=True)
    assert (af.view(np.uint32) == bf.view(np.uint32)).all() and (a == b).all()


# ------------------------------------------------------------------------------------------------ on the device ----

@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


CASES = [
    ("spheres2_d2", "spheres2.scn", 160, 90, dict(depth=2)),
    ("spheres2_shadow", "spheres2.scn", 160, 90, dict(shadow=True)),                       # --depth 3: each hit 2 lights x (refraction + reflection)
    ("spheres2_d5", "spheres2.scn", 48, 27, dict(depth=5, shadow=True)),
    ("spheres2_strict_js2", "spheres2.scn", 96, 54, dict(jsample=2, shadow=True, seed=3, strict=True)),  # 4 lights: arity 8
    ("spheres2_gi2_d3", "spheres2.scn", 64, 36, dict(gillum=2, depth=3, shadow=True, seed=5)),           # --gillum children beside them: arity 6
    ("spheres2_gi3_d2", "spheres2.scn", 64, 36, dict(gillum=3, depth=2, seed=6)),
    ("bear", "bear.scn", 96, 72, dict(shadow=True)),
    ("spheres1_d4", "spheres1.scn", 96, 54, dict(depth=4)),
    ("test_mixed", "test.scn", 96, 72, dict(shadow=True)),                                  # triangles still turn their rays black
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,scn,w,h,kw", CASES, ids=[c[0] for c in CASES])
def test_gpu_matches_the_oracle_bit_for_bit(gpu, oracle, name, scn, w, h, kw):
    strict = kw.get("strict", False)
    kw = {k: v for k, v in kw.items() if k != "strict"}
    r = skr.Renderer(skr.parse_scene(scene_path(scn), strict=strict))
    rgb, rgbf = r.render(skr.Options(w, h, legacy_reflect=True, **kw), want_float=True)
    gpu.cuda.synchronize()
    assert r.kernel_variant() == "level_pipeline_g1"
    o_rgb, o_f, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, legacy_reflect=True, strict=strict, **kw)
    g_f = rgbf.cpu().numpy()
    nb = int((g_f.view(np.uint32) != o_f.view(np.uint32)).sum())
    assert nb == 0, "%s: %d float words differ" % (name, nb)
    assert (rgb.cpu().numpy() == o_rgb).all()
    cnt = r.counters()
    assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])
    if kw.get("shadow"):
        assert cnt["shadow_rays"] == int(st[2])
    plain, _ = r.render(skr.Options(w, h, **kw))
    assert r.kernel_variant() != "level_pipeline_g1"
    if kw.get("depth", 3) > 1:
        assert (plain.cpu().numpy() != o_rgb).any()


@pytest.mark.gpu
def test_a_mesh_without_spheres_keeps_the_node_numbering(gpu, oracle):
    """The flag sets the arity of the counter RNG's node ids (N + 2 L) even where no hit can have the legacy terms: a scene of triangle
    surfaces only, under --shade-triangles --gillum.  From the fourth level down the ids differ from arity N's, so the frame does
    (found by tests/fuzz_parity.py as a count mismatch on a scene too dark to show it)."""
    scn, w, h = scene_path("dragon.scn"), 40, 23
    kw = dict(gillum=2, depth=5, shadow=True, seed=9, shade_triangles=True)
    r = skr.Renderer(skr.parse_scene(scn, strict=True))  # (--strict-scn: the scene's one light is directional)
    rgb, rgbf = r.render(skr.Options(w, h, legacy_reflect=True, **kw), want_float=True)
    gpu.cuda.synchronize()
    cnt = r.counters()
    o_rgb, o_f, st = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, legacy_reflect=True, strict=True, **kw)
    assert int((rgbf.cpu().numpy().view(np.uint32) != o_f.view(np.uint32)).sum()) == 0 and (rgb.cpu().numpy() == o_rgb).all()
    assert (cnt["radiance_rays"], cnt["sphere_hits"], cnt["shadow_rays"]) == tuple(int(v) for v in st[:3])
    _, o_plain, st_plain = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, strict=True, **kw)
    assert (o_plain.view(np.uint32) != o_f.view(np.uint32)).any()  # (the numbering matters on this scene)


@pytest.mark.gpu
def test_full_size_frame_against_the_readme_picture(gpu):
    """The picture's own command line as far as it is known (spheres2.scn, 1920x1080, --jsample 5, shadows), on the device."""
    ref = readme_picture()
    r = skr.Renderer(skr.parse_scene(scene_path("spheres2.scn")))
    rgb, _ = r.render(skr.Options(1920, 1080, jsample=5, shadow=True, seed=1, legacy_reflect=True))
    frame = rgb.cpu().numpy()
    quarter = frame.reshape(270, 4, 480, 4, 3).astype(np.float32).mean(axis=(1, 3))
    mad, cc = likeness(quarter, ref)
    assert cc > 0.985 and mad < 5.5, (mad, cc)


@pytest.mark.gpu
def test_modes_and_limits(gpu):
    from oracle import pyoracle as orc
    r = skr.Renderer(skr.parse_scene(scene_path("spheres2.scn")))
    # any positive --depth (main.cpp:318-329; the lane-per-pixel kernel of rounds 1-2 stopped at 6)
    rgb, rgbf = r.render(skr.Options(40, 23, depth=9, shadow=True, legacy_reflect=True), want_float=True)
    gpu.cuda.synchronize()
    o_rgb, o_f, st = orc.render(scene_path("spheres2.scn"), 40, 23, rng=orc.RNG_COUNTER, math=orc.MATH_SHARED, want_float=True, depth=9, shadow=True, legacy_reflect=True)
    assert r.kernel_variant() == "level_pipeline_g1" and (rgbf.cpu().numpy().view(np.uint32) == o_f.view(np.uint32)).all() and r.counters()["radiance_rays"] == int(st[0])
    # no spheres, nothing to reflect off: the flag is a no-op and HEAD's schedule stays
    d = skr.Renderer(skr.parse_scene(scene_path("dragon.scn")))
    a, _ = d.render(skr.Options(64, 48, legacy_reflect=True))
    b, _ = d.render(skr.Options(64, 48))
    gpu.cuda.synchronize()
    assert gpu.equal(a, b) and d.kernel_variant() == "direct_v3"


@pytest.mark.gpu
def test_scene_from_arrays_with_sphere_ior(gpu):
    sc = skr.parse_scene(scene_path("spheres2.scn"))
    spheres, tris, lights = sc.arrays()
    info = sc.info
    ior, cur = [], 1.0
    for ln in open(scene_path("spheres2.scn")):
        tok = ln.split()
        if tok and tok[0] == "material":
            cur = float(tok[14])
        elif tok and tok[0] == "sphere":
            ior.append(cur)
    cam = list(info.camera[:9])
    sc2 = skr.Scene.from_arrays(spheres, tris, lights, cam, tuple(info.background), tuple(info.ambient), sphere_ior=ior)
    opt = skr.Options(96, 54, shadow=True, legacy_reflect=True)
    a, af = skr.Renderer(sc).render(opt, want_float=True)
    b, bf = skr.Renderer(sc2).render(opt, want_float=True)
    gpu.cuda.synchronize()
    assert gpu.equal(af.view(gpu.int32), bf.view(gpu.int32))
    sc3 = skr.Scene.from_arrays(spheres, tris, lights, cam, tuple(info.background), tuple(info.ambient))  # every ior = 1 (material.h:16)
    c, _ = skr.Renderer(sc3).render(opt, want_float=True)
    assert not gpu.equal(c, a)


@pytest.mark.gpu
def test_both_command_lines_take_the_flag(gpu, oracle, tmp_path):
    w, h = 96, 54
    args = ["--path", scene_path("spheres2.scn"), "--width", str(w), "--height", str(h), "--shadow", "--legacy-reflect"]
    o_rgb, _, _ = oracle.render(scene_path("spheres2.scn"), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, shadow=True, legacy_reflect=True)
    out1, out2 = str(tmp_path / "native.ppm"), str(tmp_path / "py.ppm")
    subprocess.run([os.path.join(ROOT, "bin", "raytracer"), *args, "--output", out1, "--quiet"], check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "-m", "skele_raytracer_amd.render_cli", *args, "--output", out2], check=True, cwd=ROOT, stdout=subprocess.DEVNULL)
    for out in (out1, out2):
        assert (read_ppm_bytes(open(out, "rb").read()) == o_rgb).all(), out

"""--shade-triangles: triangles as surfaces instead of HEAD's black holes (raytrace.h:221-224).

PARITY UNPINNED against the reference: the reference has no such mode, so nothing it holds or can write covers it.  The
mode's rules are stated in include/skr.h (skr_options.shade_triangles) and oracle/skr_oracle.h; these tests hold the GPU
to the oracle's restatement of those rules, bit for bit, and check the properties the rules imply.  Without the flag
every other test in this directory keeps the default (black) behaviour pinned to the reference's own output.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import skele_raytracer_amd as skr
from conftest import ROOT, read_ppm_bytes, scene_path
from test_gpu_parity import compare, gpu, gpu_render  # noqa: F401  (gpu: fixture)

pytestmark = pytest.mark.gpu


CASES = [
    ("test_primary", "test.scn", 160, 120, dict()),
    ("test_shadow_js2", "test.scn", 96, 72, dict(jsample=2, shadow=True, seed=4)),
    ("test_gi4_shadow", "test.scn", 80, 60, dict(gillum=4, shadow=True, seed=3)),
    ("test_gi3_d4", "test.scn", 48, 36, dict(gillum=3, depth=4, shadow=True, seed=9)),
    ("test_strict_gi2_d2", "test.scn", 64, 48, dict(gillum=2, depth=2, seed=5, strict=True)),
    ("spheres1_gi5", "spheres1.scn", 96, 54, dict(gillum=5, shadow=True, seed=6)),
    ("dragon_strict", "dragon.scn", 160, 120, dict(strict=True)),                      # no spheres: its one directional light shades the mesh
    ("dragon_strict_gi2_d3", "dragon.scn", 64, 48, dict(gillum=2, depth=3, seed=2, strict=True)),  # and a triangle hit recurses with no sphere in the scene
    ("dragon_default_lights", "dragon.scn", 96, 72, dict()),                           # HEAD's loader: no light at all, ambient only
]


@pytest.mark.parametrize("name,scn,w,h,kw", CASES, ids=[c[0] for c in CASES])
def test_shaded_triangles_match_the_oracle_bit_for_bit(gpu, oracle, name, scn, w, h, kw):
    g_rgb, g_f, cnt = gpu_render(scn, w, h, shade_triangles=True, **kw)
    assert skr.Renderer.kernel_variant() == "level_pipeline_g1"
    o_rgb, o_f, st = oracle.render(scene_path(scn), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, shade_triangles=True, **kw)
    compare(g_rgb, g_f, o_rgb, o_f, name)
    assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])
    if kw.get("shadow"):
        assert cnt["shadow_rays"] == int(st[2])
    # and the flag does what it says: where the default frame is black because a triangle won, this one is lit
    d_rgb, _, _ = gpu_render(scn, w, h, **kw)
    if scn != "spheres1.scn":  # (its two triangles lie where no ray of this camera goes)
        assert (d_rgb != g_rgb).any()


def _write_mesh_scene(path, rng, n_tris, n_spheres):
    """Spheres on a floor and a cloud of triangles, every few of them under their own material; two coincident pairs."""
    lines = ["camera 0 1.5 -9 0 -.05 1 0 1 0 30", "background .1 .2 .3", "ambient_light .3 .3 .3",
             "material .6 .6 .6 .7 .7 .7 .2 .2 .2 8 0 0 0 1", "sphere 0 -40 0 40"]
    for _ in range(n_spheres):
        lines.append("material %g %g %g %g %g %g %g %g %g %d 0 0 0 1" % (*rng.random(3), *rng.random(3), *(rng.random(3) * .5), int(rng.choice([1, 2, 9, 40]))))
        lines.append("sphere %g %g %g %g" % (rng.uniform(-5, 5), rng.uniform(.3, 3), rng.uniform(-2, 7), rng.uniform(.4, 1.1)))
    nv = 0
    tri_lines = []
    for i in range(n_tris):
        c = np.array([rng.uniform(-5, 5), rng.uniform(.2, 4.5), rng.uniform(-1, 9)])
        size = 10.0 ** rng.uniform(-1.2, .4)
        for v in (c, c + rng.normal(size=3) * size, c + rng.normal(size=3) * size):
            lines.append("vertex %.9g %.9g %.9g" % tuple(v))
        if i % 3 == 0:
            tri_lines.append("material %g %g %g %g %g %g %g %g %g %g 0 0 0 1" % (*rng.random(3), *rng.random(3), *(rng.random(3) * .6), float(rng.choice([1, 2.5, 7, 30]))))
        tri_lines.append("triangle %d %d %d" % (nv, nv + 1, nv + 2))
        if i in (4, 11):  # the same triangle again under another material: equal t, the earlier line has to win
            tri_lines.append("material .9 .1 .1 .9 .1 .1 .9 .9 .9 3 0 0 0 1")
            tri_lines.append("triangle %d %d %d" % (nv, nv + 1, nv + 2))
        nv += 3
    lines += tri_lines
    lines += ["point_light 30 30 30 6 8 -6", "point_light 10 25 40 -7 5 2"]
    open(path, "w").write("\n".join(lines) + "\n")


@pytest.mark.parametrize("n_tris,n_spheres,kw", [
    (60, 4, dict(gillum=4, depth=3, shadow=True, seed=31)),
    (300, 2, dict(jsample=2, shadow=True, seed=32)),
    (25, 0, dict(gillum=3, depth=4, seed=33)),          # floor sphere only
    (120, 6, dict(gillum=2, depth=6, shadow=True, seed=34)),
], ids=["mesh60_gi4_d3", "mesh300_aa", "mesh25_d4", "mesh120_d6"])
def test_random_meshes_with_their_own_materials(gpu, oracle, tmp_path, n_tris, n_spheres, kw):
    scn = str(tmp_path / "mesh.scn")
    _write_mesh_scene(scn, np.random.default_rng(n_tris), n_tris, n_spheres)
    w, h = 96, 54
    r = skr.Renderer(skr.parse_scene(scn))
    rgb, rgbf = r.render(skr.Options(w, h, shade_triangles=True, **kw), want_float=True)
    gpu.cuda.synchronize()
    o_rgb, o_f, st = oracle.render(scn, w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, shade_triangles=True, **kw)
    compare(rgb.cpu().numpy(), rgbf.cpu().numpy(), o_rgb, o_f, "mesh %d" % n_tris)
    cnt = r.counters()
    assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])


def test_culling_data_changes_nothing_for_the_closest_hit(gpu, tmp_path, monkeypatch):
    """The walk that finds the closest triangle uses the same conservative spheres as the any-hit walk; with them
    switched off (SKR_NO_CULL: every triangle tested) the frame has to be the same one."""
    kw = dict(gillum=2, depth=2, seed=3, strict=True, shade_triangles=True)
    a, af, _ = gpu_render("dragon.scn", 128, 96, **kw)
    monkeypatch.setenv("SKR_NO_CULL", "1")
    b, bf, _ = gpu_render("dragon.scn", 128, 96, **kw)
    monkeypatch.delenv("SKR_NO_CULL")
    assert (af.view(np.uint32) == bf.view(np.uint32)).all() and (a == b).all()
    monkeypatch.setenv("SKR_NO_CONES", "1")
    c, cf, _ = gpu_render("dragon.scn", 128, 96, **kw)
    assert (af.view(np.uint32) == cf.view(np.uint32)).all()


def test_scene_from_arrays_with_triangle_materials(gpu, tmp_path):
    scn = str(tmp_path / "mesh.scn")
    _write_mesh_scene(scn, np.random.default_rng(5), 40, 3)
    sc = skr.parse_scene(scn)
    spheres, tris, lights = sc.arrays()
    info = sc.info
    # the materials as the file gives them: walk the lines the way scene.cpp:110-137 does
    mats, cur = [], [0.0] * 9 + [1.0]
    for ln in open(scn):
        tok = ln.split()
        if tok and tok[0] == "material":
            cur = [float(x) for x in tok[1:11]]
        elif tok and tok[0] == "triangle":
            mats.append(cur)
    cam = list(info.camera[:9])
    sc2 = skr.Scene.from_arrays(spheres, tris, lights, cam, tuple(info.background), tuple(info.ambient), triangle_materials=np.array(mats, np.float32))
    opt = skr.Options(96, 54, gillum=3, shadow=True, seed=7, shade_triangles=True)
    a, af = skr.Renderer(sc).render(opt, want_float=True)
    b, bf = skr.Renderer(sc2).render(opt, want_float=True)
    gpu.cuda.synchronize()
    assert (af.cpu().numpy().view(np.uint32) == bf.cpu().numpy().view(np.uint32)).all()
    # without materials the triangles take material.h:9-17's defaults (all-zero colours): a different frame
    sc3 = skr.Scene.from_arrays(spheres, tris, lights, cam, tuple(info.background), tuple(info.ambient))
    c, _ = skr.Renderer(sc3).render(opt, want_float=True)
    assert (c.cpu().numpy() != a.cpu().numpy()).any()


def test_any_depth_and_both_modes_together(gpu, oracle):
    """main.cpp:318-329 takes any positive --depth: so does the mode (the lane-per-pixel kernel of rounds 1-2 stopped at 6), and it
    combines with --legacy-reflect (a sphere hit has the terms of raytrace.h:45-103, a triangle hit does not)."""
    for kw in (dict(gillum=2, depth=8, shadow=True, seed=3), dict(gillum=2, depth=3, shadow=True, seed=5, legacy_reflect=True), dict(depth=4, legacy_reflect=True)):
        g_rgb, g_f, cnt = gpu_render("test.scn", 48, 27, shade_triangles=True, **kw)
        assert skr.Renderer.kernel_variant() == "level_pipeline_g1"
        o_rgb, o_f, st = oracle.render(scene_path("test.scn"), 48, 27, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, want_float=True, shade_triangles=True, **kw)
        compare(g_rgb, g_f, o_rgb, o_f, str(kw))
        assert cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1])
    # a scene without triangles is not concerned: the flag is a no-op there and the node pipeline stays
    r2 = skr.Renderer(skr.parse_scene(scene_path("spheres2.scn")))
    a, _ = r2.render(skr.Options(32, 18, gillum=2, depth=7, seed=3, shade_triangles=True))
    assert skr.Renderer.kernel_variant().startswith("node_levels_v5")
    b, _ = r2.render(skr.Options(32, 18, gillum=2, depth=7, seed=3))
    gpu.cuda.synchronize()
    assert (a.cpu().numpy() == b.cpu().numpy()).all()


def test_both_command_lines_take_the_flag(gpu, oracle, tmp_path):
    w, h = 96, 72
    args = ["--path", scene_path("test.scn"), "--width", str(w), "--height", str(h), "--gillum", "2", "--shadow", "--seed", "5", "--shade-triangles"]
    o_rgb, _, _ = oracle.render(scene_path("test.scn"), w, h, rng=oracle.RNG_COUNTER, math=oracle.MATH_SHARED, gillum=2, shadow=True, seed=5, shade_triangles=True)
    out1, out2 = str(tmp_path / "native.ppm"), str(tmp_path / "py.ppm")
    exe = os.path.join(ROOT, "bin", "raytracer")
    subprocess.run([exe, *args, "--output", out1, "--quiet"], check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL)
    subprocess.run([sys.executable, "-m", "skele_raytracer_amd.render_cli", *args, "--output", out2], check=True, cwd=ROOT, stdout=subprocess.DEVNULL)
    for out in (out1, out2):
        assert (read_ppm_bytes(open(out, "rb").read()) == o_rgb).all(), out

"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every
symbol include/skr.h declares, the .scn loader matches the reference's parseScene() dumps
and the oracle's independent loader, options/PPM helpers behave like the reference, and
device entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

import skele_raytracer_amd as skr
from conftest import GOLD, ROOT, manifest, read_ppm_bytes, scene_path


def test_library_exports_every_declared_symbol():
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "skr.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(skr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(skr.EXPORTED_SYMBOLS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", skr.lib_path()], text=True)
    exported = set(re.findall(r" T (skr_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    L = skr.lib()
    for name in declared:
        getattr(L, name)


def test_library_is_gfx950_hip_code_object():
    data = open(skr.lib_path(), "rb").read()
    assert b"gfx950" in data and all(k in data for k in (b"skr_leaf_kernel2", b"skr_direct_kernel", b"skr_gtrace_kernel"))


def _hex(v):
    return " ".join("%08x" % x for x in np.asarray(v, np.float32).view(np.uint32).ravel())


@pytest.mark.parametrize("scn", ["spheres1.scn", "spheres2.scn", "bear.scn", "test.scn", "dragon.scn"])
def test_loader_matches_reference_parseScene_dump(scn):
    with gzip.open(os.path.join(GOLD, manifest()["scene_dumps"][scn]["file"]), "rt") as f:
        lines = f.read().splitlines()
    sc = skr.parse_scene(scene_path(scn))
    info = sc.info
    s, t, l = sc.arrays()
    assert lines[0] == "camera " + _hex(list(info.camera)[:12])
    assert lines[1] == "background " + _hex(list(info.background))
    assert lines[2] == "ambient " + _hex(list(info.ambient))
    assert list(map(int, lines[3].split()[1:])) == [info.n_spheres, info.n_triangles, info.n_point_lights, 0]
    i = 4
    for k in range(info.n_spheres):
        assert lines[i] == "sphere " + _hex(s[k]), k
        i += 1
    for k in range(info.n_point_lights):
        assert lines[i] == "point_light " + _hex(l[k]), k
        i += 1
    want = np.array([[int(x, 16) for x in ln.split()[1:]] for ln in lines[i:]], np.uint32).reshape(-1, 9)
    assert np.array_equal(t.view(np.uint32), want)


@pytest.mark.parametrize("scn", ["spheres1.scn", "spheres2.scn", "bear.scn", "test.scn", "dragon.scn"])
def test_loader_matches_oracle_loader(oracle, scn):
    sc = skr.parse_scene(scene_path(scn))
    osc = oracle.OracleScene(scene_path(scn))
    o, info = osc.s, sc.info
    assert (o.n_spheres, o.n_triangles, o.n_point_lights, o.n_vertices) == (info.n_spheres, info.n_triangles, info.n_point_lights, info.n_vertices)
    assert (o.n_directional_dropped, o.n_fog_skipped, o.n_unknown, o.n_bad_triangles) == (info.n_directional_dropped, info.n_fog_skipped, info.n_unknown, info.n_bad_triangles)
    assert (o.film_w, o.film_h, o.max_depth_parsed) == (info.film_width, info.film_height, info.max_depth_parsed)


def test_loader_edge_cases(tmp_path):
    p = tmp_path / "edge.scn"
    p.write_text("# comment\n\n   \nsphere 1 2 3\n #notacomment 1\nmaterial 1 1 1 .5 .5 .5\nsphere 0 0 5 1\r\n"
                 "vertex 0 0 0\nvertex 1 0 0\nvertex 0 1 0\ntriangle 0 1 2\ntriangle 0 1 7\ntriangle 2.9 1.2 0\n"
                 "ambient_light .1 .1 .1\nambient_light .2 .2 .2\nbackground 1 0 0\nbackground 0 1 0\n"
                 "directional_light 1 1 1 0 -1 0\nspherical_fog 0 0 0 1 1 1 1 .5\nfilm_resolution 320 200\nmax_depth 7\n")
    sc = skr.parse_scene(str(p))
    i = sc.info
    s, t, l = sc.arrays()
    assert (i.n_spheres, i.n_triangles, i.n_bad_triangles, i.n_unknown) == (2, 2, 1, 1)
    assert s[0].tolist() == [1, 2, 3, 0] + [0] * 9 + [1]          # missing radius stays 0, default material power 1
    assert s[1].tolist()[:4] == [0, 0, 5, 1] and s[1].tolist()[4:10] == [1, 1, 1, .5, .5, .5]
    assert t[1].tolist() == [0, 1, 0, 1, 0, 0, 0, 0, 0]           # float indices truncate: 2.9->2, 1.2->1
    assert np.allclose(list(i.ambient), [0.3] * 3) and list(i.background) == [0, 1, 0]
    assert (i.n_directional_dropped, i.n_fog_skipped, i.film_width, i.film_height, i.max_depth_parsed) == (1, 1, 320, 200, 7)
    with pytest.raises(skr.SkrError, match="Can't open file"):
        skr.parse_scene(str(tmp_path / "missing.scn"))
    empty = tmp_path / "empty.scn"
    empty.write_text("")
    e = skr.parse_scene(str(empty)).info
    assert (e.n_spheres, e.n_triangles, e.n_point_lights) == (0, 0, 0) and list(e.camera) == [0.0] * 13


def test_options_defaults_and_ray_count():
    o = skr.Options()
    c = o.c
    assert (c.width, c.height, c.fov, c.monte_carlo, c.num_path_traces, c.grid_size, c.max_depth, c.use_shadows) == (1920, 1080, 60.0, 0, 1, 0, 3, 0)
    # SURVEY.md §8d configs
    assert skr.radiance_ray_count(skr.Options(640, 360, depth=1)) == 230400
    assert skr.radiance_ray_count(skr.Options(1920, 1080, jsample=5, shadow=True)) == 51840000
    assert skr.radiance_ray_count(skr.Options(1920, 1080, gillum=16, shadow=True)) == 566092800
    assert skr.radiance_ray_count(skr.Options(3840, 2160, gillum=64, jsample=5)) == 8294400 * 25 * 4161


def test_ppm_writer_byte_layout(tmp_path):
    rgb = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    p = str(tmp_path / "o.ppm")
    skr.write_ppm(p, rgb)
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n7 5\n255\n") and len(data) == 11 + 105
    assert np.array_equal(read_ppm_bytes(data), rgb)


def test_no_device_is_a_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    sc = skr.parse_scene(scene_path("spheres1.scn"))
    with pytest.raises(skr.SkrError, match="no CPU fallback"):
        skr.Renderer(sc)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package, the CLI or include/ may mention it."""
    bad = []
    for base in ("skele_raytracer_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".h", ".hip")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"liboracle|pyoracle|skr_oracle|oracle/|sko_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_code_is_clean_under_asan_and_ubsan():
    """The loader (on the reference scenes, malformed and random input), the culling-tree builder and the getters,
    built with -fsanitize=address,undefined (CPU build; tools/sanitize_host.sh)."""
    res = subprocess.run(["sh", os.path.join(ROOT, "tools", "sanitize_host.sh")], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "sanitize_host: clean" in res.stdout, res.stdout + res.stderr


def test_strict_scn_keeps_the_directional_lights(oracle):
    """--strict-scn (SKR_SCN_STRICT): the two directional lights of spheres2.scn that scene.cpp:139-163 builds and drops are
    kept, colour clamped to <= 1, behind the point lights; the default parse is untouched.  Product loader == oracle loader."""
    import skele_raytracer_amd as skr
    d = skr.parse_scene(scene_path("spheres2.scn"))
    s = skr.parse_scene(scene_path("spheres2.scn"), strict=True)
    assert (d.info.n_directional_dropped, d.info.n_directional_lights, d.info.n_point_lights) == (2, 0, 2)
    assert (s.info.n_directional_dropped, s.info.n_directional_lights, s.info.n_point_lights) == (0, 2, 2)
    for a, b in zip(d.arrays(), s.arrays()):
        assert np.array_equal(a, b)
    o = oracle.OracleScene(scene_path("spheres2.scn"), strict=True)
    assert o.s.n_directional_lights == 2 and oracle.OracleScene(scene_path("spheres2.scn")).s.n_directional_lights == 0
    got = [(tuple(np.float32(v) for v in (l.direction.x, l.direction.y, l.direction.z)), tuple(np.float32(v) for v in (l.colour.x, l.colour.y, l.colour.z)))
           for l in (o.s.directional_lights[i] for i in range(2))]
    assert got == [((-1, -1, 1), (np.float32(.8), np.float32(.1), np.float32(.1))), ((0, -1, 0), (1, 0, 0))]
    assert (s.info.film_width, s.info.film_height) == (d.info.film_width, d.info.film_height)

"""Pins the oracle: the C restatement in (glibc-replay RNG, libm) mode must be
byte-identical to PPMs produced by the reference's own shade()/parseScene()
(tests/golden/ref_*.ppm.gz, made by tests/golden/make_golden.py with oracle/_ref) and
to the reference's own fixture renders/testcpu.ppm.  CPU only."""
import gzip
import os

import numpy as np
import pytest

from conftest import GOLD, args_to_kwargs, manifest, read_golden_ppm, scene_path

CASES = sorted(manifest()["cases"].items())


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_oracle_replay_matches_reference_ppm(oracle, name, case):
    kw = args_to_kwargs(case["args"])
    w, h = kw.pop("width"), kw.pop("height")
    gold = read_golden_ppm(case["file"])
    assert gold.shape == (h, w, 3)
    rgb, _, stats = oracle.render(scene_path(case["scene"]), w, h, rng=oracle.RNG_GLIBC_REPLAY,
                                  math=oracle.MATH_LIBM, **kw)
    ndiff = int((rgb != gold).sum())
    assert ndiff == 0, "%s: %d differing bytes vs the reference's output" % (name, ndiff)
    assert stats[0] >= w * h  # at least one radiance ray per pixel


def test_reference_own_fixture_testcpu(oracle):
    """renders/testcpu.ppm = HEAD `dragon.scn --parallel true` (640x480, depth 1): the only
    pixel-exact fixture the reference itself holds."""
    gold = read_golden_ppm("testcpu.ppm.gz")
    assert gold.shape == (480, 640, 3)
    rgb, _, _ = oracle.render(scene_path("dragon.scn"), 640, 480, depth=1, rng=oracle.RNG_GLIBC_REPLAY,
                              math=oracle.MATH_LIBM)
    assert np.array_equal(rgb, gold)
    # and the ref_render harness reproduced it too when the goldens were made
    assert np.array_equal(read_golden_ppm("ref_dragon_parallel_entry.ppm.gz"), gold)
    cols, counts = np.unique(gold.reshape(-1, 3), axis=0, return_counts=True)
    assert sorted(counts.tolist()) == [38301, 268899]


@pytest.mark.parametrize("name", ["spheres2_gi16_shadow", "spheres1_gi8_noshadow", "spheres2_shadow", "bear_shadow"])
def test_shared_math_stays_within_tolerance_of_libm(oracle, name):
    """The shared sincos/pow recipes (the ones the HIP kernel uses) replace libm's
    cosf/sinf/powf; on the reference's own RNG stream the image must stay within 1/255."""
    case = manifest()["cases"][name]
    kw = args_to_kwargs(case["args"])
    w, h = kw.pop("width"), kw.pop("height")
    gold = read_golden_ppm(case["file"]).astype(np.int32)
    rgb, _, _ = oracle.render(scene_path(case["scene"]), w, h, rng=oracle.RNG_GLIBC_REPLAY,
                              math=oracle.MATH_SHARED, **kw)
    d = np.abs(rgb.astype(np.int32) - gold)
    assert d.max() <= 1
    assert (d > 0).mean() < 1e-3


def _dump_lines(scn):
    with gzip.open(os.path.join(GOLD, manifest()["scene_dumps"][scn]["file"]), "rt") as f:
        return f.read().splitlines()


def _hex(v):
    return " ".join("%08x" % np.float32(x).view(np.uint32) for x in v)


@pytest.mark.parametrize("scn", ["spheres1.scn", "spheres2.scn", "bear.scn", "test.scn", "dragon.scn"])
def test_oracle_loader_matches_parseScene_dump(oracle, scn):
    """Field-by-field (hex floats) against what the reference's parseScene() produced."""
    lines = _dump_lines(scn)
    sc = oracle.OracleScene(scene_path(scn))  # keep alive: __del__ frees the arrays
    s = sc.s
    v = lambda p: (p.x, p.y, p.z)
    assert lines[0] == "camera " + _hex(v(s.cam_pos) + v(s.cam_dir) + v(s.cam_up) + v(s.cam_right))
    assert lines[1] == "background " + _hex(v(s.background))
    assert lines[2] == "ambient " + _hex(v(s.ambient))
    counts = list(map(int, lines[3].split()[1:]))
    assert counts == [s.n_spheres, s.n_triangles, s.n_point_lights, 0]  # directional lights never pushed
    i = 4
    for k in range(s.n_spheres):
        sp = s.spheres[k]
        assert lines[i] == "sphere " + _hex(v(sp.center) + (sp.radius,) + v(sp.ambient) + v(sp.diffuse) + v(sp.specular) + (sp.power,))
        i += 1
    for k in range(s.n_point_lights):
        pl = s.point_lights[k]
        assert lines[i] == "point_light " + _hex(v(pl.position) + v(pl.colour))
        i += 1
    tri = np.ctypeslib.as_array(s.triangles, shape=(max(s.n_triangles, 1),)).view(np.uint32).reshape(-1, 9)[:s.n_triangles] if s.n_triangles else np.zeros((0, 9), np.uint32)
    want = np.array([[int(t, 16) for t in l.split()[1:]] for l in lines[i:i + s.n_triangles]], np.uint32).reshape(-1, 9)
    assert np.array_equal(tri, want)
    assert i + s.n_triangles == len(lines)


def test_loader_quirks(oracle):
    h2 = oracle.OracleScene(scene_path("spheres2.scn"))
    s2 = h2.s
    assert s2.n_directional_dropped == 2 and s2.n_fog_skipped == 1 and s2.n_point_lights == 2 and s2.n_spheres == 15
    ht = oracle.OracleScene(scene_path("test.scn"))
    t = ht.s
    assert t.n_unknown >= 3  # max_vertices, max_normals, spot_light, normal ...
    hd = oracle.OracleScene(scene_path("dragon.scn"))
    d = hd.s
    assert (d.n_spheres, d.n_triangles, d.n_point_lights) == (0, 10002, 0)

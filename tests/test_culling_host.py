"""Host logic of the triangle culling hierarchy (csrc/scene_host.cpp build_triangle_chunks) — no GPU.

The device walk skips a node of the chunk tree (and every triangle below it) when `|e x d|^2 > R^2 |d|^2` for its sphere.  These tests restate both
sides in numpy binary32 with the device's operation order — the acceptance test of utils.h:181-213
(device_math.h triangle_hit) and the sphere test of shade_common.h any_triangle_closer — and check, on rays aimed
at the borders of the accept regions, that no accepted (ray, triangle) pair is ever hidden by its spheres.
"""
import numpy as np
import pytest

import skele_raytracer_amd as skr
from conftest import scene_path
from scenegen import write_random_mesh_scene

f32 = np.float32


def _dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross(x, y):  # glm operand order, as device_math.h cross3
    return np.stack([x[..., 1] * y[..., 2] - y[..., 1] * x[..., 2],
                     x[..., 2] * y[..., 0] - y[..., 2] * x[..., 0],
                     x[..., 0] * y[..., 1] - y[..., 0] * x[..., 1]], axis=-1)


def triangle_accepts(o, d, v0, e1, e2):
    """utils.h:181-213 in binary32, vectorised (all arrays float32, shape [n,3])."""
    with np.errstate(all="ignore"):
        p = _cross(d, e2)
        det = _dot(e1, p)
        ok = ~(np.abs(det) < f32(0.00001))
        inv = f32(1.0) / det
        tv = o - v0
        u = inv * _dot(-tv, p)
        ok &= ~((u < 0) | (u > 1))
        q = _cross(tv, e1)
        v = _dot(d, q) * inv
        ok &= ~((v < 0) | (u + v > 1))
    return ok


def sphere_culls(o, d, ent, cones=True):
    """shade_common.h line_touches, negated: skip iff dot(cr,cr) > R2 * dot(d,d), cr = cross(centre - o, d), with
    R2 = R_tight^2 for rays with (d . axis/kappa)^2 >= d . d and R^2 otherwise; NaN never culls."""
    with np.errstate(all="ignore"):
        e = ent[..., :3] - o
        cr = _cross(e, d)
        dd = _dot(d, d)
        R2 = ent[..., 3]
        if cones:
            gb = _dot(d, ent[..., 4:7])
            R2 = np.where(gb * gb >= dd, ent[..., 7], ent[..., 3])
        return _dot(cr, cr) > R2 * dd


def _scene(name):
    return skr.parse_scene(scene_path(name))


def tree_parents(links):
    """Parent index of every node of the depth-first, skip-linked tree (root: -1), and the height-1 node of every chunk."""
    n = links.shape[0]
    parent = np.full(n, -1, np.int64)
    stack = []
    for i in range(n):
        while stack and links[stack[-1], 0] <= i:
            stack.pop()
        if stack:
            parent[i] = stack[-1]
        if links[i, 2] == 0:
            stack.append(i)
    leaves = np.nonzero(links[:, 2] > 0)[0]
    n_chunks = int((links[leaves, 1] + links[leaves, 2]).max()) if leaves.size else 0
    node_of_chunk = np.full(n_chunks, -1, np.int64)
    for i in leaves:
        node_of_chunk[links[i, 1]:links[i, 1] + links[i, 2]] = i
    return parent, node_of_chunk


@pytest.mark.parametrize("name", ["dragon.scn", "test.scn", "spheres1.scn"])
def test_device_triangles_are_a_permutation_of_the_file_triangles(name):
    sc = _scene(name)
    _, raw, _ = sc.arrays()
    cs, tris, sph, links, ch = sc.culling()
    n = raw.shape[0]
    want = np.concatenate([raw[:, 0:3], raw[:, 3:6] - raw[:, 0:3], raw[:, 6:9] - raw[:, 0:3]], axis=1)  # utils.h:183-184
    got = tris[:, :, :3].reshape(n, 9)
    # row 1's .w carries the triangle's index in the file (--shade-triangles breaks ties by it): the permutation, exactly
    order = tris[:, 1, 3].copy().view(np.int32)
    assert np.all(tris[:, 0, 3] == 0) and np.all(tris[:, 2, 3] == 0)
    assert np.array_equal(np.sort(order), np.arange(n))
    assert np.array_equal(want[order], got)
    assert cs == (4 if sc.info.n_spheres == 0 else 16)


@pytest.mark.parametrize("name", ["dragon.scn", "test.scn", "spheres1.scn"])
def test_tree_shape(name):
    """Chunks are consecutive runs of chunk_size device triangles; the nodes above them are in depth-first order with
    skip links: every chunk sits under exactly one height-1 node, a node has at most 8 children and its skip link
    closes exactly its subtree."""
    sc = _scene(name)
    cs, tris, sph, links, ch = sc.culling()
    n, nt = links.shape[0], tris.shape[0]
    assert ch.shape[0] == (nt + cs - 1) // cs
    parent, node_of_chunk = tree_parents(links)
    assert np.all(node_of_chunk >= 0) and node_of_chunk.shape[0] == ch.shape[0]
    leaves = np.nonzero(links[:, 2] > 0)[0]
    assert np.array_equal(links[leaves, 1], np.arange(leaves.size) * 8)          # consecutive, in order
    assert np.all(links[leaves[:-1], 2] == 8) and 0 < links[leaves[-1], 2] <= 8
    assert np.all(links[leaves, 0] == leaves + 1) and np.all(links[leaves, 3] == 1)
    assert links[0, 0] == n and parent[0] == -1 and np.all(parent[1:] >= 0)
    inner = np.nonzero(links[:, 2] == 0)[0]
    kids = np.bincount(parent[1:], minlength=n) if n > 1 else np.zeros(n, np.int64)
    assert np.all(kids[inner] >= 1) and np.all(kids[inner] <= 8) and np.all(kids[leaves] == 0)
    assert np.all(links[1:, 3] == links[parent[1:], 3] - 1)                       # heights step by one
    assert np.all(links[1:, 0] <= links[parent[1:], 0])                           # a subtree closes inside its parent's


@pytest.mark.parametrize("name", ["dragon.scn", "test.scn", "spheres1.scn"])
def test_spheres_contain_their_accept_regions_and_children(name):
    sc = _scene(name)
    cs, tris, sph, links, ch = sc.culling()
    parent, node_of_chunk = tree_parents(links)
    t = tris.astype(np.float64)
    v0, e1, e2 = t[:, 0, :3], t[:, 1, :3], t[:, 2, :3]
    corners = np.stack([v0, v0 - e1, v0 + e2], axis=1)  # the mirrored triangle the reference's test accepts
    c = ch[np.arange(t.shape[0]) // cs].astype(np.float64)
    dist = np.linalg.norm(corners - c[:, None, :3], axis=2).max(axis=1)
    finite = np.isfinite(c[:, 3])
    assert np.all(dist[finite] < np.sqrt(c[finite, 3])), "an accept region sticks out of its chunk sphere"
    assert np.all(dist < np.sqrt(c[:, 7])) and np.all(c[:, 7] <= c[:, 3]), "... or out of its tight sphere"

    def inside(child_sph, parent_sph, what):
        for col in (3, 7):  # the general radii, and the tight ones where the parent has a cone
            both = np.isfinite(parent_sph[:, col])
            if col == 7:
                both &= np.any(parent_sph[:, 4:7] != 0, axis=1)
                assert np.all(np.any(child_sph[both, 4:7] != 0, axis=1)), "a %s without a cone under a node with one" % what
            assert np.all(np.isfinite(child_sph[both, col])), "an unbounded %s under a bounded node" % what
            reach = np.linalg.norm(child_sph[both, :3].astype(np.float64) - parent_sph[both, :3].astype(np.float64), axis=1) \
                + np.sqrt(child_sph[both, col].astype(np.float64))
            assert np.all(reach < np.sqrt(parent_sph[both, col].astype(np.float64))), "a %s sticks out of its node" % what

    inside(ch, sph[node_of_chunk], "chunk")
    if sph.shape[0] > 1:
        child = np.arange(1, sph.shape[0])
        inside(sph[child], sph[parent[child]], "node")
    # the Morton order must have made the chunks small: median chunk radius well below the mesh extent
    extent = np.linalg.norm(v0.max(axis=0) - v0.min(axis=0))
    if name == "dragon.scn":
        assert np.median(np.sqrt(ch[np.isfinite(ch[:, 3]), 3])) < 0.12 * extent


BOUNDS = (4.0, 32.0, 256.0)  # csrc/tri_chunks.h SKR_CULL_DMAX_LIST


def _border_rays(rng, sc, tris, n, level):
    """Rays from the places rays can start (camera; points on spheres) through points on or next to the border
    of randomly chosen triangles' accept regions — where rounding decides acceptance.  Camera rays are as long
    as the level's bound allows, rays from spheres (GI children) at most 3 (raytrace.h:123-125)."""
    i = sc.info
    cam = np.array(list(i.camera)[:3], np.float64)
    sph, _, _ = sc.arrays()
    nt = tris.shape[0]
    k = rng.integers(0, nt, n)
    t = tris[k].astype(np.float64)
    v0, e1, e2 = t[:, 0, :3], t[:, 1, :3], t[:, 2, :3]
    a = rng.random(n)
    edge = rng.integers(0, 3, n)
    A, B, Cc = v0, v0 - e1, v0 + e2
    P = np.where((edge == 0)[:, None], A + a[:, None] * (B - A), np.where((edge == 1)[:, None], B + a[:, None] * (Cc - B), Cc + a[:, None] * (A - Cc)))
    centroid = (A + B + Cc) / 3
    # from far inside to slightly outside the border, on a log scale of relative offsets
    off = (10.0 ** rng.uniform(-8, -0.3, n)) * rng.choice([-1.0, 1.0], n)
    P = P + off[:, None] * (centroid - P)
    o = np.repeat(cam[None], n, axis=0)
    from_cam = np.ones(n, bool)
    if sph.shape[0]:
        j = rng.integers(0, sph.shape[0], n)
        dirn = rng.normal(size=(n, 3))
        dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
        on_sphere = sph[j, :3].astype(np.float64) + dirn * np.abs(sph[j, 3:4].astype(np.float64))
        from_cam = rng.random(n) < 0.5
        o = np.where(from_cam[:, None], o, on_sphere)
    # a third of the rays graze their triangle: direction in its plane plus 1e-6 .. 1e-2 of the normal, from a point
    # no farther than the scene's ray origins (the bounds only use the distance) — this is where |det| sits near
    # 1e-5, the reference's u, v are noise and the cone test must fall back to the general radius
    m = np.cross(e1, e2)
    area2 = np.linalg.norm(m, axis=1, keepdims=True)
    nrm = m / np.where(area2 > 0, area2, 1)
    inplane = np.cross(nrm, rng.normal(size=(n, 3)))
    inplane /= np.maximum(np.linalg.norm(inplane, axis=1, keepdims=True), 1e-300)
    sinphi = (10.0 ** rng.uniform(-6, -2, (n, 1))) * rng.choice([-1.0, 1.0], (n, 1))
    graze_d = inplane * np.sqrt(1 - sinphi ** 2) + nrm * sinphi
    reach = np.minimum(np.linalg.norm(P - cam[None], axis=1, keepdims=True), 15.0)
    graze_o = P - graze_d * rng.uniform(0.05, 1.0, (n, 1)) * reach
    graze = (rng.random(n) < 0.33) & (area2[:, 0] > 0)
    o = np.where(graze[:, None], graze_o, o)
    d = P - o
    ln = np.linalg.norm(d, axis=1, keepdims=True)
    from_cam = from_cam | graze
    longest = np.where(from_cam, 0.98 * BOUNDS[level], 3.0)[:, None]
    scale = np.where(rng.random((n, 1)) < 0.4, 1.0, 10.0 ** (rng.random((n, 1)) * np.log10(longest / 0.5)) * 0.5)
    d = d / ln * scale
    return k, o.astype(f32), d.astype(f32)


def _soup_scene(tmp_path):
    """Small triangles between a camera and a few spheres: a scene whose chunk spheres are bounded at every level."""
    rng = np.random.default_rng(5)
    lines = ["camera 0 1.5 -9 0 -.05 1 0 1 0 30", "material .6 .6 .6 .7 .7 .7 .2 .2 .2 8 0 0 0 1", "sphere 0 -3 4 2",
             "sphere -2 1 1 1", "sphere 2.5 1.2 3 1.2"]
    n = 1500
    for _ in range(n):
        c = np.array([rng.uniform(-6, 6), rng.uniform(-1, 6), rng.uniform(-2, 12)])
        size = 10.0 ** rng.uniform(-3.5, -2.3)
        for v in (c, c + rng.normal(size=3) * size, c + rng.normal(size=3) * size):
            lines.append("vertex %.9g %.9g %.9g" % tuple(v))
    lines += ["triangle %d %d %d" % (3 * i, 3 * i + 1, 3 * i + 2) for i in range(n)]
    path = str(tmp_path / "soup.scn")
    open(path, "w").write("\n".join(lines) + "\n")
    return skr.parse_scene(path)


@pytest.mark.parametrize("name,level", [("dragon.scn", 0), ("dragon.scn", 1), ("dragon.scn", 2), ("soup", 0), ("soup", 2), ("test.scn", 0)])
def test_no_accepted_pair_is_hidden_by_its_spheres(tmp_path, name, level):
    sc = _soup_scene(tmp_path) if name == "soup" else _scene(name)
    n = 300000
    cs, tris, sph, links, ch = sc.culling(level)
    parent, node_of_chunk = tree_parents(links)
    rng = np.random.default_rng(7 + level)
    k, o, d = _border_rays(rng, sc, tris, n, level)
    t = tris[k]
    acc = triangle_accepts(o, d, t[:, 0, :3], t[:, 1, :3], t[:, 2, :3])
    assert 0.05 * n < acc.sum() < 0.98 * n, "the sample must straddle the border (%d of %d accepted)" % (acc.sum(), n)
    chunk = k // cs
    hidden = acc & sphere_culls(o, d, ch[chunk])
    node = node_of_chunk[chunk]
    while np.any(node >= 0):  # the height-1 node and every ancestor up to the root
        live = node >= 0
        hidden[live] |= acc[live] & sphere_culls(o[live], d[live], sph[node[live]])
        node = np.where(live, parent[np.maximum(node, 0)], -1)
    assert not hidden.any(), "%d accepted (ray, triangle) pairs would have been culled" % hidden.sum()
    other = rng.integers(0, ch.shape[0], n)
    culled = sphere_culls(o, d, ch[other]).mean()
    if name == "test.scn":
        # unit-sized triangles 30-40 units from the ray origins: at grazing incidence the reference's binary32 u, v
        # are off by more than an edge length and no general radius is finite — but the wall is planar, so every
        # entry has a cone and a finite tight radius for the rays that are not grazing it
        assert not np.isfinite(sph[:, 3]).any() and not np.isfinite(ch[:, 3]).any()
        assert np.isfinite(ch[:, 7]).all() and np.isfinite(sph[:, 7]).all() and culled > 0.5
        assert sphere_culls(o, d, ch[other], cones=False).sum() == 0
    elif not (name == "soup" and level == 2):  # at |d| <= 256 the soup's slack outgrows its chunks: mostly unbounded
        assert culled > 0.5, "the spheres must actually cull (%.2f)" % culled


def test_levels_nest():
    """A tighter |d| bound can only shrink a sphere."""
    sc = _scene("dragon.scn")
    for which in (2, 4):  # node spheres, chunk spheres
        r = [sc.culling(level)[which][:, 3] for level in range(3)]
        assert np.all(r[0] <= r[1]) and np.all(r[1] <= r[2]) and np.any(r[0] < r[2])
        tight = [sc.culling(level)[which][:, 4:8] for level in range(3)]
        both = np.any(tight[0][:, :3] != 0, axis=1) & np.any(tight[2][:, :3] != 0, axis=1)
        assert np.array_equal(tight[0][both], tight[2][both])  # cone and tight radius do not depend on the bound on |d|
        # (an entry drops its cone at the levels where the general radius is already the smaller one)
    links = [sc.culling(level)[3] for level in range(3)]
    assert np.array_equal(links[0], links[1]) and np.array_equal(links[1], links[2])  # one topology, three sets of radii


@pytest.mark.parametrize("seed", range(8))
def test_random_mesh_scenes_hide_no_accepted_pair(tmp_path, seed):
    """The scenes of tests/test_gpu_parity.py::test_random_meshes_match_oracle (planar patches of any size and
    orientation, slivers, degenerate triangles, spheres): all three |d| levels, cones included."""
    path = str(tmp_path / "mesh.scn")
    write_random_mesh_scene(path, np.random.default_rng(1000 + seed))
    sc = skr.parse_scene(path)
    for level in range(3):
        cs, tris, sph, links, ch = sc.culling(level)
        parent, node_of_chunk = tree_parents(links)
        n = 60000
        k, o, d = _border_rays(np.random.default_rng(seed * 7 + level), sc, tris, n, level)
        t = tris[k]
        acc = triangle_accepts(o, d, t[:, 0, :3], t[:, 1, :3], t[:, 2, :3])
        assert acc.sum() > 0.05 * n
        chunk = k // cs
        hidden = acc & sphere_culls(o, d, ch[chunk])
        node = node_of_chunk[chunk]
        while np.any(node >= 0):
            live = node >= 0
            hidden[live] |= acc[live] & sphere_culls(o[live], d[live], sph[node[live]])
            node = np.where(live, parent[np.maximum(node, 0)], -1)
        assert not hidden.any(), "level %d: %d accepted (ray, triangle) pairs would have been culled" % (level, hidden.sum())

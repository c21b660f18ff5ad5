"""Scene generators shared by the CPU and GPU tests (inputs in the reference's .scn grammar, scene.cpp:29-218)."""
import numpy as np


def write_random_mesh_scene(path, rng):
    """Planar patches (tessellated quads of random size, orientation and position: their chunks get cones), a few
    loose triangles of any size (slivers and degenerate ones included), 0-3 spheres, 0-2 lights, a random camera."""
    cam = rng.uniform(-1, 1, 3) + np.array([0, 1, -8.0])
    look = np.array([0, 0.5, 4.0]) + rng.uniform(-1, 1, 3) - cam
    look *= rng.choice([1.0, 1.0, 0.3, 2.5]) / np.linalg.norm(look)   # the reference keeps the file's magnitude
    lines = ["camera %.7g %.7g %.7g %.7g %.7g %.7g 0 1 0 30" % (*cam, *look), "background .2 .3 .4", "ambient_light .3 .3 .3"]
    for _ in range(int(rng.integers(0, 4))):
        lines.append("material %g %g %g %g %g %g .3 .3 .3 %d 0 0 0 1" % (*rng.random(3), *rng.random(3), int(rng.choice([1, 4, 16]))))
        lines.append("sphere %g %g %g %g" % (rng.uniform(-4, 4), rng.uniform(-1, 3), rng.uniform(0, 8), rng.uniform(0.4, 1.5)))
    for _ in range(int(rng.integers(0, 3))):
        lines.append("point_light %g %g %g %g %g %g" % (*rng.uniform(5, 30, 3), rng.uniform(-6, 6), rng.uniform(3, 9), rng.uniform(-6, 6)))
    verts, tris = [], []
    for _ in range(int(rng.integers(1, 4))):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        org = np.array([rng.uniform(-4, 4), rng.uniform(-2, 4), rng.uniform(-1, 10)])
        cell = 10.0 ** rng.uniform(-1.5, 0.5)
        nu, nv = int(rng.integers(2, 9)), int(rng.integers(2, 9))
        base = len(verts)
        for i in range(nu + 1):
            for j in range(nv + 1):
                verts.append(org + q[0] * (i - nu / 2) * cell + q[1] * (j - nv / 2) * cell)
        idx = lambda i, j: base + i * (nv + 1) + j
        for i in range(nu):
            for j in range(nv):
                tris += [(idx(i, j), idx(i + 1, j), idx(i, j + 1)), (idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1))]
    for _ in range(int(rng.integers(0, 40))):
        c = np.array([rng.uniform(-5, 5), rng.uniform(-2, 5), rng.uniform(-2, 10)])
        size = 10.0 ** rng.uniform(-3, 0.8)
        a, b = rng.normal(size=3) * size, rng.normal(size=3) * size
        kind = rng.random()
        if kind < 0.15:
            b = a * rng.uniform(0.5, 2) + rng.normal(size=3) * size * 1e-3   # sliver
        elif kind < 0.2:
            b = a * 2.0                                                      # degenerate
        base = len(verts)
        verts += [c, c + a, c + b]
        tris.append((base, base + 1, base + 2))
    lines += ["vertex %.9g %.9g %.9g" % tuple(v) for v in verts]
    lines += ["triangle %d %d %d" % t for t in tris]
    open(path, "w").write("\n".join(lines) + "\n")

#!/usr/bin/env python3
"""Randomised GPU-vs-oracle campaign (development aid; the committed tests hold fixed cases): random sphere / mesh scenes, random
options over every mode the product has (--gillum at random N and depth through both schedules of the node pipeline and through the
general level pipeline, --jsample, --shadow, --strict-scn, --shade-triangles, --legacy-reflect — also together, at any depth —,
--progressive), small frames, bit-for-bit comparison of the float image,
the bytes and the ray / hit / shadow-ray counts.

    python tests/fuzz_parity.py [cases=200] [seed=1]
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import skele_raytracer_amd as skr  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402  (the checker: this is a test tool)
from scenegen import write_random_mesh_scene  # noqa: E402


def write_sphere_scene(path, rng):
    lines = ["camera %g %g %g 0 -.05 1 0 1 0 30" % (rng.uniform(-1, 1), rng.uniform(1, 2), rng.uniform(-10, -7)), "background %g %g %g" % tuple(rng.random(3) * .5),
             "ambient_light .3 .3 .3", "material .6 .6 .6 .7 .7 .7 .2 .2 .2 8 0 0 0 1.3", "sphere 0 -40 0 40"]
    for _ in range(int(rng.integers(1, 40))):
        lines.append("material %g %g %g %g %g %g %g %g %g %g 0 0 0 %g" % (*rng.random(3), *rng.random(3), *(rng.random(3) * rng.choice([0, .5])),
                                                                           float(rng.choice([1, 2, 2.5, 16, 60])), rng.uniform(.8, 1.8)))
        lines.append("sphere %g %g %g %g" % (rng.uniform(-6, 6), rng.uniform(.2, 4), rng.uniform(-3, 8), rng.uniform(.3, 1.3)))
    for _ in range(int(rng.integers(0, 4))):
        lines.append("point_light %g %g %g %g %g %g" % (*rng.uniform(5, 40, 3), rng.uniform(-8, 8), rng.uniform(3, 9), rng.uniform(-8, 8)))
    for _ in range(int(rng.integers(0, 3))):
        lines.append("directional_light %g %g %g %g %g %g" % (*rng.uniform(.1, 1.2, 3), rng.uniform(-1, 1), rng.uniform(-1, -.2), rng.uniform(-1, 1)))
    open(path, "w").write("\n".join(lines) + "\n")


def generate(rng, cases, tmp):
    """The campaign's cases, in order: (index, scene file, width, height, option keywords, strict, passes, environment switches).
    The draws do not depend on any render, so a case can be replayed by its index (FUZZ_ONLY)."""
    for c in range(cases):
        scn = os.path.join(tmp, "s%d.scn" % c)
        mesh = rng.random() < 0.35
        (write_random_mesh_scene if mesh else write_sphere_scene)(scn, rng)
        w, h = int(rng.integers(8, 97)), int(rng.integers(8, 65))
        kw = dict(seed=int(rng.integers(1, 2 ** 40)), shadow=bool(rng.random() < .6), fov=float(rng.choice([30, 60, 90, 140])))
        mode = rng.choice(["gi", "gi", "gi", "plain", "legacy", "surfaces"])
        strict = bool(rng.random() < .3)
        if mode in ("gi", "legacy", "surfaces") and rng.random() < .8:
            n = int(rng.choice([1, 2, 3, 4, 5, 8, 16, 33]))
            d = int(rng.integers(2, 6))
            while n ** (d - 1) * w * h > 3e6:
                d -= 1
            kw.update(gillum=n, depth=max(d, 1))
        if mode == "legacy":
            kw.update(legacy_reflect=True)
            if rng.random() < .3:
                kw.update(shade_triangles=True)
            # (the tree has N + 2 L children per node: keep the frame small)
            while kw.get("depth", 3) > 2 and (kw.get("gillum", 0) + 8) ** (kw.get("depth", 3) - 1) * w * h > 3e6:
                kw["depth"] = kw.get("depth", 3) - 1
        if mode == "surfaces":
            kw.update(shade_triangles=True)
        if rng.random() < .25:
            kw["jsample"] = int(rng.integers(1, 4))
        passes = int(rng.choice([1, 1, 1, 2, 3]))
        env = {}
        if "gillum" in kw and rng.random() < .5:
            env["SKR_FLAT"] = "0"
        if "gillum" in kw and rng.random() < .2:
            env["SKR_LEVELS_BUDGET_MB"] = str(int(rng.choice([8, 16, 64])))
        if mode == "gi" and "gillum" in kw and rng.random() < .25:
            env["SKR_PIPELINE"] = "generic"

        yield c, scn, w, h, kw, strict, passes, env


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    tmp = tempfile.mkdtemp()
    bad = 0
    seen = {}
    ONLY = int(os.environ["FUZZ_ONLY"]) if os.environ.get("FUZZ_ONLY") else None  # replay one case of a campaign
    for c, scn, w, h, kw, strict, passes, env in generate(rng, cases, tmp):
        for k in ("SKR_FLAT", "SKR_LEVELS_BUDGET_MB", "SKR_PIPELINE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        if ONLY is not None and c != ONLY:
            continue
        try:
            r = skr.Renderer(skr.parse_scene(scn, strict=strict))
            rgb, rgbf = r.render(skr.Options(w, h, progressive=passes, **kw), want_float=True)
            torch.cuda.synchronize()
            okw = {k: v for k, v in kw.items()}
            if passes > 1:
                o_rgb, o_f, st, _ = orc.render_progressive(scn, w, h, passes, strict=strict, **okw)
            else:
                o_rgb, o_f, st = orc.render(scn, w, h, want_float=True, strict=strict, **okw)
            cnt = r.counters()
            seen[r.kernel_variant()] = seen.get(r.kernel_variant(), 0) + 1
            same = (rgbf.cpu().numpy().view(np.uint32) == o_f.view(np.uint32)).all() and (rgb.cpu().numpy() == o_rgb).all()
            counts = cnt["radiance_rays"] == int(st[0]) and cnt["sphere_hits"] == int(st[1]) and (not kw.get("shadow") or cnt["shadow_rays"] == int(st[2]))
            if not (same and counts):
                bad += 1
                print("MISMATCH case %d: %s %dx%d %s strict=%s passes=%d env=%s variant=%s same=%s counts=%s device %s oracle %s" % (c, scn, w, h, kw, strict, passes, env, r.kernel_variant(), same, counts, dict(cnt), [int(v) for v in st[:4]]), flush=True)
                if ONLY is not None:
                    print(open(scn).read(), flush=True)
        except skr.SkrError as e:
            print("case %d refused (%s): %s" % (c, kw, str(e)[:100]), flush=True)
        if c % 25 == 24:
            print("%d cases, %d mismatches" % (c + 1, bad), flush=True)
    print("done: %d cases, %d mismatches; kernel variants: %s" % (cases, bad, ", ".join("%s x %d" % kv for kv in sorted(seen.items()))))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

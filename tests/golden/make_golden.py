#!/usr/bin/env python3
"""Generate tests/golden/ from the reference itself (this container only).

Runs oracle/_ref/ref_render — the reference's own shade()/bp::*/parseScene()
compiled unmodified from /root/reference/src by oracle/Makefile — on the
reference's scene files and commits the OUTPUTS (PPMs, parsed-scene dumps) as
fixtures, with the exact command line and seed of each in manifest.json.
Also copies the data inputs the tests need on the GPU box, where
/root/reference does not exist: the .scn scene files (data, not source) and the
reference's one pixel-exact fixture renders/testcpu.ppm.

Usage: python tools/make_golden.py        (needs /root/reference and gcc/g++)
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_render")

# name, scene, args (exactly what ref_render gets, besides --path/--output)
CASES = [
    # must equal the reference's own renders/testcpu.ppm
    ("dragon_parallel_entry", "dragon.scn", ["--parallel-entry"]),
    # G2/G3 (depth-1 plumbing, with and without shadows)
    ("spheres1_d1_noshadow", "spheres1.scn", ["--width", "160", "--height", "120", "--depth", "1"]),
    ("spheres1_d1_shadow", "spheres1.scn", ["--width", "160", "--height", "120", "--depth", "1", "--shadow"]),
    ("spheres2_shadow", "spheres2.scn", ["--width", "320", "--height", "180", "--shadow"]),
    ("spheres2_noshadow_fov90", "spheres2.scn", ["--width", "200", "--height", "150", "--fov", "90"]),
    # G4 = BASELINE config 1 at its real size
    ("spheres1_640x360_d1", "spheres1.scn", ["--width", "640", "--height", "360", "--depth", "1"]),
    # G5-G7: rand()-consuming paths, serial, seed pinned
    ("spheres2_gi16_shadow", "spheres2.scn", ["--width", "160", "--height", "90", "--shadow", "--gillum", "16", "--seed", "20261004"]),
    ("spheres2_js3_shadow", "spheres2.scn", ["--width", "160", "--height", "90", "--shadow", "--jsample", "3", "--seed", "7"]),
    ("spheres2_gi4_js2_d2", "spheres2.scn", ["--width", "160", "--height", "90", "--shadow", "--gillum", "4", "--jsample", "2", "--depth", "2", "--seed", "5"]),
    ("spheres1_gi8_noshadow", "spheres1.scn", ["--width", "160", "--height", "90", "--gillum", "8", "--seed", "99"]),
    ("spheres2_gi3_d4", "spheres2.scn", ["--width", "96", "--height", "54", "--shadow", "--gillum", "3", "--depth", "4", "--seed", "12"]),
    # G8: mixed spheres + triangles + unknown commands; many spheres
    ("test_shadow", "test.scn", ["--width", "160", "--height", "120", "--shadow"]),
    ("test_gi4", "test.scn", ["--width", "80", "--height", "60", "--shadow", "--gillum", "4", "--seed", "3"]),
    ("bear_shadow", "bear.scn", ["--width", "160", "--height", "120", "--shadow"]),
    ("dragon_160x120", "dragon.scn", ["--width", "160", "--height", "120", "--depth", "1"]),
    # round 2: a deep, narrow tree (--depth beyond the old GPU cap of 6; main.cpp:318-329 takes any positive depth)
    ("spheres2_gi2_d8", "spheres2.scn", ["--width", "48", "--height", "27", "--shadow", "--gillum", "2", "--depth", "8", "--seed", "8"]),
    # the headline GI case under a second srand() seed: what two independent Monte-Carlo frames of the REFERENCE look like
    # against each other (tests/test_statistics.py compares the counter-RNG frame with both)
    ("spheres2_gi16_shadow_seed2", "spheres2.scn", ["--width", "160", "--height", "90", "--shadow", "--gillum", "16", "--seed", "777"]),
    # --strict-scn: ref_driver.cpp --strict pushes the directional lights scene.cpp:139-163 parses and drops;
    # the reference's own blinn_phong.h:77-85,122-131 and utils.h:60-76 shade them (spheres2.scn has two)
    ("spheres2_strict_shadow", "spheres2.scn", ["--width", "160", "--height", "90", "--shadow", "--strict"]),
    ("spheres2_strict_noshadow", "spheres2.scn", ["--width", "160", "--height", "90", "--strict"]),
    ("spheres2_strict_gi4", "spheres2.scn", ["--width", "96", "--height", "54", "--shadow", "--gillum", "4", "--strict", "--seed", "3"]),
    # --legacy-reflect (round 3): ref_driver.cpp --legacy runs the control flow of raytrace.h:36-103 without its early
    # return around the reference's own bp::fresnel / refraction / reflect_direction / ambient / diffuse / specular and its
    # intersection functions (composition restated, every value the reference's)
    ("spheres2_legacy_d2", "spheres2.scn", ["--width", "160", "--height", "90", "--depth", "2", "--legacy"]),
    ("spheres2_legacy_d3_shadow", "spheres2.scn", ["--width", "160", "--height", "90", "--depth", "3", "--shadow", "--legacy"]),
    ("spheres2_legacy_strict_d3", "spheres2.scn", ["--width", "96", "--height", "54", "--depth", "3", "--shadow", "--strict", "--legacy"]),
    ("spheres1_legacy_d4", "spheres1.scn", ["--width", "96", "--height", "54", "--depth", "4", "--shadow", "--legacy"]),
]


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def gz_write(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        g.write(f.read())


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/ref_render"])
    os.makedirs(os.path.join(GOLD, "scenes"), exist_ok=True)
    manifest = {"generator": "tools/make_golden.py", "binary": "oracle/_ref/ref_render (reference shade()/parseScene() unmodified)",
                "cases": {}, "scene_dumps": {}, "reference_fixture": {}}
    # data inputs
    for scn in sorted(os.listdir(os.path.join(REF, "scenes"))):
        if scn.endswith(".scn"):
            shutil.copyfile(os.path.join(REF, "scenes", scn), os.path.join(GOLD, "scenes", scn))
            os.chmod(os.path.join(GOLD, "scenes", scn), 0o644)
    # the reference's own pixel-exact fixture
    gz_write(os.path.join(REF, "renders", "testcpu.ppm"), os.path.join(GOLD, "testcpu.ppm.gz"))
    manifest["reference_fixture"]["testcpu.ppm.gz"] = {
        "source": "renders/testcpu.ppm", "sha256_uncompressed": sha(os.path.join(REF, "renders", "testcpu.ppm")),
        "meaning": "HEAD `--path scenes/dragon.scn --parallel true` (640x480, depth 1)"}
    tmp = "/tmp/golden_tmp.ppm"
    for name, scn, args in CASES:
        cmd = [BIN, "--path", os.path.join(REF, "scenes", scn), "--output", tmp] + args
        print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)  # (blinn_phong.h:124 prints a line per specular call once a directional light exists)
        dst = os.path.join(GOLD, "ref_%s.ppm.gz" % name)
        gz_write(tmp, dst)
        manifest["cases"][name] = {"scene": scn, "args": args, "sha256_uncompressed": sha(tmp), "file": os.path.basename(dst)}
    # parsed-scene dumps of the real parseScene(), hex floats
    for scn in ["spheres1.scn", "spheres2.scn", "bear.scn", "test.scn", "dragon.scn"]:
        d = "/tmp/golden_dump.txt"
        subprocess.check_call([BIN, "--path", os.path.join(REF, "scenes", scn), "--dump-scene", d])
        dst = os.path.join(GOLD, "scene_dump_%s.txt.gz" % scn[:-4])
        gz_write(d, dst)
        manifest["scene_dumps"][scn] = {"file": os.path.basename(dst), "sha256_uncompressed": sha(d)}
    # bp::fresnel / bp::refraction / bp::reflect_direction themselves on 10 000 triples (ref_driver.cpp --eval-legacy): 14 words per triple
    # [dir.xyz normal.xyz ior | fresnel | refraction.xyz | reflect_direction(normalize(dir), normal).xyz], kept as a uint32 array
    try:
        import io
        import numpy as np
        ev = "/tmp/golden_legacy_eval.txt"
        subprocess.check_call([BIN, "--eval-legacy", ev])
        words = np.array([[int(t, 16) for t in ln.split()] for ln in open(ev)], dtype=np.uint32)
        assert words.shape == (10000, 14)
        buf = io.BytesIO()
        np.save(buf, words)
        with open(os.path.join(GOLD, "ref_legacy_eval.npy.gz"), "wb") as f:
            f.write(gzip.compress(buf.getvalue(), 9, mtime=0))
        manifest["reference_fixture"]["ref_legacy_eval.npy.gz"] = {
            "source": "oracle/_ref/ref_render --eval-legacy", "sha256_text": sha(ev),
            "meaning": "the reference's bp::fresnel (blinn_phong.h:156), bp::refraction (:143), bp::reflect_direction (:137) on 10 000 (direction, normal, ior) triples"}
    except ImportError:
        print("numpy not importable: legacy eval fixture left as it is")
    # one of the reference's README pictures (a PNG screenshot, made when raytrace.h:45-103 still ran), box-filtered 4x: the visual
    # check of --legacy-reflect (tests/test_legacy_reflect.py).  Data only; PIL decodes the PNG.
    try:
        import io
        import numpy as np
        from PIL import Image
        src = os.path.join(REF, "renders", "shadows", "sample_pngs", "bp_jsample5_parallel_shadows.png")
        im = np.asarray(Image.open(src).convert("RGB"))
        h4, w4 = im.shape[0] // 4 * 4, im.shape[1] // 4 * 4
        small = im[:h4, :w4].reshape(h4 // 4, 4, w4 // 4, 4, 3).astype(np.float32).mean(axis=(1, 3)).round().astype(np.uint8)
        buf = io.BytesIO()
        np.save(buf, small)
        with open(os.path.join(GOLD, "readme_bp_jsample5_parallel_shadows_quarter.npy.gz"), "wb") as f:
            f.write(gzip.compress(buf.getvalue(), 9, mtime=0))
        manifest["reference_fixture"]["readme_bp_jsample5_parallel_shadows_quarter.npy.gz"] = {
            "source": "renders/shadows/sample_pngs/bp_jsample5_parallel_shadows.png", "sha256_source": sha(src),
            "meaning": "README picture (1919x1003 screenshot of the 1920x1080 window), mean of 4x4 pixel blocks, uint8 [250, 479, 3]"}
    except ImportError:
        print("PIL not importable: README picture fixture left as it is")
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", GOLD)


if __name__ == "__main__":
    main()

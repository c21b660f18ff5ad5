import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLD, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_path(name):
    return os.path.join(SCENES, name)


def read_ppm_bytes(data):
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255", parts[:3]
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8, w * h * 3).reshape(h, w, 3)


def read_golden_ppm(fname):
    with gzip.open(os.path.join(GOLD, fname), "rb") as f:
        return read_ppm_bytes(f.read())


def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


def args_to_kwargs(args):
    """ref_render argument list (manifest.json) -> keyword arguments of the python bindings."""
    kw = dict(width=1920, height=1080, fov=60.0, gillum=None, jsample=0, depth=3, shadow=False, seed=1)  # (+ strict=True for --strict)
    it = iter(args)
    for a in it:
        if a == "--width":
            kw["width"] = int(next(it))
        elif a == "--height":
            kw["height"] = int(next(it))
        elif a == "--fov":
            kw["fov"] = float(next(it))
        elif a == "--gillum":
            kw["gillum"] = int(next(it))
        elif a == "--jsample":
            kw["jsample"] = int(next(it))
        elif a == "--depth":
            kw["depth"] = int(next(it))
        elif a == "--seed":
            kw["seed"] = int(next(it))
        elif a == "--shadow":
            kw["shadow"] = True
        elif a == "--strict":  # ref_driver.cpp: the directional lights pushed (--strict-scn)
            kw["strict"] = True
        elif a == "--legacy":  # ref_driver.cpp: raytrace.h:45-103 runs (--legacy-reflect)
            kw["legacy_reflect"] = True
        elif a == "--parallel-entry":  # main.cpp:21-24
            kw.update(width=640, height=480, depth=1, jsample=0)
    return kw


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle

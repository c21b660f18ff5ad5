"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package (skele_raytracer_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "ref_render")

RNG_GLIBC_REPLAY, RNG_COUNTER = 0, 1
MATH_LIBM, MATH_SHARED = 0, 1


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_float), ("ambient", Vec3), ("diffuse", Vec3),
                ("specular", Vec3), ("power", C.c_float), ("transmissive", Vec3), ("ior", C.c_float)]


class Triangle(C.Structure):
    _fields_ = [("v0", Vec3), ("v1", Vec3), ("v2", Vec3)]


class PointLight(C.Structure):
    _fields_ = [("position", Vec3), ("colour", Vec3)]


class DirectionalLight(C.Structure):
    _fields_ = [("direction", Vec3), ("colour", Vec3)]


class Scene(C.Structure):
    _fields_ = [("cam_pos", Vec3), ("cam_dir", Vec3), ("cam_up", Vec3), ("cam_right", Vec3),
                ("cam_half_angle", C.c_float), ("background", Vec3), ("ambient", Vec3),
                ("n_spheres", C.c_int), ("n_triangles", C.c_int), ("n_point_lights", C.c_int),
                ("n_vertices", C.c_int),
                ("spheres", C.POINTER(Sphere)), ("triangles", C.POINTER(Triangle)),
                ("point_lights", C.POINTER(PointLight)),
                ("film_w", C.c_int), ("film_h", C.c_int), ("max_depth_parsed", C.c_int),
                ("n_directional_dropped", C.c_int), ("n_fog_skipped", C.c_int),
                ("n_unknown", C.c_int), ("n_bad_triangles", C.c_int),
                ("n_directional_lights", C.c_int), ("directional_lights", C.POINTER(DirectionalLight)),
                ("triangle_materials", C.POINTER(Sphere))]


class Options(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fov", C.c_float),
                ("monte_carlo", C.c_int32), ("num_path_traces", C.c_int32), ("grid_size", C.c_int32),
                ("max_depth", C.c_int32), ("use_shadows", C.c_int32), ("rng_mode", C.c_int32),
                ("math_mode", C.c_int32), ("seed", C.c_uint64), ("y0", C.c_int32), ("y1", C.c_int32),
                ("threads", C.c_int32), ("shade_triangles", C.c_int32), ("legacy_reflect", C.c_int32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.sko_scene_load.argtypes = [C.c_char_p, C.POINTER(Scene)]
        L.sko_scene_load.restype = C.c_int
        L.sko_scene_load_ex.argtypes = [C.c_char_p, C.c_int, C.POINTER(Scene)]
        L.sko_scene_load_ex.restype = C.c_int
        L.sko_scene_free.argtypes = [C.POINTER(Scene)]
        L.sko_render.argtypes = [C.POINTER(Scene), C.POINTER(Options), C.c_void_p, C.c_void_p, C.c_void_p]
        L.sko_render.restype = C.c_int
        L.sko_philox4x32_10.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.sko_philox4x32_spec.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.sko_philox4x32_r.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        L.sko_sincos_shared.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.sko_powf_shared.argtypes = [C.c_float, C.c_float]
        L.sko_powf_shared.restype = C.c_float
        L.sko_smallest_root.argtypes = [C.c_float] * 3
        L.sko_smallest_root.restype = C.c_float
        L.sko_quantise.argtypes = [C.c_float]
        L.sko_quantise.restype = C.c_uint8
        L.sko_counter_jitter.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.sko_counter_jitter.restype = C.c_float
        L.sko_counter_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                        C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.sko_primary_direction.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_float)]
        L.sko_primary_direction.restype = None
        L.sko_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.sko_write_ppm.restype = C.c_int
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, path, strict=False):
        """strict: the --strict-scn loader (directional lights kept: skr_oracle.h sko_scene_load_ex)."""
        self.s = Scene()
        if lib().sko_scene_load_ex(os.fsencode(path), int(bool(strict)), C.byref(self.s)) != 0:
            raise FileNotFoundError(path)

    def __del__(self):
        try:
            lib().sko_scene_free(C.byref(self.s))
        except Exception:
            pass


def primary_direction(scene, width, height, fov, x, y, jitter, r=0.0):
    """The primary ray direction the render loop forms (skr_oracle.h sko_primary_direction)."""
    out = (C.c_float * 3)()
    lib().sko_primary_direction(C.byref(scene.s), width, height, fov, x, y, int(bool(jitter)), r, out)
    return np.array(out[:], np.float32)


def host_cores():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return n


def render(scene, width, height, *, fov=60.0, gillum=None, jsample=0, depth=3, shadow=False,
           rng=RNG_COUNTER, math=MATH_SHARED, seed=1, y0=0, y1=None, threads=None, want_float=False, strict=False, shade_triangles=False, legacy_reflect=False):
    """Returns (rgb uint8 [rows,W,3], float image or None, stats uint64[5])."""
    if isinstance(scene, (str, os.PathLike)):
        scene = OracleScene(scene, strict=strict)
    y1 = height if y1 is None else y1
    o = Options(width, height, fov, 0 if gillum is None else 1, 1 if gillum is None else gillum, jsample,
                depth, int(bool(shadow)), rng, math, seed, y0, y1, threads or host_cores(), int(bool(shade_triangles)), int(bool(legacy_reflect)))
    rows = y1 - y0
    rgb = np.zeros((rows, width, 3), np.uint8)
    rgbf = np.zeros((rows, width, 3), np.float32) if want_float else None
    stats = np.zeros(5, np.uint64)
    rc = lib().sko_render(C.byref(scene.s), C.byref(o), rgb.ctypes.data,
                          rgbf.ctypes.data if want_float else None, stats.ctypes.data)
    if rc != 0:
        raise RuntimeError("sko_render failed: %d" % rc)
    return rgb, rgbf, stats


def quantise(img):
    """main.cpp:205 on a float array: sko_quantise element by element (vectorised restatement of skr_oracle.c sko_quantise)."""
    c = np.asarray(img, np.float32)
    with np.errstate(invalid="ignore"):
        m = np.where(c < np.float32(1.0), c, np.float32(1.0))  # std::min(1.0f, c): NaN -> 1
        s = m * np.float32(255)
        ok = s > np.float32(-2147483648.0)
        return np.where(ok, np.where(ok, s, 0).astype(np.int64) & 0xff, 0).astype(np.uint8)


def render_progressive(scene, width, height, passes, *, seed=1, **kw):
    """--progressive K (no counterpart in the reference beyond its SDL viewer's purpose, main.cpp:183-197): K whole
    frames under the seeds seed, seed+1, ..., summed in binary32 in that order, divided by (float) K once, quantised like a single
    frame.  Returns (rgb, mean float image, summed stats, [mean after 1, 2, ... K passes])."""
    if isinstance(scene, (str, os.PathLike)):
        scene = OracleScene(scene, strict=kw.get("strict", False))
    acc, stats, means = None, np.zeros(5, np.uint64), []
    for k in range(passes):
        _, f, st = render(scene, width, height, seed=(seed + k) & 0xFFFFFFFFFFFFFFFF, want_float=True, **{a: b for a, b in kw.items() if a != "want_float"})
        acc = f.copy() if acc is None else acc + f
        stats += st
        with np.errstate(invalid="ignore", divide="ignore"):
            means.append(acc / np.float32(k + 1))
    return quantise(means[-1]), means[-1], stats, means


def read_ppm(path):
    with open(path, "rb") as f:
        data = f.read()
    # P6\nW H\n255\n
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6" and parts[2] == b"255", parts[:3]
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8, w * h * 3).reshape(h, w, 3)


def have_ref():
    return os.path.exists(REF_BIN)


def ref_render(scn, out_ppm, width, height, *, fov=None, gillum=None, jsample=None, depth=None, shadow=False,
               seed=1, float_out=None, parallel_entry=False):
    """Run the reference's own shade()/parseScene() (oracle/_ref/ref_render)."""
    cmd = [REF_BIN, "--path", scn, "--output", out_ppm, "--width", str(width), "--height", str(height),
           "--seed", str(seed)]
    if fov is not None:
        cmd += ["--fov", repr(float(fov))]
    if gillum is not None:
        cmd += ["--gillum", str(gillum)]
    if jsample:
        cmd += ["--jsample", str(jsample)]
    if depth is not None:
        cmd += ["--depth", str(depth)]
    if shadow:
        cmd += ["--shadow"]
    if float_out:
        cmd += ["--float-out", float_out]
    if parallel_entry:
        cmd += ["--parallel-entry"]
    subprocess.check_call(cmd)
    return read_ppm(out_ppm)

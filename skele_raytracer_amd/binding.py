"""ctypes binding of lib/libskr.so (include/skr.h).  No compute happens here."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/skr.h declares (tests/test_abi.py checks the library exports them)
EXPORTED_SYMBOLS = [
    "skr_scene_create_from_scn", "skr_scene_create_from_scn_ex", "skr_scene_create_from_arrays", "skr_scene_set_triangle_materials", "skr_scene_set_sphere_ior", "skr_scene_destroy", "skr_scene_get_info",
    "skr_scene_get_arrays", "skr_scene_get_culling", "skr_options_default", "skr_radiance_ray_count", "skr_device_count",
    "skr_renderer_create", "skr_renderer_clone", "skr_renderer_destroy", "skr_render_tiles", "skr_render_tile_list", "skr_tile_costs", "skr_tile_count", "skr_render_rows",
    "skr_renderer_read_counters", "skr_renderer_read_work", "skr_renderer_read_triangle_work", "skr_renderer_count_triangle_work", "skr_renderer_kernel_work", "skr_renderer_reload_switches", "skr_renderer_kernel_timing", "skr_renderer_kernel_ms", "skr_renderer_last_parent_count", "skr_renderer_last_level1_count", "skr_render_frame_host", "skr_render_progressive_host", "skr_accumulate", "skr_resolve_accumulated", "skr_write_png", "skr_write_pfm", "skr_write_ppm", "skr_last_error",
    "skr_kernel_variant", "skr_debug_eval",
    "skr_rccl_available", "skr_multi_create", "skr_multi_destroy", "skr_multi_device_count", "skr_multi_renderer", "skr_multi_render_frame",
    "skr_multi_render_frame_host", "skr_comm_unique_id", "skr_comm_create", "skr_comm_destroy", "skr_comm_render_frame", "skr_comm_render_frame_async", "skr_comm_flush", "skr_comm_frame_to_host",
    "skr_shard_tiles_per_rank", "skr_shard_deinterleave_host", "skr_shard_lpt", "skr_shard_by_cost", "skr_shard_plan", "skr_shard_deinterleave_map_host", "skr_multi_render_frame_async", "skr_multi_flush",
]


class SkrError(RuntimeError):
    pass


class COptions(C.Structure):
    # struct skr_options == reference struct Options (utils.h:26-34) + width/height/use_shadows
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fov", C.c_float), ("monte_carlo", C.c_int32),
                ("num_path_traces", C.c_int32), ("grid_size", C.c_int32), ("max_depth", C.c_int32),
                ("use_shadows", C.c_int32), ("seed", C.c_uint64), ("shade_triangles", C.c_int32), ("progressive_passes", C.c_int32), ("legacy_reflect", C.c_int32)]


# include/skr.h skr_progress_fn: (user, passes_done, passes, h_rgb, h_rgbf) -> non-zero stops the render
PROGRESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p)


class CSceneInfo(C.Structure):
    _fields_ = [("n_spheres", C.c_int32), ("n_triangles", C.c_int32), ("n_point_lights", C.c_int32),
                ("n_vertices", C.c_int32), ("n_directional_dropped", C.c_int32), ("n_fog_skipped", C.c_int32),
                ("n_unknown", C.c_int32), ("n_bad_triangles", C.c_int32), ("film_width", C.c_int32),
                ("film_height", C.c_int32), ("max_depth_parsed", C.c_int32), ("camera", C.c_float * 13),
                ("background", C.c_float * 3), ("ambient", C.c_float * 3), ("n_directional_lights", C.c_int32)]


def lib_path():
    """lib/libskr.so, or the build named by SKR_LIBRARY (kernel experiments: same ABI, other tuning macros)."""
    return os.environ.get("SKR_LIBRARY") or os.path.join(_HERE, "lib", "libskr.so")


_lib = None


def lib():
    """Load libskr.so.  Import torch first when torch is in the process, so that the one
    libamdhip64.so.7 torch ships is the HIP runtime both sides use."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SkrError("%s is missing: run `make lib` (or __graft_entry__.build()); there is no fallback path" % path)
    try:
        import torch  # noqa: F401  (HIP runtime load order, see docstring)
    except ImportError:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p
    L.skr_scene_create_from_scn.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.skr_scene_create_from_scn_ex.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.skr_scene_create_from_arrays.argtypes = [vp, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp, vp, C.POINTER(vp)]
    L.skr_scene_set_triangle_materials.argtypes = [vp, vp]
    L.skr_scene_set_sphere_ior.argtypes = [vp, vp]
    L.skr_scene_destroy.argtypes = [vp]
    L.skr_scene_destroy.restype = None
    L.skr_scene_get_info.argtypes = [vp, C.POINTER(CSceneInfo)]
    L.skr_scene_get_arrays.argtypes = [vp, vp, vp, vp]
    L.skr_scene_get_culling.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp, vp, vp, vp]
    L.skr_options_default.argtypes = [C.POINTER(COptions)]
    L.skr_options_default.restype = None
    L.skr_radiance_ray_count.argtypes = [C.POINTER(COptions)]
    L.skr_radiance_ray_count.restype = C.c_uint64
    L.skr_device_count.restype = C.c_int
    L.skr_renderer_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.skr_renderer_clone.argtypes = [vp, C.POINTER(vp)]
    L.skr_renderer_destroy.argtypes = [vp]
    L.skr_renderer_destroy.restype = None
    L.skr_render_tiles.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.skr_tile_count.argtypes = [C.POINTER(COptions), C.c_uint32, C.c_uint32, C.c_uint32]
    L.skr_tile_count.restype = C.c_uint32
    L.skr_render_rows.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.c_uint32, vp, vp, vp]
    L.skr_renderer_read_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.skr_renderer_read_work.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.skr_renderer_read_triangle_work.argtypes = [vp, C.POINTER(C.c_uint64), C.c_int]
    L.skr_renderer_kernel_work.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.skr_renderer_count_triangle_work.argtypes = [vp, C.c_int]
    L.skr_renderer_reload_switches.argtypes = [vp]
    L.skr_renderer_kernel_timing.argtypes = [vp, C.c_int]
    L.skr_renderer_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.skr_renderer_last_parent_count.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.skr_renderer_last_level1_count.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.skr_render_frame_host.argtypes = [vp, C.POINTER(COptions), vp, C.POINTER(C.c_float)]
    L.skr_write_ppm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.skr_write_png.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.skr_write_pfm.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, vp]
    L.skr_render_progressive_host.argtypes = [vp, C.POINTER(COptions), C.c_uint32, vp, vp, PROGRESS_FN, vp, C.POINTER(C.c_float)]
    L.skr_accumulate.argtypes = [vp, vp, C.c_uint64, C.c_int, vp]
    L.skr_resolve_accumulated.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.skr_last_error.restype = C.c_char_p
    L.skr_kernel_variant.restype = C.c_char_p
    L.skr_debug_eval.argtypes = [C.c_int, vp, vp, C.c_uint32, vp]
    L.skr_rccl_available.restype = C.c_int
    L.skr_multi_create.argtypes = [vp, C.c_int, vp, C.POINTER(vp)]
    L.skr_multi_destroy.argtypes = [vp]
    L.skr_multi_destroy.restype = None
    L.skr_multi_device_count.argtypes = [vp]
    L.skr_multi_renderer.argtypes = [vp, C.c_int]
    L.skr_multi_renderer.restype = vp
    L.skr_multi_render_frame.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.POINTER(vp), C.POINTER(C.c_float)]
    L.skr_multi_render_frame_host.argtypes = [vp, C.POINTER(COptions), C.c_uint32, vp, C.POINTER(C.c_float)]
    L.skr_comm_unique_id.argtypes = [vp]
    L.skr_comm_create.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.skr_comm_destroy.argtypes = [vp]
    L.skr_comm_destroy.restype = None
    L.skr_comm_render_frame.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.POINTER(vp), vp]
    L.skr_comm_frame_to_host.argtypes = [vp, vp, vp]
    L.skr_comm_render_frame_async.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.POINTER(vp), vp]
    L.skr_comm_flush.argtypes = [vp, C.POINTER(vp), vp]
    L.skr_multi_render_frame_async.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.POINTER(vp)]
    L.skr_multi_flush.argtypes = [vp, C.POINTER(vp)]
    L.skr_render_tile_list.argtypes = [vp, C.POINTER(COptions), C.c_uint32, vp, C.c_uint32, vp, vp, vp]
    L.skr_tile_costs.argtypes = [vp, C.POINTER(COptions), C.c_uint32, vp]
    L.skr_shard_lpt.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.skr_shard_by_cost.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.skr_shard_plan.argtypes = [vp, C.POINTER(COptions), C.c_uint32, C.c_uint32, vp]
    L.skr_shard_deinterleave_map_host.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_uint32, vp]
    L.skr_shard_tiles_per_rank.argtypes = [C.c_int32, C.c_uint32, C.c_uint32]
    L.skr_shard_tiles_per_rank.restype = C.c_uint32
    L.skr_shard_deinterleave_host.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise SkrError("%s failed (status %d): %s" % (what, rc, (lib().skr_last_error() or b"").decode()))


class Options:
    """Reference struct Options (utils.h:26-34) with the reference's defaults, plus the
    width/height/use_shadows main() folds in (main.cpp:393-396) and the RNG seed."""

    def __init__(self, width=1920, height=1080, fov=60.0, gillum=None, jsample=0, depth=3, shadow=False, seed=1, shade_triangles=False, progressive=1, legacy_reflect=False):
        c = COptions()
        lib().skr_options_default(C.byref(c))
        c.width, c.height, c.fov = width, height, fov
        if gillum is not None:  # main.cpp:252-253: --gillum N sets monte_carlo and num_path_traces
            c.monte_carlo, c.num_path_traces = 1, gillum
        c.grid_size, c.max_depth, c.use_shadows, c.seed = jsample, depth, int(bool(shadow)), seed
        c.shade_triangles = int(bool(shade_triangles))  # --shade-triangles (include/skr.h): triangles as surfaces, not black holes
        c.progressive_passes = max(1, int(progressive))  # --progressive K: the mean of K frames under the seeds seed .. seed+K-1
        c.legacy_reflect = int(bool(legacy_reflect))  # --legacy-reflect (include/skr.h): the reflection / refraction code behind raytrace.h:44's early return
        self.c = c

    @property
    def width(self):
        return self.c.width

    @property
    def height(self):
        return self.c.height


def radiance_ray_count(opt):
    return int(lib().skr_radiance_ray_count(C.byref(opt.c)))


class Scene:
    """Host scene (reference struct Scene, scene.h:13-28) in SoA form; see parse_scene()."""

    def __init__(self, handle):
        self.h = C.c_void_p(handle)

    def close(self):
        if self.h:
            try:
                lib().skr_scene_destroy(self.h)
            except TypeError:  # (interpreter teardown)
                pass
            self.h = None

    __del__ = close

    @property
    def info(self):
        i = CSceneInfo()
        _check(lib().skr_scene_get_info(self.h, C.byref(i)), "skr_scene_get_info")
        return i

    def arrays(self):
        i = self.info
        s = np.zeros((i.n_spheres, 14), np.float32)
        t = np.zeros((i.n_triangles, 9), np.float32)
        l = np.zeros((i.n_point_lights, 6), np.float32)
        _check(lib().skr_scene_get_arrays(self.h, s.ctypes.data, t.ctypes.data, l.ctypes.data), "skr_scene_get_arrays")
        return s, t, l

    def culling(self, level=0):
        """(chunk_size, device_tris [n,3,4], node_spheres [nn,8], node_links [nn,4] int32, chunk_spheres [nc,8]) —
        include/skr.h skr_scene_get_culling."""
        cs, nn, nc = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().skr_scene_get_culling(self.h, level, C.byref(cs), C.byref(nn), C.byref(nc), None, None, None, None), "skr_scene_get_culling")
        tris = np.zeros((self.info.n_triangles, 3, 4), np.float32)
        sph = np.zeros((nn.value, 8), np.float32)
        links = np.zeros((nn.value, 4), np.int32)
        ch = np.zeros((nc.value, 8), np.float32)
        _check(lib().skr_scene_get_culling(self.h, level, None, None, None, tris.ctypes.data, sph.ctypes.data, links.ctypes.data, ch.ctypes.data),
               "skr_scene_get_culling")
        return cs.value, tris, sph, links, ch

    @staticmethod
    def from_arrays(spheres, triangles, point_lights, camera, background=(0, 0, 0), ambient=(0, 0, 0), triangle_materials=None, sphere_ior=None):
        s = np.ascontiguousarray(spheres, np.float32).reshape(-1, 14)
        t = np.ascontiguousarray(triangles, np.float32).reshape(-1, 9)
        l = np.ascontiguousarray(point_lights, np.float32).reshape(-1, 6)
        cam = np.ascontiguousarray(camera, np.float32).reshape(9)
        bg = np.ascontiguousarray(background, np.float32).reshape(3)
        am = np.ascontiguousarray(ambient, np.float32).reshape(3)
        h = C.c_void_p()
        _check(lib().skr_scene_create_from_arrays(s.ctypes.data, len(s), t.ctypes.data, len(t), l.ctypes.data, len(l),
                                                  cam.ctypes.data, bg.ctypes.data, am.ctypes.data, C.byref(h)),
               "skr_scene_create_from_arrays")
        sc = Scene(h.value)
        if triangle_materials is not None:  # [n_triangles][10] ambient diffuse specular power (--shade-triangles)
            m = np.ascontiguousarray(triangle_materials, np.float32).reshape(len(t), 10)
            _check(lib().skr_scene_set_triangle_materials(sc.h, m.ctypes.data), "skr_scene_set_triangle_materials")
        if sphere_ior is not None:  # [n_spheres] (--legacy-reflect)
            q = np.ascontiguousarray(sphere_ior, np.float32).reshape(len(s))
            _check(lib().skr_scene_set_sphere_ior(sc.h, q.ctypes.data), "skr_scene_set_sphere_ior")
        return sc


def parse_scene(path, echo=False, strict=False):
    """Reference `Scene parseScene(std::string)` (scene.cpp:12); strict = SKR_SCN_STRICT (--strict-scn: directional lights kept)."""
    h = C.c_void_p()
    _check(lib().skr_scene_create_from_scn_ex(os.fsencode(path), int(echo), 1 if strict else 0, C.byref(h)), "skr_scene_create_from_scn_ex")
    return Scene(h.value)


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    _check(lib().skr_write_ppm(os.fsencode(path), w, h, rgb.ctypes.data), "skr_write_ppm")


def write_png(path, rgb):
    """The bytes of the PPM as an 8-bit RGB PNG (include/skr.h skr_write_png)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    _check(lib().skr_write_png(os.fsencode(path), w, h, rgb.ctypes.data), "skr_write_png")


def write_pfm(path, rgbf):
    """The unquantised float frame [H, W, 3] (top row first) as a PFM file (include/skr.h skr_write_pfm)."""
    rgbf = np.ascontiguousarray(rgbf, np.float32)
    h, w, _ = rgbf.shape
    _check(lib().skr_write_pfm(os.fsencode(path), w, h, rgbf.ctypes.data), "skr_write_pfm")


class Renderer:
    """Device context (scene resident in HBM) + the launches.  Needs a gfx950 GPU."""

    def __init__(self, scene, device=0):
        self.scene = scene
        self.device = device
        h = C.c_void_p()
        _check(lib().skr_renderer_create(scene.h, device, C.byref(h)), "skr_renderer_create")
        self.h = h
        self._env = self._switch_env()

    def clone(self):
        """skr_renderer_clone: a second renderer on the same uploaded scene with its own tables (a second frame in flight); it shares this
        renderer's work counters and must be closed before it."""
        other = Renderer.__new__(Renderer)
        other.scene, other.device, other._source = self.scene, self.device, self
        h = C.c_void_p()
        _check(lib().skr_renderer_clone(self.h, C.byref(h)), "skr_renderer_clone")
        other.h = h
        other._env = self._env
        return other

    @staticmethod
    def _switch_env():
        return tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith("SKR_")))

    def _sync_switches(self):
        """libskr reads its SKR_* development switches once per renderer; tests and A/B tools change them between
        frames, so the binding asks for a re-read when this process's environment has changed since."""
        env = self._switch_env()
        if env != self._env:
            _check(lib().skr_renderer_reload_switches(self.h), "skr_renderer_reload_switches")
            self._env = env

    def close(self):
        if getattr(self, "h", None):
            try:
                lib().skr_renderer_destroy(self.h)
            except TypeError:  # interpreter teardown: the module's globals are gone, the process is about to free everything
                pass
            self.h = None

    __del__ = close

    def tile_count(self, opt, tile_rows, first_tile=0, tile_stride=1):
        return int(lib().skr_tile_count(C.byref(opt.c), tile_rows, first_tile, tile_stride))

    def render_tiles_into(self, opt, tile_rows, first_tile, tile_stride, rgb_ptr, rgbf_ptr=None, stream=None):
        """Enqueue the kernels of this partition of the frame (skr_render_tiles); pointers are raw device addresses."""
        self._sync_switches()
        _check(lib().skr_render_tiles(self.h, C.byref(opt.c), tile_rows, first_tile, tile_stride, rgb_ptr, rgbf_ptr,
                                      stream), "skr_render_tiles")

    def render_tile_list_into(self, opt, tile_rows, tiles_ptr, n_slots, rgb_ptr, rgbf_ptr=None, stream=None):
        """skr_render_tile_list: slot k of the compact output holds tile tiles[k] (a device array of uint32; 0xFFFFFFFF = empty)."""
        self._sync_switches()
        _check(lib().skr_render_tile_list(self.h, C.byref(opt.c), tile_rows, tiles_ptr, n_slots, rgb_ptr, rgbf_ptr, stream), "skr_render_tile_list")

    def tile_costs(self, opt, tile_rows):
        """Per tile, its counted work in flops (include/skr.h skr_tile_costs)."""
        self._sync_switches()
        n = (opt.height + tile_rows - 1) // tile_rows
        out = np.zeros(n, np.uint64)
        _check(lib().skr_tile_costs(self.h, C.byref(opt.c), tile_rows, out.ctypes.data), "skr_tile_costs")
        return out

    def shard_plan(self, opt, tile_rows, world):
        """slot_of_tile of the map a frame step of `world` ranks uses (include/skr.h skr_shard_plan)."""
        n = (opt.height + tile_rows - 1) // tile_rows
        out = np.zeros(n, np.uint32)
        _check(lib().skr_shard_plan(self.h, C.byref(opt.c), tile_rows, world, out.ctypes.data), "skr_shard_plan")
        return out

    def render(self, opt, want_float=False, tile_rows=None, first_tile=0, tile_stride=1):
        """Render (a partition of) the frame into torch tensors on this device.  Returns
        (rgb uint8 [rows, W, 3], float32 image or None); rows are tile-major compact."""
        import torch
        dev = torch.device("cuda", self.device)
        tile_rows = tile_rows or opt.height
        n = self.tile_count(opt, tile_rows, first_tile, tile_stride)
        rows = n * tile_rows
        rgb = torch.zeros((rows, opt.width, 3), dtype=torch.uint8, device=dev)
        rgbf = torch.zeros((rows, opt.width, 3), dtype=torch.float32, device=dev) if want_float else None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            self.render_tiles_into(opt, tile_rows, first_tile, tile_stride, rgb.data_ptr(),
                                   rgbf.data_ptr() if want_float else None, stream)
        return rgb, rgbf

    def render_rows(self, opt, y0, y1, want_float=False):
        import torch
        dev = torch.device("cuda", self.device)
        rgb = torch.zeros((y1 - y0, opt.width, 3), dtype=torch.uint8, device=dev)
        rgbf = torch.zeros((y1 - y0, opt.width, 3), dtype=torch.float32, device=dev) if want_float else None
        self._sync_switches()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _check(lib().skr_render_rows(self.h, C.byref(opt.c), y0, y1, rgb.data_ptr(),
                                         rgbf.data_ptr() if want_float else None, stream), "skr_render_rows")
        return rgb, rgbf

    def render_progressive_host(self, opt, every=0, want_float=False, progress=None):
        """include/skr.h skr_render_progressive_host: the whole frame (the mean of opt's progressive passes) into host arrays;
        progress(passes_done, passes, rgb, rgbf) is called after every `every` passes with the mean so far (views of the
        returned arrays; return True to stop).  Returns (rgb uint8 [H, W, 3], rgbf float32 or None, device ms)."""
        self._sync_switches()
        rgb = np.zeros((opt.height, opt.width, 3), np.uint8)
        rgbf = np.zeros((opt.height, opt.width, 3), np.float32) if want_float else None
        cb = PROGRESS_FN((lambda user, done, total, b, f: int(bool(progress(done, total, rgb, rgbf)))) if progress else (lambda *a: 0))
        ms = C.c_float()
        _check(lib().skr_render_progressive_host(self.h, C.byref(opt.c), every, rgb.ctypes.data, rgbf.ctypes.data if want_float else None,
                                                 cb if progress else PROGRESS_FN(), None, C.byref(ms)), "skr_render_progressive_host")
        return rgb, rgbf, ms.value

    def counters(self, reset=True):
        out = (C.c_uint64 * 3)()
        _check(lib().skr_renderer_read_counters(self.h, out, int(reset)), "skr_renderer_read_counters")
        return {"radiance_rays": int(out[0]), "sphere_hits": int(out[1]), "shadow_rays": int(out[2])}

    def work(self, reset=True):
        """counters() plus sphere_tests (include/skr.h skr_renderer_read_work)."""
        out = (C.c_uint64 * 4)()
        _check(lib().skr_renderer_read_work(self.h, out, int(reset)), "skr_renderer_read_work")
        return {"radiance_rays": int(out[0]), "sphere_hits": int(out[1]), "shadow_rays": int(out[2]), "sphere_tests": int(out[3])}

    def kernel_work(self):
        """work() of the one kernel kernel_ms() times, in the last launch made with kernel timing on (include/skr.h skr_renderer_kernel_work)."""
        out = (C.c_uint64 * 4)()
        _check(lib().skr_renderer_kernel_work(self.h, out), "skr_renderer_kernel_work")
        return {"radiance_rays": int(out[0]), "sphere_hits": int(out[1]), "shadow_rays": int(out[2]), "sphere_tests": int(out[3])}

    def count_triangle_work(self, enable=True):
        _check(lib().skr_renderer_count_triangle_work(self.h, int(enable)), "skr_renderer_count_triangle_work")

    def triangle_work(self, reset=True):
        """What the triangle walks executed (include/skr.h skr_renderer_read_triangle_work): call it BEFORE work(reset=True)."""
        out = (C.c_uint64 * 3)()
        _check(lib().skr_renderer_read_triangle_work(self.h, out, int(reset)), "skr_renderer_read_triangle_work")
        return {"cull_tests": int(out[0]), "triangle_tests": int(out[1]), "reference_triangle_tests": int(out[2])}

    def kernel_timing(self, enable=True):
        _check(lib().skr_renderer_kernel_timing(self.h, int(enable)), "skr_renderer_kernel_timing")

    def kernel_ms(self):
        """(mean ms of the dominant kernel, number of launches) since the last call; synchronises."""
        ms, n = C.c_float(), C.c_int32()
        _check(lib().skr_renderer_kernel_ms(self.h, C.byref(ms), C.byref(n)), "skr_renderer_kernel_ms")
        return ms.value, n.value

    def last_parent_count(self):
        n = C.c_uint32()
        _check(lib().skr_renderer_last_parent_count(self.h, C.byref(n)), "skr_renderer_last_parent_count")
        return n.value

    def last_level1_count(self):
        n = C.c_uint32()
        _check(lib().skr_renderer_last_level1_count(self.h, C.byref(n)), "skr_renderer_last_level1_count")
        return n.value

    @staticmethod
    def kernel_variant():
        return (lib().skr_kernel_variant() or b"").decode()


COMM_ID_BYTES = 128


def rccl_available():
    return bool(lib().skr_rccl_available())


def comm_unique_id():
    """An RCCL id (rank 0 makes it; the caller broadcasts the 128 bytes by whatever transport it has)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().skr_comm_unique_id(buf), "skr_comm_unique_id")
    return bytes(buf)


class Comm:
    """One rank of the native frame step (include/skr.h skr_comm_*): this rank's tiles, ONE RCCL all-gather, rank 0's
    de-interleave on the device — all inside libskr, enqueued on the caller's stream."""

    def __init__(self, renderer, rank, world, unique_id=None):
        self.renderer, self.rank, self.world = renderer, rank, world
        h = C.c_void_p()
        idbuf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id) if unique_id is not None else None
        _check(lib().skr_comm_create(renderer.h, renderer.device, idbuf, rank, world, C.byref(h)), "skr_comm_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            try:
                lib().skr_comm_destroy(self.h)
            except TypeError:  # (interpreter teardown)
                pass
            self.h = None

    __del__ = close

    def render_frame(self, opt, tile_rows, stream=None):
        """Enqueue one frame; returns the device address of rank 0's finished frame (None on the other ranks)."""
        self.renderer._sync_switches()
        d = C.c_void_p()
        _check(lib().skr_comm_render_frame(self.h, C.byref(opt.c), tile_rows, C.byref(d), stream), "skr_comm_render_frame")
        return d.value

    def render_frame_async(self, opt, tile_rows, stream=None, want_previous=True):
        """Enqueue one frame with its collective on the communicator's own stream (include/skr.h skr_comm_render_frame_async);
        returns the device address of the PREVIOUS call's frame (rank 0; None on the first call, on the other ranks, and when not asked for)."""
        self.renderer._sync_switches()
        d = C.c_void_p()
        _check(lib().skr_comm_render_frame_async(self.h, C.byref(opt.c), tile_rows, C.byref(d) if want_previous else None, stream), "skr_comm_render_frame_async")
        return d.value

    def flush(self, stream=None):
        """Ends a run of render_frame_async calls: `stream` waits for the last collective; returns the last frame's address (rank 0)."""
        d = C.c_void_p()
        _check(lib().skr_comm_flush(self.h, C.byref(d), stream), "skr_comm_flush")
        return d.value

    def frame_to_host(self, opt, stream=None):
        """Rank 0: the frame of the last render_frame as a numpy array (waits for the stream)."""
        rgb = np.zeros((opt.height, opt.width, 3), np.uint8)
        _check(lib().skr_comm_frame_to_host(self.h, rgb.ctypes.data, stream), "skr_comm_frame_to_host")
        return rgb


class Multi:
    """One process, N devices (include/skr.h skr_multi_*)."""

    def __init__(self, scene, n_devices):
        h = C.c_void_p()
        _check(lib().skr_multi_create(scene.h, n_devices, None, C.byref(h)), "skr_multi_create")
        self.h, self.n = h, n_devices

    def close(self):
        if getattr(self, "h", None):
            try:
                lib().skr_multi_destroy(self.h)
            except TypeError:  # (interpreter teardown)
                pass
            self.h = None

    __del__ = close

    def render_frame_host(self, opt, tile_rows=8):
        rgb = np.zeros((opt.height, opt.width, 3), np.uint8)
        ms = C.c_float()
        _check(lib().skr_multi_render_frame_host(self.h, C.byref(opt.c), tile_rows, rgb.ctypes.data, C.byref(ms)), "skr_multi_render_frame_host")
        return rgb, ms.value

    def render_frame_async(self, opt, tile_rows=8, want_previous=True):
        """skr_multi_render_frame_async: returns the device address of the PREVIOUS call's frame (0 on the first call)."""
        prev = C.c_void_p()
        _check(lib().skr_multi_render_frame_async(self.h, C.byref(opt.c), tile_rows, C.byref(prev) if want_previous else None), "skr_multi_render_frame_async")
        return prev.value or 0

    def flush(self):
        last = C.c_void_p()
        _check(lib().skr_multi_flush(self.h, C.byref(last)), "skr_multi_flush")
        return last.value or 0

    def counters(self, reset=True):
        tot = {"radiance_rays": 0, "sphere_hits": 0, "shadow_rays": 0}
        for i in range(self.n):
            out = (C.c_uint64 * 3)()
            _check(lib().skr_renderer_read_counters(lib().skr_multi_renderer(self.h, i), out, int(reset)), "skr_renderer_read_counters")
            for k, name in enumerate(tot):
                tot[name] += int(out[k])
        return tot


def shard_tiles_per_rank(height, tile_rows, world):
    return int(lib().skr_shard_tiles_per_rank(height, tile_rows, world))


def shard_lpt(cost, world):
    """The cost-aware map: slot_of_tile[t] = rank * k_max + slot (include/skr.h skr_shard_lpt)."""
    c = np.ascontiguousarray(cost, np.uint64)
    out = np.zeros(len(c), np.uint32)
    _check(lib().skr_shard_lpt(c.ctypes.data, len(c), world, out.ctypes.data), "skr_shard_lpt")
    return out


def shard_by_cost(cost, world):
    """The frame steps' rule on given costs: `t mod world` unless that is more than 10 % off balance, then LPT (include/skr.h skr_shard_by_cost)."""
    c = np.ascontiguousarray(cost, np.uint64)
    out = np.zeros(len(c), np.uint32)
    _check(lib().skr_shard_by_cost(c.ctypes.data, len(c), world, out.ctypes.data), "skr_shard_by_cost")
    return out


def shard_deinterleave_map_host(gathered, width, height, tile_rows, slot_of_tile):
    g = np.ascontiguousarray(gathered, np.uint8)
    m = np.ascontiguousarray(slot_of_tile, np.uint32)
    out = np.zeros((height, width, 3), np.uint8)
    _check(lib().skr_shard_deinterleave_map_host(g.ctypes.data, out.ctypes.data, width, height, tile_rows, m.ctypes.data), "skr_shard_deinterleave_map_host")
    return out


def shard_deinterleave_host(gathered, width, height, tile_rows, world):
    g = np.ascontiguousarray(gathered, np.uint8)
    out = np.zeros((height, width, 3), np.uint8)
    _check(lib().skr_shard_deinterleave_host(g.ctypes.data, out.ctypes.data, width, height, tile_rows, world), "skr_shard_deinterleave_host")
    return out


def debug_eval(op, inp, out_words_per_record, device=0):
    """Run the arithmetic-spec debug kernel on n records (uint32 words)."""
    import torch
    dev = torch.device("cuda", device)
    a = torch.from_numpy(np.ascontiguousarray(inp).view(np.uint32).astype(np.int64)).to(torch.int64)
    a = a.to(dev).to(torch.int32).contiguous()  # same bits as uint32
    n = a.shape[0]
    o = torch.zeros((n, out_words_per_record), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _check(lib().skr_debug_eval(op, a.data_ptr(), o.data_ptr(), n, torch.cuda.current_stream(dev).cuda_stream),
               "skr_debug_eval")
    torch.cuda.synchronize(dev)
    return o.cpu().numpy().view(np.uint32)

"""skele_raytracer_amd — MI355X-native drop-in for the per-pixel hot path of
lilinitsy/skele-raytracer.

The product is the C-ABI library ``lib/libskr.so`` (include/skr.h: hand-written
HIP kernels for gfx950 + the .scn loader and PPM writer) and the ``raytracer``
command line built on it.  This package is the thin Python host binding used by
tests/ and bench.py; names follow the reference (parseScene -> parse_scene,
struct Options -> Options, generate_rays -> render).  PyTorch only supplies
device memory, streams and torch.distributed.

There is no CPU fallback: every render entry point raises if libskr.so or a
gfx950 device is missing.
"""
from .binding import (Options, Renderer, Scene, SkrError, lib, lib_path, parse_scene, radiance_ray_count,
                      write_ppm, write_png, write_pfm, EXPORTED_SYMBOLS)

__all__ = ["Options", "Renderer", "Scene", "SkrError", "lib", "lib_path", "parse_scene", "radiance_ray_count",
           "write_ppm", "write_png", "write_pfm", "EXPORTED_SYMBOLS"]

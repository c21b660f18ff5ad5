#!/usr/bin/env python3
"""Event counts of one headline frame from a diagnostic build (-DSKR_DIAG=1, loaded through SKR_LIBRARY):
how often the sphere loops take their candidate paths, how full the shading batches are."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import skele_raytracer_amd as skr
from skele_raytracer_amd import binding
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
L = binding.lib()
out = np.zeros(32, np.uint64)
r.render(opt); torch.cuda.synchronize()
L.skr_diag_read_nodes(C.c_void_p(out.ctypes.data), 1)
r.render(opt); torch.cuda.synchronize()
L.skr_diag_read_nodes(C.c_void_p(out.ctypes.data), 1)
# (the counters of render_nodes.hip's translation unit: the trace, leaf and finalize kernels of the node pipeline)
names = ["closest-pair iterations", "closest-pair candidate paths", "  lanes in them", "exact-root fallbacks (lanes)", "shadow-pair iterations", "shadow candidate paths",
         "  lanes in them", "exact-root fallbacks (wave events)", "bracket overlaps -> exact loop (wave events)", "leaf shading batches", "  hits in them",
         "activation batches", "  records in them", "closest-pair candidate RAYS", "  accepted", "shadow candidate RAYS", "  occluders found", "  closest-pair candidates a t2 <= 1 pre-test rejects", "  closest-pair candidate paths left with it"] + ["counter %d" % i for i in range(19, 32)]
print(r.kernel_variant())
for n, v in zip(names, out):
    if v: print("%-48s %d" % (n, v))

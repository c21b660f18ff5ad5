#!/usr/bin/env python3
"""Per-phase cycle shares of the node pipeline's leaf kernel (diagnostic build -DSKR_STAMPS=1 via SKR_LIBRARY)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
r.render(opt); torch.cuda.synchronize(); r.counters()
r.render(opt); torch.cuda.synchronize()
print(r.kernel_variant(), "phases: 0 pull, 1 activation, 2 trace, 3 leaf shading, 4 pushes + window sums, 5 unit end", flush=True)
os.environ["SKR_PRINT_STAMPS"] = "1"; r.counters(); os.environ.pop("SKR_PRINT_STAMPS")

import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, skele_raytracer_amd as skr
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
for mode in ("default", "levels"):
    if mode == "levels": os.environ["SKR_PIPELINE"] = "levels"
    r.render(opt); torch.cuda.synchronize(); r.counters()
    r.render(opt); torch.cuda.synchronize()
    print(mode, r.kernel_variant(), flush=True)
    os.environ["SKR_PRINT_STAMPS"] = "1"; r.counters(); os.environ.pop("SKR_PRINT_STAMPS")

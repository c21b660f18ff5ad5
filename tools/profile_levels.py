import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SKR_PIPELINE"] = "levels"
import torch, skele_raytracer_amd as skr
r = skr.Renderer(skr.parse_scene(os.path.join(ROOT, "tests/golden/scenes/spheres2.scn")))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
opt = skr.Options(1920, 1080, gillum=16, shadow=True, seed=20261004)
n = r.tile_count(opt, 8, 0, G)
buf = torch.zeros((n * 8, 1920, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream()
for _ in range(10):
    r.render_tiles_into(opt, 8, 0, G, buf.data_ptr(), None, st.cuda_stream)
torch.cuda.synchronize()
print(r.kernel_variant())

import os, sys
sys.path.insert(0, '/root/repo'); os.environ['SKR_PRINT_STAMPS']='1'
import torch, skele_raytracer_amd as skr
sc = skr.parse_scene('/root/repo/tests/golden/scenes/spheres2.scn'); r = skr.Renderer(sc)
opt = skr.Options(1920,1080,gillum=16,shadow=True,seed=20261004)
r.render(opt); torch.cuda.synchronize(); 
os.environ.pop('SKR_PRINT_STAMPS'); r.counters(); os.environ['SKR_PRINT_STAMPS']='1'
r.render(opt); torch.cuda.synchronize(); print(r.counters())

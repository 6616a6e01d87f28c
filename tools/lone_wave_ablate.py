import os, sys
sys.path.insert(0, "/root/repo")
import torch
from path_tracer_amd import abi, render as R, scenes
from path_tracer_amd.scene import pack, triangle, box, constant_medium, sphere, _rect
spp = 2048
hs, cam_args, atlas = scenes.smoke_sphere_scene()
cam_args = dict(cam_args, look_at=(0.0, 1.0, 0.0), vfov=1.0, aperture=0.0)
cam = scenes.make_camera(cam_args, 8, 8)
def run(name, hs2):
    ps = pack(hs2, atlas)
    ds = R.DeviceScene(ps)
    R.render(8, 8, 16, ds, cam); torch.cuda.synchronize()
    ms = min(R.render(8, 8, spp, ds, cam, flags=abi.PT_FLAG_NO_LPT, timed=True)[1] for _ in range(3))
    print(f"{name:44s} {len(hs2):4d} hittables {ms / spp * 1e3:7.2f} us per sample", flush=True)
run("original", hs)
run("- pyramid (4 triangles)", [h for h in hs if not isinstance(h, triangle)])
run("- monolith (box)", [h for h in hs if not isinstance(h, box)])
run("- smoke ball (medium)", [h for h in hs if not isinstance(h, constant_medium)])
run("- rect", [h for h in hs if not isinstance(h, _rect)])
small = [h for h in hs if isinstance(h, sphere) and abs(h.radius) < 0.5 and h.center0[1] < 0.5]
run("- the small spheres (grid run)", [h for h in hs if h not in small])
run("- all of the above", [h for h in hs if isinstance(h, sphere) and h not in small])

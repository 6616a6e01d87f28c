"""Where the SmokeSphere frame's time goes, by substitution: the scene re-rendered with one family of shading work after the
other replaced by the cheapest one (solid lambertian).  Not the same paths, so only the size of the steps means something.
    python tools/exp_smoke_materials.py [spp]"""
import dataclasses, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import render as R, scenes
from path_tracer_amd.scene import (checker_texture, constant_medium, dielectric_material, image_texture, isotropic_material,
                                   lambertian_material, lightsource_material, metal_material, pack, solid_texture)

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1920, 1080
hs, cam_args, atlas = scenes.smoke_sphere_scene()
cam = scenes.make_camera(cam_args, W, H)
grey = lambertian_material((0.5, 0.5, 0.5))

def with_mat(h, m):
    import copy
    g = copy.copy(h)
    object.__setattr__(g, "material_type", m)
    return g

def swap(hs, pred):
    out = []
    for h in hs:
        m = getattr(h, "material_type", None)
        out.append(with_mat(h, grey) if m is not None and pred(m) else h)
    return out

def is_tex(m, cls):
    t = getattr(m, "albedo", None) if isinstance(m, lambertian_material) else None
    return isinstance(t, cls)

steps = [("original", lambda hs: hs),
         ("checker ground -> solid", lambda hs: swap(hs, lambda m: is_tex(m, checker_texture))),
         ("+ image textures -> solid", lambda hs: swap(hs, lambda m: is_tex(m, image_texture))),
         ("+ metal -> lambertian", lambda hs: swap(hs, lambda m: isinstance(m, metal_material))),
         ("+ glass -> lambertian", lambda hs: swap(hs, lambda m: isinstance(m, dielectric_material))),
         ("+ smoke ball removed", lambda hs: [h for h in hs if not isinstance(h, constant_medium)])]
cur = hs
for name, f in steps:
    cur = f(cur)
    ps = pack(cur, atlas)
    ds = R.DeviceScene(ps)
    R.render(W, H, 8, ds, cam); torch.cuda.synchronize()
    ms = min(R.render(W, H, spp, ds, cam, timed=True)[1] for _ in range(2))
    print(f"{name:32s} {ms:8.1f} ms  {W * H * spp / ms / 1e3:8.1f} Msamples/s", flush=True)

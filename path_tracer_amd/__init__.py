"""path_tracer_amd — MI355X-native (gfx950) implementation of triSYCL/path_tracer's render() hot path.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels + the C ABI of include/pt_render.h  -> libpt_render.so
  abi.py           ctypes mirror of the C ABI (fails loudly if the extension is missing)
  scene.py         host mirror of the reference's scene-description types + packer to the ABI tables
  scenes.py        the benchmark scenes (Cornell-style, SmokeSphere, triangle mesh)
  render.py        render() / render_distributed() / tonemap — torch only for memory, streams, RCCL
  png.py           dependency-free PNG writer (output stage, main.cpp:57)
"""
from . import abi  # noqa: F401
from .scene import (TextureAtlas, box, camera, checker_texture, constant_medium, dielectric_material,  # noqa: F401
                    image_texture, isotropic_material, lambertian_material, lightsource_material, metal_material,
                    pack, solid_texture, sphere, triangle, xy_rect, xz_rect, yz_rect)

__all__ = ["abi", "TextureAtlas", "box", "camera", "checker_texture", "constant_medium", "dielectric_material",
           "image_texture", "isotropic_material", "lambertian_material", "lightsource_material", "metal_material",
           "pack", "solid_texture", "sphere", "triangle", "xy_rect", "xz_rect", "yz_rect"]

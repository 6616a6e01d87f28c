"""render() — the host-side mirror of the reference's hot-path entry point, over the C ABI.

Reference: `render<width,height,samples>(queue, frame_buf, hittables, cam)` include/render.hpp:141-160.
Here:      `render(width, height, samples, scene, cam, depth=50, ...) -> torch.Tensor [H][W][3]` on the GPU.

PyTorch is plumbing only: it owns the device framebuffer, the current HIP stream and (for N GPUs)
the RCCL process group.  All arithmetic happens in the hand-written gfx950 kernels of
`csrc/pt_render.hip`, reached through `include/pt_render.h`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import abi
from .scene import PackedScene, camera, pack


class DeviceScene:
    """The flattened, HBM-resident scene (pt_scene_create).  Replaces the sycl::buffer wrapping of the
    hittable vector and image_texture::freeze() (render.hpp:146-148); may be reused across renders."""

    def __init__(self, scene: PackedScene, tuning: "abi.PtTuning | None" = None):
        """tuning: an abi.PtTuning (performance-only knobs, include/pt_render.h; abi.tuning(sphere_grid=-1, ...)); None = the
        library's defaults with the PT_* environment applied."""
        self.lib = abi.load_library()
        self.handle = C.c_void_p()
        self._packed = scene  # keep the host tables alive
        if tuning is None:
            abi.check(self.lib.pt_scene_create(C.byref(scene.desc), C.byref(self.handle)), "pt_scene_create")
        else:
            abi.check(self.lib.pt_scene_create_tuned(C.byref(scene.desc), C.byref(tuning), C.byref(self.handle)), "pt_scene_create_tuned")

    def reserve(self, width, height, samples, depth=50, shard_index=0, shard_count=1, flags=0) -> None:
        """pt_scene_reserve: allocate the launch workspaces for these parameters now, so that render() never allocates."""
        p = _params(width, height, samples, depth, shard_index, shard_count, flags)
        abi.check(self.lib.pt_scene_reserve(self.handle, C.byref(p)), "pt_scene_reserve")

    def close(self):
        if self.handle:
            self.lib.pt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def _params(width, height, samples, depth, shard_index=0, shard_count=1, flags=0) -> abi.PtRenderParams:
    return abi.PtRenderParams(int(width), int(height), int(samples), int(depth), int(shard_index), int(shard_count),
                              int(flags), 0)


_DEVICE_SCENE_CACHE = 4  # device scenes kept per PackedScene (one per (device, stream) in use)


def _as_device_scene(scene, cache_key=None) -> DeviceScene:
    """DeviceScene of whatever the caller handed over.  With `cache_key` (the asynchronous torch path) the device scene of a
    PackedScene is kept on it, per (device, stream): no re-flatten / re-upload / hipMalloc per call, and — the point — no temporary
    whose destructor (pt_scene_destroy -> hipFree, an implicit device synchronisation) would run while the kernels it
    launched are still in flight."""
    if isinstance(scene, DeviceScene):
        return scene
    if not isinstance(scene, PackedScene):
        scene = pack(scene)  # a list of hittables, like std::vector<hittable_t>
        cache_key = None
    if cache_key is None:
        return DeviceScene(scene)
    # a small LRU per PackedScene: a device scene holds a full copy of the flattened scene (25 MB for the 100 k-triangle mesh) plus
    # launch workspaces, and the key contains the raw stream handle — programs that create streams on the fly must not pile up a
    # copy per stream that ever existed (a handle reused after its stream died aliases at worst a scene that is still valid: scene
    # data is immutable and its workspaces are only touched in launch order of whichever stream uses them next)
    cache = scene.__dict__.setdefault("_pt_device_scenes", {})
    if cache_key in cache:
        cache[cache_key] = cache.pop(cache_key)  # most recently used last
        return cache[cache_key]
    while len(cache) >= _DEVICE_SCENE_CACHE:
        cache.pop(next(iter(cache)))  # destroyed when the last frame that references it is gone (DeviceScene.__del__)
    cache[cache_key] = DeviceScene(scene)
    return cache[cache_key]


def _stream_ptr(torch) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def render(width: int, height: int, samples: int, scene, cam: camera, depth: int = 50, *, flags: int = 0,
           out=None, shard_index: int = 0, shard_count: int = 1, timed: bool = False):
    """Launch the render kernel on torch's current device/stream; asynchronous like queue.submit
    (render.hpp:151) unless `timed`.  `scene`: a DeviceScene (reuse it across renders), a PackedScene (its device scene is
    created once per device and kept on it) or a list of hittables (packed + uploaded for this call; kept alive by the
    returned tensor).  Returns the framebuffer tensor — [H][W][3] float32, y=0 bottom
    row — or, for shard_count>1, this shard's tiles [tiles][64][3].  With `timed`, returns
    (tensor, kernel_ms) measured with HIP events on the launch stream."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("path_tracer_amd.render needs a HIP device: there is no CPU path in the product")
    lib = abi.load_library()
    # one device scene per (device, stream): a PtScene owns per-scene launch workspaces (tile costs / order / partial sums),
    # so renders on ONE PtScene must be stream-ordered (include/pt_render.h); two streams get two scenes
    ds = _as_device_scene(scene, cache_key=("cuda", torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream))
    p = _params(width, height, samples, depth, shard_index, shard_count, flags)
    n = lib.pt_framebuffer_floats(C.byref(p))
    if n < 0:
        abi.check(abi.PT_ERR_INVALID_ARG, "pt_framebuffer_floats")
    shape = (height, width, 3) if shard_count == 1 else (n // (abi.PT_TILE_PIXELS * 3), abi.PT_TILE_PIXELS, 3)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device="cuda")
    elif out.numel() != n or out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 CUDA tensor of pt_framebuffer_floats() elements")
    if timed:
        ms = C.c_float()
        abi.check(lib.pt_render_timed(ds.handle, C.byref(cam.c), C.byref(p), C.c_void_p(out.data_ptr()),
                                      _stream_ptr(torch), C.byref(ms)), "pt_render_timed")
        return out, float(ms.value)
    abi.check(lib.pt_render(ds.handle, C.byref(cam.c), C.byref(p), C.c_void_p(out.data_ptr()), _stream_ptr(torch)),
              "pt_render")
    out._pt_scene = ds  # a scene built from a list of hittables lives as long as the frame it is still rendering into
    return out


def render_host(width: int, height: int, samples: int, scene, cam: camera, depth: int = 50, *, flags: int = 0,
                shard_index: int = 0, shard_count: int = 1) -> np.ndarray:
    """Torch-free path: pt_render_host renders into a numpy array (allocates, copies back, syncs)."""
    lib = abi.load_library()
    ds = _as_device_scene(scene)
    p = _params(width, height, samples, depth, shard_index, shard_count, flags)
    n = lib.pt_framebuffer_floats(C.byref(p))
    if n < 0:
        abi.check(abi.PT_ERR_INVALID_ARG, "pt_framebuffer_floats")
    fb = np.empty(n, dtype=np.float32)
    abi.check(lib.pt_render_host(ds.handle, C.byref(cam.c), C.byref(p), fb.ctypes.data_as(C.POINTER(C.c_float))),
              "pt_render_host")
    return fb.reshape((height, width, 3) if shard_count == 1 else (-1, abi.PT_TILE_PIXELS, 3))


def unshard(gathered, width: int, height: int, shard_count: int):
    """[shard_count][tiles_per_shard][64][3] (the RCCL-gathered buffer) -> [H][W][3] on the root GPU."""
    import torch

    lib = abi.load_library()
    p = _params(width, height, 1, 1, 0, shard_count)
    fb = torch.empty((height, width, 3), dtype=torch.float32, device=gathered.device)
    abi.check(lib.pt_unshard_tiles(C.c_void_p(gathered.data_ptr()), C.byref(p), C.c_void_p(fb.data_ptr()),
                                   _stream_ptr(torch)), "pt_unshard_tiles")
    return fb


def gather_frame(local, width: int, height: int, group=None, unshard_fn=None):
    """The exchange step of the N-GPU path: ONE gather of every rank's float tiles to rank 0 (RCCL over xGMI
    with the nccl backend; gloo in the CPU tests), then the un-interleave on the root.  `local` is this rank's
    [tiles_per_shard][64][3] tensor.  Returns the [H][W][3] frame on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # one contiguous [world][tiles][64][3] receive buffer, the gather list = its slices (no re-pack before the un-interleave)
    gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device) if rank == 0 else None
    bufs = list(gathered.unbind(0)) if rank == 0 else None
    dist.gather(local, bufs, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank != 0:
        return None
    if world == 1:
        return gathered[0]  # a one-shard render is already [H][W][3]
    return (unshard_fn or unshard)(gathered, width, height, world)


def render_distributed(width: int, height: int, samples: int, scene, cam: camera, depth: int = 50, *,
                       flags: int = 0, group=None, gather: bool = True):
    """One process per GPU (torch.distributed, backend nccl == RCCL).  Tiles are dealt round-robin to
    ranks (tile g -> rank g % world), each rank renders its tiles with the pixels' GLOBAL seeds
    (render.hpp:130-131), so the assembled frame is bit-identical to a single-GPU render.  No collective
    inside the render; one gather of the framebuffer at the end (gather_frame).
    Returns (frame on rank 0 | None elsewhere, local tiles)."""
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        fb = render(width, height, samples, scene, cam, depth, flags=flags)
        return fb, fb
    local = render(width, height, samples, scene, cam, depth, flags=flags, shard_index=rank, shard_count=world)
    if not gather:
        return None, local
    return gather_frame(local, width, height, group), local


def tonemap_rgb8(fb):
    """Output stage of main.cpp:33-59 on the device: sqrt gamma, clamp [0,0.999], x256 -> u8, rows flipped
    (row 0 = top).  Returns a uint8 tensor [H][W][3]."""
    import torch

    lib = abi.load_library()
    h, w, _ = fb.shape
    out = torch.empty((h, w, 3), dtype=torch.uint8, device=fb.device)
    abi.check(lib.pt_tonemap_rgb8(C.c_void_p(fb.data_ptr()), w, h, C.c_void_p(out.data_ptr()), _stream_ptr(torch)),
              "pt_tonemap_rgb8")
    return out

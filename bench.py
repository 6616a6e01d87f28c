"""bench.py — Msamples/s of the render() hot path on BASELINE.json's headline config.

  python bench.py --gpus N --steps K --warmup W          (N > 1 from a bare shell: bench.py starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

Workload (config.workload): configs[1] — Cornell-style scene (7 box + 1 xy_rect, diffuse light), 1920x1080,
1024 spp, depth 50.  One step = one full render of that frame.  N = 1: the frame on one GPU.  N > 1: the SAME
frame (north_star: "a 1920x1080 Cornell-box-style scene at 1024 spp reported at 1/2/4/8 GPUs"), its 8x8 tiles dealt
round-robin to the ranks (no collective inside the render), then one RCCL gather of the float tiles to rank 0 over
xGMI and a device-side un-interleave, both inside the timed step: STRONG scaling, `value` = the frame's samples /
the slowest rank's time.  At this kernel speed the fixed frame is bounded by its heaviest pixel — one sequential RNG
chain — so the line also carries `weak_scaling`: the same measurement on a frame with N x the pixels
(same scene, camera, aspect, spp; width and height x sqrt(N)), i.e. fixed work per GPU.  `--scaling weak` makes that
the headline instead (and says so in `metric`, `scaling` and `config`).
Other BASELINE configs: `--config cfg1` (SmokeSphere 400x225x64, the reference's own CPU case), `--config cfg3` (SmokeSphere 1920x1080x1024), `--config cfg4` (SmokeSphere 3840x2160x4096,
the 8-GPU config), `--config cfg5` (100 k triangles 1920x1080x256); `--scene/--width/--height/--spp` override.
`--mode fast` = the opt-in decorrelated-RNG mode (NOT parity: judged by PSNR; never the default).
The scene is resident in HBM before the timed region (it is ~1 KB; the boundary hands over host tables, and
uploading them costs microseconds — see DESIGN.md).

Extra objects on the JSON line:
  roofline      bound = VALU issue (not HBM, not MFMA).  achieved = algorithmic lane-ops/sample
                (oracle exit-point counters x the per-exit op costs of SURVEY.md §8d) x samples/s of the render
                kernel, measured with HIP events on the launch stream; `executed_over_algorithmic` = the PMC's
                VALU lane-instructions per sample / that figure.
  cpu_baseline  the CPU oracle (kind "port": the reference itself needs triSYCL and cannot be built)
                timed on this host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12  # 78.6 T lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)

# Per-exit algorithmic op costs as written in the reference — SURVEY.md §8(d)'s table (1 op = one fp32
# add/sub/mul/div/cmp/sqrt/cvt or int32 shift/xor; a transcendental call T counts 4).  RNG draws are priced once, by
# the draw counter, and taken out of the per-event figures that §8(d) quotes with their draws included.
T = 4
OPS = dict(
    rng_draw=8,                                   # xorshift.hpp:72-74 + rtweekend.hpp:40-41
    sphere_nodisc=25, sphere_roots_rejected=25 + 12,  # discriminant <= 0 | sqrt + two roots + their comparisons, none inside
    sphere_accept=58 + 2 * T, sphere_moving=12,   # + center(time) per test of a moving sphere (sphere.hpp:51-56)
    rect=(4, 12, 33),                             # t-reject | bounds-reject | accept   (a box = 6 of these)
    tri=(22, 34, 52, 60, 84),                     # |a|<eps | u | v | t-range | accept  (triangle.hpp:71-91)
    medium_extra=25 + T,                          # constant_medium beyond its two boundary tests (its draw: rng_draw)
    camera=91 - 5 * 8, sky=24,
    lambertian=46 - 3 * 8, metal=71 + 4 * T - 3 * 8, dielectric=60 + T, isotropic=41 + 4 * T - 3 * 8, light=4,
    tex=(8 + 3 * T, 0, 20),                       # checker | solid | image (texture.hpp:154 order)
)


# Algorithmic lane-ops per sample of the standard scenes at depth 50: oracle exit-point counters (a 480x270x4 render;
# 96x54x1 for the mesh) priced with OPS above.  Recorded so that ranks of an N>1 job, where the cpu_baseline leg does
# not run, need nothing from oracle/; the N=1 cpu_baseline leg re-derives the figure live and reports that.
ALGORITHMIC_OPS_PER_SAMPLE = {"cornell": 2062.7, "smoke": 39448.4, "triangles": 8220663.0}  # (smoke: round 6's scene — main.cpp:83 in g++'s argument order)  # the REFERENCE's algorithm as written (never the culled figures below)

# The 496-hittable scene does NOT run the reference's algorithm as written: 476 of its 489 spheres sit in an exact culling grid
#  and a ray tests the spheres of the cells it crosses instead of all of them.  Pricing the kernel against the
# reference's 483 sphere tests per ray gave "fractions" above 1; the roofline of such a kernel is priced for the algorithm it
# runs: everything but the gridded spheres as counted by the oracle, plus the grid walk as counted IN the kernel
# (profiles/r03_smoke_walk_counters.json: diagnostic build, tools/stamps.py): cells visited x the ops of a DDA step, walks x the
# ops of a walk's set-up, grid sphere tests priced like the oracle's sphere tests.
def counted_in_kernel(kind: str, scene: str, profiles_dir=None):
    """The in-kernel counters a culled algorithm is priced with — `kind` "walk" (sphere grid: profiles/*_walk_counters.json, written
    by tools/stamps.py with PT_WALK_JSON) or "tripool" (triangle pool: profiles/*_tripool_counters.json, tools/tri_counters.py with
    PT_TRI_JSON) — selected like the PMC summaries: only records marked `"final": true`, the highest `"round"` wins.  Returns the
    record with its file name under "source", or None (the kernel is then priced as the reference's algorithm as written)."""
    best, best_round = None, -1
    for f in (Path(profiles_dir) if profiles_dir else ROOT / "profiles").glob(f"*_{kind}_counters.json"):
        try:
            d = json.loads(f.read_text())
        except Exception:  # noqa: BLE001
            continue
        if d.get("final") and d.get("scene") == scene and int(d.get("round", 0)) > best_round:
            best, best_round = dict(d, source=f"profiles/{f.name}"), int(d.get("round", 0))
    return best


def grid_walk_counters(scene: str, profiles_dir=None):
    d = counted_in_kernel("walk", scene, profiles_dir)
    if not d:
        return None
    return dict(cells_per_sample=d["per_sample"]["cells_visited"], tests_per_sample=d["per_sample"]["grid_sphere_tests"], source=d["source"])


GRID_WALK = {"smoke": grid_walk_counters("smoke")}
OPS_GRID = dict(step=19,    # min3 (2) + axis select (3) + index step (3) + bounds (3) + boundary update (3) + header decode (2) + limit (3)
                setup=45)   # origin and reciprocal direction in cell units (9), slab clip (14), entry cell (12), first boundaries (9), sign selects (1)


def ops_per_sample_culled(ctr: dict, n_spheres: int, grid_spheres: int, walk: dict) -> float:
    """Algorithmic lane-ops per sample of the CULLED algorithm: ops_per_sample() with the sphere part replaced.  The oracle's
    sphere counters cover all spheres; the kernel tests the (n_spheres - grid_spheres) listed ones for every ray and the
    gridded ones per cell.  Sphere accepts are priced as counted (a hit is a hit in any order); every other sphere test — the
    listed ones of every ray and the walk's — at the oracle's own mix of its two rejecting exits; moving-sphere extras in
    proportion."""
    n = ctr["samples"]
    se = ctr["sphere_exit"]
    brute_spheres = (se[0] * OPS["sphere_nodisc"] + se[1] * OPS["sphere_roots_rejected"] + se[2] * OPS["sphere_accept"]
                     + ctr["sphere_moving"] * OPS["sphere_moving"])
    rest = ops_per_sample(ctr) * n - brute_spheres
    reject_price = (se[0] * OPS["sphere_nodisc"] + se[1] * OPS["sphere_roots_rejected"]) / max(1, se[0] + se[1])
    moving_share = ctr["sphere_moving"] / max(1, sum(se))
    tests = ctr["rays"] * (n_spheres - grid_spheres) + walk["tests_per_sample"] * n
    spheres = (se[2] * OPS["sphere_accept"] + max(0.0, tests - se[2]) * reject_price + tests * moving_share * OPS["sphere_moving"])
    walk_ops = walk["cells_per_sample"] * n * OPS_GRID["step"] + ctr["rays"] * OPS_GRID["setup"]
    return (rest + spheres + walk_ops) / n


# The 100 k-triangle mesh (cfg5) likewise: since round 3 its long triangle run is culled exactly by a triangle pool —
# per ray the kernel runs the reference's test on the candidates of the grid cells it crosses (round 5: minus the ones the previous
# cell listed too), the band test on the records of its direction's bin in the direction map, the noise-radius filter on the
# survivors and the reference's test on what is left.  Counted in the kernel (diagnostic build `make stamps EXTRA=-DPT_STAMPS_TRI`,
# tools/tri_counters.py -> profiles/r05_tripool_counters.json).
def tri_pool_counters(scene: str, profiles_dir=None):
    d = counted_in_kernel("tripool", scene, profiles_dir)
    if not d:
        return None
    r = d["per_ray"]
    return dict(exact_per_ray=r["exact_tests"], grid_filter_per_ray=r["grid_filter_tests"], band_per_ray=r["band_tests"] + r["always_tests"],
                noise_per_ray=r["noise_radius_tests"], cells_per_ray=r["grid_cells"], source=d["source"])


TRI_POOL = {"triangles": tri_pool_counters("triangles")}
OPS_TRI_POOL = dict(band=8,        # d . n (5) + |.| + rho + c, compare (3)
                    grid_filter=20,  # (rounds 3-4: the grid candidates' line test; the round-5 pool has no such stage)
                    noise=33,      # the noise-radius filter of a pair past the band test: |a'| - ea |d| (4), radius (10), line test (19)
                    setup=45 + 20)   # the grid walk's set-up + the direction-map bin of the ray (face, two quotients, two floors)


def ops_per_sample_culled_tri(ctr: dict, pool: dict) -> float:
    """ops_per_sample() with the triangle part replaced by what the pool runs: exact tests priced at the oracle's own mix of the
    triangle test's five exits (the candidates are the triangles NEAR the ray, so this under-prices them if anything), band
    records at OPS_TRI_POOL['band'], grid candidates' line test at OPS_TRI_POOL['grid_filter'], grid steps as the sphere grid's."""
    n = ctr["samples"]
    te = ctr["tri_exit"]
    brute_tri = sum(c * p for c, p in zip(te, OPS["tri"]))
    rest = ops_per_sample(ctr) * n - brute_tri
    tri_price = brute_tri / max(1, sum(te))
    per_ray = (pool["exact_per_ray"] * tri_price + pool["band_per_ray"] * OPS_TRI_POOL["band"]
               + pool.get("grid_filter_per_ray", 0.0) * OPS_TRI_POOL["grid_filter"] + pool.get("noise_per_ray", 0.0) * OPS_TRI_POOL["noise"]
               + pool["cells_per_ray"] * OPS_GRID["step"] + OPS_TRI_POOL["setup"])
    return (rest + ctr["rays"] * per_ray) / n


# recorded like ALGORITHMIC_OPS_PER_SAMPLE (the N = 1 cpu_baseline leg re-derives them live): the culled algorithms' figures
ALGORITHMIC_OPS_PER_SAMPLE_CULLED = {"smoke": 2072.8, "triangles": 88771.3}  # (round 6: profiles/r06_smoke_walk_counters.json, r06_tripool_counters.json — camera rays take their grazing candidates from the pixel's cache, the grid's slack re-derived and its cells listed exactly: 185 pair tests per ray after 581; round 5: 2 116.7 / 205 k; rounds 3-4: 402 k)


def ops_per_sample(ctr: dict) -> float:
    """Algorithmic lane-ops per sample from the oracle's exit-point counters, priced per exit."""
    n = ctr["samples"]
    dot = lambda counts, prices: sum(c * p for c, p in zip(counts, prices))  # noqa: E731
    se = ctr["sphere_exit"]
    total = (ctr["rng_draws"] * OPS["rng_draw"]
             + se[0] * OPS["sphere_nodisc"] + se[1] * OPS["sphere_roots_rejected"] + se[2] * OPS["sphere_accept"]
             + ctr["sphere_moving"] * OPS["sphere_moving"]
             + dot(ctr["rect_exit"], OPS["rect"])
             + dot(ctr["tri_exit"], OPS["tri"])
             + ctr["tests"][4] * OPS["medium_extra"]
             + n * OPS["camera"] + ctr["end_sky"] * OPS["sky"]
             + ctr["scatters"][0] * OPS["lambertian"] + ctr["scatters"][1] * OPS["metal"]
             + ctr["scatters"][2] * OPS["dielectric"] + ctr["scatters"][3] * OPS["light"]
             + ctr["scatters"][4] * OPS["isotropic"]
             + dot(ctr["tex_evals"], OPS["tex"]))
    return total / n


KERNEL_SOURCES = ("pt_render.hip", "pt_device.hpp", "pt_math.hpp", "pt_flatten.hpp", "pt_tripool.hpp", "pt_binned.hpp")  # csrc/Makefile: KSRC, same order


def kernels_sha16(root=None):
    """sha256 over the sources the render kernels are compiled from: identifies the build a PMC recording belongs to (tools/pmc_summary.py
    writes the same hash into every summary).  The PMC-derived fields of the bench line are RECORDINGS (profiles/*_pmc_summary.json), not
    measurements of this run: when the recording's hash differs from the tree's they are nulled and the line says so."""
    import hashlib
    hsh = hashlib.sha256()
    for name in KERNEL_SOURCES:
        hsh.update(((Path(root) if root else ROOT) / "path_tracer_amd" / "csrc" / name).read_bytes())
    return hsh.hexdigest()[:16]


def cgroup_cpu_max():
    """The container's CPU quota in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable: a box that shows 256
    CPUs may let this process use a dozen — which is what a thread-scaling efficiency of 0.08 on 128 OpenMP threads then means."""
    try:
        txt = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if txt and txt[0] != "max":
            return round(int(txt[0]) / int(txt[1]), 2)
        return None
    except Exception:  # noqa: BLE001
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            p = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            return round(q / p, 2) if q > 0 else None
        except Exception:  # noqa: BLE001
            return None


def pmc_traffic(scene: str, w: int, h: int, spp: int, profiles_dir=None):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC summary of the same workload
    (FETCH_SIZE and WRITE_SIZE are collected in their own --pmc passes: tools/pmc_summary.py).
    Selection is EXPLICIT, never by file name: only summaries that carry `"final": true` (written by
    `tools/pmc_summary.py --final ROUND`: the counters of a round's final build) qualify, and among those the highest
    `"round"` wins; two finals of one round for one workload are an error.  Returns (bytes, file name, derived dict)."""
    best, best_round = None, None
    for f in (Path(profiles_dir) if profiles_dir else ROOT / "profiles").glob("*_pmc_summary.json"):
        try:
            d = json.loads(f.read_text())
        except Exception:  # noqa: BLE001
            continue
        if not d.get("final") or d.get("workload") != f"{w}x{h}x{spp}" or d.get("scene", "cornell") != scene:
            continue
        if "hbm_bytes_per_launch" not in d.get("derived", {}):
            continue
        rnd = int(d.get("round", 0))
        if best is not None and rnd == best_round:
            raise RuntimeError(f"two final PMC summaries of round {rnd} for {scene} {w}x{h}x{spp}: {best[1]} and {f.name}")
        if best is None or rnd > best_round:
            best, best_round = (d["derived"]["hbm_bytes_per_launch"], f.name, d["derived"], d.get("kernels_sha16"), d.get("recorded_at_head")), rnd
    return best


def predicted_chain_floor_ms(scene: str, w: int, h: int, spp: int, n_gpus: int, profiles_dir=None):
    """Kernel ms of shard 0 of `n_gpus` of this frame as measured on ONE GPU (tools/shard_table.py -> profiles/*_shard_table_*.json,
    final-marked like the PMC summaries): the step time an N-GPU parity run cannot beat — a shard is bounded by its heaviest
    pixel's sequential RNG chain, so strong scaling flattens where this figure stops falling."""
    best, best_round = None, -1
    for f in (Path(profiles_dir) if profiles_dir else ROOT / "profiles").glob("*_shard_table_*.json"):
        try:
            d = json.loads(f.read_text())
        except Exception:  # noqa: BLE001
            continue
        if d.get("final") and d.get("scene") == scene and d.get("workload") == f"{w}x{h}x{spp}" and str(n_gpus) in d.get("parity", {}):
            if int(d.get("round", 0)) > best_round:
                best, best_round = (d["parity"][str(n_gpus)], f.name), int(d.get("round", 0))
    return best


def weak_frame(width: int, height: int, n_gpus: int):
    """Frame of an N-GPU weak-scaling step: N x the pixels of the 1-GPU frame, aspect kept (every rank gets one 1-GPU
    frame's worth of 8x8 tiles).  1920x1080 -> 2715x1527, 3840x2160, 5431x3055 for N = 2, 4, 8."""
    return int(round(width * n_gpus ** 0.5)), int(round(height * n_gpus ** 0.5))


CONFIGS = {  # BASELINE.json `configs`
    "cfg1": dict(scene="smoke", width=400, height=225, spp=64),          # the reference's own CPU-runnable case (main.cpp defaults)
    "cfg2": dict(scene="cornell", width=1920, height=1080, spp=1024),    # the headline: what `metric` is quoted on
    "cfg3": dict(scene="smoke", width=1920, height=1080, spp=1024),
    "cfg4": dict(scene="smoke", width=3840, height=2160, spp=4096),      # the 8-GPU tile-sharded config
    "cfg5": dict(scene="triangles", width=1920, height=1080, spp=256),
}
SCENE_TEXT = {"cornell": "Cornell-style 7 box + 1 xy_rect + diffuse light",
              "smoke": "SmokeSphere scene of main.cpp:67-161 (496 hittables, the reference's two image textures)",
              "triangles": "100 k triangles + ground sphere + emissive xy_rect"}
PT_FLAG_FAST_RNG = 1 << 9  # include/pt_render.h


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` from a bare shell: start the N ranks as children (torch.distributed.run) and relay
    rank 0's JSON line.  Runs BEFORE anything in this process touches the GPU, and nothing is exec()ed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    if os.environ.get("PT_BENCH_DRY_LAUNCH"):  # testing aid (no GPU here): show what would be started
        print(json.dumps({"launch": cmd}))
        return 0
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    result = [ln for ln in lines if ln.lstrip().startswith('{"metric"')]
    for ln in lines:
        if ln not in result:
            print(ln, file=sys.stderr)
    if result:
        print(result[-1], flush=True)  # the JSON line is the last line on stdout
    return proc.returncode if proc.returncode else (0 if result else 1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS), help="BASELINE.json config (default: the headline)")
    ap.add_argument("--scene", default=None, choices=["cornell", "smoke", "triangles"])
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--mode", choices=["parity", "fast"], default="parity",
                    help="fast = opt-in decorrelated per-(pixel, sample) RNG streams: NOT the reference's image bit for bit")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1 headline: strong = the fixed frame on N GPUs (default; the other one is reported beside it), "
                         "weak = a frame with N x the pixels (same aspect)")
    ap.add_argument("--dist-single", action="store_true",
                    help="testing aid: run the N>1 code path (RCCL process group, sharded render, gather) with world size 1")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("scene", "width", "height", "spp"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    if args.mode == "fast":
        args.flags |= PT_FLAG_FAST_RNG

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    from path_tracer_amd import render as R
    from path_tracer_amd import scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dist_path = world > 1 or args.dist_single
    if dist_path:
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()  # creates the communicator now
        # With NCCL_DEBUG=VERSION (exported by the harness) RCCL printf()s a version banner into C stdio's buffer,
        # which would otherwise be flushed at exit, AFTER the result: push it out now so that the JSON line is the
        # last line on stdout.
        import ctypes
        ctypes.CDLL(None).fflush(None)

    scene_name, W1, H1, SPP, DEPTH = cfg["scene"], cfg["width"], cfg["height"], cfg["spp"], args.depth
    kw = {"n_triangles": 100_000} if scene_name == "triangles" else {}
    packed, cam_args = scenes.build(scene_name, **kw)
    # The scene is resident in HBM before the timed region — what that costs is on the line all the same (cfg5's 3 s of host
    # work and 2.9 GB of tables were invisible): pt_scene_create = validate + flatten + build the culling structures + upload.
    torch.cuda.synchronize()
    t_build = time.perf_counter()
    ds = R.DeviceScene(packed)
    torch.cuda.synchronize()
    scene_build_first_s = time.perf_counter() - t_build  # the first one of the process also pays the HIP runtime's one-off costs (its fill / copy kernels are loaded on first use: ~0.1 s)
    ds.close()
    t_build = time.perf_counter()
    ds = R.DeviceScene(packed)
    torch.cuda.synchronize()
    scene_build_s = time.perf_counter() - t_build
    from path_tracer_amd import abi as _abi
    scene_device_bytes = int(_abi.load_library().pt_scene_device_bytes(ds.handle))
    library_build_id = _abi.load_library().pt_build_id().decode()

    def barrier():
        if dist_path:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(W, H, steps, warmup):
        """`warmup` untimed + exactly `steps` timed steps of the W x H frame; returns (seconds, kernel ms, last frame)."""
        cam = scenes.make_camera(cam_args, W, H)
        kernel_ms = []
        exchange = []  # (start, end) events around gather + un-interleave of every step (torch's current stream waits for the collective)
        fb = None
        # launch workspaces sized before the timed region, so that no timed step allocates (also with --warmup 0)
        ds.reserve(W, H, SPP, DEPTH, rank if dist_path else 0, world if dist_path else 1, args.flags)

        def step():
            if not dist_path:
                fb, ms = R.render(W, H, SPP, ds, cam, DEPTH, flags=args.flags, timed=True)
                kernel_ms.append(ms)
                return fb
            local, ms = R.render(W, H, SPP, ds, cam, DEPTH, flags=args.flags, shard_index=rank, shard_count=world, timed=True)
            kernel_ms.append(ms)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = R.gather_frame(local, W, H)  # one RCCL gather of the float tiles to rank 0 + un-interleave
            e1.record()
            exchange.append((e0, e1))
            return out

        for _ in range(warmup):
            step()
        kernel_ms.clear()
        exchange.clear()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fb = step()
        barrier()
        elapsed = time.perf_counter() - t0
        kern = sum(kernel_ms) / max(1, len(kernel_ms))
        detail = None
        if dist_path:
            xch = sum(a.elapsed_time(b) for a, b in exchange) / max(1, len(exchange))  # ms per step on this rank
            mine = torch.tensor([kern, xch], dtype=torch.float64, device="cuda")
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)  # every rank's own figures: the slowest rank sets the step, the others show the balance
            detail = {"kernel_ms_per_rank": [round(float(t[0]), 3) for t in every],
                      "gather_unshard_ms_per_rank": [round(float(t[1]), 3) for t in every],
                      "gather_unshard_ms_root": round(float(every[0][1]), 3)}
            t = torch.tensor([elapsed, kern], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the slowest rank's clock
            elapsed, kern = float(t[0]), float(t[1])
        return elapsed, kern, fb, cam, detail

    Ww, Hw = weak_frame(W1, H1, world)
    W, H = (Ww, Hw) if (world > 1 and args.scaling == "weak") else (W1, H1)
    elapsed, kern_ms, fb, cam, dist_detail = measure(W, H, args.steps, args.warmup)
    other = None
    if world > 1:  # the other scaling mode, reported beside the headline (same steps / warmup)
        Wo, Ho = (W1, H1) if args.scaling == "weak" else (Ww, Hw)
        eo, ko, _, _, _ = measure(Wo, Ho, args.steps, args.warmup)
        other = {"scaling": "strong" if args.scaling == "weak" else "weak", "frame": f"{Wo}x{Ho}",
                 "value": round(Wo * Ho * SPP * args.steps / eo / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(eo / args.steps * 1e3, 3), "kernel_ms": round(ko, 3)}

    if rank == 0:
        samples_per_step = W * H * SPP
        value = samples_per_step * args.steps / elapsed / 1e6
        ops = ALGORITHMIC_OPS_PER_SAMPLE[scene_name]
        ops_culled = ALGORITHMIC_OPS_PER_SAMPLE_CULLED.get(scene_name)
        # which algorithm the kernel runs is read off the scene as pt_scene_create(desc) flattens it — the library's defaults with the PT_*
        # variables applied, which is how THIS program creates its scene (pt_debug_tri_pool: spheres in a culling grid, triangles in a pool);
        # a caller that creates scenes with an explicit PtTuning would have to ask pt_debug_flatten_tuned instead
        import ctypes as C_
        from path_tracer_amd import abi as abi_
        st_scene = (C_.c_int32 * 8)()
        abi_.check(abi_.load_library().pt_debug_tri_pool(C_.byref(packed.desc), st_scene), "pt_debug_tri_pool")
        grid_culled, tri_culled = st_scene[7] > 0, st_scene[0] > 0
        if (scene_name == "smoke" and not grid_culled) or (scene_name == "triangles" and not tri_culled):
            ops_culled = None  # the A/B knobs select the reference's algorithm as written
        cpu_line = None
        if world == 1 and not args.no_cpu_baseline:
            # --- cpu_baseline leg: the only place bench.py touches oracle/ (test infrastructure) ---------------
            from oracle import binding as orc
            orc.set_math(True)
            cw, ch, cs = (480, 270, 4) if scene_name != "triangles" else (64, 36, 1)
            _, ctr = orc.render(packed, scenes.make_camera(cam_args, cw, ch).c, cw, ch, cs, DEPTH, counters=True)
            ops = ops_per_sample(ctr.as_dict())  # exit-point counters -> algorithmic ops per sample, live
            if GRID_WALK.get(scene_name) and grid_culled:
                st = st_scene
                ops_culled = ops_per_sample_culled(ctr.as_dict(), sum(1 for k in packed.kinds() if k == abi_.PT_HIT_SPHERE), st[7], GRID_WALK[scene_name])
            if TRI_POOL.get(scene_name) and tri_culled:
                ops_culled = ops_per_sample_culled_tri(ctr.as_dict(), TRI_POOL[scene_name])
            # bounded sample of the same workload.  Round 6: ~3 s of OpenMP work, ~1 s each for the glibc-math and the
            # one-thread legs — the whole CPU leg now costs about what the GPU region does (it was 22 s against 3: the driver's gpu_busy
            # sampler saw an idle device on all of its samples).  The rate does not depend on the sample's size (Msamples/s is resolution-
            # independent: SURVEY.md §8d), only its noise does.
            bw, bh = (W // 2, H // 2) if scene_name != "triangles" else (160, 90)
            bcam = scenes.make_camera(cam_args, bw, bh)
            # threads: what the container may actually run.  A box that shows 256 CPUs under a 16-CPU cgroup quota ran 128 OpenMP threads in
            # bursts (a one-sample probe at 65 Msamples/s, the sample it sized at 10: 19.6 s instead of 3); the quota, rounded up, is the
            # thread count now, and the sample is sized from a second, longer probe.
            quota = cgroup_cpu_max()
            if quota:
                orc.load().orc_set_threads(int(max(1, min(orc.load().orc_max_threads(), math.ceil(quota)))))
            t1 = time.perf_counter()
            orc.render(packed, bcam.c, bw, bh, 1, DEPTH)
            probe = time.perf_counter() - t1
            ps = int(max(1, min(SPP, round(0.4 / max(probe, 1e-3)))))
            if ps > 1:
                t1 = time.perf_counter()
                orc.render(packed, bcam.c, bw, bh, ps, DEPTH)
                probe = (time.perf_counter() - t1) / ps
            bs = int(max(1, min(SPP, round(2.6 / max(probe, 1e-3)))))
            t1 = time.perf_counter()
            orc.render(packed, bcam.c, bw, bh, bs, DEPTH)
            dt = time.perf_counter() - t1
            # the same sample with glibc's float libm (the reference's own libm semantics; transcendental-free scenes such as
            # the Cornell-style one give the same image either way), half the sample: it rides beside, it is not `value`
            orc.set_math(False)
            gs = max(1, bs // 3)
            t1 = time.perf_counter()
            orc.render(packed, bcam.c, bw, bh, gs, DEPTH)
            dtg = time.perf_counter() - t1
            orc.set_math(True)
            # one thread on its own, on a slice of the same workload sized for ~1 s (a smaller frame of the same camera where even one
            # sample per pixel of the sample's frame would take longer: the 100 k-triangle mesh does ~40 samples/s per thread): what a
            # "core" of this host is worth, and how the OpenMP run scales over the threads it used (row-dynamic schedule)
            nthreads = orc.load().orc_max_threads()
            per_thread_guess = bw * bh * bs / dt / max(nthreads, 1)              # samples / s / thread if the scaling were perfect
            want = max(64.0, 1.0 * per_thread_guess)                             # samples for ~1 s (a thread alone is at least that fast)
            s1 = int(max(1, min(bs, want // (bw * bh))))
            w1, h1 = bw, bh
            if want < bw * bh:                                                   # fewer samples than pixels: shrink the frame
                k = (want / (bw * bh)) ** 0.5
                w1, h1 = max(8, int(bw * k)), max(8, int(bh * k))
            cam1 = scenes.make_camera(cam_args, w1, h1)
            orc.load().orc_set_threads(1)
            t1 = time.perf_counter()
            orc.render(packed, cam1.c, w1, h1, s1, DEPTH)
            dt1 = time.perf_counter() - t1
            orc.load().orc_set_threads(nthreads)
            one_thread = w1 * h1 * s1 / dt1 / 1e6
            cpu_line = {"value": round(bw * bh * bs / dt / 1e6, 3), "unit": "Msamples/s",
                        "cores": nthreads, "kind": "port",
                        "affinity_cpus": len(os.sched_getaffinity(0)), "os_cpu_count": os.cpu_count(), "cgroup_cpu_max": cgroup_cpu_max(),
                        "one_thread_value": round(one_thread, 4), "one_thread_sample": f"{w1}x{h1}, {s1} spp ({dt1:.1f} s), 1 OpenMP thread",
                        "thread_scaling_efficiency": round(bw * bh * bs / dt / 1e6 / (one_thread * max(nthreads, 1)), 3),
                        "sample": f"same scene, {bw}x{bh}, {bs} spp, depth {DEPTH} ({bw * bh * bs / 1e6:.1f} Msamples, {dt:.1f} s), OpenMP CPU oracle, portable math",
                        "value_glibc_math": round(bw * bh * gs / dtg / 1e6, 3),
                        "sample_glibc_math": f"{bw}x{bh}, {gs} spp ({dtg:.1f} s), same oracle with glibc's float libm"}
            # the "PSNR vs CPU ref" half of the metric: sampled pixels of the LAST timed GPU frame re-rendered by the
            # oracle at full spp (a pixel depends only on its own RNG stream)
            import numpy as np
            rng_ = np.random.default_rng(1)
            npx = 256 if scene_name == "cornell" else 24 if scene_name == "smoke" else 4
            xy = np.stack([rng_.integers(0, W, npx), rng_.integers(0, H, npx)], axis=1).astype(np.int32)
            ref = orc.render_pixels(packed, cam.c, W, H, SPP, xy, DEPTH)
            got = fb.cpu().numpy()[xy[:, 1], xy[:, 0]]
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            g8, r8 = orc.tonemap_rgb8(got[None]), orc.tonemap_rgb8(ref[None])
            mse = float(np.mean((g8.astype(np.float64) - r8.astype(np.float64)) ** 2))
            parity = {"pixels_checked": int(npx), "bit_identical_pixels": int(same.all(axis=1).sum()),
                      "psnr_db_8bit": None if mse == 0 else round(10 * np.log10(255.0 ** 2 / mse), 2),
                      "note": "GPU frame vs CPU oracle (portable math) at sampled pixels, full spp; null PSNR = identical"
                              + ("; fast mode is NOT expected to be bit-identical" if args.mode == "fast" else "")}
        # the committed counters are those of the parity kernels with default flags: any other mode / flag set gets nulls
        pmc = pmc_traffic(scene_name, W, H, SPP) if (world == 1 and args.flags == 0) else None
        # ... and they are RECORDINGS: they describe the build they were recorded on.  A recording of another build (its kernel-source hash
        # differs from this tree's) is not reported as if it were this run's: every PMC-derived field is nulled, the line says which
        # recording was refused.  (tools/pmc_summary.py stamps kernels_sha16 / recorded_at_head; older summaries carry neither = stale.)
        # "this build" is the LIBRARY that ran (pt_build_id: the hash of the kernel sources it was compiled from, csrc/Makefile), not the
        # tree beside it (a stale .so or a PT_RENDER_LIB override would otherwise pass for the tree's build)
        kernels_now = library_build_id
        pmc_record = {"file": pmc[1], "kernels_sha16": pmc[3], "recorded_at_head": pmc[4], "matches_this_build": pmc[3] == kernels_now} if pmc else None
        if pmc and pmc[3] != kernels_now:
            pmc = None
        kernel_samples_per_s = samples_per_step / (kern_ms * 1e-3)  # whole job; each rank renders 1/world of it
        ops_reference = ops
        if ops_culled:  # a kernel that provably skips tests is priced for the algorithm it runs; the reference's figure rides beside
            ops = ops_culled
        achieved = ops * kernel_samples_per_s / 1e12 / world  # per GPU
        # algorithmic HBM bytes of one launch: the frame written once (12 B per pixel of this rank's share) + the scene read once
        # (the flattened blob with its culling tables, the material table, the texture atlas)
        # (the triangle pools' tables are a buffer of their own since round 5: pt_debug_flatten_pool)
        # (round 6: what the scene occupies on the device, from the scene itself — pt_scene_device_bytes — instead of a second flatten of it)
        scene_bytes = scene_device_bytes
        algorithmic_bytes = W * H * 12 // world + scene_bytes
        valu_frac = achieved / PEAK_TLANEOPS
        # memory leg: bytes that went past L2 (FETCH_SIZE + WRITE_SIZE of the committed PMC passes of this workload) / kernel time / 8 TB/s
        mem_frac = (pmc[0] / (kern_ms * 1e-3) / 8e12) if pmc else None
        bound = "hbm" if (mem_frac is not None and mem_frac > valu_frac) else "valu"
        headline = (scene_name, W1, H1, SPP) == ("cornell", 1920, 1080, 1024)
        at = "1080p 1024spp" if (W, H, SPP) == (1920, 1080, 1024) else f"{W}x{H} {SPP}spp"
        scaling = args.scaling  # (at N = 1 both modes coincide; the label stays the one the N > 1 lines of the same sweep carry)
        line = {
            "metric": f"Msamples/s (W x H x spp / s) at {at}" + ("" if headline else f" [{args.config}: {scene_name}]")
                      + (" [fast mode: decorrelated RNG, not parity]" if args.mode == "fast" else ""),
            "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "frame": f"{W}x{H}",
            # outside the timed region, and therefore said: seconds of pt_scene_create (flatten + culling structures + upload) and the
            # device bytes of the scene's data (records, materials, triangle-pool tables, atlas)
            "scene_build_s": round(scene_build_s, 4), "scene_build_first_in_process_s": round(scene_build_first_s, 4), "scene_device_bytes": scene_device_bytes,
            "config": {"workload": f"{scene_name}: {SCENE_TEXT[scene_name]}, {W}x{H}, {SPP} spp, depth {DEPTH}, "
                                   + ("seed = pixel linear id" if args.mode == "parity" else "FAST MODE: one RNG stream per (pixel, sample)"),
                       "baseline_config": args.config, "hittables": packed.n_hittables, "mode": args.mode,
                       "frame_at_1_gpu": f"{W1}x{H1}",
                       "sharding": "whole frame" if world == 1 else f"8x8 tiles round-robin over {world} ranks + RCCL gather"},
            # `bound`: the larger of the VALU fraction (algorithmic lane-ops / s over the issue peak) and the memory fraction (bytes past
            # L2 per second over 8 TB/s).  achieved / peak / unit / frac describe THAT leg; the other one rides in "valu" / "hbm".
            "roofline": {"bound": bound,
                         "achieved": round(achieved, 3) if bound == "valu" else round(pmc[0] / (kern_ms * 1e-3) / 1e9, 1),
                         "peak": round(PEAK_TLANEOPS, 1) if bound == "valu" else 8000.0,
                         "unit": "Tlaneop/s" if bound == "valu" else "GB/s",
                         "frac": round(valu_frac if bound == "valu" else mem_frac, 4),
                         "valu": {"achieved_tlaneops": round(achieved, 3), "peak_tlaneops": round(PEAK_TLANEOPS, 1), "frac": round(valu_frac, 4)},
                         # `achieved` prices the algorithm the kernel RUNS: the reference's as written, except where an exact
                         # culling structure provably skips tests (sphere grid: ops_per_sample_culled); the reference's own
                         # figure rides beside as algorithmic_ops_per_sample_reference
                         "frac_note": ("exceeds 1: the kernel skips tests the pricing still counts" if achieved / PEAK_TLANEOPS > 1 else None),
                         "priced_algorithm": (("culled: " + ("sphere grid" if GRID_WALK.get(scene_name) else "triangle pool") + " (counted in-kernel: "
                                               + (GRID_WALK.get(scene_name) or TRI_POOL.get(scene_name) or {"source": "recorded figure"})["source"] + ")")
                                              if ops_culled else "the reference's algorithm as written"),
                         "algorithmic_ops_per_sample_reference": round(ops_reference, 1),
                         "traffic": pmc[0] if pmc else None, "traffic_source": pmc[1] if pmc else None,
                         # which recording the PMC-derived fields (traffic, hbm, *_pmc, executed_over_algorithmic) come from; nulled when stale
                         "pmc_recording": pmc_record, "kernels_sha16": kernels_now, "tree_kernels_sha16": kernels_sha16(), "recorded_at_head": pmc_record["recorded_at_head"] if pmc_record else None,
                         # north-star evidence: HBM is not the limiter, VALU issue is busy (PMC of the committed profile)
                         "hbm": {"achieved_gbs": round(pmc[0] / (kern_ms * 1e-3) / 1e9, 3), "peak_gbs": 8000.0,
                                 "frac": round(mem_frac, 6), "traffic_over_algorithmic": round(pmc[0] / algorithmic_bytes, 2)} if pmc else None,
                         "valu_issue_occupancy_pmc": round(pmc[2].get("valu_issue_occupancy", 0.0), 3) if pmc else None,
                         "valu_lane_utilisation_pmc": round(pmc[2].get("valu_lane_utilisation", 0.0), 3) if pmc else None,
                         "issue_slot_occupancy_pmc": round(pmc[2]["issue_slot_occupancy"], 3) if pmc and "issue_slot_occupancy" in pmc[2] else None,
                         "valu_lane_instr_per_sample_pmc": round(pmc[2].get("valu_lane_instr_per_sample", 0.0), 1) if pmc else None,
                         # how much more the kernel executes than the reference's arithmetic as written
                         "executed_over_algorithmic": round(pmc[2]["valu_lane_instr_per_sample"] / ops, 3)
                         if pmc and pmc[2].get("valu_lane_instr_per_sample") else None,
                         "kernel": "render_kernel", "kernel_ms": round(kern_ms, 3),
                         "algorithmic_ops_per_sample": round(ops, 1),
                         "kernel_msamples_per_s_per_gpu": round(kernel_samples_per_s / world / 1e6, 2),
                         "hbm_algorithmic_bytes": algorithmic_bytes, "hbm_algorithmic_bytes_scene": scene_bytes},
        }
        if dist_path:
            # the N > 1 line explains itself: every rank's kernel time, the exchange step on its own, what the collective layer reports,
            # and (parity mode) the shard's own chain floor as measured on ONE GPU — where strong scaling must flatten
            line["distributed"] = dict(dist_detail or {}, backend=dist.get_backend(), world_size=dist.get_world_size(),
                                       rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()),
                                       # what THIS path issues (render.py: gather_frame): torch.distributed.gather, which the NCCL backend runs as
                                       # grouped point-to-point sends / receives — every peer -> root transfer on its own xGMI link, like the one
                                       # ncclGather the C ABI's pt_dist_gather_frame issues (csrc/pt_dist.cpp), which this Python path does not call
                                       exchange="torch.distributed.gather of float tiles to rank 0 (NCCL backend: RCCL grouped send / recv) + device un-interleave (pt_unshard_tiles)",
                                       bytes_gathered_per_rank=int(((W + 7) // 8) * ((H + 7) // 8) + world - 1) // world * 64 * 12)
            if args.mode == "parity":
                floor = predicted_chain_floor_ms(scene_name, W1, H1, SPP, world)
                line["predicted_chain_floor_ms"] = floor[0] if floor else None
                line["predicted_chain_floor_source"] = floor[1] if floor else None
        if other:
            line[other["scaling"] + "_scaling"] = other
        if cpu_line:
            line["cpu_baseline"] = cpu_line
            line["parity"] = parity
        print(json.dumps(line), flush=True)
    if dist_path:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

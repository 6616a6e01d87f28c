"""Reads the s_memtime shares of a diagnostic build (csrc built with -DPT_STAMPS -> libpt_stamps.so).
    PT_RENDER_LIB=path_tracer_amd/libpt_stamps.so python tools/stamps.py cornell 256"""
import ctypes as C, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from path_tracer_amd import abi, scenes
from path_tracer_amd import render as R
scene, spp = sys.argv[1], int(sys.argv[2])
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = abi.load_library()
lib.pt_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
W, H = (int(os.environ.get('PT_W', 1920)), int(os.environ.get('PT_H', 1080)))
N = int(os.environ.get('PT_SHARDS', 1))
packed, cam_args = scenes.build(scene)
cam = scenes.make_camera(cam_args, W, H)
ds = R.DeviceScene(packed)
R.render(W, H, 8, ds, cam, shard_index=0, shard_count=N); torch.cuda.synchronize()
lib.pt_debug_stamps(None, 1)
if os.environ.get("PT_STAMPS_RUNS"):
    lib.pt_debug_runs.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.pt_debug_runs(None, 1)
WALK = bool(os.environ.get("PT_STAMPS_WALK"))
if WALK:
    lib.pt_debug_walk.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    lib.pt_debug_walk(None, 1)
no_lpt = 0 if (len(sys.argv) > 4 and sys.argv[4] == 'lpt') else abi.PT_FLAG_NO_LPT
fb, ms = R.render(W, H, spp, ds, cam, flags=flags | no_lpt, shard_index=0, shard_count=N, timed=True)
out = (C.c_ulonglong * 8)()
lib.pt_debug_stamps(out, 0)
prep, trav, shade, iters = out[0], out[1], out[2], out[3]
tot = prep + trav + shade
if WALK:
    wk = (C.c_ulonglong * 8)()
    lib.pt_debug_walk(wk, 0)
    cyc, walks, wsteps, splits, lsteps, ltests, wtrips = [wk[i] for i in range(7)]
    samples = W * H * spp / N
    print(f"  grid walks per wave-iteration {walks/iters:.3f}; cycles per walk {cyc/max(walks,1):.0f} = {cyc/max(trav,1):.3f} of the traversal, {cyc/max(tot,1):.3f} of the iteration")
    print(f"  per walk: wave-steps {wsteps/max(walks,1):.2f}, lane-steps {lsteps/max(walks,1):.1f} (lanes stepping per wave-step {lsteps/max(wsteps,1):.1f}), "
          f"test trips {wtrips/max(walks,1):.2f}, lane sphere tests {ltests/max(walks,1):.1f} (lanes testing per trip {ltests/max(wtrips,1):.1f}), split phases {splits/max(walks,1):.3f}")
    print(f"  per sample: cells visited {lsteps/samples:.2f}, grid sphere tests {ltests/samples:.2f}")
    if os.environ.get("PT_WALK_JSON"):  # the record bench.py's culled-algorithm pricing is built on (profiles/<tag>_walk_counters.json)
        import json
        final = {"final": True, "round": int(os.environ["PT_FINAL_ROUND"])} if os.environ.get("PT_FINAL_ROUND") else {}  # what bench.py selects
        json.dump({**final, "scene": scene, "workload": f"{W}x{H}x{spp}", "shards": N, "note": "in-kernel counters of the sphere-grid walk, diagnostic build "
                   "(make -C path_tracer_amd/csrc stamps EXTRA=-DPT_STAMPS_WALK), probe pass included in the wave counts, not in the per-sample figures' denominators",
                   "per_sample": {"cells_visited": lsteps / samples, "grid_sphere_tests": ltests / samples, "walks_waves": walks / samples},
                   "per_walk": {"wave_steps": wsteps / max(walks, 1), "test_trips": wtrips / max(walks, 1), "lanes_per_step": lsteps / max(wsteps, 1),
                                "lanes_per_trip": ltests / max(wtrips, 1), "cycles": cyc / max(walks, 1)},
                   "cycles_per_wave_iteration": {"prepare": prep / iters, "traversal": trav / iters, "shade": shade / iters},
                   "walk_share_of_iteration": cyc / max(tot, 1)}, open(os.environ["PT_WALK_JSON"], "w"), indent=1)
elif not os.environ.get("PT_STAMPS_POOL"):
    print(f"  gridded sphere runs scanned {out[4]:.3e}; full-list fallback in {out[5]/max(out[4],1):.4f} of them (far origin in {out[6]/max(out[4],1):.4f}), {out[7]/max(out[5],1):.1f} lanes at fault on average")
if os.environ.get("PT_STAMPS_RUNS"):  # build with EXTRA="-DPT_STAMPS_RUNS": cycles per run of the hittable list
    lib.pt_debug_runs.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    rr = (C.c_ulonglong * 16)()
    lib.pt_debug_runs(rr, 0)
    print("  cycles per wave-iteration by run of the list:", "  ".join(f"run {i}: {rr[i]/iters:.0f}" for i in range(16) if rr[i]))
if os.environ.get("PT_STAMPS_POOL"):
    print(f"  slab pool scans {out[4]:.3e}; exact trips per scan {out[5]/max(out[4],1):.3f}, lanes busy per trip {out[7]/max(out[5],1):.1f}, repeated passes per scan {out[6]/max(out[4],1):.4f}")
print(f"{scene} {spp} spp: kernel {ms:.1f} ms; wave-iterations {iters:.3e}; cycles per wave-iteration: prepare {prep/iters:.0f}  traversal {trav/iters:.0f}  shade {shade/iters:.0f}  (total {tot/iters:.0f}); shares {prep/tot:.2f} {trav/tot:.2f} {shade/tot:.2f}")

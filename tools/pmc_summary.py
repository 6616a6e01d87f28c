"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per CSV) into one JSON summary for the
render kernel: per-launch counter sums plus the derived figures DESIGN.md quotes.

  python tools/pmc_summary.py [--final ROUND] r02_smoke smoke 1920 1080 1024 profiles/r02_smoke_pmc_*.csv > profiles/r02_smoke_pmc_summary.json

--final ROUND marks the summary as THE counters of round ROUND's final build for this (scene, workload): bench.py's
`pmc_traffic` only ever reads summaries so marked (highest round wins) — never "the newest file by name".
"""
import collections
import csv
import hashlib
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KERNEL_SOURCES = ("pt_render.hip", "pt_device.hpp", "pt_math.hpp", "pt_flatten.hpp", "pt_tripool.hpp", "pt_binned.hpp")  # csrc/Makefile: KSRC


def kernels_sha16(root=ROOT):
    """sha256 over the sources the render kernels are compiled from (path_tracer_amd/csrc): what a PMC recording belongs to.
    bench.py carries the same function; a recording whose hash differs from the tree's is reported as stale (its fields nulled)."""
    hsh = hashlib.sha256()
    for name in KERNEL_SOURCES:
        hsh.update((root / "path_tracer_amd" / "csrc" / name).read_bytes())
    return hsh.hexdigest()[:16]


def git_head(root=ROOT):
    try:
        return subprocess.run(["git", "-C", str(root), "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()[:12]
    except Exception:  # noqa: BLE001  (the GPU box's snapshot has no .git: the source hash below is what identifies the build)
        return None


def main():
    final_round = None
    if sys.argv[1] == "--final":
        final_round = int(sys.argv[2])
        del sys.argv[1:3]
    tag, scene, w, h, spp = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    # A render is two launches of the render kernel (a short cost-probe pass, then the frame).  The FRAME launch is the dispatch
    # that runs longest; every counter is read from that dispatch of its own pass.  (Round 2 took each counter's maximum over
    # the dispatches instead, which for WRITE_SIZE picked the probe pass — 66 MB of per-pixel cost atomics in 1.6 ms — and made
    # the frame's write traffic look 2.7x its algorithmic 12 B/pixel; it is 1.06x.)
    meta = {}
    per = {}
    for path in sys.argv[6:]:
        rows = [r for r in csv.DictReader(open(path)) if "render_kernel" in r["Kernel_Name"]]
        if not rows:
            continue
        dur = collections.defaultdict(float)
        for r in rows:
            dur[r["Dispatch_Id"]] = max(dur[r["Dispatch_Id"]], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        frame = max(dur, key=dur.get)
        here = {}  # this pass's counters (summed over the rows of the frame dispatch: one row per XCD / instance)
        for r in rows:
            if r["Dispatch_Id"] == frame:
                here[r["Counter_Name"]] = here.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                meta = {k: r[k] for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
        for k, v in here.items():  # a counter collected in several passes (GRBM_GUI_ACTIVE rides along as each pass's clock): the first pass's value
            per.setdefault(k, v)
    samples = w * h * spp
    out = {"tag": tag, "scene": scene, "workload": f"{w}x{h}x{spp}", "kernel": meta, "per_launch": per, "derived": {}}
    if final_round is not None:
        out["final"], out["round"] = True, final_round
    import os
    # the build these counters were recorded on: the hash tools/profile_round.sh took on the GPU box (PT_KERNELS_SHA16), else this tree's
    out["kernels_sha16"] = os.environ.get("PT_KERNELS_SHA16") or kernels_sha16()
    out["recorded_at_head"] = git_head()     # (None on the GPU box; tools/collect_profiles.sh fills it in when it copies the summary)
    d = out["derived"]
    if "GRBM_GUI_ACTIVE" in per:
        cyc = per["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
        d["cycles_per_xcd"] = cyc
        if "SQ_INSTS_VALU" in per:
            d["valu_lane_instr_per_sample"] = per["SQ_INSTS_VALU"] * 64 / samples
            d["valu_issue_occupancy"] = per["SQ_INSTS_VALU"] * 2 / (1024 * cyc)  # wave64 on SIMD-32: 2 cycles per instruction
        if "SQ_INSTS_VALU" in per and "SQ_INSTS_SALU" in per:
            # a SIMD issues one instruction per two cycles whatever its kind: scalar and LDS instructions
            # take the same slots as vector ones
            d["issue_slot_occupancy"] = (per["SQ_INSTS_VALU"] + per["SQ_INSTS_SALU"] + per.get("SQ_INSTS_LDS", 0.0)) * 2 / (1024 * cyc)
        if "SQ_WAVE_CYCLES" in per:
            d["mean_waves_per_simd"] = per["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc)  # SQ_WAVE_CYCLES counts quad-cycles
    if "SQ_THREAD_CYCLES_VALU" in per and "SQ_ACTIVE_INST_VALU" in per:
        d["valu_lane_utilisation"] = per["SQ_THREAD_CYCLES_VALU"] / (64 * per["SQ_ACTIVE_INST_VALU"])
    if "SQ_INSTS_SALU" in per and "SQ_INSTS_VALU" in per:
        d["salu_per_valu"] = per["SQ_INSTS_SALU"] / per["SQ_INSTS_VALU"]
    if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
        d["l2_hit_rate"] = per["TCC_HIT_sum"] / max(1.0, per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    if "TCC_EA0_RDREQ_sum" in per:
        # what leaves L2 towards the fabric: 64-byte requests except the 32-byte ones; Infinity-Cache hits are among them (MI355X_MICROARCH.md)
        d["bytes_read_past_l2"] = (per["TCC_EA0_RDREQ_sum"] - per.get("TCC_EA0_RDREQ_32B_sum", 0.0)) * 64 + per.get("TCC_EA0_RDREQ_32B_sum", 0.0) * 32
        if "TCC_EA0_RDREQ_DRAM_sum" in per:
            d["read_requests_to_dram_share"] = per["TCC_EA0_RDREQ_DRAM_sum"] / max(1.0, per["TCC_EA0_RDREQ_sum"])
    if "TCP_PENDING_STALL_CYCLES_sum" in per and "GRBM_GUI_ACTIVE" in per:
        d["l1_pending_stall_share"] = per["TCP_PENDING_STALL_CYCLES_sum"] / (256 * per["GRBM_GUI_ACTIVE"] / 8)  # per TCP (one per CU) and cycle
    if "FETCH_SIZE" in per or "WRITE_SIZE" in per:
        # KiB units. The guide's gfx950 x2 FETCH_SIZE correction is calibrated for 16 B/lane streaming reads only;
        # this kernel's reads are scalar/LDS-staged (uncalibrated), so both readings are reported.
        f, wr = per.get("FETCH_SIZE", 0.0) * 1024, per.get("WRITE_SIZE", 0.0) * 1024
        d["hbm_bytes_per_launch"] = f + wr
        d["hbm_bytes_per_launch_fetch_x2"] = 2 * f + wr
        d["algorithmic_bytes_per_launch"] = w * h * 12
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""The measurement tables of DESIGN.md §5 / §6, generated from the committed records under profiles/ (round tag as argument):
    python tools/design_tables.py r04"""
import json
import sys
from pathlib import Path

P = Path(__file__).resolve().parent.parent / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def load(name):
    f = P / name
    return json.loads(f.read_text()) if f.exists() else None


rows = [("cfg2 Cornell-style 1920×1080×1024 spp (headline)", "cornell"), ("cfg3 SmokeSphere (496 hittables, the reference's image textures) 1920×1080×1024 spp", "smoke"),
        ("cfg1 SmokeSphere 400×225×64 spp (the reference's own CPU case)", "cfg1"), ("cfg5 100 k triangles 1920×1080×256 spp (triangle pool)", "triangles")]
print("| config (`profiles/%s_<tag>_*`) | Msamples/s | kernel ms | bound | frac | priced algorithm, ops/sample | executed / algorithmic | VALU issue | all-instruction issue slots | lane utilisation | waves/SIMD |" % tag)
print("|---|---|---|---|---|---|---|---|---|---|---|")
for title, t in rows:
    b, s = load(f"{tag}_{t}_bench_n1.json"), load(f"{tag}_{t}_pmc_summary.json")
    if not b:
        continue
    r, d = b["roofline"], (s or {}).get("derived", {})
    valu = r.get("valu", {"frac": r["frac"]})
    eoa = d.get("valu_lane_instr_per_sample", 0) / r["algorithmic_ops_per_sample"] if d.get("valu_lane_instr_per_sample") else None
    print(f"| {title} `{t}` | **{b['value']:,.0f}** | {r['kernel_ms']:,.1f} | {r['bound']} | VALU {valu['frac']:.3f}" + (f", memory {r['hbm']['frac']:.3f}" if r.get("hbm") else "")
          + f" | {r['priced_algorithm'].split(' (')[0]}, {r['algorithmic_ops_per_sample']:,.0f}" + (f" (reference: {r['algorithmic_ops_per_sample_reference']:,.0f})" if r['algorithmic_ops_per_sample_reference'] != r['algorithmic_ops_per_sample'] else "")
          + f" | {eoa:.2f}× | {d.get('valu_issue_occupancy', 0):.2f} | {d.get('issue_slot_occupancy', 0):.2f} | {d.get('valu_lane_utilisation', 0):.2f} | {d.get('mean_waves_per_simd', 0):.1f} |".replace("None×", "—"))
    if b.get("cpu_baseline") and t == "cornell":
        c = b["cpu_baseline"]
        cpu = f"{c['value']} Msamples/s on {c['cores']} host threads (portable math; {c.get('value_glibc_math')} with glibc's libm) — {c['sample']}"
print()
print("CPU:", cpu)
t = load(f"{tag}_triangles_pmc_summary.json")
if t:
    d = t["derived"]
    ms = load(f"{tag}_triangles_bench_n1.json")["roofline"]["kernel_ms"]
    print(f"cfg5 memory: bytes past L2 {d.get('bytes_read_past_l2', 0) / 1e12:.1f} TB (FETCH_SIZE + WRITE_SIZE {d['hbm_bytes_per_launch'] / 1e12:.1f} TB) in {ms / 1e3:.2f} s = {d['hbm_bytes_per_launch'] / ms / 1e9:.2f} TB/s; "
          f"L2 hit rate {d.get('l2_hit_rate', 0):.2f}; L1 pending-stall share {d.get('l1_pending_stall_share', 0):.2f}")
print()
print("| frame (`profiles/%s_shard_table_*`) | mode | N = 1 | 2 | 4 | 8 | predicted 8-GPU speed-up |" % tag)
print("|---|---|---|---|---|---|---|")
for title, name in (("Cornell 1920×1080×1024 spp (cfg2)", "cornell_1080p_1024spp"), ("SmokeSphere 3840×2160×4096 spp (cfg4 itself)", "smoke_4k_4096spp"),
                    ("SmokeSphere 3840×2160×512 spp", "smoke_4k_512spp"), ("100 k triangles 1920×1080×64 spp (cfg5's scene)", "triangles_1080p_64spp")):
    d = load(f"{tag}_shard_table_{name}.json")
    if not d:
        continue
    for mode in ("parity", "fast"):
        v = d[mode]
        print(f"| {title} | {mode}{' (not the reference image)' if mode == 'fast' else ''} | " + " | ".join(f"{v[str(n)]:,.1f}" for n in (1, 2, 4, 8)) + f" | **{v['1'] / v['8']:.1f}×** |")

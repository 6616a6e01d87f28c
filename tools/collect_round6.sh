#!/bin/bash
# After `gpurun -- 'bash tools/evidence_round6.sh'`: copy what DESIGN.md / bench.py quote from gpurun_out/ into profiles/ (tracked).
export PT_FINAL_ROUND=6
tools/collect_profiles.sh r06_cornell cornell 1920 1080 1024
tools/collect_profiles.sh r06_smoke smoke 1920 1080 1024
tools/collect_profiles.sh r06_cfg1 smoke 400 225 64
tools/collect_profiles.sh r06_triangles triangles 1920 1080 256
O=gpurun_out/r06f
[ -f $O/gpu_tests.log ] && cp $O/gpu_tests.log profiles/r06_gpu_tests.log
for f in gpu_health.txt bench_cfg2_steps20.json bench_cfg2_fast_mode.json bench_cfg2_dist_single.json shard_table_cornell_1080p_1024spp.txt shard_table_cornell_1080p_1024spp.json \
         shard_table_smoke_4k_512spp.txt shard_table_smoke_4k_512spp.json shard_table_smoke_4k_4096spp.txt shard_table_smoke_4k_4096spp.json \
         shard_table_triangles_1080p_64spp.txt shard_table_triangles_1080p_64spp.json smoke_walk_counters.json smoke_walk_stamps.txt \
         tripool_counters.json tripool_counters.txt soak_triangle_fields.log soak_all_kinds.log soak_tri_renderers_final.log soak_scheduling.log; do
  [ -f $O/$f ] && cp $O/$f profiles/r06_$f
done
ls profiles | grep -c r06

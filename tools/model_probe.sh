#!/bin/bash
# tools/model_probe.sh lib...   cfg1 + whole frame + shards of the 496-hittable scene for several builds (makespan-model tuning)
for lib in "$@"; do
  echo "== $lib"
  PT_RENDER_LIB=$PWD/$lib python tools/cfg1_probe.py 2>/dev/null | head -1
  PT_RENDER_LIB=$PWD/$lib python tools/shard_probe.py smoke 256 2>&1 | grep tile-queue | awk '{print $4, $5, $6}' | paste - - - -
done

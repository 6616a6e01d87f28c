#!/bin/bash
# Shard-0-of-N kernel times for several builds of the extension in ONE gpurun call:
#   tools/lib_shards.sh scene spp lib1.so lib2.so ...
scene=$1; spp=$2; shift 2
for lib in "$@"; do
  echo "== $lib"; PT_RENDER_LIB=$PWD/$lib python tools/shard_probe.py $scene $spp 2>&1 | grep tile-queue
done

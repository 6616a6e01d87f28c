#!/bin/bash
# tools/ab3.sh scene spp flags : current build vs libpt_w7.so vs libpt_w8.so in one process group (same GPU)
for i in 1 2; do
  echo -n "cur "; tools/quick_bench.sh $1 $2 $3
  echo -n "w7  "; PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_w7.so tools/quick_bench.sh $1 $2 $3
  echo -n "w8  "; PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_w8.so tools/quick_bench.sh $1 $2 $3
done

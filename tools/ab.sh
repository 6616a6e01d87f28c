#!/bin/bash
# A/B two builds of the extension in ONE gpurun call (different boxes differ by ~10 %):
#   tools/ab.sh scene spp flags rounds   compares path_tracer_amd/libpt_render_base.so (A) with libpt_render.so (B)
for i in $(seq 1 ${4:-2}); do
  echo -n "A(base) "; PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_render_base.so tools/quick_bench.sh $1 $2 $3
  echo -n "B(new)  "; tools/quick_bench.sh $1 $2 $3
done

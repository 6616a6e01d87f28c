#!/bin/bash
# A/B two builds at the cfg4 regime (4K frame, throughput-bound): tools/ab4k.sh spp rounds
for i in $(seq 1 ${2:-1}); do
  for lib in base new; do
    if [ $lib = base ]; then export PT_RENDER_LIB=$PWD/path_tracer_amd/libpt_render_base.so; else unset PT_RENDER_LIB; fi
    echo -n "$lib "; python bench.py --steps 1 --warmup 1 --scene smoke --width 3840 --height 2160 --spp $1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], 'Msamples/s  kernel_ms', d['roofline']['kernel_ms'])"
  done
done

#!/bin/bash
# bench.py itself with the probe's samples kept (default) and thrown away (PT_NO_PROBE_RESUME=1), alternating, one gpurun call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_resume; O=gpurun_out/r05_resume/bench_ab.log; : > $O
for rep in 1 2; do
  for cfg in cfg3 cfg2 cfg1; do
    for mode in kept again; do
      if [ $mode = again ]; then export PT_NO_PROBE_RESUME=1; else unset PT_NO_PROBE_RESUME; fi
      python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg $mode', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" >> $O
    done
  done
done
cat $O

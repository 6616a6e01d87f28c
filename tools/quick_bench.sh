#!/bin/bash
# usage: tools/quick_bench.sh scene spp flags [steps]   -> one line: flags value kernel_ms
python bench.py --steps ${4:-1} --warmup 1 --scene $1 --spp $2 --flags $3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 spp=$2 flags=$3', d['value'], 'Msamples/s  kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"

#!/bin/bash
# Sweep of the sphere grid's margin (PT_GRID_M, in median radii) and cell size (PT_GRID_CELL, in median radius + margin)
# on the SmokeSphere frame (median radius 0.2).   tools/grid_sweep.sh "0.25 0.5 0.75" "1.2 1.5 1.8 2.2"   (cell edges in units)
MS=${1:-"1.5 1.0 0.5 0.25"}; CS=${2:-"1.0 1.4 2.0"}
for m in $MS; do for c in $CS; do
  f=$(python3 -c "print($c / (0.2 * (1 + $m)))")
  echo -n "m=$m cell=$c (factor $f)  "
  PT_GRID_M=$m PT_GRID_CELL=$f python bench.py --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c75-110
done; done

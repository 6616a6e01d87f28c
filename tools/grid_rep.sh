for rep in 1 2 3; do
for mc in "1.5 2.8" "0.5 3.0" "0.5 3.3333" "0.5 4.0" "0.75 3.4286" "0.4 3.4"; do
  set -- $mc
  echo -n "m=$1 factor=$2  "
  PT_GRID_M=$1 PT_GRID_CELL=$2 python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c75-110
done; done

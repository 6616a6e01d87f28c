cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
PT_TRICULL=1 PT_TRI_M=12 PT_TRI_CELL=1.0 python bench.py --config cfg5 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | cut -c1-220
PT_TRICULL=1 PT_TRI_M=12 PT_TRI_CELL=1.0 python bench.py --config cfg5 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-220
python bench.py --config cfg5 --spp 32 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | cut -c1-220

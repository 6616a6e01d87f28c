python -m pytest tests/test_gpu_parity.py -x -q -k "config1 or cfg1 or smoke or sphere_grid or sphere_runs or bounce_bit_exact or 4k or cfg4" 2>&1 | tail -2
python -m pytest tests/test_gpu_fuzz.py -x -q -k "sphere or path_rays_on_the_baseline" 2>&1 | tail -2
tools/abn.sh "libpt_render.so" smoke 1024 1 2>/dev/null
tools/abn.sh "libpt_render.so" smoke 256 1 3840 2160 2>/dev/null
tools/abn.sh "libpt_render.so" smoke 512 8 3840 2160 2>/dev/null
tools/abn.sh "libpt_render.so" smoke 64 1 400 225 2>/dev/null
tools/abn.sh "libpt_render.so" smoke 256 8 1920 1080 2>/dev/null
tools/abn.sh "libpt_render.so" smoke 256 16 1920 1080 2>/dev/null

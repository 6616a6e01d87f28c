#!/bin/bash
for q in 0 256 512 1024; do
echo "PT_PRIO_STEP=$q"
export PT_PRIO_STEP=$q
tools/abn.sh "libpt_render.so" cornell 1024 1; tools/abn.sh "libpt_render.so" cornell 1024 4; tools/abn.sh "libpt_render.so" cornell 1024 8
tools/abn.sh "libpt_render.so" smoke 512 1; tools/abn.sh "libpt_render.so" smoke 512 8
tools/abn.sh "libpt_render.so" smoke 512 1 3840 2160; tools/abn.sh "libpt_render.so" smoke 512 8 3840 2160
tools/abn.sh "libpt_render.so" smoke 64 1 400 225
done

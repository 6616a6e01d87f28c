#!/bin/bash
for b in 0 1 2 3 4 6; do
  echo "PT_BLOCKS_PER_CU=$b"
  if [ $b = 0 ]; then tools/abn.sh "libpt_render.so" cornell 1024 8; tools/abn.sh "libpt_render.so" cornell 1024 4;
  else PT_BLOCKS_PER_CU=$b tools/abn.sh "libpt_render.so" cornell 1024 8; PT_BLOCKS_PER_CU=$b tools/abn.sh "libpt_render.so" cornell 1024 4; fi
done

#!/bin/bash
tools/abn.sh "libpt_var_nosaddr.so libpt_render.so libpt_var_nosaddr.so libpt_render.so" triangles 16 1

#!/bin/bash
# build_variant.sh NAME FILE.hip "-DFLAG ..." : ab_libs/libNAME.so = the library with FILE recompiled with extra flags (A/B of kernel builds; MCL_LIB_PATH selects it)
set -e
cd $(dirname $0)/..
mkdir -p ab_libs/obj
name=$1; src=$2; shift 2
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $@ -c mclstexp_amd/csrc/$src -o ab_libs/obj/$name.o
objs=$(ls mclstexp_amd/csrc/build/*.o | grep -v "/${src%.hip}.o")
hipcc -shared -fPIC --offload-arch=gfx950 $objs ab_libs/obj/$name.o -o ab_libs/lib$name.so
echo built ab_libs/lib$name.so

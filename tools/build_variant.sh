#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>: rebuild ONE translation unit with extra -D flags and link
# it with the other (already built) objects into tools/tmp_libs/lib_<name>.so  (use with IMH_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/tmp_libs
obj=tools/tmp_libs/${name}_$(basename $src).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c imagharmony_amd/csrc/$src -o $obj
others=$(ls imagharmony_amd/csrc/_obj/*.o | grep -v "/$(basename $src).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/tmp_libs/lib_${name}.so $obj $others
echo built tools/tmp_libs/lib_${name}.so

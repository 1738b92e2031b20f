#!/bin/bash
# A variant of the library with ONE source recompiled under extra flags (A/B on one box: BN_HIP_LIB=<it>):
#   tools/build_variant.sh <name> <source.hip> <flags...>   ->  tools/lab/bin/libbn_<name>.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../behavenet_amd/csrc"
obj=../../tools/lab/bin/${src%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -c $src -o $obj
objs=$(ls *.o | grep -v tuning | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/bin/libbn_$name.so $objs $obj
echo tools/lab/bin/libbn_$name.so

#!/bin/bash
# usage: tools/runs/build_variant.sh <name> <source.hip> [-Dflags...]: one kernel source recompiled with extra flags, linked with the
# in-tree build's other objects into cyclical-visual-captioning_amd/cvc/lib/variants/libcvc_<name>.so (git-ignored; travels with gpurun) (measurement builds, A/B runs)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
pkg=$root/cyclical-visual-captioning_amd
out=$pkg/cvc/lib/variants
mkdir -p $pkg/build/variants $out
obj=$pkg/build/variants/$(basename $src).$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-comment "$@" -c $pkg/csrc/$src -o $obj
others=$(ls $pkg/build/*.hip.o | grep -v "/$(basename $src).o" | grep -v "gemm_gsk\|gemm_packed_ks")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcvc_$name.so $obj $others -ldl
echo $out/libcvc_$name.so

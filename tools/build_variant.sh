#!/bin/bash
# Builds lib/libvohip_<tag>.so with extra compiler flags for one source file (kernel tuning experiments):
#   tools/build_variant.sh <tag> <source.hip> <flags...>      then run with VOHIP_LIB=<that library>
set -e
TAG=$1; SRC=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/visual-odometry-pipeline_amd
python3 "$PKG/build.py" > /dev/null
OBJ=$PKG/build/$(basename "$SRC" .hip)_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function \
    -I"$ROOT/include" -I"$PKG/csrc" "$@" -c "$PKG/csrc/$SRC" -o "$OBJ"
OBJS=$(ls "$PKG"/build/*.o | grep -v "_[A-Za-z0-9]*\.o$" | grep -v "/$(basename "$SRC" .hip)\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/lib/libvohip_$TAG.so" $OBJS "$OBJ"
echo "$PKG/lib/libvohip_$TAG.so"

#!/bin/bash
# Build a second library from a given version of one csrc/*.hip (a git revision of conv.hip, or a source file) for same-box
# A/B runs:   tools/build_variant.sh <name> <git-rev | path/to/file.hip> [extra hipcc flags]
#   -> neural-audio-fp_amd/_abl/libnafp_<name>.so          (select it with NAFP_LIB=...)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
P=neural-audio-fp_amd
mkdir -p $P/_abl /tmp/nafp_variant
if [ -f "$SRC" ]; then BASE=$(basename "$SRC" .hip); cp "$SRC" /tmp/nafp_variant/$BASE.hip; else BASE=conv; git show "$SRC:$P/csrc/conv.hip" > /tmp/nafp_variant/conv.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -fno-gpu-rdc -I$P/csrc "$@" -c /tmp/nafp_variant/$BASE.hip -o /tmp/nafp_variant/$BASE.o
OBJS=$(ls $P/build/*.o | grep -v /$BASE.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc $OBJS /tmp/nafp_variant/$BASE.o -o $P/_abl/libnafp_$NAME.so
echo $P/_abl/libnafp_$NAME.so

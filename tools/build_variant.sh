#!/bin/bash
# Diagnostic builds: recompiles ONE source of the library with extra -D flags and links it with the shipped objects
# of the others -> sparch_amd/libsparch_hip_<NAME>.so (git-ignored; select it with SPARCH_HIP_LIB).  Never shipped.
# usage: tools/build_variant.sh NAME SOURCE.hip [-DFOO=1 ...]
set -eu
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/../sparch_amd/csrc"
make -s -j8 all
mkdir -p ../../.ab/$NAME
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
EXTRA=""; [ "$SRC" = cell.hip ] && EXTRA="-fno-slp-vectorize"   # as the Makefile's FLAGS_cell
$HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -I../../include -Wall -Wno-unused-function $EXTRA "$@" -c $SRC -o ../../.ab/$NAME/${SRC%.hip}.o
OBJS=$(ls *.o | grep -v "^${SRC%.hip}.o$")
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS ../../.ab/$NAME/${SRC%.hip}.o -o ../libsparch_hip_$NAME.so
echo built sparch_amd/libsparch_hip_$NAME.so

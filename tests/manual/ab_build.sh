#!/bin/bash
# tests/manual/ab_build.sh <name> <translation unit> [extra hipcc flags...]: abtest/<name>.so = the library with ONE
# translation unit rebuilt under extra flags (-D switches, -I for an alternative header) and the other objects of
# longtr_amd/csrc/build as they are.  Variants run through LTR_GPU_LIB=abtest/<name>.so (same box, interleaved).
set -e
cd "$(dirname "$0")/../.."
NAME=$1; TU=$2; shift 2
B=longtr_amd/csrc/build
mkdir -p abtest/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-honor-nans -std=c++17 -fPIC -pthread -Wall "$@" -c longtr_amd/csrc/$TU -o abtest/obj/$NAME.$TU.o
OBJS=$(ls $B/*.o | grep -v "/$TU.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread $OBJS abtest/obj/$NAME.$TU.o -lz -o abtest/$NAME.so
echo built abtest/$NAME.so

#!/bin/bash
# build_from.sh <name> <git-rev> [file ...]: scratch/lib_<name>/libgs360hip.so from the working tree's csrc with the named files taken from
# <git-rev> instead (A/B of one kernel file against an earlier commit on the SAME box: GS360_LIB=scratch/lib_<name>/libgs360hip.so)
set -e
cd "$(dirname "$0")/../.."
name=$1; rev=$2; shift 2
d=scratch/src_$name
rm -rf $d && mkdir -p $d/csrc scratch/lib_$name
cp 360cam-pgm-3dgs-tools_amd/csrc/* $d/csrc/
for f in "$@"; do git show $rev:360cam-pgm-3dgs-tools_amd/csrc/$f > $d/csrc/$f; done
sed -i "s#../../include/gs360.h#$PWD/include/gs360.h#" $d/csrc/gs360_kernels.h
cd $d/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-flush-denormals-to-zero -Wno-unused-result -shared -o ../../lib_$name/libgs360hip.so \
    gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2>&1 | grep -i " error" || true
ls -la ../../lib_$name/libgs360hip.so

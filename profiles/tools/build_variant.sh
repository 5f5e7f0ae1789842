#!/bin/bash
# build_variant.sh <name> [-DFLAG=V ...]: builds scratch/lib_<name>/libgs360hip.so from the working tree with extra defines
# (A/B probes: select with GS360_LIB=scratch/lib_<name>/libgs360hip.so).  scratch/ is git-ignored but travels with gpurun.
set -e
cd "$(dirname "$0")/../../360cam-pgm-3dgs-tools_amd/csrc"
name=$1; shift
out=../../scratch/lib_$name
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result "$@" \
    -Rpass-analysis=kernel-resource-usage -shared -o $out/libgs360hip.so gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2> $out/kernel_resources.txt
grep -A8 "eq_views_kernelILi3ELb[01]ELb[01]ELi1" $out/kernel_resources.txt | grep -E "Function|VGPRs:|Scratch|VGPRs Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - -

#!/bin/bash
# build_sm_nl.sh <name> <NL>: source-major kernel with NL loader wavefronts sharing every image's rows (row % NL == loader), same two buffers
set -e
cd /root/repo
name=$1; NL=$2
d=scratch/src_$name
rm -rf $d && mkdir -p $d/csrc scratch/lib_$name
cp 360cam-pgm-3dgs-tools_amd/csrc/* $d/csrc/
python3 - "$d/csrc/gs360_srcmajor.hip" "$NL" <<'PY'
import sys
p, NL = sys.argv[1], sys.argv[2]
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
rep("__global__ __launch_bounds__(64 * (CW + 1)) void eq_srcmajor_kernel", f"__global__ __launch_bounds__(64 * (CW + {NL})) void eq_srcmajor_kernel")
rep("                for (int row = 0; row < T.nrows; ++row, y += ystep) {", f"                y += wave * ystep;\n                for (int row = wave; row < T.nrows; row += {NL}, y += {NL} * ystep) {{")
rep("        if constexpr (MASKED) {\n            // the keep bits of the box", "        if (MASKED && wave == 0) {\n            // the keep bits of the box")
rep("""    if (wave == 0) {
        const uint8_t* ge = reinterpret_cast<const uint8_t*>(P.entries + T.eoff);
        const int eb = 20 * nq;                          // a multiple of 64 bytes (nq % 16 == 0)
        for (int o = 0; o < eb; o += 1024)
            if (o + lane * 16 < eb)
                __builtin_amdgcn_global_load_lds((global_void_t*)(ge + o + lane * 16), (lds_void_t*)(s_ent + o), 16, 0, 0);
        dma(g0, s_tile);""", f"""    if (wave < {NL}) {{
        const uint8_t* ge = reinterpret_cast<const uint8_t*>(P.entries + T.eoff);
        const int eb = 20 * nq;                          // a multiple of 64 bytes (nq % 16 == 0)
        for (int o = 1024 * wave; o < eb; o += 1024 * {NL})
            if (o + lane * 16 < eb)
                __builtin_amdgcn_global_load_lds((global_void_t*)(ge + o + lane * 16), (lds_void_t*)(s_ent + o), 16, 0, 0);
        dma(g0, s_tile);""")
rep("        if (wave == 0) {\n            if (g + 1 < G) dma(", f"        if (wave < {NL}) {{\n            if (g + 1 < G) dma(")
rep("((wave - 1) * 64 + lane) * 4;", f"((wave - {NL}) * 64 + lane) * 4;")
rep("(((wave - 1) * 64 + lane) >> 2) * 4;", f"(((wave - {NL}) * 64 + lane) >> 2) * 4;")
rep("            int i0 = (wave - 1) * 64;", f"            int i0 = (wave - {NL}) * 64;")
s = s.replace("dim3(64 * (kSmConsumers + 1))", f"dim3(64 * (kSmConsumers + {NL}))")
open(p, "w").write(s)
PY
sed -i "s#../../include/gs360.h#$PWD/include/gs360.h#" $d/csrc/gs360_kernels.h
cd $d/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
    -fno-gpu-flush-denormals-to-zero -Wno-unused-result -shared -o ../../lib_$name/libgs360hip.so \
    gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2>&1 | grep -i " error" || true
ls -la ../../lib_$name/libgs360hip.so

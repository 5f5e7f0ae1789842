#!/bin/bash
# timing probe v4: block 0 and 9, loader wave 0 and consumer wave NL+2: s_memtime at phase boundaries -> second job's output buffer
set -e
cd /root/repo
name=p5time
rm -rf scratch/r06/csrc_$name && mkdir -p scratch/r06/csrc_$name/csrc scratch/lib_$name
cp 360cam-pgm-3dgs-tools_amd/csrc/* scratch/r06/csrc_$name/csrc/
python3 - <<'PY'
p = "/root/repo/scratch/r06/csrc_p5time/csrc/gs360_tablestage.hip"
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
rep("    int g = 0;\n    for (int t = t0; t < t_end; t += nj, ++g) {",
    "    int g = 0;\n    unsigned long long tm[5] = {0, 0, 0, 0, 0};\n    unsigned long long* const dbg = reinterpret_cast<unsigned long long*>(P.job[1].dst);\n    const bool rec = (b == 0 || b == 9) && (wave == 0 || wave == NL + 2);\n    const int recbase = ((b == 0 ? 0 : 2) + (wave == 0 ? 0 : 1)) * 200;\n    for (int t = t0; t < t_end; t += nj, ++g) {\n        tm[0] = __builtin_amdgcn_s_memtime();")
rep("            __builtin_amdgcn_s_waitcnt(0x0F70);          // this loader's share of tile g + 1 has landed\n",
    "            tm[1] = __builtin_amdgcn_s_memtime();\n            __builtin_amdgcn_s_waitcnt(0x0F70);\n            tm[2] = __builtin_amdgcn_s_memtime();\n")
rep("            for (int r0 = wave - NL; r0 < R; r0 += 4 * CW) {", "            tm[1] = __builtin_amdgcn_s_memtime();\n            for (int r0 = wave - NL; r0 < R; r0 += 4 * CW) {")
rep("        __builtin_amdgcn_s_barrier();                    // tile g + 1 has landed AND every consumer is done with tile g's buffer",
    "        tm[3] = __builtin_amdgcn_s_memtime();\n        __builtin_amdgcn_s_barrier();\n        tm[4] = __builtin_amdgcn_s_memtime();\n        if (rec && lane == 0 && g < 40) for (int k = 0; k < 5; ++k) dbg[8 + recbase + g * 5 + k] = tm[k];")
open(p, "w").write(s)
PY
cd scratch/r06/csrc_$name/csrc
sed -i 's#../../include/gs360.h#/root/repo/include/gs360.h#' gs360_kernels.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result \
    -shared -o /root/repo/scratch/lib_$name/libgs360hip.so gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2>&1 | grep -i error || true
ls -la /root/repo/scratch/lib_$name/ | tail -1

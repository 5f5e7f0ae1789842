#!/bin/bash
# dmaflood probe: every wave of the workgroup copies tiles back to back (no render, no barriers): the copy path's own ceiling
set -e
cd /root/repo
name=$1; waves=$2
rm -rf scratch/r06/csrc_$name && mkdir -p scratch/r06/csrc_$name/csrc scratch/lib_$name
cp 360cam-pgm-3dgs-tools_amd/csrc/* scratch/r06/csrc_$name/csrc/
python3 - "$name" "$waves" <<'PY'
import sys
name, waves = sys.argv[1], int(sys.argv[2])
p = f"/root/repo/scratch/r06/csrc_{name}/csrc/gs360_tablestage.hip"
s = open(p).read()
a = s.index("    TsTile Tn;                                           // (loader) head of the tile whose copy is issued next")
b = s.index("}\n\n}  // namespace\n\n// ---- host side")
s = s[:a] + f"""    if (wave < {waves}) {{
        for (int tt = t + wave * nj; tt < t_end; tt += {waves} * nj) dma(tt, head_of(tt), s_lds + (wave & 1) * P.buf_bytes);
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }}
""" + s[b:]
open(p, "w").write(s)
PY
cd scratch/r06/csrc_$name/csrc
sed -i 's#../../include/gs360.h#/root/repo/include/gs360.h#' gs360_kernels.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result \
    -shared -o /root/repo/scratch/lib_$name/libgs360hip.so gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2>&1 | grep -i error || true
ls -la /root/repo/scratch/lib_$name/ | tail -1

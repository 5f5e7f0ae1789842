#!/bin/bash
# probes of the shared-copy kernel: norender | onlyfirstdma (render from tile t0's data every iteration) | nostore | nodrain (no vmcnt(0) at the end of an iteration: INVALID, timing only)
set -e
cd /root/repo
name=$1
rm -rf scratch/r06/csrc_$name && mkdir -p scratch/r06/csrc_$name/csrc scratch/lib_$name
cp 360cam-pgm-3dgs-tools_amd/csrc/* scratch/r06/csrc_$name/csrc/
python3 - "$name" <<'PY'
import sys
name = sys.argv[1]
p = f"/root/repo/scratch/r06/csrc_{name}/csrc/gs360_tablestage.hip"
s = open(p).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b)
if "norender" in name:
    rep("            for (int r = wave; r < R; r += 2 * NW) {", "            if (P.R < 0) for (int r = wave; r < R; r += 2 * NW) {")
if "firstdma" in name:
    rep("        if (t + nj < t_end) {\n            dma(t + nj, (g + 1) & 3, s_buf + ((g + 1) & 1) * P.buf_bytes);", "        if (P.R < 0) {\n            dma(t + nj, (g + 1) & 3, s_buf + ((g + 1) & 1) * P.buf_bytes);")
    rep("        uint8_t* const cur = s_buf + (g & 1) * P.buf_bytes;", "        uint8_t* const cur = s_buf;")
    rep("            const int ty = __builtin_amdgcn_readfirstlane((int)hd.z), tx = __builtin_amdgcn_readfirstlane((int)hd.w);",
        "            const int txs = (J.w + 3 + 63) / 64; const int ty = lt / txs + 0 * (int)hd.z, tx = lt - ty * txs;")
if "nostore" in name:
    rep("                if (sg.live) *reinterpret_cast<uint32_t*>(__builtin_assume_aligned(dstp + off, 4)) = dwq;",
        "                if (sg.live && dwq == 0x12345678u) *reinterpret_cast<uint32_t*>(__builtin_assume_aligned(dstp + off, 4)) = dwq;")
if "nodrain" in name:
    rep("        __builtin_amdgcn_s_waitcnt(0x0F70);              // this wavefront's share of tile g + 1 has landed (and its stores have left)\n", "")
open(p, "w").write(s)
PY
cd scratch/r06/csrc_$name/csrc
sed -i 's#../../include/gs360.h#/root/repo/include/gs360.h#' gs360_kernels.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result \
    -shared -o /root/repo/scratch/lib_$name/libgs360hip.so gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2>&1 | grep -i error || true
ls -la /root/repo/scratch/lib_$name/ | tail -1

#!/bin/bash
# build_probe.sh <name> <python-expr patch>: copies csrc, applies a text patch to gs360_tablestage.hip, builds scratch/lib_<name>
set -e
cd /root/repo
name=$1
rm -rf scratch/r06/csrc_$name && mkdir -p scratch/r06/csrc_$name/csrc scratch/lib_$name
cp 360cam-pgm-3dgs-tools_amd/csrc/* scratch/r06/csrc_$name/csrc/
mkdir -p scratch/r06/include && cp include/gs360.h scratch/r06/include/
python3 - "$name" <<'PY'
import sys
name = sys.argv[1]
p = f"/root/repo/scratch/r06/csrc_{name}/csrc/gs360_tablestage.hip"
s = open(p).read()
s = s.replace('#include "gs360_cvremap.h"', '#include "gs360_cvremap.h"\n#define PROBE_' + name.upper() + ' 1')
# no render: consumers skip the row loop
s = s.replace("            for (int r = wave - 1; r < R; r += CW) {\n                const int y = ty * R + r;",
              "#ifdef PROBE_NORENDER\n            if (P.R < 0)\n#endif\n            for (int r = wave - 1; r < R; r += CW) {\n                const int y = ty * R + r;")
# no dma: loader skips copies
s = s.replace("        for (int o = 0; o < words_bytes; o += 1024)\n            __builtin_amdgcn_global_load_lds(",
              "#ifdef PROBE_NODMA\n        if (P.R < 0)\n#endif\n        for (int o = 0; o < words_bytes; o += 1024)\n            __builtin_amdgcn_global_load_lds(")
s = s.replace("        for (int c0 = 0; c0 < T.chunks; c0 += 64) {", "#if defined(PROBE_NODMA) || defined(PROBE_NOBOX)\n        if (P.R < 0)\n#endif\n        for (int c0 = 0; c0 < T.chunks; c0 += 64) {")
s = s.replace("                *reinterpret_cast<uint32_t*>(__builtin_assume_aligned(dstp + off, 4)) = dwq;",
              "#ifdef PROBE_NOSTORE\n                if (dwq == 0x12345678u)\n#endif\n                *reinterpret_cast<uint32_t*>(__builtin_assume_aligned(dstp + off, 4)) = dwq;")
open(p, "w").write(s)
PY
cd scratch/r06/csrc_$name/csrc
sed -i 's#../../include/gs360.h#/root/repo/include/gs360.h#' gs360_kernels.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -Wno-unused-result \
    -shared -o /root/repo/scratch/lib_$name/libgs360hip.so gs360_kernels.hip gs360_table.hip gs360_tablestage.hip gs360_srcmajor.hip gs360_u16.hip gs360_color.hip gs360_capi.hip 2>/dev/null
ls -la /root/repo/scratch/lib_$name/

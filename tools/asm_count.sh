#!/bin/bash
# Static instruction mix of k_lq in the gfx950 assembly (a proxy while iterating: the kernel is mostly straight-line code).
#   tools/asm_count.sh            whole kernel
#   tools/asm_count.sh --phases   per phase (between the ordering points HB_ABLATE_STOP leaves; -DHB_PHASE_MARK names them by source line)
cd "$(dirname "$0")/../hunter_bipedal_control_amd/csrc"
if [ "$1" = "--phases" ]; then shift; set -- -DHB_PHASE_MARK "$@"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/hb_k.s hb_kernels.hip "$@" 2>/dev/null
L=$(grep -n "^_ZN12_GLOBAL__N_14k_lqE" /tmp/hb_k.s | cut -d: -f1)
sed -n "${L},\$p" /tmp/hb_k.s | awk '
function flush(name) { printf "%-28s valu %5d salu %5d lds %4d vmem %4d mfma %3d\n", name, v, s, d, g, m; tv+=v; ts+=s; td+=d; tg+=g; tm+=m; v=s=d=g=m=0 }
/s_endpgm/{flush("(end)"); v=tv; s=ts; d=td; g=tg; m=tm; tv=ts=td=tg=tm=0; flush("total"); exit}
/HB_PHASE line/{flush("-> " $0); next}
/^\tv_mfma/{m++; next} /^\tv_/{v++} /^\ts_/{s++} /^\tds_/{d++} /^\tglobal_|^\tscratch_|^\tbuffer_/{g++}'

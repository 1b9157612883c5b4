#!/bin/bash
# static instruction mix of k_lq in the gfx950 assembly (a proxy while iterating: the kernel is mostly straight-line code)
cd "$(dirname "$0")/../hunter_bipedal_control_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/hb_k.s hb_kernels.hip "$@" 2>/dev/null
L=$(grep -n "^_ZN12_GLOBAL__N_14k_lqE" /tmp/hb_k.s | cut -d: -f1)
sed -n "${L},\$p" /tmp/hb_k.s | awk '/s_endpgm/{exit} /^\tv_mfma/{m++; next} /^\tv_/{v++} /^\ts_/{s++} /^\tds_/{d++} /^\tglobal_|^\tscratch_|^\tbuffer_/{g++} END{print "valu " v " salu " s " lds " d " vmem " g " mfma " m}'

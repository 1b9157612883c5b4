#!/bin/bash
# Builds libhunter_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
# The build FAILS if one of the hot kernels of the update (either WBC flavour included) spills to scratch memory: every scratch reload is followed by
# s_waitcnt vmcnt(0), which drains the software-pipelined record prefetch of the sweeps.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libhunter_hip.so
# --ablate: the profiling variant with the phase-by-phase exits compiled in (tools/perf_quick.py --lib variants/libhunter_hip_ablate.so --ablate-lq)
# (the profiling variant may spill: its phase exits change the register allocation; scratch there is reported, not fatal)
ABLATE=0
if [ "$1" = "--ablate" ]; then shift; mkdir -p ../../variants; OUT=../../variants/libhunter_hip_ablate.so; ABLATE=1; set -- -DHB_ABLATE "$@"; fi
LOG=$(mktemp)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -fPIC -shared -Rpass-analysis=kernel-resource-usage -o $OUT hb_kernels.hip "$@" 2> "$LOG" || { cat "$LOG" >&2; rm -f "$LOG"; exit 1; }
grep -E "error|warning:" "$LOG" | grep -v "Wcomment" >&2 || true
python3 - "$LOG" "$ABLATE" <<'PY'
import re, sys
hot = ["k_lqE", "k_lq_tripE", "8k_refgenE", "k_refgen_ikE", "k_refgen_nodesE", "k_estimatorE", "k_warm_shiftE", "k_ric_bwdE", "k_ric_bwd4E", "k_ric_fwdE", "k_ric_fwd_wE", "5k_wbcE", "6k_hwbcE", "k_ls_evalE", "k_ls_tail_evalE", "k_ls_tail_decideE", "k_policy_evalE"]
txt = open(sys.argv[1]).read()
rows, bad = [], []
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?LDS Size \[bytes/block\]: (\d+)", txt, re.S):
    name, vg, ag, scr, occ, lds = m.group(1), *map(int, m.groups()[1:])
    if any(h in name for h in hot):
        rows.append(f"  {name[:48]:48s} vgpr {vg:3d} agpr {ag:3d} scratch {scr:3d} B  waves/SIMD {occ}  lds {lds} B")
        if scr:
            bad.append(name)
print("hot-kernel resources (gfx950):")
print("\n".join(rows))
if bad and sys.argv[2] == "1":
    print("build.sh --ablate: scratch memory in " + ", ".join(bad) + " (profiling variant: tolerated)")
elif bad:
    sys.exit("build.sh: scratch memory in hot kernel(s): " + ", ".join(bad))
PY
rm -f "$LOG"

#!/bin/bash
# Order of global loads / stores / vmcnt waits / matrix instructions in the gfx950 ISA of one kernel — the check behind "no load queues behind
# the record's stores"  and "Q~ is requested before it is needed" (3.2):
#     tools/isa_vmseq.sh k_lq_trip          [LINE]LDxN = N consecutive global loads, STxN stores, Wn = s_waitcnt vmcnt(n), MxN = v_mfma
# A load that follows stores and is waited for with a small n (the compiler's count of the stores in between is the minimum over all paths
# around lane-masked blocks) drains them: loads and stores retire in order on one counter.
set -e
k=${1:?kernel name (substring of the mangled symbol)}
d=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm -fPIC -S --cuda-device-only -o $d/hb.s "$(dirname "$0")/../hunter_bipedal_control_amd/csrc/hb_kernels.hip" 2>/dev/null
python3 - "$d/hb.s" "$k" <<'P'
import re, sys
src, key = open(sys.argv[1]).read().split("\n"), sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w*" + re.escape(key) + r"E\w*:", l))
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
out = []
for i, l in enumerate(src[start:end]):
    t = l.strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith(("global_load", "scratch_load")): out.append((i, "LD"))
    elif op.startswith(("global_store", "scratch_store")): out.append((i, "ST"))
    elif op == "s_waitcnt" and "vmcnt" in t: out.append((i, "W" + re.search(r"vmcnt\((\d+)\)", t).group(1)))
    elif op.startswith("v_mfma"): out.append((i, "M"))
res, prev, cnt, first = [], None, 0, 0
for i, e in out:
    if e == prev: cnt += 1
    else:
        if prev: res.append((first, prev, cnt))
        prev, cnt, first = e, 1, i
res.append((first, prev, cnt))
print(f"{src[start].split(':')[0]}: {end - start} lines")
print(" ".join(f"[{f}]{e}{'x%d' % c if c > 1 else ''}" for f, e, c in res))
P
rm -rf $d

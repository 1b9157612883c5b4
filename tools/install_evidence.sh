#!/bin/bash
# Copies what one `tools/evidence.sh <tag>` call left under gpurun_out/<tag>/final/ into profiles/ under the round's name:
#   bash tools/install_evidence.sh r04e r04      ->  profiles/r04_* (+ profiles/pmc_latest.json)
set -e
tag=$1; name=${2:-$1}
src=gpurun_out/$tag/final
[ -d "$src" ] || { echo "no $src" >&2; exit 1; }
for f in $src/${tag}_*; do
  b=$(basename "$f")
  cp "$f" profiles/${b/${tag}_/${name}_}
done
sed -i "1s#of .*/prof_\([a-z0-9_]*\)/.*#of gpurun_out/$name/prof_\1/run_results.db#" profiles/${name}_kernel_stats*.txt
sed -i "s/\"tag\": \"[^\"]*\"/\"tag\": \"$name\"/" profiles/${name}_pmc.json
sed -i "1s/(\([^)]*\))/($name)/" profiles/${name}_pmc.txt
# stamp the commit: the counters belong to the kernel sources they were collected on (source_sha256, written on the GPU box); refuse to
# install them as "latest" over a tree whose sources differ, so that a stale file can never feed a later bench line
python3 - "$name" <<'PY'
import json, subprocess, sys, importlib.util
name = sys.argv[1]
spec = importlib.util.spec_from_file_location("b", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
p = f"profiles/{name}_pmc.json"
j = json.load(open(p))
if j.get("source_sha256") != b.source_fingerprint():
    sys.exit(f"install_evidence: {p} was collected on kernel sources {str(j.get('source_sha256'))[:12]}, the working tree has {b.source_fingerprint()[:12]}")
dirty = subprocess.run(["git", "status", "--porcelain", "hunter_bipedal_control_amd/csrc"], capture_output=True, text=True).stdout.strip()
head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
j["git_head"] = head + (" + uncommitted kernel-source changes" if dirty else "")
json.dump(j, open(p, "w"), indent=1)
PY
cp profiles/${name}_pmc.json profiles/pmc_latest.json
ls profiles/${name}_* | wc -l

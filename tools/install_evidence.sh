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
cp profiles/${name}_pmc.json profiles/pmc_latest.json
ls profiles/${name}_* | wc -l

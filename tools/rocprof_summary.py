#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) or pmc CSVs into a small text/JSON file for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_kernel_stats.txt
"""
import json
import sqlite3
import sys
from collections import defaultdict


def kernel_stats(db):
    c = sqlite3.connect(db)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tables if "kernel_dispatch" in t][0]
    ks = [t for t in tables if "kernel_symbol" in t][0]
    names = {r[0]: (r[1], r[2], r[3], r[4], r[5]) for r in c.execute(
        f"select id, display_name, arch_vgpr_count, accum_vgpr_count, group_segment_size, private_segment_size from {ks}")}
    agg = defaultdict(list)
    for kid, st, en in c.execute(f"select kernel_id, start, end from {kd}"):
        agg[kid].append(en - st)
    total = sum(sum(v) for v in agg.values())
    rows = []
    for kid, d in agg.items():
        n, vg, ag, lds, scr = names[kid]
        rows.append(dict(kernel=n.replace("(anonymous namespace)::", "").split("(")[0], calls=len(d), total_ms=sum(d) / 1e6, avg_us=sum(d) / len(d) / 1e3,
                         min_us=min(d) / 1e3, max_us=max(d) / 1e3, pct=100.0 * sum(d) / total, vgpr=vg, agpr=ag, lds=lds, scratch=scr))
    rows.sort(key=lambda r: -r["total_ms"])
    return rows


def main():
    rows = kernel_stats(sys.argv[1])
    out = sys.argv[2]
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}",
             f"{'kernel':58s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>6s} {'scratch':>7s}"]
    for r in rows:
        lines.append(f"{r['kernel'][:58]:58s} {r['calls']:6d} {r['total_ms']:10.3f} {r['avg_us']:10.1f} {r['min_us']:10.1f} {r['max_us']:10.1f} "
                     f"{r['pct']:6.2f} {r['vgpr']:5d} {r['agpr']:5d} {r['lds']:6d} {r['scratch']:7d}")
    open(out, "w").write("\n".join(lines) + "\n")
    json.dump(rows, open(out.rsplit(".", 1)[0] + ".json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()

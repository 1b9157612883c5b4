#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters from rocpd databases -> profiles/<tag>_pmc.{txt,json}.

    python tools/pmc_summary.py r01 gpurun_out/pmc_r01_fetch/p_results.db gpurun_out/pmc_r01_write/p_results.db gpurun_out/pmc_r01_sq/p_results.db
"""
import json
import sqlite3
import sys
from collections import defaultdict


def per_kernel(db):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    f = lambda key: [x for x in t if key in x][0]
    pe, ip, kd, ks = f("pmc_event"), f("info_pmc"), f("kernel_dispatch"), f("kernel_symbol")
    pmc_name = {r[0]: r[1] for r in c.execute(f"select id, name from {ip}")}
    kname = {r[0]: r[1] for r in c.execute(f"select id, display_name from {ks}")}
    disp = {r[0]: (r[1], r[2] - r[3]) for r in c.execute(f"select event_id, kernel_id, end, start from {kd}")}
    acc = defaultdict(lambda: defaultdict(list))
    for ev, pid, val in c.execute(f"select event_id, pmc_id, value from {pe}"):
        if ev in disp:
            k = kname[disp[ev][0]].replace("(anonymous namespace)::", "").split("(")[0]
            acc[k][pmc_name[pid]].append(val)
    dur = defaultdict(list)
    for ev, (kid, d) in disp.items():
        dur[kname[kid].replace("(anonymous namespace)::", "").split("(")[0]].append(d)
    out = {}
    for k, counters in acc.items():
        out[k] = {name: sum(v) / len(v) for name, v in counters.items()}
        out[k]["launches"] = len(dur[k])
        out[k]["avg_us_under_profiling"] = sum(dur[k]) / len(dur[k]) / 1e3
    return out


def main():
    tag, dbs = sys.argv[1], sys.argv[2:]
    merged = defaultdict(dict)
    for db in dbs:
        for k, v in per_kernel(db).items():
            for name, val in v.items():
                merged[k][name] = val
    # HBM bytes per launch (MI355X_MICROARCH.md §HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts a wide
    # (16 B/lane) coalesced read at 1/2 -> the corrected figure doubles it; both are reported.
    for k, v in merged.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_bytes_per_launch_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
            v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        # Matrix-pipe busy fraction, normalised.  SQ_VALU_MFMA_BUSY_CYCLES is reported as (pipe-busy cycles summed over all SIMDs) / 32
        # on this stack: k_ric_bwd issues 63 v_mfma_f64_16x16x4_f64 per stage, 64 pipe cycles each, and the counter reads
        # 409 600 stages x 126 = 63 x 64 / 32 per stage (profiles/r01m_pmc.txt).  Hence
        #     mfma_pipe_busy_frac = 32 * SQ_VALU_MFMA_BUSY_CYCLES / (N_SIMD * GRBM_GUI_ACTIVE),   N_SIMD = 256 CUs x 4
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
            v["mfma_pipe_busy_frac"] = 32.0 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * v["GRBM_GUI_ACTIVE"])
        if "hbm_bytes_per_launch" in v and v.get("avg_us_under_profiling", 0) > 0:
            v["hbm_GBs_under_profiling"] = v["hbm_bytes_per_launch"] / (v["avg_us_under_profiling"] * 1e-6) / 1e9
        # wave-cycle breakdown (MI355X_MICROARCH.md §rocprofv3 PMC slots): parked at s_waitcnt / barrier, issue-stalled, issuing
        if v.get("SQ_WAVE_CYCLES", 0) > 0:
            for src, dst in (("SQ_WAIT_ANY", "wave_frac_parked_waitcnt"), ("SQ_WAIT_INST_ANY", "wave_frac_issue_stalled"),
                             ("SQ_ACTIVE_INST_ANY", "wave_frac_issuing")):
                if src in v:
                    v[dst] = v[src] / v["SQ_WAVE_CYCLES"]
        # lanes doing work in the average VALU instruction; LDS cycles lost to bank conflicts
        if v.get("SQ_ACTIVE_INST_VALU", 0) > 0 and "SQ_THREAD_CYCLES_VALU" in v:
            v["valu_lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
        if v.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in v:
            v["lds_bank_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
        # Average resident wavefronts per CU.  SQ_WAVE_CYCLES is reported in units of 4 cycles and sampled 1 / 32 on this stack
        # (checked on k_lq at 8 workgroups per CU, profiles/r02a_pmc.txt: 272.8 M x 128 / 17.3 M cycles = 2 018 = 7.9 x 256)
        if v.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_WAVE_CYCLES" in v:
            v["avg_resident_waves_per_cu"] = 128.0 * v["SQ_WAVE_CYCLES"] / v["GRBM_GUI_ACTIVE"] / 256.0
        if v.get("SQ_WAVES", 0) > 0:
            for src in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_MFMA"):
                if src in v:
                    v[src.lower() + "_per_wave"] = v[src] / v["SQ_WAVES"]
    # what the counters were collected ON: the sha256 of the kernel sources of the tree that ran (bench.source_fingerprint; there is no
    # .git on the GPU box — tools/install_evidence.sh adds the commit when it copies the file into profiles/).  bench.py drops
    # roofline.traffic when the running tree's sources differ from this stamp.
    import importlib.util
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("bench_for_fingerprint", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    json.dump(dict(merged, tag=tag, source_sha256=bench.source_fingerprint(), git_head=None), open(f"profiles/{tag}_pmc.json", "w"), indent=1)
    lines = [f"# rocprofv3 --pmc per-kernel means ({tag}); bytes corrected per MI355X_MICROARCH.md §HBM (FETCH x2 for wide reads)"]
    for k, v in sorted(merged.items(), key=lambda kv: -kv[1].get("avg_us_under_profiling", 0)):
        lines.append(k)
        for name, val in sorted(v.items()):
            lines.append(f"    {name:34s} {val:18.3f}")
    open(f"profiles/{tag}_pmc.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:80]))


if __name__ == "__main__":
    main()

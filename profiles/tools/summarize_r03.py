#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/tools/collect_r03.sh (gpurun_out/) into the files kept under profiles/:

  r03_<wl>_kernel_stats.csv     the --stats summaries (wl = m1, mhd8, m2, m3, dmr2d, mhd2d, axi2d, mhdaxi2d)
  r03_<wl>_under_rocprof.json   the JSON line bench.py printed in that profiled run
  r03_pmc_traffic.json          bytes per stage-kernel launch that left L2, per workload and per kernel instance:
                                FETCH_SIZE / WRITE_SIZE from separate --pmc passes, KB -> bytes, read counter x the
                                gfx950 correction measured with tools/calib_traffic.hip at the same 8 B/lane access
                                width -- on aligned streams AND on the stage kernel's own x tiling (62-cell tiles of 64
                                lanes starting off the 128-byte line grid)
  r03_pmc_l2_fabric.json        M1: the raw fabric-side read counters (requests, 32-byte requests, DRAM-routed
                                requests), L2 hit / miss, write requests: request sizes and the DRAM share of the reads
  r03_pmc_sq.json               SQ counters per stage-kernel instance, per workload
  r03_clock.json                effective clock of the stage kernels from GRBM_GUI_ACTIVE (MI355X_MICROARCH.md, DVFS)
Every JSON carries the hash of the kernel sources it was measured on (bench.py quotes them only while it matches)."""
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
WLS = ["m1", "mhd8", "m2", "m3", "dmr2d", "mhd2d", "axi2d", "mhdaxi2d"]
KB = 1024.0   # FETCH_SIZE / WRITE_SIZE are reported in KB


def kernel_source_hash():
    """the same function as bench.py's (code only: // comments and blank lines dropped)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pion_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")) or f == "Makefile":
            h.update(f.encode())
            with open(os.path.join(d, f), encoding="utf-8", errors="replace") as fh:
                for line in fh:
                    line = re.sub(r"(//|#(?!\s*(include|define|if|else|endif|ifdef|ifndef|undef|pragma|error))).*$", "", line).strip()
                    if line:
                        h.update(line.encode())
                        h.update(b"\n")
    return h.hexdigest()[:16]


def short(n):
    return n.split("(")[0].replace("void pion::", "")


def counters_per_kernel(d):
    """{kernel: {counter: (mean per dispatch, dispatches)}} of one --pmc pass"""
    acc = {}
    for f in glob.glob(os.path.join(OUT, d, "*", "*_counter_collection.csv")):
        per = {}
        for r in csv.DictReader(open(f)):
            k = (r["Dispatch_Id"], short(r["Kernel_Name"]), r["Counter_Name"])
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        for (_, name, ctr), v in per.items():
            acc.setdefault(name, {}).setdefault(ctr, []).append(v)
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in d2.items()} for k, d2 in acc.items()}


def durations(wl):
    out = {}
    for f in glob.glob(os.path.join(OUT, "prof_r03_" + wl, "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])}
    return out


def is_stage(k):
    return "k_stage" in k


def main():
    src = kernel_source_hash()
    # ---- stats + bench lines
    for wl in WLS:
        stats = glob.glob(os.path.join(OUT, "prof_r03_" + wl, "*", "*_kernel_stats.csv"))
        if stats:
            shutil.copy(stats[0], os.path.join(PROF, "r03_%s_kernel_stats.csv" % wl))
        log = os.path.join(OUT, "prof_r03_%s.log" % wl)
        if os.path.exists(log):
            lines = [ln for ln in open(log, errors="replace").read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(PROF, "r03_%s_under_rocprof.json" % wl), "w").write(lines[-1] + "\n")
    # ---- calibration
    n = 1 << 27
    cal = {}
    cf, cw = counters_per_kernel("pmc_r03_calib_fetch"), counters_per_kernel("pmc_r03_calib_write")
    cr = counters_per_kernel("pmc_r03_calib_rdreq")

    def find(d, frag):
        for k in d:
            if frag in k:
                return d[k]
        return None
    read_corr, write_corr, tile_corr = 2.0, 1.0, None
    a = find(cf, "k_calib<4")
    if a and "FETCH_SIZE" in a:
        read_corr = (4.0 * n * 8) / (a["FETCH_SIZE"][0] * KB)
    b = find(cw, "k_calib<1")
    if b and "WRITE_SIZE" in b:
        write_corr = (2.0 * n * 8) / (b["WRITE_SIZE"][0] * KB)
    t = find(cf, "k_calib_tiles")
    nrows = (4 * n) // 516
    if t and "FETCH_SIZE" in t:
        tile_corr = (nrows * 516 * 8.0) / (t["FETCH_SIZE"][0] * KB)
    cal = {"what": "profiles/tools/calib_traffic.hip: 8 B/lane reads / writes of known size: aligned 1 GiB planes "
                   "(k_calib) and the stage kernel's x tiling on rows of 516 doubles (k_calib_tiles: 62-cell tiles of 64 "
                   "lanes starting off the 128-byte line grid, neighbouring tiles overlapping by two cells)",
           "read_bytes_per_reported_byte_aligned_stream": read_corr,
           "read_bytes_per_reported_byte_stage_tiling": tile_corr,
           "write_bytes_per_reported_byte": write_corr}
    rr = find(cr, "k_calib<4")
    if rr:
        cal["aligned_stream_raw"] = {c: v[0] for c, v in rr.items()}
        if "TCC_EA0_RDREQ_sum" in rr:
            cal["aligned_stream_bytes_per_read_request"] = (4.0 * n * 8) / rr["TCC_EA0_RDREQ_sum"][0]
    rt = find(cr, "k_calib_tiles")
    if rt:
        cal["stage_tiling_raw"] = {c: v[0] for c, v in rt.items()}
        if "TCC_EA0_RDREQ_sum" in rt:
            cal["stage_tiling_unique_bytes_per_read_request"] = (nrows * 516 * 8.0) / rt["TCC_EA0_RDREQ_sum"][0]
    # ---- traffic per workload
    traffic = {"kernel_source_hash": src, "calibration": cal, "workloads": {},
               "note": "FETCH_SIZE counts requests that leave L2, Infinity-Cache hits included (MI355X_MICROARCH.md); read "
                       "bytes = FETCH_SIZE x 1024 x the aligned-stream correction (the stage-tiling correction is given "
                       "beside it: if the two differ, the truth for the stage kernels lies between)"}
    for wl in ("m1", "m2", "m3", "dmr2d", "mhd2d", "axi2d", "mhdaxi2d"):
        f, w = counters_per_kernel("pmc_r03_%s_fetch" % wl), counters_per_kernel("pmc_r03_%s_write" % wl)
        dur = durations(wl)
        inst = {}
        for k in sorted(f):
            if "FETCH_SIZE" not in f[k]:
                continue
            rb = f[k]["FETCH_SIZE"][0] * KB * read_corr
            wb = w.get(k, {}).get("WRITE_SIZE", (0.0, 0))[0] * KB * write_corr
            inst[k] = {"read_bytes": rb, "write_bytes": wb, "launches_seen": f[k]["FETCH_SIZE"][1],
                       "avg_ms": dur.get(k, {}).get("avg_ms")}
            if inst[k]["avg_ms"]:
                inst[k]["l2_miss_traffic_GBs"] = (rb + wb) / (inst[k]["avg_ms"] * 1e-3) / 1e9
        st = {k: v for k, v in inst.items() if is_stage(k)}
        if st:
            traffic["workloads"][wl] = {
                "per_kernel": inst,
                "stage_read_bytes_per_launch": sum(v["read_bytes"] for v in st.values()) / len(st),
                "stage_write_bytes_per_launch": sum(v["write_bytes"] for v in st.values()) / len(st),
            }
            traffic["workloads"][wl]["stage_traffic_bytes_per_launch"] = (
                traffic["workloads"][wl]["stage_read_bytes_per_launch"] + traffic["workloads"][wl]["stage_write_bytes_per_launch"])
    if "m1" in traffic["workloads"]:
        traffic["traffic_bytes_per_launch"] = traffic["workloads"]["m1"]["stage_traffic_bytes_per_launch"]
    json.dump(traffic, open(os.path.join(PROF, "r03_pmc_traffic.json"), "w"), indent=1)
    # ---- L2 / fabric
    l2 = {"kernel_source_hash": src, "workload": "m1 (512^3 GLM-MHD HLLD, fast build)", "per_kernel": {}}
    for d in ("pmc_r03_m1_rdreq", "pmc_r03_m1_l2", "pmc_r03_m1_wrreq"):
        for k, cs in counters_per_kernel(d).items():
            if is_stage(k) or "k_prepass" in k:
                l2["per_kernel"].setdefault(k, {}).update({c: v[0] for c, v in cs.items()})
    for k, c in l2["per_kernel"].items():
        if "TCC_EA0_RDREQ_sum" in c and c["TCC_EA0_RDREQ_sum"] > 0:
            c["read_requests_32B_share"] = c.get("TCC_EA0_RDREQ_32B_sum", 0.0) / c["TCC_EA0_RDREQ_sum"]
            c["read_requests_to_DRAM_share"] = c.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / c["TCC_EA0_RDREQ_sum"]
        if "TCC_HIT_sum" in c and (c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0)) > 0:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    json.dump(l2, open(os.path.join(PROF, "r03_pmc_l2_fabric.json"), "w"), indent=1)
    # ---- SQ
    sq = {"kernel_source_hash": src, "workloads": {}}
    for wl in ("m1", "m2", "m3", "dmr2d", "mhd2d", "axi2d", "mhdaxi2d"):
        per = {k: {c: v[0] for c, v in cs.items()} for k, cs in counters_per_kernel("pmc_r03_%s_sq" % wl).items() if is_stage(k)}
        for k, c in per.items():
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            if wc > 0:
                c["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
                c["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / wc
        if per:
            sq["workloads"][wl] = per
    if "m1" in sq["workloads"]:
        names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU",
                 "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD"]
        m1 = sq["workloads"]["m1"]
        sq["counters"] = {c: sum(d.get(c, 0.0) for d in m1.values()) / len(m1) for c in names}
    json.dump(sq, open(os.path.join(PROF, "r03_pmc_sq.json"), "w"), indent=1)
    # ---- clock
    clk = {"kernel_source_hash": src, "what": "GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS give-back)", "per_kernel": {}}
    g = counters_per_kernel("pmc_r03_m1_clk")
    dur = durations("m1")
    for k, cs in g.items():
        if is_stage(k) and "GRBM_GUI_ACTIVE" in cs and dur.get(k, {}).get("avg_ms"):
            clk["per_kernel"][k] = {"GRBM_GUI_ACTIVE": cs["GRBM_GUI_ACTIVE"][0], "avg_ms_unprofiled_pass": dur[k]["avg_ms"],
                                    "effective_GHz": cs["GRBM_GUI_ACTIVE"][0] / 8.0 / (dur[k]["avg_ms"] * 1e-3) / 1e9}
    json.dump(clk, open(os.path.join(PROF, "r03_clock.json"), "w"), indent=1)
    print(json.dumps({"traffic_m1": traffic.get("traffic_bytes_per_launch"), "calibration": cal,
                      "sq_m1": sq.get("counters"), "clock": clk["per_kernel"]}, indent=1)[:3000])


if __name__ == "__main__":
    sys.exit(main())

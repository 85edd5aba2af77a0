#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/tools/collect_r01.sh (gpurun_out/) into the files kept
under profiles/:  r01_bench512_kernel_stats.csv (the --stats summary), r01_pmc_traffic.json (HBM
bytes per stage-kernel launch from FETCH_SIZE / WRITE_SIZE, calibrated as MI355X_MICROARCH.md's HBM
section prescribes: separate --pmc passes, counters in KB, gfx950 read correction measured on a
known byte count with the same 8 B/lane access width)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def counter_per_kernel(d, counter):
    """mean counter value per dispatch, keyed by kernel name (summed over the counter's instances)"""
    acc = {}
    for f in glob.glob(os.path.join(OUT, d, "*", "*_counter_collection.csv")):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = (r["Dispatch_Id"], r["Kernel_Name"])
            per_dispatch[k] = per_dispatch.get(k, 0.0) + float(r["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            acc.setdefault(name, []).append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pick(d, frag):
    for k, v in d.items():
        if frag in k:
            return k, v
    raise KeyError(frag)


def main():
    stats = glob.glob(os.path.join(OUT, "prof_r01", "*", "*_kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(PROF, "r01_bench512_kernel_stats.csv"))
    KB = 1024.0  # FETCH_SIZE / WRITE_SIZE are reported in KB (derived: requests * 64 B / 1024)
    cf = counter_per_kernel("pmc_calib_fetch", "FETCH_SIZE")
    cw = counter_per_kernel("pmc_calib_write", "WRITE_SIZE")
    n = 1 << 27
    # calibration kernels: k_calib<4,1> reads 4 planes, k_calib<1,2> writes 2 planes of n doubles
    _, (f41, _) = pick(cf, "k_calib<4")
    _, (w12, _) = pick(cw, "k_calib<1")
    read_corr = (4.0 * n * 8) / (f41 * KB)
    write_corr = (2.0 * n * 8) / (w12 * KB)
    bf = counter_per_kernel("pmc_fetch", "FETCH_SIZE")
    bw = counter_per_kernel("pmc_write", "WRITE_SIZE")
    def stage_mean(d):
        """per-launch mean over every k_stage_rows instance (the first- and the second-order stage are
        separate template instances; a step launches each once)"""
        tot = sum(v * n for k, (v, n) in d.items() if "k_stage_rows" in k)
        cnt = sum(n for k, (v, n) in d.items() if "k_stage_rows" in k)
        names = sorted(k for k in d if "k_stage_rows" in k)
        return " + ".join(names), tot / cnt, cnt
    name, fs, nf = stage_mean(bf)
    _, ws, nw = stage_mean(bw)
    out = {
        "workload": "bench.py --steps 3 --warmup 1 (512^3 GLM-MHD HLLD, fast mode)",
        "kernel": name,
        "launches_averaged": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
        "FETCH_SIZE_KB_per_launch": fs,
        "WRITE_SIZE_KB_per_launch": ws,
        "calibration": {
            "what": "profiles/tools/calib_traffic.hip: 8 B/lane SoA plane reads/writes of known size (1 GiB planes)",
            "read_bytes_per_reported_byte": read_corr,
            "write_bytes_per_reported_byte": write_corr,
        },
        "read_bytes_per_launch": fs * KB * read_corr,
        "write_bytes_per_launch": ws * KB * write_corr,
    }
    out["traffic_bytes_per_launch"] = out["read_bytes_per_launch"] + out["write_bytes_per_launch"]
    json.dump(out, open(os.path.join(PROF, "r01_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())

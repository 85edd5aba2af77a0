#!/usr/bin/env python3
"""Turns the rocprofv3 output of profiles/tools/collect_r02.sh (gpurun_out/) into the files kept under
profiles/: r02_{m1,m2,m3}_kernel_stats.csv (the --stats summaries), r02_pmc_traffic.json (bytes per
stage-kernel launch from FETCH_SIZE / WRITE_SIZE, calibrated as MI355X_MICROARCH.md's HBM section prescribes:
separate --pmc passes, counters in KB, gfx950 read correction measured on a known byte count with the same
8 B/lane access width), r02_pmc_sq_stage_kernel.json (SQ counters per stage-kernel instance).  Every JSON
carries the hash of the kernel sources it was measured on (bench.py quotes it only while that still matches)."""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def kernel_source_hash():
    """the same function as bench.py's (code only: // comments and blank lines dropped)"""
    import re
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


def counter_per_kernel(d, counter):
    """mean counter value per dispatch, keyed by kernel name (summed over the counter's instances)"""
    acc = {}
    for f in glob.glob(os.path.join(OUT, d, "*", "*_counter_collection.csv")):
        per_dispatch = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = (r["Dispatch_Id"], short(r["Kernel_Name"]))
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
    src = kernel_source_hash()
    for wl in ("m1", "m2", "m3"):
        stats = glob.glob(os.path.join(OUT, "prof_r02_" + wl, "*", "*_kernel_stats.csv"))
        if stats:
            shutil.copy(stats[0], os.path.join(PROF, "r02_%s_kernel_stats.csv" % wl))
        log = os.path.join(OUT, "prof_r02_%s.log" % wl)
        if os.path.exists(log):
            lines = [ln for ln in open(log, errors="replace").read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(PROF, "r02_%s_under_rocprof.json" % wl), "w").write(lines[-1] + "\n")
    KB = 1024.0  # FETCH_SIZE / WRITE_SIZE are reported in KB (derived: requests * 64 B / 1024)
    cf = counter_per_kernel("pmc_r02_calib_fetch", "FETCH_SIZE")
    cw = counter_per_kernel("pmc_r02_calib_write", "WRITE_SIZE")
    n = 1 << 27
    _, (f41, _) = pick(cf, "k_calib<4")
    _, (w12, _) = pick(cw, "k_calib<1")
    read_corr = (4.0 * n * 8) / (f41 * KB)
    write_corr = (2.0 * n * 8) / (w12 * KB)
    bf = counter_per_kernel("pmc_r02_fetch", "FETCH_SIZE")
    bw = counter_per_kernel("pmc_r02_write", "WRITE_SIZE")
    inst = {}
    for k in sorted(bf):
        if "k_stage_rows" in k:
            inst[k] = {"read_bytes": bf[k][0] * KB * read_corr, "write_bytes": bw[k][0] * KB * write_corr,
                       "launches": [bf[k][1], bw[k][1]]}
    tot_r = sum(v["read_bytes"] for v in inst.values()) / max(1, len(inst))
    tot_w = sum(v["write_bytes"] for v in inst.values()) / max(1, len(inst))
    other = {k: {"read_bytes": bf[k][0] * KB * read_corr, "write_bytes": bw.get(k, (0, 0))[0] * KB * write_corr}
             for k in bf if any(s in k for s in ("k_prepass", "k_bc", "k_cooling"))}
    out = {
        "workload": "bench.py --steps 3 --warmup 1 (512^3 GLM-MHD HLLD, fast mode)",
        "kernel_source_hash": src,
        "per_instance": inst,
        "other_kernels": other,
        "calibration": {
            "what": "profiles/tools/calib_traffic.hip: 8 B/lane SoA plane reads/writes of known size (1 GiB planes)",
            "read_bytes_per_reported_byte": read_corr,
            "write_bytes_per_reported_byte": write_corr,
        },
        "read_bytes_per_launch": tot_r,
        "write_bytes_per_launch": tot_w,
        "traffic_bytes_per_launch": tot_r + tot_w,
        "note": "mean of the first- and the second-order stage instance (a step launches each once); FETCH_SIZE "
                "counts requests that leave L2, Infinity-Cache hits included (MI355X_MICROARCH.md)",
    }
    json.dump(out, open(os.path.join(PROF, "r02_pmc_traffic.json"), "w"), indent=1)
    # SQ counters
    names = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU",
             "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD"]
    sq = {}
    for c in names:
        for k, (v, cnt) in counter_per_kernel("pmc_r02_sq", c).items():
            if "k_stage_rows" in k:
                sq.setdefault(k, {})[c] = v
    mean = {c: sum(d.get(c, 0.0) for d in sq.values()) / max(1, len(sq)) for c in names}
    json.dump({"what": "SQ counters per stage-kernel launch, bench.py --steps 3 --warmup 1 (512^3 GLM-MHD HLLD, fast "
                       "mode), rocprofv3 --pmc in one pass with --kernel-trace only",
               "kernel_source_hash": src, "per_instance": sq, "counters": mean},
              open(os.path.join(PROF, "r02_pmc_sq_stage_kernel.json"), "w"), indent=1)
    print(json.dumps({"traffic": out["traffic_bytes_per_launch"], "read": tot_r, "write": tot_w, "sq_mean": mean}, indent=1))


if __name__ == "__main__":
    sys.exit(main())

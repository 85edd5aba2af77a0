#!/usr/bin/env python3
"""Copies what profiles/tools/run_final_r03.sh left under gpurun_out/final_r03/ into the files kept under profiles/:
r03_default_run.json, r03_bench_<workload>.json, r03_bench_gpus2_shm_one_gpu.json and r03_loopback_slab_shapes.json."""
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
PROF = os.path.join(HERE, "..")
SRC = os.path.join(HERE, "..", "..", "gpurun_out", "final_r03")


def last_json(path):
    with open(path) as f:
        lines = [x for x in f.read().strip().splitlines() if x.startswith("{")]
    return json.loads(lines[-1])


def main():
    d = last_json(os.path.join(SRC, "default_run.json"))
    with open(os.path.join(PROF, "r03_default_run.json"), "w") as f:
        f.write(json.dumps(d) + "\n")
    for wl in ("m2", "m3", "mhd8", "dmr2d", "mhd2d", "axi2d", "mhdaxi2d"):
        p = os.path.join(SRC, "bench_%s.json" % wl)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            with open(os.path.join(PROF, "r03_bench_%s.json" % wl), "w") as f:
                f.write(json.dumps(last_json(p)) + "\n")
    p = os.path.join(SRC, "bench_gpus2_shm.json")
    if os.path.exists(p) and os.path.getsize(p) > 0:
        with open(os.path.join(PROF, "r03_bench_gpus2_shm_one_gpu.json"), "w") as f:
            f.write(json.dumps(last_json(p)) + "\n")
    shapes = {}
    n1 = d["ms_per_step"]
    for n, nz in ((2, 256), (4, 128), (8, 64)):
        a, b = last_json(os.path.join(SRC, "slab_nz%d.json" % nz)), last_json(os.path.join(SRC, "loop_nz%d.json" % nz))
        shapes["N=%d (512x512x%d)" % (n, nz)] = {
            "plain_ms_per_step": a["ms_per_step"], "loopback_ms_per_step": b["ms_per_step"],
            "plain_kernel_ms": a["roofline"]["kernel_ms"], "loopback_kernel_ms_per_stage": b["roofline"]["kernel_ms"],
            "n1_over_plain": n1 / a["ms_per_step"], "n1_over_loopback": n1 / b["ms_per_step"],
            "per_cell_rate_vs_n1_plain": n1 / a["ms_per_step"] / n}
    out = {"what": "bench.py --no-cpu-baseline --no-parity-build [--loopback] --nz NZ on one MI355X (one box, back to back): the "
                   "slab of one rank of an N-rank run of the 512^3 headline (NZ = 512 / N), plain and with the split stages + "
                   "RCCL send/recv to self; N = 1 on the same box: profiles/r03_default_run.json",
           "n1_ms_per_step": n1, "shapes": shapes}
    with open(os.path.join(PROF, "r03_loopback_slab_shapes.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: (round(v["plain_ms_per_step"], 3), round(v["loopback_ms_per_step"], 3), round(v["n1_over_loopback"], 2),
                          round(v["per_cell_rate_vs_n1_plain"], 3)) for k, v in shapes.items()}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X flux-update path.

Workload: 3-D GLM-MHD Stone blast wave, 512^3, HLLD + FKJ98
viscosity, periodic, second order in space and time, fp64.  One "step" = one full
second-order time step of the reference's time loop: calculate_timestep (CFL
reduction) + advance_time (2 fused stages + 2 boundary updates).

  python bench.py --gpus N --steps K --warmup W

N>1 is launched by torch.distributed.run, one rank per GPU; the 512^3 grid is split
into z-slabs (strong scaling) with RCCL halo exchange.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from pion_amd import abi, driver, lib, problems, slab  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def kernel_source_hash():
    """hash of the CODE of pion_amd/csrc (// comments and blank lines dropped, so that a comment edit does not
    invalidate the committed PMC measurements; the same function as profiles/tools/summarize_r03.py)"""
    import hashlib
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


def baseline_metric():
    """BASELINE.json's metric string, verbatim (the file travels with the repo)"""
    try:
        with open(os.path.join(ROOT, "BASELINE.json"), encoding="utf-8") as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "Mcell-updates/s on 3D ideal-MHD 512\u00b3 uniform grid; achieved HBM GB/s vs peak"


def _cpu_case(workload, ng, eqn):
    """the benchmark workloads on an ng[0] x ng[1] x ng[2] grid (CPU baselines; strict arithmetic)"""
    if workload == "m1":
        eq = abi.EQGLM if eqn == "glm" else abi.EQMHD
        cfg, _ = problems.mhd_blastwave(4, 3, eq, abi.FLUX_RS_HLLD, strict_fp=1)
        for a in range(3):
            cfg.ng[a] = ng[a]
        cfg.dx = 1.0 / ng[0]
        return cfg, problems.fill_mhd_blastwave(cfg), None, None
    if workload == "m2":
        cfg, _ = problems.hd_blast_octant(4, 3, solver=abi.FLUX_RSroe, strict_fp=1)
        L = cfg.dx * 4
        for a in range(3):
            cfg.ng[a] = ng[a]
        cfg.dx = L / ng[0]
        return cfg, problems.fill_hd_blast_octant(cfg, ng[0] / 32.0), None, None
    cfg, _, _, _ = problems.wind3d(8, strict_fp=1)
    L = cfg.dx * 8
    for a in range(3):
        cfg.ng[a] = ng[a]
    cfg.dx = L / ng[0]
    P, wind, dt_lim = problems.fill_wind3d(cfg, ng[0])
    return cfg, P, wind, dt_lim


def cpu_worker(spec):
    """`bench.py --cpu-worker kind,workload,eqn,nx,ny,nz,budget_s`: one CPU process of the baseline (started by
    cpu_baseline() BEFORE or beside the GPU work, never touches the GPU).  Prints one JSON line."""
    kind, workload, eqn, nx, ny, nz, budget = spec.split(",")
    ng, budget = [int(nx), int(ny), int(nz)], float(budget)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_backends import CpuSim
    cfg, P, wind, dt_lim = _cpu_case(workload, ng, eqn)
    with CpuSim(cfg, kind) as o:
        if wind is not None:
            from pion_amd import cooling
            o.set_cooling_tables(*cooling.build_tables(cfg.min_temp, cfg.max_temp))
            if wind[0].size:
                o.set_wind_cells(*wind)
        sc = driver.SimControl(o, cfg)
        sc.first_step_dt_limit = dt_lim
        sc.init(P)
        sc.calculate_timestep()
        sc.advance_time()  # untimed first step
        t0 = time.perf_counter()
        steps = 0
        while True:
            sc.calculate_timestep()
            sc.advance_time()
            steps += 1
            el = time.perf_counter() - t0
            if el > budget or steps >= 200:
                break
    print(json.dumps({"cells": ng[0] * ng[1] * ng[2], "steps": steps, "seconds": el}))


def _run_cpu_workers(specs):
    """start one process per spec at once, wait for all -> list of their JSON results"""
    import subprocess
    env = dict(os.environ)
    env["PION_NO_TORCH"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", sp], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, env=env, text=True) for sp in specs]
    out = []
    for p in procs:
        so, _ = p.communicate()
        lines = [ln for ln in so.splitlines() if ln.startswith("{")]
        out.append(json.loads(lines[-1]) if (p.returncode == 0 and lines) else None)
    return out


def cpu_baseline(n, eqn, budget_s=10.0):
    """CPU baselines of SURVEY 8(d) on this box's host cores (the reference has no threads: "N cores" = N
    processes):
      value        M1 on ALL cores: one process per core, each a z-slab n x n x (n/C) of the n^3 problem,
                   run as C independent periodic slabs (no halo exchange between them: an upper bound on an MPI
                   run of the same decomposition); sum of the processes' rates
      single_core  the same problem, n^3, one process
      m2 / m3      single-core rates of the other two workloads at 48^3 (m3: the stellar-wind source is not in
                   oracle/_ref -- GSL -- so that one is the oracle, kind "port")
    kind "reference": oracle/_ref/libpion_ref.so (the reference's own solver objects, -O3 -DSERIAL, under
    oracle/ref_harness.cpp's time_integrator loops) when it is present, else kind "port": oracle/liboracle.so."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_backends import have_oracle, have_ref
    if not (have_ref() or have_oracle()):
        return None
    kind = "ref" if have_ref() else "orc"
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    C = 1
    while C * 2 <= min(cores, 16) and n % (C * 2) == 0 and n // (C * 2) >= 4:   # (a one-GPU box's CPU share is 16 cores)
        C *= 2
    rate = lambda r: r["cells"] * r["steps"] / r["seconds"] / 1e6
    res_all = _run_cpu_workers(["%s,m1,%s,%d,%d,%d,%g" % (kind, eqn, n, n, n // C, budget_s)] * C)
    if any(r is None for r in res_all):
        return None
    ns = min(n, 128)   # the single-process samples stay at 128^3 (6 s per step; 256^3 would take a minute per step)
    singles = _run_cpu_workers(["%s,m1,%s,%d,%d,%d,%g" % (kind, eqn, ns, ns, ns, budget_s),
                                "orc,m1,%s,%d,%d,%d,%g" % (eqn, ns, ns, ns, budget_s),
                                "%s,m2,%s,48,48,48,%g" % (kind, eqn, 0.6 * budget_s),
                                "orc,m3,%s,48,48,48,%g" % (eqn, 0.6 * budget_s)])
    lib = "oracle/_ref/libpion_ref.so (reference objects, -O3 -DSERIAL)" if kind == "ref" else "oracle/liboracle.so"
    out = {"value": sum(rate(r) for r in res_all), "unit": "Mcell-updates/s", "cores": C,
           "kind": "reference" if kind == "ref" else "port",
           "sample": "M1 (GLM-MHD HLLD blast) %d^3 as %d independent periodic z-slabs %dx%dx%d (each computes %d planes "
                     "with its ghosts for %d counted: %.0f %% overhead of the decomposition), one process per core, "
                     "%d-%d steps each in %.0f s; %s; host has %d cores available; single_core_value / port_value: "
                     "%d^3 in one process" % (
                         n, C, n, n, n // C, n // C + 4, n // C, 400.0 / (n // C),
                         min(r["steps"] for r in res_all), max(r["steps"] for r in res_all),
                         max(r["seconds"] for r in res_all), lib, cores, ns)}
    if singles[0]:
        out["single_core_value"] = rate(singles[0])
    if singles[1]:
        out["port_value"] = rate(singles[1])
    if singles[2]:
        out["m2_single_core"] = {"value": rate(singles[2]), "kind": out["kind"], "sample": "M2 Euler Roe-CV octant blast 48^3"}
    if singles[3]:
        out["m3_single_core"] = {"value": rate(singles[3]), "kind": "port",
                                 "sample": "M3 Wind3D FVS + cooling 48^3 (oracle: the stellar-wind source is not in oracle/_ref)"}
    return out


def parity_build_run(args, cfg, device, dt_lim, steps=3, warmup=1):
    """`steps` steps of the benchmark workload with the bit-parity kernels (strict_fp = 1)"""
    cfg.strict_fp = 1
    sim = lib.GpuSim(cfg, device)
    try:
        if args.workload == "m1":
            P = problems.fill_mhd_blastwave(cfg)
        elif args.workload == "m2":
            P = problems.fill_hd_blast_octant(cfg, args.n / 32.0)
        elif args.workload == "dmr2d":
            _, P = problems.double_mach_reflection(args.n, strict_fp=1)
        elif args.workload == "mhd2d":
            P = problems.fill_mhd_blastwave(cfg)
        elif args.workload == "axi2d":
            _, P = problems.blast_axi2d(args.n, abi.EQEUL, abi.FLUX_RSroe, strict_fp=1)
        elif args.workload == "mhdaxi2d":
            _, P = problems.blast_axi2d(args.n, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=1)
        else:
            from pion_amd import cooling
            P, (widx, wst), dt_lim = problems.fill_wind3d(cfg, args.n)
            sim.set_cooling_tables(*cooling.build_tables(cfg.min_temp, cfg.max_temp))
            sim.set_wind_cells(widx, wst)
        sc = driver.SimControl(sim, cfg)
        sc.first_step_dt_limit = dt_lim
        sc.init(P)
        del P
        for _ in range(warmup):
            sc.calculate_timestep()
            sc.advance_time()
        sim.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sc.calculate_timestep()
            sc.advance_time()
        sim.synchronize()
        el = time.perf_counter() - t0
    finally:
        sim.close()
    ncell = cfg.ng[0] * cfg.ng[1] * cfg.ng[2]
    return {"value": ncell * steps / el / 1e6, "unit": "Mcell-updates/s", "ms_per_step": el / steps * 1e3,
            "steps": steps, "warmup": warmup,
            "fp_mode": "strict (-ffp-contract=off, reference operation order; bit-identical to the oracle)"}


class _stdout_to_stderr:
    """gloo and RCCL print connection / version banners on fd 1 while a process group or communicator is
    created; bench.py's stdout carries exactly one JSON line, so fd 1 points at fd 2 meanwhile"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *a):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def exchange_digests(A, nbc_z, nz):
    """hashes of the four z plane groups of one rank's state array A[nvar][nz + 2 nbc][ny + 2 nbc][nx + 2 nbc]"""
    import hashlib

    def dig(x):
        return hashlib.blake2b(np.ascontiguousarray(x).tobytes(), digest_size=16).hexdigest()
    nb = int(nbc_z)
    return {"ghost_lo": dig(A[:, 0:nb]), "own_lo": dig(A[:, nb:2 * nb]),
            "own_hi": dig(A[:, nz:nz + nb]), "ghost_hi": dig(A[:, nz + nb:nz + 2 * nb]),
            "finite": bool(np.isfinite(A).all())}


def compare_exchange(every, periodic_z):
    """every[r] = exchange_digests of rank r; returns the list of faces whose ghosts are not the neighbour's planes"""
    world = len(every)
    bad = []
    for r in range(world):
        lo, hi = r - 1, r + 1
        if periodic_z:
            lo, hi = lo % world, hi % world
        if lo >= 0 and every[r]["ghost_lo"] != every[lo]["own_hi"]:
            bad.append("rank %d lower ghosts != rank %d top planes" % (r, lo))
        if hi < world and every[r]["ghost_hi"] != every[hi]["own_lo"]:
            bad.append("rank %d upper ghosts != rank %d bottom planes" % (r, hi))
        if not every[r]["finite"]:
            bad.append("rank %d holds non-finite values" % r)
    return bad


def check_exchange(sim, cfg, rank, world, periodic_z, dist):
    """After the last step of an N > 1 run (untimed): every rank's z ghost planes must be, bit for bit, the planes its
    neighbour owns -- whole planes, x / y ghost cells included, which is what the exchange moves.  Hashes of the four
    plane groups of every rank go through the bootstrap group; rank 0 compares.  This is the check that the transport
    moved the right data on THIS machine (development had one-GPU boxes only)."""
    mine = exchange_digests(sim.download(0), int(cfg.nbc), int(cfg.ng[2]))
    every = [None] * world
    dist.all_gather_object(every, mine)
    if rank != 0:
        return None
    bad = compare_exchange(every, periodic_z)
    if bad:
        sys.stderr.write("bench.py: EXCHANGE CHECK FAILED: %s\n" % "; ".join(bad))
        return "FAILED: " + "; ".join(bad)
    return "ok: after the last step every rank's z ghost planes equal its neighbours' owned planes bit for bit (%d faces)" % (
        2 * world if periodic_z else 2 * (world - 1))


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--grid", dest="n", type=int, default=512, help="cells per axis (headline: 512); use --grid under torch.distributed.run, whose own parser finds --n ambiguous")
    ap.add_argument("--eqn", default="glm", choices=["glm", "mhd"])
    ap.add_argument("--strict", type=int, default=0, help="1 = bit-parity kernels (no FMA contraction)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="do not bracket the launches with HIP events (A/B of the event overhead; roofline fields are 0)")
    ap.add_argument("--no-parity-build", action="store_true", help="skip the strict-build throughput run")
    ap.add_argument("--no-exchange-check", action="store_true",
                    help="N > 1: skip the untimed end-of-run check that every rank's z ghosts equal its neighbours' planes")
    ap.add_argument("--cpu-n", type=int, default=256,
                    help="cells per axis of the all-cores CPU baseline sample (16 slabs of n x n x n/16; the single-process "
                         "samples use min(n, 128))")
    ap.add_argument("--workload", default="m1", choices=["m1", "m2", "m3", "dmr2d", "mhd2d", "axi2d", "mhdaxi2d"],
                    help="m1 (default, the headline): MHD blast; m2: 3-D Euler Roe-CV octant Sedov blast (SURVEY 8d); "
                         "m3: Wind3D single level, FVS + cooling 8 + stellar wind (single GPU); dmr2d / mhd2d: BASELINE "
                         "configs 2 and 3 (2-D double Mach reflection, Euler Roe-CV, --grid = cells along x, ny = nx / 3.25; "
                         "2-D GLM-MHD blast wave, HLLD, --grid = nx, ny = 1.5 nx), one GPU; axi2d / mhdaxi2d: axisymmetric (z,R) "
                         "blast, Euler Roe-CV / GLM-MHD HLLD (SURVEY 8f-4; --grid = cells along z, nR = nz / 2), one GPU.  All "
                         "but m1 are extra rows for DESIGN.md")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="N>1 transport: nccl (= RCCL over xGMI, one rank per GPU) or gloo with the halo staged "
                         "through pinned host buffers (rehearsal of the multi-rank path, ranks may share a GPU)")
    ap.add_argument("--transport", default="host", choices=["host", "shm", "torch"],
                    help="who drives the time loop and the N>1 transport: host (default) = the C++ host layer "
                         "(pion_host::sim_control_gpu + slab_comm_rccl: RCCL send/recv groups and the dt all-reduce "
                         "issued from C++; torch.distributed/gloo only broadcasts the ncclUniqueId and hosts the "
                         "timing barrier); shm = the same C++ time loop with pion_host::slab_comm_shm (halo planes staged "
                         "through pinned host buffers and POSIX shared memory: no RCCL; ranks may share a GPU; the fall-back "
                         "when no RCCL communicator can be made); torch = the Python driver with torch.distributed P2P "
                         "(round-1 path; required for --backend gloo)")
    ap.add_argument("--nz", type=int, default=0,
                    help="m1 only: cells along z if not --n (with --loopback and nz = n/N this is exactly the slab, the "
                         "launches and the transfers of one rank of an N-rank run)")
    ap.add_argument("--loopback", action="store_true",
                    help="one rank, m1 only: the periodic z faces become slab faces that exchange with the rank itself "
                         "over RCCL (send/recv to self on the comm stream, split stages) -- the whole N>1 code path on a "
                         "one-GPU box; the step cost over the plain run is the split + exchange overhead")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run) BEFORE anything in this process touches the GPU, relay rank 0's JSON
        # line (the children inherit stdout) and exit with the launcher's code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
              [("--grid" if a == "--n" else a) for a in sys.argv[1:]]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d started with WORLD_SIZE=%d" % (args.gpus, world))
    eq = abi.EQGLM if args.eqn == "glm" else abi.EQMHD
    solver = abi.FLUX_RS_HLLD

    comm = None
    torch = None
    use_host = (args.transport in ("host", "shm") and args.backend == "nccl")
    use_shm = (args.transport == "shm")
    if world > 1 and use_host:
        import torch
        import torch.distributed as dist
        with _stdout_to_stderr():
            dist.init_process_group("gloo")   # bootstrap (ncclUniqueId / segment name) and timing barrier only
            dist.barrier()
    elif world > 1:
        import torch
        import torch.distributed as dist
        with _stdout_to_stderr():
            if args.backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                local_rank = local_rank % torch.cuda.device_count()
                torch.cuda.set_device(local_rank)
                dist.init_process_group("gloo")
            dist.barrier()

    loopback = args.loopback and world == 1 and args.workload == "m1"
    if loopback and not use_host:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (23400 + os.getpid() % 4000),
                                rank=0, world_size=1, device_id=torch.device("cuda", 0))

    n = args.n
    wl_name = None
    dt_lim = None
    periodic_z = True
    hs = None

    transport_note = []

    def make_sim(cfg, periodic_z):
        """the handle the benchmark drives: owned by the C++ host layer (transport host / shm) or by Python"""
        if not use_host:
            return None, lib.GpuSim(cfg, local_rank)
        from pion_amd import host_rccl
        ngpu = 1
        if world > 1:
            import torch
            ngpu = max(1, torch.cuda.device_count())
        dev = local_rank % ngpu
        if world > 1 and (use_shm or ngpu < world):
            # host-staged transport: ranks agree on a shared-memory name through the bootstrap group
            box = ["/pion_bench_%d_%d" % (os.getpid(), time.time_ns() % 1000000007) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            transport_note.append("shm")
            h = host_rccl.HostSim(cfg, dev, rank=rank, world=world, periodic_z=periodic_z, shm_name=box[0])
            return h, lib.GpuSim(cfg, borrowed_handle=h.gpu_handle())
        uid = None
        if world > 1:
            box = [host_rccl.new_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        elif loopback:
            uid = host_rccl.new_unique_id()
        try:
            h = host_rccl.HostSim(cfg, dev, rank=rank, world=world, periodic_z=periodic_z, unique_id=uid)
            ok = 1
        except RuntimeError as e:
            sys.stderr.write("bench.py rank %d: RCCL communicator failed (%s)\n" % (rank, e))
            h, ok = None, 0
        if world > 1:
            # every rank must take the same transport: fall back to the host-staged one if ANY rank failed
            oks = [None] * world
            dist.all_gather_object(oks, ok)
            if not all(oks):
                if h is not None:
                    h.close()
                box = ["/pion_bench_%d_%d" % (os.getpid(), time.time_ns() % 1000000007) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                transport_note.append("shm (fall-back: no RCCL communicator)")
                h = host_rccl.HostSim(cfg, dev, rank=rank, world=world, periodic_z=periodic_z, shm_name=box[0])
        elif h is None:
            raise SystemExit("bench.py: RCCL loop-back communicator failed")
        return h, lib.GpuSim(cfg, borrowed_handle=h.gpu_handle())

    if args.workload == "m1":
        cfg_g, _ = problems.mhd_blastwave(4, 3, eq, solver, strict_fp=args.strict)  # template
        cfg_g.ng[0] = cfg_g.ng[1] = cfg_g.ng[2] = n
        if args.nz:
            cfg_g.ng[2] = args.nz
        cfg_g.dx = 1.0 / n
        cfg = slab.slab_config(cfg_g, rank, world)
        if loopback:
            cfg.bc_type[4] = cfg.bc_type[5] = abi.BC_SLAB
        P = problems.fill_mhd_blastwave(cfg)
        hs, sim = make_sim(cfg, True)
    elif args.workload == "m2":
        # BASELINE configs[3]: 3-D HD blast 512^3, z-slabs over the GPUs (physical z faces on the end ranks)
        cfg_g, _ = problems.hd_blast_octant(4, 3, solver=abi.FLUX_RSroe, strict_fp=args.strict)   # template
        L = cfg_g.dx * 4
        cfg_g.ng[0] = cfg_g.ng[1] = cfg_g.ng[2] = n
        cfg_g.dx = L / n
        cfg = slab.slab_config(cfg_g, rank, world)
        P = problems.fill_hd_blast_octant(cfg, n / 32.0)
        wl_name = "M2: 3-D Euler octant Sedov blast %d^3, Roe-CV + FKJ98 0.1, reflecting/outflow, OA2/OA2" % n
        periodic_z = False
        hs, sim = make_sim(cfg, False)
        eq = cfg.eqntype
    elif args.workload in ("dmr2d", "mhd2d", "axi2d", "mhdaxi2d"):
        # BASELINE configs[1] / [2] on one GPU (2-D grids do not shard along z); SURVEY 8f-4: cylindrical (z,R) grids
        if world > 1:
            raise SystemExit("bench.py: the 2-D workloads run on one GPU")
        if args.workload == "axi2d":
            cfg, P = problems.blast_axi2d(n, abi.EQEUL, abi.FLUX_RSroe, strict_fp=args.strict)
            wl_name = ("AXI2D: axisymmetric (z,R) blast %d x %d, Euler Roe-CV + FKJ98 0.1, outflow / axis, OA2/OA2"
                       % (cfg.ng[0], cfg.ng[1]))
        elif args.workload == "mhdaxi2d":
            cfg, P = problems.blast_axi2d(n, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=args.strict)
            wl_name = ("MHDAXI2D: axisymmetric (z,R) blast %d x %d, GLM-MHD HLLD + FKJ98 0.1, outflow / axis, OA2/OA2"
                       % (cfg.ng[0], cfg.ng[1]))
        elif args.workload == "dmr2d":
            cfg, P = problems.double_mach_reflection(n, strict_fp=args.strict)
            wl_name = ("DMR2D: double Mach reflection %d x %d, Euler Roe-CV + FKJ98 0.1, inflow/outflow/reflecting/DMR, "
                       "OA2/OA2" % (cfg.ng[0], cfg.ng[1]))
        else:
            cfg, _ = problems.mhd_blastwave(4, 2, abi.EQGLM, abi.FLUX_RS_HLLD, strict_fp=args.strict)
            cfg.ng[0], cfg.ng[1] = n, (3 * n) // 2
            cfg.dx = 1.0 / n
            cfg.xmin[1] = -0.75
            P = problems.fill_mhd_blastwave(cfg)
            wl_name = "MHD2D: GLM-MHD Stone blast wave %d x %d, HLLD + FKJ98 eta 0.1, periodic, OA2/OA2" % (cfg.ng[0], cfg.ng[1])
        cfg_g = cfg
        periodic_z = False
        hs, sim = make_sim(cfg, False)
        eq = cfg.eqntype
    else:
        # BASELINE configs[4]: Wind3D with the cooling source term, z-slabs over the GPUs
        from pion_amd import cooling
        cfg_g, _, _, _ = problems.wind3d(8, strict_fp=args.strict)   # template
        L = cfg_g.dx * 8
        cfg_g.ng[0] = cfg_g.ng[1] = cfg_g.ng[2] = n
        cfg_g.dx = L / n
        cfg = slab.slab_config(cfg_g, rank, world)
        P, (widx, wst), dt_lim = problems.fill_wind3d(cfg, n)
        wl_name = ("M3: Wind3D single level %d^3, Euler + tracer, FVS + FKJ98 0.15, cooling 8, stellar wind, "
                   "reflecting/one-way, OA2/OA2" % n)
        periodic_z = False
        hs, sim = make_sim(cfg, False)
        sim.set_cooling_tables(*cooling.build_tables(cfg.min_temp, cfg.max_temp))
        if widx.size:
            sim.set_wind_cells(widx, wst)   # (ranks away from the source hold no wind cell)
        eq = cfg.eqntype
    if hs is not None:
        hs.init(P, first_step_dt_limit=dt_lim)
        step = hs.step
        finish_halo = hs.finish_halo
    else:
        if world > 1 or loopback:
            comm = slab.SlabComm(rank, world, periodic_z, sim.halo_count(), torch.device("cuda", local_rank),
                                 loopback=loopback)
            comm.use_streams(sim)   # exchange under the interior part of each stage, no host waits
        sc = driver.SimControl(sim, cfg, comm=comm)
        sc.first_step_dt_limit = dt_lim
        sc.init(P)
        finish_halo = sc.finish_halo

        def step():
            sc.calculate_timestep()
            sc.advance_time()
    del P

    def barrier():
        finish_halo()
        sim.synchronize()
        if hs is None and (world > 1 or loopback):
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    # kernel durations for the roofline: HIP events around every launch, on the stream it is launched on, over
    # a few MORE steps after the timed region (event records between dependent kernels on two streams cost
    # ~0.5 ms per step in the slab path, so they stay out of `value`)
    ev_steps = 0
    tm = {"stage_ms": 0.0, "stage_n": 0, "prepass_ms": 0.0, "bc_ms": 0.0, "dt_ms": 0.0}
    if not args.no_kernel_timing:
        ev_steps = max(1, min(3, args.steps))
        sim.enable_timing(True)
        for _ in range(ev_steps):
            step()
        barrier()
        tm = sim.get_timing()
    exchange_check = None
    if world > 1 and not args.no_exchange_check:
        exchange_check = check_exchange(sim, cfg, rank, world, periodic_z, dist)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device="cuda" if (args.backend == "nccl" and hs is None) else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    if rank == 0:
        ncell = cfg_g.ng[0] * cfg_g.ng[1] * cfg_g.ng[2]
        value = ncell * args.steps / el / 1e6
        nvar = cfg.nvar
        # algorithmic bytes of one stage launch on this rank: a step moves 5*nvar*8 B per
        # cell in two launches (stage 1: read P, write Ph; stage 2: read P and Ph, write P)
        cells_rank = cfg.ng[0] * cfg.ng[1] * cfg.ng[2]
        alg_bytes = 2.5 * nvar * 8 * cells_rank
        # (N > 1 splits a stage into interior + 2 z-boundary launches: sum them per stage)
        stage_ms = tm["stage_ms"] * tm["stage_n"] / (2.0 * ev_steps) if ev_steps else 0.0
        achieved = alg_bytes / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
        # Bytes per stage launch that left L2, from the PMC passes of this same command (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 read correction calibrated on 8 B/lane accesses):
        # profiles/r03_pmc_traffic.json, written by profiles/tools/summarize_r03.py.  Counters cannot be read
        # from inside this process, so the committed measurement is quoted -- only for the workload it was
        # taken on (512^3, GLM, fast mode, 1 GPU) and only while the kernel sources still hash to what it was
        # taken on; otherwise null.
        traffic = None
        valu = None
        pdir = os.path.join(ROOT, "profiles")
        tfile, sfile = os.path.join(pdir, "r03_pmc_traffic.json"), os.path.join(pdir, "r03_pmc_sq.json")
        if (world == 1 and not loopback and n == 512 and not args.nz and eq == abi.EQGLM and not args.strict
                and args.workload == "m1"):
            src = kernel_source_hash()
            if os.path.exists(tfile):
                with open(tfile) as f:
                    t = json.load(f)
                if t.get("kernel_source_hash") == src:
                    traffic = t.get("traffic_bytes_per_launch")
            # where the time goes (the kernel is fp64-VALU bound, not HBM bound): instructions per launch,
            # share of wave cycles issuing VALU / parked on s_waitcnt, from the committed SQ counter pass
            if os.path.exists(sfile):
                with open(sfile) as f:
                    t = json.load(f)
                if t.get("kernel_source_hash") == src:
                    c = t["counters"]
                    valu = {"valu_insts_per_launch": c["SQ_INSTS_VALU"],
                            "valu_active_frac_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                            "waitcnt_frac_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                            "waves_per_simd": 2, "source": "profiles/r03_pmc_sq.json"}
                    # the roofline that actually binds: every wave-level VALU instruction of this fp64 kernel holds
                    # its SIMD's vector pipe for 4 cycles (16 fp64 lanes per clock per SIMD, 78.6 TFLOP/s spec);
                    # fraction of that issue capacity the launch used, at the nominal 2.4 GHz
                    if stage_ms > 0:
                        valu["fp64_issue_frac_at_2.4GHz"] = c["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9 * stage_ms * 1e-3)
        out = {
            "metric": baseline_metric(),
            "value": value, "unit": "Mcell-updates/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name or "M1: 3-D %s Stone blast wave %d^3, HLLD + FKJ98 eta 0.1, periodic, OA2/OA2"
                                   % ("GLM-MHD (nvar 9)" if eq == abi.EQGLM else "ideal MHD (nvar 8)", n),
                       "grid": [int(v) for v in cfg_g.ng[:3]], "nvar": nvar, "decomposition": "z-slab x%d" % world, "transport": ("RCCL send/recv to self (loopback)" if loopback else "none" if world == 1 else
                                     ((transport_note[0] + ": pinned host buffers + POSIX shared memory" if transport_note else "RCCL P2P") if args.backend == "nccl" else "gloo via pinned host buffers (rehearsal)"))
                       + ("; time loop and transport issued from C++ (libpion_host)" if hs is not None else "; Python driver")
                       + ("; NOTE: RCCL transfers between two different GPUs were never exercised in development (one-GPU boxes; "
                          "tests/test_gpu_host_two_ranks.py covers them where two devices exist) -- `--transport shm` (host-staged) "
                          "and `--transport torch` are the verified fall-backs" if (world > 1 and not transport_note and hs is not None) else ""),
                       "fp_mode": "strict (no FMA)" if args.strict else "fast (FMA contraction)",
                       "exchange_check": exchange_check},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling_6290GBs": achieved / 6290.0,
                         "step_frac": value * 1e6 * 5 * nvar * 8 / (HBM_PEAK_GBS * 1e9 * world),
                         "step_frac_what": "SURVEY 8(d): whole-step rate x 5 nvar 8 B per cell-update / (peak x GPUs): "
                                           "prepass, boundary and reduction launches included",
                         "traffic": traffic, "traffic_unit": "bytes per launch that left L2 (PMC FETCH_SIZE + WRITE_SIZE, calibrated; profiles/r03_pmc_traffic.json; null when the kernel sources have changed since)",
                         "kernel": {"m1": "k_stage_rows2<GLM,0,HLLD>" if eq == abi.EQGLM else "k_stage_rows2<MHD,0,HLLD>",
                                    "m2": "k_stage_rows2<EUL,0,Roe-CV>", "m3": "k_stage_rows2<EUL,1,FVS>",
                                    "dmr2d": "2-D stage kernel <EUL,0,Roe-CV>", "mhd2d": "2-D stage kernel <GLM,0,HLLD>",
                                    "axi2d": "2-D cylindrical stage kernel <EUL,0,Roe-CV,CYL>",
                                    "mhdaxi2d": "2-D cylindrical stage kernel <GLM,0,HLLD,CYL>"}[args.workload]
                                   + " (first-order + second-order instance, mean per launch)",
                         "kernel_ms": stage_ms, "kernel_ms_from": "HIP events over %d steps after the timed region" % ev_steps, "launches_per_stage": (tm["stage_n"] / (2.0 * ev_steps) if ev_steps else 0.0), "prepass_ms": tm["prepass_ms"], "bc_ms": tm["bc_ms"],
                         "dt_ms": tm["dt_ms"], "algorithmic_bytes_per_launch": alg_bytes, "issue": valu},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "m1":
            out["cpu_baseline"] = cpu_baseline(args.cpu_n, args.eqn)
        if world == 1 and not loopback and not args.strict and not args.no_parity_build:
            # the same workload through the PARITY build (strict_fp=1: -ffp-contract=off, the reference's
            # operation order, bit-identical to the oracle): its throughput beside the headline
            sim.close()
            sim = None
            if hs is not None:
                hs.close()
                hs = None
            out["parity_build"] = parity_build_run(args, cfg, local_rank, dt_lim)
        print(json.dumps(out))
    if sim is not None:
        sim.close()
    if hs is not None:
        hs.close()
    if world > 1 or (loopback and not use_host):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

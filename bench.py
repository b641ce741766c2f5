#!/usr/bin/env python
"""Headline benchmark: BASELINE.json configs[1] -- J2 isotropic-hardening (Voce) stress update + vjp
w.r.t. the material parameters, 10M synthetic Gauss points per GPU, fp64.

A "step" is one pass of the hot path over the resident batch: the fused kernel behind
`cm_update_and_vjp` (Newton return mapping -> xi, sigma; reverse IFT sweep -> grad[12] reduced on chip),
plus, for N > 1, one RCCL all-reduce of the 12 gradient doubles.  Points shard across ranks with no
data-path collective (weak scaling).  Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher in the environment (no WORLD_SIZE) the command starts its own ranks: the parent -- which never
imports torch or touches the GPU -- runs

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

as a child (`--dry-launch` prints that command), relays rank 0's JSON line and returns the child's exit code; started under a
launcher it is simply one of the ranks.  The line's `rccl` block states the process group's world size, backend and the number
of data-path collectives inside the timed region; `objective` is BASELINE.json configs[4] on the same shards (fused calibration
objective + gradient, one all-reduce of 13 doubles per evaluation), timed by the same rule right after the headline.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Gauss-point stress-updates/sec (fp64) + grad, 1/2/4/8 MI355X; % HBM roofline"
BYTES_PER_UPDATE = 280           # read gradu 72 + xi_prev 56 + sigma_bar 48; write xi 56 + sigma 48
HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


# GB/s of a kernel that only reads / writes the same SoA rows (profiles/r02_hbm_stream_ceilings.txt, 10^7 points)
LAYOUT_CEILING_GBS = {("j2_update_vjp", "full_3d"): 6331.0, ("j2_objective_grad", "full_3d"): 6919.0,
                      ("j2_update", "full_3d"): 6464.0}


def effective_cores():
    """Host cores this process can really use: the affinity mask, capped by the cgroup CPU quota (a GPU box exposes
    all of the host's hardware threads in the mask but grants the job only a share of them)."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2
            q, per = f.read().split()
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    return cores


def cpu_baseline(values, budget_s=6.0):
    """the CMAD path restated for the CPU, timed on this box's host cores on the same synthetic batch --
    update + vjp per point -- on ONE core and on ALL usable cores, median of 5 timed runs after a warm-up run, the sample
    sized from a probe so each leg costs about `budget_s` seconds.  Two restatements, both reported:
      * kind "port" (the `value`): oracle/cmad_port.cpp -- the SAME algorithm and arithmetic the HIP kernels run (host build of
        cmad_amd/csrc: hand-derived blocks, structured solve, J2 radial-line Newton, closed-form gradient), C++/OpenMP, g++ -O3;
      * "oracle_ad": oracle/cmad_oracle.cpp -- the parity checker: 7-dof Newton on the reference's residual with Jacobians by
        nested forward-mode AD, as jacfwd(residual) o grad(effective_stress) does in the reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    from cmad_amd.models.device import build_desc
    from cmad_amd.synthetic import gauss_point_batch
    cores = effective_cores()
    mat = ol.Material(values)
    st = ol.newton_settings()
    desc, _ = build_desc(values)

    def run_oracle(nthreads, g, xp, sb):
        t0 = time.perf_counter()
        xi, sig, it, cv = mat.update_batch(st, g, xp, nthreads=nthreads)
        mat.update_vjp_batch(g, xp, xi, sb, nthreads=nthreads, want_bars=False)
        return time.perf_counter() - t0

    def run_port(nthreads, g, xp, sb):
        t0 = time.perf_counter()
        ol.port_update_and_vjp(desc, g, xp, sb, nthreads=nthreads)
        return time.perf_counter() - t0

    def leg(run, nthreads, cap):
        probe = 2048 * nthreads
        g = gauss_point_batch(probe); xp = np.zeros((7, probe)); sb = np.random.default_rng(0).normal(size=(6, probe))
        run(nthreads, g, xp, sb)                                          # thread pool + page warm-up
        t = run(nthreads, g, xp, sb)
        B = int(max(probe, min(cap, probe * (budget_s / 6.0) / max(t, 1e-6))))          # 1 warm-up + 5 timed runs
        g = gauss_point_batch(B); xp = np.zeros((7, B)); sb = np.random.default_rng(0).normal(size=(6, B))
        run(nthreads, g, xp, sb)
        ts = sorted(run(nthreads, g, xp, sb) for _ in range(5))
        return B, ts[2], ts

    Bp, tp, tsp = leg(run_port, cores, 10_000_000)
    Bp1, tp1, _ = leg(run_port, 1, 10_000_000)
    Bo, to, tso = leg(run_oracle, cores, 1_000_000)
    Bo1, to1, _ = leg(run_oracle, 1, 1_000_000)
    return {"value": Bp / tp, "unit": "updates/s", "cores": cores, "kind": "port",
            "one_core": {"value": Bp1 / tp1, "unit": "updates/s", "cores": 1, "sample_points": Bp1},
            "hardware_threads_visible": os.cpu_count(),
            "sample": f"{Bp} points of the same synthetic batch (seed 22), update + vjp per point, median of 5 runs after "
                      f"warm-up ({tp:.2f} s, min {tsp[0]:.2f} / max {tsp[-1]:.2f}), OpenMP {cores} threads = the cores the job "
                      f"may use (affinity mask capped by the cgroup CPU quota; {os.cpu_count()} hardware threads visible); C++/OpenMP "
                      "port of the kernels' own algorithm (oracle/cmad_port.cpp: host build of cmad_amd/csrc), g++ -O3, not JAX",
            "oracle_ad": {"value": Bo / to, "unit": "updates/s", "cores": cores, "kind": "oracle (AD)",
                          "one_core": {"value": Bo1 / to1, "unit": "updates/s", "cores": 1, "sample_points": Bo1},
                          "sample": f"{Bo} points, median of 5 ({to:.2f} s, min {tso[0]:.2f} / max {tso[-1]:.2f}); the parity "
                                    "checker: 7-dof Newton, Jacobians by nested forward-mode AD as jacfwd o grad does, g++ -O3"}}


def load_traffic(points):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if int(t.get("points_per_launch", -1)) == int(points):
            return float(t["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def history_workload(args, dev, rank, world, distributed):
    """Side measurement for BASELINE.json configs[4] with a K-step history per point, the way the reference's
    calibration objectives are used (tests/objectives/test_calibrations.py: PLANE_STRESS, biaxial ramp):
    one evaluation = one cm_objective_grad_history launch (K updates forward with the states stored, K adjoint steps
    backward; --per-step-history: K cm_update launches + K cm_adjoint_step launches) + one all-reduce of (J, grad).  A "step" is one objective + gradient evaluation; value = point-steps per second."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from cmad_amd.models import DefType, SmallElasticPlastic
    from cmad_amd.objectives import BatchedCalibrationObjective
    from cmad_amd.parameters import Parameters
    from cmad_amd.synthetic import gauss_point_batch, j2_voce_values
    K = 10
    B = min(args.points, 2_000_000)
    from cmad_amd.parameters.parameters import tree_map
    values = j2_voce_values()
    flags = tree_map(lambda leaf: False, values)
    flags["plastic"]["flow stress"] = tree_map(lambda leaf: True, flags["plastic"]["flow stress"])     # Y, S, D active
    ps = args.workload == "ps_calibration_history" or args.def_type == "plane_stress"
    model = SmallElasticPlastic(Parameters(values, flags, tree_map(lambda leaf: None, values)),
                                DefType.PLANE_STRESS if ps else DefType.FULL_3D)
    g1 = torch.from_numpy(gauss_point_batch(B, seed=22 + rank, ndims=2 if ps else 3)).to(dev)
    ramp = torch.linspace(0.0, 1.5, K + 1, dtype=torch.float64, device=dev)
    gradu_hist = (ramp[:, None, None] * g1[None]).contiguous()                    # proportional ramp to 6 eps_y
    gen = torch.Generator(device=dev); gen.manual_seed(99 + rank)
    data_hist = 50.0 * torch.randn((K + 1, 6, B), dtype=torch.float64, device=dev, generator=gen)
    weight = np.zeros((3, 3)); weight[0, 0] = weight[1, 1] = 1.0
    fused = not args.per_step_history
    obj = BatchedCalibrationObjective(model, gradu_hist, data_hist, weight, fused_history=fused)
    for _ in range(max(1, args.warmup)):
        r = obj.evaluate_native()
    # a generation-2 collection of the interpreter (tens of ms with torch loaded) inside a window of a few 1-ms evaluations
    # would be timed as if it were the evaluation: collect before the window and keep the collector off inside it (as timeit does)
    import gc
    gc.collect()
    gc.disable()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r = obj.evaluate_native()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    assert np.isfinite(r.J) and np.isfinite(r.grad).all()
    if rank == 0:
        # one launch per evaluation (cm_objective_grad_history): forward read grad u 32, write xi 64; adjoint read
        # grad u 32 + previous xi 64 + data 48.  --per-step-history (one launch per step and direction): forward read
        # grad u 32 + xi_prev 64, write xi 64; adjoint read grad u 32 + xi_prev 64 + xi 64 + data 48 + history 64,
        # write history 64
        bytes_per = ((96 + 144) if fused else (160 + 336)) if ps else ((128 + 176) if fused else (184 + 344))
        value = world * B * K * args.steps / elapsed
        print(json.dumps({
            "metric": METRIC, "value": value, "unit": "point-steps/s (objective + gradient)", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"J2 {'PLANE_STRESS' if ps else 'FULL_3D'} calibration objective + gradient over a {K}-step history per point "
                                   "(side measurement for configs[4])", "points_per_gpu": B, "history_steps": K},
            "roofline": {"bound": "hbm", "achieved": bytes_per * B * K * args.steps / elapsed / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bytes_per * B * K * args.steps / elapsed / 1e9 / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_history" if fused else "k_update + k_reverse<MODE 2> per history step",
                         "kernel_ms": elapsed / args.steps * 1e3, "algorithmic_bytes_per_update": bytes_per}}))
    if distributed:
        dist.destroy_process_group()


def launcher_command(n_ranks, port, argv):
    """The command `bench.py --gpus N` runs itself under when it was not started by a launcher: one rank per GPU."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as FRESH child processes (this
    parent never imports torch and never touches the GPU -- a process that has initialised the GPU must not be replaced or
    forked), relay rank 0's JSON line(s) on stdout, everything else on stderr, and return the children's exit code."""
    import subprocess
    argv = [a for a in argv if a != "--dry-launch"]
    cmd = launcher_command(args.gpus, free_port(), argv)
    if args.dry_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this pool (task statement)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        is_json = False
        if line.startswith("{"):
            try:
                is_json = "metric" in json.loads(line)
            except ValueError:
                pass
        (sys.stdout if is_json else sys.stderr).write(line)
        (sys.stdout if is_json else sys.stderr).flush()
    return proc.wait()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-launch", action="store_true",
                    help="print the child command `--gpus N` would start (one rank per GPU under torch.distributed.run) and exit")
    ap.add_argument("--sustain", action="store_true",
                    help="sustained-clock measurement: 250 back-to-back steps after an idle gap, no warm-up; reports the mean of "
                         "launches 50-250 beside the first 14 and the per-launch trace (`sustained` in the JSON line)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000, help="Gauss points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch each step through the C-ABI from Python instead of replaying the captured HIP graph")
    ap.add_argument("--coherent", action="store_true",
                    help="side measurement: order the synthetic points by deviatoric strain magnitude, so that the lanes of a "
                         "wavefront sit in similar states (as neighbouring Gauss points of a mesh do) instead of the default "
                         "uncorrelated order, where every wavefront mixes elastic and plastic points")
    ap.add_argument("--per-step-history", action="store_true",
                    help="ps_calibration_history workload: one launch per step and direction instead of cm_objective_grad_history")
    ap.add_argument("--general-newton", action="store_true",
                    help="CM_SOLVER_GENERAL_NEWTON: force the general 7-dof Newton where the J2 radial-line "
                         "restriction (same iterates) would apply; side measurement")
    ap.add_argument("--reference-iterates", action="store_true",
                    help="CM_SOLVER_REFERENCE_ITERATES: no scalar return map / analytic warm start -- the Newton iteration starts at "
                         "x_prev and reproduces the reference's iterates and iteration counts (Hill FULL_3D, J2 PLANE_STRESS, "
                         "Hosford a >= 20 FULL_3D; side measurement)")
    ap.add_argument("--lockstep", action="store_true",
                    help="CM_SOLVER_LOCKSTEP: one point per lane for the whole kernel instead of the work-pool kernel (A/B)")
    ap.add_argument("--ls-evals", type=int, default=0,
                    help="J2 workloads: line-search evaluations per Newton iteration (0 = newton_solve defaults, "
                         "4 = make_newton_solve defaults)")
    ap.add_argument("--yield-surface", default="j2", choices=["j2", "hill", "hosford8", "barlat8"],
                    help="j2_* workloads with another yield surface (side measurements): Al7079 Hill coefficients, "
                         "Hosford a = 8, Yld2004-18p with the Al7079 coefficients and a = 8")
    ap.add_argument("--def-type", default="full_3d", choices=["full_3d", "plane_stress", "uniaxial_stress"],
                    help="J2 workloads: plane_stress is a side measurement (the reference's material-point tests' type)")
    ap.add_argument("--workload", default="j2_update_vjp",
                    choices=["j2_update_vjp", "j2_update", "j2_update_tangent", "j2_objective_grad", "hosford_update",
                             "hybrid_update", "hosford_update_vjp", "hybrid_update_vjp", "hosford_update_tangent", "hybrid_update_tangent",
                             "ps_calibration_history", "calibration_history"],
                    help="default = BASELINE.json configs[1]; the others are side measurements (DESIGN.md section 6)")
    args = ap.parse_args()
    # Not started by a launcher (no WORLD_SIZE): N > 1 -- or the 1-rank rehearsal of the process-group path -- runs under
    # torch.distributed.run as child processes, started before anything here imports torch or touches the GPU.
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_launch or os.environ.get("CMAD_BENCH_FORCE_DIST") == "1"):
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.sustain:
        args.steps, args.warmup = max(args.steps, 250), 0

    import numpy as np
    import torch
    import torch.distributed as dist
    from cmad_amd.models.device import DeviceEvaluator, NewtonSettings, build_desc
    from cmad_amd.synthetic import gauss_point_batch, j2_voce_values

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    # CMAD_BENCH_FORCE_DIST=1 exercises the process-group path with a single rank (rehearsal on a 1-GPU box)
    distributed = world > 1 or os.environ.get("CMAD_BENCH_FORCE_DIST") == "1"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # CMAD_BENCH_SHARED_GPU=1: rehearsal of the N-rank path on a box with ONE GPU -- every rank computes on device 0 and the
    # collectives run over gloo (RCCL refuses two ranks on one device).  Exercises the launcher, the sharding, the barriers, the
    # max-over-ranks timing and rank 0's line; its numbers mean nothing (the ranks share the card) and the line says so.
    shared_gpu = os.environ.get("CMAD_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")           # set by torch.distributed.run; only the 1-rank rehearsal lacks it
        # RCCL prints a version banner on stdout when its communicator comes up; stdout is reserved for the one JSON
        # line, so the communicator is created (init + one tiny all-reduce) with fd 1 pointed at stderr.
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if shared_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            probe = torch.zeros(1, dtype=torch.float64, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    B = args.points
    wl = args.workload
    if wl in ("ps_calibration_history", "calibration_history"):   # the second takes --def-type (full_3d / plane_stress)
        return history_workload(args, dev, rank, world, distributed)
    values = j2_voce_values()
    if args.yield_surface != "j2":
        assert wl.startswith("j2_"), "--yield-surface applies to the j2_* workloads"
        from cmad_amd.synthetic import AL7079_HILL
        al_barlat = [0.4555, 1.0274, 0.7101, 1.3755, 0.5314, 0.8817, 1.0558, 1.1133, 0.9220,
                     1.2431, 1.5438, 1.2204, 0.7632, 0.5327, 0.3015, 0.9722, 0.7399, 1.0760, 8.0]
        from cmad_amd.models.device import BARLAT_NAMES
        values["plastic"]["effective stress"] = {
            "hill": {"hill": dict(zip("FGHLMN", AL7079_HILL))},
            "hosford8": {"hosford": {"a": 8.0}},
            "barlat8": {"barlat": dict(zip(BARLAT_NAMES, al_barlat))}}[args.yield_surface]
    newton = NewtonSettings(j2_radial_line=not args.general_newton)   # newton_solve defaults: 10 iters, 1e-14, no line search
    if args.ls_evals > 0:                          # make_newton_solve: same tolerances + Armijo line search
        newton = NewtonSettings.traced(line_search_settings={"max evals": args.ls_evals})
    eps_y, hybrid = 1e-3, None
    if wl.startswith("hosford_update"):            # configs[2]: notch_hosford.yaml material + solver settings
        from cmad_amd.synthetic import hosford_values
        values, eps_y = hosford_values(), 2e-3
        newton = NewtonSettings.traced(max_iters=500, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 100})
        newton.j2_radial_line = not args.general_newton       # --general-newton: the reference's iteration from x_prev (work pool)
    elif wl.startswith("hybrid_update"):           # configs[3]: hybrid Hill + ICNN [6,16,1]
        from cmad_amd.models.device import HybridHillEffectiveStress
        from cmad_amd.synthetic import al7079_hybrid_setup
        icnn, values = al7079_hybrid_setup()
        hybrid, eps_y = HybridHillEffectiveStress(icnn), 525.0 / 70.2e3
        newton = NewtonSettings.traced(max_iters=50, abs_tol=1e-12, rel_tol=1e-12, line_search_settings={"max evals": 10})
    ps = args.def_type == "plane_stress"
    ux = args.def_type == "uniaxial_stress"
    if ps or ux:
        assert wl.startswith("j2_"), "--def-type applies to the J2 workloads"
    n_gradu, n_xi = (4, 8) if ps else ((1, 9) if ux else (9, 7))

    def algorithmic_bytes(w):
        # algorithmic bytes per point: rows of 8 bytes read (grad u, xi_prev, sigma_bar / data) and written (xi,
        # sigma, tangent); FULL_3D n_gradu = 9, n_xi = 7 -> 232 / 280 / 176 / 664; PLANE_STRESS n_gradu = 4, n_xi = 8 -> 208 / 256 / 144 / 400
        reads = n_gradu + n_xi + (6 if (w.endswith("_vjp") or w == "j2_objective_grad") else 0)
        writes = 0 if w == "j2_objective_grad" else n_xi + 6 + (6 * n_gradu if w.endswith("_update_tangent") else 0)
        return 8 * (reads + writes)

    bytes_per_update = algorithmic_bytes(wl)
    from cmad_amd.models.deformation_types import DefType
    newton.lockstep = bool(args.lockstep)
    newton.warm_start = not args.reference_iterates
    desc, info = build_desc(values, def_type=DefType.PLANE_STRESS if ps else (DefType.UNIAXIAL_STRESS if ux else DefType.FULL_3D),
                            newton=newton, hybrid=hybrid)
    ev = DeviceEvaluator(desc, info)
    nxi = n_xi

    # resident inputs (disjoint shard per rank: seed + rank)
    g_host = gauss_point_batch(B, seed=22 + rank, eps_y=eps_y, ndims=2 if ps else (1 if ux else 3))
    if args.coherent:
        nd = 2 if ps else 3
        e = 0.5 * (g_host.reshape(nd, nd, B) + g_host.reshape(nd, nd, B).transpose(1, 0, 2))
        mag = np.einsum("ijb,ijb->b", e, e) - np.einsum("iib->b", e) ** 2 / 3.0       # |dev eps|^2 (in-plane part for PLANE_STRESS)
        g_host = np.ascontiguousarray(g_host[:, np.argsort(mag, kind="stable")])
    gradu = torch.from_numpy(g_host).to(dev)
    xi_prev = torch.zeros((nxi, B), dtype=torch.float64, device=dev)
    if ps or ux:
        xi_prev[7:] = 1.0                          # F33 (lateral stretches) start at 1 (small_elastic_plastic.py:161-180)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    sigma_bar = torch.randn((6, B), dtype=torch.float64, device=dev, generator=gen)
    out = {"xi": torch.empty((nxi, B), dtype=torch.float64, device=dev),
           "sigma": torch.empty((6, B), dtype=torch.float64, device=dev),
           "grad": torch.empty(12, dtype=torch.float64, device=dev)}
    if wl.endswith("_update_tangent"):
        out["dsigma"] = torch.empty((6 * n_gradu, B), dtype=torch.float64, device=dev)
    wsq6 = [1., 1., 1., 1., 1., 1.]
    use_graph = not args.no_graph

    def timed_region(w, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` steps of workload `w` on the resident batch, bracketed by barrier +
        synchronize on both sides; returns the max-over-ranks wall time and the device-side timeline of this rank."""
        # two result buffers: the all-reduce of step k (RCCL's own stream) overlaps the kernel of step k+1
        grads = [torch.empty(12, dtype=torch.float64, device=dev) for _ in range(2)]
        res13 = [torch.empty(13, dtype=torch.float64, device=dev) for _ in range(2)]
        pending = [None, None]
        calls = [0]

        def launch(k):
            i = k & 1
            if pending[i] is not None:                 # the buffer's previous all-reduce must have finished
                pending[i].wait()
                pending[i] = None
            if w.endswith("_update_vjp"):
                out["grad"] = grads[i]
                ev.update_and_vjp(gradu, xi_prev, sigma_bar, out=out)
                return grads[i]
            if w == "j2_objective_grad":               # sigma_bar doubles as the "measured stress" array
                ev.objective_grad(gradu, xi_prev, sigma_bar, wsq6, out=res13[i])
                return res13[i]
            ev.update(gradu, xi_prev, want_status=False, out=out, tangent=w.endswith("_update_tangent"))
            return None

        def reduce_async(k, r):
            if distributed and r is not None:
                pending[k & 1] = dist.all_reduce(r, async_op=True)
                calls[0] += 1

        # The step is replayed from a captured HIP graph (one per result buffer): the entry points allocate and synchronise
        # nothing (include/cmad_hip.h), so a replay is the main kernel + the two reduction kernels with one host call and
        # no Python between the launches.  --no-graph launches through the C-ABI each step instead (A/B).
        graphs = [None, None]
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for i in range(2):
                    launch(i)                              # warm-up on the capture stream (workspace allocated here)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for i in range(2):
                graphs[i] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graphs[i], stream=side):
                    launch(i)
        result_of = res13 if w == "j2_objective_grad" else (grads if w.endswith("_update_vjp") else [None, None])   # what each graph writes

        def step(k):
            i = k & 1
            if not use_graph:
                return launch(k)
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None
            graphs[i].replay()
            return result_of[i]

        # every timing event is created AND recorded once before the timed region: hipEventCreate / the first record of an
        # event must not land between two timed launches
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        for e in starts + ends:
            e.record()
        for k in range(warmup):
            reduce_async(k, step(k))
        for i in range(2):
            if pending[i] is not None:
                pending[i].wait(); pending[i] = None
        if args.sustain:
            torch.cuda.synchronize()
            time.sleep(1.0)                            # start from an idle card: the first launches run at the boost clock
        calls[0] = 0
        host_t = [0.0] * (steps + 1)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            host_t[k] = time.perf_counter()
            starts[k].record()                         # events on the stream the kernels are launched on
            r = step(k)
            ends[k].record()
            reduce_async(k, r)
        host_t[steps] = time.perf_counter()
        for i in range(2):
            if pending[i] is not None:
                pending[i].wait()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        per_step_ms = [s_.elapsed_time(e_) for s_, e_ in zip(starts, ends)]
        kernel_ms = float(np.mean(per_step_ms))
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        kms = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
        per_rank_kernel_ms = [kernel_ms]
        if distributed:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            gathered = [torch.zeros_like(kms) for _ in range(world)]
            dist.all_gather(gathered, kms)
            per_rank_kernel_ms = [float(t.item()) for t in gathered]
        finite = all(bool(torch.isfinite(r_).all()) for r_ in result_of if r_ is not None)
        return {"elapsed": float(tmax.item()), "local_elapsed": elapsed, "per_step_ms": per_step_ms, "kernel_ms": kernel_ms,
                "device_span_ms": float(starts[0].elapsed_time(ends[-1])),           # first launch start -> last launch end, device clock
                "start_gaps_ms": [starts[k].elapsed_time(starts[k + 1]) for k in range(steps - 1)],
                "host_issue_ms": [(host_t[k + 1] - host_t[k]) * 1e3 for k in range(steps)],
                "per_rank_kernel_ms": per_rank_kernel_ms, "collective_calls": calls[0], "finite": finite,
                "payload_doubles": 13 if w == "j2_objective_grad" else (12 if w.endswith("_update_vjp") else 0)}

    tr = timed_region(wl, args.steps, args.warmup)
    elapsed, local_elapsed, per_step_ms, kernel_ms = tr["elapsed"], tr["local_elapsed"], tr["per_step_ms"], tr["kernel_ms"]
    device_span_ms, start_gaps_ms, host_issue_ms = tr["device_span_ms"], tr["start_gaps_ms"], tr["host_issue_ms"]
    per_rank_kernel_ms = tr["per_rank_kernel_ms"]
    headline_cfg = wl == "j2_update_vjp" and not (ps or ux) and args.yield_surface == "j2" and args.ls_evals == 0 and not args.general_newton
    # BASELINE.json configs[4] on the same resident shards: the fused calibration objective + gradient, one all-reduce of
    # (J, grad) = 13 doubles per evaluation over RCCL -- timed by the same rule (every rank takes part: it holds collectives)
    tobj = timed_region("j2_objective_grad", args.steps, args.warmup) if (headline_cfg and not args.sustain) else None
    # informative only, never `value`: the same step 120 more times -- a 20-step region after 5 warm-up steps sits inside the
    # clock transient of a back-to-back sequence (launches ~3-30, profiles/r04_sustained.txt); the later launches are what a
    # calibration's thousands of evaluations run at
    tlate = timed_region(wl, 120, 0) if (headline_cfg and not args.sustain) else None

    # a cheap self-check so a broken run cannot report a number: all points converged, finite gradient
    stride = max(1, B // 65536)                   # a strided sample: representative for any point order
    xi, sig, status = ev.update(gradu[:, ::stride][:, :65536].contiguous(), xi_prev[:, ::stride][:, :65536].contiguous())
    status = status.cpu().numpy().astype(np.uint32)
    assert ((status >> 16) & 1).mean() > 0.999, "points failed to converge"
    assert tr["finite"] and (tobj is None or tobj["finite"]), "non-finite objective / gradient"
    plastic_frac = float(((status & 0xFFFF) > 0).mean())

    if rank == 0:
        n = world
        value = n * B * args.steps / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        # roofline.achieved comes from the SAME interval as `value` (barrier-to-barrier wall clock, max over ranks), per GPU:
        # algorithmic bytes per launch / ms_per_step.  The kernel-only figure (HIP events around each step on the launch
        # stream: main kernel + the two reduction kernels) is kept beside it under roofline.kernel_only.
        achieved = bytes_per_update * B / (ms_per_step * 1e-3) / 1e9
        achieved_kernel = bytes_per_update * B / (kernel_ms * 1e-3) / 1e9
        res = {
            "metric": METRIC, "value": value, "unit": "updates/s", "n_gpus": n, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": {"j2_update_vjp": "J2 isotropic-hardening (Voce) stress update + vjp w.r.t. parameters, FULL_3D, "
                                              "synthetic Gauss points fp64 (BASELINE.json configs[1])",
                             "j2_update": "J2 + Voce stress update only, FULL_3D (side measurement)",
                             "j2_update_tangent": "J2 + Voce stress update + consistent tangent d sigma / d grad u (the FE "
                                                  "caller's per-integration-point call; side measurement)",
                             "j2_objective_grad": "fused J2 calibration objective + gradient, single step (configs[4] per GPU)",
                             "hosford_update": "Hosford a=100 stress update, notch_hosford.yaml material (configs[2])",
                             "hybrid_update": "hybrid Hill + ICNN[6,16,1] stress update (configs[3])",
                             "hosford_update_vjp": "Hosford a=100 stress update + vjp w.r.t. parameters (configs[2] material; side measurement)",
                             "hosford_update_tangent": "Hosford a=100 stress update + consistent tangent: the per-integration-point call of "
                                                       "the notch_hosford.yaml FE deck (configs[2] material and solver settings; side measurement)",
                             "hybrid_update_tangent": "hybrid Hill + ICNN[6,16,1] stress update + consistent tangent (configs[3] material; "
                                                      "side measurement)",
                             "hybrid_update_vjp": "hybrid Hill + ICNN[6,16,1] stress update + vjp w.r.t. parameters (configs[3] material; "
                                                  "side measurement)"}[wl]
                            .replace("FULL_3D", "PLANE_STRESS (side measurement)" if ps else ("UNIAXIAL_STRESS (side measurement)" if ux else "FULL_3D"))
                            .replace(" (BASELINE.json configs[1])", "" if (ps or ux or args.yield_surface != "j2") else " (BASELINE.json configs[1])"),
                "def_type": args.def_type, "yield_surface": args.yield_surface, "points_per_gpu": B, "plastic_fraction": round(plastic_frac, 4),
                "point_order": "sorted by deviatoric strain (coherent wavefronts)" if args.coherent else "uncorrelated",
                "newton": {"max_iters": newton.max_iters, "abs_tol": newton.abs_tol, "rel_tol": newton.rel_tol,
                           "line_search_max_evals": newton.line_search["max evals"],
                           "solver": ("make_newton_solve's Newton + Armijo search started at the analytic warm start (cm::hosford_warm_start); "
                                      "the returned state passes the reference's convergence test on the reference's residual"
                                      if (wl.startswith("hosford_") and not (args.general_newton or args.reference_iterates)) else
                                      "make_newton_solve's iteration from x_prev (work-pool kernel)"
                                      if wl.startswith("hosford_") else
                                      "plane-stress scalar return map, then the 8-dof Newton in the coordinates of the J2 plane started at "
                                      "its result (verified by the reference's convergence test; CM_SOLVER_REFERENCE_ITERATES starts at x_prev)"
                                      if (ps and args.yield_surface == "j2" and wl.startswith("j2_") and not (args.general_newton or args.reference_iterates)) else
                                      "Hill scalar return map, then the general 7-dof Newton started at its result (verified by the "
                                      "reference's convergence test; CM_SOLVER_REFERENCE_ITERATES starts at x_prev)"
                                      if (args.yield_surface == "hill" and not (ps or ux) and wl.startswith("j2_") and not (args.general_newton or args.reference_iterates)) else
                                      "8-dof Newton in the coordinates of the J2 plane it never leaves (identical iterates and "
                                      "iteration counts; CM_SOLVER_GENERAL_NEWTON turns it off)"
                                      if (ps and args.yield_surface == "j2" and wl.startswith("j2_") and not (args.general_newton or args.ls_evals > 0)) else
                                      "general 7-dof Newton, structured block solve"
                                      if (args.general_newton or args.ls_evals > 0 or ps or not wl.startswith("j2_")) else
                                      "7-dof Newton restricted to the J2 radial line it never leaves (identical iterates "
                                      "and iteration counts; CM_SOLVER_GENERAL_NEWTON turns it off)")},
                "parallelism": f"dp{n}: disjoint point shards; one RCCL all-reduce of the 12 fp64 gradient entries per "
                               "step, double-buffered so it overlaps the next step's kernel",
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(B) if (wl == "j2_update_vjp" and not (ps or ux) and args.yield_surface == "j2") else None,
                         # PMC counters cannot be read from inside this process: the figure is the committed result of the
                         # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/profile_gpu.sh)
                         "traffic_source": "profiles/traffic.json (rocprofv3 --pmc passes of this command, not measured in this run)",
                         "kernel": {"j2_update_vjp": "k_reverse<FULL_3D,J2,noROT,noLS,fused update+vjp,radial-line>"
                                    if not (args.general_newton or args.ls_evals > 0 or ps) else
                                    "k_reverse<J2,noROT,fused update+vjp>",
                                    "j2_objective_grad": "k_reverse<FULL_3D,J2,noROT,fused objective+grad>"}.get(wl, "k_update"),
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_update": bytes_per_update,
                         # what the same rows stream at with no arithmetic (tools/microbench/hbm_stream.hip)
                         "layout_streaming_ceiling": LAYOUT_CEILING_GBS.get((wl, args.def_type)),
                         "interval": "ms_per_step (same wall-clock interval as value)",
                         "kernel_only": {"ms": kernel_ms, "achieved": achieved_kernel, "frac": achieved_kernel / HBM_PEAK_GBS,
                                         "how": "mean HIP-event duration of one step on the launch stream"}},
            # where the time of the timed region went (rank 0): a host stall, a launch-bound loop and a slow kernel look
            # different here.  device_span = first launch start -> last launch end on the device clock;
            # start_gap = distance between consecutive launch starts; host_issue = host time spent issuing one step.
            "timeline": {"launch": "hip graph replay" if use_graph else "eager C-ABI calls",
                         "wall_ms": local_elapsed * 1e3, "device_span_ms": device_span_ms,
                         "device_span_per_step_ms": device_span_ms / args.steps,
                         "step_kernel_ms": {"min": min(per_step_ms), "max": max(per_step_ms), "mean": kernel_ms},
                         "start_gap_ms": ({"min": min(start_gaps_ms), "max": max(start_gaps_ms),
                                           "mean": float(np.mean(start_gaps_ms))} if start_gaps_ms else None),
                         "host_issue_ms": {"min": min(host_issue_ms), "max": max(host_issue_ms),
                                           "mean": float(np.mean(host_issue_ms))},
                         "per_rank_kernel_ms": per_rank_kernel_ms},
        }
        # proof of what ran: the process group's own world size and backend, and the number of data-path collectives inside
        # the timed region (one per step: the 12 gradient doubles of update + vjp, or (J, grad) = 13 of the objective)
        res["rccl"] = {"world_size": dist.get_world_size() if distributed else 1,
                       "backend": dist.get_backend() if distributed else None,
                       "collective_calls": tr["collective_calls"], "payload_doubles": tr["payload_doubles"],
                       "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else None}
        if shared_gpu:
            res["rccl"]["rehearsal"] = "CMAD_BENCH_SHARED_GPU=1: every rank on device 0, collectives over gloo -- not a measurement"
        if tobj is not None:
            ob, oms = algorithmic_bytes("j2_objective_grad"), tobj["elapsed"] / args.steps * 1e3
            res["objective"] = {
                "workload": "fused J2 calibration objective + gradient, single step, FULL_3D, points sharded over the ranks, one "
                            "RCCL all-reduce of (J, grad) = 13 fp64 per evaluation (BASELINE.json configs[4]: 8 x 1e7 = 8e7 points at N = 8)",
                "value": n * B * args.steps / tobj["elapsed"], "unit": "point objective+gradient evaluations/s",
                "total_points": n * B, "ms_per_step": oms, "steps": args.steps, "warmup": args.warmup,
                "roofline": {"bound": "hbm", "achieved": ob * B / (oms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ob * B / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_update": ob,
                             "kernel": "k_reverse<FULL_3D,J2,noROT,fused objective+grad>", "kernel_ms": tobj["kernel_ms"],
                             "kernel_only_frac": ob * B / (tobj["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "rccl": {"collective_calls": tobj["collective_calls"], "payload_doubles": tobj["payload_doubles"]},
                "per_rank_kernel_ms": tobj["per_rank_kernel_ms"]}
        if tlate is not None:
            late_ms = float(np.mean(tlate["per_step_ms"][60:]))
            res["later_launches"] = {"launches": 120, "mean_last_60_kernel_ms": late_ms,
                                     "value_at_that_rate": n * B / (late_ms * 1e-3),
                                     "frac": bytes_per_update * B / (late_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "note": "the same step repeated after the timed region; informative, not `value`"}
        if args.sustain:
            # a calibration runs thousands of evaluations back to back: after an idle gap the engine starts at its boost clock
            # and sustained fp64 issue pulls it down within milliseconds, so the sustained figure is the late one
            first, late = per_step_ms[:14], per_step_ms[50:250]
            to_frac = lambda ms: bytes_per_update * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            res["sustained"] = {"first_14_ms": float(np.mean(first)), "launches_50_250_ms": float(np.mean(late)),
                                "first_14_frac": to_frac(float(np.mean(first))), "sustained_frac": to_frac(float(np.mean(late))),
                                "sustained_value": B / (float(np.mean(late)) * 1e-3),
                                "per_launch_ms": [round(x, 4) for x in per_step_ms]}
        if n == 1 and headline_cfg and not args.sustain:
            # the same workload (i) on the general 7-dof Newton path (no J2 specialisation) and (ii) with
            # make_newton_solve's default line search (4 evaluations; for J2 every full step passes the Armijo test,
            # so the iterates are the same and the acceptance bookkeeping is the only extra work).
            # Reported beside the headline, never as `value`.
            def side(settings):
                d2, i2 = build_desc(values, newton=settings)
                ev2 = DeviceEvaluator(d2, i2)
                for _ in range(2):
                    ev2.update_and_vjp(gradu, xi_prev, sigma_bar, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = max(3, min(args.steps, 10))
                torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    ev2.update_and_vjp(gradu, xi_prev, sigma_bar, out=out)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                return {"kernel_ms": ms, "value": B / (ms * 1e-3), "unit": "updates/s",
                        "roofline_frac": bytes_per_update * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            res["general_newton"] = side(NewtonSettings(j2_radial_line=False))
            res["with_line_search"] = dict(side(NewtonSettings.traced()), max_evals=4)
        if n == 1 and not args.no_cpu_baseline and wl == "j2_update_vjp" and not args.sustain:
            res["cpu_baseline"] = cpu_baseline(values)
        print(json.dumps(res))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

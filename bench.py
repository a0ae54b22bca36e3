#!/usr/bin/env python3
"""bench.py — forward+backward steps/s of the MI355X DiffCloth stepper on BASELINE.json's headline workload.

Workload (SURVEY.md §8d, config C4 "10k-vertex cloth + contact + self-contact, batch = 256"): synthetic 100x100 grid
cloth (N = 10 000 vertices, T = 19 602 triangles, E = 29 205 bending flaps) of the reference's `sphereFabric`
(k_stretch 150, k_bend 1e-5, density 0.3, 4.5 x 4.5) dropped on the `rotatingSphereScene` sphere (r = 2, Signorini–Coulomb
contact), h = 1/180, with a flap — the last `--fold-rows` grid rows — folded back over the cloth and pressed onto it
(`--flap-force` x its own weight, a constant per-vertex force), so that every step carries `100 x fold-rows` LOADED self
contacts (detection, layering, layered friction and its transpose in the adjoint all run). 256 independent rollouts in total
(per-rollout start offset and friction coefficient, seed = global rollout id), sharded over the ranks.
One "step" = one forward time step (Simulation::step) + one backward step (Simulation::stepBackward) of all rollouts of the
job; `value` = rollout-steps per second over the whole job = B_total * K / t.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
One process per GPU; rollouts are independent, so ranks share nothing on the data path. Default: the metric's batch of 256
rollouts is SHARDED over the ranks ("strong" scaling: 256 / N per GPU; with fewer rollouts than CUs the engine spreads each
rollout over several workgroups, diffcloth_hip.h: dc_get_cluster). `--batch B` instead gives every rank B rollouts ("weak").
The only collectives are the optimiser-level all-reduce of the summed parameter gradient at the end of the backward sweep
(diffcloth_amd/distributed.py) and the barrier / MAX-reduce of the timing.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from diffcloth_amd import workloads  # noqa: E402  (the workload definitions live in the package; tests/ is only imported by cpu_baseline)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def scene(args):
    """Rest mesh, folded start shape, flap mask, sphere centre (fp32-representable, as the device sees them)."""
    return workloads.c4_scene(args.grid, args.fold_rows, args.fold_gap)      # reference grid builder, Simulation.cpp:2611-2757


def make_engine(device, args, V, F, center):
    from diffcloth_amd import capi
    e = capi.Engine(device)
    e.set_mesh(V, F)
    e.set_params(time_step=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, forward_tol=args.fwd_tol,
                 backward_tol=args.bwd_tol, cg_rel_tol=args.cg_tol, cg_max_iter=args.cg_max,
                 gradient_clipping=1, selfcollision_enabled=args.selfcollision, adjoint_mode=args.adjoint_mode,
                 adjoint_rel_tol=args.adjoint_rel_tol, adjoint_block_precond=args.block_precond)
    e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=center, radius=2.0, mu=0.9)])
    e.build()
    return e


def flap_force(args, mass, flap):
    """Constant per-vertex force (3N): the flap pressed onto the cloth with `flap_force` times its own weight."""
    return workloads.c4_flap_force(mass, flap, args.flap_force)


rollout_inputs = workloads.c4_rollout_inputs      # per-rollout start state and friction coefficient, seeded by the global rollout id


def cpu_baseline(args, V, F, center, field, x0, v0, mu, steps, gscale):
    """Reference algorithm (fp64 oracle port, OpenMP at the reference's sites) on the host cores: the SAME window of steps of
    rollout 0 as the GPU timed, forward and backward, direct adjoint solve (solveDirect semantics) like adjoint_mode 1."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))      # the oracle's ctypes wrapper (test infrastructure) — the checker, timed here as the CPU baseline
    import orc
    threads = args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 32)
    o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=args.fwd_tol,
                   bwd_tol=args.bwd_tol, selfcollision=bool(args.selfcollision), gradient_clipping=True, threads=threads)
    o.add_sphere(center, 2.0, float(mu))
    o.build()
    if field is not None:
        o.set_force_extras(None, field, 1.0)
    x = x0.copy(); v = v0.copy()
    t0 = time.perf_counter()
    recs = []
    for s in range(steps):
        out = o.step(x, v)
        recs.append(out)
        x, v = out["x"], out["v"]
    gx = gscale * (x - V.reshape(-1)); gv = np.zeros_like(gx)
    for s in reversed(range(steps)):
        b = o.step_backward(recs[s]["id"], gx, gv, is_start=False, direct=bool(args.adjoint_mode == 1))
        gx, gv = b["dL_dx"], b["dL_dv"]
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit="rollout-steps/s", cores=threads, kind="port",
                sample=f"rollout 0 of this job from its state after the {args.warmup} warm-up steps, the same {steps} fwd+bwd steps the GPU "
                       f"timed, fp64 oracle (oracle/, OpenMP at the reference's sites, sparse direct solves), mean PD iters "
                       f"{np.mean([r['iters'] for r in recs]):.0f}, self contacts {np.mean([r['nself'] for r in recs]):.0f}, "
                       + ("direct adjoint solve" if args.adjoint_mode == 1 else "reference adjoint iteration"))


def tshirt_evaluation0():
    """Secondary, like-for-like line: evaluation 0 of the reference's shipped L-BFGS run output/tshirt-exampleopt (wind_tshirt demo,
    1426 vertices, ONE rollout of 250 steps with self-collision, MATCH_TRAJECTORY loss, backward sweep) through this repository's
    diffcloth_py.OptimizeHelper — the only workload the reference publishes timings for (its own CPU run: forwardLog.txt:3-5 41.0 s
    forward, backwardLog.txt:3-9 2.67 s backward). Parameters of that evaluation from tests/golden/tshirt_golden.npz."""
    sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))
    try:
        import diffcloth_py as d
        g = np.load(os.path.join(workloads.GOLDEN, "tshirt_golden.npz"))
        V, F = workloads.load_mesh("tshirt")
        sim = d.makeSimFromMesh("wind_tshirt", V.reshape(-1), F.reshape(-1).tolist())
        h = d.makeOptimizeHelperWithSim("wind_tshirt", sim)
        x = np.array([*g["log_wind"][0], g["log_k"][0]])
        h.runSimulationAndGetLoss(x)                      # warm-up (allocations, first launches)
        t0 = time.perf_counter(); loss = h.runSimulationAndGetLoss(x); t_f = time.perf_counter() - t0
        t0 = time.perf_counter(); recs = h.runSimulationAndGetLossGradient(x); t_fb = time.perf_counter() - t0
        return {"workload": "wind_tshirt evaluation 0 of output/tshirt-exampleopt: 1 rollout, 250 steps fwd + 250 bwd, N=1426, self-collision on, "
                            "through diffcloth_py.OptimizeHelper (device-resident evaluation: wind factors and per-frame loss gradients as schedules, one launch per direction)",
                "forward_s": t_f, "forward_plus_backward_s": t_fb, "backward_s": max(t_fb - t_f, 0.0), "steps_per_s_fwd_bwd": 250.0 / t_fb,
                "loss": float(loss), "loss_logged_by_reference": float(g["losses"][0]), "pd_iterations": int(sim.getStateInfo().cumulateIter),
                "adjoint_iterations": int(recs[0].backwardTotalIters),
                "reference_cpu": {"forward_s": 41.0, "backward_s": 2.67, "steps_per_s_fwd_bwd": 250.0 / 43.67,
                                  "source": "/root/reference/output/tshirt-exampleopt/forwardLog.txt:3-5, backwardLog.txt:3-9 (authors' machine)"}}
    except Exception as ex:      # secondary information must never take the headline down
        return {"error": f"{type(ex).__name__}: {ex}"}


def secondary_config(device, key, K=10):
    """One of BASELINE.json's other configurations in a LOADED contact state (diffcloth_amd/workloads.py: hat pressed onto the head, sock
    pulled along the foot, squashed dress with its sheets in self contact, slope fabric sliding on its plane): lead-in steps run by the
    engine itself, then K timed fwd+bwd steps as two fused launches — contact counts and per-kernel times of the timed steps."""
    from diffcloth_amd import capi
    try:
        w = workloads.SECONDARY_WORKLOADS[key]()
        e = capi.Engine(device)
        e.set_mesh(w["P"], w["F"]); e.set_attachments(w["att"])
        e.set_params(forward_tol=w["fwd_tol"], backward_tol=5e-4, cg_rel_tol=1e-4, cg_max_iter=2000, gradient_clipping=1, adjoint_mode=1,
                     adjoint_rel_tol=1e-6, **w["params"])
        e.set_primitives(w["prims"]); e.build()
        B = w["B"]
        X0, V0, lead_xf, timed_xf, mus = w["start"](B, np.random.default_rng(0))
        L = len(lead_xf) if lead_xf is not None else 2
        e.alloc_batch(B, L + K)
        if mus is not None:
            e.set_mu(mus)
        e.set_state(0, X0, V0)
        if lead_xf is not None:
            e.set_fixed_point_schedule(0, np.concatenate([lead_xf, timed_xf(K)]))
        e.rollout_forward(0, L)
        e.seed_gradient(L, None, 1e-4); e.rollout_backward(L, 1); e.sync(); e.kernel_times(reset=True)
        t0 = time.perf_counter()
        e.rollout_forward(L, K); e.seed_gradient(L + K, None, 2.0 / ((K + 1) * e.N)); e.rollout_backward(L + K, K); e.sync()
        dt = time.perf_counter() - t0
        kt = e.kernel_times()
        fs = [e.get_stats(s) for s in range(L + 1, L + K + 1)]
        mean = lambda side, f: float(np.mean([st[side][f].mean() for st in fs]))      # noqa: E731
        per_rollout_adj = np.sum([st[1]["adjoint_iters"] for st in fs], axis=0) / K
        pd = mean(0, "pd_iters")
        out = {"workload": w["name"], "N": e.N, "rollouts": B, "steps": K, "lead_in_steps": L, "workgroups_per_rollout": e.cluster(),
               "rollout_steps_per_s": B * K / dt, "ms_per_batch_step": dt / K * 1e3, "fwd_ms_per_step": kt["fwd_ms"] / K, "bwd_ms_per_step": kt["bwd_ms"] / K,
               "mean_prim_contacts_per_step": mean(0, "prim_contacts"), "mean_self_contacts_per_step": mean(0, "self_contacts"),
               "mean_pd_iters_per_step": pd, "mean_cg_iters_per_pd_iter": mean(0, "cg_iters") / max(pd, 1e-30),
               "mean_adjoint_iters_per_step": mean(1, "adjoint_iters"), "mean_adjoint_cg_iters_per_step": mean(1, "cg_iters"),
               "slowest_rollout_adjoint_iters_per_step": float(per_rollout_adj.max()),
               "fp64_fallback_iters_per_step": mean(1, "fp64_iters"),
               "forward_converged_fraction": float(np.mean([(st[0]["converged"] > 0).mean() for st in fs])),
               "adjoint_converged_fraction": float(np.mean([(st[1]["converged"] != 0).mean() for st in fs])),
               "fwd_tol": w["fwd_tol"], "adjoint_rel_tol": 1e-6}
        dx, dv, _ = e.get_gradient()
        out["gradients_finite"] = bool(np.isfinite(dx).all() and np.isfinite(dv).all())
        e.close()
        return out
    except Exception as ex:      # secondary information must never take the headline down
        return {"workload": key, "error": f"{type(ex).__name__}: {ex}"}


def c4_share_config(device, args, V, F, V0, flap, center, B=32):
    """The headline workload at the per-GPU share of the 8-GPU job (256 rollouts sharded over 8 ranks = 32 per GPU): the engine splits every rollout over
    8 workgroups (dc_get_cluster) — the figure one GPU can measure of BASELINE.json's "batch 256 sharded over 8 GPUs". Same steps, same settings."""
    try:
        K, W = args.steps, args.warmup
        e = make_engine(device, args, V, F, center)
        e.alloc_batch(B, W + K)
        X0, MU = rollout_inputs(V0, np.arange(B))
        e.set_mu(MU)
        e.set_state(0, X0, np.zeros_like(X0))
        if flap.any() and args.flap_force != 0:
            e.set_vertex_forces(np.tile(flap_force(args, e.vertex_data()[0], flap), (B, 1)))
        if W > 0:
            e.rollout_forward(0, W)
        gscale = 2.0 / ((K + 1) * e.N)
        e.seed_gradient(W, None, gscale)
        if W > 0:
            e.rollout_backward(W, 1)
        e.sync(); e.kernel_times(reset=True)
        t0 = time.perf_counter()
        e.rollout_forward(W, K); e.seed_gradient(W + K, None, gscale); e.rollout_backward(W + K, K)
        dmu = e.get_mu_gradient().sum(axis=0)
        dt = time.perf_counter() - t0
        kt = e.kernel_times()
        st = [e.get_stats(s) for s in range(W + 1, W + K + 1)]
        pd = float(np.mean([q[0]["pd_iters"].mean() for q in st]))
        out = {"workload": f"C4 (the headline workload) at {B} rollouts: one rank's share of the 8-GPU job, every rollout split over workgroups",
               "rollouts": B, "steps": K, "workgroups_per_rollout": e.cluster(), "rollout_steps_per_s": B * K / dt, "ms_per_batch_step": dt / K * 1e3,
               "fwd_ms_per_step": kt["fwd_ms"] / K, "bwd_ms_per_step": kt["bwd_ms"] / K, "mean_pd_iters_per_step": pd,
               "mean_cg_iters_per_pd_iter": float(np.mean([q[0]["cg_iters"].mean() for q in st])) / max(pd, 1e-30),
               "mean_adjoint_operator_applications_per_step": float(np.mean([(2 * q[1]["adjoint_iters"] + q[1]["cg_iters"]).mean() for q in st])),
               "mean_self_contacts_per_step": float(np.mean([q[0]["self_contacts"].mean() for q in st])),
               "backward_workgroups_per_rollout": int(st[0][1]["workgroups"].max()), "dL_dmu_sum": float(np.asarray(dmu).sum()),
               "eight_gpu_projection_rollout_steps_per_s": 8.0 * B * K / dt,
               "projection_note": "8 x this figure is what 8 such GPUs deliver on the 256-rollout job if the ranks run as this one does (no data-path collective; unmeasured)"}
        e.close()
        return out
    except Exception as ex:      # secondary information must never take the headline down
        return {"workload": "C4 at 32 rollouts", "error": f"{type(ex).__name__}: {ex}"}


def config_key(args, B, K, W, N):
    return (f"N{N}_B{B}_K{K}_W{W}_fold{args.fold_rows}x{args.flap_force:g}_sc{args.selfcollision}_ft{args.fwd_tol:g}_cg{args.cg_tol:g}"
            f"_am{args.adjoint_mode}_ar{args.adjoint_rel_tol:g}_bp{args.block_precond}")


def load_profiles():
    """profiles/r*_roofline.json (tools/profile_round.sh -> tools/roofline_from_pmc.py), newest first: per kernel the calibrated
    HBM-side bytes of the timed sweep (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes), the LDS-array / VALU busy
    fractions (SQ counter passes) and the algorithmic bytes of the profiled run."""
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline.json")), reverse=True):
        with open(f) as fh:
            p = json.load(fh)
        p["file"] = os.path.relpath(f, ROOT)
        out.append(p)
    return out


def profile_for(profiles, key, kernel):
    """the profile of this very configuration if there is one, else the newest one that contains the kernel"""
    for p in profiles:
        if p.get("config_key") == key and kernel in p.get("kernels", {}):
            return p, True
    for p in profiles:
        if kernel in p.get("kernels", {}):
            return p, False
    return None, False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--total-batch", dest="total_batch", type=int, default=256, help="rollouts of the whole job, sharded over the ranks")
    ap.add_argument("--batch", type=int, default=0, help="> 0: rollouts PER GPU instead (weak scaling)")
    ap.add_argument("--grid", type=int, default=100, help="grid cloth resolution (grid x grid vertices)")
    ap.add_argument("--fold-rows", dest="fold_rows", type=int, default=5, help="grid rows of the folded flap (100 self contacts per row); 0 = flat cloth")
    ap.add_argument("--fold-gap", dest="fold_gap", type=float, default=0.02)
    ap.add_argument("--flap-force", dest="flap_force", type=float, default=2.0, help="flap pressed down with this multiple of its weight")
    ap.add_argument("--h", type=float, default=1.0 / 180)
    ap.add_argument("--fwd-tol", dest="fwd_tol", type=float, default=1e-8)   # hatController.py:83 / tshirtScene
    ap.add_argument("--bwd-tol", dest="bwd_tol", type=float, default=5e-4)   # every scene table
    ap.add_argument("--cg-tol", dest="cg_tol", type=float, default=1e-4)
    ap.add_argument("--cg-max", dest="cg_max", type=int, default=500)
    ap.add_argument("--adjoint-mode", dest="adjoint_mode", type=int, default=1,
                    help="1: direct adjoint solve (reference's solveDirect semantics); 0: reference fixed-point iteration")
    ap.add_argument("--adjoint-rel-tol", dest="adjoint_rel_tol", type=float, default=1e-6)
    ap.add_argument("--block-precond", dest="block_precond", type=int, default=0,
                    help="direct adjoint solve preconditioned with K's own 3x3 diagonal blocks (the engine's default, 1.3-1.8 x fewer iterations on "
                         "the garment scenes) or with diag(P) (0). On this soft fabric the blocks buy nothing (37.0 vs 37.4 BiCGSTAB iterations, "
                         "measured r02r) and cost 8 % per iteration, so the C4 workload runs with 0")
    ap.add_argument("--selfcollision", type=int, default=1,
                    help="self-collision detection + layered self friction (the reference's default: selfcollisionEnabled = true); 0 = off")
    ap.add_argument("--cluster", type=int, default=-1, help="workgroups per rollout: -1 = engine's choice, 1 = one workgroup per rollout")
    ap.add_argument("--cpu-steps", type=int, default=-1,
                    help="steps of the CPU baseline sample (rollout 0 from its state after the warm-up steps); -1 = the timed steps, 0 disables")
    ap.add_argument("--tshirt", type=int, default=1, help="also time evaluation 0 of the reference's T-shirt L-BFGS run (secondary line; 0 = skip)")
    ap.add_argument("--secondary", type=str, default="hat,sock,dress,perf_fabric",
                    help="other BASELINE.json configurations in loaded contact states, 10 fwd+bwd steps each (N = 1 only; 'none' = skip)")
    ap.add_argument("--cpu-threads", dest="cpu_threads", type=int, default=0,
                    help="OpenMP threads of the CPU baseline (0: min(host cores, 32), the fastest setting measured on the MI355X host)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # development switches for exercising the N > 1 control flow on a box with one GPU: all ranks on one device, gloo
    # instead of RCCL (RCCL refuses two ranks on one device). Never set by the driver.
    backend = os.environ.get("DC_BENCH_BACKEND", "nccl")
    if "DC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["DC_BENCH_DEVICE"])
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the stepper has no CPU path")

    from diffcloth_amd.distributed import allreduce_loss_and_grads, shard_rollouts
    V, F, V0, flap, center = scene(args)
    K, W = args.steps, args.warmup
    if args.batch > 0:          # weak scaling: B rollouts per rank
        B, scaling = args.batch, "weak"
        ids = np.arange(rank * B, (rank + 1) * B)
        total = world * B
    else:                       # the metric's batch sharded over the ranks
        total, scaling = args.total_batch, "strong"
        first, count = shard_rollouts(total, rank, world)
        ids = np.arange(first, first + count)
        B = len(ids)
    if args.cluster >= 0:
        os.environ["DC_CLUSTER"] = str(args.cluster)
    e = make_engine(local_rank, args, V, F, center)
    e.alloc_batch(B, W + K)
    X0, MU = rollout_inputs(V0, ids)
    e.set_mu(MU)
    e.set_state(0, X0, np.zeros_like(X0))
    field = None
    if flap.any() and args.flap_force != 0:
        field = flap_force(args, e.vertex_data()[0], flap)
        e.set_vertex_forces(np.tile(field, (B, 1)))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W untimed forward steps (contact onset) + one untimed backward step
    if W > 0:
        e.rollout_forward(0, W)
    # loss gradient of MATCH_TRAJECTORY (Simulation.cpp:3260-3274): dL/dx = 2 (x - target) / (frames * N)
    gscale = 2.0 / ((K + 1) * e.N)
    e.seed_gradient(W, None, gscale)
    if W > 0:
        e.rollout_backward(W, 1)
    e.sync()
    e.kernel_times(reset=True)

    barrier()
    t0 = time.perf_counter()
    e.rollout_forward(W, K)                 # K forward steps, tape on the device
    e.seed_gradient(W + K, None, gscale)    # dL/dx_K of the match-trajectory loss, on the device
    e.rollout_backward(W + K, K)            # K backward steps
    # optimiser-level reduction: the friction-coefficient gradient summed over all rollouts of the job — one
    # fused all-reduce (RCCL over xGMI at N > 1, nothing at N = 1), the only exchange between ranks
    dmu_local = e.get_mu_gradient().sum(axis=0)      # also waits for the sweep
    t_sweep = time.perf_counter() - t0
    _, (dmu_total,) = allreduce_loss_and_grads(0.0, [dmu_local])
    t_reduce = time.perf_counter() - t0 - t_sweep
    e.sync()
    barrier()
    dt = time.perf_counter() - t0
    rank_ms = [t_sweep * 1e3]
    if dist is not None:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tr = torch.zeros(world, device=dev, dtype=torch.float64)
        tr[rank] = t_sweep * 1e3
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        rank_ms = [float(v) for v in tr.tolist()]

    kt = e.kernel_times()
    # iteration statistics of the timed steps (needed for the byte / cycle models)
    pd = cg_f = adj = cg_b = selfc = primc = cyc = it64 = 0.0
    conv = 0
    cg_per_rollout = np.zeros(B); adj_per_rollout = np.zeros(B)       # load balance: a launch lasts as long as its slowest rollout
    for s in range(W + 1, W + K + 1):
        fs, bs = e.get_stats(s)
        pd += fs["pd_iters"].sum(); cg_f += fs["cg_iters"].sum(); conv += int((fs["converged"] > 0).sum())
        selfc += fs["self_contacts"].sum(); primc += fs["prim_contacts"].sum()
        adj += bs["adjoint_iters"].sum(); cg_b += bs["cg_iters"].sum(); cyc += bs["refine_cycles"].sum(); it64 += bs["fp64_iters"].sum()
        cg_per_rollout += fs["cg_iters"]; adj_per_rollout += bs["adjoint_iters"]
    N, T, E = e.N, e.T, e.E
    cl = e.cluster() if hasattr(e, "cluster") else 1
    # ---- roofline models (DESIGN.md "Roofline model"), per launch = the K timed steps of the rank's B rollouts ----
    # Forward kernel: the PCG vectors live in LDS / registers, so what the design HAS to move through HBM is the PD-level state only:
    #   compulsory bytes  108 * I_pd * N  (x_n, v, g, f, r, contact record per PD iteration) + 64 * N per step of tape
    # and the resource its inner loop loads is the LDS array. LDS-array cycles per CU (MI355X_MICROARCH.md LDS table: ds_read_b32 / b64
    # 2 cycles per wave-instruction, ds_write_b32 / b64 2 / 4):
    #   per CG iteration   rows/64 * (12 neighbours * (2 + 2) + 14 for reading / rewriting p and the LDS part of x)
    #   per PD iteration   staging 12 * 1.15 N/64 + triangles (6 * 4 + 12) * T/64 + flaps (8 * 4 + 6) * E/64 + vertex gather 20 * 4 * N/64
    #                      (a valence-6 vertex reads 8 signed triangle entries + 12 flap entries since round 3)
    # The streaming model of SURVEY.md section 8d ((108 I_pd + 132 I_cg) N: every CG vector through HBM) is kept as a number for reference;
    # it is what the resident design AVOIDS, not a roof for it.
    # Adjoint kernel: its Krylov vectors do stream through HBM — 72 N per step + 388 N per BiCGSTAB iteration (2 operator applications
    # x 108 B + 172 B of vector updates) + 96 N per fp64 residual evaluation + 100 N per step of fp64 gradient assembly.
    rows64 = (N + 63) // 64
    bytes_fwd = (108.0 * pd + 64.0 * B * K) * N
    bytes_fwd_stream = (108.0 * pd + 132.0 * cg_f) * N
    lds_cycles_fwd = (cg_f * rows64 * (12 * 4 + 14) + pd * (12 * 1.15 * rows64 + 36.0 * T / 64 + 38.0 * E / 64 + 80.0 * rows64)) / max(B * cl, 1)
    if args.adjoint_mode == 1:
        # (cg_b: iterations of the CG correction solves since round 6 — one operator application of 108 B + two vector passes of 72 + 36 B)
        bytes_bwd = (72.0 * B * K + 388.0 * adj + 216.0 * cg_b + 96.0 * cyc + 100.0 * B * K) * N
    else:
        bytes_bwd = (72.0 * B * K + 24.0 * adj + 132.0 * cg_b + 100.0 * B * K) * N
    lds_cycles_bwd = (2.0 * adj + (cg_b if args.adjoint_mode == 1 else 0.0) + cyc) * (12 * 1.15 * rows64 + 36.0 * T / 64 + 38.0 * E / 64 + 80.0 * rows64) / max(B * cl, 1)
    try:
        clock_hz = torch.cuda.get_device_properties(local_rank).clock_rate * 1e3
    except Exception:
        clock_hz = 2.4e9
    key = config_key(args, B, K, W, N)
    profiles = load_profiles()

    def kernel_entry(name, nbytes, lds_cycles, ms, launches, extra):
        """Contract fields: achieved = algorithmic bytes of the launch / its duration (HIP events on the context's stream,
        dc_kernel_times), frac = achieved / HBM peak. Next to them the LDS side (modelled LDS-array cycles per CU / kernel cycles)
        and, when profiles/ holds a rocprofv3 --pmc profile, the MEASURED quantities: fabric traffic (calibrated FETCH_SIZE +
        WRITE_SIZE; of this very configuration, else that profile's traffic per algorithmic byte applied to this run and said so),
        LDS-array / VALU busy fractions, share of wave cycles spent waiting. Every fraction is <= 1 by construction."""
        sec = max(ms * 1e-3, 1e-12)
        ent = {"kernel": name, "avg_launch_ms": ms / max(launches, 1), "launches": launches, "steps_per_launch": K / max(launches, 1),
               "ms_per_step": ms / K, "algorithmic_bytes": nbytes, "achieved": nbytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": nbytes / sec / 1e9 / HBM_PEAK_GBS,
               "lds_model_cycles_per_cu": lds_cycles, "lds_frac_model": lds_cycles / (sec * clock_hz), "clock_hz": clock_hz,
               "traffic": None, "traffic_frac": None, "traffic_source": None, "lds_frac": None, "lds_conflict_share": None,
               "valu_frac": None, "wait_frac": None, "scratch_write_bytes": None,
               "scratch_write_bytes_note": "measured WRITE_SIZE minus the modelled stores (compulsory_write_bytes): an upper bound on spill traffic, not a measurement of it — "
                                           "profiles/r06_scratch_static.txt locates the scratch stores of the code objects (bench adjoint: none inside a loop; bench forward: 2)"}
        ent.update(extra)
        prof, exact = profile_for(profiles, key, name)
        if prof:
            pk = prof["kernels"][name]
            if exact and abs(pk["launch_ms_total"] - ms) > 0.1 * ms:       # same configuration, other kernels: lends ratios only
                exact = False
            pa = max(pk.get("algorithmic_bytes", 0.0), 1.0)
            if exact:
                traffic, src = pk["hbm_bytes"], f"{prof['file']} (this configuration)"
            else:
                traffic = nbytes * pk["hbm_bytes"] / pa
                src = f"{prof['file']} (profiled at {prof.get('config_key')}: its traffic per algorithmic byte applied to this run)"
            ent.update(traffic=traffic / max(launches, 1), traffic_frac=min(traffic / sec / 1e9 / HBM_PEAK_GBS, 1.0), traffic_source=src,
                       lds_frac=pk.get("lds_frac"), lds_conflict_share=pk.get("lds_conflict_share"), valu_frac=pk.get("valu_frac"),
                       wait_frac=pk.get("wait_frac"))
            if pk.get("scratch_write_bytes") is not None:
                ent["scratch_write_bytes"] = pk["scratch_write_bytes"] * (1.0 if exact else nbytes / pa) / max(launches, 1)
        # the binding resource: the largest of the fractions known for this kernel (measured ones where there are any)
        cand = {"hbm": ent["traffic_frac"] if ent["traffic_frac"] is not None else ent["frac"],
                "lds": ent["lds_frac"] if ent["lds_frac"] is not None else ent["lds_frac_model"]}
        if ent["valu_frac"] is not None:
            cand["valu"] = ent["valu_frac"]
        ent["bound"] = max(cand, key=cand.get)
        ent["bound_frac"] = cand[ent["bound"]]
        return ent
    # The roof the resident forward design is actually up against: instruction issue. Floor of the PCG products alone — 6 instructions per
    # non-zero (bit-field extract, address add, ds_read_b64, three v_fma_mix_f32), each a wave-instruction of 4 cycles on its SIMD
    # (VERDICT r04's accounting; MI355X_MICROARCH.md quotes 2 cycles for a plain v_fma_f32, which halves the figure) — per CU:
    #   cycles = products of the CU's rollouts x 6 x nnz x (1 / 64 lanes) x (1 / 4 SIMDs) x 4
    nnz_p = float(getattr(e, "nnz", 12.8 * N))
    issue_floor_cycles = (cg_f / max(B * cl, 1)) * 6.0 * nnz_p / 64.0 / 4.0 * 4.0
    k_fwd = kernel_entry("k_pd_step_cl" if cl > 1 else "k_pd_step_pk", bytes_fwd, lds_cycles_fwd, kt["fwd_ms"], kt["fwd_launches"],
                         {"issue_floor_cycles_per_cu": issue_floor_cycles,
                          "issue_frac": issue_floor_cycles / max(kt["fwd_ms"] * 1e-3 * clock_hz, 1e-30),
                          "issue_model": "VALU / LDS issue floor of the PCG products alone: 6 instructions per non-zero x 4 cycles per wave-instruction and SIMD, over the launch time",
                          "streaming_model_bytes": bytes_fwd_stream,
                          # SURVEY.md section 8d's figure taken as a rate, side by side with `frac`: NOT a roof for this design — the CG vectors
                          # it counts never pass through HBM (they live in LDS / registers), so the value exceeds 1
                          "frac_streaming_model": bytes_fwd_stream / max(kt["fwd_ms"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
                          "frac_streaming_model_note": "SURVEY 8d bytes (108 I_pd + 132 I_cg) N over the launch time and 8 TB/s; not a roof: the resident PCG moves none of the 132 I_cg N through HBM",
                          "compulsory_write_bytes": (60.0 * pd + 64.0 * B * K) * N,
                          "model": "compulsory HBM bytes of the resident design: (108 I_pd + 64 per step) N; CG vectors never leave the CU"})
    k_bwd = kernel_entry("k_adjoint_step_cl" if cl > 1 else "k_adjoint_step", bytes_bwd, lds_cycles_bwd, kt["bwd_ms"], kt["bwd_launches"],
                         {"streaming_model_bytes": bytes_bwd, "compulsory_write_bytes": (72.0 * adj + (48.0 * cg_b if args.adjoint_mode == 1 else 0.0) + 48.0 * cyc + 60.0 * B * K) * N,
                          "model": "Krylov vectors stream through HBM: (72 + 100) N per step + 388 N per BiCGSTAB iteration + 216 N per CG iteration + 96 N per fp64 residual"})
    dom = dict(k_fwd if kt["fwd_ms"] >= kt["bwd_ms"] else k_bwd)
    bound = dom["bound"]
    dx, dv, dmu = e.get_gradient()
    finite = bool(np.isfinite(dx).all() and np.isfinite(dv).all())

    if rank == 0:
        out = {
            "metric": "fwd+bwd steps/sec (node), 10k-vtx cloth+contact, batch=256",
            "value": total * K / dt,
            "unit": "rollout-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "dtype_note": "state, tape and solver vectors fp32; the forward PCG's search direction is stored as power-of-two-scaled fp16 at 20 rows per thread "
                          "(the N = 10 000 instance: iterates, residuals and all sums stay fp32, step length = exact line search along the stored direction); "
                          "fp64 for element strains, the adjoint's residual / fall-back and all gradient sums",
            "config": {"workload": f"C4 grid {args.grid}x{args.grid} cloth (N={N}, T={e.T}, E={e.E}) on sphere r=2, "
                                   f"h=1/{round(1 / args.h)}, primitive Signorini-Coulomb contact"
                                   + (" + self-collision" if args.selfcollision else "")
                                   + (f", flap of {args.fold_rows} rows folded back and pressed down ({args.flap_force:g} x weight)" if flap.any() else ""),
                       "config_key": key, "rollouts_per_gpu": B, "rollouts_total": total, "workgroups_per_rollout": cl,
                       "fwd_tol": args.fwd_tol, "bwd_tol": args.bwd_tol, "cg_rel_tol": args.cg_tol, "adjoint_mode": args.adjoint_mode,
                       "adjoint_rel_tol": args.adjoint_rel_tol, "adjoint_block_precond": args.block_precond, "selfcollision": bool(args.selfcollision),
                       "mean_self_contacts_per_step": selfc / (B * K), "mean_prim_contacts_per_step": primc / (B * K),
                       "mean_pd_iters_per_step": pd / (B * K), "mean_cg_iters_per_pd_iter": cg_f / max(pd, 1),
                       "mean_adjoint_iters_per_step": adj / (B * K), "mean_adjoint_cg_iters_per_step": (cg_b / (B * K)) if args.adjoint_mode == 1 else 0.0,
                       "mean_adjoint_operator_applications_per_step": (2.0 * adj + (cg_b if args.adjoint_mode == 1 else 0.0)) / (B * K),
                       "mean_fp32_solves_per_adjoint": cyc / (B * K),
                       "fp64_fallback_iters": it64, "adjoint_precision": "mixed: fp32 CG (BiCGSTAB once a CG cycle fails to contract) corrections of the fp64 residual",
                       "converged_fraction": conv / (B * K),
                       "slowest_rollout_over_mean": {"forward_pcg_iterations": float(cg_per_rollout.max() / max(cg_per_rollout.mean(), 1e-30)),
                                                     "adjoint_iterations": float(adj_per_rollout.max() / max(adj_per_rollout.mean(), 1e-30)),
                                                     # what starting a rollout's backward sweep as soon as ITS forward sweep ends could save: per-rollout
                                                     # sweep times modelled as the launch times scaled by the rollout's share of the iterations
                                                     "two_launches_over_per_rollout_handover": float(
                                                         (kt["fwd_ms"] + kt["bwd_ms"]) / max((kt["fwd_ms"] * cg_per_rollout / max(cg_per_rollout.max(), 1e-30)
                                                                                             + kt["bwd_ms"] * adj_per_rollout / max(adj_per_rollout.max(), 1e-30)).max(), 1e-30)),
                                                     "correlation_forward_backward_work": float(np.corrcoef(cg_per_rollout, adj_per_rollout)[0, 1]) if B > 1 else 0.0},
                       "batch_steps_per_s": K / dt, "gradients_finite": finite,
                       "dL_dmu_sum_over_job": float(np.asarray(dmu_total).sum()),
                       "per_rank_sweep_ms": rank_ms, "allreduce_ms": t_reduce * 1e3,
                       "parallelism": f"rollout-sharded x{world}"},
            # dominant kernel first (contract fields), then both kernels
            "roofline": {"bound": bound, **dom, "kernels": [k_fwd, k_bwd]},
        }
        ncpu = K if args.cpu_steps < 0 else args.cpu_steps
        if ncpu > 0:      # rank 0, at every N: the same rollout 0 of the job
            xw, vw = e.get_state(W)
            out["cpu_baseline"] = cpu_baseline(args, V, F, center, field, xw[0], vw[0], MU[0, 0], ncpu, gscale)
        if world == 1 and args.tshirt:
            out["secondary"] = tshirt_evaluation0()
        if world == 1 and args.secondary and args.secondary.lower() not in ("none", "0", "off"):
            e.close()          # the headline's 40 GB of tape go back before the other configurations allocate theirs
            out["secondary_configs"] = [secondary_config(local_rank, k.strip()) for k in args.secondary.split(",") if k.strip()]
            if args.cluster < 0 and args.total_batch == 256 and args.batch == 0:
                out["secondary_configs"].append(c4_share_config(local_rank, args, V, F, V0, flap, center))
        line = json.dumps(out)
    else:
        line = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # The JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio at communicator set-up, which sits in
    # the C buffer (stdout is a pipe) until someone flushes it — flush it out first, then print the line.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()

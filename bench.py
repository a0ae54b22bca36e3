#!/usr/bin/env python3
"""bench.py — forward+backward steps/s of the MI355X DiffCloth stepper on BASELINE.json's headline workload.

Workload: synthetic 100x100 grid cloth (N = 10 000 vertices, T = 19 602 triangles,
E = 29 205 bending flaps) of the reference's `sphereFabric` (k_stretch 150, k_bend 1e-5, density 0.3, 4.5 x 4.5)
dropped on the `rotatingSphereScene` sphere (r = 2, Signorini–Coulomb contact, self-collision on), h = 1/180, 256 independent
rollouts per GPU (per-rollout start offset and friction coefficient, seed = global rollout id).
One "step" = one forward time step (Simulation::step) + one backward step (Simulation::stepBackward) of all
rollouts of the job; `value` = rollout-steps per second over the whole job = B_total * K / t.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
One process per GPU; rollouts are independent, so ranks share nothing on the data path ("weak" scaling: 256
rollouts per GPU). The only collectives are the optimiser-level all-reduce of the summed parameter gradient at the end of
the backward sweep (diffcloth_amd/distributed.py) and the barrier / MAX-reduce of the timing.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def grid_cloth(nx, dim):
    """Reference grid builder (Simulation.cpp:2611-2757), orientation DOWN — same numbering as tests/meshes.py."""
    import meshes
    return meshes.grid_cloth(nx, nx, dim, dim, "DOWN")


def make_engine(device, args, V, F, center):
    from diffcloth_amd import capi
    e = capi.Engine(device)
    e.set_mesh(V, F)
    e.set_params(time_step=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, forward_tol=args.fwd_tol,
                 backward_tol=args.bwd_tol, cg_rel_tol=args.cg_tol, cg_max_iter=args.cg_max,
                 gradient_clipping=1, selfcollision_enabled=args.selfcollision, adjoint_mode=args.adjoint_mode,
                 adjoint_rel_tol=args.adjoint_rel_tol)
    e.set_primitives([dict(kind=capi.DC_PRIM_SPHERE, group=0, center=center, radius=2.0, mu=0.9)])
    e.build()
    return e


def rollout_inputs(V, ids):
    """Per-rollout start state and friction coefficient, seeded by the global rollout id."""
    X = np.empty((len(ids), V.size)); MU = np.empty((len(ids), 1))
    for k, gid in enumerate(ids):
        rng = np.random.default_rng(1000 + int(gid))
        shift = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.09, -0.02), rng.uniform(-0.5, 0.5)])
        X[k] = (V + shift).astype(np.float32).reshape(-1)
        MU[k, 0] = rng.uniform(0.1, 0.9)
    return X, MU


def cpu_baseline(args, V, F, center, x0, v0, mu, steps, gscale):
    """Reference algorithm (fp64 oracle port, OpenMP at the reference's sites) on the host cores, one rollout."""
    import orc
    threads = args.cpu_threads if args.cpu_threads > 0 else min(os.cpu_count() or 1, 32)
    o = orc.Oracle(V, F, h=args.h, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=args.fwd_tol,
                   bwd_tol=args.bwd_tol, selfcollision=bool(args.selfcollision), gradient_clipping=True, threads=threads)
    o.add_sphere(center, 2.0, float(mu))
    o.build()
    x = x0.copy(); v = v0.copy()
    t0 = time.perf_counter()
    recs = []
    for s in range(steps):
        out = o.step(x, v)
        recs.append(out)
        x, v = out["x"], out["v"]
    gx = gscale * (x - V.reshape(-1)); gv = np.zeros_like(gx)
    for s in reversed(range(steps)):
        b = o.step_backward(recs[s]["id"], gx, gv, is_start=False, direct=False)
        gx, gv = b["dL_dx"], b["dL_dv"]
    dt = time.perf_counter() - t0
    return dict(value=steps / dt, unit="rollout-steps/s", cores=threads, kind="port",
                sample=f"rollout 0 of this job from its state after the {args.warmup} warm-up steps, {steps} fwd+bwd steps, fp64 "
                       f"oracle (oracle/, OpenMP at the reference's sites), mean PD iters {np.mean([r['iters'] for r in recs]):.0f}, "
                       f"reference adjoint iteration")


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r01g_traffic.json")


def measured_traffic(args, K, W, B):
    """{kernel: HBM-side bytes per launch} measured by rocprofv3 --pmc for the default workload (profiles/), {} otherwise."""
    default = (K == 10 and W == 5 and B == 256 and args.grid == 100 and args.selfcollision == 1 and args.fwd_tol == 1e-8
               and args.cg_tol == 1e-4 and args.adjoint_mode == 1 and args.adjoint_rel_tol == 1e-6)
    if not default or not os.path.exists(TRAFFIC_FILE):
        return {}
    with open(TRAFFIC_FILE) as f:
        return {k: v["hbm_bytes_per_launch"] for k, v in json.load(f)["kernels"].items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="rollouts per GPU")
    ap.add_argument("--grid", type=int, default=100, help="grid cloth resolution (grid x grid vertices)")
    ap.add_argument("--h", type=float, default=1.0 / 180)
    ap.add_argument("--fwd-tol", dest="fwd_tol", type=float, default=1e-8)   # hatController.py:83 / tshirtScene
    ap.add_argument("--bwd-tol", dest="bwd_tol", type=float, default=5e-4)   # every scene table
    ap.add_argument("--cg-tol", dest="cg_tol", type=float, default=1e-4)
    ap.add_argument("--cg-max", dest="cg_max", type=int, default=500)
    ap.add_argument("--adjoint-mode", dest="adjoint_mode", type=int, default=1,
                    help="1: direct adjoint solve (reference's solveDirect semantics); 0: reference fixed-point iteration")
    ap.add_argument("--adjoint-rel-tol", dest="adjoint_rel_tol", type=float, default=1e-6)
    ap.add_argument("--selfcollision", type=int, default=1,
                    help="self-collision detection + layered self friction (the reference's default: selfcollisionEnabled = true); 0 = off")
    ap.add_argument("--cpu-steps", type=int, default=10,
                    help="steps of the CPU baseline sample: rollout 0 from its state after the warm-up steps (0 disables); the default, "
                         "the 10 timed steps of that rollout, is ~0.5 s of wall time = 10-20 s of CPU work on 32 threads")
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

    from diffcloth_amd.distributed import allreduce_loss_and_grads
    V, F = grid_cloth(args.grid, 4.5)
    V = V.astype(np.float32).astype(np.float64)
    import meshes
    center = meshes.sphere_scene_center(V, 2.0).astype(np.float32).astype(np.float64)
    B, K, W = args.batch, args.steps, args.warmup
    e = make_engine(local_rank, args, V, F, center)
    e.alloc_batch(B, W + K)
    ids = np.arange(rank * B, (rank + 1) * B)
    X0, MU = rollout_inputs(V, ids)
    e.set_mu(MU)
    e.set_state(0, X0, np.zeros_like(X0))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W untimed forward steps (contact onset) + one untimed backward step
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
    _, (dmu_total,) = allreduce_loss_and_grads(0.0, [dmu_local])
    e.sync()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kt = e.kernel_times()
    # iteration statistics of the timed steps (needed for the algorithmic-byte count)
    pd = cg_f = adj = cg_b = selfc = 0.0
    conv = 0
    for s in range(W + 1, W + K + 1):
        fs, bs = e.get_stats(s)
        pd += fs["pd_iters"].sum(); cg_f += fs["cg_iters"].sum(); conv += int((fs["converged"] > 0).sum())
        selfc += fs["self_contacts"].sum()
        adj += bs["adjoint_iters"].sum(); cg_b += bs["cg_iters"].sum()
    N = e.N
    # Algorithmic bytes (fp32, per rollout; DESIGN.md "Roofline model"): what ONE streaming pass per vector sweep
    # would move if nothing stayed on chip.
    #   forward    (108 * I_pd + 132 * I_cg) * N
    #   backward, mode 0 (reference iteration)  72 N + (24 * I_adj + 132 * I_cg) * N
    #   backward, mode 1 (BiCGSTAB on K)        72 N + 388 * I_adj * N   (2 operator applications x 108 B + 172 B of vector updates)
    bytes_fwd = (108.0 * pd + 132.0 * cg_f) * N
    if args.adjoint_mode == 1:
        bytes_bwd = (72.0 * B * K + 388.0 * adj) * N
    else:
        bytes_bwd = (72.0 * B * K + 24.0 * adj + 132.0 * cg_b) * N

    def kernel_entry(name, nbytes, ms, launches):
        # one launch = the K timed steps of all B rollouts (dc_rollout_* runs a rollout's steps inside one launch)
        gbs = nbytes / max(ms * 1e-3, 1e-12) / 1e9
        return {"kernel": name, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": nbytes / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                "steps_per_launch": K / max(launches, 1), "ms_per_step": ms / K}
    k_fwd = kernel_entry("k_pd_step_pk", bytes_fwd, kt["fwd_ms"], kt["fwd_launches"])
    k_bwd = kernel_entry("k_adjoint_step", bytes_bwd, kt["bwd_ms"], kt["bwd_launches"])
    # HBM-side bytes per launch from the PMC passes of the round profile (tools/profile_round.sh -> profiles/*_traffic.json:
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command, FETCH_SIZE doubled as the
    # calibration kernels show for this part) — only quoted when this run is the profiled workload, else null
    traffic = measured_traffic(args, K, W, B) if world == 1 else {}
    k_fwd["traffic"] = traffic.get("k_pd_step_pk"); k_bwd["traffic"] = traffic.get("k_adjoint_step")
    dom = k_fwd if kt["fwd_ms"] >= kt["bwd_ms"] else k_bwd
    dx, dv, dmu = e.get_gradient()
    finite = bool(np.isfinite(dx).all() and np.isfinite(dv).all())

    if rank == 0:
        out = {
            "metric": "fwd+bwd steps/sec (node), 10k-vtx cloth+contact, batch=256",
            "value": world * B * K / dt,
            "unit": "rollout-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C4 grid {args.grid}x{args.grid} cloth (N={N}, T={e.T}, E={e.E}) on sphere r=2, "
                                   f"h=1/{round(1 / args.h)}, primitive Signorini-Coulomb contact" + (" + self-collision" if args.selfcollision else ""),
                       "rollouts_per_gpu": B, "rollouts_total": world * B, "fwd_tol": args.fwd_tol,
                       "bwd_tol": args.bwd_tol, "cg_rel_tol": args.cg_tol, "adjoint_mode": args.adjoint_mode,
                       "adjoint_rel_tol": args.adjoint_rel_tol, "selfcollision": bool(args.selfcollision), "mean_self_contacts_per_step": selfc / (B * K),
                       "mean_pd_iters_per_step": pd / (B * K), "mean_cg_iters_per_pd_iter": cg_f / max(pd, 1),
                       "mean_adjoint_iters_per_step": adj / (B * K), "converged_fraction": conv / (B * K),
                       "batch_steps_per_s": world * K / dt, "gradients_finite": finite,
                       "dL_dmu_sum_over_job": float(np.asarray(dmu_total).sum()),
                       "parallelism": f"rollout-sharded x{world}"},
            # dominant kernel first (contract fields), then both kernels. `achieved` prices the ALGORITHMIC bytes; the
            # forward kernel keeps the PCG vectors in LDS/registers, so its figure can exceed the HBM peak — `traffic` is
            # what it really moves through the fabric (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, profiles/).
            "roofline": {"bound": "hbm", **dom, "kernels": [k_fwd, k_bwd]},
        }
        if world == 1 and args.cpu_steps > 0:
            xw, vw = e.get_state(W)
            out["cpu_baseline"] = cpu_baseline(args, V, F, center, xw[0], vw[0], MU[0, 0], args.cpu_steps, gscale)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

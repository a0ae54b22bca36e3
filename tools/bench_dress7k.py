"""Developer script: a garment-sized self-contact workload — the reference's 7 742-vertex dress (src/assets/meshes/remeshed/
dress-v7k-f14k.obj, frozen in tests/golden/meshes.npz) hung by its top rim and twirled like the dress_twirl demo
(TRAJECTORY_DRESS_TWIRL, Simulation.cpp:1005-1016: the attached rim rotates 0.02 rad per step about the vertical axis through the
rest shape's mid point), self-collision on. The rim targets of every step are a device schedule (dc_set_fixed_point_schedule), so
forward and backward are one launch each. Not the headline metric (bench.py is); numbers go into DESIGN.md §6.
  python tools/bench_dress7k.py [rollouts] [timed steps] [warm-up steps]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from diffcloth_amd import capi


def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    V, F = scenes.load_mesh("dress7k")
    P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0)
    P = f32(P)
    att = np.where(P[:, 1] >= np.quantile(P[:, 1], 0.995))[0]          # the top rim (39 vertices)
    e = capi.Engine(0)
    e.set_mesh(P, F); e.set_attachments(att.tolist())
    e.set_params(time_step=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05, forward_tol=1e-8, backward_tol=5e-4, cg_rel_tol=1e-4,
                 cg_max_iter=2000, gradient_clipping=1, selfcollision_enabled=1, adjoint_mode=1, adjoint_rel_tol=1e-6)
    e.set_primitives([])
    e.build()
    S = W + K
    e.alloc_batch(B, S + 1)
    rng = np.random.default_rng(0)
    # (1e-5: the fine regions of this garment hold ~380 non-connected vertex pairs inside the collision radii at rest; a perturbation of 1e-3 pushes
    #  some of them through each other and the step of such a rollout explodes — in the fp64 oracle as well, it is the reference's algorithm)
    X = np.stack([f32(P.reshape(-1) + 1e-5 * rng.standard_normal(P.size)) for _ in range(B)])
    e.set_state(0, X, np.zeros_like(X))
    # twirl: rim targets of step s = rest rim rotated by 0.02 (s + 1) rad about the vertical axis through the bounding-box mid point
    mid = 0.5 * (rmin + rmax)
    XF = np.zeros((S, B, 3 * len(att)))
    for s in range(S):
        a = 0.02 * (s + 1) * (1.0 + 0.0)
        c, sn = np.cos(a), np.sin(a)
        q = P[att].copy()
        rel = q - mid
        q[:, 0] = mid[0] + c * rel[:, 0] + sn * rel[:, 2]
        q[:, 2] = mid[2] - sn * rel[:, 0] + c * rel[:, 2]
        XF[s] = f32(q.reshape(-1))[None, :]
    e.set_fixed_point_schedule(0, XF)
    e.rollout_forward(0, W)
    e.seed_gradient(W, None, 1e-4); e.rollout_backward(W, 1); e.sync(); e.kernel_times(reset=True)
    t0 = time.perf_counter()
    e.rollout_forward(W, K); e.seed_gradient(S, None, 2.0 / ((K + 1) * e.N)); e.rollout_backward(S, K); e.sync()
    dt = time.perf_counter() - t0
    kt = e.kernel_times()
    st = [e.get_stats(s) for s in range(W + 1, S + 1)]
    pd = np.mean([a["pd_iters"].mean() for a, _ in st]); cg = np.mean([a["cg_iters"].mean() for a, _ in st])
    sc = np.mean([a["self_contacts"].mean() for a, _ in st]); adj = np.mean([b["adjoint_iters"].mean() for _, b in st])
    conv = np.mean([(a["converged"] != 0).mean() for a, _ in st])
    xs, vs = e.get_state(S)
    print(f"largest |v| at the last step {np.abs(vs).max():.2f}, PD iterations per rollout at the last step {st[-1][0]['pd_iters']}")
    f64 = np.mean([b["fp64_iters"].mean() for _, b in st]); f64n = np.mean([(b["fp64_iters"] > 0).mean() for _, b in st])
    cyc = np.mean([b["refine_cycles"].mean() for _, b in st]); bconv = np.mean([(b["converged"] != 0).mean() for _, b in st])
    print(f"adjoint: fp32 BiCGSTAB {adj:.0f} iterations in {cyc:.1f} solves, fp64 fall-back in {f64n:.2f} of the solves ({f64:.0f} iterations on average), converged {bconv:.2f}; "
          f"deflation {e.deflation()}")
    sc0 = e.get_self_contacts(S, 0, cap=16000)
    dx, dv, _ = e.get_gradient()
    print(f"dress7k twirl: N={e.N} T={e.T} rim {len(att)} vertices, B={B} x {e.cluster()} workgroups, steps {W}+{K}: {B * K / dt:.0f} rollout-steps/s, "
          f"{dt / K * 1e3:.2f} ms per batch step (fwd {kt['fwd_ms'] / K:.2f} ms, bwd {kt['bwd_ms'] / K:.2f} ms), PD iters {pd:.0f} (PCG {cg / max(pd, 1):.1f} each), "
          f"BiCGSTAB {adj:.0f}, self contacts {sc:.0f} per step in {sc0['layers']} layers (rollout 0, last step), converged {conv:.2f}, gradients finite {bool(np.isfinite(dx).all() and np.isfinite(dv).all())}")


if __name__ == "__main__":
    main()

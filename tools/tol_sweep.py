"""Developer script: effect of the inner-solver tolerances (PCG relative tolerance of the PD global step, relative
residual of the direct adjoint solve) on time, iteration counts and results of the bench.py workload, against a run with
tight inner tolerances. The outer criteria (forward x_diff threshold = the reference's forwardConvergenceThreshold) stay."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, meshes

class A: pass

def run(cg_tol, adj_tol, B=256, W=5, K=10, sc=1, grid=100):
    a = A(); a.h = 1 / 180; a.fwd_tol = 1e-8; a.bwd_tol = 5e-4; a.cg_tol = cg_tol; a.cg_max = 500; a.selfcollision = sc
    a.adjoint_mode = 1; a.adjoint_rel_tol = adj_tol
    V, F = bench.grid_cloth(grid, 4.5); V = V.astype(np.float32).astype(np.float64)
    c = meshes.sphere_scene_center(V, 2.0).astype(np.float32).astype(np.float64)
    e = bench.make_engine(0, a, V, F, c)
    e.alloc_batch(B, W + K)
    X0, MU = bench.rollout_inputs(V, np.arange(B))
    e.set_mu(MU); e.set_state(0, X0, np.zeros_like(X0))
    e.rollout_forward(0, W)
    g = 2.0 / ((K + 1) * e.N)
    e.seed_gradient(W, None, g); e.rollout_backward(W, 1); e.sync(); e.kernel_times(reset=True)
    t0 = time.perf_counter()
    e.rollout_forward(W, K); e.seed_gradient(W + K, None, g); e.rollout_backward(W + K, K); e.sync()
    dt = time.perf_counter() - t0
    kt = e.kernel_times()
    pd = np.mean([e.get_stats(s)[0]["pd_iters"].mean() for s in range(W + 1, W + K + 1)])
    cg = np.mean([e.get_stats(s)[0]["cg_iters"].mean() for s in range(W + 1, W + K + 1)])
    adj = np.mean([e.get_stats(s)[1]["adjoint_iters"].mean() for s in range(W + 1, W + K + 1)])
    conv = np.mean([(e.get_stats(s)[0]["converged"] > 0).mean() for s in range(W + 1, W + K + 1)])
    x, v = e.get_state(W + K)
    dx, dv, dmu = e.get_gradient()
    return dict(rate=B * K / dt, fwd=kt["fwd_ms"] / K, bwd=kt["bwd_ms"] / K, pd=pd, cg=cg / pd, adj=adj, conv=conv, x=x, dx=dx, dv=dv)

if __name__ == "__main__":
    ref = run(1e-7, 1e-9)
    print(f"tight: cg 1e-7 adj 1e-9: {ref['rate']:.0f} r-steps/s fwd {ref['fwd']:.2f} bwd {ref['bwd']:.2f} ms, PD {ref['pd']:.1f} x CG {ref['cg']:.1f}, adj {ref['adj']:.1f}")
    for cg_tol, adj_tol in [(1e-4, 1e-6), (1e-3, 1e-6), (1e-2, 1e-6), (3e-2, 1e-6), (1e-1, 1e-6), (1e-4, 1e-5), (1e-4, 1e-4), (1e-2, 1e-5), (1e-2, 1e-4)]:
        r = run(cg_tol, adj_tol)
        ex = np.abs(r["x"] - ref["x"]).max(axis=1)
        eg = np.linalg.norm(r["dx"] - ref["dx"], axis=1) / np.linalg.norm(ref["dx"], axis=1)
        print(f"cg {cg_tol:g} adj {adj_tol:g}: {r['rate']:.0f} r-steps/s fwd {r['fwd']:.2f} bwd {r['bwd']:.2f} ms, PD {r['pd']:.1f} x CG {r['cg']:.1f}, adj {r['adj']:.1f}, "
              f"conv {r['conv']:.3f} | max|x-x_tight| median {np.median(ex):.1e} max {ex.max():.1e} | dL/dx0 rel err median {np.median(eg):.1e} max {eg.max():.1e}")

"""Developer script: per-rollout iteration statistics of the bench workload (load imbalance across the 256 workgroups)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, meshes
import argparse
args = argparse.Namespace(h=1 / 180, fwd_tol=1e-8, bwd_tol=5e-4, cg_tol=1e-4, cg_max=500, adjoint_mode=1, adjoint_rel_tol=1e-6)
V, F = bench.grid_cloth(100, 4.5); V = V.astype(np.float32).astype(np.float64)
center = meshes.sphere_scene_center(V, 2.0).astype(np.float32).astype(np.float64)
W, K, B = int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 10, 256
e = bench.make_engine(0, args, V, F, center)
e.alloc_batch(B, W + K)
X0, MU = bench.rollout_inputs(V, np.arange(B))
e.set_mu(MU); e.set_state(0, X0, np.zeros_like(X0))
e.rollout_forward(0, W + K)
e.seed_gradient(W + K, None, 2.0 / ((K + 1) * e.N))
e.rollout_backward(W + K, K)
e.sync()
for s in range(W + 1, W + K + 1):
    fs, bs = e.get_stats(s)
    cg = fs["cg_iters"]
    print(f"step {s}: pd iters min {fs['pd_iters'].min()} mean {fs['pd_iters'].mean():.1f} max {fs['pd_iters'].max()} | cg total per rollout min {cg.min()} mean {cg.mean():.0f} max {cg.max()} "
          f"(max/mean {cg.max() / cg.mean():.2f}) | adjoint iters min {bs['adjoint_iters'].min()} mean {bs['adjoint_iters'].mean():.1f} max {bs['adjoint_iters'].max()}")

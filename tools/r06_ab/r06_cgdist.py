"""Developer diagnostic (round 6): distribution over (step, rollout) of the adjoint's CG / BiCGSTAB iterations on the bench workload."""
import os, sys, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
args = types.SimpleNamespace(grid=100, fold_rows=5, fold_gap=0.02, flap_force=2.0, h=1.0 / 180, fwd_tol=1e-8, bwd_tol=5e-4, cg_tol=1e-4, cg_max=500,
                             adjoint_mode=1, adjoint_rel_tol=1e-6, block_precond=0, selfcollision=1, warmup=5, cpu_threads=0)
B, W, K = 256, 5, 10
V, F, V0, flap, center = bench.scene(args)
e = bench.make_engine(0, args, V, F, center)
e.alloc_batch(B, W + K)
X0, MU = bench.rollout_inputs(V0, np.arange(B))
e.set_mu(MU); e.set_state(0, X0, np.zeros_like(X0))
e.set_vertex_forces(np.tile(bench.flap_force(args, e.vertex_data()[0], flap), (B, 1)))
e.rollout_forward(0, W + K)
e.seed_gradient(W + K, None, 2.0 / ((K + 1) * e.N))
e.kernel_times(reset=True)
e.rollout_backward(W + K, K); e.sync()
kt = e.kernel_times()
cg = np.array([e.get_stats(s)[1]["cg_iters"] for s in range(W + 1, W + K + 1)])
bi = np.array([e.get_stats(s)[1]["adjoint_iters"] for s in range(W + 1, W + K + 1)])
cyc = np.array([e.get_stats(s)[1]["refine_cycles"] for s in range(W + 1, W + K + 1)])
apps = cg + 2 * bi
print(f"bwd {kt['bwd_ms'] / K:.2f} ms per batch step; applications per step: mean {apps.mean():.1f} median {np.median(apps):.0f} p90 {np.percentile(apps, 90):.0f} p99 {np.percentile(apps, 99):.0f} max {apps.max()}")
print(f"CG iterations: mean {cg.mean():.1f} max {cg.max()}; BiCGSTAB iterations: mean {bi.mean():.1f} max {bi.max()}; cycles mean {cyc.mean():.2f} max {cyc.max()}")
per_rollout = apps.sum(axis=0)
print(f"per rollout over {K} steps: mean {per_rollout.mean():.0f} max {per_rollout.max()} (slowest / mean {per_rollout.max() / per_rollout.mean():.2f})")
print("histogram of applications per step:", np.histogram(apps, bins=[0, 40, 60, 80, 100, 150, 200, 400, 1000, 10000])[0])

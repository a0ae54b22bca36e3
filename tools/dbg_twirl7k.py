"""Developer script: the twirl workload of tools/bench_dress7k.py with per-step adjoint statistics (B rollouts, S steps)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from diffcloth_amd import capi
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 8
V, F = scenes.load_mesh("dress7k")
P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0); P = f32(P)
att = np.where(P[:, 1] >= np.quantile(P[:, 1], 0.995))[0]
e = capi.Engine(0)
e.set_mesh(P, F); e.set_attachments(att.tolist())
e.set_params(time_step=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05, forward_tol=1e-8, backward_tol=5e-4, cg_rel_tol=float(os.environ.get("CGTOL", "1e-4")),
             cg_max_iter=2000, gradient_clipping=int(os.environ.get("CLIP", "1")), selfcollision_enabled=1, adjoint_mode=1, adjoint_rel_tol=1e-6)
e.set_primitives([]); e.build()
e.alloc_batch(B, S + 1)
rng = np.random.default_rng(0)
X = np.stack([f32(P.reshape(-1) + 0.001 * rng.standard_normal(P.size)) for _ in range(B)])
e.set_state(0, X, np.zeros_like(X))
mid = 0.5 * (rmin + rmax)
XF = np.zeros((S, B, 3 * len(att)))
for s in range(S):
    a = 0.02 * (s + 1); c, sn = np.cos(a), np.sin(a)
    q = P[att].copy(); rel = q - mid
    q[:, 0] = mid[0] + c * rel[:, 0] + sn * rel[:, 2]; q[:, 2] = mid[2] - sn * rel[:, 0] + c * rel[:, 2]
    XF[s] = f32(q.reshape(-1))[None, :]
e.set_fixed_point_schedule(0, XF)
e.rollout_forward(0, S); e.sync()
t0 = time.perf_counter()
e.seed_gradient(S, None, 2.0 / (4 * e.N)); e.rollout_backward(S, min(S, 3)); e.sync()
dt = time.perf_counter() - t0
print(f"[{os.environ.get('TAG', '')}] deflation {e.deflation()}, {e.cluster()} workgroups, backward of 3 steps {dt:.2f} s")
for s in range(S, S - 3, -1):
    a, b = e.get_stats(s)
    print(f"  step {s}: PD {a['pd_iters']}, self {a['self_contacts']}, fwd conv {a['converged']}; adjoint conv {b['converged']}, fp32 {b['adjoint_iters']} in {b['refine_cycles']}, fp64 {b['fp64_iters']}, residual {b['last_udiff']}")

"""Developer script: the squashed dress-7742 step of tests/test_gpu_configs.py, forward + adjoint statistics only (no oracle)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from diffcloth_amd import capi
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
V, F = scenes.load_mesh("dress7k")
P, rmin, rmax = scenes.normalise_model(V, "FRONT", 8.0); P = f32(P)
top = np.argsort(-P[:, 1])[:6].tolist()
e = capi.Engine(0)
e.set_mesh(P, F); e.set_attachments(top)
e.set_params(time_step=1.0 / 120, density=0.2, k_stretch=800.0, k_bend=0.05, forward_tol=1e-8, backward_tol=1e-9, cg_rel_tol=1e-6, cg_max_iter=3000,
             gradient_clipping=0, selfcollision_enabled=1, adjoint_mode=1, adjoint_rel_tol=1e-7)
e.set_primitives([]); e.build()
rng = np.random.default_rng(8)
X = P.copy(); X[:, 2] *= 0.9
vel = np.zeros_like(X); vel[:, 2] = -0.1 * np.sign(P[:, 2])
x0 = f32((X + 0.0005 * rng.standard_normal(X.shape)).reshape(-1))[None, :]
v0 = f32((vel + 0.005 * rng.standard_normal(X.shape)).reshape(-1))[None, :]
xf = f32(X[top].reshape(-1))[None, :]
e.alloc_batch(1, 1); e.set_state(0, x0, v0)
st = e.step_forward(0, fixed_pts=xf)
gx = f32(rng.standard_normal(x0.shape)); gv = f32(0.01 * rng.standard_normal(x0.shape))
t0 = time.perf_counter()
gb = e.step_backward(1, gx, gv, is_start=False)
dt = time.perf_counter() - t0
print(f"[{os.environ.get('TAG', '')}] deflation {e.deflation()}, {e.cluster()} workgroup(s): PD {st['pd_iters'][0]}, PCG/PD {st['cg_iters'][0] / st['pd_iters'][0]:.0f}; adjoint converged {gb['converged'][0]}, "
      f"fp32 {gb['adjoint_iters'][0]} iterations in {gb['refine_cycles'][0]} solve(s), fp64 {gb['fp64_iters'][0]}, residual {gb['last_udiff'][0]:.2e}, |dL_dx| {np.linalg.norm(gb['dL_dx'][0]):.6e}, {dt:.2f} s")

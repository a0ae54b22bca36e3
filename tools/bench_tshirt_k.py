"""Developer script: single-rollout T-shirt (N = 1426) forward/backward time per step against the number of workgroups per rollout."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from diffcloth_amd import capi
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
cfg = scenes.TSHIRT
V, F = scenes.load_mesh("tshirt")
P, rmin, rmax = scenes.normalise_model(V, "BACK", 6.0); P = f32(P)
att = scenes.corner_attachments(P, rmin, rmax)
for K in (1, 2, 3, 4, 6, 8):
    os.environ["DC_CLUSTER"] = str(K)
    e = capi.Engine(0); e.set_mesh(P, F); e.set_attachments(att)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], forward_tol=1e-8, backward_tol=5e-4,
                 cg_rel_tol=1e-4, cg_max_iter=2000, gradient_clipping=1, selfcollision_enabled=1, adjoint_mode=1, adjoint_rel_tol=1e-6)
    e.build(); e.alloc_batch(1, 22)
    x = f32(P.reshape(-1)); e.set_state(0, x[None], np.zeros((1, x.size)))
    wind = np.array([1.0, 0.1, 1.0]); wind = wind / np.linalg.norm(wind) * 0.015
    e.set_uniform_force(wind[None])
    e.rollout_forward(0, 2); e.seed_gradient(2, None, 1e-4); e.rollout_backward(2, 1); e.sync(); e.kernel_times(reset=True)
    e.rollout_forward(2, 20); e.seed_gradient(22, None, 1e-4); e.rollout_backward(22, 20); e.sync()
    kt = e.kernel_times()
    pd = np.mean([e.get_stats(s)[0]["pd_iters"].mean() for s in range(3, 23)]); cg = np.mean([e.get_stats(s)[0]["cg_iters"].mean() for s in range(3, 23)])
    adj = np.mean([e.get_stats(s)[1]["adjoint_iters"].mean() for s in range(3, 23)])
    print(f"K requested {K} -> {e.cluster()}: fwd {kt['fwd_ms'] / 20:.2f} ms/step, bwd {kt['bwd_ms'] / 20:.2f} ms/step; PD {pd:.0f} x CG {cg / pd:.1f}, BiCGSTAB {adj:.0f}; "
          f"{kt['fwd_ms'] / 20 / cg * 1e3:.2f} us per CG iteration", flush=True)

import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, orc, scenes
from diffcloth_amd import capi
def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)
V, F = scenes.load_mesh("hat"); cfg = scenes.HAT
P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
P = f32(P)
for (kb, att, sph) in [(120.0, cfg["attachments"], True), (120.0, [], False), (0.0, cfg["attachments"], False), (1.0, [], False)]:
    o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=kb, fwd_tol=1e-9, bwd_tol=1e-9, attachments=att, selfcollision=False, gradient_clipping=False)
    c = f32(scenes.hat_head_center(rmin, rmax, 2.1))
    if sph: o.add_sphere(c, 2.1, 0.1)
    o.build()
    e = capi.Engine(0); e.set_mesh(P, F); e.set_attachments(att)
    e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=kb, forward_tol=1e-9, backward_tol=1e-9, cg_rel_tol=1e-6, cg_max_iter=3000, gradient_clipping=0)
    if sph: e.set_primitives([dict(kind=0, group=0, center=c, radius=2.1, mu=0.1)])
    e.build(); e.alloc_batch(1, 1)
    x = P.reshape(-1).copy(); v = np.zeros_like(x); a = P[att].reshape(-1) if att else None
    for s in range(3):
        if att: a = f32(a + np.tile([0.02, -0.05, 0.01], len(att)))
        e.set_state(0, x, v); st = e.step_forward(0, fixed_pts=a); x1, v1 = e.get_state(1)
        ref = o.step(x, v, a)
        print(f"kb={kb} att={len(att)} sph={sph} step {s}: err {np.abs(x1[0]-ref['x']).max():.2e} gpu pd {st['pd_iters'][0]} conv {st['converged'][0]} cg {st['cg_iters'][0]} xdiff {st['last_xdiff'][0]:.1e} | ref pd {ref['iters']} conv {ref['converged']}")
        x, v = f32(ref["x"]), f32(ref["v"])

import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/diffcloth_amd/lib")
import numpy as np, orc, scenes
import diffcloth_py as d
def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)
V, F = scenes.load_mesh("hat"); cfg = scenes.HAT
sim = d.makeSimFromMesh("wear_hat", V.reshape(-1), F.reshape(-1).tolist())
P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"])
o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=cfg["k_stretch"], k_bend=cfg["k_bend"], fwd_tol=1e-8, bwd_tol=1e-9, attachments=cfg["attachments"], selfcollision=False, gradient_clipping=False)
o.add_sphere(scenes.hat_head_center(rmin, rmax, 2.1), 2.1, 0.1); o.build()
d.Simulation.forwardConvergenceThreshold = 1e-8
sim.resetSystem()
rec = sim.getStateInfo(); x, v = f32(rec.x), f32(rec.v); a = f32(rec.x_fixedpoints)
print("rest diff", np.abs(rec.x - P.reshape(-1)).max(), "a", a)
for s in range(4):
    a = f32(a + np.array([0.02, -0.05, 0.01, 0.02, -0.05, 0.01]))
    sim.stepNN(s + 1, x, v, a); new = sim.getStateInfo(); ref = o.step(x, v, a)
    print(s, "err", np.abs(new.x - ref["x"]).max(), "conv", new.converged, new.convergeIter, "ref", ref["iters"], ref["converged"], "xf", np.abs(new.x_fixedpoints - a).max(), "ncontact", len(new.collisionInfos[0][0]), ref["nprim"])
    x, v = f32(ref["x"]), f32(ref["v"])

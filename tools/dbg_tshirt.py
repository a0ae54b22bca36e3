import sys, os
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, orc, scenes
from diffcloth_amd import capi
def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)
g = np.load(os.path.join(scenes.GOLDEN, "tshirt_golden.npz"))
V, F = scenes.load_mesh("tshirt"); cfg = scenes.TSHIRT
P, rmin, rmax = scenes.normalise_model(V, cfg["orientation"], cfg["cloth_dim"]); P = f32(P)
att = scenes.corner_attachments(P, rmin, rmax)
o = orc.Oracle(P, F, h=cfg["h"], density=cfg["density"], k_stretch=float(g["k_stretch"]), k_bend=cfg["k_bend"], fwd_tol=1e-8, bwd_tol=5e-4, attachments=att, contact=True, selfcollision=True, threads=4)
fw = g["f_wind"]; o.set_wind(True, 2, fw[0:3]/np.linalg.norm(fw[0:3]), float(np.linalg.norm(fw[0:3])), float(fw[3]), float(fw[4])); o.build()
e = capi.Engine(0); e.set_mesh(P, F); e.set_attachments(att)
e.set_params(time_step=cfg["h"], density=cfg["density"], k_stretch=float(g["k_stretch"]), k_bend=cfg["k_bend"], forward_tol=1e-8, cg_rel_tol=1e-5, cg_max_iter=2000, selfcollision_enabled=1, stall_window=int(os.environ.get("SW", "40")))
e.build(); e.alloc_batch(1, 1)
x = P.reshape(-1).copy(); v = np.zeros_like(x); xf = P[att].reshape(-1)
for k in range(1, 41):
    t = k*cfg["h"]; factor = (np.sin(fw[3]*t + fw[4]) + 1)/2
    xr, vr = f32(x), f32(v)
    ref = o.step(xr, vr, xf, t_prev=(k-1)*cfg["h"])
    if k >= 26:
        e.set_uniform_force(fw[0:3]*factor); e.set_state(0, xr, vr); st = e.step_forward(0, fixed_pts=xf); x1, _ = e.get_state(1)
        sc = o.self_contacts(ref["id"]); got = e.get_self_contacts(1)
        want = sorted(zip(sc["layer"].tolist(), sc["p1"].tolist(), sc["p2"].tolist())); have = sorted(zip(got["layer"].tolist(), got["pairs"][:,0].tolist(), got["pairs"][:,1].tolist()))
        print(k, "conv", st["converged"][0], f"xdiff {st['last_xdiff'][0]:.2e}", "err", f"{np.abs(x1[0]-ref['x']).max():.2e}", "pd", st["pd_iters"][0], ref["iters"], "self", got["count"], ref["nself"], "same" if want == have else f"DIFF want {want} have {have}")
    x, v = ref["x"], ref["v"]

"""Developer script: the T-shirt system-identification demo end to end through diffcloth_py.OptimizeHelper."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import diffcloth_py as d
import scenes
V, F = scenes.load_mesh("tshirt")
sim = d.makeSimFromMesh("wind_tshirt", V.reshape(-1), F.reshape(-1).tolist())
t0 = time.time()
h = d.makeOptimizeHelperWithSim("wind_tshirt", sim)
print("helper (ground-truth rollout of", h.forward_steps, "steps):", round(time.time() - t0, 1), "s; params", list(h.paramName))
xt = h.getActualParam(); print("actual", xt)
t0 = time.time(); recs = h.runSimulationAndGetLossGradient(xt); print("loss at truth", recs[0].loss, round(time.time() - t0, 1), "s")
K = int(sys.argv[1]) if len(sys.argv) > 1 else h.forward_steps
h.forward_steps = K
if len(sys.argv) > 2:
    flags = [bool(int(c)) for c in sys.argv[2]]
    print('collision flags (contact, self):', flags)
    sim.setWindAndCollision(True, flags[0], flags[1], False)
x = xt.copy(); x[5] *= 0.8; x[0] *= 1.3
t0 = time.time(); recs = h.runSimulationAndGetLossGradient(x); g = h.gradientInfoToVecXd(recs[0]); print("loss", recs[0].loss, "grad", g, round(time.time() - t0, 1), "s")
for k, eps in ((5, 2.0), (0, 2e-4), (1, 2e-4), (3, 0.05), (4, 0.01)):
    xp = x.copy(); xp[k] += eps; xm = x.copy(); xm[k] -= eps
    fd = (h.runSimulationAndGetLoss(xp) - h.runSimulationAndGetLoss(xm)) / (2 * eps)
    print("param", k, h.paramName[k], "adjoint", g[k], "fd", fd)

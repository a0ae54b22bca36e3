import sys, os, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import diffcloth_py as d
import scenes
V, F = scenes.load_mesh("tshirt")
sim = d.makeSimFromMesh("wind_tshirt", V.reshape(-1), F.reshape(-1).tolist())
h = d.makeOptimizeHelperWithSim("wind_tshirt", sim)
xt = h.getActualParam()
x = xt.copy(); x[5] *= 0.9
for rep in range(2):
    t0 = time.time(); L = h.runSimulationAndGetLoss(x); tf = time.time() - t0
    t0 = time.time(); recs = h.runSimulationAndGetLossGradient(x); tfb = time.time() - t0
    print(f"T-shirt demo through diffcloth_py.OptimizeHelper ({h.forward_steps} steps): forward only {tf:.2f} s = {tf / h.forward_steps * 1e3:.2f} ms/step; forward+backward {tfb:.2f} s = {tfb / h.forward_steps * 1e3:.2f} ms/step -> {h.forward_steps / tfb:.1f} fwd+bwd steps/s (reference run: 5.55)")

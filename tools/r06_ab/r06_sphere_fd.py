"""Developer diagnostic (round 6): sphere demo, rollout-level dL/dmu by the adjoint against central finite differences for several horizons / steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))
import diffcloth_py as d
sim = d.makeSim("sphere")
h = d.makeOptimizeHelperWithSim("sphere", sim)
for steps in (30, 50, 80, 120, 200):
    h.forward_steps = steps
    for mu in (0.55, 0.15):
        x = np.array([mu])
        recs = h.runSimulationAndGetLossGradient(x)
        g = h.gradientInfoToVecXd(recs[0])[0]
        out = []
        for eps in (0.002, 0.005, 0.01, 0.02):
            fd = (h.runSimulationAndGetLoss(x + eps) - h.runSimulationAndGetLoss(x - eps)) / (2 * eps)
            out.append(f"eps {eps}: fd {fd:.4e} ratio {g / fd if fd else float('nan'):.3f}")
        print(f"steps {steps} mu {mu}: loss {recs[0].loss:.4e} adjoint {g:.4e} | " + " | ".join(out), flush=True)

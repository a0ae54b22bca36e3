# developer script: OpenMP thread scaling of the fp64 oracle (the CPU baseline of bench.py) on the GPU box's host
cd $GRAFT_REPO_ROOT
for t in 1 8 16 32 64 128; do
  python - <<PY
import sys, time, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, meshes, orc
V, F = meshes.grid_cloth(100, 100, 4.5, 4.5, "DOWN"); V = V.astype(np.float32).astype(np.float64)
c = meshes.sphere_scene_center(V, 2.0)
o = orc.Oracle(V, F, h=1/180, density=0.3, k_stretch=150.0, k_bend=1e-5, fwd_tol=1e-8, bwd_tol=5e-4, selfcollision=False, gradient_clipping=True, threads=$t)
o.add_sphere(c, 2.0, 0.5); o.build()
x = (V + np.array([0.1, -0.05, 0.1])).reshape(-1).copy(); v = np.zeros_like(x)
t0 = time.perf_counter(); recs = []
for s in range(2):
    out = o.step(x, v); recs.append(out); x, v = out["x"], out["v"]
tf = time.perf_counter() - t0
gx = x - V.reshape(-1); gv = np.zeros_like(gx)
for s in reversed(range(2)):
    b = o.step_backward(recs[s]["id"], gx, gv, is_start=(s == 0), direct=False); gx, gv = b["dL_dx"], b["dL_dv"]
dt = time.perf_counter() - t0
print(f"threads $t: fwd {tf/2*1e3:.0f} ms/step, fwd+bwd {dt/2*1e3:.0f} ms/step -> {2/dt:.3f} rollout-steps/s (host cores {os.cpu_count()})")
PY
done

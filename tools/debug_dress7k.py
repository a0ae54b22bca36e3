import os, sys, time
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes
from diffcloth_amd import capi
def f32(a): return np.asarray(a, dtype=np.float32).astype(np.float64)
V,F=scenes.load_mesh("dress7k")
P,rmin,rmax=scenes.normalise_model(V,"FRONT",8.0); P=f32(P)
top=np.argsort(-P[:,1])[:6].tolist()
rng=np.random.default_rng(8)
X=P.copy(); X[:,2]*=0.9; vel=np.zeros_like(X); vel[:,2]=-0.1*np.sign(P[:,2])
x0=f32((X+0.0005*rng.standard_normal(X.shape)).reshape(-1))[None,:]; v0=f32((vel+0.005*rng.standard_normal(X.shape)).reshape(-1))[None,:]
xf=f32(X[top].reshape(-1))[None,:]
for selfc in (0,1):
    e=capi.Engine(0); e.set_mesh(P,F); e.set_attachments(top)
    e.set_params(time_step=1/120,density=0.2,k_stretch=800.0,k_bend=0.05,forward_tol=1e-8,backward_tol=1e-9,cg_rel_tol=1e-6,cg_max_iter=3000,gradient_clipping=0,selfcollision_enabled=selfc,adjoint_mode=1,adjoint_rel_tol=1e-7)
    e.set_primitives([]); e.build(); e.alloc_batch(1,1)
    print("layout",e.layout(),"cluster",e.cluster())
    e.set_state(0,x0,v0)
    t=time.time(); st=e.step_forward(0,fixed_pts=xf); dt=time.time()-t
    x1,v1=e.get_state(1)
    print("selfc",selfc,"fwd",{k:v[0] for k,v in st.items()},"time %.3f s"%dt,"finite",np.isfinite(x1).all(), "max|x|",np.abs(x1).max())
    gx=f32(rng.standard_normal(x0.shape)); gv=f32(0.01*rng.standard_normal(x0.shape))
    t=time.time(); gb=e.step_backward(1,gx,gv,is_start=False); dt=time.time()-t
    print("   bwd",{k:gb[k][0] for k in ("converged","adjoint_iters","used_direct","last_udiff")},"time %.3f s"%dt,"finite",np.isfinite(gb["dL_dx"]).all())

#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the READ-ONLY reference checkout (run in the build container only).

The GPU box has no /root/reference, so everything the tests need from it is frozen here as small fixtures:
  meshes.npz         raw OBJ vertices / triangles of the demo meshes (src/assets/meshes/remeshed/...)
  tshirt_golden.npz  the reference's only golden data: output/tshirt-exampleopt/iter0 — first frames of the
                     1426-vertex T-shirt rollout of L-BFGS evaluation 0, with the parameters of that run
                     (iter0/param.txt), the losses / parameters / iteration counts of forwardLog.txt and the gradients of backwardLog.txt
  reference_callers/ the reference's own PyTorch caller layer, src/python_code/pySim/{__init__,functional,pySim}.py, frozen byte
                     for byte: tests/test_gpu_reference_callers.py imports and runs it UNMODIFIED against this repository's
                     diffcloth_py module (BASELINE.json north_star: "drops into the repo's ... hatController optimisation loops
                     unchanged"). These three files are the reference's, not this repository's work; they are test input.
"""
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_obj(path):
    V, F = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            V.append([float(p[1]), float(p[2]), float(p[3])])
        elif p[0] == "f":
            idx = [int(t.split("/")[0]) - 1 for t in p[1:]]
            for k in range(1, len(idx) - 1):
                F.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(V, dtype=np.float64), np.asarray(F, dtype=np.int32)


def freeze_reference_callers():
    import shutil
    dst = os.path.join(OUT, "reference_callers", "pySim")
    os.makedirs(dst, exist_ok=True)
    for f in ("__init__.py", "functional.py", "pySim.py"):
        shutil.copyfile(os.path.join(REF, "src/python_code/pySim", f), os.path.join(dst, f))
    print("froze", dst)
    # the controller-training script the north star names (hatController.py:78-105 is the loop) with the modules it imports, and the
    # target shape file its OptimizeHelper reads: run unmodified for one epoch by tests/test_gpu_reference_callers.py
    top = os.path.join(OUT, "reference_callers")
    os.makedirs(os.path.join(top, "clothNN"), exist_ok=True)
    for f in ("hatController.py", "common.py", "utils.py", "clothNN/__init__.py", "clothNN/controller.py"):
        shutil.copyfile(os.path.join(REF, "src/python_code", f), os.path.join(top, f))
    shutil.copyfile(os.path.join(REF, "src/assets/meshes/remeshed/Hat/hat_target.txt"), os.path.join(top, "hat_target.txt"))
    print("froze hatController.py, common.py, utils.py, clothNN/, hat_target.txt")


def main():
    freeze_reference_callers()
    meshes = {
        "hat": "src/assets/meshes/remeshed/agenthat2-579-rotated.obj",
        "tshirt": "src/assets/meshes/remeshed/T-shirt/tshirt1000-tri.obj",
        "sock": "src/assets/meshes/remeshed/sock1055-2081.obj",
        "dress": "src/assets/meshes/remeshed/dress-handsup-drape.obj",
        "dress7k": "src/assets/meshes/remeshed/dress-v7k-f14k.obj",      # 7 742 vertices: the garment-sized self-contact workload of tools/bench_dress7k.py
        # the reference's own 10k-class meshes: the 17 562-vertex dress (needs >= 3 workgroups per rollout) and the
        # 96 x 96 performance fabric lying on the slope plane (9 216 vertices: the one-workgroup kernels at 18 rows per thread)
        "dress17k": "src/assets/meshes/remeshed/dress-v17k-f34k.obj",
        "perf96": "src/assets/meshes/remeshed/Slope/perfFabric4-96x96-onPlane.obj",
    }
    out = {}
    for name, rel in meshes.items():
        V, F = load_obj(os.path.join(REF, rel))
        out[name + "_v"] = V
        out[name + "_f"] = F
        print(name, V.shape, F.shape)
    np.savez_compressed(os.path.join(OUT, "meshes.npz"), **out)

    run = os.path.join(REF, "output/tshirt-exampleopt")
    frames = []
    for k in range(0, 41):
        V, _ = load_obj(os.path.join(run, "iter0", f"{k}.obj"))
        frames.append(V)
    last, _ = load_obj(os.path.join(run, "iter0", "250.obj"))
    param = open(os.path.join(run, "iter0", "param.txt")).read()
    k_stretch = float(re.search(r"k_CONSTRAINT_TRIANGLE:([-\d.eE]+)", param).group(1))
    wind = [float(v) for v in re.search(r"f_wind:\(([^)]*)\)", param).group(1).split(",")]
    clips = [[float(v) for v in m.split(",")] for m in re.findall(r"CLIP_\d:([-\d.,eE]+)", param)]
    flog = open(os.path.join(run, "forwardLog.txt")).read()
    losses = [float(v) for v in re.findall(r"Loss:([-\d.eE]+)", flog)]
    pditers = [int(v) for v in re.findall(r"Total PD Iters:(\d+)", flog)]
    # parameters of every logged forward evaluation (forwardLog.txt prints them after each record)
    log_k = [float(v) for v in re.findall(r"k_CONSTRAINT_TRIANGLE:([-\d.eE]+)", flog)]
    log_wind = [[float(t) for t in m.split(",")] for m in re.findall(r"f_wind:\(([^)]*)\)", flog)]
    assert len(log_k) == len(log_wind) == len(losses)
    # backwardLog.txt: gradients (4-5 decimals) and adjoint iteration totals of every backward sweep
    blog = open(os.path.join(run, "backwardLog.txt")).read()
    grad_k = [float(v) for v in re.findall(r"dL/dk_CONSTRAINT_TRIANGLE:([-\d.eE]+)", blog)]
    grad_wind = [[float(t) for t in m.split(",")] for m in re.findall(r"dL/df_wind:\(([^)]*)\)", blog)]
    bwd_iters = [int(v) for v in re.findall(r"Total Backward Iter:(\d+)", blog)]
    bwd_fwd_idx = [int(v) for v in re.findall(r"Corresponding forward Idx: (\d+)", blog)]
    np.savez_compressed(os.path.join(OUT, "tshirt_golden.npz"), frames=np.asarray(frames, dtype=np.float64),
                        frame250=last, k_stretch=k_stretch, f_wind=np.asarray(wind), clips=np.asarray(clips),
                        losses=np.asarray(losses), pd_iters=np.asarray(pditers), log_k=np.asarray(log_k),
                        log_wind=np.asarray(log_wind), grad_k=np.asarray(grad_k), grad_wind=np.asarray(grad_wind),
                        bwd_iters=np.asarray(bwd_iters), bwd_fwd_idx=np.asarray(bwd_fwd_idx))
    print("golden frames", np.asarray(frames).shape, "k", k_stretch, "wind", wind, "clips", clips, "loss0", losses[0], "pd0", pditers[0])


if __name__ == "__main__":
    main()

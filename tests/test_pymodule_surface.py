"""The pybind11 module `diffcloth_py` must expose the reference's Python surface
(/root/reference/src/code/python_interface.cpp:164-378) — checked here by name, without a GPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))


def test_module_surface():
    d = pytest.importorskip("diffcloth_py")
    for fn in ["makeSim", "makeOptimizeHelper", "makeOptimizeHelperWithSim", "enableOpenMP", "render"]:
        assert callable(getattr(d, fn)), fn
    for cls in ["WindConfig", "SceneConfiguration", "PrimitiveCollisionInformation", "SelfCollisionInformation",
                "ForwardInformation", "BackwardInformation", "BackwardTaskInformation", "LossInfo", "Primitive",
                "Simulation", "OptimizeHelper"]:
        assert hasattr(d, cls), cls
    sim_attrs = ["taskLossInfo", "primitives", "sceneConfig", "forwardRecords", "useCustomRLFixedPoint", "perStepGradient",
                 "gradientClipping", "gradientClippingThreshold", "ndof_u", "num_particles", "forwardConvergenceThreshold",
                 "backwardConvergenceThreshold", "resetSystem", "step", "getCurrentPosVelocityVec", "appendPerStepGradient",
                 "stepNN", "setWindAndCollision", "getStateInfo", "setAction", "exportCurrentMeshPos", "setPrintVerbose",
                 "getPastStateInfo", "exportCurrentSimulation", "stepBackward", "stepBackwardNN"]
    for a in sim_attrs:
        assert hasattr(d.Simulation, a), a
    for a in ["x", "v", "x_prev", "v_prev", "f", "r", "x_fixedpoints", "stepIdx", "sysMatId", "t", "avgDeformation",
              "maxDeformation", "collisionInfos"]:
        assert hasattr(d.ForwardInformation, a), a
    for a in ["dL_dx", "dL_dv", "dL_dfext", "dL_dxfixed", "dL_dwind", "dL_ddensity", "dL_dk_pertype", "dL_dmu", "loss",
              "totalRuntime", "converged", "convergedAccum", "backwardIters", "backwardTotalIters"]:
        assert hasattr(d.BackwardInformation, a), a
    for a in ["taskInfo", "lossInfo", "lossType", "sim", "forward_steps"]:
        assert hasattr(d.OptimizeHelper, a), a
    assert d.WindConfig.WIND_SIN != d.WindConfig.NO_WIND
    assert d.Primitive.PrimitiveType.SPHERE is not None


def test_unknown_example_raises_like_the_reference():
    d = pytest.importorskip("diffcloth_py")
    with pytest.raises(Exception, match=r"Undefined example name \(nope\)"):
        d.makeSim("nope")


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = pytest.importorskip("diffcloth_py")
    import scenes
    V, F = scenes.load_mesh("hat")
    with pytest.raises(Exception, match="no CPU path"):
        d.makeSimFromMesh("wear_hat", V.reshape(-1), F.reshape(-1).tolist())

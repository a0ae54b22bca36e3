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
                "ForwardInformation", "BackwardInformation", "BackwardTaskInformation", "LossInfo", "CorresPondenceTargetInfo", "Primitive",
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
    for a in ["targetLoc", "targetTranslation", "targetFrameShape", "targetPosPairs"]:      # python_interface.cpp:252-256
        assert hasattr(d.LossInfo, a), a
    c = d.CorresPondenceTargetInfo()                                                        # python_interface.cpp:245-249
    c.frameIdx = 7; c.targetPos = [1.0, 2.0, 3.0]; c.particleIndices = [4, 5]
    assert c.frameIdx == 7 and list(c.targetPos) == [1.0, 2.0, 3.0] and list(c.particleIndices) == [4, 5]
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


def test_spline_trajectory_math():
    """Cubic Hermite fixed-point trajectories (reference Spline.h): end-point interpolation, multi-segment lookup, and
    dxfixed_dcontrolPoints against finite differences of evalute() for the three parameterisations."""
    d = pytest.importorskip("diffcloth_py")
    import numpy as np
    rng = np.random.default_rng(0)
    p0, p1, p2 = rng.standard_normal(3), rng.standard_normal(3), rng.standard_normal(3)
    for typ, nper in ((d.SplineType.ENDPOINT, 3), (d.SplineType.ENDPOINT_AND_UP, 4), (d.SplineType.ENDPOINT_AND_TANGENTS, 9)):
        s = d.Spline(p0, p1, 8.0, 1, 0.0, 0.5)
        s.addSegment(p2, 3.0, 0.5, 1.0)
        s.type = typ
        assert s.getParameterNumber() == 2 * nper and s.pFixed == 1
        np.testing.assert_allclose(s.evalute(0.0), p0, atol=1e-14)
        np.testing.assert_allclose(s.evalute(0.5), p1, atol=1e-14)
        np.testing.assert_allclose(s.evalute(1.0), p2, atol=1e-14)
        np.testing.assert_allclose(s.evalute(1.7), p2, atol=1e-14)          # clamped
        for t in (0.13, 0.41, 0.77):
            J = s.dxfixed_dcontrolPoints(t)
            assert J.shape == (3, 2 * nper)
            seg = 0 if t <= 0.5 else 1
            for q in range(2 * nper):
                if q // nper != seg:
                    assert np.all(J[:, q] == 0)          # the reference differentiates the active segment only
                    continue
                e = np.zeros(2 * nper); e[q] = 1e-6
                sp = d.Spline(p0, p1, 8.0, 1, 0.0, 0.5); sp.addSegment(p2, 3.0, 0.5, 1.0); sp.type = typ
                sm = d.Spline(p0, p1, 8.0, 1, 0.0, 0.5); sm.addSegment(p2, 3.0, 0.5, 1.0); sm.type = typ
                sp.updateControlPoints(e); sm.updateControlPoints(-e)
                fd = (sp.evalute(t) - sm.evalute(t)) / 2e-6
                np.testing.assert_allclose(J[:, q], fd, atol=1e-6)
    # first-order evaluation = derivative w.r.t. the local spline time
    s = d.Spline(p0, p1, 8.0, 0)
    fd = (s.evalute(0.3 + 1e-6) - s.evalute(0.3 - 1e-6)) / 2e-6
    np.testing.assert_allclose(s.evalute(0.3, 1), fd, atol=1e-6)


def test_batched_autograd_module_imports_without_a_device():
    """diffcloth_amd.functional is importable on a CPU-only host; it refuses to wrap an engine without a batch."""
    sys.path.insert(0, ROOT)
    from diffcloth_amd import functional

    class NoBatch:
        B = 0
        tape = 0
    with pytest.raises(ValueError):
        functional.BatchedSim(NoBatch(), 10)
    assert callable(functional.sim_step)

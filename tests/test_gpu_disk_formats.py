"""GPU: on-disk formats of a run — frames written in the reference's layouts
(Simulation.cpp:3788-3851, 4003-4238; MeshFileHandler.h:137-160), read back by the folder reader the reference's
viewer uses (Simulation.h:574-620), and the per-evaluation log files of an optimisation run."""
import os
import re
import sys

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))


@pytest.fixture()
def out_root(tmp_path):
    import diffcloth_py as d
    saved = d.Simulation.outputRoot
    d.Simulation.outputRoot = str(tmp_path)
    yield tmp_path
    d.Simulation.outputRoot = saved


def test_frames_round_trip_through_the_viewer_layout(out_root):
    import diffcloth_py as d
    V, F = scenes.load_mesh("hat")
    sim = d.makeSimFromMesh("wear_hat", V.reshape(-1), F.reshape(-1).tolist())
    sim.resetSystem()
    for _ in range(3):
        sim.step()
    frames = [np.array(r.x) for r in sim.forwardRecords]
    clips = [np.array(r.x_fixedpoints) for r in sim.forwardRecords]
    assert len(frames) == 4

    sim.exportSimulation("run/iter0")                     # <root>/run/iter0/<i>.obj + info.txt
    folder = out_root / "run" / "iter0"
    assert sorted(p.name for p in folder.iterdir()) == ["0.obj", "1.obj", "2.obj", "3.obj", "info.txt"]
    first = (folder / "2.obj").read_text().splitlines()
    assert first[0].startswith("v ") and first[579].startswith("f ") and len(first) == 579 + len(F)
    assert [int(t) for t in first[579].split()[1:]] == [int(i) + 1 for i in F[0]]      # 1-based, caller's vertex numbering
    info = (folder / "info.txt").read_text().splitlines()
    assert len(info) == 2 and all(re.fullmatch(r"CLIP_\d:-?\d+\.\d{5},-?\d+\.\d{5},-?\d+\.\d{5}", l) for l in info)
    np.testing.assert_allclose([float(t) for t in info[1].split(":")[1].split(",")], clips[3][3:6], atol=5.1e-6)

    other = d.makeSimFromMesh("wear_hat", V.reshape(-1), F.reshape(-1).tolist())
    other.resetSystem()
    assert other.resetForwardRecordsFromFolder("run/iter0") == 4
    recs = other.forwardRecords
    assert len(recs) == 5                                 # the reset record + one per frame file
    att = other.getAttachmentVertices()
    for i in range(4):
        x = np.array(recs[1 + i].x)
        np.testing.assert_allclose(x, frames[i], rtol=6e-6, atol=1e-6)          # OBJ text keeps 6 significant digits
        assert recs[1 + i].stepIdx == i and abs(recs[1 + i].t - 0.01 * i) < 1e-12
        np.testing.assert_array_equal(np.array(recs[1 + i].x_fixedpoints), x.reshape(-1, 3)[att].reshape(-1))

    sim.exportCurrentSimulation("full")                   # <root>/full/<i>/0-CLOTH.obj + info.txt, <root>/area.txt
    assert sorted(p.name for p in (out_root / "full" / "3").iterdir()) == ["0-CLOTH.obj", "info.txt"]
    areas = (out_root / "area.txt").read_text().splitlines()
    assert len(areas) == 4 and areas[0].startswith("Frame 0:")
    P = frames[0].reshape(-1, 3)
    a0 = 0.5 * np.linalg.norm(np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]]), axis=1).sum()
    assert abs(float(areas[0].split(":")[1]) - a0) < 1e-6
    sim.exportCurrentMeshPos(1, "pose")                   # <root>/pose.txt (3 decimals) + pose.obj
    txt = np.loadtxt(out_root / "pose.txt")
    np.testing.assert_allclose(txt, frames[1].reshape(-1, 3), atol=5.1e-4)
    pts, tris = d.loadObjFile(str(out_root / "pose.obj"))
    np.testing.assert_array_equal(tris.reshape(-1, 3), F)


def test_optimisation_run_writes_the_reference_log_files(out_root):
    import diffcloth_py as d
    V, F = scenes.load_mesh("tshirt")
    sim = d.makeSimFromMesh("wind_tshirt", V.reshape(-1), F.reshape(-1).tolist())
    h = d.makeOptimizeHelperWithSim("wind_tshirt", sim)
    h.forward_steps = 6
    x = h.getActualParam(); x[5] = 556.0163134; x[:5] = [-0.0211231, 0.0566203, 0.0596879, 13.6755941, -3.0244862]
    # the parameter text of the reference's own iter0 (output/tshirt-exampleopt/iter0/param.txt)
    assert d.Simulation.parameterToString(h.taskInfo, h.vecXdToParamInfo(x)) == (
        "============Parameter Info:======================\nk_CONSTRAINT_TRIANGLE:556.016313\n"
        "f_wind:(-0.021123,0.056620,0.059688,13.675594,-3.024486)\n")
    L0, g0 = h.evaluate(x)
    x2 = x.copy(); x2[5] = 500.0
    L1, g1 = h.evaluate(x2)
    assert L0 > 0 and L1 > 0 and g0.shape == (6,) and np.isfinite(g1).all()
    run = out_root / (h.experimentName + "-LBFGS")
    names = sorted(p.name for p in run.iterdir())
    assert names == ["backwardLog.txt", "forwardLog.txt", "iter0", "iter1", "iters.txt", "last_frame_meshes", "perf.txt",
                     "scene-config.txt", "task_info.txt"]
    assert (run / "iters.txt").read_text() == "Total forward:2\nTotal backprop:2"
    fl = (run / "forwardLog.txt").read_text()
    assert fl.count("Record ") == 2 and f"Loss:{L0:.5f}\n" in fl and f"Loss:{L1:.5f}\n" in fl
    assert "k_CONSTRAINT_TRIANGLE:500.000000\n" in fl and re.search(r"Total PD Iters:\d+\nTotal Frames Converged:6\n", fl)
    bl = (run / "backwardLog.txt").read_text()
    assert bl.count("Record ") == 2 and "dL/dk_CONSTRAINT_TRIANGLE:" in bl and "dL/df_wind:(" in bl and "Corresponding forward Idx: 1\n" in bl
    assert sorted(p.name for p in (run / "iter1").iterdir()) == [f"{i}.obj" for i in range(7)] + ["info.txt", "param.txt"]
    assert (run / "iter0" / "param.txt").read_text().splitlines()[1] == "k_CONSTRAINT_TRIANGLE:556.016313"
    perf = (run / "perf.txt").read_text()
    assert "Total Particles:1426\n" in perf and perf.count("iter1:") == 2 and "Total Time:" in perf
    task = (run / "task_info.txt").read_text()
    assert "CONSTRAINT_TRIANGLE: ON\n" in task and "f_wind: ON\n" in task
    # replay: the frames of iteration 1 are the rollout the helper just ran
    assert sim.resetForwardRecordsFromFolder(h.experimentName + "-LBFGS/iter1") == 7

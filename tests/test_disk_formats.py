"""On-disk formats of the host class that need no device: the parameter text the reference
writes next to every optimisation iteration, and the OBJ reader.

Known answer: output/tshirt-exampleopt/iter0/param.txt of the reference repository (written by
Simulation::parameterToString, Simulation.cpp:4283-4352) for the T-shirt task (stretching stiffness + 5 wind parameters).
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "diffcloth_amd", "lib"))

REFERENCE_ITER0_PARAM_TXT = ("============Parameter Info:======================\n"
                             "k_CONSTRAINT_TRIANGLE:556.016313\n"
                             "f_wind:(-0.021123,0.056620,0.059688,13.675594,-3.024486)\n")


def test_parameter_text_reproduces_the_reference_param_txt():
    m = pytest.importorskip("diffcloth_py")
    task = m.BackwardTaskInformation()
    task.dL_dfwind = True
    p = m.ParamInfo()
    p.f_extwind = np.array([-0.0211231, 0.0566203, 0.0596879, 13.6755941, -3.0244862])
    lines = REFERENCE_ITER0_PARAM_TXT.splitlines(keepends=True)
    # (ParamInfo.k_pertype is read-only from Python as in the reference; the stiffness line is covered on the GPU through
    # OptimizeHelper.vecXdToParamInfo, tests/test_gpu_pymodule.py)
    assert m.Simulation.parameterToString(task, p) == lines[0] + lines[2]


def test_obj_reader_accepts_the_face_forms_obj_files_use(tmp_path):
    m = pytest.importorskip("diffcloth_py")
    f = tmp_path / "a.obj"
    f.write_text("# comment\nv 0 0 0\nv 1 0 0.5\nvn 0 0 1\nv 0 1e-3 -2\nvt 0 0\nf 1 2 3\nf 3/1 2/1 1/1\nf 1//1 3//1 2//1\nf 2/1/1 3/1/1 1/1/1\n")
    pts, tris = m.loadObjFile(str(f))
    np.testing.assert_array_equal(pts.reshape(-1, 3), [[0, 0, 0], [1, 0, 0.5], [0, 1e-3, -2]])
    np.testing.assert_array_equal(tris.reshape(-1, 3), [[0, 1, 2], [2, 1, 0], [0, 2, 1], [1, 2, 0]])

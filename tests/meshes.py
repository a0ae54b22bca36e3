"""Synthetic cloth meshes for the tests: the definitions live in the package (diffcloth_amd/workloads.py — bench.py builds its workload
from the same functions); this module re-exports them under the names the tests have always used, so that oracle and product are
fed the same raw mesh."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffcloth_amd.workloads import axis_to_rotation, fold_flap, grid_cloth, orient, sphere_scene_center  # noqa: E402,F401

"""Scene construction for the tests: model-mesh normalisation, attachment search, demo tables and primitive placement are defined in
diffcloth_amd/workloads.py (they restate Simulation.cpp:2170-2226, :2258-2310, :1894-1944 and OptimizationTaskConfigurations.cpp so the
oracle can be fed the same scenes as the product's C++ host class, which implements them independently); re-exported here."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffcloth_amd.workloads import (DRESS, GOLDEN, HAT, PERF_FABRIC, SOCK, TSHIRT, corner_attachments, hat_head_center, load_mesh,  # noqa: E402,F401
                                     normalise_model, slope_plane, sock_leg)

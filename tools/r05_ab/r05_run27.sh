# the remaining 1024-thread adjoint instances behind the LDS y list: coarse level (dress 7742), block preconditioner (dress 3634 x 256), general kernels (dress 17562)
OUT=gpurun_out/r05_run27; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 180 python -m pytest "tests/test_gpu_configs.py::test_dress_7742_vertices_forward_step_and_adjoint_fallback" "tests/test_gpu_configs.py::test_c4_dress_self_contact_batch" tests/test_gpu_garments10k.py::test_dress_17562_vertices_self_contacts_and_clips -v -x 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed|Error|assert" | cut -c1-200 > $OUT/tests.log; tail -20 $OUT/tests.log

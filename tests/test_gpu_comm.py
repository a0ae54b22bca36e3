"""dc_comm_* / dc_allreduce_sum: the RCCL collective of the C-ABI (C++ callers; SURVEY.md section 8 (b) "dc_allreduce"). One GPU is all the
test box has, so this pins the run-time binding of RCCL, communicator set-up and the host -> device -> ncclAllReduce -> host path at
world size 1; the N > 1 exchange itself is the same call (bench.py's Python path over torch.distributed is what the driver scales)."""
import numpy as np
import pytest

from diffcloth_amd import capi

pytestmark = pytest.mark.gpu


def test_allreduce_sum_through_the_c_abi_world_size_one():
    e = capi.Engine(0)
    with pytest.raises(capi.DcError, match="dc_comm_init has not been called"):
        e.allreduce_sum([1.0, 2.0])
    uid = capi.Engine.comm_unique_id()
    assert len(uid) == 128 and any(b != 0 for b in uid)
    e.comm_init(1, 0, uid)
    v = np.array([1.5, -2.25, 3.0e10, 1e-300])
    out = e.allreduce_sum(v)
    np.testing.assert_array_equal(out, v)            # sum over one rank, in double precision end to end
    e.comm_destroy()
    with pytest.raises(capi.DcError, match="dc_comm_init has not been called"):
        e.allreduce_sum([1.0])
    with pytest.raises(capi.DcError):
        e.comm_init(2, 5, uid)                        # rank outside the communicator

"""GPU: the multi-GPU exchange entry points (elfi_amd/csrc/comm.hip) at world_size 1 -- all this single-GPU box can run:
RCCL loads at run time, a communicator comes up on the context's GPU, the collectives move the data they should and
elfihip_comm_bcast_factor leaves a factorised GP usable.  (World sizes > 1 are exercised by the driver's multi-GPU
bench through torch.distributed; the partitioning and merge logic around the exchanges by tests/test_sharding_gloo.py.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_round_trip(hip_ctx):
    import torch
    from elfi_amd.gp import GPHandle
    lib = hip_ctx.lib
    uid = (C.c_char * 128)()
    hip_ctx.call("elfihip_comm_unique_id", uid)
    comm = C.c_void_p()
    hip_ctx.call("elfihip_comm_init_rank", uid, 0, 1, C.byref(comm))
    try:
        a = torch.arange(1000, dtype=torch.float64, device='cuda') * 0.5
        out = torch.zeros(1000, dtype=torch.float64, device='cuda')
        torch.cuda.synchronize()
        assert lib.elfihip_comm_allgather_f64(comm, a.data_ptr(), 1000, out.data_ptr()) == 0
        hip_ctx.synchronize()
        assert torch.equal(out, a)
        out.zero_()
        torch.cuda.synchronize()
        assert lib.elfihip_comm_gather_f64(comm, a.data_ptr(), 1000, out.data_ptr(), 0) == 0
        assert lib.elfihip_comm_bcast_f64(comm, out.data_ptr(), 1000, 0) == 0
        hip_ctx.synchronize()
        assert torch.equal(out, a)
        assert lib.elfihip_comm_gather_f64(comm, a.data_ptr(), 10, out.data_ptr(), 3) != 0      # no such root
        # a factorised GP survives its own broadcast
        rs = np.random.RandomState(0)
        X, y = rs.uniform(-1, 1, (300, 3)), rs.randn(300)
        gp = GPHandle(3, 300)
        gp.set_hyper(1.0, 0.5, 0.1, 0.05)
        gp.set_data(X, y)
        lz = gp.factorize()
        mu0, var0 = gp.predict(X[:7])
        assert lib.elfihip_comm_bcast_factor(comm, gp.h, 0) == 0
        mu1, var1 = gp.predict(X[:7])
        assert np.array_equal(mu0, mu1) and np.array_equal(var0, var1) and np.isfinite(lz)
        gp.close()
    finally:
        assert lib.elfihip_comm_free(comm) == 0

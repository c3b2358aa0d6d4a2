"""licv_allreduce_small (SURVEY.md section 8(b) minimum export, 8(e)): the C-ABI wrapper over the RCCL of the calling process.  The GPU
boxes have one GPU: a one-rank communicator (ncclCommInitAll) exercises the whole call path - symbol lookup in the already loaded RCCL,
argument order, in-place operation, stream ordering; the two-rank arithmetic of the trainer's collective is tests/test_dp_gpu.py's."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rccl():
    return ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)


def test_allreduce_small_on_a_one_rank_communicator():
    from licv import _lib
    torch.cuda.init()
    rccl = _rccl()
    comm = ctypes.c_void_p()
    devs = (ctypes.c_int * 1)(0)
    assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs) == 0
    try:
        n = 131072 + 32                                            # icv (32 x 4096) + alpha (32)
        x = torch.randn(n, device="cuda")
        ref = x.clone()
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for avg in (0, 1):
                _lib.check(_lib.lib().licv_allreduce_small(comm, ctypes.c_void_p(x.data_ptr()), n, avg, ctypes.c_void_p(st.cuda_stream)))
        st.synchronize()
        assert torch.equal(x, ref)                                 # one rank: the sum and the mean are the input itself
        _lib.check(_lib.lib().licv_allreduce_small(comm, ctypes.c_void_p(x.data_ptr()), 0, 0, None))       # empty: no call into RCCL
    finally:
        rccl.ncclCommDestroy(comm)


def test_allreduce_small_rejects_bad_arguments():
    from licv import _lib
    x = torch.zeros(8, device="cuda")
    with pytest.raises(Exception):
        _lib.check(_lib.lib().licv_allreduce_small(None, ctypes.c_void_p(x.data_ptr()), 8, 0, None))
    with pytest.raises(Exception):
        _lib.check(_lib.lib().licv_allreduce_small(ctypes.c_void_p(1), None, 8, 0, None))

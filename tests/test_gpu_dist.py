"""GPU: the C-ABI of include/fbsmi_dist.h on ONE rank (RCCL refuses two ranks on one card, so a one-GPU box can only run the
communicator with world = 1; the peer windows run with 2 and 3 ranks on one card in tests/test_gpu_sharded.py):
fbsmi_dist_logsumexp / fbsmi_dist_resample_exchange must return exactly what the single-GPU entries return."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rccl", [True, False])
def test_one_rank_context_equals_single_gpu_entries(rccl, dev):
    from fbs_amd import ops
    from fbs_amd.sharded import DistContext
    R = 70001
    ctx = DistContext(R, rccl=rccl, device=dev)
    assert (ctx.n, ctx.offset, ctx.count) == (R, 0, R)
    lw = 3.0 * ops.normal(ops.PRNGKey(5), (R,), device=dev) - 40.0
    want, lse, ess = ops.normalise(lw, log_space=True, return_lse=True, return_ess=True)
    for log_space in (True, False):
        got, l2, e2 = ctx.logsumexp(lw, log_space=log_space)
        ref = want if log_space else ops.normalise(lw, log_space=False)
        assert torch.equal(got, ref) and torch.equal(l2.reshape(()), lse.reshape(())) and torch.equal(e2.reshape(()), ess.reshape(()))
    A = torch.from_numpy(np.random.default_rng(0).integers(0, R, R).astype(np.int32)).to(dev)
    for shape in ((1,), (3,), (8,), (5, 4), (300,), (1028,)):
        rows = ops.normal(ops.PRNGKey(6), (R,) + shape, device=dev)
        want_rows = rows[A.long()]
        assert torch.equal(ctx.exchange(A, rows, mode="all_gather"), want_rows)
    # peer windows with a single owner: publish -> (collective) -> gather, twice, so that both buffers are used
    ctx.open_windows(1028)
    for shape, seed in (((1028,), 7), ((257, 4), 8), ((6,), 9), ((1,), 10)):
        rows = ops.normal(ops.PRNGKey(seed), (R,) + shape, device=dev)
        ctx.publish(rows)
        assert torch.equal(ctx.exchange(A, mode="peer", rowshape=shape), rows[A.long()])
        assert torch.equal(ctx.read_row(R - 1, shape), rows[R - 1])
    with pytest.raises(RuntimeError):
        ctx.exchange(A, mode="peer", rowshape=(5,))            # not the published row size
    with pytest.raises(RuntimeError):
        ctx.publish(torch.zeros((R, 2000), device=dev))        # wider than the window
    ctx.close()


def test_context_without_communicator_refuses_collectives(dev):
    from fbs_amd import _lib
    import ctypes as C
    D = _lib.dist_lib()
    h = C.c_void_p()
    # rank 1 of 2 without an id: shard arithmetic works, the collective entries refuse
    assert D.fbsmi_dist_create(None, 1, 2, 101, C.byref(h)) == 0
    n, off, cnt = C.c_int64(), C.c_int64(), C.c_int64()
    assert D.fbsmi_dist_shard(h, C.byref(n), C.byref(off), C.byref(cnt)) == 0
    assert (n.value, off.value, cnt.value) == (51, 51, 50)
    lw = torch.zeros(50, device=dev)
    out = torch.zeros(101, device=dev)
    assert D.fbsmi_dist_logsumexp(h, lw.data_ptr(), 1, out.data_ptr(), None, None, None) == -1
    assert b"communicator" in D.fbsmi_dist_last_error()
    A = torch.zeros(101, dtype=torch.int32, device=dev)
    assert D.fbsmi_dist_resample_exchange(h, lw.data_ptr(), A.data_ptr(), 1, out.data_ptr(), 1, None) == -1   # no windows
    assert D.fbsmi_dist_destroy(h) == 0

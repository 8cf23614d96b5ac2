"""The C-ABI communicator handle (mhimx_comm_*, csrc/comm.hip) on one GPU: a world of one rank (RCCL refuses two ranks on one
device, so the multi-rank sum itself is exercised by the driver's multi-GPU runs; the 2-rank control flow is covered over gloo in
tests/test_dp_cpu.py)."""
import pytest
import torch

from mhim_mil_amd import comm, engine

pytestmark = pytest.mark.gpu


def test_world_of_one_allreduce_is_identity_in_both_modes():
    c = comm.NativeComm(0, 1)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    c.allreduce(x, mode=0)
    c.allreduce(x, mode=1)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                      # enqueue-only on the caller's stream
        c.allreduce(x)
    s.synchronize()
    assert torch.equal(x, ref)
    c.close()
    c.close()


def test_unique_ids_differ_and_are_128_bytes():
    a, b = comm.unique_id(), comm.unique_id()
    assert len(a) == 128 and len(b) == 128 and a != b


def test_sync_flat_gradient_takes_the_native_handle():
    c = comm.NativeComm(0, 1)
    grad = torch.randn(1000, device="cuda")
    student = torch.randn(1000, device="cuda")
    g0, s0 = grad.clone(), student.clone()
    # world = 2 code path with a one-rank communicator: SUM over one rank, tail averaged by 1/2 (exactly what two equal ranks give / 2)
    scale = engine.sync_flat_gradient(grad, student, 900, 2, None, c)
    torch.cuda.synchronize()
    assert scale == 0.5
    assert torch.equal(grad[:900], g0[:900]) and float(grad[900:].abs().max()) == 0.0
    assert torch.allclose(student[900:], s0[900:] * 0.5)
    c.close()

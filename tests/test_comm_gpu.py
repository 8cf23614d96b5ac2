"""The C-ABI communicator handle (mhimx_comm_*, csrc/comm.hip) on one GPU: a world of one rank (RCCL refuses two ranks on one
device; the 2-rank control flow is covered over gloo in tests/test_dp_cpu.py) - and, wherever TWO GPUs are visible, the c4 and c5
equivalence tests with one device per rank over RCCL itself, plus bench.py's self-launch with its run-time choice of the exchange."""
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


# ------------------------------------------------------------------------------------------------------------------------------
# Two DEVICES, RCCL itself: skipped on the one-GPU boxes of this pool, run wherever two GPUs are visible
# ------------------------------------------------------------------------------------------------------------------------------
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks on one device")


@two_gpus
@pytest.mark.parametrize("mode", ["eager", "split_graph"])
def test_c4_two_devices_over_rccl_equal_accumulation_two(tmp_path, mode):
    """tests/test_sharded_gpu.py's c4 equivalence (two ranks, two different bags per step == one process with accumulation_steps = 2) with
    one DEVICE per rank and the gradient exchange over RCCL."""
    import os
    import numpy as np
    import torch.multiprocessing as mp
    from tests import test_sharded_gpu as T
    from mhim_mil_amd.engine import FusedTrainer
    s, t = T._models()
    tr = FusedTrainer(s, t, aux_alpha=0.5, mm=0.999, accumulation_steps=2)
    ref_logits = []
    for step in range(2):
        for rank in range(2):
            x, lab = T._dp_bag(0 if mode == "split_graph" else step, rank)
            perm, shuf = T._dp_draws(0 if mode == "split_graph" else step, rank)
            lg, _ = tr.train_step(x, lab, perm=perm, ids_shuffle=shuf)
            ref_logits.append(lg.cpu().clone())
    torch.cuda.synchronize()
    s_ref = {k: v.detach().cpu() for k, v in s.state_dict().items()}
    port = 43500 + (os.getpid() % 1000) + (0 if mode == "eager" else 1000)
    mp.spawn(T._dp_diff_worker, args=(port, str(tmp_path), mode, "nccl"), nprocs=2, join=True)
    res = [torch.load(os.path.join(tmp_path, f"dd{r}.pt")) for r in range(2)]
    for rank, r in enumerate(res):
        for step in range(2):
            np.testing.assert_allclose(r["logits"][step].numpy(), ref_logits[2 * step + rank].numpy(), atol=3e-5, rtol=0)
        for k, v in s_ref.items():
            err = (r["stu"][k].double() - v.double()).abs()
            assert err.max().item() <= (3e-6 if "global_q" in k else 2 * 4.1e-4), (k, err.max().item())
    for k in res[0]["stu"]:
        assert torch.equal(res[0]["stu"][k], res[1]["stu"][k]), k


@two_gpus
def test_c5_two_devices_over_rccl_equal_single_process(tmp_path):
    """The instance-sharded step (sharded Merge included) with one device per rank over RCCL == the single-process trainer."""
    import os
    import numpy as np
    import torch.multiprocessing as mp
    from tests import test_sharded_gpu as T
    outs, s_ref, t_ref = T._reference_run()
    port = 45500 + (os.getpid() % 1000)
    mp.spawn(T._worker, args=(port, str(tmp_path), "nccl"), nprocs=2, join=True)
    res = [torch.load(os.path.join(tmp_path, f"g{r}.pt")) for r in range(2)]
    for r in res:
        for step in range(2):
            np.testing.assert_allclose(r["logits"][step].numpy(), outs[step][0].numpy(), atol=2e-5, rtol=0)
        for ref, got in ((s_ref, r["stu"]), (t_ref, r["tea"])):
            for k, v in ref.items():
                err = (got[k].double() - v.double()).abs()
                assert err.mean().item() <= 2e-6 and err.max().item() <= 4.1e-4, (k, err.mean().item(), err.max().item())
    for k in res[0]["stu"]:
        assert torch.equal(res[0]["stu"][k], res[1]["stu"][k]), k


def test_bench_self_launches_and_picks_a_collective(tmp_path):
    """`python bench.py --gpus 2` with no launcher (the shape of the driver's command): re-executes itself under torch.distributed.run, times
    the candidate forms of the gradient exchange and prints ONE JSON line.  MHIMX_BENCH_SELFTEST=1: both ranks on this box's one GPU over
    gloo - the code path, not the numbers."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MHIMX_BENCH_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-steps", "0",
                        "--no-extras", "--no-kernel-events"], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-400:], r.stderr[-800:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    cands = d["config"]["collective_candidates_ms_per_step"]
    assert d["config"]["collective"] in cands and isinstance(cands[d["config"]["collective"]], float)
    assert any("eager" in k for k in cands) and any("torch.distributed all_reduce" in k for k in cands)
    wire = cands["flat_gradient_all_reduce_alone"]["torch.distributed all_reduce"]                  # the bare exchange beside the step times
    assert wire["bytes"] > 6e6 and wire["ms"] > 0 and wire["bus_GBps"] > 0


@pytest.mark.parametrize("workload", ["c5", "c3-sharded"])
def test_bench_sharded_workloads_self_launch(tmp_path, workload):
    """`python bench.py --workload c5 --gpus 2` (one bag sharded by rows, the exchanges between hipGraph segments) and the
    sequence-parallel MHIM(TransMIL) step (`--workload c3-sharded`) through the SAME argument plumbing the driver's multi-GPU run uses -
    self-launch under torch.distributed.run, barriers, max-over-ranks timing, rank 0's ONE JSON line.  MHIMX_BENCH_SELFTEST=1: both ranks on
    this box's one GPU over gloo - the code path, not the numbers."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MHIMX_BENCH_SELFTEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--cpu-steps", "0"], capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-400:], r.stderr[-1200:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert "2 GPU(s)" in d["config"]["parallelism"] and workload in d["config"]["workload"]
    if workload == "c5":
        assert d["config"]["launch"].startswith("hipGraph segments"), d["config"]["launch"]

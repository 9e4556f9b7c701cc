"""The N > 1 path on real GPUs, SELF-ARMING: when the box has two or more GPUs these tests launch
`torch.distributed.run` with one rank per GPU over RCCL, shard 8 (and 7: a ragged last shard) pairs, gather the match records with
shard.gather_records AND with imx_gather_records, and compare the record bytes with the single-GPU batch; on a 1-GPU box the
world-2 cases skip and the same worker runs at world 1 (every statement of it over the real RCCL backend), so that the day a
multi-GPU node exists the suite proves the world >= 2 path without a builder turn.  SURVEY section 8(e)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, n_pairs, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IMX_MULTI_PAIRS=str(n_pairs))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multi_gpu_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("MULTI-GPU OK")]
    assert len(ok) == 1 and f"world={world} pairs={n_pairs}" in ok[0], r.stdout[-1500:]
    print("[multi-gpu]", ok[0])


def test_worker_at_world_1_over_rccl():
    """The worker's own control flow (process group with a device id, RCCL gather forced at world 1, a communicator built from a
    broadcast unique id, imx_gather_records, byte comparison with the single batch) on whatever box this is."""
    _launch(1, 4, 29541)


@pytest.mark.parametrize("n_pairs", [8, 7])
def test_two_ranks_over_rccl_equal_the_single_gpu_batch(n_pairs):
    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs >= 2 GPUs (this box has {torch.cuda.device_count()}): the world-1 form of the same worker ran instead")
    _launch(2, n_pairs, 29543 + n_pairs)


def test_every_gpu_of_the_node_over_rccl():
    n = torch.cuda.device_count()
    if n < 3:
        pytest.skip(f"needs >= 3 GPUs (this box has {n})")
    _launch(n, 4 * n, 29561)

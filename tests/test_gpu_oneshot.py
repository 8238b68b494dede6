"""The one-shot exchange at the cut (rn_set_exchange_transport(ctx, 1)): the kernel that produces a rank's
partial children sums writes them as {payload, sequence tag} packets into an inbox on every peer, the crown kernel of every
rank adds the contributions in rank order -- no collective launch between the chain walks and the crown.

Functionally verified on ONE GPU: (i) several shard contexts of one process share an address space, so their inboxes are wired
directly (rn_debug_peer_inbox_connect_local) and the device-resident batches run with the one-shot transport on 2-4 ranks --
bitwise the iterates of the stand-in transport, and the oracle's at 1e-9; (ii) two PROCESSES on one GPU exchange
hipIpcMemHandle_t handles over gloo and map each other's inboxes (what `bench.py --gpus N` does on a multi-GPU node);
(iii) a rank whose peer never writes gives up after the time-out with RN_E_COMM.  Each case runs in a process of its own
(tests/oneshot_worker.py): the ranks' streams must sit on different hardware queues (GPU_MAX_HW_QUEUES)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "oneshot_worker.py")


def run(args, env_extra=None, timeout=600, launcher=None):
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", RAPIDNET_ONESHOT_TIMEOUT_MS="4000", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra or {})
    cmd = (launcher or [sys.executable]) + [WORKER] + [str(a) for a in args]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, "rc %d\n%s\n%s" % (p.returncode, p.stdout.decode()[-2000:], p.stderr.decode()[-4000:])
    return p.stdout.decode()


@pytest.mark.parametrize("args", [("medium", 2, 0), ("medium", 4, 1), ("medium", 3, 2, "structured"), ("ragged", 3, 1), ("small", 2, 3),
                                  ("medium", 2, 0, "f32"), ("medium", 4, 0, "trip"), ("medium", 3, 0, "wrap")])
def test_one_shot_exchange_in_process(args):
    out = run(("inprocess",) + args)
    assert "oneshot inprocess ok" in out, out
    if "trip" in args:          # the soft-constraint thresholds trip: the optimistic batch is replayed through the exact path, one-shot too
        assert "'replayed': 1" in out or "'replayed': 2" in out, out


@pytest.mark.parametrize("args", [("medium", 2, 0), ("ragged", 3, 1)])
def test_exchange_chooses_itself(args):
    """RN_EXCHANGE_AUTO: both transports timed on the context's own iterations, one decision for all ranks, no trace in the iterates"""
    out = run(("auto",) + args)
    assert "oneshot auto ok" in out, out


def test_one_shot_exchange_gives_up_instead_of_hanging():
    out = run(("timeout",), env_extra={"RAPIDNET_ONESHOT_TIMEOUT_MS": "300", "RAPIDNET_GROUP_TIMEOUT_S": "3"})
    assert "oneshot timeout ok" in out, out


def test_a_time_out_on_some_ranks_fails_the_batch_on_every_rank():
    out = run(("latecomer",), env_extra={"RAPIDNET_ONESHOT_TIMEOUT_MS": "400", "RAPIDNET_GROUP_TIMEOUT_S": "30"})
    assert "oneshot latecomer ok" in out, out


@pytest.mark.parametrize("name,cut", [("medium", 0), ("ragged", 1)])
def test_one_shot_exchange_between_two_processes(name, cut):
    """hipIpcMemHandle_t inboxes between two processes sharing the box's one GPU"""
    import socket

    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    out = run(("ipc", name, cut), launcher=launcher, timeout=900)
    assert "oneshot ipc ok" in out, out

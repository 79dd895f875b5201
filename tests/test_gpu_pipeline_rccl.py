"""The RCCL path of the layer pipeline on REAL devices: two ranks, one process per GPU, residual by
ncclSend/ncclRecv -- each stream must reproduce the single-GPU greedy tokens exactly (fp32 copies are
exact).  Needs two visible GPUs; on a one-GPU box the test skips (the same code is exercised there by
the single-process loopback self-test and, on CPU, by tests/test_pipeline_gloo.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def test_two_rank_rccl_pipeline_matches_single_gpu(hip):
    if hip.q3_device_count() < 2:
        pytest.skip("needs 2 GPUs (one process per GPU)")
    path = os.path.join(Q.tmp_dir(), "4Bmini.bin")
    Q.synth("4Bmini", path)
    n = 24
    m = hip.q3_model_open(path.encode(), 256, 0)
    want = (C.c_int * n)()
    assert hip.q3_generate_greedy(m, 11, 0, n, want) == n
    hip.q3_model_close(m)
    idfile = os.path.join(Q.tmp_dir(), f"rccl_test_id_{os.getpid()}")
    if os.path.exists(idfile):
        os.remove(idfile)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "pipeline_rank.py"), path, str(n), idfile],
                                      env=env, stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), [p.returncode for p in procs]
    streams = json.loads(outs[1].strip().splitlines()[-1])["streams"]
    assert streams[0] == list(want) and streams[1] == list(want)
    os.remove(idfile)


def test_one_rank_rccl_self_exchange_matches_plain_loop(hip):
    """What a one-GPU box CAN run of the RCCL transport: a ONE-rank communicator whose rank sends every tick's message
    (residual + token slot) to itself (Q3_PIPE_SELF=1) -- ncclCommInitRank, grouped ncclSend / ncclRecv on the launch
    stream between hipGraph replays, the token travelling back through the message instead of the on-device shortcut.
    The tokens must be the plain greedy loop's."""
    path = os.path.join(Q.tmp_dir(), "4Bmini.bin")
    Q.synth("4Bmini", path)
    n = 40
    m = hip.q3_model_open(path.encode(), 256, 0)
    want = (C.c_int * n)()
    assert hip.q3_generate_greedy(m, 11, 0, n, want) == n
    hip.q3_model_close(m)
    idfile = os.path.join(Q.tmp_dir(), f"rccl_self_id_{os.getpid()}")
    if os.path.exists(idfile):
        os.remove(idfile)
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", Q3_PIPE_SELF="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "pipeline_rank.py"), path, str(n), idfile],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    streams = json.loads(p.stdout.strip().splitlines()[-1])["streams"]
    assert streams[0] == list(want)
    os.remove(idfile)


def test_bench_refuses_more_ranks_than_gpus(hip):
    """`python bench.py --gpus N` without a launcher starts N ranks itself -- and says so loudly when the box
    has fewer than N devices instead of measuring one GPU and printing n_gpus: 1."""
    ndev = hip.q3_device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ndev + 1), "--steps", "2", "--warmup", "1",
                        "--model", "4Bmini"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "HIP device(s) visible" in (r.stderr + r.stdout)
    assert "n_gpus" not in r.stdout

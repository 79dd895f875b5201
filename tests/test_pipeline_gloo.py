"""The N > 1 path on CPU: two processes over gloo run the layer pipeline with the same
stage split (q3_pipeline_layers) and tick schedule (q3_pipeline_schedule) the GPU code
uses, the oracle standing in for the kernels and dist.send/recv for the RCCL hand-offs.
Every stream must reproduce the single-process greedy tokens."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import q3lib as Q

HERE = os.path.dirname(os.path.abspath(__file__))


def _stage_worker(rank, world, path, nsteps, port, q):
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=True)      # a wedged rank must not hang the suite
    import torch
    import torch.distributed as dist
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")     # the container's hostname may not resolve
    dist.init_process_group("gloo", rank=rank, world_size=world)
    host, orc = Q.host_lib(), Q.oracle_lib()   # plain-C host side only: torch's wheel carries its own HIP runtime
    hip = host
    orc.orc_set_mode(Q.ORC_TREE)
    ms = [host.q3_model_open(path.encode(), 0, 1) for _ in range(world)]   # one KV cache per stream
    p = ms[0].contents.params
    first, count = C.c_int(), C.c_int()
    hip.q3_pipeline_layers(C.byref(p), rank, world, C.byref(first), C.byref(count))
    dim, V = p.dim, p.vocab_size
    tokens = {s: [] for s in range(world)}
    # one message per tick around the ring, as in the GPU code (ring_exchange): x[dim] + token slot
    msg_in = torch.zeros(dim + 1, dtype=torch.float32)
    msg_out = torch.zeros(dim + 1, dtype=torch.float32)
    T = nsteps * world + world - 1
    for tick in range(T):
        s, k = C.c_int(), C.c_int()
        if hip.q3_pipeline_schedule(rank, world, nsteps, tick, C.byref(s), C.byref(k)):
            m = ms[s.value]
            pos = k.value
            if rank == 0:
                tok = 7 if k.value == 0 else int(msg_in[dim].item())
                fe = np.ctypeslib.as_array(m.contents.weights.fe, (V * dim,))
                x = fe[tok * dim:(tok + 1) * dim].copy()
            else:
                x = msg_in[:dim].numpy().copy()
            for layer in range(first.value, first.value + count.value):
                y = np.zeros(dim, np.float32)
                orc.orc_layer_step(m, layer, pos, Q.fptr(x), Q.fptr(y))
                x = y
            if rank < world - 1:
                msg_out[:dim] = torch.from_numpy(x)
            else:
                w = m.contents.weights
                xn = np.zeros(dim, np.float32)
                orc.orc_rmsnorm(Q.fptr(xn), Q.fptr(x), w.out_rms_norm, dim)
                xq = np.zeros(dim, np.int8); xs = np.zeros(dim // 64, np.float32); qt = Q.q8view(xq, xs)
                orc.orc_q8_quantize(C.byref(qt), Q.fptr(xn), dim, 64)
                logits = np.zeros(V, np.float32)
                orc.orc_matmul(Q.fptr(logits), C.byref(qt), w.cls, dim, V, 64)
                tok = int(logits.argmax())
                tokens[s.value].append(tok)
                msg_out[dim] = float(tok)
        if tick < T - 1:
            reqs = [dist.isend(msg_out.clone(), dst=(rank + 1) % world),
                    dist.irecv(msg_in, src=(rank + world - 1) % world)]
            for r in reqs:
                r.wait()
    dist.barrier()
    if rank == world - 1:
        q.put(tokens)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipeline_over_gloo_matches_single_process(world):
    """world = 2: the two-stage case; world = 3: one layer per stage on the 3-layer model, the token returning from
    rank 2 to rank 0 across two idle hops of the ring"""
    import torch.multiprocessing as mp
    host, orc = Q.host_lib(), Q.oracle_lib()
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    nsteps = 6
    m = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_mode(Q.ORC_TREE); orc.orc_set_threads(1)
    want, tok = [], 7
    for pos in range(nsteps):
        tok = int(Q.logits_array(m, orc.orc_forward(m, tok, pos)).argmax())
        want.append(tok)
    host.q3_model_close(m)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() * 7 + world) % 300
    procs = [ctx.Process(target=_stage_worker, args=(r, world, path, nsteps, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for s in range(world):
        assert got[s] == want, f"stream {s}"


def test_stage_split_is_contiguous_and_complete():
    hip = Q.host_lib()
    for L in (3, 28, 36):
        p = Q.ModelParams(0, 0, 64, 64, L, 1, 1, 64, 64, 64, 1, 64)
        for world in (1, 2, 4, 8):
            nxt = 0
            for r in range(world):
                f, c = C.c_int(), C.c_int()
                hip.q3_pipeline_layers(C.byref(p), r, world, C.byref(f), C.byref(c))
                assert f.value == nxt and (c.value >= 1 or L < world)
                nxt += c.value
            assert nxt == L


def test_stage_split_weighs_the_classifier():
    """Qwen3-4B shapes: the classifier is worth ~1.6 layers of time, so the last stage gets fewer layers and the
    slowest stage (which sets the tick) is no slower than with the even split."""
    hip = Q.host_lib()
    spec = Q.SynthSpec()
    assert hip.q3_synth_preset(b"4B", C.byref(spec)) == 0
    p = Q.ModelParams()
    for k in ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len", "head_dim"):
        setattr(p, k, getattr(spec, k))
    cls = 1.62
    for world, want in ((2, [19, 17]), (4, [10, 9, 9, 8]), (8, [5, 5, 5, 5, 5, 4, 4, 3])):
        counts = []
        for r in range(world):
            f, c = C.c_int(), C.c_int()
            hip.q3_pipeline_layers(C.byref(p), r, world, C.byref(f), C.byref(c))
            assert f.value == sum(counts)
            counts.append(c.value)
        assert counts == want, counts
        even = [36 // world + (1 if r < 36 % world else 0) for r in range(world)]
        assert max(counts[:-1] + [counts[-1] + cls]) <= max(even[:-1] + [even[-1] + cls])

"""End-to-end and layer-level parity of forward() on the GPU (SURVEY.md 8(c') tiers B-D)."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def open_pair(hip, host, name, seq=0):
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    Q.synth(name, path)
    mg = hip.q3_model_open(path.encode(), seq, 0)
    mo = host.q3_model_open(path.encode(), seq, 1)
    assert mg and mo
    return mg, mo


def run_stream(step, n, first=7):
    tok, toks, logits = first, [], []
    for pos in range(n):
        lg = step(tok, pos)
        logits.append(lg)
        tok = int(lg.argmax())
        toks.append(tok)
    return toks, logits


@pytest.mark.parametrize("name,steps", [("tiny", 64), ("small", 140)])
def test_forward_matches_tree_oracle_bit_exact(hip, host, orc, name, steps):
    mg, mo = open_pair(hip, host, name)
    orc.orc_set_mode(Q.ORC_TREE)
    tg, lg = run_stream(lambda t, p: Q.logits_array(mg, hip.forward(mg, t, p)), steps)
    to, lo = run_stream(lambda t, p: Q.logits_array(mo, orc.orc_forward(mo, t, p)), steps)
    assert tg == to
    for pos in range(steps):
        assert np.array_equal(lg[pos], lo[pos]), f"pos {pos}"
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_forward_vs_reference_order_tiny(hip, host, orc):
    """Tier C: on the tiny fixture no int8 code flips, so the GPU agrees with the
    reference's own order to 1e-5 of max|logit| and picks the same greedy tokens."""
    mg, mo = open_pair(hip, host, "tiny")
    orc.orc_set_mode(Q.ORC_REF)
    tg, lg = run_stream(lambda t, p: Q.logits_array(mg, hip.forward(mg, t, p)), 64)
    to, lo = run_stream(lambda t, p: Q.logits_array(mo, orc.orc_forward(mo, t, p)), 64)
    assert tg == to
    worst = max(float(np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(lg, lo))
    assert worst <= 1e-5, worst
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_graph_and_eager_agree_and_rewind(hip, host):
    """hipGraph replay == plain launches; and callers may rewind pos to 0 without
    clearing the cache (reference completion.c:281-284)."""
    mg, _mo = open_pair(hip, host, "small")
    a_t, a_l = run_stream(lambda t, p: Q.logits_array(mg, hip.forward(mg, t, p)), 80)
    b_t, b_l = run_stream(lambda t, p: Q.logits_array(mg, hip.forward(mg, t, p)), 80)   # rewound
    assert a_t == b_t and all(np.array_equal(x, y) for x, y in zip(a_l, b_l))
    hip.q3_tap_enable(mg, 1)     # the tap forces plain launches
    c_t, c_l = run_stream(lambda t, p: Q.logits_array(mg, hip.forward(mg, t, p)), 80)
    hip.q3_tap_enable(mg, 0)
    assert a_t == c_t and all(np.array_equal(x, y) for x, y in zip(a_l, c_l))
    hip.q3_model_close(mg); host.q3_model_close(_mo)


def test_layer_taps_and_teacher_forced_layers(hip, host, orc):
    mg, mo = open_pair(hip, host, "small")
    p = mg.contents.params
    L, dim = p.n_layers, p.dim
    orc.orc_set_mode(Q.ORC_TREE)
    tap = np.zeros(L * dim, np.float32)
    orc.orc_set_tap(Q.fptr(tap))
    hip.q3_tap_enable(mg, 1)
    tok = 11
    for pos in range(70):
        lo = Q.logits_array(mo, orc.orc_forward(mo, tok, pos))
        lg = Q.logits_array(mg, hip.forward(mg, tok, pos))
        gt = np.ctypeslib.as_array(hip.q3_tap_data(mg), shape=(L * dim,))
        assert np.array_equal(gt, tap), f"pos {pos}"
        assert np.array_equal(lg, lo)
        tok = int(lo.argmax())
    orc.orc_set_tap(None)
    hip.q3_tap_enable(mg, 0)
    # tier B: one layer from the oracle's (reference-order) residual
    orc.orc_set_mode(Q.ORC_REF)
    rng = np.random.default_rng(3)
    for layer in range(L):
        x = rng.standard_normal(dim).astype(np.float32)
        xo = np.zeros(dim, np.float32); xg = np.zeros(dim, np.float32)
        orc.orc_layer_step(mo, layer, 70, Q.fptr(x), Q.fptr(xo))
        hip.q3_layer_step(mg, layer, 70, Q.fptr(x), Q.fptr(xg))
        assert float(np.abs(xg - xo).max() / np.abs(xo).max()) <= 1e-3
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_device_argmax_and_greedy_loop(hip, host):
    mg, _mo = open_pair(hip, host, "small")
    toks, _ = run_stream(lambda t, p: Q.logits_array(mg, hip.forward(mg, t, p)), 20)
    out = (C.c_int * 20)()
    assert hip.q3_generate_greedy(mg, 7, 0, 20, out) == 20
    assert list(out) == toks
    hip.q3_model_close(mg); host.q3_model_close(_mo)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_pipeline_selftest_matches_single_gpu(hip, host, world):
    """The layer pipeline (stage split, per-stream KV caches, tick schedule, token feedback)
    run as `world` stages on this one GPU, with device copies in place of the RCCL
    hand-offs: every stream must reproduce the single-GPU greedy tokens exactly."""
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    mg = hip.q3_model_open(path.encode(), 0, 0)
    nsteps = 70            # crosses the 64-position chunk boundary
    want = (C.c_int * nsteps)()
    assert hip.q3_generate_greedy(mg, 7, 0, nsteps, want) == nsteps
    hip.q3_model_close(mg)
    got = (C.c_int * (world * nsteps))()
    assert hip.q3_pipeline_selftest(path.encode(), 0, world, 7, 0, nsteps, got) == 0
    for s in range(world):
        assert list(got[s * nsteps:(s + 1) * nsteps]) == list(want), f"stream {s}"


@pytest.mark.parametrize("world,streams", [(4, 1), (3, 2)])
def test_pipeline_selftest_with_fewer_streams_than_stages(hip, world, streams):
    """q3_pipeline_run_streams' schedule (bench.py's single-stream figure at N > 1): only the first `streams`
    streams run, the others' ticks stay idle, the stale messages of idle ticks are never consumed."""
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    mg = hip.q3_model_open(path.encode(), 0, 0)
    nsteps = 20
    want = (C.c_int * nsteps)()
    assert hip.q3_generate_greedy(mg, 7, 0, nsteps, want) == nsteps
    hip.q3_model_close(mg)
    got = (C.c_int * (streams * nsteps))()
    assert hip.q3_pipeline_selftest_streams(path.encode(), 0, world, streams, 7, 0, nsteps, got) == 0
    for s in range(streams):
        assert list(got[s * nsteps:(s + 1) * nsteps]) == list(want), f"stream {s}"


def test_pipeline_schedule_covers_every_token_once(hip):
    for world in (1, 2, 4, 8):
        nsteps = 5
        seen = {}
        for tick in range(nsteps * world + world - 1):
            for r in range(world):
                s, k = C.c_int(), C.c_int()
                if hip.q3_pipeline_schedule(r, world, nsteps, tick, C.byref(s), C.byref(k)):
                    seen.setdefault((s.value, k.value), []).append((tick, r))
        assert len(seen) == world * nsteps
        for (s, k), visits in seen.items():
            # each token of each stream visits ranks 0..world-1 on consecutive ticks
            assert [r for _, r in visits] == list(range(world))
            assert [t for t, _ in visits] == list(range(visits[0][0], visits[0][0] + world))
            assert visits[0][0] == s + k * world


def test_model_created_by_the_reference_loader_is_accepted(hip, host, orc):
    """forward() on a Model that the REFERENCE's model_create() built (oracle/_ref, compiled from
    /root/reference): same struct layout, weights read through its mmap views."""
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref not available")
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    mr = ref.model_create(path.encode(), 0)
    mo = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    feed = np.random.default_rng(4).integers(0, 1024, size=10)
    for pos, tok in enumerate(feed):
        a = Q.logits_array(mr, hip.forward(mr, int(tok), pos))
        b = Q.logits_array(mo, orc.orc_forward(mo, int(tok), pos))
        assert np.array_equal(a, b)
    hip.q3_device_detach(mr)
    ref.model_free(mr)
    host.q3_model_close(mo)


# (dim, hidden_dim, n_heads, n_kv_heads): the layer shapes of the Qwen3 family, 2 layers, small vocabulary.
# Each picks a different launch plan in q3_gemv.hip (rows per wave, preparing waves, the generic
# kernel for hidden sizes beyond 16384), so each is checked bit for bit.
FAMILY = {"0.6B": (1024, 3072, 16, 8), "1.7B": (2048, 6144, 16, 8), "4B": (2560, 9728, 32, 8),
          "8B": (4096, 12288, 32, 8), "14B": (5120, 17408, 40, 8), "32B": (5120, 25600, 64, 8)}


@pytest.mark.parametrize("family", sorted(FAMILY))
def test_family_layer_shapes_bit_exact(hip, host, orc, family):
    dim, hid, heads, kv = FAMILY[family]
    path = os.path.join(Q.tmp_dir(), f"shape_{family}.bin")
    Q.synth("4Bmini", path, dim=dim, hidden_dim=hid, n_heads=heads, n_kv_heads=kv, n_layers=2,
            vocab_size=2048, seq_len=256)
    mg = hip.q3_model_open(path.encode(), 0, 0)
    mo = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(8)
    feed = np.random.default_rng(3).integers(0, 2048, size=8)
    for pos in range(8):
        lg = Q.logits_array(mg, hip.forward(mg, int(feed[pos]), pos))
        lo = Q.logits_array(mo, orc.orc_forward(mo, int(feed[pos]), pos))
        assert np.array_equal(lg, lo), (family, pos)
    # the same positions again through the batched prompt path (int8-MFMA GEMM) on a fresh model
    mp = hip.q3_model_open(path.encode(), 0, 0)
    arr = (C.c_int * 8)(*[int(t) for t in feed])
    lp = Q.logits_array(mp, hip.q3_prefill(mp, arr, 8, 0))
    assert np.array_equal(lp, lg), family
    hip.q3_model_close(mp)
    hip.q3_model_close(mg)
    host.q3_model_close(mo)


def test_back_to_back_async_steps_do_not_share_the_ctl_slot(hip, host):
    """q3_forward_device() returns without a sync; several of them queued back to back must each run with
    their own {token, pos} (round-1 ADVICE: the shared pinned slot was overwritten before the first copy ran)."""
    mg, _mo = open_pair(hip, host, "small")
    ma, _mb = open_pair(hip, host, "small")
    feed = [int(t) for t in np.random.default_rng(12).integers(0, 1024, size=24)]
    want = None
    for pos, tok in enumerate(feed):
        want = Q.logits_array(mg, hip.forward(mg, tok, pos))
    for pos, tok in enumerate(feed):
        hip.q3_forward_device(ma, tok, pos)           # no sync in between
    hip.q3_logits_fetch(ma)
    got = Q.logits_array(ma)
    assert np.array_equal(got, want)
    assert hip.q3_device_argmax(ma) == int(want.argmax())
    for m in (mg, ma):
        hip.q3_model_close(m)
    host.q3_model_close(_mo); host.q3_model_close(_mb)


def test_model_freed_by_the_reference_and_recreated(hip, host, orc):
    """The real drop-in order of calls: model_create -> forward -> model_free (which knows nothing of the
    device state) -> model_create -> forward.  The registry must neither touch the freed Model nor run the
    new one on the old one's weights and KV cache."""
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref not available")
    orc.orc_set_mode(Q.ORC_TREE)
    feed = [int(t) for t in np.random.default_rng(14).integers(0, 512, size=6)]
    for round_, (name, vocab) in enumerate([("small", 1024), ("tiny", 512), ("small", 1024)]):
        path = os.path.join(Q.tmp_dir(), f"{name}.bin")
        Q.synth(name, path)
        mr = ref.model_create(path.encode(), 0)
        mo = host.q3_model_open(path.encode(), 0, 1)
        for pos, tok in enumerate(feed):
            a = Q.logits_array(mr, hip.forward(mr, tok % vocab, pos))
            b = Q.logits_array(mo, orc.orc_forward(mo, tok % vocab, pos))
            assert np.array_equal(a, b), (round_, pos)
        ref.model_free(mr)                 # no q3_device_detach: the reference's own teardown order
        host.q3_model_close(mo)


def test_exported_attention_has_the_reference_semantics(hip, host, orc):
    """attention(Model*, layer, pos) (reference src/forward.c:141-195): q and the host KV cache in, head
    outputs to x_rms_norm, nothing else touched."""
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    mg = host.q3_model_open(path.encode(), 0, 1)      # host state present; never attached to the device
    mo = host.q3_model_open(path.encode(), 0, 1)
    p = mg.contents.params
    P, KVD, seq = p.n_heads * p.head_dim, p.n_kv_heads * p.head_dim, p.seq_len
    rng = np.random.default_rng(21)
    layer, pos = 1, 70
    for m in (mg, mo):
        st = m.contents.state
        np.ctypeslib.as_array(st.q, shape=(P,))[:] = 0
        np.ctypeslib.as_array(st.k_cache, shape=(p.n_layers, seq, KVD))[:] = 0
        np.ctypeslib.as_array(st.v_cache, shape=(p.n_layers, seq, KVD))[:] = 0
    q = rng.standard_normal(P).astype(np.float32)
    k = rng.standard_normal((pos + 1, KVD)).astype(np.float32)
    v = rng.standard_normal((pos + 1, KVD)).astype(np.float32)
    for m in (mg, mo):
        st = m.contents.state
        np.ctypeslib.as_array(st.q, shape=(P,))[:] = q
        np.ctypeslib.as_array(st.k_cache, shape=(p.n_layers, seq, KVD))[layer, :pos + 1] = k
        np.ctypeslib.as_array(st.v_cache, shape=(p.n_layers, seq, KVD))[layer, :pos + 1] = v
    hip.attention(mg, layer, pos)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_attention(mo, layer, pos)
    got = np.ctypeslib.as_array(mg.contents.state.x_rms_norm, shape=(P,)).copy()
    tree = np.ctypeslib.as_array(mo.contents.state.x_rms_norm, shape=(P,)).copy()
    assert np.array_equal(got, tree)
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_attention(mo, layer, pos)
    refo = np.ctypeslib.as_array(mo.contents.state.x_rms_norm, shape=(P,)).copy()
    assert float(np.abs(got - refo).max() / np.abs(refo).max()) <= 2e-6
    # the cache and q are inputs only
    assert np.array_equal(np.ctypeslib.as_array(mg.contents.state.k_cache, shape=(p.n_layers, seq, KVD))[layer, :pos + 1], k)
    assert np.array_equal(np.ctypeslib.as_array(mg.contents.state.q, shape=(P,)), q)
    host.q3_model_close(mg); host.q3_model_close(mo)


@pytest.mark.parametrize("name,T", [("small", 100), ("4Bmini", 700), ("4Bmini", 1500)])
def test_decode_over_a_pseudo_random_cache_vs_oracle(hip, host, orc, name, T):
    """q3_kv_fill_random and its oracle twin put the same rows into both caches (device layout
    [kv head][position] vs the reference's [position][kv head]); three decode steps on top are bit-identical
    to the tree oracle.  The small-model counterpart of test_gpu_fullsize.py::test_4b_long_context_vs_oracle."""
    seq = T + 16
    path = os.path.join(Q.tmp_dir(), f"{name}_w2048.bin")
    Q.synth(name, path, seq_len=2048)           # the preset's weights with a longer window in the header
    mg = hip.q3_model_open(path.encode(), seq, 0)
    mo = host.q3_model_open(path.encode(), seq, 1)
    hip.q3_kv_fill_random(mg, T, 5)
    orc.orc_kv_fill_random(mo, T, 5)
    orc.orc_set_mode(Q.ORC_TREE)
    tok = 3
    for k in range(3):
        a = Q.logits_array(mg, hip.forward(mg, tok, T + k))
        b = Q.logits_array(mo, orc.orc_forward(mo, tok, T + k))
        assert np.array_equal(a, b), (name, T, k)
        tok = int(a.argmax())
    hip.q3_model_close(mg); host.q3_model_close(mo)


@pytest.mark.parametrize("name", ["tiny", "small", "4Bmini"])
def test_fused_attention_wo_launch_equals_separate_launches(hip, name):
    """k_attn_wo (attention + Wo in one launch, the attention output handed over as tagged granules) against the
    attention launch followed by the Wo GEMV (Q3_FUSE=0, read at attach): logits bit-identical at every position
    through the four one-chunk launch shapes (rows_cap 16/32/48/64) and into the in-launch-merge shape, and again
    after a rewind of pos (same position twice in a row: the step counter, not the position, tags the hand-off)."""
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    spec = Q.synth(name, path)
    n = min(150, spec.seq_len - 2)
    os.environ["Q3_FUSE"] = "0"
    try:
        ma = hip.q3_model_open(path.encode(), 0, 0)
        assert hip.q3_device_attach(ma) == 0
    finally:
        del os.environ["Q3_FUSE"]
    mb = hip.q3_model_open(path.encode(), 0, 0)
    assert hip.q3_device_attach(mb) == 0
    tok = 3
    for pos in list(range(n)) + [7, 7, 70 if n > 71 else 5, 0]:
        a = Q.logits_array(ma, hip.forward(ma, tok, pos)).copy()
        b = Q.logits_array(mb, hip.forward(mb, tok, pos))
        assert np.array_equal(a, b), (name, pos)
        tok = int(a.argmax())
    hip.q3_model_close(ma); hip.q3_model_close(mb)


def _nan_checkpoint(name):
    """A synthetic checkpoint whose final RMSNorm weights are NaN: every logit of every step is NaN."""
    import shutil, struct
    src = os.path.join(Q.tmp_dir(), f"{name}.bin")
    spec = Q.synth(name, src)
    dst = os.path.join(Q.tmp_dir(), f"{name}_nan_out_norm.bin")
    shutil.copyfile(src, dst)
    # layout (q3_model.c): 256-byte header, att_rms_norm[L][dim], ffn_rms_norm[L][dim], out_rms_norm[dim], ...
    off = 256 + 2 * spec.n_layers * spec.dim * 4
    with open(dst, "r+b") as f:
        f.seek(off)
        f.write(struct.pack("<f", float("nan")) * spec.dim)
    return dst, spec


@pytest.mark.parametrize("fp16", [False, True])
def test_nan_logits_never_become_an_out_of_range_token_on_the_device(hip, fp16):
    """Round-3 GPU fault, regression: all-NaN logits made the device argmax return its INT_MAX sentinel, and the next
    step's k_begin fetched embedding row INT_MAX * dim.  The argmax now yields a valid id, and k_begin bounds whatever
    token id device memory hands it."""
    import ctypes as C
    path, spec = _nan_checkpoint("small")
    m = hip.q3_model_open(path.encode(), 0, 0)
    if fp16:
        assert hip.q3_device_attach_fp16(m) == 0
    lg = Q.logits_array(m, hip.forward(m, 5, 0))
    assert np.isnan(lg).all()
    assert 0 <= hip.q3_argmax(hip.forward(m, 5, 0), spec.vocab_size) < spec.vocab_size          # host argmax
    hip.q3_forward_device(m, 7, 1)
    t = hip.q3_device_argmax(m)                                                                  # device argmax
    assert 0 <= t < spec.vocab_size, t
    out = (C.c_int * 3)()
    assert hip.q3_generate_greedy(m, 9, 2, 3, out) == 3                                          # the loop feeds its own picks back
    assert all(0 <= out[i] < spec.vocab_size for i in range(3)), list(out)
    seed = C.c_uint64(1)
    out2 = (C.c_int * 3)()
    hip.q3_generate_sampled(m, 9, 5, 3, C.c_float(0.8), C.c_float(0.9), C.byref(seed), out2)   # NaN probabilities: any valid id
    assert all(0 <= out2[i] < spec.vocab_size for i in range(3)), list(out2)
    hip.q3_model_close(m)

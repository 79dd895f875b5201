"""The oracle itself (CPU, no GPU):
  * ORC_REF against golden vectors captured from the REAL reference (tests/golden/, made by
    make_golden.py from oracle/_ref = /root/reference compiled -O2 -DNDEBUG, 1 thread);
  * ORC_REF against that reference build directly, bit for bit, where it is present;
  * ORC_TREE (the order the GPU uses) against ORC_REF within the tolerances of SURVEY.md 8(c')."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

HERE = os.path.dirname(os.path.abspath(__file__))
os.environ.setdefault("OMP_NUM_THREADS", "1")


def golden(name):
    return np.load(os.path.join(HERE, "golden", name))


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def forced(lib_forward, m, feed):
    return np.stack([Q.logits_array(m, lib_forward(m, int(t), pos)) for pos, t in enumerate(feed)])


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_ref_order_reproduces_reference_logits_bit_for_bit(host, orc, name):
    g = golden(f"{name}_ref_logits.npz")
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    Q.synth(name, path)
    m = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_set_threads(1)
    got = forced(orc.orc_forward, m, g["feed"])
    keep = g["positions"] if "positions" in g.files else np.arange(len(g["feed"]))
    assert np.array_equal(got[keep], g["logits"])
    host.q3_model_close(m)


def test_ref_order_reproduces_reference_ops_bit_for_bit(orc):
    g = golden("ops_ref.npz")
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_set_threads(1)
    n, d = g["mm_x"].shape[0], g["mm_out"].shape[0]
    xq = np.zeros(n, np.int8); xs = np.zeros(n // 64, np.float32); t = Q.q8view(xq, xs)
    orc.orc_q8_quantize(C.byref(t), Q.fptr(g["mm_x"].copy()), n, 64)
    assert np.array_equal(xq, g["mm_xq"]) and np.array_equal(xs, g["mm_xs"])
    wq, ws = g["mm_wq"].copy(), g["mm_ws"].copy()
    wt = Q.q8view(wq.reshape(-1), ws.reshape(-1))
    out = np.zeros(d, np.float32)
    orc.orc_matmul(Q.fptr(out), C.byref(t), C.byref(wt), n, d, 64)
    assert np.array_equal(out, g["mm_out"])
    o = np.zeros(n, np.float32)
    orc.orc_rmsnorm(Q.fptr(o), Q.fptr(g["mm_x"].copy()), Q.fptr(g["rms_w"].copy()), n)
    assert np.array_equal(o, g["rms_out"])
    z = g["sm_in"].copy(); orc.orc_softmax(Q.fptr(z), len(z))
    assert np.array_equal(z, g["sm_out"])
    for pos in (0, 3, 4097):
        r = g["rope_in"].copy(); orc.orc_rotary(Q.fptr(r), 128, pos)
        assert np.array_equal(r, g[f"rope_{pos}"])
    s = g["sw_g"].copy(); orc.orc_swiglu(Q.fptr(s), Q.fptr(g["sw_u"].copy()), len(s))
    assert np.array_equal(s, g["sw_out"])


@pytest.mark.parametrize("name,steps", [("tiny", 40), ("small", 100)])
def test_ref_order_equals_reference_build(host, orc, name, steps):
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref is not built and /root/reference is absent")
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    spec = Q.synth(name, path)
    feed = np.random.default_rng(1).integers(0, spec.vocab_size, size=steps)
    mr = ref.model_create(path.encode(), 0)
    mo = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_set_threads(1)
    a = forced(ref.forward, mr, feed)
    b = forced(orc.orc_forward, mo, feed)
    assert np.array_equal(a, b)
    ref.model_free(mr)
    host.q3_model_close(mo)


def test_tree_order_ops_within_1e6_of_reference_order(orc):
    rng = np.random.default_rng(11)
    orc.orc_set_threads(1)
    for n, d in ((128, 64), (2560, 96), (9728, 40)):
        wq = rng.integers(-127, 128, size=(d, n), dtype=np.int8)
        ws = (0.02 / 73.3 * rng.uniform(0.75, 1.25, size=(d, n // 64))).astype(np.float32)
        xq = rng.integers(-127, 128, size=n, dtype=np.int8)
        xs = rng.uniform(0.5, 1.5, size=n // 64).astype(np.float32)
        xt, wt = Q.q8view(xq, xs), Q.q8view(wq.reshape(-1), ws.reshape(-1))
        a = np.zeros(d, np.float32); b = np.zeros(d, np.float32)
        orc.orc_set_mode(Q.ORC_REF); orc.orc_matmul(Q.fptr(a), C.byref(xt), C.byref(wt), n, d, 64)
        orc.orc_set_mode(Q.ORC_TREE); orc.orc_matmul(Q.fptr(b), C.byref(xt), C.byref(wt), n, d, 64)
        assert rel(b, a) <= 1e-6
        x = rng.standard_normal(n).astype(np.float32); w = rng.standard_normal(n).astype(np.float32)
        a = np.zeros(n, np.float32); b = np.zeros(n, np.float32)
        orc.orc_set_mode(Q.ORC_REF); orc.orc_rmsnorm(Q.fptr(a), Q.fptr(x), Q.fptr(w), n)
        orc.orc_set_mode(Q.ORC_TREE); orc.orc_rmsnorm(Q.fptr(b), Q.fptr(x), Q.fptr(w), n)
        assert rel(b, a) <= 1e-6
    for T in (1, 3, 64, 65, 300):
        H, KV, hd = 4, 1, 128
        q = rng.standard_normal((H, hd)).astype(np.float32)
        k = rng.standard_normal((T, KV, hd)).astype(np.float32)
        v = rng.standard_normal((T, KV, hd)).astype(np.float32)
        a = np.zeros((H, hd), np.float32); b = np.zeros((H, hd), np.float32)
        orc.orc_set_mode(Q.ORC_REF); orc.orc_attention_raw(Q.fptr(q), Q.fptr(k), Q.fptr(v), T, H, KV, hd, Q.fptr(a))
        orc.orc_set_mode(Q.ORC_TREE); orc.orc_attention_raw(Q.fptr(q), Q.fptr(k), Q.fptr(v), T, H, KV, hd, Q.fptr(b))
        assert rel(b, a) <= 2e-6
    # q3_expf vs libm over the working range: < 2 ulp
    xs = np.linspace(-86, 88.7, 100001).astype(np.float32)
    orc.orc_set_mode(Q.ORC_TREE)
    mine = np.array([orc.orc_expf(float(t)) for t in xs[::7]], np.float64)
    true = np.exp(xs[::7].astype(np.float64))
    assert np.max(np.abs(mine - true) / true) < 2.5e-7
    assert orc.orc_expf(0.0) == 1.0 and orc.orc_expf(-100.0) == 0.0 and np.isinf(orc.orc_expf(89.0))


def test_tree_order_end_to_end_on_tiny(host, orc):
    """Tier C: on the tiny fixture no activation code flips, so the two orders agree to
    1e-5 of max|logit| and choose the same tokens."""
    path = os.path.join(Q.tmp_dir(), "tiny.bin")
    Q.synth("tiny", path)
    feed = np.random.default_rng(2).integers(0, 512, size=48)
    ma = host.q3_model_open(path.encode(), 0, 1)
    mb = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_threads(1)
    orc.orc_set_mode(Q.ORC_REF); a = forced(orc.orc_forward, ma, feed)
    orc.orc_set_mode(Q.ORC_TREE); b = forced(orc.orc_forward, mb, feed)
    assert rel(b, a) <= 1e-5
    assert np.array_equal(a.argmax(1), b.argmax(1))
    host.q3_model_close(ma); host.q3_model_close(mb)


def test_oracle_threads_do_not_change_results(host, orc):
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    feed = np.random.default_rng(3).integers(0, 1024, size=12)
    outs = []
    for threads in (1, 4):
        m = host.q3_model_open(path.encode(), 0, 1)
        orc.orc_set_mode(Q.ORC_TREE); orc.orc_set_threads(threads)
        outs.append(forced(orc.orc_forward, m, feed))
        host.q3_model_close(m)
    orc.orc_set_threads(1)
    assert np.array_equal(outs[0], outs[1])


def sampler_logits(rng, vocab, kind):
    """logits of three shapes: peaked (a real model), flat (random-init), with exact ties"""
    if kind == "peaked":
        x = rng.standard_normal(vocab).astype(np.float32) * 2.0
        x[rng.integers(0, vocab, size=5)] += np.float32(9.0)
    elif kind == "flat":
        x = (rng.standard_normal(vocab) * 0.05).astype(np.float32)
    else:
        x = rng.integers(-3, 4, size=vocab).astype(np.float32)      # many equal probabilities
    return x


@pytest.mark.parametrize("vocab", [512, 5000, 151936])
def test_sampler_restatement_equals_reference_sample(orc, vocab):
    """ORC_REF orc_sample() == the reference's sample() (src/sampler.c:189-201): same token, same
    probabilities left in the logits buffer, same RNG state, over runs of draws."""
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref is not built and /root/reference is absent")
    orc.orc_set_mode(Q.ORC_REF)
    rng = np.random.default_rng(vocab)
    for kind in ("peaked", "flat", "ties"):
        for temperature, top_p in ((1.0, 0.9), (0.7, 0.95), (1.3, 1.0), (0.0, 0.5), (1.0, 0.0)):
            seed = int(rng.integers(1, 2**63))
            s = ref.sampler_create(vocab, temperature, top_p, seed)
            t = np.array([temperature], np.float32); pp = np.array([top_p], np.float32)
            orc.orc_sampler_clamp(Q.fptr(t), Q.fptr(pp))
            assert t[0] == s.contents.temperature and pp[0] == s.contents.top_p
            state = C.c_uint64(seed)
            for _ in range(3 if vocab > 10000 else 8):
                x = sampler_logits(rng, vocab, kind)
                a, b = x.copy(), x.copy()
                ta = ref.sample(s, Q.fptr(a))
                tb = orc.orc_sample(Q.fptr(b), vocab, float(t[0]), float(pp[0]), C.byref(state))
                assert ta == tb
                assert np.array_equal(a, b, equal_nan=True)
                assert s.contents.seed == state.value
            ref.sampler_free(s)


def test_kv_fill_twin_follows_the_documented_generator():
    """orc_kv_fill_random (the oracle's twin of the product's q3_kv_fill_random, used by the long-context
    parity tests) against a numpy restatement of the same splitmix64 recipe, in the reference's cache layout
    [layer][seq_len][n_kv][head_dim] (src/model.c:360-361); rows beyond T stay untouched."""
    host, orc = Q.host_lib(), Q.oracle_lib()
    path = os.path.join(Q.tmp_dir(), "small.bin")
    Q.synth("small", path)
    m = host.q3_model_open(path.encode(), 0, 1)
    p = m.contents.params
    T, seed = 37, 99
    kvd = p.n_kv_heads * p.head_dim
    n = p.n_layers * p.seq_len * kvd
    kc = np.ctypeslib.as_array(m.contents.state.k_cache, (n,)).reshape(p.n_layers, p.seq_len, p.n_kv_heads, p.head_dim)
    vc = np.ctypeslib.as_array(m.contents.state.v_cache, (n,)).reshape(p.n_layers, p.seq_len, p.n_kv_heads, p.head_dim)
    orc.orc_kv_fill_random(m, T, seed)

    def gen(sd):
        with np.errstate(over="ignore"):
            i = np.arange(T * p.head_dim, dtype=np.uint64)
            z = np.uint64(sd) + i * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z ^= z >> np.uint64(31)
        u = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        return ((u - np.float32(0.5)) * np.float32(2.0)).reshape(T, p.head_dim)

    for l in range(p.n_layers):
        for g in range(p.n_kv_heads):
            assert np.array_equal(kc[l, :T, g, :], gen(seed + 2 * (l * 64 + g)))
            assert np.array_equal(vc[l, :T, g, :], gen(seed + 2 * (l * 64 + g) + 1))
    assert not kc[:, T:].any() and not vc[:, T:].any()
    assert np.abs(kc[:, :T]).max() <= 1.0 and abs(float(kc[:, :T].mean())) < 0.05
    host.q3_model_close(m)

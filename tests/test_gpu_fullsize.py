"""Full-size parity (SURVEY.md 8(c') tiers C' and D) on the BASELINE.json shapes.

* 4B (headline config): GPU logits bit-identical to the oracle's tree order, run-to-run
  determinism, and the KV rewind semantics, at full width (2560 / 9728 / 151936).
* 0.6B (config 1, the reference's own CPU case): GPU vs the REAL reference build
  (oracle/_ref, 1 thread = golden), reported next to the reference's own self-noise
  (golden vs the same code at 16 threads): the reference does not reproduce itself to
  1e-3 at this size, so the bar is max(1e-3, 1.5 x self-noise), and every
  argmax disagreement must sit where the golden's top-2 gap is inside that noise."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def test_4b_logits_bit_exact_vs_tree_oracle(hip, host, orc):
    path = os.path.join(Q.tmp_dir(), "4B.bin")
    Q.synth("4B", path)
    mg = hip.q3_model_open(path.encode(), 128, 0)
    mo = host.q3_model_open(path.encode(), 128, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(16)
    feed = np.random.default_rng(8).integers(0, 151936, size=5)
    first = None
    for pos, tok in enumerate(feed):
        a = Q.logits_array(mg, hip.forward(mg, int(tok), pos))
        b = Q.logits_array(mo, orc.orc_forward(mo, int(tok), pos))
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), f"pos {pos}: max diff {np.abs(a - b).max()}"
        if pos == 0:
            first = a
    # rewind without clearing the cache (reference completion.c:281-284) reproduces position 0
    again = Q.logits_array(mg, hip.forward(mg, int(feed[0]), 0))
    assert np.array_equal(again, first)
    orc.orc_set_threads(1)
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_06b_against_the_reference_build_and_its_self_noise(hip):
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref not available")
    gomp = C.CDLL("libgomp.so.1")
    path = os.path.join(Q.tmp_dir(), "0.6B.bin")
    Q.synth("0.6B", path)
    feed = np.random.default_rng(9).integers(0, 151936, size=10)

    def run_ref(threads):
        gomp.omp_set_num_threads(threads)
        m = ref.model_create(path.encode(), 64)
        rows = [Q.logits_array(m, ref.forward(m, int(t), p)) for p, t in enumerate(feed)]
        ref.model_free(m)
        gomp.omp_set_num_threads(1)
        return np.stack(rows)

    golden = run_ref(1)
    noisy = run_ref(16)
    mg = hip.q3_model_open(path.encode(), 64, 0)
    gpu = np.stack([Q.logits_array(mg, hip.forward(mg, int(t), p)) for p, t in enumerate(feed)])
    hip.q3_model_close(mg)
    scale = np.abs(golden).max(axis=1)
    self_noise = float((np.abs(noisy - golden).max(axis=1) / scale).max())
    gpu_err = float((np.abs(gpu - golden).max(axis=1) / scale).max())
    print(f"0.6B: GPU vs golden {gpu_err:.3e}; reference 16 threads vs golden {self_noise:.3e}")
    Q.record_parity("0.6B_vs_reference_build", {"positions": len(feed), "gpu_vs_golden_rel": gpu_err,
                                               "reference_16_threads_vs_golden_rel": self_noise,
                                               "argmax_disagreements": int(sum(gpu[p].argmax() != golden[p].argmax() for p in range(len(feed))))})
    assert gpu_err <= max(1e-3, 1.5 * self_noise)
    for p in range(len(feed)):
        if gpu[p].argmax() != golden[p].argmax():
            top2 = np.sort(golden[p])[-2:]
            assert (top2[1] - top2[0]) <= 2 * max(self_noise, gpu_err) * scale[p]


def _tier_c_prime(hip, name, positions, seq=64):
    """GPU vs the REAL reference build (1 thread = golden) next to the reference's own
    self-noise (same code, 16 threads)."""
    ref = Q.reference_lib()
    if ref is None:
        pytest.skip("oracle/_ref not available")
    gomp = C.CDLL("libgomp.so.1")
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    Q.synth(name, path)
    feed = np.random.default_rng(19).integers(0, 151936, size=positions)

    def run_ref(threads):
        gomp.omp_set_num_threads(threads)
        m = ref.model_create(path.encode(), seq)
        rows = [Q.logits_array(m, ref.forward(m, int(t), p)) for p, t in enumerate(feed)]
        ref.model_free(m)
        gomp.omp_set_num_threads(1)
        return np.stack(rows)

    golden = run_ref(1)
    noisy = run_ref(16)
    mg = hip.q3_model_open(path.encode(), seq, 0)
    gpu = np.stack([Q.logits_array(mg, hip.forward(mg, int(t), p)) for p, t in enumerate(feed)])
    hip.q3_model_close(mg)
    scale = np.abs(golden).max(axis=1)
    self_noise = float((np.abs(noisy - golden).max(axis=1) / scale).max())
    gpu_err = float((np.abs(gpu - golden).max(axis=1) / scale).max())
    flips = int(sum(gpu[p].argmax() != golden[p].argmax() for p in range(positions)))
    Q.record_parity(f"{name}_vs_reference_build", {"positions": positions, "gpu_vs_golden_rel": gpu_err,
                                                  "reference_16_threads_vs_golden_rel": self_noise,
                                                  "argmax_disagreements": flips})
    assert gpu_err <= max(1e-3, 1.5 * self_noise)
    for p in range(positions):
        if gpu[p].argmax() != golden[p].argmax():
            top2 = np.sort(golden[p])[-2:]
            assert (top2[1] - top2[0]) <= 2 * max(self_noise, gpu_err) * scale[p]


def test_4b_against_the_reference_build_and_its_self_noise(hip):
    """tier C' on the headline configuration (round-1 VERDICT: only 0.6B had met the real reference)"""
    _tier_c_prime(hip, "4B", 6)


@pytest.mark.parametrize("name", ["1.7B", "8B"])
def test_full_size_configs_bit_exact_vs_tree_oracle(hip, host, orc, name):
    """BASELINE configs 2 (1.7B) and 4 (8B, untied classifier) at full width: the looping classifier
    kernel k_gemv2 at n = 2048 (NJ = 2) and n = 4096 (NJ = 4) over d = 151,936 rows, and every layer
    kernel of those shapes, against the oracle's tree order, bit for bit."""
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    Q.synth(name, path)
    mg = hip.q3_model_open(path.encode(), 64, 0)
    mo = host.q3_model_open(path.encode(), 64, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(16)
    feed = np.random.default_rng(23).integers(0, 151936, size=4)
    worst = 0.0
    for pos, tok in enumerate(feed):
        a = Q.logits_array(mg, hip.forward(mg, int(tok), pos))
        b = Q.logits_array(mo, orc.orc_forward(mo, int(tok), pos))
        assert np.isfinite(a).all()
        worst = max(worst, float(np.abs(a - b).max()))
        assert np.array_equal(a, b), f"{name} pos {pos}: max diff {np.abs(a - b).max()}"
    orc.orc_set_threads(1)
    Q.record_parity(f"{name}_vs_tree_oracle", {"positions": len(feed), "max_abs_diff": worst, "bit_exact": True})
    # the batched prompt pass at these widths (GEMM slabs of n = 2048 / 4096 and 6144 / 12288, both tile widths):
    # 40 tokens through q3_prefill on a second handle == 40 calls of forward() on the first
    mp = hip.q3_model_open(path.encode(), 64, 0)
    prompt = np.random.default_rng(29).integers(0, 151936, size=40).astype(np.int32)
    arr = (C.c_int * 40)(*[int(t) for t in prompt])
    lp = Q.logits_array(mp, hip.q3_prefill(mp, arr, 40, 0))
    for pos in range(40):
        lf = Q.logits_array(mg, hip.forward(mg, int(prompt[pos]), pos))
    assert np.array_equal(lp, lf), f"{name}: prompt pass differs from token-by-token decode"
    Q.record_parity(f"prefill_{name}_40_tokens", {"bit_identical_to_token_by_token": True})
    hip.q3_model_close(mp)
    hip.q3_model_close(mg); host.q3_model_close(mo)


@pytest.mark.parametrize("n", [2048, 4096])
def test_classifier_loop_kernel_at_full_vocabulary(hip, orc, n):
    """k_gemv2 (the looping GEMV) with d = 151,936 at the 1.7B / 8B widths, op level: fused rmsnorm +
    quantise feeding the exported matmul path, vs orc_matmul in tree order (bit-exact) and reference
    order (1e-6)."""
    d = 151936
    rng = np.random.default_rng(n)
    wq = rng.integers(-127, 128, size=(d, n), dtype=np.int8)
    ws = (rng.random((d, n // 64), dtype=np.float32) * 0.5 + 0.75) * np.float32(0.02 / 73.3)
    xq = rng.integers(-127, 128, size=n, dtype=np.int8)
    xs = (rng.random(n // 64, dtype=np.float32) + 0.5) * np.float32(1.0 / 127.0)
    out = np.zeros(d, np.float32)
    hip.q3_op_gemv(Q.i8ptr(wq), Q.fptr(ws), Q.i8ptr(xq), Q.fptr(xs), n, d, Q.fptr(out))
    xt, wt = Q.q8view(xq, xs), Q.q8view(wq.reshape(-1), ws.reshape(-1))
    tree = np.zeros(d, np.float32); refo = np.zeros(d, np.float32)
    orc.orc_set_threads(16)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_matmul(Q.fptr(tree), C.byref(xt), C.byref(wt), n, d, 64)
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_matmul(Q.fptr(refo), C.byref(xt), C.byref(wt), n, d, 64)
    orc.orc_set_threads(1)
    rel = float(np.abs(out - refo).max() / np.abs(refo).max())
    Q.record_parity(f"classifier_gemv_n{n}_d{d}", {"bit_exact_vs_tree": bool(np.array_equal(out, tree)), "rel_vs_reference_order": rel})
    assert np.array_equal(out, tree)
    assert rel <= 1e-6


def test_4b_prompt_pass_equals_token_by_token(hip):
    """full-size 4B: a 200-token prompt through q3_prefill (LDS-staged int8-MFMA GEMMs at every layer width,
    block attention over 1-4 chunks) leaves the logits and the next decode steps bit-identical to 200 calls
    of forward() -- which tier D ties to the tree oracle"""
    path = os.path.join(Q.tmp_dir(), "4B.bin")
    Q.synth("4B", path)
    ma = hip.q3_model_open(path.encode(), 512, 0)
    mb = hip.q3_model_open(path.encode(), 512, 0)
    n = 200
    prompt = np.random.default_rng(21).integers(0, 151936, size=n).astype(np.int32)
    arr = (C.c_int * n)(*[int(t) for t in prompt])
    la = Q.logits_array(ma, hip.q3_prefill(ma, arr, n, 0))
    for pos in range(n):
        lb = Q.logits_array(mb, hip.forward(mb, int(prompt[pos]), pos))
    assert np.array_equal(la, lb)
    tok = int(la.argmax())
    for pos in range(n, n + 4):
        la = Q.logits_array(ma, hip.forward(ma, tok, pos))
        lb = Q.logits_array(mb, hip.forward(mb, tok, pos))
        assert np.array_equal(la, lb), pos
        tok = int(la.argmax())
    Q.record_parity("prefill_4B_200_tokens", {"bit_identical_to_token_by_token": True, "decode_steps_after": 4})
    hip.q3_model_close(ma)
    hip.q3_model_close(mb)


@pytest.mark.parametrize("name,world", [("8B", 2), ("8B", 4), ("8B", 8), ("4B", 8)])
def test_full_size_pipeline_selftest_matches_single_gpu(hip, name, world):
    """BASELINE config 4 on its own shapes: the layer pipeline on full-size DeepSeek-R1-0528-Qwen3-8B-shaped
    weights (untied classifier on the last stage only, embedding on the first) as 2 / 4 / 8 stages -- the 8-stage
    split is 5/5/5/5/5/4/4/3 -- and on tied 4B as 8 stages (embedding matrix on the first AND the last stage).
    All stages live in this one process (q3_pipeline_selftest: the stage split, per-stream KV caches, tick
    schedule, token feedback and graphs are the code the RCCL run executes; the hand-offs are device copies),
    and every one of the `world` streams must reproduce the single-GPU greedy tokens."""
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    Q.synth(name, path)
    nsteps = 12
    mg = hip.q3_model_open(path.encode(), 128, 0)
    p = mg.contents.params
    counts = []
    for r in range(world):
        first, count = C.c_int(), C.c_int()
        hip.q3_pipeline_layers(C.byref(p), r, world, C.byref(first), C.byref(count))
        assert first.value == sum(counts)
        counts.append(count.value)
    assert sum(counts) == p.n_layers and min(counts) >= 1
    if (name, world) == ("8B", 8):
        assert counts == [5, 5, 5, 5, 5, 4, 4, 3]
    want = (C.c_int * nsteps)()
    assert hip.q3_generate_greedy(mg, 9707, 0, nsteps, want) == nsteps
    hip.q3_model_close(mg)
    got = (C.c_int * (world * nsteps))()
    assert hip.q3_pipeline_selftest(path.encode(), 128, world, 9707, 0, nsteps, got) == 0
    for s in range(world):
        assert list(got[s * nsteps:(s + 1) * nsteps]) == list(want), f"{name} world {world} stream {s}"
    Q.record_parity(f"pipeline_selftest_{name}_world{world}", {"layers_per_stage": counts, "tokens_per_stream": nsteps,
                                                              "streams_equal_single_gpu": True})


@pytest.mark.parametrize("T", [512, 4096])
def test_4b_long_context_vs_oracle(hip, host, orc, T):
    """BASELINE config 3 at full size against the oracle (round-2 VERDICT: beyond position 4 the full-size 4B
    had met the oracle only at op level and on a 2-layer model).  Both sides fill rows 0..T-1 of every layer's
    K/V cache with the same pseudo-random values (q3_kv_fill_random on the device, its twin orc_kv_fill_random
    in the reference's cache layout), then run two teacher-forced steps at positions T and T+1 through all 36
    layers: logits bit-identical to the tree-order oracle.  Launch shapes: T = 512 -> in-launch merge of the
    chunk partials, T = 4096 -> wide merge.  Then, on the last layer's own q and cache (reference
    src/forward.c:141-195), the exported attention() against the REFERENCE-order restatement within
    max(2e-6, 2e-9 * T) of max|out| -- the bar the op-level tests use (the slack is the reference's own
    sequential fp32 sum over T terms)."""
    path = os.path.join(Q.tmp_dir(), "4B.bin")
    Q.synth("4B", path)
    seq = T + 64
    mg = hip.q3_model_open(path.encode(), seq, 0)
    mo = host.q3_model_open(path.encode(), seq, 1)
    orc.orc_set_threads(16)
    hip.q3_kv_fill_random(mg, T, 99)
    orc.orc_kv_fill_random(mo, T, 99)
    orc.orc_set_mode(Q.ORC_TREE)
    feed = np.random.default_rng(31 + T).integers(0, 151936, size=2)
    for k, tok in enumerate(feed):
        a = Q.logits_array(mg, hip.forward(mg, int(tok), T + k))
        b = Q.logits_array(mo, orc.orc_forward(mo, int(tok), T + k))
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), f"T={T} step {k}: max diff {np.abs(a - b).max()}"
    # op level, reference order: the last layer's attention at position T+1 on the oracle's host state
    p = mo.contents.params
    P = p.n_heads * p.head_dim
    layer, pos = p.n_layers - 1, T + 1
    out = np.ctypeslib.as_array(mo.contents.state.x_rms_norm, (P,))
    orc.orc_set_mode(Q.ORC_REF)
    orc.orc_attention(mo, layer, pos)
    ref = out.copy()
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_attention(mo, layer, pos)
    tree = out.copy()
    hip.attention(mo, layer, pos)
    got = out.copy()
    orc.orc_set_threads(1)
    rel = float(np.abs(got - ref).max() / np.abs(ref).max())
    Q.record_parity(f"4B_at_{T}_cached_positions", {
        "steps": 2, "logits_bit_exact_vs_tree_oracle": True,
        "attention_last_layer_bit_exact_vs_tree": bool(np.array_equal(got, tree)),
        "attention_last_layer_rel_vs_reference_order": rel, "bar": max(2e-6, 2e-9 * (T + 2))})
    assert np.array_equal(got, tree)
    assert rel <= max(2e-6, 2e-9 * (T + 2))
    hip.q3_model_close(mg); host.q3_model_close(mo)

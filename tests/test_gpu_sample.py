"""Device-side temperature / top-p sampling against the oracle's restatement of the
reference's sample() (src/sampler.c:189-201), which tests/test_oracle.py pins to the reference build."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q
from test_oracle import sampler_logits

pytestmark = pytest.mark.gpu

PARAMS = ((1.0, 0.9), (0.7, 0.95), (1.3, 1.0), (0.0, 0.5), (1.0, 0.0), (2.0, 0.3))


@pytest.mark.parametrize("vocab", [512, 4096, 5000, 151936])
def test_sample_op_bit_exact_vs_tree_oracle(hip, orc, vocab):
    """token, probabilities left in the buffer and RNG state: identical to orc_sample in tree mode;
    against the reference-order restatement the probabilities agree to 4e-6 of the largest and the token almost always."""
    rng = np.random.default_rng(vocab + 1)
    same_as_ref, total = {}, {}
    for kind in ("peaked", "flat", "ties"):
        for temperature, top_p in PARAMS:
            seed = int(rng.integers(1, 2**63))
            sg, st, sr = C.c_uint64(seed), C.c_uint64(seed), C.c_uint64(seed)
            t = np.array([temperature], np.float32); pp = np.array([top_p], np.float32)
            orc.orc_sampler_clamp(Q.fptr(t), Q.fptr(pp))
            for _ in range(2 if vocab > 10000 else 4):
                x = sampler_logits(rng, vocab, kind)
                g, a, r = x.copy(), x.copy(), x.copy()
                tg = hip.q3_op_sample(Q.fptr(g), vocab, temperature, top_p, C.byref(sg))
                orc.orc_set_mode(Q.ORC_TREE)
                ta = orc.orc_sample(Q.fptr(a), vocab, float(t[0]), float(pp[0]), C.byref(st))
                orc.orc_set_mode(Q.ORC_REF)
                tr = orc.orc_sample(Q.fptr(r), vocab, float(t[0]), float(pp[0]), C.byref(sr))
                assert np.array_equal(g, a), (kind, temperature, top_p)
                assert tg == ta, (kind, temperature, top_p)
                assert sg.value == st.value == sr.value
                # q3_expf vs libm, and the reference's sequential sum over `vocab` terms, which drops addends below half an ulp of
                # the running sum once a large probability has been added (measured: 2e-5 at 5000 terms)
                assert np.abs(g - r).max() <= max(4e-6, 1e-8 * vocab) * max(r.max(), 1e-30) + 1e-12
                same_as_ref[kind] = same_as_ref.get(kind, 0) + int(tg == tr)
                total[kind] = total.get(kind, 0) + 1
    # Tree-order and reference-order softmax differ in the last bits (see the tolerance above), which moves
    # a draw only when the coin lands within that distance of a boundary of the cumulative distribution:
    # rare for the peaked distribution of a trained model, common when 150k near-equal probabilities put a
    # boundary every 7e-6 (there the reference does not reproduce itself across summation orders either).
    assert same_as_ref["peaked"] >= 0.9 * total["peaked"], (same_as_ref, total)


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_sampled_decode_matches_oracle_stream(hip, host, orc, name):
    """forward on the device + q3_device_sample, step by step, == orc_forward + orc_sample (tree mode);
    the on-device loop q3_generate_sampled reproduces the same stream and RNG state."""
    path = os.path.join(Q.tmp_dir(), f"{name}.bin")
    spec = Q.synth(name, path)
    V = spec.vocab_size
    mg = hip.q3_model_open(path.encode(), 0, 0)
    mo = host.q3_model_open(path.encode(), 0, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    temperature, top_p, seed, steps, first = 0.9, 0.92, 20250101, 40, 5
    sg, so = C.c_uint64(seed), C.c_uint64(seed)
    tok_g, tok_o, stream_g, stream_o = first, first, [], []
    for pos in range(steps):
        hip.q3_forward_device(mg, tok_g, pos)
        tok_g = hip.q3_device_sample(mg, temperature, top_p, C.byref(sg))
        lo = Q.logits_array(mo, orc.orc_forward(mo, tok_o, pos)).copy()
        tok_o = orc.orc_sample(Q.fptr(lo), V, temperature, top_p, C.byref(so))
        stream_g.append(tok_g); stream_o.append(tok_o)
    assert stream_g == stream_o
    assert sg.value == so.value
    assert len(set(stream_g)) > 3          # it is sampling, not repeating one token
    out = (C.c_int * steps)()
    sl = C.c_uint64(seed)
    n = hip.q3_generate_sampled(mg, first, 0, steps, temperature, top_p, C.byref(sl), out)
    assert n == steps and list(out) == stream_g and sl.value == sg.value
    hip.q3_model_close(mg)
    host.q3_model_close(mo)


def reference_loop(orc, mo, ids, V, seq, temperature, top_p, seed, stops, max_new):
    """completion()'s token loop (src/completion.c:57-84) on the oracle, with the caller's cap on new tokens"""
    state = C.c_uint64(seed)
    out, token = [], int(ids[0])
    for pos in range(seq):
        lo = Q.logits_array(mo, orc.orc_forward(mo, token, pos)).copy()
        if pos + 1 < len(ids):
            nxt = int(ids[pos + 1])
        else:
            if len(out) >= max_new:
                break
            nxt = orc.orc_sample(Q.fptr(lo), V, temperature, top_p, C.byref(state))
        if nxt in stops:
            break
        if pos + 1 >= len(ids):
            out.append(nxt)
        token = nxt
    return out, state.value


@pytest.mark.parametrize("case", ["cap", "stop", "window", "stop_in_prompt"])
def test_complete_equals_the_reference_token_loop(hip, host, orc, case):
    path = os.path.join(Q.tmp_dir(), "tiny.bin")
    spec = Q.synth("tiny", path)
    V, seq = spec.vocab_size, spec.seq_len
    orc.orc_set_mode(Q.ORC_TREE)
    temperature, top_p, seed = 0.9, 0.95, 777
    prompt = [5, 17, 300, 44, 9]
    stops, max_new = (-1, -1), 20
    if case == "window":
        max_new = 10 * seq                      # the context window ends the loop
    mo = host.q3_model_open(path.encode(), 0, 1)
    if case == "stop":                          # a token the stream is known to produce becomes the stop id
        dry, _ = reference_loop(orc, mo, prompt, V, seq, temperature, top_p, seed, stops, 40)
        stops = (dry[7], -1)
        max_new = 40
    if case == "stop_in_prompt":
        stops = (-1, prompt[3])
    want, want_seed = reference_loop(orc, mo, prompt, V, seq, temperature, top_p, seed, stops, max_new)
    mg = hip.q3_model_open(path.encode(), 0, 0)
    out = (C.c_int * (seq + 8))()
    sg = C.c_uint64(seed)
    arr = (C.c_int * len(prompt))(*prompt)
    n = hip.q3_complete(mg, arr, len(prompt), temperature, top_p, C.byref(sg), stops[0], stops[1], out, min(max_new, seq + 8))
    assert list(out[:n]) == want
    assert sg.value == want_seed
    if case == "stop":
        assert n == 7
    if case == "stop_in_prompt":
        assert n == 0 and sg.value == seed
    hip.q3_model_close(mg)
    host.q3_model_close(mo)

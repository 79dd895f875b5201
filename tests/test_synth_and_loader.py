"""Host side: the synthetic checkpoint writer, the `.bin` reader and the byte accounting."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import q3lib as Q

HERE = os.path.dirname(os.path.abspath(__file__))


def test_synthetic_files_are_reproducible_and_match_committed_checksums(host):
    sums = json.load(open(os.path.join(HERE, "golden", "checksums.json")))
    for name, want in sums.items():
        path = os.path.join(Q.tmp_dir(), f"chk_{name}.bin")
        if os.path.exists(path):
            os.remove(path)
        Q.synth(name, path)
        assert os.path.getsize(path) == want["bytes"]
        assert "%016x" % host.q3_file_checksum(path.encode()) == want["fnv1a64"]
        os.remove(path)
    # the survey's tiny fixture is 386,816 bytes
    assert sums["tiny"]["bytes"] == 386816


@pytest.mark.parametrize("name,size", [("0.6B", 0.633), ("1.7B", 1.828), ("4B", 4.274), ("8B", 8.704)])
def test_preset_sizes_match_the_survey(host, name, size):
    spec = Q.SynthSpec()
    assert host.q3_synth_preset(name.encode(), C.byref(spec)) == 0
    assert abs(host.q3_synth_bytes(C.byref(spec)) / 1e9 - size) < 0.002


def test_bytes_per_token_matches_the_baseline_table(host):
    """4B 4.275 / 4.426 / 5.483 GB at T = 1 / 512 / 4096, 0.6B 0.634, 8B 8.043."""
    def bpt(name, T):
        spec = Q.SynthSpec()
        host.q3_synth_preset(name.encode(), C.byref(spec))
        p = Q.ModelParams(0, 0, spec.dim, spec.hidden_dim, spec.n_layers, spec.n_heads, spec.n_kv_heads,
                          spec.vocab_size, spec.seq_len, spec.head_dim, spec.shared_classifier, 64)
        return host.q3_bytes_per_token(C.byref(p), T) / 1e9
    assert abs(bpt("4B", 1) - 4.275) < 0.001
    assert abs(bpt("4B", 512) - 4.426) < 0.001
    assert abs(bpt("4B", 4096) - 5.483) < 0.001
    assert abs(bpt("0.6B", 1) - 0.634) < 0.001
    assert abs(bpt("1.7B", 1) - 1.829) < 0.001
    assert abs(bpt("8B", 1) - 8.043) < 0.001


def test_loader_views_match_a_direct_parse(host):
    path = os.path.join(Q.tmp_dir(), "small.bin")
    spec = Q.synth("small", path)
    raw = np.fromfile(path, dtype=np.uint8)
    hdr = raw[:48].view(np.int32)
    assert hdr[0] == 0x7177656E and hdr[1] == 1 and hdr[11] == 64
    assert not raw[48:256].any()
    m = host.q3_model_open(path.encode(), 100, 1)
    p = m.contents.params
    assert (p.dim, p.hidden_dim, p.n_layers, p.seq_len) == (spec.dim, spec.hidden_dim, spec.n_layers, 100)
    L, dim, hd, hid, V = p.n_layers, p.dim, p.head_dim, p.hidden_dim, p.vocab_size
    P, KVD = p.n_heads * hd, p.n_kv_heads * hd
    off = 256
    w = m.contents.weights

    def take_f32(n):
        nonlocal off
        a = raw[off:off + 4 * n].view(np.float32); off += 4 * n
        return a

    def take_q8(n):
        nonlocal off
        q = raw[off:off + n].view(np.int8); off += n
        s = raw[off:off + 4 * (n // 64)].view(np.float32); off += 4 * (n // 64)
        return q, s

    assert np.array_equal(np.ctypeslib.as_array(w.att_rms_norm, (L * dim,)), take_f32(L * dim))
    assert np.array_equal(np.ctypeslib.as_array(w.ffn_rms_norm, (L * dim,)), take_f32(L * dim))
    assert np.array_equal(np.ctypeslib.as_array(w.out_rms_norm, (dim,)), take_f32(dim))
    assert np.array_equal(np.ctypeslib.as_array(w.q_rms_norm, (L * hd,)), take_f32(L * hd))
    assert np.array_equal(np.ctypeslib.as_array(w.k_rms_norm, (L * hd,)), take_f32(L * hd))
    for arr, count, numel in ((w.qe, 1, V * dim), (w.wq, L, P * dim), (w.wk, L, KVD * dim), (w.wv, L, KVD * dim),
                              (w.wo, L, dim * P), (w.w1, L, hid * dim), (w.w2, L, dim * hid), (w.w3, L, hid * dim),
                              (w.cls, 1, V * dim)):
        for i in range(count):
            q, s = take_q8(numel)
            assert np.array_equal(np.ctypeslib.as_array(arr[i].q, (numel,)), q)
            assert np.array_equal(np.ctypeslib.as_array(arr[i].s, (numel // 64,)), s)
    assert off == len(raw)
    # fp32 embedding copy = q*s (reference q8_dequantize, src/q8.c:32-36)
    fe = np.ctypeslib.as_array(w.fe, (V * dim,))
    qe = np.ctypeslib.as_array(w.qe[0].q, (V * dim,)).astype(np.float32)
    se = np.repeat(np.ctypeslib.as_array(w.qe[0].s, (V * dim // 64,)), 64)
    assert np.array_equal(fe, qe * se)
    host.q3_model_close(m)


def test_loader_rejects_bad_files(host):
    bad = os.path.join(Q.tmp_dir(), "bad.bin")
    open(bad, "wb").write(b"\0" * 300)
    assert not host.q3_model_open(bad.encode(), 0, 0)
    good = os.path.join(Q.tmp_dir(), "tiny.bin")
    Q.synth("tiny", good)
    data = open(good, "rb").read()
    open(bad, "wb").write(data[:-100])            # truncated
    assert not host.q3_model_open(bad.encode(), 0, 0)
    assert not host.q3_model_open(b"/nonexistent/x.bin", 0, 0)
    os.remove(bad)


def test_argmax_is_first_maximum(host):
    a = np.array([1, 5, 5, 2], np.float32)
    assert host.q3_argmax(Q.fptr(a), 4) == 1


def test_host_argmax_first_maximum_with_ties_and_nans():
    """q3_argmax (SSE2 / AVX2 picked at run time) == the scalar walk `if (x[i] > best)` from the left: first of
    equal maxima, NaNs never win, lengths around the vector widths and the full vocabulary."""
    host = Q.host_lib()
    rng = np.random.default_rng(5)
    for n in (1, 2, 31, 32, 33, 63, 64, 65, 127, 1000, 151936):
        for trial in range(12):
            y = rng.standard_normal(n).astype(np.float32)
            if trial % 3 == 0 and n > 3:
                y[rng.integers(0, n, 3)] = y.max()
            if trial % 4 == 1 and n > 5:
                y[rng.integers(1, n)] = np.nan
            want, best = 0, y[0]
            for i in range(1, n):
                if y[i] > best:
                    best, want = y[i], i
            assert host.q3_argmax(Q.fptr(y), n) == want, (n, trial)

"""Soak (tools/soak.py in test form): graph replay == eager launches over 2,300 positions -- all three attention
launch shapes, the in-launch ticket merge and the wide merge launch -- and a 1,500-token prompt through
q3_prefill == token by token.  Size-independent property: every logit bit-identical."""
import ctypes as C
import os

import numpy as np
import pytest

import q3lib as Q

pytestmark = pytest.mark.gpu


def test_graph_eager_and_prefill_agree_over_long_runs(hip):
    path = os.path.join(Q.tmp_dir(), "soak.bin")
    Q.synth("4Bmini", path, seq_len=4096, vocab_size=4096)
    a = hip.q3_model_open(path.encode(), 0, 0)       # graph replay
    b = hip.q3_model_open(path.encode(), 0, 0)       # eager: layer taps force plain launches
    hip.q3_tap_enable(b, 1)
    tok = 11
    for pos in range(2300):
        la = Q.logits_array(a, hip.forward(a, tok, pos))
        lb = Q.logits_array(b, hip.forward(b, tok, pos))
        assert np.array_equal(la, lb), pos
        tok = int(la.argmax()) if pos % 7 else int((pos * 2654435761) % 4096)
    hip.q3_model_close(b)
    prompt = np.random.default_rng(1).integers(0, 4096, size=1500).astype(np.int32)
    arr = (C.c_int * len(prompt))(*[int(t) for t in prompt])
    c = hip.q3_model_open(path.encode(), 0, 0)
    lc = Q.logits_array(c, hip.q3_prefill(c, arr, len(prompt), 0))
    for pos, t in enumerate(prompt):                 # model `a` is reused from position 0
        la = hip.forward(a, int(t), pos)
    assert np.array_equal(lc, Q.logits_array(a, la))
    hip.q3_model_close(a)
    hip.q3_model_close(c)


def test_decode_meets_the_oracle_past_1100_positions(hip, host, orc):
    """End to end against the ORACLE (not the GPU against itself) through all three attention launch
    shapes: one chunk, in-launch merge (64..1023) and the wide merge launch (>= 1024 positions), on the
    Qwen3-4B layer shapes: logits bit-identical to the tree-order oracle at every compared position."""
    import numpy as np
    path = os.path.join(Q.tmp_dir(), "4Bmini_long.bin")
    Q.synth("4Bmini", path, seq_len=1280)
    mg = hip.q3_model_open(path.encode(), 1280, 0)
    mo = host.q3_model_open(path.encode(), 1280, 1)
    orc.orc_set_mode(Q.ORC_TREE)
    orc.orc_set_threads(16)
    n = 1180
    feed = np.random.default_rng(31).integers(0, 8192, size=n)
    compared = 0
    for pos, tok in enumerate(feed):
        lg = hip.forward(mg, int(tok), pos)
        lo = orc.orc_forward(mo, int(tok), pos)
        if pos % 16 == 0 or pos in (63, 64, 65, 1023, 1024, 1025) or pos >= n - 8:
            a = Q.logits_array(mg, lg); b = Q.logits_array(mo, lo)
            assert np.array_equal(a, b), f"pos {pos}: max diff {np.abs(a - b).max()}"
            compared += 1
    orc.orc_set_threads(1)
    Q.record_parity("4Bmini_decode_vs_tree_oracle_long", {"positions": n, "compared": compared, "bit_exact": True,
                                                          "attention_shapes": ["single", "merge", "long"]})
    hip.q3_model_close(mg); host.q3_model_close(mo)


def test_in_launch_hand_off_under_foreign_load(hip):
    """The attention -> Wo hand-off inside k_attn_wo / k_merge_wo (tagged granules, bounded polls) while ANOTHER
    model's launches -- prompt passes on its own stream -- compete for the CUs and the memory system: the consumer
    workgroups may start long before their attention workgroups get a CU, polls queue behind foreign traffic, the
    chip is unevenly loaded.  Three hundred decode steps queued without a host sync in between (one-chunk shapes,
    the in-launch merge and, with the window filled, the merge + Wo launch) must leave the logits of the same steps
    run alone with separate attention and Wo launches (Q3_FUSE=0), bit for bit."""
    import ctypes as C
    import os
    path = os.path.join(Q.tmp_dir(), "4Bmini_w2048.bin")
    Q.synth("4Bmini", path, seq_len=2048)
    rng = np.random.default_rng(41)
    for T in (0, 1100):
        feed = [int(t) for t in rng.integers(0, 8192, size=300)]
        os.environ["Q3_FUSE"] = "0"
        try:
            mref = hip.q3_model_open(path.encode(), 1500, 0)
            assert hip.q3_device_attach(mref) == 0
        finally:
            del os.environ["Q3_FUSE"]
        if T:
            hip.q3_kv_fill_random(mref, T, 3)
        for k, tok in enumerate(feed):
            hip.q3_forward_device(mref, tok, T + k)
        hip.q3_logits_fetch(mref)
        want = Q.logits_array(mref).copy()
        hip.q3_model_close(mref)

        ma = hip.q3_model_open(path.encode(), 1500, 0)      # fused launches (the default)
        mb = hip.q3_model_open(path.encode(), 1500, 0)      # the foreign load
        assert hip.q3_device_attach(ma) == 0 and hip.q3_device_attach(mb) == 0
        if T:
            hip.q3_kv_fill_random(ma, T, 3)
        prompt = (C.c_int * 256)(*[int(t) for t in rng.integers(0, 8192, size=256)])
        hip.q3_prefill(mb, prompt, 256, 0)                  # first use of its launch shapes, out of the way
        for k, tok in enumerate(feed):
            hip.q3_forward_device(ma, tok, T + k)           # queued on ma's stream, no sync
            if k % 60 == 30:
                hip.q3_prefill(mb, prompt, 256, 0)          # runs on mb's stream beside the queued steps; returns when done
        hip.q3_logits_fetch(ma)
        got = Q.logits_array(ma)
        assert np.array_equal(got, want), f"context {T}"
        hip.q3_model_close(ma); hip.q3_model_close(mb)

#!/usr/bin/env python3
"""Decode-throughput bench of the MI355X forward() path.

    python bench.py --gpus N --steps K --warmup W [--model 4B] [--context T]

A "step" is one forward(model, token, pos) through the C-ABI (libq3hip.so): the whole
decode step of the Qwen3 Q8_0 model plus the copy of the logits to the host, with the
weights and the KV cache already resident in HBM.  Workload: BASELINE.json's headline
configuration, Qwen3-4B-shaped random-init Q8_0 weights (SURVEY.md 8(d) recipe), greedy
feedback from token 9707.

N > 1 (launched by torch.distributed.run, one process per GPU; RANK / WORLD_SIZE /
LOCAL_RANK / MASTER_PORT from the environment): the layers are split into N contiguous
stages; the residual travels stage to stage by RCCL send/recv over xGMI
(q3_pipeline_*), and N independent greedy token streams keep every stage busy, so
`value` = tokens/s summed over the N streams ("weak" scaling: one stream per GPU).
The timed region is bracketed by an RCCL all-reduce (barrier) + stream sync on every
rank and the MAX over ranks is reported.  torch is not imported: its wheel carries its
own HIP/RCCL runtimes, which cannot share a process with the ones the library links.

The JSON line also carries
  roofline      HBM roofline of the dominant kernel (the gate/up GEMV): algorithmic bytes
                per launch / mean launch duration from HIP events on the launch stream
  cpu_baseline  the reference's own CPU path (oracle/_ref, upstream -Ofast flags, all
                host cores) or, when that build is absent, the oracle's port, on a
                bounded sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
START_TOKEN = 9707


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def host_cores():
    """Cores this process may use: the affinity mask, capped at the GPU box's per-GPU CPU
    share (16) -- os.cpu_count() reports the whole host and oversubscribes OpenMP."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("Q3_MAX_CORES", "16"))))


def cpu_baseline(path, vocab, budget_s=20.0):
    """Tokens/s of the CPU path on the host cores of this box (rank 0, N=1 only)."""
    import numpy as np
    import q3lib as Q
    cores = int(os.environ["OMP_NUM_THREADS"])
    ref = Q.reference_lib(fast=True)
    first = START_TOKEN % vocab
    if ref is not None:
        m = ref.model_create(path.encode(), 256)
        step = lambda t, p: ref.forward(m, t, p)   # noqa: E731
        kind = "reference"
    else:
        orc, host = Q.oracle_lib(), Q.host_lib()
        orc.orc_set_mode(Q.ORC_TREE)
        orc.orc_set_threads(cores)
        m = host.q3_model_open(path.encode(), 256, 1)
        step = lambda t, p: orc.orc_forward(m, t, p)   # noqa: E731
        kind = "port"
    tok, pos = first, 0
    t0 = time.perf_counter()
    lg = step(tok, pos)           # one untimed step (page-in)
    log(f"[bench] cpu baseline ({kind}, {cores} threads): first step {time.perf_counter() - t0:.2f}s")
    tok = int(np.ctypeslib.as_array(lg, shape=(vocab,)).argmax()); pos = 1
    t0 = time.perf_counter()
    n = 0
    while n < 1 or (time.perf_counter() - t0 < budget_s and n < 64 and pos < 250):
        lg = step(tok, pos)
        tok = int(np.ctypeslib.as_array(lg, shape=(vocab,)).argmax())
        pos += 1
        n += 1
        if n % 4 == 0:
            log(f"[bench] cpu baseline: {n} steps in {time.perf_counter() - t0:.1f}s")
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": "tokens/s", "cores": cores, "kind": kind,
            "sample": f"{n} greedy decode steps at pos 1..{pos - 1} of the same checkpoint, "
                      f"OMP_NUM_THREADS={cores}" + (", built -Ofast -march=x86-64-v3 -fopenmp (upstream CMakeLists.txt:17 "
                                                     "uses -Ofast -march=native; the build host is not the GPU box)"
                                                     if kind == "reference" else "")}


def chain_floor_from_log(path):
    """The committed measurement of tools/micro/chain_floor.hip: per launch shape, microseconds per
    token-equivalent of a hipGraph of this step's dependent launches that only stream their bytes."""
    import re
    try:
        us = [float(x) for x in re.findall(r"launches, ([0-9.]+) us per token-equivalent", open(path).read())]
    except OSError:
        return None
    if not us:
        return None
    return {"us_per_token": [round(min(us), 1), round(max(us), 1)],
            "tokens_per_s": [round(1e6 / max(us), 1), round(1e6 / min(us), 1)],
            "source": f"{os.path.relpath(path, ROOT)} ({len(us)} launch shapes, one MI355X; committed, not re-run here)"}


def job_nonce():
    """8 bytes that every rank of THIS job agrees on and another job does not share: the launcher's run id
    (spawn_ranks exports Q3_JOB_NONCE; torch.distributed.run exports TORCHELASTIC_RUN_ID to every worker -- but a
    default standalone run has the constant id "none", so then, and without either, the tag is built from the
    launcher's pid and start time, which the ranks of one launch share), hashed."""
    import hashlib
    key = os.environ.get("Q3_JOB_NONCE") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    if key in ("", "none"):
        # a default standalone torch.distributed.run exports the constant run id "none": fall back to what the ranks of one
        # launch share and another launch does not -- the launcher (their common parent process) and its start time
        ppid = os.getppid()
        try:
            started = open(f"/proc/{ppid}/stat").read().rsplit(")", 1)[1].split()[19]      # field 22: starttime
        except (OSError, IndexError):
            started = ""
        key = f"{os.environ.get('MASTER_PORT', '')}:{ppid}:{started}"
    return hashlib.sha256(key.encode()).digest()[:8]


def unlink_id_file(world, tmp):
    """Rank 0, after the barrier that proves every rank holds the id: the file must not outlive the rendezvous."""
    port = int(os.environ.get("MASTER_PORT", "0"))
    try:
        os.remove(os.path.join(tmp, f"rccl_id_{port}_{world}"))
    except OSError:
        pass


def share_id(rank, world, payload, tmp, timeout_s=300.0):
    """Rank 0 hands `payload` (the 128-byte RCCL unique id) to the other ranks of this node.  First choice: a
    TCP exchange on MASTER_ADDR:MASTER_PORT (nothing else of this job uses that port: torch.distributed is not
    loaded), every message tagged so that a foreign listener on the port is recognised.  Fallback when rank 0
    cannot bind it: a file keyed by port and world size, accepted only when written after this job started."""
    import socket
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "0"))
    tag = b"Q3ID" + bytes([world & 255]) + job_nonce()
    idfile = os.path.join(tmp, f"rccl_id_{port}_{world}")
    started = time.time()
    if rank == 0:
        assert payload is not None and len(payload) == 128
        with open(idfile + ".tmp", "wb") as f:
            f.write(tag + payload)
        os.replace(idfile + ".tmp", idfile)
        srv = None
        if port:
            try:
                srv = socket.socket()
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                srv.listen(world)
            except OSError:
                srv = None
        if srv is not None:
            srv.settimeout(1.0)
            served = 0
            while served < world - 1 and time.time() - started < timeout_s:
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                with c:
                    c.sendall(tag + payload)
                served += 1
            srv.close()
        return payload
    while time.time() - started < timeout_s:
        if port:
            try:
                with socket.create_connection((addr, port), timeout=1.0) as c:
                    c.settimeout(5.0)
                    got = b""
                    while len(got) < len(tag) + 128:
                        part = c.recv(len(tag) + 128 - len(got))
                        if not part:
                            break
                        got += part
                if got[:len(tag)] == tag and len(got) == len(tag) + 128:
                    return got[len(tag):]
            except OSError:
                pass
        # the file: only after rank 0 has had time to bind, and only a file of THIS job
        if time.time() - started > 15.0 and os.path.exists(idfile) and os.path.getmtime(idfile) > started - 120.0:
            got = open(idfile, "rb").read()
            if got[:len(tag)] == tag and len(got) == len(tag) + 128:
                return got[len(tag):]
        time.sleep(0.05)
    raise SystemExit("[bench] rank 0 never published the RCCL id")


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher: start N fresh rank processes (one per GPU) before
    anything in THIS process has touched HIP, relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    probe = subprocess.run([sys.executable, "-c",
                            f"import sys; sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); import q3lib; "
                            "print(q3lib.hip_lib().q3_device_count())"],
                           capture_output=True, text=True, timeout=600)
    try:
        ndev = int(probe.stdout.strip().splitlines()[-1])
    except Exception:
        raise SystemExit(f"[bench] cannot count HIP devices: {probe.stderr[-400:]}")
    if ndev < n:
        raise SystemExit(f"[bench] --gpus {n} but only {ndev} HIP device(s) visible: one process per GPU, no oversubscription")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    nonce = f"{os.getpid()}-{time.time_ns()}"
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), Q3_BENCH_CHILD="1", Q3_JOB_NONCE=nonce)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    line, bad = supervise(procs, float(os.environ.get("Q3_BENCH_TIMEOUT", "900")) + 60.0)
    if bad:
        raise SystemExit(f"[bench] rank(s) {bad} failed")
    sys.stdout.write(line)
    sys.stdout.flush()


def supervise(procs, timeout_s):
    """Watch ALL rank processes: as soon as one exits non-zero, or the deadline passes, the others are killed
    (a rank that died would otherwise leave its peers blocked in the RCCL rendezvous, holding the GPUs).
    Rank 0's stdout is drained on a thread so that a long line cannot block it.  Returns (rank 0's stdout,
    list of failed ranks)."""
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + timeout_s
    bad = []
    while True:
        codes = [pr.poll() for pr in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad or all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            bad = [r for r, c in enumerate(codes) if c is None]
            log(f"[bench] ranks {bad} still running after {timeout_s:.0f}s: killing them")
            break
        time.sleep(0.05)
    if bad:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for pr in procs:
        pr.wait()
    reader.join(10.0)
    return "".join(c or "" for c in chunks), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="4B")
    ap.add_argument("--context", type=int, default=0, help="cached positions before the timed steps")
    ap.add_argument("--seq-len", type=int, default=8192, help="context window allocated on the device")
    ap.add_argument("--dtype", default="q8", choices=["q8", "fp16"],
                    help="fp16 = the contrast path of BASELINE config 5 (weights dequantised to binary16 at attach)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 512 / 4096 cached-position points")
    ap.add_argument("--no-dropin", action="store_true", help="skip the reference-loader + host-sampler loop")
    ap.add_argument("--no-models", action="store_true", help="skip the 1.7B / 8B points of the default line")
    args = ap.parse_args()

    os.environ["OMP_NUM_THREADS"] = str(host_cores())   # before libgomp is first loaded
    # RCCL's bootstrap must stay on loopback: all ranks are on this node and the boxes'
    # other interfaces refuse connections
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)        # no launcher: be one (before any HIP call in this process)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: refusing to measure a different job")
    ngpu = world

    import numpy as np
    import q3lib as Q
    hip = Q.hip_lib()
    if hip.q3_device_count() <= 0:
        raise SystemExit("[bench] no HIP device: the product path has no CPU fallback")

    if ngpu > 1:
        import signal
        signal.alarm(int(os.environ.get("Q3_BENCH_TIMEOUT", "900")))   # never hang a node: die instead
        # Rendezvous without torch: torch's wheel bundles its own HIP/RCCL runtimes, which must
        # not share a process with the ROCm 7.2 ones libq3hip.so links.  All ranks are on one
        # node, so rank 0 publishes the RCCL unique id in a file keyed by MASTER_PORT.
        if rank == 0:
            buf = (C.c_char * 128)()
            assert hip.q3_pipeline_unique_id(buf) == 0
            raw = share_id(0, ngpu, bytes(buf), Q.tmp_dir())
        else:
            raw = share_id(rank, ngpu, None, Q.tmp_dir())
        if hip.q3_device_count() < ngpu:
            raise SystemExit(f"[bench] {ngpu} ranks but {hip.q3_device_count()} HIP device(s): one process per GPU")
        assert hip.q3_pipeline_init(rank, ngpu, raw) == 0
        assert hip.q3_pipeline_size() == ngpu, (hip.q3_pipeline_size(), ngpu)
        hip.q3_pipeline_allreduce_max(0.0)            # everybody has joined
        if rank == 0:
            unlink_id_file(ngpu, Q.tmp_dir())

    tmp = Q.tmp_dir()
    path = os.path.join(tmp, f"{args.model}.bin")
    t0 = time.perf_counter()
    if rank == 0:
        Q.synth(args.model, path)
    if ngpu > 1:
        hip.q3_pipeline_allreduce_max(0.0)            # the checkpoint is on disk
    log(f"[bench] rank {rank}: checkpoint {path} ready in {time.perf_counter() - t0:.1f}s")
    seq = max(args.seq_len, args.context + args.steps + args.warmup + 8)
    m = hip.q3_model_open(path.encode(), seq, 0)
    assert m, "cannot open checkpoint"
    p = m.contents.params
    vocab = p.vocab_size
    t0 = time.perf_counter()
    assert (hip.q3_device_attach_fp16(m) if args.dtype == "fp16" else hip.q3_device_attach(m)) == 0
    log(f"[bench] weights resident in HBM after {time.perf_counter() - t0:.1f}s")

    K, W = args.steps, args.warmup
    pos0 = args.context
    if pos0 > 0:
        hip.q3_kv_fill_random(m, pos0, 99)

    if ngpu == 1:
        def run(n, tok, pos):
            for _ in range(n):
                lg = hip.forward(m, tok, pos)
                tok = hip.q3_argmax(lg, vocab)
                pos += 1
            return tok, pos
        tok, pos = run(W, START_TOKEN % vocab, pos0)
        hip.q3_device_sync(m)
        t0 = time.perf_counter()
        tok, pos = run(K, tok, pos)
        hip.q3_device_sync(m)
        elapsed = time.perf_counter() - t0
        total_tokens = K
        # `value` is the K steps the caller asked for.  A short K stays inside the cheapest attention launch
        # shape (fewer than 64 cached positions), so the same loop is ALSO timed over 20 and over 256 steps from
        # position pos0 + W, whatever K is, and both rates go into the line (`by_steps`).
        by_steps = {}
        for ks in (20, 256):
            if ks == K:
                by_steps[str(ks)] = round(K / elapsed, 2)
                continue
            if pos0 + W + ks + 8 > seq:
                continue
            t_tok, t_pos = run(W, START_TOKEN % vocab, pos0)
            hip.q3_device_sync(m)
            t1 = time.perf_counter()
            run(ks, t_tok, t_pos)
            hip.q3_device_sync(m)
            by_steps[str(ks)] = round(ks / (time.perf_counter() - t1), 2)
    else:
        # N streams x (W + K) tokens: untimed fill + warm-up, then K timed tokens per stream
        first = START_TOKEN % vocab
        hip.q3_pipeline_run(m, first, pos0, W)
        hip.q3_device_sync(m)
        hip.q3_pipeline_allreduce_max(0.0)
        t0 = time.perf_counter()
        hip.q3_pipeline_run(m, first, pos0 + W, K)
        hip.q3_device_sync(m)
        elapsed = hip.q3_pipeline_allreduce_max(time.perf_counter() - t0)
        total_tokens = K * ngpu
        tok, pos = first, pos0 + W + K
        # ONE token stream through the N stages (each stage busy one tick in N, a hand-off per
        # stage and token) -- the figure a single user sees; its roofline is ONE GPU's (only one is busy)
        hip.q3_pipeline_run_streams(m, first, pos0, W, 1)
        hip.q3_device_sync(m)
        hip.q3_pipeline_allreduce_max(0.0)
        t0 = time.perf_counter()
        hip.q3_pipeline_run_streams(m, first, pos0 + W, K, 1)
        hip.q3_device_sync(m)
        single_elapsed = hip.q3_pipeline_allreduce_max(time.perf_counter() - t0)

    out = {
        "metric": "decode tokens/sec Qwen3-4B Q8_0" if args.model == "4B" else f"decode tokens/sec Qwen3-{args.model} Q8_0",
        "value": round(total_tokens / elapsed, 2),
        "unit": "tokens/s",
        "n_gpus": ngpu,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(1000.0 * elapsed / K, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int8 x int8 -> int32 group dots, fp32 scale-accumulate",
        "data": "synthetic",
        "config": {"workload": f"Qwen3-{args.model}-shaped random-init Q8_0 (group 64), single-token decode, "
                               f"{pos0} cached positions at start, greedy feedback; one forward() per token incl. "
                               f"D2H of {vocab} logits",
                   "parallelism": "1 GPU" if ngpu == 1 else f"pp{ngpu}: layer pipeline, RCCL send/recv of the residual, "
                                                            f"{ngpu} concurrent streams"},
    }
    out["config"]["launch"] = "hipGraph of per-stage kernels: per layer qkv, attention+Wo (one launch), gate/up, down"
    out["rccl_comm_size"] = hip.q3_pipeline_size()
    bpt = hip.q3_bytes_per_token(C.byref(p), pos0 + W + K // 2)
    if args.dtype == "fp16":
        # binary16 weights: 2 bytes per element instead of 1 + 4/64; everything else as in the Q8_0 count
        Pd, KVD = p.n_heads * p.head_dim, p.n_kv_heads * p.head_dim
        elems = p.n_layers * (p.dim * (Pd + 2 * KVD) + Pd * p.dim + 3 * p.dim * p.hidden_dim) + p.vocab_size * p.dim
        bpt = bpt - elems * (1.0 + 4.0 / 64.0) + elems * 2.0
        out["metric"] = out["metric"].replace("Q8_0", "fp16 (dequantised, contrast path)")
        out["dtype"] = "binary16 weights x fp32 activations, fp32 accumulate"
        out["config"]["workload"] = out["config"]["workload"].replace("random-init Q8_0 (group 64)", "random-init Q8_0 dequantised to binary16 at attach")
    per_gpu_rate = out["value"] / ngpu
    out["hbm_roofline_frac_step"] = round(per_gpu_rate * bpt / 1e9 / HBM_PEAK_GBS * (1 if ngpu == 1 else 1.0), 4)
    out["bytes_per_token"] = int(bpt)
    if ngpu == 1:
        out["by_steps"] = {"tokens_per_s": by_steps,
                           "note": "the same decode loop timed over 20 and over 256 steps from the same start position: a "
                                   "20-step run never leaves the one-chunk attention launch shape (< 64 cached positions)"}
    else:
        single = K / single_elapsed
        out["single_stream"] = {"tokens_per_s": round(single, 2), "ms_per_token": round(1000.0 * single_elapsed / K, 4),
                                "frac_of_one_gpu_hbm_roofline": round(single * bpt / 1e9 / HBM_PEAK_GBS, 4),
                                "note": f"one token stream through the {ngpu} stages (q3_pipeline_run_streams(..., 1)): stages "
                                        "run one after the other, so the roofline it is quoted against is ONE GPU's; `value` is "
                                        f"the aggregate of {ngpu} concurrent streams"}

    if ngpu == 1:
        # the same decode loop kept on the device (argmax feeds the next step, no D2H per token)
        hip.q3_pipeline_run(m, tok, pos, 8); hip.q3_device_sync(m)
        t0 = time.perf_counter()
        hip.q3_pipeline_run(m, tok, pos + 8, K); hip.q3_device_sync(m)
        out["device_loop_tokens_per_s"] = round(K / (time.perf_counter() - t0), 2)
        pos += 8 + K
    if ngpu == 1 and args.dtype == "q8" and not args.no_sweep:
        # BASELINE config 3: the same decode step at 512 and 4096 cached positions (KV-tile + roofline
        # sweep).  The cache is filled with finite pseudo-random rows (untimed), then real steps are timed.
        ctx = {str(pos0): {"tokens_per_s": out["value"], "frac_of_hbm_roofline": out["hbm_roofline_frac_step"],
                           "bytes_per_token": out["bytes_per_token"], "steps": K}}
        Kc, Wc = min(K, 64), 8
        for T in (512, 4096):
            if T == pos0 or T + Kc + Wc + 8 > seq:
                continue
            hip.q3_kv_fill_random(m, T, 99)
            t_tok, t_pos = run(Wc, tok, T)
            hip.q3_device_sync(m)
            t0 = time.perf_counter()
            t_tok, t_pos = run(Kc, t_tok, t_pos)
            hip.q3_device_sync(m)
            rate = Kc / (time.perf_counter() - t0)
            b_t = hip.q3_bytes_per_token(C.byref(p), T + Wc + Kc // 2)
            ctx[str(T)] = {"tokens_per_s": round(rate, 2), "frac_of_hbm_roofline": round(rate * b_t / 1e9 / HBM_PEAK_GBS, 4),
                           "bytes_per_token": int(b_t), "steps": Kc}
        out["contexts"] = ctx
    if ngpu == 1 and args.dtype == "q8" and not args.no_dropin:
        # the drop-in as a reference user runs it: the reference's own loader and HOST sampler (whose
        # softmax() call is this library's export) around forward(), oracle/_ref/libqwen3_dropin.so
        drop = Q.dropin_lib()
        if drop is not None:
            md = drop.model_create(path.encode(), 512)
            smp = drop.sampler_create(vocab, 1e-6, 0.9, 1234)      # the reference's "-t 0"
            t_tok = START_TOKEN % vocab
            for p_ in range(4):
                t_tok = drop.sample(smp, drop.forward(md, t_tok, p_))
            t0 = time.perf_counter()
            nd = min(K, 64)
            for p_ in range(4, 4 + nd):
                t_tok = drop.sample(smp, drop.forward(md, t_tok, p_))
            out["dropin_loop_tokens_per_s"] = round(nd / (time.perf_counter() - t0), 2)
            out["dropin_loop"] = ("reference model_create + forward + host sample() (softmax export, qsort, xorshift) per token, "
                                  "temperature 1e-6 / top-p 0.9, oracle/_ref/libqwen3_dropin.so; the reference's host qsort of "
                                  f"{vocab} entries per token dominates this figure, not forward()")
            drop.sampler_free(smp)
            hip.q3_device_detach(md)
            drop.model_free(md)
    if ngpu == 1 and seq >= 512:
        # prompt ingestion (q3_prefill: up to 64 positions per pass; Q8_0 products on int8 MFMA, bit-identical
        # to feeding the prompt through forward(); binary16 products on f16 MFMA for --dtype fp16) --
        # reported next to the decode rate, not part of `value`
        n_pf = 256
        prompt = (C.c_int * n_pf)(*[int(t) for t in np.random.default_rng(5).integers(0, vocab, size=n_pf)])
        hip.q3_prefill(m, prompt, n_pf, 0)             # untimed: first use of every launch shape of this prompt
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            hip.q3_prefill(m, prompt, n_pf, 0)         # a prompt from position 0 (the rows it rewrites stay finite)
            times.append(time.perf_counter() - t0)
        out["prefill_tokens_per_s"] = round(n_pf / sorted(times)[1], 1)
        out["prefill_prompt_tokens"] = n_pf
        out["prefill_timing"] = "median of 3 passes after one untimed pass of the same prompt"
    if rank == 0 and ngpu == 1 and not args.no_roofline and args.dtype == "q8":
        # the roofline is quoted against the vendor HBM peak; next to it, what a plain device copy reaches here
        copy = hip.q3_measure_copy_gbps(1 << 30, 8)
        out["hbm_copy_gbps_measured"] = round(copy, 1)
        if args.model == "4B":
            # context for `value`: what the platform charges for this step's 146 dependent launches (four per layer) when
            # every launch only streams its stage's bytes (tools/micro/chain_floor.hip) -- a committed measurement of
            # round 4, not re-run here
            floor = chain_floor_from_log(os.path.join(ROOT, "profiles", "r04_chain_floor.log"))
            if floor:
                floor["launches_per_token"] = 146
                floor["launches_per_layer"] = 4
                out["dependent_launch_floor"] = floor
        out["frac_of_measured_copy"] = round(per_gpu_rate * bpt / 1e9 / copy, 4)
        hip.q3_prof_enable(m, 1)
        hip.q3_prof_reset(m)
        for _ in range(3):           # untimed: first launches of the non-graph path
            lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, vocab); pos += 1
        hip.q3_prof_reset(m)
        for _ in range(16):
            lg = hip.forward(m, tok, pos); tok = hip.q3_argmax(lg, vocab); pos += 1
        ents = (Q.ProfEntry * 16)()
        n = hip.q3_prof_get(m, ents, 16)
        hip.q3_prof_enable(m, 0)
        kern = {}
        for e in ents[:n]:
            if e.launches:
                us = 1000.0 * e.ms_total / e.launches
                kern[e.name.decode()] = {"launches": int(e.launches), "us": round(us, 3),
                                         "GBps": round(e.bytes_per_launch / us / 1e3, 1) if e.bytes_per_launch else None}
        for name in kern:      # the same launches by the device clock read inside the kernel
            dus = hip.q3_prof_device_us(m, name.encode())
            if dus > 0:
                kern[name]["us_device_clock"] = round(dus, 3)
        dom = kern.get("gateup")
        dom_bytes = hip.q3_gemv_bytes(2 * p.hidden_dim, p.dim)
        if dom:
            # duration of a launch = first workgroup in .. last workgroup out by s_memrealtime inside
            # the kernel when available (agrees with rocprofv3's dispatch durations); the HIP-event
            # bracket of the same launches is reported next to it
            us_best = dom.get("us_device_clock") or dom["us"]
            ach = round(dom_bytes / us_best / 1e3, 1)
            traffic, traffic_check = None, None
            try:   # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE)
                pm = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))["kernels"]["gateup"]
                if args.model == "4B":
                    traffic = pm["hbm_read_bytes_corrected"] + pm["hbm_write_bytes"]
                    # in-run check: the bytes this run prices the launch at against the committed counter figure
                    ratio = traffic / float(dom_bytes)
                    traffic_check = {"traffic_over_bytes_per_launch": round(ratio, 4), "within_3_percent": bool(abs(ratio - 1.0) <= 0.03)}
            except Exception:
                pass
            out["roofline"] = {"bound": "hbm", "kernel": "k_gemv3<PRO_NORM,EPI_SWIGLU,3,8> (gate/up GEMV)",
                               "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "traffic_source": "profiles/r04_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes "
                                                 "of this workload (FETCH_SIZE doubled per the gfx950 note), committed -- not re-measured by this run",
                               "traffic_check": traffic_check,
                               "timing": "in-kernel device clock (s_memrealtime, first workgroup in .. last out) of the "
                                         "same launches that the HIP events bracket on the launch stream; "
                                         "us_per_launch_events includes the end-of-kernel release (~2 us)",
                               "us_per_launch_events": dom["us"],
                               "bytes_per_launch": int(dom_bytes),
                               "us_per_launch": us_best}
        out["kernels"] = kern
    hip.q3_model_close(m)

    if rank == 0 and ngpu == 1 and args.model == "4B" and args.dtype == "q8" and not args.no_models:
        # BASELINE configs 2 and 4 in the driver-visible line: the same decode loop on the 1.7B shapes and on the 8B
        # shapes (untied classifier), 64 steps each after 8 untimed ones, one GPU; `value` stays the 4B figure
        models = {}
        for name in ("1.7B", "8B"):
            try:
                mp = os.path.join(tmp, f"{name}.bin")
                Q.synth(name, mp)
                mm = hip.q3_model_open(mp.encode(), 1024, 0)
                pp = mm.contents.params
                vv = pp.vocab_size
                t_tok, t_pos = START_TOKEN % vv, 0
                for _ in range(8):
                    t_tok = hip.q3_argmax(hip.forward(mm, t_tok, t_pos), vv); t_pos += 1
                hip.q3_device_sync(mm)
                t1 = time.perf_counter()
                for _ in range(64):
                    t_tok = hip.q3_argmax(hip.forward(mm, t_tok, t_pos), vv); t_pos += 1
                hip.q3_device_sync(mm)
                rate = 64 / (time.perf_counter() - t1)
                b_m = hip.q3_bytes_per_token(C.byref(pp), 8 + 32)
                models[name] = {"tokens_per_s": round(rate, 2), "frac_of_hbm_roofline": round(rate * b_m / 1e9 / HBM_PEAK_GBS, 4),
                                "bytes_per_token": int(b_m), "steps": 64, "warmup": 8}
                hip.q3_model_close(mm)
            except Exception as exc:      # a reported extra, never a reason to lose the line
                models[name] = {"error": repr(exc)}
        out["models"] = models

    if rank == 0 and ngpu == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(path, vocab)
        except Exception as exc:   # the baseline is a reported number, never a reason to lose the line
            out["cpu_baseline"] = {"value": None, "error": repr(exc)}
    if ngpu > 1:
        hip.q3_pipeline_allreduce_max(0.0)
        hip.q3_pipeline_shutdown()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

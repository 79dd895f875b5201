"""bench.py's launch contract without a GPU: a WORLD_SIZE that disagrees with --gpus is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "refusing" in r.stderr


def _share(rank, world, port, tmp, q, payload=None, bind_busy=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import bench
    q.put((rank, bench.share_id(rank, world, payload, tmp, timeout_s=60.0)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rccl_id_reaches_every_rank_over_tcp(tmp_path):
    """rank 0 -> ranks 1..3 on MASTER_PORT; a stale id file of an earlier job with the same port is ignored"""
    import multiprocessing as mp
    world, port = 4, _free_port()
    stale = tmp_path / f"rccl_id_{port}_{world}"
    stale.write_bytes(b"Q3ID" + bytes([world]) + b"\x01" * 128)
    os.utime(stale, (1.0, 1.0))
    payload = bytes(range(128))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_share, args=(r, world, port, str(tmp_path), q, payload if r == 0 else None)) for r in range(world)]
    for p in reversed(ps):            # rank 0 last: the others must wait for it
        p.start()
    got = dict(q.get(timeout=90) for _ in range(world))
    for p in ps:
        p.join(30)
        assert p.exitcode == 0
    assert all(got[r] == payload for r in range(world))


def test_rccl_id_falls_back_to_the_file_when_the_port_is_taken(tmp_path):
    """somebody else listens on MASTER_PORT and answers nothing useful: the ranks take the fresh file"""
    import multiprocessing as mp
    import socket
    world = 2
    with socket.socket() as other:
        other.bind(("127.0.0.1", 0))
        other.listen(4)
        port = other.getsockname()[1]
        payload = bytes(reversed(range(128)))
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        ps = [ctx.Process(target=_share, args=(r, world, port, str(tmp_path), q, payload if r == 0 else None)) for r in range(world)]
        for p in ps:
            p.start()
        got = dict(q.get(timeout=120) for _ in range(world))
        for p in ps:
            p.join(30)
            assert p.exitcode == 0
    assert got[1] == payload


def test_launcher_kills_the_other_ranks_when_one_dies():
    """bench.supervise: a rank that exits non-zero must not leave its peers waiting in the RCCL
    rendezvous -- the survivors are killed and the failure is reported, well inside the overall timeout."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    sleeper = [sys.executable, "-c", "import time; print('line', flush=True); time.sleep(120)"]
    dier = [sys.executable, "-c", "import sys, time; time.sleep(0.5); sys.exit(3)"]
    procs = [subprocess.Popen(sleeper, stdout=subprocess.PIPE, text=True), subprocess.Popen(dier),
             subprocess.Popen(sleeper, stdout=subprocess.DEVNULL)]
    t0 = time.time()
    line, bad = bench.supervise(procs, 60.0)
    assert bad == [1] and time.time() - t0 < 20.0
    assert all(p.poll() is not None for p in procs)
    assert "line" in line


def test_launcher_times_out_hung_ranks():
    sys.path.insert(0, ROOT)
    import bench
    sleeper = [sys.executable, "-c", "import time; time.sleep(120)"]
    procs = [subprocess.Popen(sleeper, stdout=subprocess.PIPE, text=True), subprocess.Popen(sleeper)]
    _line, bad = bench.supervise(procs, 1.0)
    assert bad == [0, 1]
    assert all(p.poll() is not None for p in procs)


def test_id_file_of_another_job_is_not_accepted(tmp_path, monkeypatch):
    """file fallback: an id file left by a job with a different launcher nonce carries a different tag"""
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setenv("Q3_JOB_NONCE", "job-a")
    a = bench.job_nonce()
    monkeypatch.setenv("Q3_JOB_NONCE", "job-b")
    assert bench.job_nonce() != a and len(a) == 8

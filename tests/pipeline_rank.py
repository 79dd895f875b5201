#!/usr/bin/env python3
"""One rank of the RCCL layer pipeline (started by tests/test_gpu_pipeline_rccl.py, one process per
GPU): joins the communicator, runs `n` greedy tokens of `world` concurrent streams, and -- on the last
rank -- prints the tokens of every stream as one JSON line."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import q3lib as Q


def main():
    path, n, idfile = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    hip = Q.hip_lib()
    if hip.q3_device_count() < world:
        raise SystemExit(f"{world} ranks need {world} GPUs")
    if rank == 0:
        buf = (C.c_char * 128)()
        assert hip.q3_pipeline_unique_id(buf) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.replace(idfile + ".tmp", idfile)
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("rank 0 never published the RCCL id")
        time.sleep(0.02)
    assert hip.q3_pipeline_init(rank, world, open(idfile, "rb").read()) == 0
    assert hip.q3_pipeline_size() == world
    m = hip.q3_model_open(path.encode(), 256, 0)
    ticks = hip.q3_pipeline_run(m, 11, 0, n)
    assert ticks == n * world, ticks          # ticks this rank computed on (include/q3_ext.h)
    hip.q3_device_sync(m)
    if rank == world - 1:
        streams = []
        for s in range(world):
            out = (C.c_int * n)()
            assert hip.q3_pipeline_tokens(m, s, out, n) == n
            streams.append(list(out))
        print(json.dumps({"streams": streams}), flush=True)
    hip.q3_pipeline_allreduce_max(0.0)
    hip.q3_model_close(m)
    hip.q3_pipeline_shutdown()


if __name__ == "__main__":
    main()

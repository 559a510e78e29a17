"""The TCP bootstrap of the RCCL unique id (include/dvs_comm.h, dvs_comm_bootstrap) on CPU: three ranks, the launcher's environment
(MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE) honoured, the id on a dedicated port next to a socket that already listens on
MASTER_PORT (as torch.distributed.run's store does), a stray connection and a duplicate rank ignored, and a deadline instead of a
hang when a rank never shows up. No GPU and no RCCL involved: this is host code."""
import ctypes as C
import os
import socket
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_CODE = r"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ["DVS_ROOT"])
import divshot_amd as dv
rank = int(os.environ["RANK"])
buf = (C.c_ubyte * 128)(*([(7 * i + 3) % 251 for i in range(128)] if rank == 0 else [0] * 128))
rc = dv.lib.dvs_comm_bootstrap(-1, -1, None, 0, buf)          # everything from the environment
print("RESULT", rc, bytes(buf).hex(), dv.lib.dvs_last_error().decode() if rc else "")
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn(rank, world, port, extra=None):
    env = dict(os.environ, DVS_ROOT=ROOT, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("DVS_COMM_PORT", None)
    env.update(extra or {})
    return subprocess.Popen([sys.executable, "-c", RANK_CODE], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)


def _result(p, timeout=120):
    out, err = p.communicate(timeout=timeout)
    line = [l for l in out.splitlines() if l.startswith("RESULT ")]
    assert line, out + err
    f = line[-1].split(" ", 3)
    return int(f[1]), f[2], (f[3] if len(f) > 3 else "")


def test_three_ranks_next_to_a_busy_master_port_with_strays():
    port = _free_port()
    store = socket.socket(); store.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    store.bind(("127.0.0.1", port)); store.listen(4)          # the launcher's own store occupies MASTER_PORT
    boot = port + 1789 if port + 1789 < 65536 else 1024 + (port + 1789 - 65536) % (65536 - 1024)
    r0 = _spawn(0, 3, port, {"MASTER_ADDR": "localhost"})      # rank 0 only listens: it may know the host under another name
    # strays on the bootstrap port while rank 0 waits: garbage, a silent connection, and a well-formed hello of the wrong job
    deadline = time.time() + 30
    while True:
        try:
            s = socket.create_connection(("127.0.0.1", boot), timeout=1); break
        except OSError:
            assert time.time() < deadline, "rank 0 never listened on the bootstrap port"
            time.sleep(0.1)
    s.sendall(b"GET / HTTP/1.0\r\n\r\n"); s.close()
    silent = socket.create_connection(("127.0.0.1", boot), timeout=1)
    wrong = socket.create_connection(("127.0.0.1", boot), timeout=1)
    wrong.sendall(struct.pack("<IIQ", 0x44565343, 1, 12345)); wrong.close()
    r1 = _spawn(1, 3, port)
    r2 = _spawn(2, 3, port)
    res = [_result(p) for p in (r0, r1, r2)]
    silent.close(); store.close()
    want = bytes((7 * i + 3) % 251 for i in range(128)).hex()
    for rc, got, msg in res:
        assert rc == 0, msg
        assert got == want


def test_listener_answers_a_served_rank_again_during_the_grace_period():
    """a peer whose receive of rank 0's confirmation failed AFTER rank 0 counted it used to retry against a closed listener
    while every other rank already sat in ncclCommInitRank. Rank 0 now keeps the listener open for a grace period (DVS_COMM_GRACE_S,
    default 1 s) after the last rank was counted and answers already-served ranks again. Played with a raw socket as rank 1 of 2:
    complete handshake, then the same hello once more — the id and the confirmation come back a second time."""
    port = _free_port()
    boot = port + 1789 if port + 1789 < 65536 else 1024 + (port + 1789 - 65536) % (65536 - 1024)
    r0 = _spawn(0, 2, port, {"DVS_COMM_GRACE_S": "6", "DVS_COMM_NONCE": "grace-test"})
    import hashlib
    key = (":" + str(port) + "|grace-test").encode()
    h = 1469598103934665603
    for ch in key:
        h ^= ch; h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF                      # FNV-1a, as dvs_comm.cpp job_nonce()
    want = bytes((7 * i + 3) % 251 for i in range(128))

    def handshake():
        deadline = time.time() + 30
        while True:
            try:
                s = socket.create_connection(("127.0.0.1", boot), timeout=5); break
            except OSError:
                assert time.time() < deadline, "rank 0 does not listen"
                time.sleep(0.1)
        s.sendall(struct.pack("<IIQ", 0x44565343, 1, h))
        buf = b""
        while len(buf) < 136:
            chunk = s.recv(136 - len(buf)); assert chunk, "rank 0 closed the connection before the id"
            buf += chunk
        assert struct.unpack("<I", buf[:4])[0] == 0x44565343 and buf[8:] == want
        s.sendall(struct.pack("<I", 0x44565343))
        assert s.recv(1) == b"\x01"
        s.close()

    env_keys = [k for k in ("TORCHELASTIC_RUN_ID", "SLURM_JOB_ID") if k in os.environ]
    if env_keys:
        import pytest
        pytest.skip("the job nonce also hashes " + ", ".join(env_keys))
    handshake()                  # rank 0 has now counted its only peer
    time.sleep(0.3)
    handshake()                  # ... and still answers it (before round 6: connection refused)
    rc, got, msg = _result(r0)
    assert rc == 0, msg


def test_missing_rank_times_out_with_an_error():
    port = _free_port()
    t0 = time.time()
    rc, _, msg = _result(_spawn(0, 2, port, {"DVS_COMM_TIMEOUT_S": "3"}), timeout=60)
    assert rc != 0 and "timed out" in msg and time.time() - t0 < 30
    rc, _, msg = _result(_spawn(1, 2, port, {"DVS_COMM_TIMEOUT_S": "2"}), timeout=60)      # nobody serves
    assert rc != 0 and "no RCCL id" in msg


COLL_CODE = r"""
import ctypes as C, os, sys, json
import numpy as np
sys.path.insert(0, os.environ["DVS_ROOT"])
import divshot_amd as dv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
c = dv.lib.dvs_comm_create(-1, -1, -1, None, 0)          # device -1 + DVS_COMM_BACKEND=tcp: host buffers, no HIP
assert c, dv.lib.dvs_last_error()
res = {}
a = (np.arange(1000, dtype=np.float32) * (rank + 1) + np.float32(0.1 * rank))
want = sum((np.arange(1000, dtype=np.float32) * (r + 1) + np.float32(0.1 * r)) for r in range(world))      # rank order: the sum the star forms
assert dv.lib.dvs_comm_all_reduce_sum_f32(c, None, a.ctypes.data, a.size) == 0
res["allreduce_bits_equal_rank_order_sum"] = bool(np.array_equal(a, want.astype(np.float32)))
m = np.array([rank, 10 - rank, 5], np.int32)
assert dv.lib.dvs_comm_all_reduce_max_i32(c, None, m.ctypes.data, m.size) == 0
res["max"] = m.tolist()
s = np.full(7, rank + 1, np.float32); g = np.zeros(7 * world, np.float32)
assert dv.lib.dvs_comm_all_gather_f32(c, None, s.ctypes.data, g.ctypes.data, s.size) == 0
res["gather"] = g.reshape(world, 7)[:, 0].tolist()
send = np.arange(4 * world, dtype=np.float32) + rank; recv = np.zeros(4, np.float32)
assert dv.lib.dvs_comm_reduce_scatter_sum_f32(c, None, send.ctypes.data, recv.ctypes.data, recv.size) == 0
res["reduce_scatter"] = recv.tolist()
b = np.full(5, 7 if rank == 1 else -1, np.int32)
assert dv.lib.dvs_comm_broadcast(c, None, b.ctypes.data, b.nbytes, 1) == 0
res["broadcast"] = b.tolist()
assert dv.lib.dvs_comm_group_start(c) == 0 and dv.lib.dvs_comm_group_end(c) == 0
dv.lib.dvs_comm_destroy(c)
print("RESULT 0 " + json.dumps(res).replace(" ", ""))
"""


def test_tcp_backend_collectives_three_ranks_on_host_buffers():
    """The TEST backend of include/dvs_comm.h (DVS_COMM_BACKEND=tcp) with device = -1: every collective of the data-parallel step on host
    buffers, three ranks, no GPU — sums formed in rank order (all ranks receive the same bits), max, all-gather slot order,
    reduce-scatter slices, broadcast from a non-zero root."""
    import json
    port = _free_port()
    procs = []
    for r in range(3):
        env = dict(os.environ, DVS_ROOT=ROOT, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DVS_COMM_BACKEND="tcp",
                   DVS_COMM_TIMEOUT_S="60")
        env.pop("DVS_COMM_PORT", None)
        procs.append(subprocess.Popen([sys.executable, "-c", COLL_CODE], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=180)
        line = [l for l in so.splitlines() if l.startswith("RESULT 0 ")]
        assert line, so + se
        outs.append(json.loads(line[-1][9:]))
    for r, res in enumerate(outs):
        assert res["allreduce_bits_equal_rank_order_sum"]
        assert res["max"] == [2, 10, 5]
        assert res["gather"] == [1.0, 2.0, 3.0]
        want = [float(sum(4 * r + k + q for q in range(3))) for k in range(4)]
        assert res["reduce_scatter"] == want, (r, res["reduce_scatter"], want)
        assert res["broadcast"] == [7] * 5
    assert outs[0] == {**outs[0]} and outs[0]["allreduce_bits_equal_rank_order_sum"]

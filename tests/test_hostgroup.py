"""CPU: the torch-free side of the sharded searches.

* ``pygps_amd.hostgroup.HostGroup``: collectives + ticket counter over sockets, 3 processes.
* the library's host collectives (``pgp_comm_bcast_host`` / ``allreduce`` / ``allgather``) on a host-only communicator whose
  call-backs the HostGroup serves -- no torch in the process (the worker asserts ``"torch" not in sys.modules``).
* ``ShardedMinimize`` (Core/opt.py:301-327 sharded) over that communicator: every rank gets what the sequential ``Minimize``
  gets, static and dynamic deal.
* ``valid.sharded_k_fold`` (Validation/valid.py:20-66 sharded): fold -> rank bookkeeping and the one all-gather, on a numpy
  stand-in model; ``k_fold_validation`` / ``k_fold_index`` / metrics against the fixture recorded from the reference (G19).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import golden

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "workers", "search_worker.py")


from conftest import free_port as _free_port  # noqa: E402  (below the ephemeral range, port + 17 free as well)


def launch(world, case, out_dir, *args, no_torch=True, extra_env=None, timeout=600):
    """`world` processes with the environment a launcher exports; returns when all have exited with status 0."""
    return _launch_once(world, case, out_dir, args, no_torch, extra_env, timeout, attempts_left=2)


def _launch_once(world, case, out_dir, args, no_torch, extra_env, timeout, attempts_left):
    port = _free_port()
    procs = []
    for r in range(world):
        env = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK", "PYGPS_AMD_TORCH_FIRST")}
        env.update(RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if no_torch:
            env["PYGPS_AMD_NO_TORCH"] = "1"
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, WORKER, case, str(out_dir)] + [json.dumps(a) for a in args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    if attempts_left > 0 and any(p.returncode != 0 and ("EADDRINUSE" in o or "ddress already in use" in o) for p, o in zip(procs, outs)):
        return _launch_once(world, case, out_dir, args, no_torch, extra_env, timeout, attempts_left - 1)   # a lost race for the port
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d:\n%s" % (r, o[-4000:])
    return [np.load(os.path.join(str(out_dir), "r%d.npz" % r)) for r in range(world)]


def test_hostgroup_collectives_and_tickets(tmp_path):
    rs = launch(3, "group_ops", tmp_path)
    for r, z in enumerate(rs):
        assert np.array_equal(z["b"], np.arange(5.0) + 20)                        # root = world - 1
        assert np.array_equal(z["s"], [3.0, 6.0, np.inf]) and np.array_equal(z["mx"], [2.0, -3.0])
        assert np.array_equal(z["ga"], [[0, 0], [1, 1], [2, 4]])
        assert sorted(z["allt"].ravel().tolist()) == list(range(9))               # 9 tickets, each handed out exactly once
    assert sorted(int(z["t2"]) for z in rs) == [0, 1, 2]


def test_library_host_collectives_without_torch(tmp_path):
    rs = launch(3, "comm_ops", tmp_path)
    for r, z in enumerate(rs):
        assert str(z["transport"]) == "host"
        assert np.array_equal(z["b"], np.arange(6.0).reshape(2, 3))
        ga = z["ga"]
        assert ga.shape == (3, 1, 2) and np.array_equal(ga[:, 0, 0], [0, 1, 2])
        assert np.isinf(ga[0, 0, 1]) and np.isnan(ga[1, 0, 1]) and np.isnan(ga[2, 0, 1])   # non-finite values arrive as they are
        assert np.array_equal(z["s"], [6.0, 3.0]) and np.array_equal(z["mx"], [3.0, 0.0])


@pytest.mark.parametrize("world,R,streams,deal", [(2, 6, 1, "auto"), (2, 5, 2, "dynamic"), (3, 7, 1, "static"), (1, 4, 2, "auto")])
def test_sharded_minimize_without_torch_equals_sequential(tmp_path, world, R, streams, deal):
    sys.path.insert(0, HERE)
    from test_host_logic import _FakeModel, _conf
    from pygps_amd import opt
    m = _FakeModel()
    o = opt.Minimize(m, _conf(m, R))
    np.random.seed(7)
    h_seq, f_seq = o.findMin(m.x, m.y, numIters=15)
    rs = launch(world, "fake_search", tmp_path, R, streams, deal)
    for z in rs:
        assert float(z["f"]) == f_seq and np.array_equal(z["h"], h_seq)           # same optimum on every rank
        assert np.array_equal(z["x"], np.arange(4.0).reshape(4, 1))               # data arrived by broadcast
        assert np.array_equal(z["runs_f"], rs[0]["runs_f"]) and len(z["runs_f"]) == R
        assert np.array_equal(z["owner"], rs[0]["owner"])
    own = rs[0]["owner"]
    if deal == "static":
        assert np.array_equal(own, np.arange(R) % world)
    assert set(own.tolist()) <= set(range(world))
    if streams == 1:
        assert sum(int(z["calls"]) for z in rs) == m.calls                        # the work was shared, nothing ran twice


def test_k_fold_helpers_match_the_reference():
    from pygps_amd import valid
    g = golden("G19_kfold")
    N, K = int(g["N"]), int(g["K"])
    folds = list(valid.k_fold_index(N, K))
    assert isinstance(folds[0][0], list) and len(folds) == K
    assert np.array_equal(folds[0][0], g["fold0_train_idx"]) and np.array_equal(folds[0][1], g["fold0_test_idx"])
    assert np.array_equal(folds[9][1], g["fold9_test_idx"])
    from conftest import synth_reg
    x, y = synth_reg(N, int(g["d"]))
    for k, (xtr, xte, ytr, yte) in enumerate(valid.k_fold_validation(x, y, K)):
        assert np.array_equal(xtr, x[folds[k][0]]) and np.array_equal(yte, y[folds[k][1]])
    # randomise=True: the global generator shuffles [x | y]; y comes back one-dimensional (Validation/valid.py:34-38)
    np.random.seed(int(g["rand_seed"]))
    xtr, xte, ytr, yte = next(valid.k_fold_validation(x[:50], y[:50], 5, randomise=True))
    assert np.array_equal(xtr, g["rand_x_train"]) and np.array_equal(xte, g["rand_x_test"])
    assert ytr.ndim == 1 and np.array_equal(ytr, g["rand_y_train"]) and np.array_equal(yte, g["rand_y_test"])
    # metrics (valid.py:70-146) on made-up predictions
    p = np.array([[1.0], [-1.0], [1.0], [1.0], [-1.0]])
    t = np.array([[1.0], [1.0], [-1.0], [1.0], [-1.0]])
    assert valid.ACC(p, t) == 3.0 / 5 and valid.Prec(p, t) == 2.0 / 3 and valid.Recall(p, t) == 2.0 / 3
    assert valid.RMSE(p, t) == np.sqrt(np.mean((p - t) ** 2))
    assert str(g["nlpd_raises"]) == "NameError"                                   # the reference's NLPD cannot run as written
    mu, s2 = np.array([[0.5], [1.0]]), np.array([[0.2], [0.3]])
    yy = np.array([[0.0], [2.0]])
    assert abs(valid.NLPD(yy, mu, s2) - np.mean(0.5 * np.log(2 * np.pi * s2) + 0.5 * (yy - mu) ** 2 / s2)) < 1e-15


@pytest.mark.parametrize("world,K", [(2, 7), (3, 3)])
def test_sharded_k_fold_bookkeeping_without_torch(tmp_path, world, K):
    from pygps_amd import valid
    rs = launch(world, "kfold_fake", tmp_path, K)
    rng = np.random.RandomState(3)
    x = rng.randn(53, 2)
    y = x[:, :1] * 2 + 0.1 * rng.randn(53, 1)
    want_nlZ, want_rmse, want_nlpd = [], [], []
    for xtr, xte, ytr, yte in valid.k_fold_validation(x, y, K):
        ym = xte[:, :1] * 2.0 + np.mean(ytr)
        want_nlZ.append(np.sum(ytr)); want_rmse.append(valid.RMSE(ym, yte)); want_nlpd.append(valid.NLPD(yte, ym, np.full_like(ym, 0.5)))
    for z in rs:
        assert np.allclose(z["nlZ"], want_nlZ, rtol=1e-14) and np.allclose(z["RMSE"], want_rmse, rtol=1e-14)
        assert np.allclose(z["NLPD"], want_nlpd, rtol=1e-14)
        assert np.array_equal(z["owner"], rs[0]["owner"]) and set(z["owner"].tolist()) <= set(range(world))


def test_multi_dataset_objective_sharded_without_torch(tmp_path):
    """Demo/Clustering/pyGP_extension.py:27-76 sharded: data set i on rank i % world, ONE all-reduce of 1 + nhyp + D doubles."""
    rs = launch(2, "multi_fake", tmp_path)
    rng = np.random.RandomState(4)
    xs = [rng.randn(5 + i, 2) for i in range(5)]
    ys = [rng.randn(5 + i, 1) for i in range(5)]
    each = np.array([np.sum(y ** 2) * 1.3 for y in ys])
    g = np.array([sum(np.sum(y) for y in ys), sum(np.sum(x) for x in xs), each.sum(), 5.0])
    for z in rs:
        assert np.allclose(z["each"], each, rtol=1e-14) and abs(float(z["tot"]) - each.sum()) < 1e-12 * each.sum()
        assert np.allclose(z["g"], g, rtol=1e-13)
    assert sorted(int(z["calls"]) for z in rs) == [2, 3]                          # 5 data sets over 2 ranks


def test_hostgroup_wire_is_fixed_frames_and_survives_strangers(tmp_path):
    """nothing received is unpickled.  A stranger that connects and sends a pickle, one that sends nothing and one
    with a wrong secret are dropped while the rendezvous goes on; the real rank (right secret) gets its collectives."""
    import pickle
    import threading
    import time
    from pygps_amd import hostgroup
    src = open(hostgroup.__file__).read()
    assert "import pickle" not in src and "pickle.loads" not in src
    port = _free_port()
    os.environ["PYGPS_AMD_GROUP_SECRET"] = "s3cret"
    try:
        res = {}

        def rank0():
            g = hostgroup.HostGroup(0, 2, "127.0.0.1", port, timeout=60.0)
            res["sum"] = g.allreduce(np.array([1.0, 2.0]), "sum")
            res["t"] = g.ticket("x")
            g.barrier()
            g.close()
        th = threading.Thread(target=rank0)
        th.start()

        def connect():
            for _ in range(200):
                try:
                    return socket.create_connection(("127.0.0.1", port), timeout=5.0)
                except OSError:
                    time.sleep(0.05)
            raise AssertionError("rank 0 never listened")
        bad = connect()
        bad.sendall(struct_pack_q(1 << 20) + pickle.dumps(("coll", 0, "sum", None, np.ones(3))))     # the round-5 wire format
        silent = connect()                                                                            # says nothing
        wrong = connect()
        hostgroup._send_frame(wrong, hostgroup._HELLO, arg=1, raw=b"\x00" * 32)                       # wrong token
        time.sleep(0.3)
        g1 = hostgroup.HostGroup(1, 2, "127.0.0.1", port, timeout=60.0)
        s1 = g1.allreduce(np.array([10.0, 20.0]), "sum")
        t1 = g1.ticket("x")
        g1.barrier()
        g1.close()
        th.join(timeout=60.0)
        assert not th.is_alive()
        for s_ in (bad, silent, wrong):
            s_.close()
        assert np.array_equal(s1, [11.0, 22.0]) and np.array_equal(res["sum"], [11.0, 22.0]) and {t1, res["t"]} == {0, 1}
    finally:
        del os.environ["PYGPS_AMD_GROUP_SECRET"]


def struct_pack_q(n):
    import struct
    return struct.pack("<Q", n)


def test_hostgroup_frame_rejects_inconsistent_headers():
    from pygps_amd import hostgroup
    a, b = socket.socketpair()
    try:
        hostgroup._send_frame(a, hostgroup._COLL, 2, 7, 0, arr=np.arange(6, dtype=np.int64).reshape(2, 3))
        kind, op, seq, arg, arr, raw = hostgroup._recv_frame(b)
        assert (kind, op, seq) == (hostgroup._COLL, 2, 7) and arr.dtype == np.int64 and arr.shape == (2, 3) and arr[1, 2] == 5
        hdr = hostgroup._HDR.pack(b"PGHG", hostgroup._COLL, 0, 1, 1, 0, 0, 16, 3, 0, 0, 0, 0, 0)    # 3 doubles announced, 16 bytes of payload
        a.sendall(hdr + b"\x00" * 16)
        with pytest.raises(ConnectionError):
            hostgroup._recv_frame(b)
        a.sendall(hostgroup._HDR.pack(b"XXXX", 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0))
        with pytest.raises(ConnectionError):
            hostgroup._recv_frame(b)
    finally:
        a.close()
        b.close()

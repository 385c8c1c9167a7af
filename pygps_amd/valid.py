"""K-fold cross-validation helpers (reference: pyGPs/Validation/valid.py) and the fold loop sharded over GPUs.

``k_fold_validation`` (:20-47), ``k_fold_index`` (:50-66) and the metrics ``RMSE / ACC / Prec / Recall / NLPD`` (:70-146)
keep the reference's signatures and results.  ``sharded_k_fold`` is the loop that Demo/JHUI/demo_Validation.py:70-90 writes
by hand -- per fold: a fresh model, fit (optionally ``optimize``), ``predict`` on the held-out fold, metrics -- with fold f
on rank f % world (BASELINE north_star: "shards independent restarts / CV folds across the 8 GPUs"): every fit and predict
runs on the rank's GPU, two folds at a time per GPU on two fit streams, and ONE all-gather of K small records returns
every fold's numbers to every rank.  No collective touches the data path of a fit.

Reference quirks kept or stated:
* ``k_fold_validation(randomise=True)`` shuffles ``np.append(x, y, axis=1)`` with the GLOBAL numpy generator and hands
  ``y`` back one-dimensional (valid.py:34-38); so does this one.
* ``NLPD`` as written in the reference raises ``NameError`` (``log`` and ``math`` are never imported, valid.py:138); this
  one evaluates the formula of its docstring, ``mean(0.5 log(2 pi s2) + 0.5 (y - mu)^2 / s2)``.
"""
from copy import deepcopy

import numpy as np


def k_fold_validation(x, y, K=10, randomise=False):
    """Generates K (x_train, x_test, y_train, y_test) tuples; fold k holds the items with index % K == k."""
    if randomise:
        data = np.append(x, y, axis=1)
        np.random.shuffle(data)
        x = data[:, :-1]
        y = data[:, -1]
    n, D = x.shape
    assert n > K
    idx = np.arange(n)
    for k in range(K):
        test = idx % K == k
        yield np.array(x[~test]), np.array(x[test]), np.array(y[~test]), np.array(y[test])


def k_fold_index(n, K=10):
    """As k_fold_validation, but yields (indice_train, indice_test) lists only."""
    for k in range(K):
        yield [i for i in range(n) if i % K != k], [i for i in range(n) if i % K == k]


def RMSE(predict, target):
    error = predict - target
    return np.sqrt(np.mean(error ** 2))


def ACC(predict, target):
    n, D = target.shape
    return float(np.count_nonzero(np.asarray(predict)[:n, 0] == np.asarray(target)[:, 0])) / n


def Prec(predict, target):
    """Precision for class +1 (ZeroDivisionError when nothing is predicted +1, as the reference)."""
    p, t = np.asarray(predict)[:, 0], np.asarray(target)[:, 0]
    count_1 = float(np.count_nonzero(p == 1))
    count_2 = float(np.count_nonzero((p == 1) & (t == 1)))
    return count_2 / count_1


def Recall(predict, target):
    """Recall for class +1."""
    p, t = np.asarray(predict)[:, 0], np.asarray(target)[:, 0]
    count_1 = float(np.count_nonzero(t == 1))
    count_2 = float(np.count_nonzero((t == 1) & (p == 1)))
    return count_2 / count_1


def NLPD(y, MU, S2):
    """Negative log predictive density in the observation space."""
    return np.mean(0.5 * np.log(2 * np.pi * S2) + 0.5 * ((y - MU) ** 2) / S2)


# ---- the fold loop over GPUs ---------------------------------------------------------------------------------------------
_METRICS = {
    "RMSE": lambda ym, ys2, yt: RMSE(ym, yt),
    "NLPD": lambda ym, ys2, yt: NLPD(yt, ym, ys2),
    "ACC": lambda ym, ys2, yt: ACC(np.sign(ym), yt),
    "Prec": lambda ym, ys2, yt: Prec(np.sign(ym), yt),
    "Recall": lambda ym, ys2, yt: Recall(np.sign(ym), yt),
    "RMSE_class": lambda ym, ys2, yt: RMSE(np.sign(ym), yt),
}


def _one_fold(make_model, x, y, K, k, metrics, numIterations, set_data):
    idx = np.arange(x.shape[0])
    test = idx % K == k
    m = make_model()
    x_tr, y_tr, x_te, y_te = x[~test], y[~test], x[test], y[test]
    inf_ = getattr(m, "inffunc", None)
    for name in ("last_ttau", "last_tnu"):           # a fold never inherits EP warm-start state (cf. ShardedMinimize._cold_start)
        if hasattr(inf_, name):
            setattr(inf_, name, None)
    if set_data:                                      # the documented flow: setData (the default mean becomes Const(mean(y_train)),
        m.setData(x_tr, y_tr)                         # Core/gp.py:154-156), then optimize() / getPosterior()
        if numIterations:
            m.optimize(numIterations=numIterations)
            nlZ = m.nlZ
        else:
            nlZ = m.getPosterior()[0]
    elif numIterations:                               # Demo/JHUI/demo_Validation.py:75 literally: optimize(x_train, y_train) on a fresh model
        m.optimize(x_tr, y_tr, numIterations=numIterations)
        nlZ = m.nlZ
    else:
        nlZ = m.getPosterior(x_tr, y_tr)[0]
    ym, ys2, fm, fs2, lp = m.predict(x_te, ys=y_te)
    rec = [float(nlZ)]
    for name, fn in metrics:
        try:
            rec.append(float(fn(ym, ys2, y_te)))
        except ZeroDivisionError:
            rec.append(np.nan)
    return rec


def sharded_k_fold(model, x, y, K=10, metrics=("RMSE", "NLPD"), numIterations=0, group=None, streams_per_gpu=2,
                   deal="auto", set_data=True):
    """K-fold validation of ``model`` on (x, y), the folds sharded over the ranks of ``group``.

    model: a zero-argument callable that returns a fresh model (``lambda: pyGPs.GPR()``, as the demo builds one per fold),
           or a model instance used as a template (deep-copied per fold, before it has seen any fold's data).
    metrics: names from RMSE / NLPD / ACC / Prec / Recall / RMSE_class, or (name, callable(ym, ys2, y_test)) pairs.
    numIterations: 0 = fit at the model's hyper-parameters (``getPosterior``); > 0 = ``optimize(numIterations)`` first.
    set_data: True = ``setData(x_train, y_train)`` per fold before the fit (the default mean becomes Const(mean(y_train))); False =
           the data go in through ``optimize(x_train, y_train)`` / ``getPosterior(x_train, y_train)`` as the demo writes it (a fresh
           model keeps its Zero mean then).
    group / deal: as ``opt.ShardedMinimize`` (a torch process group, a ``hostgroup.HostGroup``, a ``sharded.Comm`` or None).
    Returns {"nlZ": (K,), <metric>: (K,), ..., "owner": (K,) rank that ran each fold}, identical on every rank.  The data are
    taken from rank 0 (one broadcast each for x and y)."""
    from . import _lib, sharded
    comm = group if isinstance(group, sharded.Comm) else sharded.search_comm(group)
    rank, world = comm.rank, comm.world
    make_model = model if callable(model) and not hasattr(model, "getPosterior") else (lambda: deepcopy(model))
    mets = [(m, _METRICS[m]) if isinstance(m, str) else (m[0], m[1]) for m in metrics]
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    x = comm.bcast(np.array(x), 0).reshape(x.shape)
    y = comm.bcast(np.array(y), 0).reshape(y.shape)
    if y.ndim == 1:
        y = y.reshape(-1, 1)
    n = x.shape[0]
    assert n > K
    W = 1 + len(mets)
    rec = np.full((K, W + 1), np.nan)
    rec[:, W] = 0.0                                   # "this rank ran it"
    seq = comm.search_seq = getattr(comm, "search_seq", 0) + 1
    dynamic = deal == "dynamic" or (deal == "auto" and K > world and world > 1)
    import threading
    lock = threading.Lock()
    if dynamic and comm.ticket("probe/%d" % seq) is not None:
        name = "fold/%d" % seq

        def take():
            with lock:
                t = comm.ticket(name)
            return t if t is not None and t < K else None
    else:
        todo = [k for k in range(K) if k % world == rank]

        def take():
            with lock:
                return todo.pop(0) if todo else None
    errors = []

    def work(slot, side_by_side=False):
        import contextlib
        with _lib.fit_stream(slot), (_lib.concurrent_fit_streams() if side_by_side else contextlib.nullcontext()):
            while True:
                k = take()
                if k is None:
                    return
                try:
                    rec[k, :W] = _one_fold(make_model, x, y, K, k, mets, numIterations, set_data)
                except Exception as e:                 # the fold is reported as failed (NaNs), the collective still completes
                    errors.append((k, e))
                rec[k, W] = 1.0
    S = max(1, int(streams_per_gpu))
    if S == 1:
        work(_lib.current_slot())
    else:
        ths = [threading.Thread(target=work, args=(s, True)) for s in range(S)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    full = comm.allgather(rec)                        # ONE all-gather: K x (2 + #metrics) doubles per rank
    ran = full[:, :, W]
    if not np.all(ran.sum(axis=0) == 1.0):
        raise RuntimeError("sharded_k_fold: folds %s were run by %s ranks" % (np.nonzero(ran.sum(axis=0) != 1.0)[0].tolist(),
                                                                              ran.sum(axis=0)[ran.sum(axis=0) != 1.0].tolist()))
    owner = np.argmax(ran, axis=0)
    out = {"nlZ": np.array([full[owner[k], k, 0] for k in range(K)]), "owner": owner}
    for j, (name, _) in enumerate(mets):
        out[name] = np.array([full[owner[k], k, 1 + j] for k in range(K)])
    if errors and not np.all(np.isfinite(out["nlZ"])):
        import logging
        logging.getLogger(__name__).warning("sharded_k_fold: %d fold(s) failed on rank %d: %r", len(errors), rank, errors[0][1])
    return out

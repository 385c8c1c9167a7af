"""A torch-free process group for the handful of tiny host collectives the sharded searches need.

The restart search (pyGPs/Core/opt.py:301-327 sharded over GPUs), the K-fold loop (pyGPs/Validation/valid.py:20-66) and
the RCCL rendezvous of ``sharded.Comm`` exchange a 128-byte id, one table, the data and one record per work item.  A pyGPs
user should not need ``torch`` for that: ``HostGroup`` is ~250 lines of sockets and fixed binary frames (no pickle: nothing received is ever evaluated) -- rank 0 listens on
``MASTER_ADDR:MASTER_PORT`` (the variables ``torchrun`` / any launcher exports), every other rank keeps ONE connection
to it, and rank 0 serves

* ``bcast`` / ``allreduce`` (sum, max) / ``allgather`` / ``barrier`` on numpy float64 / uint8 arrays (star topology:
  fine for kilobytes; the big collectives of the sharded fit go over RCCL inside the library), and
* ``ticket(name)``: an atomic counter -- restarts / folds are dealt to whichever rank asks next when there are more
  work items than ranks (line searches differ in length; a static deal leaves the fastest rank idle).

It serves two purposes: (i) the side channel that carries RCCL's unique id to ``pgp_comm_init_rccl``, (ii) the
call-backs of the library's host transport (``pgp_comm_init_host``), so that several ranks can share ONE GPU in tests.
No data-path traffic of a fit ever goes through it.
"""
import hashlib
import hmac
import os
import socket
import struct
import threading
import time

import numpy as np

# ---- wire format: fixed binary frames, nothing on the wire is ever evaluated ------------------------------------------
# header  <4s B B B B Q q Q 6Q> = magic, kind, op, dtype, ndim, seq, arg, nbytes, shape[6]   (84 bytes), then `nbytes` raw bytes
#   kind   HELLO (arg = rank, payload = the token)  COLL (op, arg = root, payload = the array)  TICKET (payload = the name, utf-8)
#          BYE  OK (COLL: the combined array; TICKET: arg = the value)  ERR (payload = the message, utf-8)
#   dtype  0 = no array (barrier, a non-root's bcast contribution), else an index into _DTYPES
# A frame that does not parse (magic, kind, dtype, ndim, a payload size that is not shape x itemsize or above the cap) closes
# the connection.  The hello is read under a timeout and carries HMAC-SHA256(secret, "pygps_amd.hostgroup|world|rank") when
# PYGPS_AMD_GROUP_SECRET is set (every rank must then share it); rank 0 binds the address the launcher exported (127.0.0.1
# for a single node).
_MAGIC = b"PGHG"
_HDR = struct.Struct("<4sBBBBQqQ6Q")
_HELLO, _COLL, _TICKET, _BYE, _OK, _ERR = 1, 2, 3, 4, 5, 6
_OPS = ("bcast", "barrier", "sum", "max", "gather")
_DTYPES = (None, np.dtype("<f8"), np.dtype("u1"), np.dtype("<i8"), np.dtype("<f4"), np.dtype("<i4"), np.dtype("<u8"))
_CODE = {dt.str: i for i, dt in enumerate(_DTYPES) if dt is not None}     # by name: `None == np.dtype("f8")` is True in numpy
_MAX_PAYLOAD = 1 << 32                                  # 4 GiB: far above X, y and any record table; a bound, not a feature


def _token(world, rank):
    secret = os.environ.get("PYGPS_AMD_GROUP_SECRET", "")
    if not secret:
        return b""
    return hmac.new(secret.encode(), b"pygps_amd.hostgroup|%d|%d" % (world, rank), hashlib.sha256).digest()


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("pygps_amd.hostgroup: peer closed the connection")
        buf.extend(chunk)
    return bytes(buf)


def _send_frame(sock, kind, op=0, seq=0, arg=0, arr=None, raw=b""):
    """One frame: an array (`arr`) or raw bytes (`raw`), never both."""
    if arr is not None:
        a = np.ascontiguousarray(arr)
        if a.dtype.str not in _CODE:
            a = a.astype(np.float64)                   # anything else numeric travels as float64 (bool / small ints are exact)
        if a.ndim > 6:
            raise ValueError("pygps_amd.hostgroup: at most 6 dimensions")
        code = _CODE[a.dtype.str]
        shape = tuple(a.shape) + (0,) * (6 - a.ndim)
        payload = a.tobytes()
        hdr = _HDR.pack(_MAGIC, kind, op, code, a.ndim, seq, int(arg), len(payload), *shape)
    else:
        payload = bytes(raw)
        hdr = _HDR.pack(_MAGIC, kind, op, 0, 0, seq, int(arg), len(payload), 0, 0, 0, 0, 0, 0)
    sock.sendall(hdr + payload)


def _recv_frame(sock):
    """(kind, op, seq, arg, array or None, raw bytes)."""
    magic, kind, op, code, ndim, seq, arg, nbytes, *shape = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if magic != _MAGIC or not _HELLO <= kind <= _ERR or code >= len(_DTYPES) or ndim > 6 or nbytes > _MAX_PAYLOAD or op >= len(_OPS):
        raise ConnectionError("pygps_amd.hostgroup: malformed frame")
    if code == 0:
        return kind, op, seq, arg, None, _recv_exact(sock, nbytes) if nbytes else b""
    dt = _DTYPES[code]
    count = 1
    for s_ in shape[:ndim]:
        count *= s_
    if count * dt.itemsize != nbytes:
        raise ConnectionError("pygps_amd.hostgroup: malformed frame (shape x itemsize != payload)")
    a = np.frombuffer(_recv_exact(sock, nbytes), dtype=dt).reshape(shape[:ndim]).copy()
    return kind, op, seq, arg, a, b""


class HostGroup(object):
    """One per process.  ``HostGroup.from_env()`` reads RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; the port used is
    MASTER_PORT + ``port_offset`` (default 17, env PYGPS_AMD_PORT_OFFSET) so that it never collides with a c10d store
    a launcher may already have bound to MASTER_PORT."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29517, timeout=300.0, op_timeout=86400.0):
        """timeout: the rendezvous; op_timeout: how long rank 0 waits inside one collective for the slowest rank (a rank may
        reach the final all-gather of a search long before the others: generous on purpose)."""
        self.rank, self.world = int(rank), int(world)
        self.addr, self.port, self.timeout, self.op_timeout = addr, int(port), float(timeout), float(op_timeout)
        self._seq = 0
        self._lock = threading.Lock()                   # one collective at a time per process
        self._closed = False
        if self.world == 1:
            self._tickets = {}
            return
        if self.rank == 0:
            self._tickets = {}
            self._state = {}                            # seq -> dict(parts={rank: payload}, result=..., left=int)
            self._cv = threading.Condition()
            self._srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self._srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self._srv.bind((addr, self.port))
            self._srv.listen(self.world)
            self._conns = []
            self._threads = []
            deadline = time.time() + self.timeout
            seen = set()
            while len(self._conns) < self.world - 1:
                self._srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = self._srv.accept()
                except socket.timeout:
                    raise TimeoutError("pygps_amd.hostgroup: %d of %d ranks connected to %s:%d within %.0f s"
                                       % (len(self._conns) + 1, self.world, addr, self.port, self.timeout))
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(max(0.1, min(10.0, deadline - time.time())))   # a connector that says nothing is dropped, not waited for
                try:
                    kind, _, _, r, _, tok = _recv_frame(c)
                except (ConnectionError, OSError, struct.error):
                    c.close()
                    continue
                if kind != _HELLO or r <= 0 or r >= self.world or r in seen or not hmac.compare_digest(tok, _token(self.world, r)):
                    c.close()                           # not one of ours (or a wrong secret): keep listening until the deadline
                    continue
                c.settimeout(None)
                seen.add(r)
                self._conns.append(c)
                th = threading.Thread(target=self._serve, args=(c, r), daemon=True)
                th.start()
                self._threads.append(th)
        else:
            deadline = time.time() + self.timeout
            while True:
                try:
                    s = socket.create_connection((addr, self.port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise TimeoutError("pygps_amd.hostgroup: rank %d cannot reach rank 0 at %s:%d"
                                           % (self.rank, addr, self.port))
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            _send_frame(s, _HELLO, arg=self.rank, raw=_token(self.world, self.rank))
            self._sock = s

    @classmethod
    def from_env(cls, timeout=300.0):
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("PYGPS_AMD_PORT_OFFSET", "17"))
        return cls(rank, world, addr, port, timeout)

    # ---- rank 0: the service ------------------------------------------------------------------------------------------
    @staticmethod
    def _combine(op, parts, world, arg):
        if op == "bcast":
            return parts[arg]
        if op == "barrier":
            return None
        arrs = [np.asarray(parts[r]) for r in range(world)]
        if op == "sum":
            out = arrs[0].copy()
            for a in arrs[1:]:                          # rank order: the result does not depend on arrival order
                out = out + a
            return out
        if op == "max":
            out = arrs[0].copy()
            for a in arrs[1:]:
                out = np.maximum(out, a)
            return out
        if op == "gather":
            return np.stack(arrs)
        raise ValueError(op)

    def _deposit(self, seq, op, arg, rank, payload):
        """Called on rank 0 for every rank's contribution (its own included); returns the combined result."""
        with self._cv:
            st = self._state.setdefault(seq, dict(parts={}, left=self.world, done=False, result=None, op=op))
            if st["op"] != op:
                raise RuntimeError("pygps_amd.hostgroup: collective mismatch at #%d: %s vs %s" % (seq, st["op"], op))
            st["parts"][rank] = payload
            if len(st["parts"]) == self.world:
                st["result"] = self._combine(op, st["parts"], self.world, arg)
                st["parts"] = None
                st["done"] = True
                self._cv.notify_all()
            else:
                ok = self._cv.wait_for(lambda: st["done"] or self._closed, timeout=self.op_timeout)
                if not ok or not st["done"]:
                    raise TimeoutError("pygps_amd.hostgroup: collective #%d (%s) timed out on rank 0" % (seq, op))
            res = st["result"]
            st["left"] -= 1
            if st["left"] == 0:
                del self._state[seq]
            return res

    def _ticket_local(self, name):
        with self._cv if self.world > 1 else self._lock:
            v = self._tickets.get(name, 0)
            self._tickets[name] = v + 1
            return v

    def _serve(self, conn, rank):
        try:
            while True:
                kind, op, seq, arg, payload, raw = _recv_frame(conn)
                if kind == _COLL:
                    try:
                        res = self._deposit(seq, _OPS[op], arg, rank, payload)
                        _send_frame(conn, _OK, op, seq, arr=res)
                    except Exception as e:             # the peer gets the error instead of a hang
                        _send_frame(conn, _ERR, op, seq, raw=repr(e).encode())
                elif kind == _TICKET:
                    _send_frame(conn, _OK, seq=seq, arg=self._ticket_local(raw.decode("utf-8", "replace")))
                elif kind == _BYE:
                    return
                else:
                    return                              # a peer never sends HELLO / OK / ERR here: drop the connection
        except (ConnectionError, OSError, EOFError, struct.error):
            return

    # ---- every rank ---------------------------------------------------------------------------------------------------
    def _collective(self, op, arg, payload):
        if self.world == 1:
            return self._combine(op, {0: payload}, 1, arg)
        with self._lock:
            seq = self._seq
            self._seq += 1
            if self.rank == 0:
                return self._deposit(seq, op, arg, 0, payload)
            _send_frame(self._sock, _COLL, _OPS.index(op), seq, 0 if arg is None else arg, arr=payload)
            kind, _, rseq, _, res, raw = _recv_frame(self._sock)
            if kind != _OK or rseq != seq:
                raise RuntimeError("pygps_amd.hostgroup: %s" % (raw.decode("utf-8", "replace") if kind == _ERR else "reply out of sequence"))
            return res

    def bcast(self, arr, root=0):
        """In place when ``arr`` is a writable numpy array; returns the array that holds the root's data."""
        a = np.asarray(arr)
        res = self._collective("bcast", int(root), a if self.rank == root else None)
        if self.rank != root:
            if isinstance(arr, np.ndarray) and arr.flags.writeable and arr.shape == res.shape:
                arr[...] = res
                return arr
            return np.array(res)
        return arr

    def allreduce(self, arr, op="sum"):
        a = np.asarray(arr)
        res = self._collective("max" if op in ("max", 1) else "sum", None, a)
        if isinstance(arr, np.ndarray) and arr.flags.writeable:
            arr[...] = res
            return arr
        return np.array(res)

    def allgather(self, arr):
        """(world,) + arr.shape, rank r's contribution at [r]."""
        return np.array(self._collective("gather", None, np.asarray(arr)))

    def barrier(self):
        self._collective("barrier", None, None)

    def ticket(self, name="default"):
        """The next value (0, 1, 2, ...) of the group-wide counter ``name``."""
        if self.world == 1 or self.rank == 0:
            return self._ticket_local(name)
        with self._lock:
            _send_frame(self._sock, _TICKET, raw=str(name).encode("utf-8"))
            kind, _, _, v, _, raw = _recv_frame(self._sock)
            if kind != _OK:
                raise RuntimeError("pygps_amd.hostgroup: %s" % raw.decode("utf-8", "replace"))
            return v

    def __deepcopy__(self, memo):
        return self

    def close(self):
        if self._closed:
            return
        self._closed = True
        if self.world == 1:
            return
        try:
            if self.rank == 0:
                for th in self._threads:               # the peers say "bye" when they close: replies still in flight
                    th.join(timeout=10.0)              # (the last barrier's) must reach them before the sockets go away
                with self._cv:
                    self._cv.notify_all()
                for c in self._conns:
                    try:
                        c.close()
                    except OSError:
                        pass
                self._srv.close()
            else:
                try:
                    _send_frame(self._sock, _BYE)
                except OSError:
                    pass
                self._sock.close()
        except Exception:
            pass

    def __del__(self):
        self.close()

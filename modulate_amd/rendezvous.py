"""Loopback control plane for bench.py's ranks: a barrier and a MAX / SUM of one float over N processes of ONE node.

The parts of an archive are independent cipher streams (Modulate/CArk.cpp:741-755, 849-897): nothing on the data path is
exchanged between GPUs, so the only thing the ranks of the bench ever say to each other is "I am here" and one number.  That
needs neither torch nor RCCL: rank 0 runs a coordinator thread on a loopback socket, every rank (rank 0 included) is a client,
and one round is "everybody sends (op, value), everybody gets the reduced value back".  Standard library only; nothing here
touches the GPU.

Addresses:  "tcp:127.0.0.1:PORT"  (bench.py's own launcher picks a free port), or  "unix:NAME"  -- an abstract Unix-domain
socket, used under torch.distributed.run, whose agent already occupies MASTER_PORT with its own store.
"""
import os
import socket
import struct
import threading
import time

_OPS = {"barrier": 0, "max": 1, "sum": 2}
_MSG = struct.Struct("<Id")  # op, value
_CONNECT_PATIENCE_S = 180.0  # a rank's first `import` on a fresh box can take minutes
_ROUND_PATIENCE_S = 900.0    # no round of the bench takes longer; a dead peer must not hang the others forever


def default_address(env=None):
    """The address every rank of one launch derives alike from its environment."""
    env = os.environ if env is None else env
    explicit = env.get("MODGPU_BENCH_RDZV")
    if explicit:
        return explicit
    # under torch.distributed.run: MASTER_PORT belongs to the agent's store, so meet on an abstract Unix socket named after it
    return f"unix:modgpu-bench-{env.get('MASTER_ADDR', '127.0.0.1')}-{env.get('MASTER_PORT', '29500')}-{env.get('TORCHELASTIC_RUN_ID', 'none')}"


def _parse(address):
    kind, _, rest = address.partition(":")
    if kind == "tcp":
        host, _, port = rest.rpartition(":")
        return socket.AF_INET, (host, int(port))
    if kind == "unix":
        return socket.AF_UNIX, "\0" + rest
    raise ValueError(f"rendezvous address {address!r}: expected tcp:HOST:PORT or unix:NAME")


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        got = conn.recv(n - len(buf))
        if not got:
            raise ConnectionError("a rank closed the rendezvous connection")
        buf += got
    return buf


def _coordinate(server, world):
    """Rank 0's coordinator: accept `world` clients, then serve rounds until the first client goes away."""
    conns = []
    try:
        server.settimeout(_CONNECT_PATIENCE_S)
        while len(conns) < world:
            c, _ = server.accept()
            c.settimeout(_ROUND_PATIENCE_S)
            if c.family == socket.AF_INET:
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conns.append(c)
        while True:
            msgs = [_MSG.unpack(_recv_exact(c, _MSG.size)) for c in conns]
            ops = {op for op, _ in msgs}
            if len(ops) != 1:
                raise RuntimeError(f"ranks disagree about the round's operation: {sorted(ops)}")
            op = ops.pop()
            vals = [v for _, v in msgs]
            out = max(vals) if op == _OPS["max"] else sum(vals) if op == _OPS["sum"] else 0.0
            reply = _MSG.pack(op, out)
            for c in conns:
                c.sendall(reply)
    except (OSError, ConnectionError, RuntimeError):
        pass  # a client left (the normal end, after close()) or died: closing every connection wakes the rest with an error
    finally:
        for c in conns:
            try:
                c.close()
            except OSError:
                pass
        server.close()


class LoopbackPlane:
    """barrier() / max(v) / sum(v) over `world` processes of one node."""

    name = "socket"

    def __init__(self, rank, world, address=None):
        self.rank, self.world = rank, world
        self.address = address or default_address()
        family, where = _parse(self.address)
        self._thread = None
        if rank == 0:
            server = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            server.bind(where)
            server.listen(world)
            self._thread = threading.Thread(target=_coordinate, args=(server, world), daemon=True, name="modgpu-rendezvous")
            self._thread.start()
        deadline = time.monotonic() + _CONNECT_PATIENCE_S
        while True:
            self._conn = socket.socket(family, socket.SOCK_STREAM)
            try:
                self._conn.connect(where)
                break
            except OSError:
                self._conn.close()
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {rank}: nobody listens at {self.address}")
                time.sleep(0.02)
        self._conn.settimeout(_ROUND_PATIENCE_S + _CONNECT_PATIENCE_S)
        if family == socket.AF_INET:
            self._conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    def _round(self, op, value):
        self._conn.sendall(_MSG.pack(_OPS[op], float(value)))
        got_op, out = _MSG.unpack(_recv_exact(self._conn, _MSG.size))
        if got_op != _OPS[op]:
            raise RuntimeError("rendezvous reply out of step")
        return out

    def barrier(self):
        self._round("barrier", 0.0)

    def max(self, value):
        return self._round("max", value)

    def sum(self, value):
        return self._round("sum", value)

    def close(self):
        try:
            self._conn.close()
        except OSError:
            pass
        if self._thread is not None:
            self._thread.join(timeout=5.0)

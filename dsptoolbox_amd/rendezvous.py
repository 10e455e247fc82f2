"""Host-side exchange between the ranks of one node: rank, world size and three small
collectives on byte strings (broadcast, all-gather, barrier).

The hot path has no exchange step; what has to travel between processes on the host is tiny --
the 128-byte RCCL id, array shapes, a few KB of results -- so the default transport is a plain
TCP star around rank 0 (standard library only).  The launcher's environment names the meeting
point: MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE as set by `python -m torch.distributed.run`
(the port used here is MASTER_PORT + DSPTOOLBOX_AMD_PORT_OFFSET, default 23, so it never collides
with the launcher's own store).  Anything with the same five methods can be installed instead
(`distributed.init(exchange=...)`), e.g. an object backed by an existing torch.distributed / MPI
group -- the package itself imports neither.
"""

from __future__ import annotations

import hashlib
import hmac
import os
import socket
import struct
import time


class Exchange:
    """Interface: rank, world, broadcast_bytes, allgather_bytes, barrier, close."""

    rank = 0
    world = 1

    def broadcast_bytes(self, data: bytes | None, src: int = 0) -> bytes:
        return data

    def allgather_bytes(self, data: bytes) -> list:
        return [data]

    def barrier(self) -> None:
        pass

    def close(self) -> None:
        pass


def _is_loopback(addr: str) -> bool:
    return addr in ("localhost", "::1") or addr.startswith("127.")


def job_token() -> bytes:
    """32 bytes every rank of ONE job derives alike and nobody else on the network can guess
    without the job's environment: DSPTOOLBOX_AMD_RDZV_TOKEN if the launcher exports one, else
    a digest of what torch.distributed.run hands to all its workers (run id, master address and
    port, world size) plus the user id."""
    explicit = os.environ.get("DSPTOOLBOX_AMD_RDZV_TOKEN")
    if explicit:
        seed = "explicit:" + explicit
    else:
        seed = "|".join(os.environ.get(k, "") for k in
                        ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE")) + f"|{os.getuid()}"
    return hashlib.sha256(("dsptoolbox_amd rendezvous:" + seed).encode()).digest()


def _send_msg(sock: socket.socket, data: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class TcpExchange(Exchange):
    """Star topology: rank 0 listens, every other rank holds one connection to it.  Collectives
    are sequences of length-prefixed messages; every rank must call them in the same order."""

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29523,
                 timeout_s: float = 120.0, token: bytes | None = None):
        assert 0 <= rank < world
        self.rank, self.world = rank, world
        self._peers = {}
        self._sock = None
        if world == 1:
            return
        if token is None:
            # the default token is a digest of the launcher's environment: good against cross-talk between jobs on one
            # node, guessable by anyone who knows that environment -- so an outside interface needs an explicit secret
            if not _is_loopback(addr) and not os.environ.get("DSPTOOLBOX_AMD_RDZV_TOKEN"):
                raise RuntimeError(f"rendezvous on a non-loopback address ({addr}) needs DSPTOOLBOX_AMD_RDZV_TOKEN "
                                   "(a secret shared by the ranks of this job)")
            token = job_token()
        if rank == 0:
            # Handshake, both ways: the server sends a fresh 16-byte challenge, the peer answers with its
            # rank, HMAC-SHA256(token, challenge || rank) and a nonce of its own; the server proves itself
            # with HMAC-SHA256(token, "rank0" || nonce).  A connection that fails it, names a rank outside
            # 1 .. world-1 or one that is already connected is dropped and the wait goes on.  (The channel
            # is authenticated at connect time only; what follows is plain TCP.)
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            # a single-node job (the launcher's MASTER_ADDR is a loopback name) never listens on
            # an outside interface
            srv.bind(("127.0.0.1" if _is_loopback(addr) else addr, port))
            srv.listen(world)
            deadline = time.time() + timeout_s
            try:
                while len(self._peers) < world - 1:
                    left = deadline - time.time()
                    if left <= 0:
                        raise TimeoutError(f"rendezvous: {world - 1 - len(self._peers)} rank(s) never arrived")
                    srv.settimeout(left)
                    conn, _ = srv.accept()
                    try:
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        conn.settimeout(min(10.0, timeout_s))
                        challenge = os.urandom(16)
                        conn.sendall(challenge)
                        raw = _recv_exact(conn, 4 + 32 + 16)
                        (r,) = struct.unpack_from("<I", raw, 0)
                        want = hmac.new(token, challenge + raw[:4], hashlib.sha256).digest()
                        if not hmac.compare_digest(want, raw[4:36]) or not (0 < r < world) or r in self._peers:
                            raise ConnectionError("rendezvous: handshake rejected")
                        conn.sendall(b"\x01" + hmac.new(token, b"rank0" + raw[36:], hashlib.sha256).digest())
                        conn.settimeout(timeout_s)
                        self._peers[r] = conn
                    except (OSError, ConnectionError, struct.error):
                        conn.close()
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout_s
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 not reachable at {addr}:{port}")
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout_s)
            challenge = _recv_exact(s, 16)
            me = struct.pack("<I", rank)
            nonce = os.urandom(16)
            s.sendall(me + hmac.new(token, challenge + me, hashlib.sha256).digest() + nonce)
            if _recv_exact(s, 1) != b"\x01":
                raise ConnectionError("rendezvous: rank 0 refused the handshake")
            if not hmac.compare_digest(_recv_exact(s, 32), hmac.new(token, b"rank0" + nonce, hashlib.sha256).digest()):
                raise ConnectionError("rendezvous: the listener at rank 0's address does not know this job's token")
            self._sock = s

    # -- collectives -------------------------------------------------------------------
    def allgather_bytes(self, data: bytes) -> list:
        if self.world == 1:
            return [data]
        if self.rank == 0:
            parts = [data] + [_recv_msg(self._peers[r]) for r in range(1, self.world)]
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for r in range(1, self.world):
                _send_msg(self._peers[r], blob)
            return parts
        _send_msg(self._sock, data)
        blob = _recv_msg(self._sock)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def broadcast_bytes(self, data: bytes | None, src: int = 0) -> bytes:
        if self.world == 1:
            return data
        # small payloads only: go through the gather so that any source rank works on a star
        parts = self.allgather_bytes(data if self.rank == src else b"")
        return parts[src]

    def barrier(self) -> None:
        self.allgather_bytes(b"")

    def close(self) -> None:
        for s in list(self._peers.values()) + ([self._sock] if self._sock else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._sock = {}, None


def from_environment(timeout_s: float | None = None) -> Exchange:
    """The launcher's rendezvous (see the module docstring); a single process gets the trivial
    exchange.  Time limit: DSPTOOLBOX_AMD_RDZV_TIMEOUT seconds (default 120)."""
    if timeout_s is None:
        timeout_s = float(os.environ.get("DSPTOOLBOX_AMD_RDZV_TIMEOUT", "120"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return Exchange()
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("DSPTOOLBOX_AMD_PORT_OFFSET", "23"))
    return TcpExchange(rank, world, addr, port, timeout_s)

"""A minimal rendezvous for one-process-per-GPU runs: plain TCP sockets, rank 0 in the middle.

The exchange between the shards of a structure is RCCL inside the library (``arp_comm_*``, ``arp_shard_exchange_*``); what a host
program has to provide is small: carry the 128-byte communicator id from rank 0 to the others once, put a barrier around a timed
region, add up a few numbers.  ``torch.distributed`` can do that, but a process that loads PyTorch-ROCm next to the library's
RCCL holds two HIP runtimes and two RCCLs; this module needs nothing but the standard library.

Environment (as ``python -m torch.distributed.run`` sets it): RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT.  The launcher's own
store listens on MASTER_PORT, so rank 0 listens on the first free port of MASTER_PORT + 1000 .. + 1007 (``ARP_RDZV_PORT``
overrides the first) and the others find it by the handshake.  Rank 0 binds MASTER_ADDR's interface only; a connection
that does not say hello within 3 s is dropped; the hello carries an optional token (``ARP_RDZV_TOKEN``, by default the launcher's
``TORCHELASTIC_RUN_ID``) so that a stray process of another launch cannot claim a rank; announced message lengths are capped.
"""
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b'ARPRDZV1'
_MAX_MESSAGE = 1 << 31       # a length prefix beyond this is not one of ours (the halo buffers of the host-buffer debug path stay far below)
_HELLO_TIMEOUT = 3.0         # a connection that does not say hello within this is dropped: it must not eat the rendezvous deadline


def _token() -> bytes:
    """Optional shared secret of one launch (ARP_RDZV_TOKEN, e.g. the launcher's run id): part of the hello both ways."""
    return os.environ.get('ARP_RDZV_TOKEN', os.environ.get('TORCHELASTIC_RUN_ID', '')).encode()[:64]


def _send(sock, data: bytes):
    sock.sendall(struct.pack('<Q', len(data)) + data)


def _recv_exact(sock, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _recv(sock, limit=_MAX_MESSAGE) -> bytes:
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError('rendezvous: message of %d bytes announced (limit %d)' % (n, limit))
    return _recv_exact(sock, n)


class TcpRendezvous:
    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=180.0):
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        if port is None:
            port = int(os.environ.get('ARP_RDZV_PORT', '0')) or (int(os.environ.get('MASTER_PORT', '29500')) + 1000)
        ports = [1024 + (int(port) + k - 1024) % (65536 - 1024) for k in range(8)]
        self.peers = {}          # rank 0: rank -> socket
        self.sock = None         # other ranks: the socket to rank 0
        if self.world == 1:
            return
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((self._bind_address(addr), p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError('rendezvous: no free port in %r' % (ports,))
            srv.listen(self.world)
            srv.settimeout(1.0)
            while len(self.peers) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError('rendezvous: %d of %d ranks arrived' % (len(self.peers) + 1, self.world))
                try:
                    s, _ = srv.accept()
                except socket.timeout:
                    continue
                s.settimeout(_HELLO_TIMEOUT)
                try:
                    hello = _recv(s, 256)
                    if hello[:8] != _MAGIC or hello[16:] != _token():
                        s.close()
                        continue
                    r, w = struct.unpack('<ii', hello[8:16])
                    if w != self.world or not (0 < r < self.world) or r in self.peers:
                        s.close()
                        continue
                    _send(s, _MAGIC + _token())
                    s.settimeout(timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.peers[r] = s
                except (OSError, struct.error, ConnectionError):
                    s.close()
            srv.close()
        else:
            k = 0
            while self.sock is None:
                if time.time() > deadline:
                    raise TimeoutError('rendezvous: rank 0 not found on ports %r of %s' % (ports, addr))
                p = ports[k % len(ports)]
                k += 1
                try:
                    s = socket.create_connection((addr, p), timeout=2.0)
                    s.settimeout(5.0)
                    _send(s, _MAGIC + struct.pack('<ii', self.rank, self.world) + _token())
                    if _recv(s, 256) != _MAGIC + _token():
                        raise ConnectionError('not the rendezvous')
                    s.settimeout(timeout)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self.sock = s
                except (OSError, ConnectionError, struct.error):
                    try:
                        s.close()
                    except Exception:        # noqa: BLE001
                        pass
                    if k % len(ports) == 0:
                        time.sleep(0.2)

    @staticmethod
    def _bind_address(addr):
        """Rank 0 listens on MASTER_ADDR's own interface (the loopback for 127.0.0.1), not on every interface; a name that does
        not resolve to a local address (a NAT'ed or virtual master address) falls back to all interfaces."""
        if addr in ('127.0.0.1', 'localhost'):
            return '127.0.0.1'
        try:
            ip = socket.gethostbyname(addr)
            probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                probe.bind((ip, 0))
            finally:
                probe.close()
            return ip
        except OSError:
            return ''

    # ---- collectives over the star --------------------------------------------------------------------------
    def gather(self, data: bytes):
        """Rank 0 gets the list of every rank's bytes (by rank); the others get None."""
        if self.world == 1:
            return [data]
        if self.rank == 0:
            out = [data] + [None] * (self.world - 1)
            for r, s in self.peers.items():
                out[r] = _recv(s)
            return out
        _send(self.sock, data)
        return None

    def broadcast(self, data) -> bytes:
        """Everybody gets rank 0's bytes."""
        if self.world == 1:
            return data
        if self.rank == 0:
            for s in self.peers.values():
                _send(s, data)
            return data
        return _recv(self.sock)

    def allgather(self, data: bytes):
        got = self.gather(data)
        if self.world == 1:
            return got
        if self.rank == 0:
            blob = struct.pack('<i', len(got)) + b''.join(struct.pack('<Q', len(g)) + g for g in got)
            self.broadcast(blob)
            return got
        blob = self.broadcast(None)
        (n,) = struct.unpack('<i', blob[:4])
        out, off = [], 4
        for _ in range(n):
            (m,) = struct.unpack('<Q', blob[off:off + 8])
            out.append(blob[off + 8:off + 8 + m])
            off += 8 + m
        return out

    def barrier(self):
        self.allgather(b'')

    def allreduce(self, a, op='max'):
        a = np.ascontiguousarray(a)
        parts = [np.frombuffer(b, a.dtype).reshape(a.shape) for b in self.allgather(a.tobytes())]
        red = {'max': np.maximum.reduce, 'sum': np.add.reduce, 'min': np.minimum.reduce}[op]
        return red(np.stack(parts), axis=0)

    def allreduce_max(self, a):      # (the transport interface of arpeggio_amd.sharding)
        return self.allreduce(a, 'max')

    def exchange(self, payload):
        """The transport interface of arpeggio_amd.sharding: payload = {-1 / +1: uint8 array for that slab neighbour};
        returns what the neighbours sent to this rank, keyed by the side it came from.  (Host buffers, through rank 0: the
        debug path; device buffers go over RCCL.)"""
        mine = b''.join(struct.pack('<iQ', s_, int(np.asarray(payload[s_]).size)) + np.ascontiguousarray(payload[s_], np.uint8).tobytes()
                        for s_ in (-1, +1) if s_ in payload and 0 <= self.rank + s_ < self.world)
        out = {}
        for r, blob in enumerate(self.allgather(mine)):
            off = 0
            while off < len(blob):
                side, n = struct.unpack('<iQ', blob[off:off + 12])
                if r + side == self.rank and n:
                    out[-side] = np.frombuffer(blob[off + 12:off + 12 + n], np.uint8).copy()
                off += 12 + n
        return out

    def close(self):
        for s in list(self.peers.values()) + ([self.sock] if self.sock else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None

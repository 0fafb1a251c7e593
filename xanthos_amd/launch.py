"""One process per GPU without a framework: a launcher that starts the rank processes and a small TCP rendezvous.

The reference has no distributed path; this package's multi-GPU mode (basins sharded over the GPUs of a node, one RCCL gather
at write-out: ``dist.py``) needs very little from a process group -- the ranks have to agree on a 128-byte RCCL id, on a
flag or two (min over ranks) and on a clock (max over ranks, barrier) -- so it carries its own:

* ``spawn(n, argv)``: starts ``n`` rank processes (plain children of this process, started BEFORE anything here has
  touched a GPU; never an exec) with ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT`` in their environment,
  the way ``torch.distributed.run`` would, relays their output and returns the worst exit code;
* ``SocketGroup``: the ranks of one job, a star over TCP through rank 0 on ``MASTER_ADDR:MASTER_PORT`` -- ``bcast``,
  ``allreduce`` (min / max / sum of a number), ``gather`` (picklable objects or raw bytes to one rank), ``barrier``.
  Everything that moves real data goes over RCCL (``csrc/xh_comm.hip``); the group only ever carries a few hundred bytes,
  except in the host fall-back of ``dist.OutputGather`` (RCCL unavailable, e.g. two test ranks on one GPU).
* ``current_group()``: the group of this process, made on first use from the environment; ``None`` in a one-rank job.

Any launcher that sets the five variables works (``torchrun`` too -- but it keeps its own store on ``MASTER_PORT``, so under
it set ``XH_MASTER_PORT`` to a free port for this group).
"""
import os
import pickle
import socket
import struct
import subprocess
import sys
import time

_GROUP = None
_HDR = struct.Struct('<Q')


def _send(sock, payload):
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('peer closed the rendezvous connection')
        got += k
    return bytes(buf)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return _recv_exact(sock, n)


class SocketGroup:
    """The ranks of one job.  Rank 0 listens; every other rank holds one connection to it.  Collectives are called by all
    ranks in the same order (like any process group); each is one round trip through rank 0."""

    def __init__(self, rank, size, addr='127.0.0.1', port=29400, timeout=120.0):
        self.rank, self.size = int(rank), int(size)
        self.peers = {}
        self.sock = None
        if self.size <= 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ('localhost',) else '127.0.0.1', int(port)))
            srv.listen(self.size)
            srv.settimeout(timeout)
            try:
                while len(self.peers) < self.size - 1:
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(None)
                    r = pickle.loads(_recv(conn))
                    if not (isinstance(r, int) and 0 < r < self.size) or r in self.peers:
                        conn.close()
                        raise RuntimeError('rendezvous: unexpected rank {!r}'.format(r))
                    self.peers[r] = conn
            finally:
                srv.close()
        else:
            t_end = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, int(port)), timeout=5.0)
                    break
                except OSError:
                    if time.time() > t_end:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            _send(s, pickle.dumps(self.rank))
            self.sock = s

    # ---- collectives (all ranks, same order)
    def gather(self, obj, root=0, raw=False):
        """Every rank's ``obj`` on ``root`` as a list indexed by rank (None elsewhere).  ``raw``: ``obj`` is bytes-like and
        travels as it is (the host fall-back of the write-out gather)."""
        if self.size <= 1:
            return [obj]
        enc = (lambda o: bytes(o)) if raw else pickle.dumps
        dec = (lambda b: b) if raw else pickle.loads
        if self.rank == 0:
            got = [obj] + [None] * (self.size - 1)
            for r, conn in self.peers.items():
                got[r] = dec(_recv(conn))
            if root == 0:
                return got
            _send(self.peers[root], pickle.dumps(got))
            return None
        _send(self.sock, enc(obj))
        if self.rank == root:
            return pickle.loads(_recv(self.sock))
        return None

    def bcast(self, obj, src=0):
        """``obj`` of rank ``src`` on every rank."""
        if self.size <= 1:
            return obj
        if src != 0:                                    # through rank 0
            got = self.gather(obj if self.rank == src else None, root=0)
            obj = got[src] if self.rank == 0 else None
        if self.rank == 0:
            payload = pickle.dumps(obj)
            for conn in self.peers.values():
                _send(conn, payload)
            return obj
        return pickle.loads(_recv(self.sock))

    def allreduce(self, x, op='max'):
        """min / max / sum of one number over the ranks, on every rank."""
        if self.size <= 1:
            return x
        got = self.gather(x, root=0)
        if self.rank == 0:
            x = {'max': max, 'min': min, 'sum': sum}[op](got)
        return self.bcast(x, src=0)

    def barrier(self):
        self.allreduce(0, 'max')

    def close(self):
        for c in list(self.peers.values()) + ([self.sock] if self.sock is not None else []):
            try:
                c.close()
            except OSError:
                pass
        self.peers, self.sock = {}, None


def env_world():
    """(rank, local rank, world size) of this process as a launcher set them; (0, 0, 1) without one."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0'))),
            int(os.environ.get('WORLD_SIZE', '1')))


def current_group():
    """The SocketGroup of this job (made once, on first use); None when the job has one rank."""
    global _GROUP
    rank, _, size = env_world()
    if size <= 1:
        return None
    if _GROUP is None:
        port = int(os.environ.get('XH_MASTER_PORT', os.environ.get('MASTER_PORT', '29400')))
        _GROUP = SocketGroup(rank, size, os.environ.get('MASTER_ADDR', '127.0.0.1'), port)
    return _GROUP


def close_group():
    global _GROUP
    if _GROUP is not None:
        _GROUP.close()
        _GROUP = None


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn(n, argv, env=None, one_device=False):
    """Start ``n`` rank processes running ``argv`` (a list; ``sys.executable`` is put in front when it starts with '-' or
    ends in '.py'), wait for them, return the worst exit code.  Children are started before this process touches a GPU and
    it never does afterwards; rank 0's output passes through, the other ranks' output is prefixed.  ``one_device``: every
    rank on GPU 0 (tests on a one-GPU box)."""
    if argv and (argv[0].startswith('-') or argv[0].endswith('.py')):
        argv = [sys.executable] + list(argv)
    base = dict(os.environ if env is None else env)
    base.setdefault('MASTER_ADDR', '127.0.0.1')
    base['MASTER_PORT'] = str(base.get('XH_MASTER_PORT') or free_port())
    base.pop('XH_MASTER_PORT', None)
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # RCCL across processes needs dmabuf IPC on this image
    import tempfile
    procs, logs = [], []
    for r in range(n):
        e = dict(base)
        e.update({'RANK': str(r), 'LOCAL_RANK': '0' if one_device else str(r), 'WORLD_SIZE': str(n)})
        # (the other ranks' output goes to a file and is shown when they have ended: a pipe nobody reads could block a rank
        # inside a collective the others are waiting in)
        log = None if r == 0 else tempfile.TemporaryFile(mode='w+')
        logs.append(log)
        procs.append(subprocess.Popen(list(argv), env=e, stdout=log, stderr=None if r == 0 else subprocess.STDOUT))
    worst = 0
    for r, p in enumerate(procs):
        p.wait()
        if logs[r] is not None:
            logs[r].seek(0)
            for line in logs[r].read().splitlines():
                print('[rank {}] {}'.format(r, line), flush=True)
            logs[r].close()
        if p.returncode != 0 and worst == 0:
            worst = p.returncode
    return worst

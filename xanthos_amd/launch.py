"""One process per GPU without a framework: a launcher that starts the rank processes and a small TCP rendezvous.

The reference has no distributed path; this package's multi-GPU mode (basins sharded over the GPUs of a node, one RCCL gather
at write-out: ``dist.py``) needs very little from a process group -- the ranks have to agree on a 128-byte RCCL id, on a
flag or two (min over ranks) and on a clock (max over ranks, barrier) -- so it carries its own:

* ``spawn(n, argv)``: starts ``n`` rank processes (plain children of this process, started BEFORE anything here has
  touched a GPU; never an exec) with ``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT`` in their environment,
  the way ``torch.distributed.run`` would, relays their output and returns the worst exit code;
* ``SocketGroup``: the ranks of one job, a star over TCP through rank 0 on ``MASTER_ADDR:MASTER_PORT`` -- ``bcast``,
  ``allreduce`` (min / max / sum of a number), ``gather`` (numbers, byte strings, lists, float64 arrays -- a small typed
  encoding, nothing is ever unpickled -- or raw bytes to one rank), ``barrier``.  A peer must present the job's token
  (``XH_JOB_TOKEN``, drawn by ``spawn``) in a fixed-size hello; message lengths are capped; waits are bounded.
  Everything that moves real data goes over RCCL (``csrc/xh_comm.hip``); the group only ever carries a few hundred bytes,
  except in the host fall-back of ``dist.OutputGather`` (RCCL unavailable, e.g. two test ranks on one GPU).
* ``current_group()``: the group of this process, made on first use from the environment; ``None`` in a one-rank job.

Any launcher that sets the five variables works (``torchrun`` too -- but it keeps its own store on ``MASTER_PORT``, so under
it set ``XH_MASTER_PORT`` to a free port for this group, and ``XH_JOB_TOKEN`` to a secret of the job).
"""
import hmac
import os
import socket
import struct
import subprocess
import sys
import time

import numpy as np

_GROUP = None
_HDR = struct.Struct('<Q')
_HELLO = struct.Struct('<4sI64s')                 # magic, rank, job token (zero-padded)
_MAGIC = b'XHG1'
# A rank is a process of THIS job: it knows the job token (spawn() draws one and hands it down in XH_JOB_TOKEN; under another
# launcher set it yourself -- without one the hello is accepted on the rank number alone, which is only as private as the
# port).  Nothing is ever unpickled: the group carries numbers, byte strings, lists of them and float64 arrays, in a small
# typed encoding; a hello is a fixed 72-byte record, and no message may be longer than XH_GROUP_MAX_MESSAGE bytes (4 GiB).
_MAX_MESSAGE = int(os.environ.get('XH_GROUP_MAX_MESSAGE', str(1 << 32)))
_TIMEOUT = float(os.environ.get('XH_GROUP_TIMEOUT', '900'))      # seconds a rank waits inside a collective before giving up


def _token():
    return os.environ.get('XH_JOB_TOKEN', '').encode()[:64].ljust(64, b'\0')


def _encode(obj):
    """None, bool / int, float, bytes-like, str, list / tuple / dict (string keys) of those, float64 arrays -> bytes (no pickle)."""
    if obj is None:
        return b'N'
    if isinstance(obj, (bool, int, np.integer)):
        return b'I' + struct.pack('<q', int(obj))
    if isinstance(obj, (float, np.floating)):
        return b'F' + struct.pack('<d', float(obj))
    if isinstance(obj, (bytes, bytearray, memoryview)):
        b = bytes(obj)
        return b'B' + _HDR.pack(len(b)) + b
    if isinstance(obj, str):
        b = obj.encode()
        return b'S' + _HDR.pack(len(b)) + b
    if isinstance(obj, np.ndarray):
        a = np.ascontiguousarray(obj, dtype=np.float64)
        return b'A' + struct.pack('<I', a.ndim) + struct.pack('<%dq' % a.ndim, *a.shape) + a.tobytes()
    if isinstance(obj, (list, tuple)):
        return b'L' + struct.pack('<I', len(obj)) + b''.join(_encode(x) for x in obj)
    if isinstance(obj, dict):
        return b'D' + struct.pack('<I', len(obj)) + b''.join(_encode(str(k)) + _encode(v) for k, v in obj.items())
    raise TypeError('the process group carries numbers, bytes, strings, lists and float64 arrays, not {}'.format(type(obj).__name__))


def _decode(buf, pos=0):
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b'N':
        return None, pos
    if tag == b'I':
        return struct.unpack_from('<q', buf, pos)[0], pos + 8
    if tag == b'F':
        return struct.unpack_from('<d', buf, pos)[0], pos + 8
    if tag in (b'B', b'S'):
        (n,) = _HDR.unpack_from(buf, pos)
        pos += _HDR.size
        if n > len(buf) - pos:
            raise ValueError('process group: truncated message')
        b = bytes(buf[pos:pos + n])
        return (b if tag == b'B' else b.decode()), pos + n
    if tag == b'A':
        (nd,) = struct.unpack_from('<I', buf, pos)
        pos += 4
        if nd > 8:
            raise ValueError('process group: bad array header')
        shape = struct.unpack_from('<%dq' % nd, buf, pos)
        pos += 8 * nd
        n = int(np.prod(shape, dtype=np.int64)) if nd else 1
        if min(shape, default=0) < 0 or n * 8 > len(buf) - pos:
            raise ValueError('process group: truncated array')
        return np.frombuffer(buf, dtype=np.float64, count=n, offset=pos).reshape(shape).copy(), pos + 8 * n
    if tag == b'L':
        (n,) = struct.unpack_from('<I', buf, pos)
        pos += 4
        out = []
        for _ in range(n):
            x, pos = _decode(buf, pos)
            out.append(x)
        return out, pos
    if tag == b'D':
        (n,) = struct.unpack_from('<I', buf, pos)
        pos += 4
        out = {}
        for _ in range(n):
            k, pos = _decode(buf, pos)
            out[k], pos = _decode(buf, pos)
        return out, pos
    raise ValueError('process group: unknown tag {!r}'.format(tag))


def _loads(buf):
    obj, pos = _decode(buf)
    if pos != len(buf):
        raise ValueError('process group: trailing bytes')
    return obj


def _send(sock, payload):
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv_exact(sock, n):
    # (grown as the bytes arrive, 16 MiB at a time: a length word alone never allocates anything)
    chunks, got = [], 0
    while got < n:
        buf = bytearray(min(n - got, 1 << 24))
        view, k0 = memoryview(buf), 0
        while k0 < len(buf):
            k = sock.recv_into(view[k0:], len(buf) - k0)
            if k == 0:
                raise ConnectionError('peer closed the rendezvous connection')
            k0 += k
        chunks.append(buf)
        got += len(buf)
    return bytes(chunks[0]) if len(chunks) == 1 else b''.join(chunks)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > _MAX_MESSAGE:
        raise ConnectionError('process group: a message of {} bytes exceeds XH_GROUP_MAX_MESSAGE'.format(n))
    return _recv_exact(sock, n)


class SocketGroup:
    """The ranks of one job.  Rank 0 listens; every other rank holds one connection to it.  Collectives are called by all
    ranks in the same order (like any process group); each is one round trip through rank 0.  A rank that waits longer than
    XH_GROUP_TIMEOUT seconds (900) for a peer gives up with an error instead of hanging behind a rank that died."""

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
            t_end = time.time() + timeout
            try:
                while len(self.peers) < self.size - 1:
                    srv.settimeout(max(t_end - time.time(), 0.01))
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(10.0)
                    try:                      # a fixed-size hello: magic, rank, job token -- anything else is not a rank of this job
                        magic, r, tok = _HELLO.unpack(_recv_exact(conn, _HELLO.size))
                        good = magic == _MAGIC and 0 < r < self.size and r not in self.peers and hmac.compare_digest(tok, _token())
                    except (OSError, struct.error):
                        good = False
                    if not good:
                        conn.close()          # (a stranger on the port costs the job nothing but this connection)
                        continue
                    conn.settimeout(_TIMEOUT)
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
            s.settimeout(_TIMEOUT)
            s.sendall(_HELLO.pack(_MAGIC, self.rank, _token()))
            self.sock = s

    # ---- collectives (all ranks, same order)
    def gather(self, obj, root=0, raw=False):
        """Every rank's ``obj`` on ``root`` as a list indexed by rank (None elsewhere).  ``raw``: ``obj`` is bytes-like and
        travels as it is (the host fall-back of the write-out gather)."""
        if self.size <= 1:
            return [obj]
        enc = (lambda o: bytes(o)) if raw else _encode
        dec = (lambda b: b) if raw else _loads
        if self.rank == 0:
            got = [enc(obj) if raw else obj] + [None] * (self.size - 1)
            for r, conn in self.peers.items():
                got[r] = dec(_recv(conn))
            if root == 0:
                return got
            _send(self.peers[root], _encode(got))
            return None
        _send(self.sock, enc(obj))
        if self.rank == root:
            return _loads(_recv(self.sock))
        return None

    def bcast(self, obj, src=0):
        """``obj`` of rank ``src`` on every rank."""
        if self.size <= 1:
            return obj
        if src != 0:                                    # through rank 0
            got = self.gather(obj if self.rank == src else None, root=0)
            obj = got[src] if self.rank == 0 else None
        if self.rank == 0:
            payload = _encode(obj)
            for conn in self.peers.values():
                _send(conn, payload)
            return obj
        return _loads(_recv(self.sock))

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
    if not base.get('XH_JOB_TOKEN'):                             # what makes a process a rank of THIS job (SocketGroup's hello)
        import secrets
        base['XH_JOB_TOKEN'] = secrets.token_hex(24)
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
    # Supervise ALL ranks: the first one that ends badly takes the job with it -- the others sit in a collective (or an RCCL
    # gather) that can never complete, so they get `grace` seconds and are then terminated; the job returns the first failure
    # noticed (ranks that end within one poll are looked at in rank order).
    worst, failed_at = 0, None
    grace = float(base.get('XH_SPAWN_GRACE', '20'))
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.discard(r)
            if rc != 0 and worst == 0:
                worst, failed_at = rc, time.time()
        if failed_at is not None and alive and time.time() - failed_at > grace:
            for r in alive:
                procs[r].terminate()
            t_kill = time.time() + 5.0
            for r in sorted(alive):
                try:
                    procs[r].wait(timeout=max(t_kill - time.time(), 0.1))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
            alive.clear()
        elif alive:
            time.sleep(0.05)
    for r in range(n):
        if logs[r] is not None:
            logs[r].seek(0)
            for line in logs[r].read().splitlines():
                print('[rank {}] {}'.format(r, line), flush=True)
            logs[r].close()
    return worst

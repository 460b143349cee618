"""Launcher plumbing for one-process-per-GPU runs, standard library only (no torch, no MPI).

The reference has no parallelism (SURVEY.md sections 2, 5); the multi-GPU path of this build is one
process per GPU with RCCL between them (wavelets_amd/parallel.py).  What the processes need from a
launcher is small: start N of them, hand every rank the 128-byte RCCL unique id of rank 0, a
barrier, a MAX over ranks of one float, and a wall-clock limit.  Until round 3 that was
torch.distributed (gloo) - which made every rank import torch and with it torch's bundled ROCm
runtime, so the N > 1 numbers ran on a different HIP stack than the N = 1 ones.  This module does
the same with sockets:

  * ``SocketGroup(rank, world)``: a star over ONE abstract AF_UNIX socket (single node, Linux; no
    file, nothing to clean up, nothing stale).  Rank 0 listens, the others connect and announce
    their rank.  ``bcast`` / ``gather`` / ``allreduce`` / ``barrier`` are length-prefixed pickles
    through rank 0.  The socket's name comes from the environment: ``WT_RDZV`` (set by ``spawn``
    below) or, under ``python -m torch.distributed.run`` - the driver's launcher; torch then lives
    in the launcher process only - from ``MASTER_PORT``, which is unique per job on a host.
  * ``spawn(nranks, argv, ...)``: start the ranks as child processes (RANK / LOCAL_RANK /
    WORLD_SIZE / WT_RDZV in their environment, one process group each), forward their output, watch
    the clock: when a rank fails or the limit passes every rank's process group is killed and the
    caller gets a non-zero code - a hung ncclCommInitRank cannot eat a GPU lease.
  * ``Watchdog(limit, on_expire)``: the same limit from inside a rank (a daemon thread: the main
    thread may sit in a C call that never returns), for ranks started by somebody else's launcher.
"""
import os
import pickle
import signal
import socket
import struct
import subprocess
import sys
import threading
import time

__all__ = ["SocketGroup", "spawn", "Watchdog", "group_from_env"]


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection (a rank died?)")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


def _check_peer(sock):
    """Abstract sockets carry no file permissions: any local user could connect (or bind the name
    first) and feed pickles to a rank.  SO_PEERCRED names the peer's uid as the kernel saw it at
    connect() / listen() time; anything but our own uid is refused before a byte is read."""
    cred = sock.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
    _pid, uid, _gid = struct.unpack("3i", cred)
    if uid != os.getuid():
        sock.close()
        raise PermissionError(f"rendezvous: peer runs as uid {uid}, this job as uid {os.getuid()}")


def default_name(env=None):
    """Name of the rendezvous socket of this job (see the module docstring)."""
    env = os.environ if env is None else env
    if env.get("WT_RDZV"):
        return env["WT_RDZV"]
    if env.get("MASTER_PORT"):
        return f"wt-rdzv-{os.getuid()}-{env.get('MASTER_ADDR', '127.0.0.1')}-{env['MASTER_PORT']}"
    raise RuntimeError("no rendezvous name: neither WT_RDZV nor MASTER_PORT is set "
                       "(start the ranks with wavelets_amd.launch.spawn or torch.distributed.run)")


class SocketGroup:
    """The ranks of one job, connected through rank 0 (see the module docstring).

    connect_timeout: seconds a rank waits for the others to show up; op_timeout: seconds any later
    operation may block (a dead peer then raises instead of hanging)."""

    def __init__(self, rank, world, name=None, connect_timeout=120.0, op_timeout=900.0):
        if world < 1 or not 0 <= rank < world:
            raise ValueError(f"bad rank {rank} of {world}")
        self.rank, self.world = rank, world
        self._peers, self._up = {}, None
        if world == 1:
            return
        addr = "\0" + (name or default_name())
        deadline = time.monotonic() + connect_timeout
        if rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(addr)
            srv.listen(world)
            try:
                while len(self._peers) < world - 1:
                    srv.settimeout(max(0.05, deadline - time.monotonic()))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        raise TimeoutError(f"rendezvous: {world - 1 - len(self._peers)} of {world - 1} ranks did "
                                           f"not connect within {connect_timeout:.0f} s") from None
                    conn.settimeout(op_timeout)
                    _check_peer(conn)
                    r = _recv(conn)
                    if not isinstance(r, int) or not 0 < r < world or r in self._peers:
                        conn.close()
                        raise RuntimeError(f"rendezvous: unexpected rank announcement {r!r}")
                    self._peers[r] = conn
            finally:
                srv.close()
            for r in sorted(self._peers):               # everyone is here: let them go
                _send(self._peers[r], world)
        else:
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 did not listen within {connect_timeout:.0f} s") from None
                    time.sleep(0.02)
            s.settimeout(max(op_timeout, connect_timeout))
            _check_peer(s)
            _send(s, rank)
            if _recv(s) != world:
                raise RuntimeError("rendezvous: the ranks disagree about the world size")
            s.settimeout(op_timeout)
            self._up = s

    # -- collectives (all through rank 0; every rank must call them in the same order) -------------
    def gather(self, obj):
        """rank 0: [obj of rank 0, obj of rank 1, ...]; other ranks: None."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(self._peers[r]) for r in range(1, self.world)]
        _send(self._up, obj)
        return None

    def bcast(self, obj, src=0):
        """Every rank gets rank `src`'s object."""
        if self.world == 1:
            return obj
        if self.rank == 0:
            if src != 0:
                obj = _recv(self._peers[src])
            for r in range(1, self.world):
                _send(self._peers[r], obj)
            return obj
        if self.rank == src:
            _send(self._up, obj)
        return _recv(self._up)

    def allreduce(self, value, op=max):
        """op over the ranks' values (op takes an iterable: max, min, sum, all, any, ...)."""
        vals = self.gather(value)
        return self.bcast(op(vals) if self.rank == 0 else None)

    def barrier(self):
        self.allreduce(0)

    def close(self):
        for s in list(self._peers.values()) + ([self._up] if self._up else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._up = {}, None


def group_from_env(connect_timeout=120.0, op_timeout=900.0):
    """The SocketGroup of this process from RANK / WORLD_SIZE (+ WT_RDZV or MASTER_PORT)."""
    return SocketGroup(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                       connect_timeout=connect_timeout, op_timeout=op_timeout)


def _kill_group(proc, sig):
    try:
        os.killpg(proc.pid, sig)          # start_new_session: the child leads its own group
    except (ProcessLookupError, PermissionError):
        pass


def spawn(nranks, argv, time_limit=900.0, env=None, grace=5.0, tee_rank0=None):
    """Start `nranks` copies of `argv` (rank r with RANK = LOCAL_RANK = r), wait for them.

    Returns (returncode, reason): 0 / None when every rank exited with 0; otherwise the first
    failing rank's code (124 for the time limit) and a one-line reason.  On a failure or at the
    limit every rank's process group gets SIGTERM, then SIGKILL after `grace` seconds.  The
    children inherit stdout / stderr; with `tee_rank0` (a list) rank 0's stdout goes through a pipe
    instead: every line is forwarded to this process's stdout as it arrives and appended to the
    list, so that the caller knows what rank 0 has reported."""
    base = dict(os.environ if env is None else env)
    name = f"wt-rdzv-{os.getuid()}-{os.getpid()}-{time.monotonic_ns()}"
    procs = []
    for r in range(nranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), WT_RDZV=name)
        out = subprocess.PIPE if (r == 0 and tee_rank0 is not None) else None
        procs.append(subprocess.Popen(list(argv), env=e, start_new_session=True, stdout=out))
    reader = None
    if tee_rank0 is not None:
        def pump():
            for raw in procs[0].stdout:
                line = raw.decode("utf-8", "replace")
                tee_rank0.append(line)
                sys.stdout.write(line)
                sys.stdout.flush()
        reader = threading.Thread(target=pump, name="wt-rank0-stdout", daemon=True)
        reader.start()
    t0 = time.monotonic()
    rc, reason = 0, None
    live = set(range(nranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                if code < 0:                    # killed by a signal: the shell's convention
                    rc, reason = 128 - code, f"rank {r} was killed by signal {-code}"
                else:
                    rc, reason = code, f"rank {r} exited with code {code}"
        if rc != 0:
            break
        if live and time.monotonic() - t0 > time_limit:
            rc, reason = 124, f"time limit of {time_limit:.0f} s exceeded ({len(live)} of {nranks} ranks still running)"
            break
        if live:
            time.sleep(0.05)
    if live:                                    # a failure or the limit: nobody is left behind
        for r in live:
            _kill_group(procs[r], signal.SIGTERM)
        t1 = time.monotonic()
        while any(procs[r].poll() is None for r in live) and time.monotonic() - t1 < grace:
            time.sleep(0.05)
        for r in live:
            if procs[r].poll() is None:
                _kill_group(procs[r], signal.SIGKILL)
        for r in live:
            try:
                procs[r].wait(timeout=grace)
            except subprocess.TimeoutExpired:
                pass
    if reader is not None:
        reader.join(timeout=grace)
    return rc, reason


class Watchdog:
    """Calls ``on_expire()`` and then ``os._exit(code)`` (code: an int or a callable returning one)
    from a daemon thread when `limit` seconds pass before ``cancel()``: the main thread may be stuck
    inside a C call (ncclCommInitRank with a peer that never arrives), where neither signals nor
    exceptions reach it."""

    def __init__(self, limit, on_expire=None, code=124):
        self._ev = threading.Event()
        self.limit = limit

        def run():
            if self._ev.wait(limit):
                return
            try:
                if on_expire:
                    on_expire()
            finally:
                try:
                    sys.stderr.flush()
                except Exception:
                    pass
                os._exit(code() if callable(code) else code)

        self._t = threading.Thread(target=run, name="wt-watchdog", daemon=True)
        self._t.start()

    def cancel(self):
        self._ev.set()

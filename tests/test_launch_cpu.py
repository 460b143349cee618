"""CPU tests of the torch-free launcher plumbing (wavelets_amd/launch.py) that `bench.py --gpus N`
and tools/check_rccl_ranks.py use between their ranks: rendezvous over one abstract AF_UNIX socket,
the collectives the bench needs (unique-id broadcast, barrier, MAX over ranks), the wall-clock limit,
and that no torch import sits on bench.py's default path."""
import ast
import json
import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    from wavelets_amd.launch import group_from_env
    g = group_from_env(connect_timeout=30, op_timeout=30)
    rank, world = g.rank, g.world
    uid = g.bcast(bytes(range(128)) if rank == 0 else None, 0)          # the RCCL unique id's path
    assert uid == bytes(range(128))
    assert g.bcast("from-last" if rank == world - 1 else None, world - 1) == "from-last"
    assert g.allreduce(float(rank), max) == float(world - 1)
    assert g.allreduce(rank, sum) == world * (world - 1) // 2
    assert g.allreduce(rank < world, all) is True
    got = g.gather({{"rank": rank}})
    if rank == 0:
        assert [d["rank"] for d in got] == list(range(world))
    else:
        assert got is None
    for _ in range(50):
        g.barrier()
    g.close()
    if rank == 0:
        print("group ok", world, "torch" in sys.modules, flush=True)
""")


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_socket_group_collectives_between_real_processes(tmp_path, world):
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT))
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from wavelets_amd import launch
        seen = []
        rc, reason = launch.spawn({world}, [sys.executable, {str(script)!r}], time_limit=60, tee_rank0=seen)
        print("RC", rc, reason, len(seen))
        sys.exit(rc)
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"group ok {world} False" in r.stdout                 # (and torch was never imported)
    assert "RC 0 None 1" in r.stdout


def test_group_under_a_foreign_launcher_uses_master_port(tmp_path):
    """Ranks started by somebody else's launcher (the driver's torch.distributed.run) have no WT_RDZV:
    the socket's name then comes from MASTER_ADDR / MASTER_PORT, which that launcher made unique."""
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT.format(root=ROOT))
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(40000 + os.getpid() % 20000))
        env.pop("WT_RDZV", None)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, text=True))
        if r == 1:
            time.sleep(0.3)            # late joiners and an early rank 1 (before rank 0 listens) both work
    outs = [p.communicate(timeout=60)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "group ok 3 False" in outs[0]


def test_spawn_kills_a_hung_launch_at_the_time_limit(tmp_path):
    """Rank 1 never reaches the rendezvous (a hung ncclCommInitRank looks like this from outside):
    at the limit every rank's process group is killed and the caller gets code 124 and the reason."""
    script = tmp_path / "hang.py"
    marker = tmp_path / "pids"
    script.write_text(textwrap.dedent(f"""
        import os, time
        open({str(marker)!r} + os.environ["RANK"], "w").write(str(os.getpid()))
        time.sleep(600)
    """))
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from wavelets_amd import launch
        t = time.monotonic()
        rc, reason = launch.spawn(2, [sys.executable, {str(script)!r}], time_limit=2.0, grace=2.0)
        print("RC", rc, "|", reason, "|", round(time.monotonic() - t, 1))
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert "RC 124 | time limit of 2 s exceeded" in r.stdout, r.stdout + r.stderr
    for k in (0, 1):                                         # nobody is left behind
        pid = int(open(str(marker) + str(k)).read())
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)


def test_spawn_reports_the_first_failing_rank_and_stops_the_others(tmp_path):
    script = tmp_path / "fail.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(600)\n")
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from wavelets_amd import launch
        print("RC", *launch.spawn(3, [sys.executable, {str(script)!r}], time_limit=60, grace=2.0))
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert "RC 7 rank 1 exited with code 7" in r.stdout, r.stdout + r.stderr


def test_spawn_reports_a_rank_killed_by_a_signal_the_way_a_shell_does(tmp_path):
    script = tmp_path / "segv.py"
    script.write_text("import os, signal, time\nif os.environ['RANK'] == '0':\n    os.kill(os.getpid(), signal.SIGKILL)\ntime.sleep(600)\n")
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from wavelets_amd import launch
        print("RC", *launch.spawn(2, [sys.executable, {str(script)!r}], time_limit=60, grace=2.0))
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert "RC 137 rank 0 was killed by signal 9" in r.stdout, r.stdout + r.stderr


def test_rendezvous_refuses_a_peer_of_another_user(monkeypatch):
    """Abstract sockets have no permissions; the group checks SO_PEERCRED before it unpickles anything."""
    import socket
    from wavelets_amd import launch
    a, b = socket.socketpair(socket.AF_UNIX, socket.SOCK_STREAM)
    launch._check_peer(a)                                   # our own uid: accepted
    monkeypatch.setattr(os, "getuid", lambda: 54321)
    with pytest.raises(PermissionError, match="uid"):
        launch._check_peer(a)
    assert a.fileno() == -1                                 # and the connection is gone
    b.close()


def test_watchdog_fires_inside_a_stuck_rank():
    code = textwrap.dedent(f"""
        import sys, time
        sys.path.insert(0, {ROOT!r})
        from wavelets_amd.launch import Watchdog
        Watchdog(0.5, lambda: print("expired", flush=True), code=lambda: 9)
        time.sleep(60)          # stands for a C call that never returns
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=30)
    assert r.returncode == 9 and "expired" in r.stdout


def _torch_import_sites(path):
    """(line, enclosing function / class names) of every `import torch...` in a source file"""
    tree = ast.parse(open(path).read())
    sites = []

    def walk(node, stack):
        for child in ast.iter_child_nodes(node):
            st = stack
            if isinstance(child, (ast.FunctionDef, ast.ClassDef)):
                st = stack + [child.name]
            if isinstance(child, ast.Import) and any(a.name.split(".")[0] == "torch" for a in child.names):
                sites.append((child.lineno, stack))
            if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "torch":
                sites.append((child.lineno, stack))
            walk(child, st)
    walk(tree, [])
    return sites


def test_bench_does_not_import_torch_on_the_default_path():
    """N = 1 and N > 1 must run on the same ROCm stack: torch (and its bundled runtime) may only be
    imported behind --launcher torch.  Static part: every `import torch` of bench.py sits in the
    TorchGroup adapter or directly behind the launcher switch; the product package has none at all."""
    src = open(os.path.join(ROOT, "bench.py")).read().splitlines()
    for lineno, stack in _torch_import_sites(os.path.join(ROOT, "bench.py")):
        if "TorchGroup" in stack:
            continue
        guard = "\n".join(src[max(0, lineno - 4):lineno])
        assert 'args.launcher == "torch"' in guard, f"bench.py:{lineno}: torch imported outside the launcher switch"
    for name in os.listdir(os.path.join(ROOT, "wavelets_amd")):
        if name.endswith(".py"):
            assert not _torch_import_sites(os.path.join(ROOT, "wavelets_amd", name)), name
    assert not _torch_import_sites(os.path.join(ROOT, "__graft_entry__.py"))


def test_bench_multi_rank_launch_fails_loudly_without_a_gpu():
    """Dynamic part, on this CPU-only box: `python bench.py --gpus 2` starts two ranks through the
    stdlib launcher; they cannot create a device context, so the launcher must end both, print ONE
    JSON error line and return non-zero - and the failing ranks never imported torch."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "WT_RDZV"):
        env.pop(k, None)
    env["WT_BENCH_REPORT_MODULES"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "256", "--steps", "1",
                        "--no-cpu", "--no-build", "--time-limit", "120"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout + r.stderr[-2000:]
    err = json.loads(lines[0])
    assert "error" in err and err["n_gpus"] == 2 and err["launcher"] == "stdlib"
    assert "no HIP device" in r.stderr or "HIP" in r.stderr
    assert "torch_in_sys_modules=False" in r.stderr

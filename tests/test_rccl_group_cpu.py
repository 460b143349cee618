"""The ncclGroupStart / ncclGroupEnd bracket of the halo exchange (wavelets_amd/csrc/wt_rccl_group.h) against a
function table that fails on demand - compiled with g++ here, no GPU: a failing Send must leave no group open,
skip the operations behind it and surface ITS error code; and the stub RCCL library of the GPU test builds."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = textwrap.dedent(r'''
    #include <cstdio>
    #include "wt_rccl_group.h"
    struct Api {
        int depth = 0, starts = 0, ends = 0, sends = 0, recvs = 0, fail_send = 0, fail_start = 0, fail_end = 0;
        int GroupStart() { ++starts; if (fail_start) return 7; ++depth; return 0; }
        int GroupEnd() { ++ends; --depth; return fail_end ? 9 : 0; }
        int Send() { ++sends; return sends == fail_send ? 3 : 0; }
        int Recv() { ++recvs; return 0; }
    };
    // the exchange of an interior rank: up (send, recv), down (send, recv)
    static int exchange(Api &api, const char **what)
    {
        WtRcclGroup<Api> g(api);
        g.run("send up", [&] { return api.Send(); });
        g.run("recv up", [&] { return api.Recv(); });
        g.run("send down", [&] { return api.Send(); });
        g.run("recv down", [&] { return api.Recv(); });
        const int rc = g.end();
        *what = g.what;
        return rc;
    }
    static int early_return(Api &api)
    {
        WtRcclGroup<Api> g(api);
        g.run("send", [&] { return api.Send(); });
        return 42;                     // leaves without end(): the destructor closes the group
    }
    int main()
    {
        const char *what = "";
        Api ok;
        if (exchange(ok, &what) != 0 || ok.depth != 0 || ok.starts != 1 || ok.ends != 1 || ok.sends != 2 || ok.recvs != 2) return 1;
        Api bad;
        bad.fail_send = 2;             // the SECOND send fails (the review's case)
        const int rc = exchange(bad, &what);
        std::printf("rc=%d what=%s depth=%d starts=%d ends=%d sends=%d recvs=%d\n", rc, what, bad.depth, bad.starts, bad.ends, bad.sends, bad.recvs);
        if (rc != 3 || bad.depth != 0 || bad.ends != 1 || bad.sends != 2 || bad.recvs != 1) return 2;   // recv down was skipped
        if (exchange(bad, &what) != 0 || bad.depth != 0 || bad.starts != 2 || bad.ends != 2) return 3;  // the next exchange works
        Api first;
        first.fail_send = 1;
        if (exchange(first, &what) != 3 || first.depth != 0 || first.recvs != 0 || first.ends != 1) return 4;
        Api nostart;
        nostart.fail_start = 1;        // GroupStart itself fails: nothing is queued, GroupEnd is NOT called
        if (exchange(nostart, &what) != 7 || nostart.ends != 0 || nostart.sends != 0 || nostart.depth != 0) return 5;
        Api endfails;
        endfails.fail_end = 1;
        if (exchange(endfails, &what) != 9 || endfails.depth != 0) return 6;
        Api early;
        if (early_return(early) != 42 || early.depth != 0 || early.ends != 1) return 7;
        return 0;
    }
''')


def test_rccl_group_is_always_closed_and_reports_the_first_error(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(PROGRAM)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "wavelets_amd", "csrc"),
                           str(src), "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "rc=3 what=send down depth=0" in r.stdout


def test_halo_exchange_uses_the_group_bracket_and_no_bare_group_calls_remain():
    """every ncclGroupStart of the library goes through WtRcclGroup (a bare WT_NCCL(GroupStart) is the bug class)"""
    text = open(os.path.join(ROOT, "wavelets_amd", "csrc", "wt_core.hip")).read()
    assert "g_rccl.GroupStart()" not in text and "g_rccl.GroupEnd()" not in text
    assert text.count("WtRcclGroup<RcclApi>") >= 2


def test_stub_rccl_library_builds_and_exports_what_the_engine_binds(tmp_path):
    so = tmp_path / "librccl_stub.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-o", str(so),
                           os.path.join(ROOT, "tests", "stubs", "rccl_stub.c")])
    import ctypes
    lib = ctypes.CDLL(str(so))
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclCommUserRank",
                 "ncclGroupStart", "ncclGroupEnd", "ncclSend", "ncclRecv", "ncclAllReduce", "ncclGetErrorString",
                 "ncclGetVersion", "rccl_stub_state"):
        assert hasattr(lib, name), name
    # the names wt_core.hip resolves with dlsym are exactly these
    text = open(os.path.join(ROOT, "wavelets_amd", "csrc", "wt_core.hip")).read()
    import re
    for sym in re.findall(r'SYM\(\w+, "(nccl\w+)"\)', text):
        assert hasattr(lib, sym), sym

// ncclGroupStart / ncclGroupEnd bracket that cannot be left open (round 5).  A failing ncclSend / ncclRecv
// between the two used to return straight out of the entry point: the group stayed open on this thread and
// the next call on the communicator hung or failed with an unrelated message.  Here every operation of the
// group goes through op(); after the first failure the remaining operations are skipped, the group is ALWAYS
// closed (end(), or the destructor on an early return), and the first error code is what the caller sees.
// No HIP or RCCL types: the function table is a template parameter, so tests/test_rccl_group_cpu.py compiles
// this header with g++ against a stub that fails on demand.
#pragma once

template <typename Api>
struct WtRcclGroup {
    Api &api;
    bool open = false;
    int err = 0;            // first non-zero RCCL result inside the group (or of GroupStart / GroupEnd)
    const char *what = "";  // which call produced it

    explicit WtRcclGroup(Api &a) : api(a)
    {
        err = api.GroupStart();
        if (err) what = "ncclGroupStart";
        else open = true;
    }
    // rc = the result of an ncclSend / ncclRecv the caller has just made, or use run() to skip it after a failure
    template <typename F>
    void run(const char *name, F &&call)
    {
        if (err) return;     // an earlier operation failed: do not queue more work into a group that will be reported as failed
        const int rc = call();
        if (rc) {
            err = rc;
            what = name;
        }
    }
    int end()
    {
        if (open) {
            open = false;
            const int rc = api.GroupEnd();
            if (rc && !err) {
                err = rc;
                what = "ncclGroupEnd";
            }
        }
        return err;
    }
    ~WtRcclGroup() { (void)end(); }
    WtRcclGroup(const WtRcclGroup &) = delete;
    WtRcclGroup &operator=(const WtRcclGroup &) = delete;
};

/* A stand-in for librccl.so with just the entry points libwatroo_hip.so binds (wt_core.hip, rccl_load), for
 * tests of the error paths: WATROO_HIP_RCCL_LIB=<this library>.  Nothing moves: Send / Recv / AllReduce return
 * success without touching their buffers.  RCCL_STUB_FAIL_SEND=<n> makes the n-th ncclSend of the process
 * (1-based) return ncclInternalError once, RCCL_STUB_FAIL_ALLREDUCE=<n> the n-th ncclAllReduce.  The group depth and the call counts are exported so that a test
 * can check that a failed exchange left no group open. */
#include <stdlib.h>
#include <string.h>

static int g_depth, g_starts, g_ends, g_sends, g_recvs, g_nested_starts;
static int g_comm_obj;

int ncclGetUniqueId(void *id) { memset(id, 0x5a, 128); return 0; }
struct id128 { char b[128]; };
int ncclCommInitRank(void **comm, int nranks, struct id128 id, int rank) { (void)nranks; (void)id; (void)rank; *comm = &g_comm_obj; return 0; }
int ncclCommDestroy(void *comm) { (void)comm; return 0; }
static int g_nranks = 2, g_rank = 0;
int ncclCommCount(void *comm, int *n) { (void)comm; *n = getenv("RCCL_STUB_NRANKS") ? atoi(getenv("RCCL_STUB_NRANKS")) : g_nranks; return 0; }
int ncclCommUserRank(void *comm, int *r) { (void)comm; *r = g_rank; return 0; }
int ncclGroupStart(void) { if (g_depth > 0) g_nested_starts++; g_depth++; g_starts++; return 0; }
int ncclGroupEnd(void) { if (g_depth <= 0) return 5; g_depth--; g_ends++; return 0; }
int ncclSend(const void *b, size_t n, int t, int peer, void *comm, void *st)
{
    (void)b; (void)n; (void)t; (void)peer; (void)comm; (void)st;
    g_sends++;
    const char *f = getenv("RCCL_STUB_FAIL_SEND");
    if (f && atoi(f) == g_sends) return 3;      /* ncclInternalError */
    return 0;
}
int ncclRecv(void *b, size_t n, int t, int peer, void *comm, void *st) { (void)b; (void)n; (void)t; (void)peer; (void)comm; (void)st; g_recvs++; return 0; }
static int g_allreduces;
int ncclAllReduce(const void *s, void *r, size_t n, int t, int op, void *comm, void *st)
{
    (void)s; (void)r; (void)n; (void)t; (void)op; (void)comm; (void)st;
    g_allreduces++;
    const char *f = getenv("RCCL_STUB_FAIL_ALLREDUCE");     /* the n-th ncclAllReduce of the process (1-based) fails once */
    if (f && atoi(f) == g_allreduces) return 3;
    return 0;
}
int rccl_stub_allreduces(void) { return g_allreduces; }
const char *ncclGetErrorString(int e) { return e == 3 ? "internal error (stub)" : (e == 5 ? "invalid usage (stub)" : "stub error"); }
int ncclGetVersion(int *v) { *v = 29999; return 0; }
/* {open groups, GroupStart calls, GroupEnd calls, Sends, Recvs, GroupStart calls made while a group was open} */
void rccl_stub_state(int out[6]) { out[0] = g_depth; out[1] = g_starts; out[2] = g_ends; out[3] = g_sends; out[4] = g_recvs; out[5] = g_nested_starts; }

/* rccl_stub.c — a stand-in for librccl.so for ONE purpose: holding the RCCL calls of libgpuart_hip.so's multi-GPU read-out so that
 * its bounded waits can be exercised on a box with one GPU (tests/test_stall_path.py). Test infrastructure, never shipped or
 * loaded by the product unless GPUART_HIP_RCCL_LIBRARY names it. It implements the entry points gather.h resolves, for a
 * communicator of ONE rank: nothing is ever transferred (a single rank's gather posts no send / recv), every call succeeds at once —
 * except the one RCCL_STUB_HOLD names (init_all | init_rank | destroy | group_end | all_gather), which sleeps for RCCL_STUB_HOLD_MS
 * milliseconds (default: one hour, i.e. "never returns" on the scale of a test) before it succeeds.
 * Build: gcc -shared -fPIC -O1 -o librccl_stub.so rccl_stub.c */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;

static void hold(const char *what) {
    const char *h = getenv("RCCL_STUB_HOLD");
    if (!h || strcmp(h, what) != 0) return;
    const char *m = getenv("RCCL_STUB_HOLD_MS");
    long ms = m ? atol(m) : 3600000L;
    struct timespec ts = {ms / 1000, (ms % 1000) * 1000000L};
    while (nanosleep(&ts, &ts) != 0) {}
}

static int g_handles[64];

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof *id); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    (void)id;
    if (nranks != 1 || rank != 0) return 4; /* ncclInvalidArgument: the stub knows communicators of one rank only */
    hold("init_rank");
    *comm = &g_handles[0];
    return 0;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comms, int n, const int *devs) {
    (void)devs;
    if (n != 1) return 4;
    hold("init_all");
    comms[0] = &g_handles[1];
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { (void)comm; hold("destroy"); return 0; }
ncclResult_t ncclGroupStart(void) { return 0; }
ncclResult_t ncclGroupEnd(void) { hold("group_end"); return 0; }
ncclResult_t ncclSend(const void *b, size_t n, int t, int peer, ncclComm_t c, void *s) { (void)b; (void)n; (void)t; (void)peer; (void)c; (void)s; return 5; }
ncclResult_t ncclRecv(void *b, size_t n, int t, int peer, ncclComm_t c, void *s) { (void)b; (void)n; (void)t; (void)peer; (void)c; (void)s; return 5; }
/* one rank: the gathered table is the rank's own entry — the stub cannot copy device memory (it does not link HIP), so the
 * all-gather of gpuart_hip_gather is not served; the tests drive gpuart_hip_gather_all, which never calls it */
ncclResult_t ncclAllGather(const void *s, void *r, size_t n, int t, ncclComm_t c, void *st) { (void)s; (void)r; (void)n; (void)t; (void)c; (void)st; hold("all_gather"); return 5; }
ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { (void)c; *n = 1; return 0; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { (void)c; *r = 0; return 0; }
const char *ncclGetErrorString(ncclResult_t e) { return e == 0 ? "no error" : e == 4 ? "invalid argument (rccl_stub: one rank only)" : "rccl_stub: not served"; }

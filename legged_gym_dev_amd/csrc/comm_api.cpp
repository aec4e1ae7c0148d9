// lg_comm_*: the learner's collectives on RCCL, for hosts without torch.distributed (include/legged_hip.h).
// The library does not link librccl: it resolves the few entry points it needs at first use from the librccl.so the process
// already holds (torch ships one), else from the loader path -- one RCCL instance per process either way.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/legged_hip.h"

void lg_set_error(const std::string &s);

namespace {
typedef struct { char internal[128]; } nccl_uid;                      // ncclUniqueId (rccl.h:43)
typedef void *nccl_comm_t;
typedef int (*fn_get_uid)(nccl_uid *);
typedef int (*fn_init_rank)(nccl_comm_t *, int, nccl_uid, int);
typedef int (*fn_destroy)(nccl_comm_t);
typedef int (*fn_allreduce)(const void *, void *, size_t, int /*dtype*/, int /*op*/, nccl_comm_t, hipStream_t);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_errstr)(int);
struct Rccl {
    void *h = nullptr;
    fn_get_uid get_uid; fn_init_rank init_rank; fn_destroy destroy; fn_allreduce allreduce; fn_bcast bcast;
    fn_group group_start, group_end; fn_errstr errstr;
} R;
const int kFloat32 = 7, kSum = 0;                                      // ncclFloat32, ncclSum (rccl.h:448,466)

bool load_rccl() {
    if (R.h) return true;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);            // the instance torch loaded, when there is one
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { lg_set_error(std::string("librccl.so not found: ") + dlerror()); return false; }
#define SYM(field, name) do { R.field = (decltype(R.field))dlsym(h, name); if (!R.field) { lg_set_error("librccl.so lacks " name); return false; } } while (0)
    SYM(get_uid, "ncclGetUniqueId"); SYM(init_rank, "ncclCommInitRank"); SYM(destroy, "ncclCommDestroy");
    SYM(allreduce, "ncclAllReduce"); SYM(bcast, "ncclBroadcast"); SYM(group_start, "ncclGroupStart"); SYM(group_end, "ncclGroupEnd");
    SYM(errstr, "ncclGetErrorString");
#undef SYM
    R.h = h;
    return true;
}
int chk(int rc, const char *what) {
    if (rc == 0) return 0;
    lg_set_error(std::string(what) + ": " + (R.errstr ? R.errstr(rc) : "rccl error"));
    return -200;
}
}  // namespace

struct lg_comm {
    nccl_comm_t comm;
    int rank, nranks;
    hipStream_t stream;                 // collectives of the overlapped gradient reduction run here
    hipEvent_t done;
};

extern "C" {

int lg_comm_get_unique_id(void *id_out) {
    if (!id_out) { lg_set_error("null id buffer"); return -1; }
    if (!load_rccl()) return -201;
    nccl_uid u;
    if (int rc = chk(R.get_uid(&u), "ncclGetUniqueId")) return rc;
    memcpy(id_out, &u, sizeof(u));
    return 0;
}

int lg_comm_init(int rank, int nranks, const void *id, lg_comm **out) {
    if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) { lg_set_error("lg_comm_init: bad arguments"); return -1; }
    if (!load_rccl()) return -201;
    nccl_uid u;
    memcpy(&u, id, sizeof(u));
    lg_comm *c = new lg_comm();
    c->rank = rank; c->nranks = nranks; c->comm = nullptr; c->stream = nullptr; c->done = nullptr;
    if (int rc = chk(R.init_rank(&c->comm, nranks, u, rank), "ncclCommInitRank")) { delete c; return rc; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        lg_set_error("lg_comm_init: stream/event creation failed"); R.destroy(c->comm); delete c; return -100;
    }
    *out = c;
    return 0;
}

int lg_comm_destroy(lg_comm *c) {
    if (!c) return 0;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->comm) R.destroy(c->comm);
    delete c;
    return 0;
}

int lg_comm_rank(lg_comm *c) { return c->rank; }
int lg_comm_size(lg_comm *c) { return c->nranks; }

int lg_comm_allreduce_sum(lg_comm *c, float *buf, int64_t n, void *stream) {
    return chk(R.allreduce(buf, buf, (size_t)n, kFloat32, kSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}
int lg_comm_broadcast(lg_comm *c, float *buf, int64_t n, int root, void *stream) {
    return chk(R.bcast(buf, buf, (size_t)n, kFloat32, root, c->comm, (hipStream_t)stream), "ncclBroadcast");
}

// internal: used by ppo_api.hip for the overlapped per-layer gradient reduction
hipStream_t lg_comm_stream_(lg_comm *c) { return c->stream; }
hipEvent_t lg_comm_event_(lg_comm *c) { return c->done; }
int lg_comm_group_allreduce_(lg_comm *c, float *const *bufs, const int64_t *counts, int n, hipStream_t s) {
    if (int rc = chk(R.group_start(), "ncclGroupStart")) return rc;
    int rc = 0;
    for (int k = 0; k < n && !rc; ++k)
        if (counts[k] > 0) rc = chk(R.allreduce(bufs[k], bufs[k], (size_t)counts[k], kFloat32, kSum, c->comm, s), "ncclAllReduce");
    int rc2 = chk(R.group_end(), "ncclGroupEnd");
    return rc ? rc : rc2;
}

}  // extern "C"

// The exchange step of the data-parallel path in the C ABI (reference distribution/distribution_utils.py:158-169 all_reduce_values /
// ReplicaContext.all_reduce SUM; SyncBN statistics and gradient sums): a thin layer over RCCL.  One process per GPU; the 128-byte unique id
// is created on rank 0 and handed to the other ranks by the host program (file, socket, MPI, torch's store -- not this library's business).
// RCCL is resolved at run time (an already loaded copy first -- a Python process with torch has one -- then librccl.so.1 / librccl.so), so the
// library has no link-time dependency on it and the product path (torch.distributed over the same RCCL) is untouched.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {

typedef struct { char internal[128]; } rccl_unique_id;      // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* rccl_comm;
typedef int (*fn_get_unique_id)(rccl_unique_id*);
typedef int (*fn_comm_init_rank)(rccl_comm*, int, rccl_unique_id, int);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_comm_destroy)(rccl_comm);
typedef const char* (*fn_error_string)(int);

struct Rccl {
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_error_string error_string = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static const Rccl r = [] {
        Rccl t;
        void* handles[3] = {RTLD_DEFAULT, nullptr, nullptr};
        for (int i = 0; i < 3 && !t.ok; ++i) {
            void* h = handles[i];
            if (i == 1) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (i == 2) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (i > 0 && !h) continue;
            t.get_unique_id = (fn_get_unique_id)dlsym(h, "ncclGetUniqueId");
            t.comm_init_rank = (fn_comm_init_rank)dlsym(h, "ncclCommInitRank");
            t.all_reduce = (fn_all_reduce)dlsym(h, "ncclAllReduce");
            t.comm_destroy = (fn_comm_destroy)dlsym(h, "ncclCommDestroy");
            t.error_string = (fn_error_string)dlsym(h, "ncclGetErrorString");
            t.ok = t.get_unique_id && t.comm_init_rank && t.all_reduce && t.comm_destroy;
        }
        return t;
    }();
    return r;
}

struct Comm {
    rccl_comm handle;
    int world, rank;
};

int fail(const char* what, int code) {
    const Rccl& r = rccl();
    iseg_set_error("%s: RCCL error %d (%s)", what, code, r.error_string ? r.error_string(code) : "?");
    return ISEG_ERR_HIP;
}

}  // namespace

extern "C" int iseg_comm_unique_id(void* id128) {
    ISEG_REQUIRE(id128, "iseg_comm_unique_id: null pointer");
    const Rccl& r = rccl();
    if (!r.ok) {
        iseg_set_error("iseg_comm_unique_id: RCCL not found (librccl.so.1)");
        return ISEG_ERR_UNSUPPORTED;
    }
    rccl_unique_id id;
    const int rc = r.get_unique_id(&id);
    if (rc != 0) return fail("iseg_comm_unique_id", rc);
    memcpy(id128, id.internal, 128);
    return ISEG_OK;
}

extern "C" int iseg_comm_init(void** comm, int world_size, int rank, const void* id128) {
    ISEG_REQUIRE(comm && id128 && world_size >= 1 && rank >= 0 && rank < world_size, "iseg_comm_init: bad arguments");
    const Rccl& r = rccl();
    if (!r.ok) {
        iseg_set_error("iseg_comm_init: RCCL not found (librccl.so.1)");
        return ISEG_ERR_UNSUPPORTED;
    }
    rccl_unique_id id;
    memcpy(id.internal, id128, 128);
    rccl_comm h = nullptr;
    const int rc = r.comm_init_rank(&h, world_size, id, rank);
    if (rc != 0) return fail("iseg_comm_init", rc);
    *comm = new Comm{h, world_size, rank};
    return ISEG_OK;
}

// in place: buf[i] = sum over ranks of buf[i]; fp32 (SyncBN messages, gradient buckets) or bf16
extern "C" int iseg_allreduce_sum(void* comm, void* buf, size_t count, int dtype, hipStream_t stream) {
    ISEG_REQUIRE(comm && (buf || count == 0), "iseg_allreduce_sum: bad arguments");
    ISEG_REQUIRE(dtype == ISEG_F32 || dtype == ISEG_BF16, "iseg_allreduce_sum: dtype must be f32 or bf16");
    if (count == 0) return ISEG_OK;
    const Comm* c = (const Comm*)comm;
    const int rc = rccl().all_reduce(buf, buf, count, dtype == ISEG_F32 ? 7 /* ncclFloat32 */ : 9 /* ncclBfloat16 */, 0 /* ncclSum */, c->handle, stream);
    if (rc != 0) return fail("iseg_allreduce_sum", rc);
    return ISEG_OK;
}

extern "C" int iseg_comm_destroy(void* comm) {
    if (!comm) return ISEG_OK;
    Comm* c = (Comm*)comm;
    const int rc = rccl().comm_destroy(c->handle);
    delete c;
    if (rc != 0) return fail("iseg_comm_destroy", rc);
    return ISEG_OK;
}

// RCCL behind the C ABI: the one collective of the path without torch.
//
// The reference has no multi-device code; a Julia (or any C-ABI) caller that shards the
// batch over the 8 GPUs of a node needs exactly: a communicator, and `all-reduce(sum)` of the two
// Float64 partial sums between fx3d_chamfer_sums and fx3d_chamfer_finalize.  librccl is resolved at
// run time (dlopen) so the library also loads on hosts without RCCL; one process per GPU, the unique id
// travels through whatever the host already has (MPI, a file, torch.distributed, Julia Distributed).
#include <dlfcn.h>

#include <cstring>

#include "fx3d_common.h"

using namespace fx3d;

namespace {

typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void *ncclComm_t_;
typedef int (*GetUniqueId_fn)(ncclUniqueId_t *);
typedef int (*CommInitRank_fn)(ncclComm_t_ *, int, ncclUniqueId_t, int);
typedef int (*CommDestroy_fn)(ncclComm_t_);
typedef int (*AllReduce_fn)(const void *, void *, size_t, int, int, ncclComm_t_, hipStream_t);
typedef const char *(*GetErrorString_fn)(int);

struct Rccl {
    void *h = nullptr;
    GetUniqueId_fn get_id = nullptr;
    CommInitRank_fn init_rank = nullptr;
    CommDestroy_fn destroy = nullptr;
    AllReduce_fn allreduce = nullptr;
    GetErrorString_fn errstr = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (r.h) {
            r.get_id = (GetUniqueId_fn)dlsym(r.h, "ncclGetUniqueId");
            r.init_rank = (CommInitRank_fn)dlsym(r.h, "ncclCommInitRank");
            r.destroy = (CommDestroy_fn)dlsym(r.h, "ncclCommDestroy");
            r.allreduce = (AllReduce_fn)dlsym(r.h, "ncclAllReduce");
            r.errstr = (GetErrorString_fn)dlsym(r.h, "ncclGetErrorString");
        }
    }
    return (r.h && r.get_id && r.init_rank && r.destroy && r.allreduce) ? &r : nullptr;
}

fx3d_status rccl_fail(int rc, const char *what) {
    Rccl *r = rccl();
    set_error("%s failed: %s", what, (r && r->errstr) ? r->errstr(rc) : "RCCL error");
    return FX3D_ERR_RCCL;
}

constexpr int kNcclSum = 0, kNcclFloat64 = 8;  // rccl.h: ncclSum = 0, ncclFloat64 = 8

}  // namespace

extern "C" {

fx3d_status fx3d_comm_unique_id(uint8_t *id128) {
    FX3D_REQUIRE(id128, "fx3d_comm_unique_id: null output");
    Rccl *r = rccl();
    if (!r) { set_error("librccl could not be loaded"); return FX3D_ERR_RCCL; }
    ncclUniqueId_t id;
    const int rc = r->get_id(&id);
    if (rc) return rccl_fail(rc, "ncclGetUniqueId");
    memcpy(id128, id.internal, 128);
    return FX3D_OK;
}

fx3d_status fx3d_comm_init_rank(fx3d_comm_t *comm, int32_t nranks, const uint8_t *id128, int32_t rank) {
    FX3D_REQUIRE(comm && id128, "fx3d_comm_init_rank: null pointer");
    FX3D_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, "fx3d_comm_init_rank: bad rank %d of %d", rank, nranks);
    Rccl *r = rccl();
    if (!r) { set_error("librccl could not be loaded"); return FX3D_ERR_RCCL; }
    ncclUniqueId_t id;
    memcpy(id.internal, id128, 128);
    ncclComm_t_ c = nullptr;
    const int rc = r->init_rank(&c, nranks, id, rank);  // uses the calling thread's current device
    if (rc) return rccl_fail(rc, "ncclCommInitRank");
    *comm = c;
    return FX3D_OK;
}

fx3d_status fx3d_comm_destroy(fx3d_comm_t comm) {
    if (!comm) return FX3D_OK;
    Rccl *r = rccl();
    if (!r) return FX3D_ERR_RCCL;
    const int rc = r->destroy(comm);
    return rc ? rccl_fail(rc, "ncclCommDestroy") : FX3D_OK;
}

fx3d_status fx3d_comm_allreduce_sum_f64(fx3d_comm_t comm, double *buf_dev, int64_t count, fx3d_stream_t s) {
    FX3D_REQUIRE(comm && buf_dev && count > 0, "fx3d_comm_allreduce_sum_f64: bad argument");
    Rccl *r = rccl();
    if (!r) { set_error("librccl could not be loaded"); return FX3D_ERR_RCCL; }
    const int rc = r->allreduce(buf_dev, buf_dev, (size_t)count, kNcclFloat64, kNcclSum, comm, as_stream(s));
    return rc ? rccl_fail(rc, "ncclAllReduce") : FX3D_OK;
}

// chamfer_distance over a batch sharded across the ranks of `comm`: this rank's shard in, the GLOBAL
// loss out on every rank.  kernel (partials reduced in-launch) -> all-reduce(sum) of 2 Float64 ->
// finalise with the global batch size; all on `s`.
fx3d_status fx3d_chamfer_fwd_sharded(fx3d_comm_t comm, const float *x, int32_t N, const float *y, int32_t M,
                                     int32_t B_local, int32_t D, int64_t B_global, float w1, float w2,
                                     double *sums_dev, float *loss_dev, float *loss_host, void *ws,
                                     size_t ws_bytes, fx3d_stream_t s) {
    FX3D_REQUIRE(comm && sums_dev && loss_dev, "fx3d_chamfer_fwd_sharded: null pointer");
    FX3D_REQUIRE(B_global >= B_local && B_local >= 0, "fx3d_chamfer_fwd_sharded: bad batch sizes");
    fx3d_status rc;
    if (B_local > 0) {
        rc = fx3d_chamfer_sums(x, N, y, M, B_local, D, sums_dev, nullptr, nullptr, ws, ws_bytes, s);
        if (rc) return rc;
    } else {  // more ranks than batch elements: this rank contributes zeros
        FX3D_HIP(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), as_stream(s)));
    }
    rc = fx3d_comm_allreduce_sum_f64(comm, sums_dev, 2, s);
    if (rc) return rc;
    rc = fx3d_chamfer_finalize(sums_dev, N, M, B_global, D, w1, w2, loss_dev, s);
    if (rc) return rc;
    if (loss_host) {
        FX3D_HIP(hipMemcpyAsync(loss_host, loss_dev, sizeof(float), hipMemcpyDeviceToHost, as_stream(s)));
        FX3D_HIP(hipStreamSynchronize(as_stream(s)));
    }
    return FX3D_OK;
}

}  // extern "C"

// RCCL behind the C ABI: the one collective of the path without torch.
//
// The reference has no multi-device code; a Julia (or any C-ABI) caller that shards the
// batch over the 8 GPUs of a node needs exactly: a communicator, and `all-reduce(sum)` of the two
// Float64 partial sums between fx3d_chamfer_sums and fx3d_chamfer_finalize.  librccl is resolved at
// run time (dlopen) so the library also loads on hosts without RCCL; one process per GPU, the unique id
// travels through whatever the host already has (MPI, a file, torch.distributed, Julia Distributed).
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>

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
typedef int (*GetVersion_fn)(int *);
typedef int (*CommCount_fn)(ncclComm_t_, int *);
typedef int (*CommUserRank_fn)(ncclComm_t_, int *);

struct Rccl {
    void *h = nullptr;
    GetUniqueId_fn get_id = nullptr;
    CommInitRank_fn init_rank = nullptr;
    CommDestroy_fn destroy = nullptr;
    AllReduce_fn allreduce = nullptr;
    GetErrorString_fn errstr = nullptr;
    GetVersion_fn version = nullptr;
    CommCount_fn count = nullptr;
    CommUserRank_fn user_rank = nullptr;
};

Rccl *rccl() {
    static Rccl r;
    static std::once_flag once;  // any thread may make the first call (the lazy init was not thread safe)
    std::call_once(once, [] {
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
            r.version = (GetVersion_fn)dlsym(r.h, "ncclGetVersion");
            r.count = (CommCount_fn)dlsym(r.h, "ncclCommCount");
            r.user_rank = (CommUserRank_fn)dlsym(r.h, "ncclCommUserRank");
        }
    });
    return (r.h && r.get_id && r.init_rank && r.destroy && r.allreduce) ? &r : nullptr;
}

fx3d_status rccl_fail(int rc, const char *what) {
    Rccl *r = rccl();
    set_error("%s failed: %s", what, (r && r->errstr) ? r->errstr(rc) : "RCCL error");
    return FX3D_ERR_RCCL;
}

constexpr int kNcclSum = 0, kNcclFloat64 = 8;  // rccl.h: ncclSum = 0, ncclFloat64 = 8

// ---- unique-id exchange without torch / MPI: rank 0 hands the 128 bytes to the other ranks over a rendezvous ----------
//   "tcp://host:port"  rank 0 listens on `port` (all interfaces) and serves nranks-1 connections; the others connect to
//                      host:port, retrying while rank 0 is not up yet.  Nothing persists, a re-run cannot see stale state.
//   "file://path"      rank 0 writes path.tmp and renames it to `path`; the others poll for `path`.  Rank 0 removes the
//                      file again once every rank has confirmed (path.<rank> markers), so the next job starts clean.
constexpr int kBootTimeoutS = 120;

bool send_all(int fd, const void *buf, size_t n) {
    const char *p = static_cast<const char *>(buf);
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k <= 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}
bool recv_all(int fd, void *buf, size_t n) {
    char *p = static_cast<char *>(buf);
    while (n) {
        const ssize_t k = ::recv(fd, p, n, 0);
        if (k <= 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

fx3d_status boot_tcp(const std::string &host, int port, int nranks, int rank, uint8_t *id128) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(kBootTimeoutS);
    if (rank == 0) {
        const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) { set_error("fx3d_comm_exchange_id: socket() failed"); return FX3D_ERR_RCCL; }
        int one = 1;
        ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{};
        a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)port);
        if (::bind(ls, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || ::listen(ls, nranks) != 0) {
            ::close(ls);
            set_error("fx3d_comm_exchange_id: cannot listen on port %d", port);
            return FX3D_ERR_RCCL;
        }
        timeval tv{kBootTimeoutS, 0};
        ::setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));  // accept() honours it
        for (int served = 0; served < nranks - 1; ++served) {
            const int c = ::accept(ls, nullptr, nullptr);
            if (c < 0) { ::close(ls); set_error("fx3d_comm_exchange_id: %d of %d ranks connected within %d s", served, nranks - 1, kBootTimeoutS); return FX3D_ERR_RCCL; }
            int32_t peer = -1;
            const bool ok = recv_all(c, &peer, sizeof(peer)) && send_all(c, id128, 128);
            ::close(c);
            if (!ok || peer <= 0 || peer >= nranks) { ::close(ls); set_error("fx3d_comm_exchange_id: bad hand-shake from a peer"); return FX3D_ERR_RCCL; }
        }
        ::close(ls);
        return FX3D_OK;
    }
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
    if (::getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res) {
        set_error("fx3d_comm_exchange_id: cannot resolve %s", host.c_str());
        return FX3D_ERR_RCCL;
    }
    fx3d_status rc = FX3D_ERR_RCCL;
    while (std::chrono::steady_clock::now() < deadline) {
        const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            const int32_t me = rank;
            const bool ok = send_all(fd, &me, sizeof(me)) && recv_all(fd, id128, 128);
            ::close(fd);
            if (ok) { rc = FX3D_OK; break; }
        } else if (fd >= 0) {
            ::close(fd);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));  // rank 0 is not listening yet
    }
    ::freeaddrinfo(res);
    if (rc != FX3D_OK) set_error("fx3d_comm_exchange_id: rank %d could not reach rank 0 at %s:%d within %d s", rank, host.c_str(), port, kBootTimeoutS);
    return rc;
}

fx3d_status boot_file(const std::string &path, int nranks, int rank, uint8_t *id128) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(kBootTimeoutS);
    auto exists = [](const std::string &f) { struct stat st; return ::stat(f.c_str(), &st) == 0; };
    if (rank == 0) {
        const std::string tmp = path + ".tmp";
        FILE *fh = ::fopen(tmp.c_str(), "wb");
        if (!fh || ::fwrite(id128, 1, 128, fh) != 128) { if (fh) ::fclose(fh); set_error("fx3d_comm_exchange_id: cannot write %s", tmp.c_str()); return FX3D_ERR_RCCL; }
        ::fclose(fh);
        if (::rename(tmp.c_str(), path.c_str()) != 0) { set_error("fx3d_comm_exchange_id: cannot publish %s", path.c_str()); return FX3D_ERR_RCCL; }
        for (int r = 1; r < nranks; ++r) {  // wait for every reader, then leave nothing behind
            const std::string mark = path + "." + std::to_string(r);
            while (!exists(mark)) {
                if (std::chrono::steady_clock::now() > deadline) { set_error("fx3d_comm_exchange_id: rank %d never read %s", r, path.c_str()); return FX3D_ERR_RCCL; }
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
            ::unlink(mark.c_str());
        }
        ::unlink(path.c_str());
        return FX3D_OK;
    }
    while (!exists(path)) {
        if (std::chrono::steady_clock::now() > deadline) { set_error("fx3d_comm_exchange_id: %s did not appear within %d s", path.c_str(), kBootTimeoutS); return FX3D_ERR_RCCL; }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    FILE *fh = ::fopen(path.c_str(), "rb");
    const bool ok = fh && ::fread(id128, 1, 128, fh) == 128;
    if (fh) ::fclose(fh);
    if (!ok) { set_error("fx3d_comm_exchange_id: short read of %s", path.c_str()); return FX3D_ERR_RCCL; }
    const std::string mark = path + "." + std::to_string(rank);
    fh = ::fopen(mark.c_str(), "wb");
    if (fh) ::fclose(fh);
    return FX3D_OK;
}

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

fx3d_status fx3d_comm_exchange_id(uint8_t *id128, int32_t nranks, int32_t rank, const char *rendezvous) {
    FX3D_REQUIRE(id128 && rendezvous, "fx3d_comm_exchange_id: null pointer");
    FX3D_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, "fx3d_comm_exchange_id: bad rank %d of %d", rank, nranks);
    if (nranks == 1) return FX3D_OK;
    const std::string r(rendezvous);
    if (r.rfind("tcp://", 0) == 0) {
        const size_t colon = r.rfind(':');
        FX3D_REQUIRE(colon != std::string::npos && colon > 6, "fx3d_comm_exchange_id: expected tcp://host:port, got %s", rendezvous);
        const int port = atoi(r.c_str() + colon + 1);
        FX3D_REQUIRE(port > 0 && port < 65536, "fx3d_comm_exchange_id: bad port in %s", rendezvous);
        return boot_tcp(r.substr(6, colon - 6), port, nranks, rank, id128);
    }
    if (r.rfind("file://", 0) == 0) return boot_file(r.substr(7), nranks, rank, id128);
    set_error("fx3d_comm_exchange_id: rendezvous must be tcp://host:port or file://path, got %s", rendezvous);
    return FX3D_ERR_INVALID_ARG;
}

fx3d_status fx3d_comm_bootstrap(fx3d_comm_t *comm, int32_t nranks, int32_t rank, const char *rendezvous) {
    FX3D_REQUIRE(comm && rendezvous, "fx3d_comm_bootstrap: null pointer");
    FX3D_REQUIRE(nranks > 0 && rank >= 0 && rank < nranks, "fx3d_comm_bootstrap: bad rank %d of %d", rank, nranks);
    uint8_t id[128];
    memset(id, 0, sizeof(id));
    fx3d_status rc = FX3D_OK;
    if (rank == 0) {
        rc = fx3d_comm_unique_id(id);
        if (rc) return rc;
    }
    rc = fx3d_comm_exchange_id(id, nranks, rank, rendezvous);
    if (rc) return rc;
    return fx3d_comm_init_rank(comm, nranks, id, rank);
}

fx3d_status fx3d_comm_info(fx3d_comm_t comm, int32_t *nranks, int32_t *rank, int32_t *rccl_version) {
    Rccl *r = rccl();
    if (!r) { set_error("librccl could not be loaded"); return FX3D_ERR_RCCL; }
    if (rccl_version) {
        int v = 0;
        if (r->version && r->version(&v) == 0) *rccl_version = v; else *rccl_version = 0;
    }
    if (nranks || rank) {
        FX3D_REQUIRE(comm, "fx3d_comm_info: null communicator");
        int n = 0, u = 0;
        if (!r->count || !r->user_rank) { set_error("fx3d_comm_info: librccl lacks ncclCommCount / ncclCommUserRank"); return FX3D_ERR_RCCL; }
        int rc = r->count(comm, &n);
        if (rc) return rccl_fail(rc, "ncclCommCount");
        rc = r->user_rank(comm, &u);
        if (rc) return rccl_fail(rc, "ncclCommUserRank");
        if (nranks) *nranks = n;
        if (rank) *rank = u;
    }
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

// The same with the collective off the compute stream: the kernel runs on `s`, the 16-byte all-reduce and the
// finalisation on `comm_stream` behind `ready`; `done` marks the loss.  Nothing waits on the host, so the NEXT
// evaluation's kernel on `s` overlaps this evaluation's collective (the caller rotates sums_dev / loss_dev / events
// over a few slots and makes `s` wait for a slot's `done` before reusing it).
fx3d_status fx3d_chamfer_fwd_sharded_async(fx3d_comm_t comm, const float *x, int32_t N, const float *y, int32_t M,
                                           int32_t B_local, int32_t D, int64_t B_global, float w1, float w2,
                                           double *sums_dev, float *loss_dev, void *ws, size_t ws_bytes,
                                           fx3d_stream_t s, fx3d_stream_t comm_stream, fx3d_event_t ready,
                                           fx3d_event_t done) {
    FX3D_REQUIRE(comm && sums_dev && loss_dev && ready && done, "fx3d_chamfer_fwd_sharded_async: null pointer");
    FX3D_REQUIRE(B_global >= B_local && B_local >= 0, "fx3d_chamfer_fwd_sharded_async: bad batch sizes");
    FX3D_REQUIRE(comm_stream && comm_stream != s, "fx3d_chamfer_fwd_sharded_async: the collective needs a created stream of its own");
    fx3d_status rc;
    // slot reuse: the kernel below overwrites sums_dev, which the slot's previous collective (marked by `done`) may
    // still be reading on comm_stream.  Waiting on an event that was never recorded is a no-op.
    FX3D_HIP(hipStreamWaitEvent(as_stream(s), reinterpret_cast<hipEvent_t>(done), 0));
    if (B_local > 0) {
        rc = fx3d_chamfer_sums(x, N, y, M, B_local, D, sums_dev, nullptr, nullptr, ws, ws_bytes, s);
        if (rc) return rc;
    } else {
        FX3D_HIP(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), as_stream(s)));
    }
    FX3D_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(ready), as_stream(s)));
    FX3D_HIP(hipStreamWaitEvent(as_stream(comm_stream), reinterpret_cast<hipEvent_t>(ready), 0));
    rc = fx3d_comm_allreduce_sum_f64(comm, sums_dev, 2, comm_stream);
    if (rc) return rc;
    rc = fx3d_chamfer_finalize(sums_dev, N, M, B_global, D, w1, w2, loss_dev, comm_stream);
    if (rc) return rc;
    FX3D_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(done), as_stream(comm_stream)));
    return FX3D_OK;
}

}  // extern "C"

// RCCL behind the C ABI: the one collective of the path without torch.
//
// The reference has no multi-device code; a Julia (or any C-ABI) caller that shards the
// batch over the 8 GPUs of a node needs exactly: a communicator, and `all-reduce(sum)` of the two
// Float64 partial sums between fx3d_chamfer_sums and fx3d_chamfer_finalize.  librccl is resolved at
// run time (dlopen) so the library also loads on hosts without RCCL; one process per GPU, the unique id
// travels through whatever the host already has (MPI, a file, torch.distributed, Julia Distributed).
#include <arpa/inet.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

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
typedef int (*CommInitAll_fn)(ncclComm_t_ *, int, const int *);
typedef int (*Group_fn)(void);

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
    CommInitAll_fn init_all = nullptr;
    Group_fn group_start = nullptr, group_end = nullptr;
    CommDestroy_fn abort = nullptr;  // ncclCommAbort: same signature as ncclCommDestroy
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
            r.init_all = (CommInitAll_fn)dlsym(r.h, "ncclCommInitAll");
            r.group_start = (Group_fn)dlsym(r.h, "ncclGroupStart");
            r.group_end = (Group_fn)dlsym(r.h, "ncclGroupEnd");
            r.abort = (CommDestroy_fn)dlsym(r.h, "ncclCommAbort");
        }
    });
    return (r.h && r.get_id && r.init_rank && r.destroy && r.allreduce) ? &r : nullptr;
}

fx3d_status rccl_fail(int rc, const char *what) {
    Rccl *r = rccl();
    set_error("%s failed: %s", what, (r && r->errstr) ? r->errstr(rc) : "RCCL error");
    return FX3D_ERR_RCCL;
}

constexpr int kNcclSum = 0, kNcclMax = 2, kNcclFloat64 = 8;  // rccl.h: ncclSum = 0, ncclMax = 2, ncclFloat64 = 8

// ---- unique-id exchange without torch / MPI: rank 0 hands the 128 bytes to the other ranks over a rendezvous ----------
//   "tcp://host:port"  rank 0 listens on `port` (loopback only when `host` is a loopback name, else all interfaces) until every
//                      rank 1..n-1 has been served once; the others connect to host:port, retrying while rank 0 is not up
//                      yet.  Nothing persists, a re-run cannot see stale state.  Hand-shake: {magic, token, nranks, rank}
//                      -> {magic, id}; a connection that does not speak it (a port scanner, a health probe, a peer of
//                      another job: wrong token / nranks) is dropped and rank 0 keeps accepting; a rank that connects twice
//                      is served again but counted once.  The token is FX3D_COMM_TOKEN (any string the launcher exports to
//                      all ranks; the only real isolation), a hash of "port:nranks" without it.
//   "file://path"      single node.  Rank 0 removes whatever an earlier job left at `path` / `path.<r>`, writes
//                      {magic, nonce, wall-clock, nranks, id} to a fresh temp file (O_EXCL | O_NOFOLLOW, 0600) and renames it
//                      to `path`; the others poll for `path`, refuse payloads older than kStaleS or for another world
//                      size, and confirm with a marker `path.<r>` that carries the nonce.  Rank 0 waits for every marker
//                      OF THIS NONCE (a stale one is unlinked and waited for again), then removes all of it -- also when it
//                      gives up.  (A reader that beats rank 0 to a younger-than-kStaleS leftover of a crashed job with the
//                      same path still gets that job's id: keep the path unique per job -- default_rendezvous keys it by
//                      launcher pid and port -- or use tcp://.)
constexpr int kBootTimeoutS = 120;
constexpr int kStaleS = 300;
constexpr uint64_t kMagic = 0x3144495544335846ull;  // "FX3DUID1"

struct Hello { uint64_t magic, token; int32_t nranks, rank; };
struct FilePayload { uint64_t magic, nonce; int64_t wall; int32_t nranks, pad; uint8_t id[128]; };

uint64_t job_token(const std::string &host, int port, int nranks) {
    // FNV-1a of FX3D_COMM_TOKEN when the job exports one -- the ONLY form that isolates a job from a peer that can reach the
    // port; otherwise of "port:nranks": every rank derives it from its rendezvous string and it only keeps ACCIDENTAL cross-job
    // connections out (another port or world size; anybody can compute it).  The host spelling is NOT hashed: ranks may name
    // one endpoint differently (0.0.0.0 / a hostname on rank 0, MASTER_ADDR elsewhere) and must still agree.
    (void)host;
    const char *e = getenv("FX3D_COMM_TOKEN");
    const std::string dflt = std::to_string(port) + ":" + std::to_string(nranks);
    const char *src = (e && *e) ? e : dflt.c_str();
    uint64_t h = 1469598103934665603ull;
    for (; *src; ++src) { h ^= (unsigned char)*src; h *= 1099511628211ull; }
    return h;
}

uint64_t fresh_nonce() {
    uint64_t n = 0;
    const int fd = ::open("/dev/urandom", O_RDONLY | O_CLOEXEC);
    if (fd >= 0) { if (::read(fd, &n, sizeof(n)) != (ssize_t)sizeof(n)) n = 0; ::close(fd); }
    n ^= (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
    n ^= (uint64_t)::getpid() << 32;
    return n ? n : 1;
}

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
    const uint64_t token = job_token(host, port, nranks);
    if (rank == 0) {
        const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) { set_error("fx3d_comm_exchange_id: socket() failed"); return FX3D_ERR_RCCL; }
        int one = 1;
        ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{};
        const bool loopback = host == "127.0.0.1" || host == "localhost";
        a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(loopback ? INADDR_LOOPBACK : INADDR_ANY); a.sin_port = htons((uint16_t)port);
        if (::bind(ls, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || ::listen(ls, nranks + 8) != 0) {
            ::close(ls);
            set_error("fx3d_comm_exchange_id: cannot listen on port %d (in use?): choose another one with FX3D_COMM_RENDEZVOUS=tcp://host:port, "
                      "FX3D_COMM_PORT_OFFSET, or a file:// rendezvous", port);
            return FX3D_ERR_RCCL;
        }
        std::vector<char> served((size_t)nranks, 0);
        int nserved = 0;
        while (nserved < nranks - 1) {
            const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
            if (left <= 0) break;
            timeval tv{(time_t)(left / 1000), (suseconds_t)((left % 1000) * 1000)};
            ::setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));  // accept() honours it
            const int c = ::accept(ls, nullptr, nullptr);
            if (c < 0) continue;  // timed out or interrupted: the loop head decides
            timeval ptv{5, 0};    // a peer that connects and stays silent must not hold the others up
            ::setsockopt(c, SOL_SOCKET, SO_RCVTIMEO, &ptv, sizeof(ptv));
            Hello h{};
            const bool ok = recv_all(c, &h, sizeof(h)) && h.magic == kMagic && h.token == token && h.nranks == nranks &&
                            h.rank > 0 && h.rank < nranks;
            if (ok && send_all(c, &kMagic, sizeof(kMagic)) && send_all(c, id128, 128) && !served[(size_t)h.rank]) {
                served[(size_t)h.rank] = 1;
                ++nserved;
            }
            ::close(c);  // (anything else: dropped, keep accepting)
        }
        ::close(ls);
        if (nserved < nranks - 1) {
            set_error("fx3d_comm_exchange_id: %d of %d ranks connected within %d s", nserved, nranks - 1, kBootTimeoutS);
            return FX3D_ERR_RCCL;
        }
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
        if (fd >= 0) {  // a service that accepts on this port and stays silent must not hold this rank beyond the deadline
            timeval tv{5, 0};
            ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
            ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
        }
        if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            const Hello h{kMagic, token, nranks, rank};
            uint64_t magic = 0;
            const bool ok = send_all(fd, &h, sizeof(h)) && recv_all(fd, &magic, sizeof(magic)) && magic == kMagic && recv_all(fd, id128, 128);
            ::close(fd);
            if (ok) { rc = FX3D_OK; break; }
        } else if (fd >= 0) {
            ::close(fd);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));  // rank 0 is not listening yet (or refused us: another job's port)
    }
    ::freeaddrinfo(res);
    if (rc != FX3D_OK) set_error("fx3d_comm_exchange_id: rank %d could not reach rank 0 at %s:%d within %d s (refused or unreachable: every rank must "
                                 "use the same port, world size and FX3D_COMM_TOKEN -- the token is the only real isolation between jobs)", rank, host.c_str(), port, kBootTimeoutS);
    return rc;
}

bool write_new_file(const std::string &f, const void *buf, size_t n) {  // never through a symlink, never over an existing file
    ::unlink(f.c_str());
    const int fd = ::open(f.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return false;
    const bool ok = ::write(fd, buf, n) == (ssize_t)n;
    ::close(fd);
    if (!ok) ::unlink(f.c_str());
    return ok;
}
bool read_file(const std::string &f, void *buf, size_t n) {
    const int fd = ::open(f.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    const bool ok = ::read(fd, buf, n) == (ssize_t)n;
    ::close(fd);
    return ok;
}

fx3d_status boot_file(const std::string &path, int nranks, int rank, uint8_t *id128) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(kBootTimeoutS);
    auto marker = [&](int r) { return path + "." + std::to_string(r); };
    auto okfile = [&](int r) { return path + "." + std::to_string(r) + ".ok"; };
    auto gone = [](const std::string &f) { struct stat st; return ::lstat(f.c_str(), &st) != 0; };
    if (rank == 0) {
        auto cleanup = [&] {
            ::unlink(path.c_str());
            ::unlink((path + ".tmp").c_str());
            for (int r = 1; r < nranks; ++r) { ::unlink(marker(r).c_str()); ::unlink(okfile(r).c_str()); }
        };
        cleanup();  // whatever an earlier job left behind
        FilePayload pl{};
        pl.magic = kMagic; pl.nonce = fresh_nonce(); pl.wall = (int64_t)::time(nullptr); pl.nranks = nranks;
        memcpy(pl.id, id128, 128);
        const std::string tmp = path + ".tmp";
        if (!write_new_file(tmp, &pl, sizeof(pl))) { set_error("fx3d_comm_exchange_id: cannot write %s", tmp.c_str()); return FX3D_ERR_RCCL; }
        if (::rename(tmp.c_str(), path.c_str()) != 0) { cleanup(); set_error("fx3d_comm_exchange_id: cannot publish %s", path.c_str()); return FX3D_ERR_RCCL; }
        for (int r = 1; r < nranks; ++r) {  // every reader confirms with THIS job's nonce ...
            for (;;) {
                uint64_t got = 0;
                if (read_file(marker(r), &got, sizeof(got))) {
                    if (got == pl.nonce) break;
                    ::unlink(marker(r).c_str());  // a marker of another job (its reader saw a stale file): the reader starts over
                }
                if (std::chrono::steady_clock::now() > deadline) {
                    cleanup();
                    set_error("fx3d_comm_exchange_id: rank %d never confirmed %s", r, path.c_str());
                    return FX3D_ERR_RCCL;
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        // ... is told so (only now may it use the id), takes its two files away, and rank 0 removes the rest
        for (int r = 1; r < nranks; ++r)
            if (!write_new_file(okfile(r), &pl.nonce, sizeof(pl.nonce))) { cleanup(); set_error("fx3d_comm_exchange_id: cannot write %s", okfile(r).c_str()); return FX3D_ERR_RCCL; }
        for (int r = 1; r < nranks; ++r)
            while (!gone(okfile(r))) {
                if (std::chrono::steady_clock::now() > deadline) { cleanup(); set_error("fx3d_comm_exchange_id: rank %d never picked its acknowledgement up", r); return FX3D_ERR_RCCL; }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        cleanup();
        return FX3D_OK;
    }
    uint64_t confirmed = 0;
    for (;;) {
        FilePayload pl{};
        if (read_file(path, &pl, sizeof(pl)) && pl.magic == kMagic && pl.nranks == nranks &&
            (int64_t)::time(nullptr) - pl.wall <= kStaleS && pl.nonce != confirmed) {
            // a first payload, or rank 0 replaced a stale one after we had confirmed that
            memcpy(id128, pl.id, 128);
            if (!write_new_file(marker(rank), &pl.nonce, sizeof(pl.nonce))) { set_error("fx3d_comm_exchange_id: cannot write %s", marker(rank).c_str()); return FX3D_ERR_RCCL; }
            confirmed = pl.nonce;
        }
        if (confirmed) {
            uint64_t ack = 0;
            if (read_file(okfile(rank), &ack, sizeof(ack)) && ack == confirmed) {
                ::unlink(marker(rank).c_str());
                ::unlink(okfile(rank).c_str());
                return FX3D_OK;
            }
            if (gone(marker(rank))) confirmed = 0;  // rank 0 wiped it (it belonged to a stale payload): confirm again
        }
        if (std::chrono::steady_clock::now() > deadline) {
            ::unlink(marker(rank).c_str());
            set_error("fx3d_comm_exchange_id: %s: no (fresh) unique id from rank 0 within %d s", path.c_str(), kBootTimeoutS);
            return FX3D_ERR_RCCL;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
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

fx3d_status fx3d_comm_allreduce_max_f64(fx3d_comm_t comm, double *buf_dev, int64_t count, fx3d_stream_t s) {
    FX3D_REQUIRE(comm && buf_dev && count > 0, "fx3d_comm_allreduce_max_f64: bad argument");
    Rccl *r = rccl();
    if (!r) { set_error("librccl could not be loaded"); return FX3D_ERR_RCCL; }
    const int rc = r->allreduce(buf_dev, buf_dev, (size_t)count, kNcclFloat64, kNcclMax, comm, as_stream(s));
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

// ---- one process, several devices (fx3d_comm_init_all) ---------------------------------------------------
// The reference is ONE Julia process (src/metrics/pcloud.jl:54-70); with the per-rank entry points above such a host
// would have to spawn a process per GPU to shard a batch.  fx3d_comm_init_all builds the communicators of `ndev` devices
// of this process (ncclCommInitAll) plus one worker thread, one stream and the scratch of a sharded evaluation per device;
// fx3d_chamfer_fwd_multi hands every worker its shard: kernel -> all-reduce(sum) of the two Float64 -> finalise with the
// global batch size, each on its device's stream, the host thread of the caller only waits for the enqueues (one thread
// per device: eight launches of ~10 us each from a single thread would make a 50 us evaluation host bound; one thread per
// communicator is RCCL's supported single-process form, no group call needed).
namespace {

struct Worker {
    int dev = 0;
    ncclComm_t_ comm = nullptr;
    hipStream_t stream = nullptr;
    void *ws = nullptr;
    size_t ws_bytes = 0;
    double *sums = nullptr;
    float *loss = nullptr;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<fx3d_status()> job;
    bool has_job = false, busy = false, quit = false;
    fx3d_status rc = FX3D_OK;
    fx3d_status rc0 = FX3D_OK;   // the worker's own setup (device, stream, result buffers); written by its thread before any job runs
    std::string err;

    void loop() {
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&sums), 2 * sizeof(double)) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&loss), sizeof(float)) != hipSuccess)
            rc0 = FX3D_ERR_HIP;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) break;
            std::function<fx3d_status()> j = std::move(job);
            has_job = false;
            lk.unlock();
            fx3d_status r = rc0 ? rc0 : j();
            std::string e;
            if (r) {
                char buf[512];
                buf[0] = 0;
                if (rc0) snprintf(buf, sizeof(buf), "device %d: stream / scratch setup failed", dev);
                else fx3d_last_error(buf, sizeof(buf));
                e = buf;
            }
            lk.lock();
            rc = r; err = std::move(e); busy = false;
            cv.notify_all();
        }
        if (ws) (void)hipFree(ws);
        if (sums) (void)hipFree(sums);
        if (loss) (void)hipFree(loss);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void post(std::function<fx3d_status()> j) {
        std::lock_guard<std::mutex> lk(mu);
        job = std::move(j); has_job = true; busy = true;
        cv.notify_all();
    }
    fx3d_status wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !busy; });
        return rc;
    }
};

struct Multi {
    std::vector<Worker *> w;
    std::vector<int> devs;
    // set when an evaluation failed AFTER some device had enqueued its collective: those collectives have no partner, the
    // communicators are aborted and rebuilt before the next evaluation; if the rebuild fails the handle stays
    // broken and every later call says so
    bool broken = false;
};

fx3d_status run_all(Multi *m, const std::function<fx3d_status(Worker &, int)> &f) {
    for (size_t d = 0; d < m->w.size(); ++d) {
        Worker *w = m->w[d];
        const int di = (int)d;
        w->post([w, di, &f] { return f(*w, di); });
    }
    fx3d_status rc = FX3D_OK;
    std::string err;
    for (Worker *w : m->w) {  // (every worker is waited for: `f` is referenced until the last one returns)
        const fx3d_status r = w->wait();
        if (r && !rc) { rc = r; err = w->err; }
    }
    if (rc) set_error("%s", err.c_str());
    return rc;
}

// After a failed phase 2: end every pending collective and drop the communicators (ncclCommAbort where the library has it,
// else ncclCommDestroy after the streams drained as far as they can).  The handle is marked broken.
void multi_abort_comms(Multi *m) {
    Rccl *r = rccl();
    for (Worker *w : m->w) {
        if (r && w->comm) (void)(r->abort ? r->abort(w->comm) : r->destroy(w->comm));
        w->comm = nullptr;
    }
    m->broken = true;
}

fx3d_status multi_rebuild_comms(Multi *m) {
    Rccl *r = rccl();
    if (!r || !r->init_all) { set_error("fx3d_chamfer_fwd_multi: the communicators were aborted after a failed evaluation and librccl cannot rebuild them"); return FX3D_ERR_RCCL; }
    for (Worker *w : m->w) (void)w->wait();
    int prev = 0;
    (void)hipGetDevice(&prev);
    std::vector<ncclComm_t_> comms(m->w.size(), nullptr);
    const int rc = r->init_all(comms.data(), (int)m->w.size(), m->devs.data());
    (void)hipSetDevice(prev);
    if (rc) {
        set_error("fx3d_chamfer_fwd_multi: the communicators were aborted after a failed evaluation; ncclCommInitAll failed to rebuild them (%d): "
                  "destroy this handle (fx3d_multi_destroy) and create a new one", rc);
        return FX3D_ERR_RCCL;
    }
    for (size_t d = 0; d < m->w.size(); ++d) m->w[d]->comm = comms[d];
    m->broken = false;
    return FX3D_OK;
}

}  // namespace

extern "C" {

fx3d_status fx3d_comm_init_all(fx3d_multi_t *multi, int32_t ndev, const int32_t *devices) {
    FX3D_REQUIRE(multi && ndev > 0 && ndev <= 64, "fx3d_comm_init_all: bad argument (ndev=%d)", ndev);
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) { set_error("fx3d_comm_init_all: no HIP device"); return FX3D_ERR_NO_DEVICE; }
    std::vector<int> devs((size_t)ndev);
    for (int d = 0; d < ndev; ++d) {
        devs[(size_t)d] = devices ? devices[d] : d;
        FX3D_REQUIRE(devs[(size_t)d] >= 0 && devs[(size_t)d] < have, "fx3d_comm_init_all: device %d of %d", devs[(size_t)d], have);
        for (int e = 0; e < d; ++e) FX3D_REQUIRE(devs[(size_t)e] != devs[(size_t)d], "fx3d_comm_init_all: device %d listed twice", devs[(size_t)d]);
    }
    Rccl *r = rccl();
    if (!r || !r->init_all) { set_error("librccl (ncclCommInitAll) could not be loaded"); return FX3D_ERR_RCCL; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    std::vector<ncclComm_t_> comms((size_t)ndev, nullptr);
    const int rc = r->init_all(comms.data(), ndev, devs.data());
    (void)hipSetDevice(prev);  // (ncclCommInitAll walks the devices)
    if (rc) return rccl_fail(rc, "ncclCommInitAll");
    Multi *m = new Multi;
    m->devs = devs;
    for (int d = 0; d < ndev; ++d) {
        Worker *w = new Worker;
        w->dev = devs[(size_t)d];
        w->comm = comms[(size_t)d];
        w->th = std::thread([w] { w->loop(); });
        m->w.push_back(w);
    }
    *multi = m;
    return FX3D_OK;
}

fx3d_status fx3d_multi_destroy(fx3d_multi_t multi) {
    if (!multi) return FX3D_OK;
    Multi *m = static_cast<Multi *>(multi);
    Rccl *r = rccl();
    for (Worker *w : m->w) {
        (void)w->wait();
        { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
        if (r && w->comm) (void)r->destroy(w->comm);
        delete w;
    }
    delete m;
    return FX3D_OK;
}

fx3d_status fx3d_multi_info(fx3d_multi_t multi, int32_t *ndev, int32_t *devices, int32_t *rccl_version) {
    FX3D_REQUIRE(multi, "fx3d_multi_info: null handle");
    Multi *m = static_cast<Multi *>(multi);
    if (ndev) *ndev = (int32_t)m->w.size();
    if (devices) for (size_t d = 0; d < m->w.size(); ++d) devices[d] = m->w[d]->dev;
    if (rccl_version) return fx3d_comm_info(nullptr, nullptr, nullptr, rccl_version);
    return FX3D_OK;
}

fx3d_status fx3d_multi_sync(fx3d_multi_t multi) {
    FX3D_REQUIRE(multi, "fx3d_multi_sync: null handle");
    return run_all(static_cast<Multi *>(multi), [](Worker &w, int) -> fx3d_status {
        FX3D_HIP(hipStreamSynchronize(w.stream));
        return FX3D_OK;
    });
}

// x[d], y[d]: device d's shard ((D,N,B_local[d]) / (D,M,B_local[d]) in ITS memory; ignored where B_local[d] == 0).
// loss_host (optional): the global loss, read back from the first device (blocks until it is there); losses_dev
// (optional, [ndev] device pointers in each device's memory): every device's copy of it, valid after fx3d_multi_sync.
fx3d_status fx3d_chamfer_fwd_multi(fx3d_multi_t multi, const float *const *x, int32_t N, const float *const *y, int32_t M,
                                   const int32_t *B_local, int32_t D, int64_t B_global, float w1, float w2,
                                   float *loss_host, float *const *losses_dev) {
    FX3D_REQUIRE(multi && x && y && B_local, "fx3d_chamfer_fwd_multi: null pointer");
    FX3D_REQUIRE(N > 0 && M > 0 && D > 0 && B_global > 0, "fx3d_chamfer_fwd_multi: bad sizes");
    Multi *m = static_cast<Multi *>(multi);
    long long tot = 0;
    for (size_t d = 0; d < m->w.size(); ++d) {
        FX3D_REQUIRE(B_local[d] >= 0, "fx3d_chamfer_fwd_multi: negative shard size on device slot %zu", d);
        FX3D_REQUIRE(B_local[d] == 0 || (x[d] && y[d]), "fx3d_chamfer_fwd_multi: null shard on device slot %zu", d);
        tot += B_local[d];
    }
    FX3D_REQUIRE(tot <= B_global, "fx3d_chamfer_fwd_multi: the shards hold %lld clouds, B_global = %lld", tot, (long long)B_global);
    if (m->broken) {
        const fx3d_status rrc = multi_rebuild_comms(m);
        if (rrc) return rrc;
    }
    // Phase 1 on every device, NOTHING enqueued: the worker's setup status, the scratch query and (re)allocation.  Only when
    // every device passed does phase 2 enqueue kernel -> all-reduce -> finalise: a device that fails before its collective
    // would leave the others' collectives without a partner (the host thread of device 0 would wait for ever).
    fx3d_status rc = run_all(m, [&](Worker &w, int d) -> fx3d_status {
        const int Bl = B_local[d];
        if (Bl <= 0) return FX3D_OK;
        size_t need = 0;
        const fx3d_status r1 = fx3d_chamfer_workspace_bytes(N, M, Bl, D, &need);
        if (r1) return r1;
        if (need > w.ws_bytes) {
            FX3D_HIP(hipStreamSynchronize(w.stream));  // (an earlier evaluation may still use the old scratch)
            if (w.ws) FX3D_HIP(hipFree(w.ws));
            w.ws = nullptr; w.ws_bytes = 0;
            FX3D_HIP(hipMalloc(&w.ws, need));
            w.ws_bytes = need;
        }
        return FX3D_OK;
    });
    if (rc) return rc;
    std::atomic<int> enq_collectives{0};
    rc = run_all(m, [&](Worker &w, int d) -> fx3d_status {
        const int Bl = B_local[d];
        fx3d_stream_t st = reinterpret_cast<fx3d_stream_t>(w.stream);
        if (Bl > 0) {
            const fx3d_status r1 = fx3d_chamfer_sums(x[d], N, y[d], M, Bl, D, w.sums, nullptr, nullptr, w.ws, w.ws_bytes, st);
            if (r1) return r1;
        } else {
            FX3D_HIP(hipMemsetAsync(w.sums, 0, 2 * sizeof(double), w.stream));
        }
        fx3d_status r2 = fx3d_comm_allreduce_sum_f64(w.comm, w.sums, 2, st);
        if (r2) return r2;
        enq_collectives.fetch_add(1);
        float *out = losses_dev && losses_dev[d] ? losses_dev[d] : w.loss;
        r2 = fx3d_chamfer_finalize(w.sums, N, M, B_global, D, w1, w2, out, st);
        if (r2) return r2;
        return FX3D_OK;
    });
    if (rc) {
        // a device failed in phase 2: if any other device has its all-reduce enqueued, that collective will never complete.
        // Abort the communicators (ncclCommAbort ends the pending collectives) and mark the handle: the next evaluation
        // rebuilds them.  Never a half-enqueued state handed back as if it were usable.
        char keep[512];
        keep[0] = 0;
        fx3d_last_error(keep, sizeof(keep));
        if (enq_collectives.load() > 0 && enq_collectives.load() < (int)m->w.size()) multi_abort_comms(m);
        set_error("%s", keep);
        return rc;
    }
    if (loss_host) {  // every device has enqueued its collective: reading device 0's result back cannot wait for a missing partner
        Worker &w0 = *m->w[0];
        float *out = losses_dev && losses_dev[0] ? losses_dev[0] : w0.loss;
        rc = run_all(m, [&](Worker &w, int d) -> fx3d_status {
            if (d != 0) return FX3D_OK;
            FX3D_HIP(hipMemcpyAsync(loss_host, out, sizeof(float), hipMemcpyDeviceToHost, w.stream));
            FX3D_HIP(hipStreamSynchronize(w.stream));
            return FX3D_OK;
        });
        (void)w0;
    }
    return rc;
}

}  // extern "C"

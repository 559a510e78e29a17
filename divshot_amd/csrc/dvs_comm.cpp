// dvs_comm.cpp — include/dvs_comm.h: RCCL behind a plain-C surface (data-parallel exchange of the training step),
// plus a host-staged TEST backend over the bootstrap sockets (DVS_COMM_BACKEND=tcp) so that two ranks can share one GPU in the test suite.
// librccl is dlopen()ed on first use; the unique id travels from rank 0 to the other ranks over TCP on a dedicated bootstrap port.
#include <hip/hip_runtime.h>
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <sys/time.h>
#include "../../include/dvs_comm.h"
#include "../../include/dvs_raster.h"

// the slice of rccl.h this file needs (types and enum values as /opt/rocm/include/rccl/rccl.h, RCCL 2.x ABI)
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { rcclSum = 0, rcclMax = 2 };
enum { rcclUint8 = 1, rcclInt32 = 2, rcclFloat32 = 7 };
}

void dvs_set_last_error(const char* msg);      // dvs_api.cpp

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;          // (optional: what the COMMUNICATOR says its size is)
    bool load(std::string& err) {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (handle) break;
        }
        if (!handle) { err = std::string("cannot dlopen librccl: ") + dlerror(); return false; }
#define SYM(field, name) field = (decltype(field))dlsym(handle, name); if (!field) { err = "librccl lacks " name; return false; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(GetErrorString, "ncclGetErrorString") SYM(AllReduce, "ncclAllReduce") SYM(ReduceScatter, "ncclReduceScatter")
        SYM(AllGather, "ncclAllGather") SYM(Broadcast, "ncclBroadcast") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
#undef SYM
        CommCount = (decltype(CommCount))dlsym(handle, "ncclCommCount");
        return true;
    }
};
Rccl g_rccl;

bool send_all(int fd, const void* p, size_t n) {
    const char* c = (const char*)p;
    while (n) { ssize_t k = ::send(fd, c, n, 0); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}
bool recv_all(int fd, void* p, size_t n) {
    char* c = (char*)p;
    while (n) { ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) return false; c += k; n -= (size_t)k; }
    return true;
}
// Bootstrap of the RCCL unique id over TCP. Rank 0 listens on the bootstrap port; every other rank connects (retrying while rank 0 is
// not up yet), introduces itself with {magic, job nonce, rank}, gets {magic, id} back, acknowledges it and waits for rank 0's one-byte
// confirmation (a peer whose acknowledgement rank 0 failed to receive must not proceed as if it had been counted). Rank 0
// counts a rank once (when the confirmation is out; a rank that timed out on any step asks again while the listener is open) and keeps accepting until all
// world-1 distinct ranks have acknowledged: a stray or stale connection (wrong magic / nonce / rank, or one that sends nothing) is
// dropped without using up a slot, a hello with our magic but the wrong nonce is logged. Every socket operation has a timeout and the whole exchange a deadline
// (DVS_COMM_TIMEOUT_S, default 180 s), after which it fails with an error instead of hanging.
constexpr uint32_t kMagic = 0x44565343u;            // "DVSC"
struct Hello { uint32_t magic; uint32_t rank; uint64_t nonce; };
struct Reply { uint32_t magic; uint32_t pad; char id[128]; };

void set_timeouts(int fd, int seconds) {
    timeval tv{}; tv.tv_sec = seconds; tv.tv_usec = 0;
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
}
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// keep_fds (TCP backend): the connections stay open — rank 0 gets one socket per peer (index = rank), a peer its socket to rank 0 at [0].
bool exchange_id(void* id128, int rank, int world, const char* addr, int port, uint64_t nonce, double timeout_s, std::string& err,
                 std::vector<int>* keep_fds = nullptr) {
    if (world == 1) return true;
    if (keep_fds) keep_fds->assign(rank == 0 ? (size_t)world : 1u, -1);
    const double deadline = now_s() + timeout_s;
    if (rank == 0) {
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) { err = "socket() failed"; return false; }
        int one = 1; setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in sa{}; sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port); sa.sin_addr.s_addr = htonl(INADDR_ANY);
        // listen on the rendezvous address only when it is a literal IPv4 address of this host (127.0.0.1 in single-node runs)
        in_addr lit{};
        if (addr && inet_pton(AF_INET, addr, &lit) == 1) {
            sa.sin_addr = lit;
            if (::bind(ls, (sockaddr*)&sa, sizeof sa) != 0) sa.sin_addr.s_addr = htonl(INADDR_ANY);
            else goto bound;
        }
        if (::bind(ls, (sockaddr*)&sa, sizeof sa) != 0) {
            err = "cannot bind the bootstrap port " + std::to_string(port) + " (in use? set DVS_COMM_PORT)"; ::close(ls); return false;
        }
    bound:
        if (::listen(ls, world + 8) != 0) { err = "listen() on the bootstrap port failed"; ::close(ls); return false; }
        set_timeouts(ls, 1);                                   // accept() wakes up once a second to check the deadline
        std::vector<bool> served((size_t)world, false);
        int left = world - 1;
        // the unrecoverable failure used to sit on the LAST message — a peer whose receive of the confirmation fails after we
        // counted it would retry against a closed listener until its deadline while every other rank already waits in ncclCommInitRank
        // (which has no timeout). Two measures: the peer waits for the confirmation with the whole remaining deadline (below), and the
        // listener stays open for a grace period after the last rank was counted and answers already-served ranks again.
        double grace_until = 0.0;
        const double grace_s = [] { const char* e = getenv("DVS_COMM_GRACE_S"); const double v = e ? atof(e) : 1.0; return v >= 0.0 ? v : 1.0; }();
        while (left > 0 || now_s() < grace_until) {
            if (left > 0 && now_s() > deadline) { err = "timed out waiting for " + std::to_string(left) + " rank(s) to fetch the RCCL id"; ::close(ls); return false; }
            if (left == 0) { timeval tv{}; tv.tv_usec = 100000; setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv); }      // grace period: 0.1 s ticks
            int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) continue;                              // timeout tick / transient error
            set_timeouts(fd, 5);
            Hello h{};
            const bool got = recv_all(fd, &h, sizeof h);
            const bool ok = got && h.magic == kMagic && h.nonce == nonce && h.rank >= 1 && h.rank < (uint32_t)world;
            bool kept = false;
            if (ok) {
                // A rank that was served already may ask again (its 5 s receive can time out after our send_all returned): answer it
                // again — it only counts once.
                Reply r{}; r.magic = kMagic; memcpy(r.id, id128, 128);
                uint32_t ack = 0;                              // the peer confirms it holds the id, we confirm that its slot counts: only
                const uint8_t confirm = 1;                     // a peer that has read this last byte returns "ok" — if our receive of the ack
                                                               // fails the byte is never sent, the peer times out on it and asks again
                if (send_all(fd, &r, sizeof r) && recv_all(fd, &ack, sizeof ack) && ack == kMagic && send_all(fd, &confirm, 1)) {
                    if (!served[h.rank]) { served[h.rank] = true; --left; if (left == 0) grace_until = std::min(deadline, now_s() + grace_s); }
                    if (keep_fds) { int& slot = (*keep_fds)[h.rank]; if (slot >= 0) ::close(slot); slot = fd; kept = true; }
                }
            } else if (got && h.magic == kMagic) {
                fprintf(stderr, "[dvs_comm] rank 0: dropped a hello from rank %u (%s) — do all ranks share MASTER_PORT / DVS_COMM_PORT and the run id?\n",
                        h.rank, h.nonce != nonce ? "job nonce differs" : "rank outside the world");
            }
            if (!kept) ::close(fd);                            // (anything else: not one of ours — no slot used)
        }
        ::close(ls);
        return true;
    }
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
    char pbuf[16]; snprintf(pbuf, sizeof pbuf, "%d", port);
    if (getaddrinfo(addr, pbuf, &hints, &res) != 0 || !res) { err = std::string("cannot resolve ") + addr; return false; }
    bool ok = false;
    while (!ok && now_s() < deadline) {
        int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd >= 0) {
            set_timeouts(fd, 5);
            if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
                Hello h{}; h.magic = kMagic; h.rank = (uint32_t)rank; h.nonce = nonce;
                Reply r{};
                const uint32_t ack = kMagic;
                uint8_t confirm = 0;                           // three-way: hello -> id -> ack -> confirm. Without the confirm rank 0 has not counted us
                                                               // (its receive of the ack failed): ask again instead of walking into ncclCommInitRank alone
                // (the confirmation is awaited with everything the deadline has left, not the 5 s step timeout: once the ack is out rank 0
                // may have counted us, and giving up on this connection then is the one failure the others cannot recover from)
                auto wait_long = [&] { const double rem = deadline - now_s(); set_timeouts(fd, rem > 1.0 ? (int)rem : 1); return true; };
                if (send_all(fd, &h, sizeof h) && recv_all(fd, &r, sizeof r) && r.magic == kMagic && send_all(fd, &ack, sizeof ack) && wait_long() &&
                    recv_all(fd, &confirm, 1) && confirm == 1) { memcpy(id128, r.id, 128); ok = true; }
            }
            if (ok && keep_fds) { (*keep_fds)[0] = fd; continue; }
            ::close(fd);
        }
        if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    freeaddrinfo(res);
    if (!ok) err = "no RCCL id from rank 0 within the deadline (is it running, and MASTER_ADDR / MASTER_PORT / DVS_COMM_PORT the same on every rank?)";
    return ok;
}

// Where the bootstrap listens. The launcher's own rendezvous store (torch.distributed.run) already listens on MASTER_PORT, so the id
// travels on a dedicated port: DVS_COMM_PORT if set, otherwise MASTER_PORT + 1789 (wrapped into the unprivileged range).
int bootstrap_port(int master_port) {
    if (const char* p = getenv("DVS_COMM_PORT")) { const int v = atoi(p); if (v > 0 && v < 65536) return v; }
    int port = master_port + 1789;
    if (port >= 65536) port = 1024 + (port - 65536) % (65536 - 1024);
    return port;
}
// Ranks of one job agree on a nonce without talking: the rendezvous port plus the launcher's run id when there is one. The ADDRESS is
// deliberately not part of it: rank 0 only listens, and may know the rendezvous host under another name (NULL, localhost, 127.0.0.1)
// than the peers that connect to it.
uint64_t job_nonce(const char* /*addr*/, int master_port) {
    std::string key = ":" + std::to_string(master_port);
    for (const char* name : {"DVS_COMM_NONCE", "TORCHELASTIC_RUN_ID", "SLURM_JOB_ID"})
        if (const char* v = getenv(name)) { key += "|"; key += v; }
    uint64_t h = 1469598103934665603ull;                     // FNV-1a
    for (unsigned char ch : key) { h ^= ch; h *= 1099511628211ull; }
    return h;
}
double bootstrap_timeout() {
    if (const char* p = getenv("DVS_COMM_TIMEOUT_S")) { const double v = atof(p); if (v > 0) return v; }
    return 180.0;
}
}  // namespace

struct dvs_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    // TEST-ONLY host-staged backend (DVS_COMM_BACKEND=tcp): a star over the bootstrap sockets. Rank 0 holds one socket per peer (index =
    // rank), a peer its socket to rank 0 at [0]. Every collective is synchronous: wait for `stream`, copy to the host, exchange, reduce on
    // rank 0 IN RANK ORDER (so every rank receives the same bits), copy back. It exists so that two ranks of the product can share the
    // one GPU of a test box (RCCL cannot: both ranks would be the same device) — never a production path, and it says so when created.
    bool tcp = false;
    std::vector<int> fds;
    std::vector<char> h_send, h_recv;
    double op_timeout = 180.0;
};

namespace {
int tcp_fail(const char* who, const char* what) {
    dvs_set_last_error((std::string(who) + " (tcp backend): " + what).c_str());
    return DVS_ERR_STATE;
}
// `bytes` of device memory at dev -> rank 0 combines the ranks' buffers with `combine(acc, incoming)` in rank order -> every rank gets
// out_bytes back (reduce: out_bytes = bytes; gather: out_bytes = world * bytes, slot r = rank r's buffer).
template <class Combine>
int tcp_collective(dvs_comm* c, hipStream_t st, const void* dev_in, void* dev_out, size_t bytes, bool gather, Combine combine, const char* who) {
    const size_t out_bytes = gather ? bytes * (size_t)c->world : bytes;
    const bool host = c->device < 0;                        // device -1: the buffers are HOST memory (CPU tests of the collectives' logic)
    if (!host && hipStreamSynchronize(st) != hipSuccess) return tcp_fail(who, "hipStreamSynchronize failed");
    c->h_send.resize(bytes); c->h_recv.resize(out_bytes);
    if (host) memcpy(c->h_send.data(), dev_in, bytes);
    else if (hipMemcpy(c->h_send.data(), dev_in, bytes, hipMemcpyDeviceToHost) != hipSuccess) return tcp_fail(who, "device -> host copy failed");
    if (c->rank == 0) {
        std::vector<char> in(bytes);
        if (gather) memcpy(c->h_recv.data(), c->h_send.data(), bytes); else memcpy(c->h_recv.data(), c->h_send.data(), bytes);
        for (int r = 1; r < c->world; ++r) {
            if (!recv_all(c->fds[(size_t)r], in.data(), bytes)) return tcp_fail(who, "a peer closed its socket or timed out");
            if (gather) memcpy(c->h_recv.data() + (size_t)r * bytes, in.data(), bytes);
            else combine(c->h_recv.data(), in.data(), bytes);
        }
        for (int r = 1; r < c->world; ++r)
            if (!send_all(c->fds[(size_t)r], c->h_recv.data(), out_bytes)) return tcp_fail(who, "send to a peer failed");
    } else {
        if (!send_all(c->fds[0], c->h_send.data(), bytes) || !recv_all(c->fds[0], c->h_recv.data(), out_bytes))
            return tcp_fail(who, "rank 0 closed its socket or timed out");
    }
    if (host) memcpy(dev_out, c->h_recv.data(), out_bytes);
    else if (hipMemcpy(dev_out, c->h_recv.data(), out_bytes, hipMemcpyHostToDevice) != hipSuccess) return tcp_fail(who, "host -> device copy failed");
    return DVS_OK;
}
void sum_f32(char* acc, const char* in, size_t bytes) {
    float* a = (float*)acc; const float* b = (const float*)in;
    for (size_t i = 0; i < bytes / 4; ++i) a[i] += b[i];
}
void max_i32(char* acc, const char* in, size_t bytes) {
    int32_t* a = (int32_t*)acc; const int32_t* b = (const int32_t*)in;
    for (size_t i = 0; i < bytes / 4; ++i) a[i] = a[i] > b[i] ? a[i] : b[i];
}
void keep(char*, const char*, size_t) {}
bool want_tcp_backend() { const char* b = getenv("DVS_COMM_BACKEND"); return b && std::string(b) == "tcp"; }
}  // namespace

#define RCCLCHECK(expr)                                                                                            \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != 0) { std::string m_ = std::string(#expr ": ") + g_rccl.GetErrorString(r_); dvs_set_last_error(m_.c_str()); return DVS_ERR_HIP; } \
    } while (0)

extern "C" {

// rank / world < 1 and a null address / non-positive port each fall back to the launcher's environment, independently of one another
static void resolve_rendezvous(int& rank, int& world, const char*& master_addr, int& master_port) {
    if (world < 1) {
        const char* r = getenv("RANK"); const char* w = getenv("WORLD_SIZE");
        rank = r ? atoi(r) : 0; world = w ? atoi(w) : 1;
    }
    if (!master_addr || !*master_addr) master_addr = getenv("MASTER_ADDR");
    if (!master_addr || !*master_addr) master_addr = "127.0.0.1";
    if (master_port <= 0) { const char* p = getenv("MASTER_PORT"); master_port = p ? atoi(p) : 0; }
    if (master_port <= 0) master_port = 29500;
}

int dvs_comm_bootstrap(int rank, int world, const char* master_addr, int master_port, void* id128) {
    std::string err;
    resolve_rendezvous(rank, world, master_addr, master_port);
    if (!id128 || rank < 0 || rank >= world) { dvs_set_last_error("dvs_comm_bootstrap: bad argument"); return DVS_ERR_INVALID; }
    if (!exchange_id(id128, rank, world, master_addr, bootstrap_port(master_port), job_nonce(master_addr, master_port), bootstrap_timeout(), err)) {
        dvs_set_last_error(("dvs_comm_bootstrap: " + err).c_str());
        return DVS_ERR_STATE;
    }
    return DVS_OK;
}

dvs_comm* dvs_comm_create(int device, int rank, int world, const char* master_addr, int master_port) {
    std::string err;
    resolve_rendezvous(rank, world, master_addr, master_port);
    if (rank < 0 || rank >= world) { dvs_set_last_error("dvs_comm_create: rank outside [0, world)"); return nullptr; }
    if (want_tcp_backend()) {
        // (device -1 with the test backend: no HIP at all, the collectives then take HOST buffers — tests/test_comm_bootstrap.py)
        if (device >= 0 && hipSetDevice(device) != hipSuccess) { dvs_set_last_error("dvs_comm_create: hipSetDevice failed"); return nullptr; }
        dvs_comm* c = new dvs_comm();
        c->device = device; c->rank = rank; c->world = world; c->tcp = true; c->op_timeout = bootstrap_timeout();
        char id[128] = {0};
        if (!exchange_id(id, rank, world, master_addr, bootstrap_port(master_port), job_nonce(master_addr, master_port), bootstrap_timeout(), err, &c->fds)) {
            dvs_set_last_error(("dvs_comm_create (tcp backend): " + err).c_str());
            dvs_comm_destroy(c);
            return nullptr;
        }
        for (int fd : c->fds) if (fd >= 0) { set_timeouts(fd, (int)c->op_timeout); int one = 1; setsockopt(fd, IPPROTO_TCP, 1 /*TCP_NODELAY*/, &one, sizeof one); }
        fprintf(stderr, "[dvs_comm] rank %d of %d: DVS_COMM_BACKEND=tcp — host-staged TEST backend (no RCCL, no xGMI); never use it for measurements\n", rank, world);
        return c;
    }
    if (!g_rccl.load(err)) { dvs_set_last_error(("dvs_comm_create: " + err).c_str()); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { dvs_set_last_error("dvs_comm_create: hipSetDevice failed"); return nullptr; }
    ncclUniqueId id;
    memset(&id, 0, sizeof id);
    if (rank == 0) {
        ncclResult_t r = g_rccl.GetUniqueId(&id);
        if (r != 0) { dvs_set_last_error((std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r)).c_str()); return nullptr; }
    }
    if (!exchange_id(&id, rank, world, master_addr, bootstrap_port(master_port), job_nonce(master_addr, master_port), bootstrap_timeout(), err)) {
        dvs_set_last_error(("dvs_comm_create: " + err).c_str());
        return nullptr;
    }
    dvs_comm* c = new dvs_comm();
    c->device = device; c->rank = rank; c->world = world;
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) { dvs_set_last_error((std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r)).c_str()); delete c; return nullptr; }
    return c;
}

void dvs_comm_destroy(dvs_comm* c) {
    if (!c) return;
    for (int fd : c->fds) if (fd >= 0) ::close(fd);
    if (c->comm) { (void)hipSetDevice(c->device); (void)g_rccl.CommDestroy(c->comm); }
    delete c;
}
int dvs_comm_rank(const dvs_comm* c) { return c ? c->rank : 0; }
int dvs_comm_world(const dvs_comm* c) { return c ? c->world : 1; }
int dvs_comm_backend_ranks(const dvs_comm* c) {
    if (!c) return 0;
    if (c->tcp) { int n = 1; for (int fd : c->fds) n += fd >= 0 ? 1 : 0; return c->rank == 0 ? n : c->world; }      // (rank 0 counts its open peer sockets)
    int n = 0;
    if (c->comm && g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) == 0) return n;
    return -1;
}
const char* dvs_comm_backend_name(const dvs_comm* c) { return !c ? "none" : c->tcp ? "tcp (host-staged TEST backend)" : "rccl"; }

int dvs_comm_all_reduce_sum_f32(dvs_comm* c, void* stream, float* buf, size_t count) {
    if (!c || (count && !buf)) { dvs_set_last_error("dvs_comm_all_reduce_sum_f32: null argument"); return DVS_ERR_INVALID; }
    if (!count) return DVS_OK;
    if (c->tcp) return c->world == 1 ? DVS_OK : tcp_collective(c, (hipStream_t)stream, buf, buf, count * 4, false, sum_f32, "dvs_comm_all_reduce_sum_f32");
    RCCLCHECK(g_rccl.AllReduce(buf, buf, count, rcclFloat32, rcclSum, c->comm, (hipStream_t)stream));
    return DVS_OK;
}
int dvs_comm_all_reduce_max_i32(dvs_comm* c, void* stream, int32_t* buf, size_t count) {
    if (!c || (count && !buf)) { dvs_set_last_error("dvs_comm_all_reduce_max_i32: null argument"); return DVS_ERR_INVALID; }
    if (!count) return DVS_OK;
    if (c->tcp) return c->world == 1 ? DVS_OK : tcp_collective(c, (hipStream_t)stream, buf, buf, count * 4, false, max_i32, "dvs_comm_all_reduce_max_i32");
    RCCLCHECK(g_rccl.AllReduce(buf, buf, count, rcclInt32, rcclMax, c->comm, (hipStream_t)stream));
    return DVS_OK;
}
int dvs_comm_reduce_scatter_sum_f32(dvs_comm* c, void* stream, const float* send, float* recv, size_t recv_count) {
    if (!c || (recv_count && (!send || !recv))) { dvs_set_last_error("dvs_comm_reduce_scatter_sum_f32: null argument"); return DVS_ERR_INVALID; }
    if (!recv_count) return DVS_OK;
    if (c->tcp) {          // all-reduce on the host, then this rank's slice
        const size_t total = recv_count * (size_t)c->world;
        if (c->device < 0) {                             // host buffers
            std::vector<float> tmp(send, send + total);
            int r = c->world > 1 ? tcp_collective(c, nullptr, tmp.data(), tmp.data(), total * 4, false, sum_f32, "dvs_comm_reduce_scatter_sum_f32") : DVS_OK;
            if (r == DVS_OK) memcpy(recv, tmp.data() + recv_count * (size_t)c->rank, recv_count * 4);
            return r;
        }
        float* tmp = nullptr;
        if (hipMalloc((void**)&tmp, total * 4) != hipSuccess) return tcp_fail("dvs_comm_reduce_scatter_sum_f32", "hipMalloc failed");
        int r = hipMemcpyAsync(tmp, send, total * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? DVS_OK : DVS_ERR_HIP;
        if (r == DVS_OK && c->world > 1) r = tcp_collective(c, (hipStream_t)stream, tmp, tmp, total * 4, false, sum_f32, "dvs_comm_reduce_scatter_sum_f32");
        if (r == DVS_OK && hipMemcpy(recv, tmp + recv_count * (size_t)c->rank, recv_count * 4, hipMemcpyDeviceToDevice) != hipSuccess) r = DVS_ERR_HIP;
        (void)hipFree(tmp);
        return r;
    }
    RCCLCHECK(g_rccl.ReduceScatter(send, recv, recv_count, rcclFloat32, rcclSum, c->comm, (hipStream_t)stream));
    return DVS_OK;
}
int dvs_comm_all_gather_f32(dvs_comm* c, void* stream, const float* send, float* recv, size_t send_count) {
    if (!c || (send_count && (!send || !recv))) { dvs_set_last_error("dvs_comm_all_gather_f32: null argument"); return DVS_ERR_INVALID; }
    if (!send_count) return DVS_OK;
    if (c->tcp) {
        if (c->world == 1) {
            if (c->device < 0) { memcpy(recv, send, send_count * 4); return DVS_OK; }
            return hipMemcpyAsync(recv, send, send_count * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess ? DVS_OK : DVS_ERR_HIP;
        }
        return tcp_collective(c, (hipStream_t)stream, send, recv, send_count * 4, true, keep, "dvs_comm_all_gather_f32");
    }
    RCCLCHECK(g_rccl.AllGather(send, recv, send_count, rcclFloat32, c->comm, (hipStream_t)stream));
    return DVS_OK;
}
int dvs_comm_group_start(dvs_comm* c) {
    if (!c) { dvs_set_last_error("dvs_comm_group_start: null communicator"); return DVS_ERR_INVALID; }
    if (c->tcp) return DVS_OK;         // (each collective of the group runs on its own, in call order on every rank)
    RCCLCHECK(g_rccl.GroupStart());
    return DVS_OK;
}
int dvs_comm_group_end(dvs_comm* c) {
    if (!c) { dvs_set_last_error("dvs_comm_group_end: null communicator"); return DVS_ERR_INVALID; }
    if (c->tcp) return DVS_OK;
    RCCLCHECK(g_rccl.GroupEnd());
    return DVS_OK;
}
int dvs_comm_broadcast(dvs_comm* c, void* stream, void* buf, size_t bytes, int root) {
    if (!c || (bytes && !buf) || root < 0 || root >= c->world) { dvs_set_last_error("dvs_comm_broadcast: bad argument"); return DVS_ERR_INVALID; }
    if (!bytes) return DVS_OK;
    if (c->tcp) {          // gather everything on rank 0, keep the root's slot (test backend: simplicity over bytes)
        if (c->world == 1) return DVS_OK;
        if (c->device < 0) {                             // host buffers
            std::vector<char> all(bytes * (size_t)c->world);
            int r = tcp_collective(c, nullptr, buf, all.data(), bytes, true, keep, "dvs_comm_broadcast");
            if (r == DVS_OK) memcpy(buf, all.data() + bytes * (size_t)root, bytes);
            return r;
        }
        char* tmp = nullptr;
        if (hipMalloc((void**)&tmp, bytes * (size_t)c->world) != hipSuccess) return tcp_fail("dvs_comm_broadcast", "hipMalloc failed");
        int r = tcp_collective(c, (hipStream_t)stream, buf, tmp, bytes, true, keep, "dvs_comm_broadcast");
        if (r == DVS_OK && hipMemcpy(buf, tmp + bytes * (size_t)root, bytes, hipMemcpyDeviceToDevice) != hipSuccess) r = DVS_ERR_HIP;
        (void)hipFree(tmp);
        return r;
    }
    RCCLCHECK(g_rccl.Broadcast(buf, buf, bytes, rcclUint8, root, c->comm, (hipStream_t)stream));
    return DVS_OK;
}

}  // extern "C"

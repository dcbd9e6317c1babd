// lpx_comm.cpp -- X1, the incumbent exchange of the sharded searches, as a native RCCL communicator inside liblpx.so.
//
// The reference keeps its incumbent in a field of the one process it runs in (`BestObjective`, Models/Branch&Bound.cs:182,191;
// `_bestValue`, Models/BranchAndBoundKnapsack.cs:124,157-160).  With the node queue sharded over the GPUs of a node, that field
// becomes ONE all-reduce(MAX) per level / round over xGMI.  A C# host has no torch and no MPI: the engine
// itself owns the communicator.  One process per GPU:
//
//     lpx_init(local_rank);
//     rank 0:  lpx_comm_unique_id(id)  ->  the host ships the 128 bytes to the other ranks (file, pipe, socket ...)
//     all:     lpx_comm_init(rank, world, id)            or, with no side channel at hand,
//     all:     lpx_comm_init_tcp(rank, world, "127.0.0.1", port)      (rank 0 serves the id on that port)
//     lpx_solve(..., opts.rank = rank, opts.world = world, opts.allreduce_max = NULL)   -> the searches use the communicator
//     lpx_comm_destroy();
//
// RCCL is bound at run time (dlopen), not at link time: liblpx.so stays loadable where no librccl exists, a process that already
// holds an RCCL (PyTorch ships its own librccl.so with the same SONAME) keeps exactly one copy, and single-GPU hosts never pay
// for the 570 MB library.  The vector travels host -> pinned staging -> device buffer (H2D on the communicator's stream),
// ncclAllReduce(ncclMax, ncclDouble) in place, D2H, one stream wait: a level costs one small copy each way.
#include "lpx_internal.h"
#include "../../include/lpx_test.h"

#include <rccl/rccl.h>

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

namespace lpx {

extern int g_device;

namespace {

struct Rccl {
    void* so = nullptr;
    std::string path;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct Comm {
    std::mutex mu;
    Rccl api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;             // world == 0: no communicator
    hipStream_t stream = nullptr;
    double* dev = nullptr; double* pin = nullptr; size_t cap = 0;      // doubles
    int64_t calls = 0; double ms = 0.0;
};
Comm g_comm;       // never destroyed at exit: the HIP runtime may be gone by then (lpx_comm_destroy is the host's job)

int load_rccl(Rccl& r)
{
    if (r.so) return 0;
    // LPX_RCCL_LIB: an explicit path.  Otherwise the SONAME first: a copy the process already holds (PyTorch's) is found by
    // name and shared; then the ROCm install.
    const char* env = std::getenv("LPX_RCCL_LIB");
    const char* cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    std::string tried;
    for (const char* c : cands) {
        if (!c || !c[0]) continue;
        void* so = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (so) { r.so = so; r.path = c; break; }
        tried += std::string(tried.empty() ? "" : "; ") + c + ": " + (dlerror() ? "not loadable" : "?");
    }
    if (!r.so) { set_error("lpx_comm: no RCCL library found (" + tried + "); set LPX_RCCL_LIB"); return LPX_EDEVICE; }
    auto sym = [&](const char* n) { return dlsym(r.so, n); };
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString) {
        set_error("lpx_comm: " + r.path + " lacks the ncclGetUniqueId / ncclCommInitRank / ncclAllReduce entry points");
        dlclose(r.so); r = Rccl{};
        return LPX_EDEVICE;
    }
    return 0;
}

int nccl_fail(const Rccl& r, const char* what, ncclResult_t e)
{
    set_error(std::string("lpx_comm: ") + what + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error") +
              " (RCCL code " + std::to_string((int)e) + ")");
    return LPX_EDEVICE;
}

int ensure_buffers(Comm& c, size_t doubles)
{
    if (doubles <= c.cap) return 0;
    size_t cap = c.cap ? c.cap : 64;
    while (cap < doubles) cap *= 2;
    if (c.dev) { hipFree(c.dev); c.dev = nullptr; }
    if (c.pin) { hipHostFree(c.pin); c.pin = nullptr; }
    c.cap = 0;
    LPX_HIP_TRY(hipMalloc((void**)&c.dev, cap * sizeof(double)));
    LPX_HIP_TRY(hipHostMalloc((void**)&c.pin, cap * sizeof(double), hipHostMallocDefault));
    c.cap = cap;
    return 0;
}

void release(Comm& c, bool abort)
{
    if (c.comm) {
        if (c.stream) hipStreamSynchronize(c.stream);
        if (abort && c.api.CommAbort) c.api.CommAbort(c.comm); else c.api.CommDestroy(c.comm);
        c.comm = nullptr;
    }
    if (c.dev) { hipFree(c.dev); c.dev = nullptr; }
    if (c.pin) { hipHostFree(c.pin); c.pin = nullptr; }
    c.cap = 0; c.world = 0; c.rank = 0;
    // the stream stays: hipStreamCreate costs milliseconds on this stack and a later communicator reuses it
}

// ---- the 128-byte id over one TCP connection per rank (for hosts without a side channel) --------------------------------
bool send_all(int fd, const void* p, size_t n)
{
    const char* b = (const char*)p;
    while (n) { ssize_t k = ::send(fd, b, n, MSG_NOSIGNAL); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}
bool recv_all(int fd, void* p, size_t n)
{
    char* b = (char*)p;
    while (n) { ssize_t k = ::recv(fd, b, n, 0); if (k <= 0) return false; b += k; n -= (size_t)k; }
    return true;
}
constexpr uint32_t kMagic = 0x4c505831u;     // "LPX1"

int serve_id(const char* host, int port, int world, const uint8_t* id, int timeout_s)
{
    int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) { set_error("lpx_comm_init_tcp: socket() failed"); return LPX_EDEVICE; }
    int one = 1; setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
    if (!host || !host[0] || inet_pton(AF_INET, host, &a.sin_addr) != 1) a.sin_addr.s_addr = htonl(INADDR_ANY);
    if (::bind(ls, (sockaddr*)&a, sizeof(a)) != 0 || ::listen(ls, world) != 0) {
        ::close(ls); set_error("lpx_comm_init_tcp: cannot listen on port " + std::to_string(port) + ": " + std::strerror(errno));
        return LPX_EDEVICE;
    }
    timeval tv{timeout_s, 0}; setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    std::vector<char> seen((size_t)world, 0);
    int left = world - 1;
    while (left > 0) {
        int fd = ::accept(ls, nullptr, nullptr);
        if (fd < 0) { ::close(ls); set_error("lpx_comm_init_tcp: " + std::to_string(left) + " rank(s) never asked for the id"); return LPX_EDEVICE; }
        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        uint32_t hello[3] = {0, 0, 0};                                    // magic, rank, world
        const bool ok = recv_all(fd, hello, sizeof(hello)) && hello[0] == kMagic && (int)hello[2] == world &&
                        hello[1] >= 1 && (int)hello[1] < world && !seen[hello[1]];
        if (ok && send_all(fd, id, NCCL_UNIQUE_ID_BYTES)) { seen[hello[1]] = 1; --left; }
        ::close(fd);                                                      // a stranger on the port is ignored
    }
    ::close(ls);
    return 0;
}

int fetch_id(const char* host, int port, int rank, int world, uint8_t* id, int timeout_s)
{
    addrinfo hints{}; hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
    addrinfo* res = nullptr;
    const std::string h = host && host[0] ? host : "127.0.0.1";
    if (getaddrinfo(h.c_str(), std::to_string(port).c_str(), &hints, &res) != 0 || !res) {
        set_error("lpx_comm_init_tcp: cannot resolve " + h); return LPX_EDEVICE;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int rc = LPX_EDEVICE;
    for (;;) {                                                            // rank 0 may not be listening yet
        int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) {
            timeval tv{timeout_s, 0}; setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
            const uint32_t hello[3] = {kMagic, (uint32_t)rank, (uint32_t)world};
            const bool ok = send_all(fd, hello, sizeof(hello)) && recv_all(fd, id, NCCL_UNIQUE_ID_BYTES);
            ::close(fd);
            if (ok) { rc = 0; break; }
        } else if (fd >= 0) ::close(fd);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s)) {
            set_error("lpx_comm_init_tcp: rank 0 did not serve the id at " + h + ":" + std::to_string(port)); break;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    freeaddrinfo(res);
    return rc;
}

}  // namespace

// used by the model level (lpx_model_api.cpp): the communicator stands in for lpx_solve_opts.allreduce_max when that is NULL
bool comm_active(int* rank, int* world)
{
    std::lock_guard<std::mutex> g(g_comm.mu);
    if (g_comm.world <= 0) return false;
    if (rank) *rank = g_comm.rank;
    if (world) *world = g_comm.world;
    return true;
}

}  // namespace lpx

using namespace lpx;

extern "C" {

// test-only (include/lpx_test.h): the TCP hand-over alone
int lpx_test_comm_exchange_id(int rank, int world, const char* host, int port, uint8_t* id)
{
    if (!id || world < 1 || rank < 0 || rank >= world) { set_error("lpx_test_comm_exchange_id: bad arguments"); return LPX_EINVAL; }
    if (world == 1) return 0;
    return rank == 0 ? serve_id(host, port, world, id, 20) : fetch_id(host, port, rank, world, id, 20);
}

int lpx_comm_unique_id(uint8_t* id)
{
    if (!id) { set_error("lpx_comm_unique_id: null argument"); return LPX_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> g(g_comm.mu);
    if ((rc = load_rccl(g_comm.api))) return rc;
    static_assert(LPX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "lpx.h and rccl.h disagree on the id size");
    ncclUniqueId u;
    ncclResult_t e = g_comm.api.GetUniqueId(&u);
    if (e != ncclSuccess) return nccl_fail(g_comm.api, "ncclGetUniqueId", e);
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

int lpx_comm_init(int rank, int world, const uint8_t* id)
{
    if (!id || world < 1 || rank < 0 || rank >= world) { set_error("lpx_comm_init: bad rank / world / id"); return LPX_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    std::lock_guard<std::mutex> g(g_comm.mu);
    if (g_comm.world > 0) { set_error("lpx_comm_init: a communicator exists already (lpx_comm_destroy first)"); return LPX_EINVAL; }
    if ((rc = load_rccl(g_comm.api))) return rc;
    LPX_HIP_TRY(hipSetDevice(g_device));                 // ncclCommInitRank binds the calling thread's current device
    if (!g_comm.stream) LPX_HIP_TRY(hipStreamCreateWithFlags(&g_comm.stream, hipStreamNonBlocking));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    ncclResult_t e = g_comm.api.CommInitRank(&comm, world, u, rank);
    if (e != ncclSuccess) return nccl_fail(g_comm.api, "ncclCommInitRank", e);
    g_comm.comm = comm; g_comm.rank = rank; g_comm.world = world; g_comm.calls = 0; g_comm.ms = 0.0;
    if ((rc = ensure_buffers(g_comm, 64))) { release(g_comm, true); return rc; }
    return 0;
}

int lpx_comm_init_tcp(int rank, int world, const char* host, int port)
{
    if (world < 1 || rank < 0 || rank >= world || port <= 0 || port > 65535) { set_error("lpx_comm_init_tcp: bad rank / world / port"); return LPX_EINVAL; }
    static const int timeout_s = [] { const char* e = std::getenv("LPX_COMM_TIMEOUT_S"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 120; }();
    uint8_t id[LPX_COMM_ID_BYTES];
    int rc;
    if (rank == 0) {
        if ((rc = lpx_comm_unique_id(id))) return rc;
        if (world > 1 && (rc = serve_id(host, port, world, id, timeout_s))) return rc;
    } else if ((rc = fetch_id(host, port, rank, world, id, timeout_s))) return rc;
    return lpx_comm_init(rank, world, id);
}

int lpx_comm_allreduce_max(double* vals, int count)
{
    if (count < 0 || (count > 0 && !vals)) { set_error("lpx_comm_allreduce_max: bad arguments"); return LPX_EINVAL; }
    if (count == 0) return 0;
    std::lock_guard<std::mutex> g(g_comm.mu);
    Comm& c = g_comm;
    if (c.world <= 0) { set_error("lpx_comm_allreduce_max: no communicator (lpx_comm_init first)"); return LPX_EINVAL; }
    const double t0 = now_ms();
    int rc = ensure_buffers(c, (size_t)count);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * (size_t)count;
    std::memcpy(c.pin, vals, bytes);
    LPX_HIP_TRY(hipMemcpyAsync(c.dev, c.pin, bytes, hipMemcpyHostToDevice, c.stream));
    ncclResult_t e = c.api.AllReduce(c.dev, c.dev, (size_t)count, ncclDouble, ncclMax, c.comm, c.stream);
    if (e != ncclSuccess) return nccl_fail(c.api, "ncclAllReduce", e);
    LPX_HIP_TRY(hipMemcpyAsync(c.pin, c.dev, bytes, hipMemcpyDeviceToHost, c.stream));
    LPX_HIP_TRY(hipStreamSynchronize(c.stream));
    std::memcpy(vals, c.pin, bytes);
    c.calls++; c.ms += now_ms() - t0;
    return 0;
}

int lpx_comm_info(int* rank, int* world, int64_t* allreduces, double* allreduce_ms, int* rccl_version)
{
    std::lock_guard<std::mutex> g(g_comm.mu);
    if (rank) *rank = g_comm.world > 0 ? g_comm.rank : -1;
    if (world) *world = g_comm.world;
    if (allreduces) *allreduces = g_comm.calls;
    if (allreduce_ms) *allreduce_ms = g_comm.ms;
    if (rccl_version) { int v = 0; if (g_comm.api.GetVersion) g_comm.api.GetVersion(&v); *rccl_version = v; }
    return 0;
}

int lpx_comm_destroy(void)
{
    std::lock_guard<std::mutex> g(g_comm.mu);
    release(g_comm, false);
    return 0;
}

}  // extern "C"

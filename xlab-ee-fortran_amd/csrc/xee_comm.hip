// xee_comm.hip -- the one collective of the path: an all-gather of the fixed-size 128-byte case records
// over RCCL (xGMI inside a node), behind the C-ABI so that ANY host -- the Fortran program, bench.py -- moves its scalar
// diagnostics the same way.  One process per GPU; ranks own consecutive blocks of the case list; fields are never
// gathered.  32 cases per rank x 8 ranks = 32 KiB: latency-bound, nothing to bucket or overlap.
//
// librccl is loaded on first use (dlopen) -- libxee itself does not link against it, so single-GPU users never map the
// 570 MB library.  The rendezvous (who tells the other ranks the 128-byte unique id) belongs to the host: rank 0 calls
// xee_comm_unique_id and hands the bytes over by whatever channel it has (a file in the run directory for the Fortran
// host, a broadcast for bench.py).
#include "../../include/xee.h"
#include "xee_device.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r = [] {
        Rccl q;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            q.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (q.handle) break;
        }
        if (!q.handle) return q;
        q.GetUniqueId = reinterpret_cast<decltype(q.GetUniqueId)>(dlsym(q.handle, "ncclGetUniqueId"));
        q.CommInitRank = reinterpret_cast<decltype(q.CommInitRank)>(dlsym(q.handle, "ncclCommInitRank"));
        q.AllGather = reinterpret_cast<decltype(q.AllGather)>(dlsym(q.handle, "ncclAllGather"));
        q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.handle, "ncclCommDestroy"));
        q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.handle, "ncclGetErrorString"));
        q.ok = q.GetUniqueId && q.CommInitRank && q.AllGather && q.CommDestroy && q.GetErrorString;
        return q;
    }();
    return r;
}

int need_rccl()
{
    if (!rccl().ok) return xee::record_failure(XEE_ERR_HIP, "librccl could not be loaded (dlopen librccl.so.1)");
    return XEE_OK;
}

std::atomic<double> g_timeout_s{120.0};

int nccl_fail(const char* what, ncclResult_t r)
{
    const std::string msg = std::string(what) + ": " + rccl().GetErrorString(r);
    return xee::record_failure(XEE_ERR_HIP, msg.c_str());
}

}  // namespace

struct xee_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;
    void *send = nullptr, *recv = nullptr;      // device staging, grown on demand
    size_t cap_records = 0;
};

extern "C" {

static_assert(sizeof(ncclUniqueId) == XEE_COMM_ID_BYTES, "xee.h promises a 128-byte rendezvous id");
static_assert(sizeof(xee_case_record_t) == 128, "the gather moves 128-byte records");

int xee_comm_unique_id(char* id /* XEE_COMM_ID_BYTES */)
{
    if (!id) return xee::record_failure(XEE_ERR_INVALID_ARGUMENT, "null id");
    if (int s = need_rccl()) return s;
    ncclUniqueId u;
    const ncclResult_t r = rccl().GetUniqueId(&u);
    if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
    memcpy(id, &u, sizeof u);
    return XEE_OK;
}

int xee_comm_available(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return xee::record_failure(XEE_ERR_NO_DEVICE, "no HIP device visible");
    return need_rccl();
}

int xee_comm_library(char* path, int path_len)
{
    if (!path || path_len < 2) return xee::record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_comm_library: bad buffer");
    if (int s = need_rccl()) return s;
    Dl_info info;
    const char* name = (dladdr(reinterpret_cast<void*>(rccl().AllGather), &info) && info.dli_fname) ? info.dli_fname : "?";
    strncpy(path, name, (size_t)path_len - 1);
    path[path_len - 1] = 0;
    return XEE_OK;
}

int xee_comm_set_timeout(double seconds)
{
    g_timeout_s.store(seconds);
    return XEE_OK;
}

int xee_comm_create(xee_comm_t** out, const char* id, int rank, int world, int device)
{
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return xee::record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_comm_create: bad argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return xee::record_failure(XEE_ERR_NO_DEVICE, "no HIP device visible");
    if (int s = need_rccl()) return s;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return xee::record_failure(XEE_ERR_HIP, "hipGetDevice failed");
    if (hipSetDevice(device) != hipSuccess) return xee::record_failure(XEE_ERR_HIP, "hipSetDevice failed");
    xee_comm* c = new xee_comm();
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    // ncclCommInitRank blocks until every rank has arrived, for ever if one never does: it runs on a helper thread and
    // this one waits with a deadline (the job is shared so that a late helper never touches freed memory)
    struct Job { std::mutex m; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr; };
    auto job = std::make_shared<Job>();
    std::thread([job, u, world, rank, device] {
        ncclComm_t cm = nullptr;
        ncclResult_t r = ncclSystemError;
        if (hipSetDevice(device) == hipSuccess) r = rccl().CommInitRank(&cm, world, u, rank);
        std::lock_guard<std::mutex> lk(job->m);
        job->r = r; job->comm = cm; job->done = true;
        job->cv.notify_all();
    }).detach();
    {
        std::unique_lock<std::mutex> lk(job->m);
        const double to = g_timeout_s.load();
        if (to > 0) {
            if (!job->cv.wait_for(lk, std::chrono::duration<double>(to), [&] { return job->done; })) {
                delete c;
                return xee::record_failure(XEE_ERR_TIMEOUT, "ncclCommInitRank: not every rank arrived before the time-out (xee_comm_set_timeout)");
            }
        } else job->cv.wait(lk, [&] { return job->done; });
    }
    const ncclResult_t r = job->r;
    c->comm = job->comm;
    if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rccl().CommDestroy(c->comm); delete c; return xee::record_failure(XEE_ERR_HIP, "hipStreamCreate failed"); }
    *out = c;
    return XEE_OK;
}

int xee_comm_destroy(xee_comm_t* c)
{
    if (!c) return XEE_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->comm && rccl().ok) rccl().CommDestroy(c->comm);
    delete c;
    return XEE_OK;
}

// Every rank contributes `per_rank` records (pad a short tail with case_id = -1); every rank receives world*per_rank
// records ordered by rank.  Host buffers in, host buffers out: the records are produced on the host side of the
// ensemble API (xee_ensemble_records) and consumed there.
int xee_gather_scalars(xee_comm_t* c, const xee_case_record_t* local, int per_rank, xee_case_record_t* all)
{
    if (!c || !local || !all || per_rank < 1) return xee::record_failure(XEE_ERR_INVALID_ARGUMENT, "xee_gather_scalars: bad argument");
    if (hipSetDevice(c->device) != hipSuccess) return xee::record_failure(XEE_ERR_HIP, "hipSetDevice failed");
    const size_t bytes = sizeof(xee_case_record_t) * (size_t)per_rank;
    if (c->cap_records < (size_t)per_rank) {
        if (c->send) (void)hipFree(c->send);
        if (c->recv) (void)hipFree(c->recv);
        c->send = c->recv = nullptr;
        c->cap_records = 0;
        if (hipMalloc(&c->send, bytes) != hipSuccess || hipMalloc(&c->recv, bytes * c->world) != hipSuccess)
            return xee::record_failure(XEE_ERR_ALLOC, "hipMalloc gather staging");
        c->cap_records = (size_t)per_rank;
    }
    if (hipMemcpyAsync(c->send, local, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return xee::record_failure(XEE_ERR_HIP, "gather H2D failed");
    // 4-byte words: the record is 32 of them (sendcount = cases_per_gpu * 32)
    const ncclResult_t r = rccl().AllGather(c->send, c->recv, (size_t)per_rank * 32, ncclInt32, c->comm, c->stream);
    if (r != ncclSuccess) return nccl_fail("ncclAllGather", r);
    if (hipMemcpyAsync(all, c->recv, bytes * c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return xee::record_failure(XEE_ERR_HIP, "gather D2H failed");
    // (polled with a deadline: a rank that died before the collective would otherwise hang the others here)
    const double to = g_timeout_s.load();
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(c->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return xee::record_failure(XEE_ERR_HIP, "gather stream failed");
        if (to > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > to)
            return xee::record_failure(XEE_ERR_TIMEOUT, "ncclAllGather: not every rank arrived before the time-out (xee_comm_set_timeout)");
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return XEE_OK;
}

int xee_comm_rank(const xee_comm_t* c, int* rank, int* world)
{
    if (!c) return xee::record_failure(XEE_ERR_INVALID_ARGUMENT, "null comm");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return XEE_OK;
}

}  // extern "C"

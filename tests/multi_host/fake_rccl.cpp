// TEST DOUBLE for RCCL (TEST INFRASTRUCTURE; loaded only when MISTI_RCCL_LIB names it): the six entry points libmisti_hip.so binds
// (misti_multi.cpp: Rccl), implemented as plain copies between the ranks' buffers of ONE process, so that the gathered multi-device form
// (misti_multi_eval_batch_dev: persistent workers + NaN padding + status table + ONE grouped ncclAllGather over several communicators) can
// run with D > 1 contexts on the single GPU of a test box - or with no GPU at all.  Real RCCL refuses a device listed
// twice; this double says so itself through `misti_test_rccl_double`, which is what lets misti_multi.cpp accept {0, 0, 0} with it.
//
//   g++   -DFAKE_RCCL_HOST  buffers are host memory, copies are memcpy at ncclGroupEnd            (tests/test_multi_host_cpu.py, sanitizers)
//   hipcc (default)         buffers are device memory: copies are hipMemcpyAsync on the receiving rank's stream, ordered behind every
//                           sender's stream by events, and every sender waits for its readers         (tests/test_gpu_multi.py)
//
// Semantics kept from NCCL: all ranks of a communicator group take part in each collective; the k-th ncclAllGather issued on each rank
// between ncclGroupStart and ncclGroupEnd is one collective; in-place operation (sendbuff == recvbuff + rank * count) is allowed; on
// return from ncclGroupEnd the work is ENQUEUED on the ranks' streams, not finished.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#ifndef FAKE_RCCL_HOST
#include <hip/hip_runtime_api.h>
typedef hipStream_t stream_t;
#else
typedef void* stream_t;
#endif

namespace {

struct Group {
    int n = 0;
    int live = 0;
};
struct Comm {
    Group* g;
    int rank;
    int device;
};
struct Op {
    Comm* comm;
    const void* send;
    void* recv;
    size_t bytes;        // per rank
    stream_t stream;
};

std::mutex g_mu;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
int g_collectives = 0;   // completed collectives (a test reads it: the gather really went through the double)

size_t elem_size(int dtype) {
    switch (dtype) {
        case 0: case 1: return 1;     // int8 / uint8
        case 2: case 3: return 4;     // int32 / uint32
        case 4: case 5: return 8;     // int64 / uint64
        case 6: return 2;             // half
        case 7: return 4;             // float
        case 8: return 8;             // double
        default: return 0;
    }
}

int run(std::vector<Op>& ops) {
    // group the ops into collectives: the k-th op of every rank of a group
    while (!ops.empty()) {
        Group* g = ops.front().comm->g;
        std::vector<Op> coll((size_t)g->n);
        std::vector<bool> have((size_t)g->n, false);
        for (size_t i = 0; i < ops.size();) {
            Op& o = ops[i];
            if (o.comm->g == g && !have[(size_t)o.comm->rank]) {
                coll[(size_t)o.comm->rank] = o;
                have[(size_t)o.comm->rank] = true;
                ops.erase(ops.begin() + (long)i);
            } else ++i;
        }
        for (int r = 0; r < g->n; ++r) if (!have[(size_t)r]) return 5;              // ncclInvalidUsage: a rank is missing from the collective
        for (int r = 1; r < g->n; ++r) if (coll[(size_t)r].bytes != coll[0].bytes) return 4;    // ncclInvalidArgument
        const size_t bytes = coll[0].bytes;
#ifdef FAKE_RCCL_HOST
        // senders' blocks first into a scratch copy (in-place operation: a rank's send block lies inside its own table)
        std::vector<char> blocks((size_t)g->n * bytes);
        for (int q = 0; q < g->n; ++q) std::memcpy(blocks.data() + (size_t)q * bytes, coll[(size_t)q].send, bytes);
        for (int r = 0; r < g->n; ++r) std::memcpy(coll[(size_t)r].recv, blocks.data(), blocks.size());
#else
        std::vector<hipEvent_t> ready((size_t)g->n), done((size_t)g->n);
        for (int q = 0; q < g->n; ++q) {
            if (hipSetDevice(coll[(size_t)q].comm->device) != hipSuccess) return 1;
            if (hipEventCreateWithFlags(&ready[(size_t)q], hipEventDisableTiming) != hipSuccess) return 1;
            if (hipEventCreateWithFlags(&done[(size_t)q], hipEventDisableTiming) != hipSuccess) return 1;
            if (hipEventRecord(ready[(size_t)q], coll[(size_t)q].stream) != hipSuccess) return 1;      // q's block is produced by then
        }
        for (int r = 0; r < g->n; ++r) {
            const Op& R = coll[(size_t)r];
            if (hipSetDevice(R.comm->device) != hipSuccess) return 1;
            for (int q = 0; q < g->n; ++q) {
                const Op& Q = coll[(size_t)q];
                char* dst = static_cast<char*>(R.recv) + (size_t)q * bytes;
                if (dst == Q.send) continue;                                           // in place: a rank's own block is where it belongs
                if (q != r && hipStreamWaitEvent(R.stream, ready[(size_t)q], 0) != hipSuccess) return 1;
                if (hipMemcpyAsync(dst, Q.send, bytes, hipMemcpyDeviceToDevice, R.stream) != hipSuccess) return 1;
            }
            if (hipEventRecord(done[(size_t)r], R.stream) != hipSuccess) return 1;
        }
        // a collective is over for a rank when every rank has read its block: its stream goes on only then
        for (int q = 0; q < g->n; ++q)
            for (int r = 0; r < g->n; ++r)
                if (r != q && hipStreamWaitEvent(coll[(size_t)q].stream, done[(size_t)r], 0) != hipSuccess) return 1;
        for (int q = 0; q < g->n; ++q) { (void)hipEventDestroy(ready[(size_t)q]); (void)hipEventDestroy(done[(size_t)q]); }   // released when complete
#endif
        std::lock_guard<std::mutex> lk(g_mu);
        ++g_collectives;
    }
    return 0;
}

}  // namespace

extern "C" {

int misti_test_rccl_double = 1;                   // the marker misti_multi.cpp looks for
int misti_test_rccl_double_collectives(void) { std::lock_guard<std::mutex> lk(g_mu); return g_collectives; }

const char* ncclGetErrorString(int code) {
    switch (code) {
        case 0: return "no error";
        case 1: return "unhandled cuda error (RCCL double: a HIP call failed)";
        case 4: return "invalid argument (RCCL double)";
        case 5: return "invalid usage (RCCL double)";
        default: return "error (RCCL double)";
    }
}

int ncclCommInitAll(void** comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return 4;
    Group* g = new Group;
    g->n = ndev;
    g->live = ndev;
    for (int r = 0; r < ndev; ++r) comms[r] = new Comm{g, r, devlist ? devlist[r] : r};
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return 4;
    Group* g = c->g;
    delete c;
    std::lock_guard<std::mutex> lk(g_mu);
    if (--g->live == 0) delete g;
    return 0;
}

int ncclGroupStart(void) { ++t_depth; return 0; }

int ncclAllGather(const void* send, void* recv, size_t count, int dtype, void* comm, stream_t stream) {
    const size_t es = elem_size(dtype);
    if (!send || !recv || !comm || es == 0) return 4;
    t_ops.push_back(Op{static_cast<Comm*>(comm), send, recv, count * es, stream});
    if (t_depth == 0) { std::vector<Op> ops; ops.swap(t_ops); return run(ops); }
    return 0;
}

int ncclGroupEnd(void) {
    if (t_depth <= 0) return 5;
    if (--t_depth > 0) return 0;
    std::vector<Op> ops;
    ops.swap(t_ops);
    return run(ops);
}

}  // extern "C"

// Lanes (include/misti_hip.h, "lanes: many batches in flight on ONE device"): n engine contexts of one model on one device, each with
// its own non-blocking stream, behind one object - the overlapped rate of the headline benchmark for a caller of the C ABI (VERDICT r5
// item 5; until round 6 the pool lived in Python, misti_amd/lanes.py).  The reference's counterpart: one MigrationInference object per
// process, as many processes as cores (/root/reference/MiSTI.py:213-214 under `parallel -j 20`, README.md:110-115).
// Built on the public single-context entry points; no C++ exception leaves this file.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/misti_hip.h"

extern "C" int misti_set_error_(int code, const char* msg);     // misti_api.cpp: sets the calling thread's misti_last_error

namespace {

int faill(int code, const char* fmt, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return misti_set_error_(code, buf);
}

// One hardware queue per lane: the HIP runtime opens GPU_MAX_HW_QUEUES queues per process (default 4) and reads the variable when it
// initialises - at the process's first HIP call, which for a C caller comes after this library was loaded.  Never overrides a value
// the user has set; MISTI_KEEP_HW_QUEUES=1 leaves the environment alone.
// 22, not "as many as possible": the device runs 23 queues beside each other, and with a 24th ACTIVE one a burst of batches takes 10 ms
// instead of 2.7 (the scheduler starts time-slicing the queues; measured round 6, profiles/r06_hw_queue_cliff.txt: 22 lanes + the null
// stream 2.9 ms, 23 lanes 10.4 ms when the runtime may open 32).  Capped at 22 the runtime never opens the 24th: streams beyond the cap
// SHARE queues (24 lanes: 4.3 ms) - and what else a process creates (the null stream, RCCL's and PyTorch's own streams) cannot push a
// lane pool over the edge.
__attribute__((constructor)) void misti_lanes_queue_env() {
    const char* keep = std::getenv("MISTI_KEEP_HW_QUEUES");
    if (keep && keep[0] && keep[0] != '0') return;
    (void)setenv("GPU_MAX_HW_QUEUES", "22", 0);
}

}  // namespace

struct misti_lanes {
    int device = 0;
    std::vector<misti_ctx*> ctx;
    std::vector<hipStream_t> stream;      // each context's own stream (misti_get_stream at creation)
    std::vector<hipEvent_t> done;         // recorded behind every batch of the lane: "nothing in flight" is a query of it
    std::vector<char> used;               // the lane has had a batch (its event has been recorded)
    int next = 0;                         // round-robin position of MISTI_LANE_ANY
};

namespace {

template <class F>
int guarded(const char* where, F&& fn) noexcept {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return faill(MISTI_E_NOMEM, "%s: out of host memory", where);
    } catch (const std::exception& e) {
        return faill(MISTI_E_ARG, "%s: %s", where, e.what());
    } catch (...) {
        return faill(MISTI_E_ARG, "%s: unknown C++ exception", where);
    }
}

// 1 busy, 0 idle, < 0 error
int lane_busy(misti_lanes* L, int i) {
    if (!L->used[(size_t)i]) return 0;
    const hipError_t e = hipEventQuery(L->done[(size_t)i]);
    if (e == hipSuccess) return 0;
    (void)hipGetLastError();
    if (e == hipErrorNotReady) return 1;
    return faill(MISTI_E_HIP, "hipEventQuery on lane %d: %s", i, hipGetErrorString(e));
}

void release(misti_lanes* L) {
    for (misti_ctx* c : L->ctx) (void)misti_destroy(c);          // waits for the context's streams before releasing (misti_api.cpp)
    (void)hipSetDevice(L->device);
    for (hipEvent_t e : L->done) if (e) (void)hipEventDestroy(e);
    (void)hipGetLastError();
    delete L;
}

}  // namespace

extern "C" {

int misti_create_lanes(const misti_model_t* model, int device, int n_lanes, misti_lanes** out) {
    if (!out) return faill(MISTI_E_ARG, "out is NULL");
    *out = nullptr;
    if (n_lanes < 1 || n_lanes > MISTI_MAX_LANES) return faill(MISTI_E_LIMIT, "n_lanes %d out of range (1..%d)", n_lanes, MISTI_MAX_LANES);
    misti_lanes* L = nullptr;
    const int r = guarded("misti_create_lanes", [&]() -> int {
        L = new misti_lanes;
        L->device = device;
        for (int i = 0; i < n_lanes; ++i) {
            misti_ctx* c = nullptr;
            if (int q = misti_create(model, device, &c)) return q;            // the message is the context's
            L->ctx.push_back(c);
            void* s = nullptr;
            if (int q = misti_get_stream(c, &s)) return q;
            L->stream.push_back(static_cast<hipStream_t>(s));
            hipEvent_t e = nullptr;
            const hipError_t he = hipEventCreateWithFlags(&e, hipEventDisableTiming);
            if (he != hipSuccess) return faill(MISTI_E_HIP, "hipEventCreateWithFlags: %s", hipGetErrorString(he));
            L->done.push_back(e);
            L->used.push_back(0);
        }
        return 0;
    });
    if (r != 0) {
        if (L) { const std::string why = misti_last_error(); release(L); return faill(r, "%s", why.c_str()); }
        return r;
    }
    *out = L;
    return 0;
}

int misti_destroy_lanes(misti_lanes* L) {
    if (!L) return 0;
    release(L);
    return 0;
}

int misti_lanes_size(misti_lanes* L) { return L ? (int)L->ctx.size() : 0; }

int misti_lanes_context(misti_lanes* L, int i, misti_ctx** ctx) {
    if (!L || !ctx || i < 0 || i >= (int)L->ctx.size()) return faill(MISTI_E_ARG, "no such lane");
    *ctx = L->ctx[(size_t)i];
    return 0;
}

int misti_lanes_set_hints(misti_lanes* L, uint32_t hints) {
    if (!L) return faill(MISTI_E_ARG, "lanes is NULL");
    for (misti_ctx* c : L->ctx) if (int r = misti_set_hints(c, hints)) return r;
    return 0;
}

int misti_lanes_busy(misti_lanes* L, int lane) {
    if (!L || lane < 0 || lane >= (int)L->ctx.size()) return faill(MISTI_E_ARG, "no such lane");
    return lane_busy(L, lane);
}

int misti_lanes_eval_batch_dev(misti_lanes* L, int lane, int64_t n_cand, const double* d_split, const double* d_params, const int32_t* d_bounds,
                               int64_t n_rep, const double* d_jsfs, double* d_llk, double* d_jafs, double* d_lc, double* d_pr, int32_t* d_status, int* lane_used) {
    if (!L) return faill(MISTI_E_ARG, "lanes is NULL");
    const int n = (int)L->ctx.size();
    if (lane != MISTI_LANE_ANY && (lane < 0 || lane >= n)) return faill(MISTI_E_ARG, "lane %d out of range (0..%d, or MISTI_LANE_ANY)", lane, n - 1);
    if (lane == MISTI_LANE_ANY) {
        // a lane with nothing in flight, looked for from the round-robin position on; if every lane is busy, the round-robin one
        lane = L->next;
        for (int k = 0; k < n; ++k) {
            const int i = (L->next + k) % n;
            const int b = lane_busy(L, i);
            if (b < 0) return b;
            if (b == 0) { lane = i; break; }
        }
        L->next = (lane + 1) % n;
    }
    if (int r = misti_eval_batch_dev(L->ctx[(size_t)lane], n_cand, d_split, d_params, d_bounds, n_rep, d_jsfs, d_llk, d_jafs, d_lc, d_pr, d_status)) return r;
    const hipError_t e = hipEventRecord(L->done[(size_t)lane], L->stream[(size_t)lane]);
    if (e != hipSuccess) return faill(MISTI_E_HIP, "hipEventRecord on lane %d: %s", lane, hipGetErrorString(e));
    L->used[(size_t)lane] = 1;
    if (lane_used) *lane_used = lane;
    return 0;
}

// Waiting for a lane = waiting for the EVENT recorded behind its last batch when that batch was issued, then misti_sync.  The event is
// what makes the difference with many lanes: asked whether a stream is finished (hipStreamSynchronize, hipStreamQuery,
// hipDeviceSynchronize) the runtime first has to put a marker of its own behind the stream's last kernel and wait for it to come back -
// one round trip per hardware queue, one queue after the other, 0.02 ms each: twenty lanes report complete 0.25 - 0.36 ms after their
// last kernel ended.  The event's marker is already in the queue behind the batch; its signal fires when the batch ends, and the
// stream synchronisation after it finds the queue's last command complete (tools/ub/queue_burst.hip, wait modes 0 / 5 / 3: a burst
// of twenty batch-shaped streams WITHOUT events 2.09 / 2.03 ms, with them 1.87 ms; profiles/r06_queue_burst.txt).  Every batch of a
// context has had such an event behind it since round 3 (misti_api.cpp: last_ev), which is why the bench's device-wide fence never paid
// that price: the bench gains nothing from waiting here first (2.73 against 2.76 - 2.81 ms, run-to-run spread), a caller that
// waits lane by lane does.
static int wait_done(misti_lanes* L, size_t i) {
    if (!L->used[i]) return 0;
    const hipError_t e = hipEventSynchronize(L->done[i]);
    if (e != hipSuccess) return faill(MISTI_E_HIP, "hipEventSynchronize on lane %d: %s", (int)i, hipGetErrorString(e));
    return 0;
}

int misti_lanes_wait(misti_lanes* L, int lane) {
    if (!L || lane < 0 || lane >= (int)L->ctx.size()) return faill(MISTI_E_ARG, "no such lane");
    if (int r = wait_done(L, (size_t)lane)) return r;
    return misti_sync(L->ctx[(size_t)lane]);
}

int misti_lanes_sync(misti_lanes* L) {
    if (!L) return faill(MISTI_E_ARG, "lanes is NULL");
    (void)hipSetDevice(L->device);
    for (size_t i = 0; i < L->ctx.size(); ++i)
        if (int r = wait_done(L, i)) return r;
    for (size_t i = 0; i < L->ctx.size(); ++i)
        if (int r = misti_sync(L->ctx[i])) { const std::string why = misti_last_error(); return faill(r, "lane %d: %s", (int)i, why.c_str()); }
    return 0;
}

}  // extern "C"

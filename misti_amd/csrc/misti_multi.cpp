// Multi-device form of the C ABI (include/misti_hip.h, "several devices"): SURVEY 8b's "variant taking a device list for
// 1/2/4/8-GPU runs".  Replaces the reference's way of using more than one processor - `parallel -j 20 ... >> res.out`,
// /root/reference/README.md:110-115, and the bash loops of test.bs/*.sh: one OS process per grid point, results concatenated
// from stdout - for a caller that binds the C ABI from ONE process: one engine context and one PERSISTENT host thread per listed
// device (a device may be listed more than once), candidates dealt out by whole lambda-correction CHAINS, the costliest first.
//
//   misti_multi_eval_batch      host buffers in and out: every context gathers its rows straight from / scatters them straight
//                               into the caller's arrays (no exchange between devices, no collective: the results end in host
//                               memory, which every device's DMA engine reaches by itself);
//   misti_multi_eval_batch_dev  device buffers in and out, one shard per context, and the log-likelihoods GATHERED ON THE DEVICES
//                               by RCCL (ncclAllGather over xGMI on a single-process communicator over the device list): the
//                               in-library counterpart of the rank-per-GPU path of misti_amd/dist.py - the reference concatenates
//                               the stdout of its processes (README.md:113-114).
//
// No C++ exception leaves this file: every entry point and every worker body runs inside guarded() (include/misti_hip.h:20-21).
// Built on the public single-device entry points (+ the indexed host-buffer form misti_eval_batch_indexed_, misti_api.cpp).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/misti_hip.h"

extern "C" int misti_set_error_(int code, const char* msg);     // misti_api.cpp: sets the calling thread's misti_last_error
// misti_eval_batch on the rows idx[0 .. n) of the caller's arrays (gathered into the context's pinned block, results scattered
// back to rows idx[i]): no staging copy in between.  misti_api.cpp; not part of the public ABI.
extern "C" int misti_eval_batch_indexed_(misti_ctx* ctx, int64_t n, const int64_t* idx, const double* split, const double* params,
                                         const int32_t* band_bounds, int64_t n_rep, const double* jsfs,
                                         double* llk, double* jafs, double* lc, double* pr, int32_t* status);

namespace {

int failm(int code, const char* fmt, ...) {
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return misti_set_error_(code, buf);
}

// fn() -> int with every exception turned into an error code + message (never across the ABI, never out of a thread)
template <class F>
int guarded(const char* where, F&& fn) noexcept {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return failm(MISTI_E_NOMEM, "%s: out of host memory", where);
    } catch (const std::exception& e) {
        return failm(MISTI_E_ARG, "%s: %s", where, e.what());
    } catch (...) {
        return failm(MISTI_E_ARG, "%s: unknown C++ exception", where);
    }
}

// ---- RCCL, bound at first use ------------------------------------------------------------------------------------------------
// dlopen by soname: a process that already carries an RCCL (PyTorch-ROCm maps its own librccl.so.1) gets THAT one, never a second
// copy; a plain C caller gets the system's.  Only the handful of entry points the gather needs; types from the published ABI
// (rccl.h: ncclResult_t 0 = success; ncclInt32 = 2, ncclFloat64 = 8).
struct Rccl {
    void* handle = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    bool test_double = false;       // the loaded library exports `misti_test_rccl_double` (tests/multi_host/fake_rccl.cpp): copies inside one process
    std::string why;
};
constexpr int NCCL_INT32 = 2, NCCL_FLOAT64 = 8;

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // MISTI_RCCL_LIB=<path>: the RCCL build to bind instead of the process's / the system's (a site's own build - or the tests' double,
        // which lets the gathered form run with several contexts on one GPU); read once, at the first gathered call of the process
        const char* chosen = std::getenv("MISTI_RCCL_LIB");
        if (chosen && chosen[0]) {
            r.handle = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) { const char* e = dlerror(); r.why = std::string("MISTI_RCCL_LIB=") + chosen + " cannot be loaded: " + (e ? e : "?"); return; }
        } else {
            for (const char* name : {"librccl.so.1", "librccl.so"}) {
                r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.handle) break;
            }
            if (!r.handle) { const char* e = dlerror(); r.why = std::string("librccl.so.1 cannot be loaded: ") + (e ? e : "?"); return; }
        }
        r.test_double = dlsym(r.handle, "misti_test_rccl_double") != nullptr;
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(r.handle, n);
            if (!p && r.why.empty()) r.why = std::string("RCCL lacks ") + n;
            return p;
        };
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    });
    return &r;
}

}  // namespace

// ---- the object ----------------------------------------------------------------------------------------------------------------
struct misti_multi {
    std::vector<misti_ctx*> ctx;
    std::vector<int> device;
    int n_param = 0, n_band = 0, numT = 0;
    std::vector<int64_t> last_cands, last_chains;        // shard sizes of the last misti_multi_eval_batch
    std::vector<double> last_cost;                       // ... and the summed chain cost per context (what the dealing balances)
    std::vector<std::vector<int64_t>> shard;             // candidate rows per context (kept between calls: no allocation in steady state)
    std::vector<void*> comm;                             // RCCL communicators, one per context (created by the first _dev call)
    int throw_in_worker = -1;                            // test hook: context whose worker body throws (misti_multi_test_throw_in_worker_, below)
    // One call at a time per object: the dispatch state below (job, pending, generation, rc, msg), the shards and the last_* records
    // are per object.  Two caller threads on ONE misti_multi are serialised here (include/misti_hip.h says so); callers that want
    // concurrency use one object per thread, as with misti_ctx.
    std::mutex call_mu;

    // persistent workers: thread d serves context d
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<int(int)>* job = nullptr;
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rc;
    std::vector<std::string> msg;

    void worker(int d) noexcept {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(int)>* fn;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                fn = job;
            }
            const int r = guarded("worker thread", [&] { return (*fn)(d); });
            std::string why;
            if (r != 0) { try { why = misti_last_error(); } catch (...) {} }     // thread-local in the worker: carry it over
            {
                std::lock_guard<std::mutex> lk(mu);
                rc[d] = r;
                msg[d].swap(why);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void shutdown() noexcept {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_go.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
        th.clear();
    }
};

namespace {

// run fn(d) for every context at the same time, each on its own persistent thread; returns the first failure (its message
// becomes the caller's misti_last_error)
int on_every_device(misti_multi* m, const std::function<int(int)>& fn) {
    const int D = (int)m->ctx.size();
    if (D == 1) {
        const int r = guarded("misti_multi", [&] { return fn(0); });
        if (r != 0) { const std::string why = misti_last_error(); return failm(r, "device %d (context 0 of 1): %s", m->device[0], why.c_str()); }
        return 0;
    }
    {
        std::unique_lock<std::mutex> lk(m->mu);
        m->job = &fn;
        m->pending = D;
        std::fill(m->rc.begin(), m->rc.end(), 0);
        ++m->generation;
        m->cv_go.notify_all();
        m->cv_done.wait(lk, [&] { return m->pending == 0; });
        m->job = nullptr;
    }
    for (int d = 0; d < D; ++d)
        if (m->rc[d] != 0) return failm(m->rc[d], "device %d (context %d of %d): %s", m->device[d], d, D, m->msg[d].c_str());
    return 0;
}

// contiguous blocks of starts: block d = [lo[d], lo[d + 1])
std::vector<int64_t> blocks(int64_t n, int D) {
    std::vector<int64_t> lo(D + 1);
    for (int d = 0; d <= D; ++d) lo[d] = n * d / D;
    return lo;
}

// key of a candidate's chain: the bits of its parameter vector and of its band bounds (setup_kernel keys its table the same way)
struct Key {
    const char* p;
    size_t n;
    bool operator==(const Key& o) const { return n == o.n && std::memcmp(p, o.p, n) == 0; }
};
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = 0xcbf29ce484222325ull;
        for (size_t i = 0; i < k.n; ++i) { h ^= (unsigned char)k.p[i]; h *= 0x100000001b3ull; }
        return (size_t)h;
    }
};

// ---- whole chains per device, the costliest first ----------------------------------------------------------------------------
// Candidates with identical parameter vectors and band bounds share one lambda-correction chain, computed once per context that
// holds any of them, at its full latency however few members that context has: a chain stays on ONE device, all its split times
// with it.  What a chain costs is known on the host: its LENGTH - the corrected two-population intervals up to the largest split
// index of its members (the solver walks them one after the other; a post-split interval is a closed form) - plus a little per
// member (the spectrum kernel).  Longest-processing-time-first: chains by descending cost (ties: first appearance), each to the
// context with the least cost so far (ties: lowest index) - within 4/3 of the optimal makespan, and a grid whose chains all cost
// the same is dealt round-robin as before.  misti_amd/dist.py: chain_shards deals the ranks of the rank-per-GPU path the same way.
// Without parameters and bounds the batch is one chain: candidates are interleaved instead (every device repeats the chain; what
// is shared out is the spectrum kernel and the replicate epilogue).
constexpr double MEMBER_COST = 1.0 / 64.0;

void deal_chains(misti_multi* m, int64_t n_cand, const double* split, const double* params, const int32_t* band_bounds) {
    const int D = (int)m->ctx.size(), P = m->n_param, B = m->n_band;
    const bool with_bounds = band_bounds && B > 0;
    const size_t kp = (size_t)P * sizeof(double), kb = with_bounds ? (size_t)B * 2 * sizeof(int32_t) : 0, kl = kp + kb;
    for (int d = 0; d < D; ++d) { m->shard[d].clear(); m->last_cands[d] = m->last_chains[d] = 0; m->last_cost[d] = 0.0; }
    if (kl == 0) {
        for (int64_t c = 0; c < n_cand; ++c) m->shard[c % D].push_back(c);
        for (int d = 0; d < D; ++d) { m->last_chains[d] = m->shard[d].empty() ? 0 : 1; m->last_cands[d] = (int64_t)m->shard[d].size(); }
        return;
    }
    std::vector<char> keys((size_t)n_cand * kl);
    for (int64_t c = 0; c < n_cand; ++c) {
        char* k = keys.data() + (size_t)c * kl;
        if (kp) std::memcpy(k, params + (size_t)c * P, kp);
        if (kb) std::memcpy(k + kp, band_bounds + (size_t)c * B * 2, kb);
    }
    std::unordered_map<Key, int32_t, KeyHash> chain_of_key;
    chain_of_key.reserve((size_t)n_cand / 4 + 16);
    std::vector<int32_t> chain(n_cand);
    std::vector<double> len;                              // per chain: largest split index of its members
    std::vector<int64_t> members;
    for (int64_t c = 0; c < n_cand; ++c) {
        const Key k{keys.data() + (size_t)c * kl, kl};
        auto it = chain_of_key.find(k);
        int32_t ch;
        if (it == chain_of_key.end()) { ch = (int32_t)len.size(); chain_of_key.emplace(k, ch); len.push_back(0.0); members.push_back(0); }
        else ch = it->second;
        chain[c] = ch;
        members[ch] += 1;
        const double s = split[c];
        const double l = (s == s && s > 0.0) ? std::ceil(std::min(s, (double)m->numT)) : 0.0;     // a fractional split adds its shortened interval
        if (l > len[ch]) len[ch] = l;
    }
    const size_t nch = len.size();
    std::vector<double> cost(nch);
    for (size_t i = 0; i < nch; ++i) cost[i] = len[i] + MEMBER_COST * (double)members[i];
    std::vector<int32_t> order(nch);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cost[a] > cost[b]; });
    std::vector<int> owner(nch);
    for (int32_t ch : order) {
        int best = 0;
        for (int d = 1; d < D; ++d) if (m->last_cost[d] < m->last_cost[best]) best = d;
        owner[ch] = best;
        m->last_cost[best] += cost[ch];
        m->last_chains[best] += 1;
    }
    for (int64_t c = 0; c < n_cand; ++c) m->shard[owner[chain[c]]].push_back(c);
    for (int d = 0; d < D; ++d) m->last_cands[d] = (int64_t)m->shard[d].size();
}

}  // namespace

extern "C" {

int misti_create_multi(const misti_model_t* model, int n_dev, const int* devices, misti_multi** out) {
    if (!model || !out) return failm(MISTI_E_ARG, "model / out is NULL");
    if (n_dev < 1 || !devices) return failm(MISTI_E_ARG, "the device list is empty");
    const int have = misti_device_count();
    if (have < 1) return failm(MISTI_E_NODEV, "no usable HIP device");
    for (int d = 0; d < n_dev; ++d)
        if (devices[d] < 0 || devices[d] >= have) return failm(MISTI_E_ARG, "device %d is not one of the %d visible HIP devices", devices[d], have);
    misti_multi* m = nullptr;
    const int r = guarded("misti_create_multi", [&]() -> int {
        m = new misti_multi;
        m->n_param = model->n_param; m->n_band = model->n_band; m->numT = model->numT;
        for (int d = 0; d < n_dev; ++d) {
            misti_ctx* c = nullptr;
            const int q = misti_create(model, devices[d], &c);
            if (q != 0) { const std::string why = misti_last_error(); return failm(q, "device %d: %s", devices[d], why.c_str()); }
            m->ctx.push_back(c);
            m->device.push_back(devices[d]);
        }
        m->last_cands.assign(n_dev, 0);
        m->last_chains.assign(n_dev, 0);
        m->last_cost.assign(n_dev, 0.0);
        m->shard.resize(n_dev);
        m->rc.assign(n_dev, 0);
        m->msg.resize(n_dev);
        if (n_dev > 1) {
            m->th.reserve(n_dev);
            for (int d = 0; d < n_dev; ++d) m->th.emplace_back([m, d] { m->worker(d); });     // a failure here joins the threads already started (below)
        }
        return 0;
    });
    if (r != 0) {
        if (m) {
            const std::string why = misti_last_error();
            m->shutdown();
            for (misti_ctx* q : m->ctx) (void)misti_destroy(q);
            delete m;
            return failm(r, "%s", why.c_str());
        }
        return r;
    }
    *out = m;
    return 0;
}

int misti_destroy_multi(misti_multi* m) {
    if (!m) return 0;
    m->shutdown();
    int rc = 0;
    if (!m->comm.empty()) {                              // only a context that gathered ever loaded RCCL
        Rccl* R = rccl();
        for (void* c : m->comm) if (c && R->CommDestroy) (void)R->CommDestroy(c);
    }
    for (misti_ctx* c : m->ctx) { const int r = misti_destroy(c); if (r != 0 && rc == 0) rc = r; }
    delete m;
    return rc;
}

// TEST HOOK, not part of the public ABI (no declaration in include/misti_hip.h, trailing underscore like misti_eval_batch_indexed_): from
// now on the worker body of context `d` throws a C++ exception (d < 0: never) - tests/test_gpu_multi.py checks that such a call fails
// with an error code and message while the process, the workers and later calls live on.  Nothing in the library reads the environment
// for it any more (a set variable made every multi call of a production process fail).
int misti_multi_test_throw_in_worker_(misti_multi* m, int d) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    m->throw_in_worker = d;
    return 0;
}

int misti_multi_size(misti_multi* m) { return m ? (int)m->ctx.size() : 0; }

int misti_multi_context(misti_multi* m, int i, misti_ctx** ctx, int* device) {
    if (!m || i < 0 || i >= (int)m->ctx.size()) return failm(MISTI_E_ARG, "no such context");
    if (ctx) *ctx = m->ctx[i];
    if (device) *device = m->device[i];
    return 0;
}

int misti_multi_last_shards(misti_multi* m, int64_t* n_cand, int64_t* n_chain) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    for (size_t d = 0; d < m->ctx.size(); ++d) {
        if (n_cand) n_cand[d] = m->last_cands[d];
        if (n_chain) n_chain[d] = m->last_chains[d];
    }
    return 0;
}

int misti_multi_last_cost(misti_multi* m, double* cost) {
    if (!m || !cost) return failm(MISTI_E_ARG, "multi context / cost is NULL");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    for (size_t d = 0; d < m->ctx.size(); ++d) cost[d] = m->last_cost[d];
    return 0;
}

int misti_multi_sync(misti_multi* m) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    for (size_t d = 0; d < m->ctx.size(); ++d) {
        const int r = misti_sync(m->ctx[d]);
        if (r != 0) { const std::string why = misti_last_error(); return failm(r, "device %d (context %d): %s", m->device[d], (int)d, why.c_str()); }
    }
    return 0;
}

int misti_multi_eval_batch(misti_multi* m, int64_t n_cand, const double* split, const double* params, const int32_t* band_bounds,
                           int64_t n_rep, const double* jsfs, double* llk, double* jafs, double* lc, double* pr, int32_t* status) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    if (n_cand < 0 || n_rep < 0) return failm(MISTI_E_ARG, "negative batch size");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    const int D = (int)m->ctx.size();
    for (int d = 0; d < D; ++d) { m->last_cands[d] = m->last_chains[d] = 0; m->last_cost[d] = 0.0; }
    if (n_cand == 0) return 0;
    if (!split) return failm(MISTI_E_ARG, "split_time is NULL");
    const int P = m->n_param;
    if (P > 0 && !params) return failm(MISTI_E_ARG, "params is NULL but the model has %d parameters", P);
    if (n_rep > 0 && (!jsfs || !llk)) return failm(MISTI_E_ARG, "jsfs / llk is NULL with n_rep > 0");
    if (D == 1) {
        m->last_cands[0] = n_cand;
        return misti_eval_batch(m->ctx[0], n_cand, split, params, band_bounds, n_rep, jsfs, llk, jafs, lc, pr, status);
    }
    return guarded("misti_multi_eval_batch", [&]() -> int {
        deal_chains(m, n_cand, split, params, band_bounds);
        const std::function<int(int)> fn = [&](int d) -> int {
            if (d == m->throw_in_worker) throw std::runtime_error("misti_multi_test_throw_in_worker_");
            const std::vector<int64_t>& idx = m->shard[d];
            if (idx.empty()) return 0;
            // rows idx[] of the caller's arrays: gathered into / scattered from the context's pinned block (disjoint rows: no two threads share one)
            return misti_eval_batch_indexed_(m->ctx[d], (int64_t)idx.size(), idx.data(), split, params, band_bounds, n_rep, jsfs, llk, jafs, lc, pr, status);
        };
        return on_every_device(m, fn);
    });
}

// Device-resident form: context i evaluates ITS shard (device pointers on device i, as misti_eval_batch_dev) and the
// log-likelihoods are gathered on the devices: ncclAllGather (RCCL over xGMI), in place, on each context's stream, behind its batch.
int misti_multi_eval_batch_dev(misti_multi* m, const int64_t* n_cand, int64_t rows_per_shard,
                               const double* const* d_split_time, const double* const* d_params, const int32_t* const* d_band_bounds,
                               int64_t n_rep, const double* const* d_jsfs, double* const* d_llk_all, int32_t* const* d_status_all) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    if (!n_cand || !d_split_time || !d_llk_all || !d_jsfs) return failm(MISTI_E_ARG, "n_cand / d_split_time / d_jsfs / d_llk_all is NULL");
    if (n_rep < 1) return failm(MISTI_E_ARG, "the gathered form needs at least one replicate");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    const int D = (int)m->ctx.size(), P = m->n_param;
    if (P > 0 && !d_params) return failm(MISTI_E_ARG, "d_params is NULL but the model has %d parameters", P);
    bool repeated = false;
    for (int d = 0; d < D; ++d) {
        if (n_cand[d] < 0 || n_cand[d] > rows_per_shard) return failm(MISTI_E_ARG, "shard %d: %lld candidates do not fit rows_per_shard = %lld", d, (long long)n_cand[d], (long long)rows_per_shard);
        if (!d_llk_all[d] || !d_jsfs[d] || (n_cand[d] > 0 && (!d_split_time[d] || (P > 0 && !d_params[d])))) return failm(MISTI_E_ARG, "shard %d: a device pointer is NULL", d);
        if (d_status_all && !d_status_all[d]) return failm(MISTI_E_ARG, "shard %d: d_status_all is NULL", d);
        for (int e = 0; e < d; ++e) if (m->device[e] == m->device[d]) repeated = true;
    }
    // an RCCL communicator holds every device once (ncclCommInitAll refuses a repeated entry); only the tests' double, which copies
    // inside the process, takes several contexts on one device
    if (repeated) {
        Rccl* R = rccl();
        if (!R->test_double)
            for (int d = 0; d < D; ++d) for (int e = 0; e < d; ++e)
                if (m->device[e] == m->device[d]) return failm(MISTI_E_ARG, "device %d is listed twice: an RCCL communicator holds every device once", m->device[d]);
    }
    if (rows_per_shard == 0) return 0;
    return guarded("misti_multi_eval_batch_dev", [&]() -> int {
        Rccl* R = rccl();
        if (!R->handle || !R->why.empty()) return failm(MISTI_E_NODEV, "%s", R->why.c_str());
        if (m->comm.empty()) {
            // one communicator over the device list, created once per multi context (ncclCommInitAll: all ranks in this process)
            std::vector<void*> comms(D, nullptr);
            const int q = R->CommInitAll(comms.data(), D, m->device.data());
            if (q != 0) return failm(MISTI_E_HIP, "ncclCommInitAll over %d devices: %s", D, R->GetErrorString(q));
            m->comm = comms;
        }
        const size_t blk = (size_t)rows_per_shard * (size_t)n_rep;
        // every context issues its batch at the same time (a launch sequence costs the host ~50 us per batch): context d writes its
        // rows straight into block d of ITS gathered table, the rows beyond its shard are NaN (all-ones bytes)
        const std::function<int(int)> fn = [&](int d) -> int {
            if (d == m->throw_in_worker) throw std::runtime_error("misti_multi_test_throw_in_worker_");
            void* sv = nullptr;
            if (int q = misti_get_stream(m->ctx[d], &sv)) return q;
            hipStream_t s = static_cast<hipStream_t>(sv);
            // the fills below go to the stream's own device whatever the thread's current one is; the current device is set all the same
            // (a worker thread starts on device 0) and, with one context - this IS the caller's thread - put back afterwards
            int before = -1;
            (void)hipGetDevice(&before);
            if (hipSetDevice(m->device[d]) != hipSuccess) return failm(MISTI_E_HIP, "hipSetDevice(%d) failed", m->device[d]);
            struct Restore { int dev; ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); } } restore{D == 1 ? before : -1};
            double* mine = d_llk_all[d] + (size_t)d * blk;
            const size_t used = (size_t)n_cand[d] * (size_t)n_rep;
            if (used < blk && hipMemsetAsync(mine + used, 0xFF, (blk - used) * sizeof(double), s) != hipSuccess) return failm(MISTI_E_HIP, "hipMemsetAsync failed");
            int32_t* st = d_status_all ? d_status_all[d] + (size_t)d * (size_t)rows_per_shard : nullptr;
            if (st && n_cand[d] < rows_per_shard &&
                hipMemsetAsync(st + n_cand[d], 0xFF, (size_t)(rows_per_shard - n_cand[d]) * sizeof(int32_t), s) != hipSuccess) return failm(MISTI_E_HIP, "hipMemsetAsync failed");
            if (n_cand[d] == 0) return 0;
            return misti_eval_batch_dev(m->ctx[d], n_cand[d], d_split_time[d], P > 0 ? d_params[d] : nullptr, d_band_bounds ? d_band_bounds[d] : nullptr,
                                        n_rep, d_jsfs[d], mine, nullptr, nullptr, nullptr, st);
        };
        if (int q = on_every_device(m, fn)) return q;
        // ONE grouped collective for all ranks of this process (a single thread drives every communicator: ncclGroupStart / End)
        int q = R->GroupStart();
        for (int d = 0; d < D && q == 0; ++d) {
            void* sv = nullptr;
            if (int e = misti_get_stream(m->ctx[d], &sv)) { (void)R->GroupEnd(); return e; }
            hipStream_t s = static_cast<hipStream_t>(sv);
            q = R->AllGather(d_llk_all[d] + (size_t)d * blk, d_llk_all[d], blk, NCCL_FLOAT64, m->comm[d], s);
            if (q == 0 && d_status_all)
                q = R->AllGather(d_status_all[d] + (size_t)d * (size_t)rows_per_shard, d_status_all[d], (size_t)rows_per_shard, NCCL_INT32, m->comm[d], s);
        }
        const int qe = R->GroupEnd();
        if (q == 0) q = qe;
        if (q != 0) return failm(MISTI_E_HIP, "ncclAllGather: %s", R->GetErrorString(q));
        for (int d = 0; d < D; ++d) m->last_cands[d] = n_cand[d];
        return 0;
    });
}

// Starts are independent searches: contiguous blocks of them per device, every device running misti_nm_solve on its block at the
// same time.  A start's trajectory does not depend on what else is in its batches, so the result equals the single-device call's.
int misti_multi_nm_solve(misti_multi* m, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                         double xatol, double fatol, int32_t maxiter, double* x, double* llh, int32_t* nit, int32_t* nfev, int32_t* status) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    if (n_start < 0) return failm(MISTI_E_ARG, "negative number of starts");
    if (n_start == 0) return 0;
    if (!starts || !jsfs_row || !x || !llh) return failm(MISTI_E_ARG, "starts / jsfs_row / x / llh is NULL");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    return guarded("misti_multi_nm_solve", [&]() -> int {
        const int D = (int)m->ctx.size(), N = m->n_param;
        const std::vector<int64_t> lo = blocks(n_start, D);
        const std::function<int(int)> fn = [&](int d) -> int {
            if (d == m->throw_in_worker) throw std::runtime_error("misti_multi_test_throw_in_worker_");
            const int64_t a = lo[d], n = lo[d + 1] - lo[d];
            if (n == 0) return 0;
            return misti_nm_solve(m->ctx[d], n, starts + a * N, split_time, jsfs_row, xatol, fatol, maxiter, x + a * N, llh + a,
                                  nit ? nit + a : nullptr, nfev ? nfev + a : nullptr, status ? status + a : nullptr);
        };
        return on_every_device(m, fn);
    });
}

int misti_multi_basinhopping(misti_multi* m, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                             int32_t niter, double T, double stepsize, int32_t interval, double target_accept_rate, double stepwise_factor,
                             double xatol, double fatol, int32_t nm_maxiter, int64_t nm_maxfev, const double* uniforms,
                             double* x, double* llh, int32_t* nfev, int32_t* failures, int32_t* accepted) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    if (n_start < 0 || niter < 0) return failm(MISTI_E_ARG, "negative number of starts / hops");
    if (n_start == 0) return 0;
    if (!starts || !jsfs_row || !x || !llh || (niter > 0 && !uniforms)) return failm(MISTI_E_ARG, "starts / jsfs_row / uniforms / x / llh is NULL");
    std::lock_guard<std::mutex> one_call(m->call_mu);
    return guarded("misti_multi_basinhopping", [&]() -> int {
        const int D = (int)m->ctx.size(), N = m->n_param;
        const std::vector<int64_t> lo = blocks(n_start, D);
        const int64_t per_start = (int64_t)niter * (N + 1);
        const std::function<int(int)> fn = [&](int d) -> int {
            if (d == m->throw_in_worker) throw std::runtime_error("misti_multi_test_throw_in_worker_");
            const int64_t a = lo[d], n = lo[d + 1] - lo[d];
            if (n == 0) return 0;
            return misti_basinhopping(m->ctx[d], n, starts + a * N, split_time, jsfs_row, niter, T, stepsize, interval, target_accept_rate, stepwise_factor,
                                      xatol, fatol, nm_maxiter, nm_maxfev, uniforms ? uniforms + a * per_start : nullptr, x + a * N, llh + a,
                                      nfev ? nfev + a : nullptr, failures ? failures + a : nullptr, accepted ? accepted + a : nullptr);
        };
        return on_every_device(m, fn);
    });
}

}  // extern "C"

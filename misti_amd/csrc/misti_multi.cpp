// Multi-device form of the C ABI (include/misti_hip.h, "several devices"): SURVEY 8b's "variant taking a device list for
// 1/2/4/8-GPU runs".  Replaces the reference's way of using more than one processor - `parallel -j 20 ... >> res.out`,
// /root/reference/README.md:110-115, and the bash loops of test.bs/*.sh: one OS process per grid point, results concatenated
// from stdout - for a caller that binds the C ABI from ONE process: one engine context and one host thread per listed device
// (a device may be listed more than once), candidates dealt out by whole lambda-correction CHAINS, results written straight
// into the caller's buffers.  No exchange between devices during evaluation (candidates are independent); no collective either:
// the host-buffer form ends in host memory, which every device's DMA engine reaches by itself.  (Processes that keep their
// results on the devices use one rank per GPU and an RCCL all_gather instead: misti_amd/dist.py.)
//
// Built on the public single-device entry points only (misti_create, misti_eval_batch, misti_nm_solve, misti_basinhopping).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/misti_hip.h"

extern "C" int misti_set_error_(int code, const char* msg);     // misti_api.cpp: sets the calling thread's misti_last_error

struct misti_multi {
    std::vector<misti_ctx*> ctx;
    std::vector<int> device;
    int n_param = 0, n_band = 0, numT = 0;
    std::vector<int64_t> last_cands, last_chains;        // shard sizes of the last misti_multi_eval_batch
};

namespace {

int failm(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return misti_set_error_(code, buf);
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

// run fn(d) on one host thread per context; returns the first failure (its message becomes the caller's misti_last_error)
template <class F>
int on_every_device(misti_multi* m, F fn) {
    const int D = (int)m->ctx.size();
    std::vector<int> rc(D, 0);
    std::vector<std::string> msg(D);
    auto work = [&](int d) {
        rc[d] = fn(d);
        if (rc[d] != 0) msg[d] = misti_last_error();         // thread-local in the worker: carry it over
    };
    if (D == 1) work(0);
    else {
        std::vector<std::thread> th;
        th.reserve(D);
        for (int d = 0; d < D; ++d) th.emplace_back(work, d);
        for (auto& t : th) t.join();
    }
    for (int d = 0; d < D; ++d)
        if (rc[d] != 0) return failm(rc[d], "device %d (context %d of %d): %s", m->device[d], d, D, msg[d].c_str());
    return 0;
}

// contiguous blocks of starts: block d = [lo[d], lo[d + 1])
std::vector<int64_t> blocks(int64_t n, int D) {
    std::vector<int64_t> lo(D + 1);
    for (int d = 0; d <= D; ++d) lo[d] = n * d / D;
    return lo;
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
    misti_multi* m = new misti_multi;
    m->n_param = model->n_param; m->n_band = model->n_band; m->numT = model->numT;
    for (int d = 0; d < n_dev; ++d) {
        misti_ctx* c = nullptr;
        const int r = misti_create(model, devices[d], &c);
        if (r != 0) {
            const std::string why = misti_last_error();
            for (misti_ctx* q : m->ctx) (void)misti_destroy(q);
            delete m;
            return failm(r, "device %d: %s", devices[d], why.c_str());
        }
        m->ctx.push_back(c);
        m->device.push_back(devices[d]);
    }
    m->last_cands.assign(n_dev, 0);
    m->last_chains.assign(n_dev, 0);
    *out = m;
    return 0;
}

int misti_destroy_multi(misti_multi* m) {
    if (!m) return 0;
    int rc = 0;
    for (misti_ctx* c : m->ctx) { const int r = misti_destroy(c); if (r != 0 && rc == 0) rc = r; }
    delete m;
    return rc;
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
    for (size_t d = 0; d < m->ctx.size(); ++d) {
        if (n_cand) n_cand[d] = m->last_cands[d];
        if (n_chain) n_chain[d] = m->last_chains[d];
    }
    return 0;
}

int misti_multi_eval_batch(misti_multi* m, int64_t n_cand, const double* split, const double* params, const int32_t* band_bounds,
                           int64_t n_rep, const double* jsfs, double* llk, double* jafs, double* lc, double* pr, int32_t* status) {
    if (!m) return failm(MISTI_E_ARG, "multi context is NULL");
    if (n_cand < 0 || n_rep < 0) return failm(MISTI_E_ARG, "negative batch size");
    const int D = (int)m->ctx.size();
    for (int d = 0; d < D; ++d) m->last_cands[d] = m->last_chains[d] = 0;
    if (n_cand == 0) return 0;
    if (!split) return failm(MISTI_E_ARG, "split_time is NULL");
    const int P = m->n_param, B = m->n_band, numT = m->numT;
    if (P > 0 && !params) return failm(MISTI_E_ARG, "params is NULL but the model has %d parameters", P);
    if (n_rep > 0 && (!jsfs || !llk)) return failm(MISTI_E_ARG, "jsfs / llk is NULL with n_rep > 0");
    if (D == 1) {
        m->last_cands[0] = n_cand;
        return misti_eval_batch(m->ctx[0], n_cand, split, params, band_bounds, n_rep, jsfs, llk, jafs, lc, pr, status);
    }
    // ---- whole chains per device -----------------------------------------------------------------------------------------
    // Candidates with identical parameter vectors and band bounds share one lambda-correction chain, computed once per context
    // that holds any of them, at its full latency however few members that context has: a chain stays on ONE device, all its
    // split times with it.  Chains are dealt round-robin in order of first appearance (misti_amd/dist.py: chain_shards - the
    // rank-per-GPU path deals them the same way).  Without parameters and bounds the batch is one chain: candidates are
    // interleaved instead (every device repeats the chain; what is shared out is the spectrum kernel and the replicate epilogue).
    const bool with_bounds = band_bounds && B > 0;
    const size_t kp = (size_t)P * sizeof(double), kb = with_bounds ? (size_t)B * 2 * sizeof(int32_t) : 0;
    std::vector<std::vector<int64_t>> shard(D);
    if (kp + kb == 0) {
        for (int64_t c = 0; c < n_cand; ++c) shard[c % D].push_back(c);
        for (int d = 0; d < D; ++d) m->last_chains[d] = shard[d].empty() ? 0 : 1;
    } else {
        std::vector<char> keys((size_t)n_cand * (kp + kb));
        for (int64_t c = 0; c < n_cand; ++c) {
            char* k = keys.data() + (size_t)c * (kp + kb);
            if (kp) std::memcpy(k, params + (size_t)c * P, kp);
            if (kb) std::memcpy(k + kp, band_bounds + (size_t)c * B * 2, kb);
        }
        std::unordered_map<Key, int, KeyHash> owner;
        owner.reserve((size_t)n_cand / 4 + 16);
        int next = 0;
        for (int64_t c = 0; c < n_cand; ++c) {
            const Key k{keys.data() + (size_t)c * (kp + kb), kp + kb};
            auto it = owner.find(k);
            int d;
            if (it == owner.end()) { d = next % D; ++next; owner.emplace(k, d); m->last_chains[d] += 1; }
            else d = it->second;
            shard[d].push_back(c);
        }
    }
    for (int d = 0; d < D; ++d) m->last_cands[d] = (int64_t)shard[d].size();
    const size_t R = (size_t)n_rep, lc_row = (size_t)(numT + 1) * 2, pr_row = (size_t)(numT + 2) * 6;
    return on_every_device(m, [&](int d) -> int {
        const std::vector<int64_t>& idx = shard[d];
        const size_t n = idx.size();
        if (n == 0) return 0;
        std::vector<double> s(n), p(kp ? n * (size_t)P : 0), o_llk(R ? n * R : 0), o_jafs(jafs ? n * 7 : 0), o_lc(lc ? n * lc_row : 0), o_pr(pr ? n * pr_row : 0);
        std::vector<int32_t> b(kb ? n * (size_t)B * 2 : 0), o_st(status ? n : 0);
        for (size_t i = 0; i < n; ++i) {
            const size_t c = (size_t)idx[i];
            s[i] = split[c];
            if (kp) std::memcpy(&p[i * P], params + c * P, kp);
            if (kb) std::memcpy(&b[i * B * 2], band_bounds + c * B * 2, kb);
        }
        const int r = misti_eval_batch(m->ctx[d], (int64_t)n, s.data(), kp ? p.data() : nullptr, kb ? b.data() : nullptr, n_rep, jsfs,
                                       R ? o_llk.data() : nullptr, jafs ? o_jafs.data() : nullptr, lc ? o_lc.data() : nullptr,
                                       pr ? o_pr.data() : nullptr, status ? o_st.data() : nullptr);
        if (r != 0) return r;
        for (size_t i = 0; i < n; ++i) {                       // disjoint rows of the caller's buffers: no two threads share one
            const size_t c = (size_t)idx[i];
            if (R) std::memcpy(llk + c * R, &o_llk[i * R], R * sizeof(double));
            if (jafs) std::memcpy(jafs + c * 7, &o_jafs[i * 7], 7 * sizeof(double));
            if (lc) std::memcpy(lc + c * lc_row, &o_lc[i * lc_row], lc_row * sizeof(double));
            if (pr) std::memcpy(pr + c * pr_row, &o_pr[i * pr_row], pr_row * sizeof(double));
            if (status) status[c] = o_st[i];
        }
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
    const int D = (int)m->ctx.size(), N = m->n_param;
    const std::vector<int64_t> lo = blocks(n_start, D);
    return on_every_device(m, [&](int d) -> int {
        const int64_t a = lo[d], n = lo[d + 1] - lo[d];
        if (n == 0) return 0;
        return misti_nm_solve(m->ctx[d], n, starts + a * N, split_time, jsfs_row, xatol, fatol, maxiter, x + a * N, llh + a,
                              nit ? nit + a : nullptr, nfev ? nfev + a : nullptr, status ? status + a : nullptr);
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
    const int D = (int)m->ctx.size(), N = m->n_param;
    const std::vector<int64_t> lo = blocks(n_start, D);
    const int64_t per_start = (int64_t)niter * (N + 1);
    return on_every_device(m, [&](int d) -> int {
        const int64_t a = lo[d], n = lo[d + 1] - lo[d];
        if (n == 0) return 0;
        return misti_basinhopping(m->ctx[d], n, starts + a * N, split_time, jsfs_row, niter, T, stepsize, interval, target_accept_rate, stepwise_factor,
                                  xatol, fatol, nm_maxiter, nm_maxfev, uniforms ? uniforms + a * per_start : nullptr, x + a * N, llh + a,
                                  nfev ? nfev + a : nullptr, failures ? failures + a : nullptr, accepted ? accepted + a : nullptr);
    });
}

}  // extern "C"

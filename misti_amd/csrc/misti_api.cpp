// C-ABI host layer of libmisti_hip.so (see include/misti_hip.h).
//
// Replaces, for a batch, what the reference does once per Python object:
//   MigrationInference.__init__ / SetModel   MigrationInference.py:41-289   -> misti_create
//   JAFSLikelihood                           MigrationInference.py:566-614  -> misti_eval_batch*
// No CPU fallback: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <chrono>
#include <complex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "misti_device.h"
#include "misti_tables.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) return fail(MISTI_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

const misti::HostTables& host_tables() {
    static const misti::HostTables t = misti::build_tables();
    return t;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() { return static_cast<T*>(p); }
};

// Host-pinned staging of the host-buffer entry points: a copy from / to pageable memory makes the runtime stage the bytes itself
// and wait; from pinned memory hipMemcpyAsync is a DMA the stream orders like a kernel.
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipHostFree(p); if (e != hipSuccess) return e; p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};
constexpr size_t PIN_STAGE_MAX = (size_t)256 << 20;    // beyond this the host-buffer calls copy straight from / to the caller's memory

}  // namespace

struct misti_ctx {
    int device = 0;
    misti::DevModel dm{};
    int unfolded = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    DevBuf model_f64, model_i32;        // times | lh ; run_start | run_end
    DevBuf consts;                      // llh_const per replicate
    DevBuf ws_jafs, ws_status;          // spectra / status when the caller passes NULL
    DevBuf ws_order;                    // dispatch order (heaviest candidates first)
    DevBuf ws_diag;                     // per candidate: largest corrected rate x interval length of the last batch
    int64_t diag_n = 0;
    int32_t* hint_host = nullptr;       // pinned, device-visible: {chains, candidates} of the last batch (a launch-shape hint only)
    int32_t* hint_dev = nullptr;
    int32_t batch_seq = 0;
    misti::Tuning tune;                 // diagnostic launch-shape overrides, read from the environment once (misti_create)
    int table_cur = 0;                  // which of the two chain-table sets the next batch uses
    size_t table_clean[2] = {0, 0};     // table size each set is known to be clean for (0: not clean)
    DevBuf ws_trunk;                    // per chain: 44-state records before every interval (trunk kernel -> kernel 2)
    DevBuf ws_chain_f64, ws_chain_i32;  // chain buffers (kernel 1 -> kernel 2) and the chain table
    DevBuf st_split, st_params, st_bounds, st_jsfs, st_llk, st_jafs, st_lc, st_pr, st_status;   // staging for the host-buffer form
    PinBuf pin_in, pin_out;             // pinned staging of misti_eval_batch: inputs packed, outputs packed
    std::vector<char> heap_in, heap_out;   // pageable staging of the indexed form beyond PIN_STAGE_MAX (kept here: they outlive an error return's drain)
    bool trace = false;                 // solver trace (misti_enable_solver_trace)
    DevBuf ws_solver, ws_iters;         // per chain / per candidate solver words; trial points of small batches
    DevBuf ws_post;                     // default fit: rates after the split per candidate and interval (+ their solver words)
    int64_t trace_n = 0, trace_iter_cap = 0;   // candidates / chains covered by the trace of the last batch
    const int32_t* trace_of = nullptr;  // candidate -> chain of the last batch (device)
    hipEvent_t order_ev = nullptr;      // orders a replaced stream before its successor (misti_set_stream)
    hipEvent_t last_ev = nullptr;       // recorded behind every batch: lets OTHER contexts see whether this one has work in flight
    bool last_ev_set = false;
    unsigned hints = 0;                 // misti_set_hints: what the caller knows about its batches (MISTI_HINT_INTEGER_SPLITS)
    hipStream_t side_stream = nullptr;  // phase 1 of a two-phase batch (run_dev): what follows the chains a packed launch completed, beside its resume launch
    hipEvent_t packed_ev = nullptr, side_ev = nullptr;
    DevBuf nm_f64, nm_i32;              // batched Nelder-Mead: simplices, points, counters (misti_nm_solve)
    int32_t* nm_live_host = nullptr;    // pinned: live starts after the last two finished iterations
    int64_t nm_iterations = 0;          // iterations issued by the last misti_nm_solve
    int64_t nm_slots = 0;               // and the batch slots they had in total (live starts + the stale-count slack)
    int64_t nm_spec_iterations = 0;     // how many of those iterations were speculative (all points of a start in one batch)
    bool timing = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double ms[3] = {0, 0, 0};
    int64_t launches[3] = {0, 0, 0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[3];
};

namespace {

// Every live context of the process, so that a batch can tell whether the device is busy with other contexts' batches (the
// launch shape of mid-sized batches depends on it: latency when alone, throughput when not - misti_consts.h FOLLOW_BUSY_*).
std::mutex g_ctx_mu;
std::vector<misti_ctx*> g_ctxs;
// the device-wide table of chain waves per (compute unit, SIMD) (correct_follow_kernel's role choice): one per device, zeroed once
int32_t* g_pair_table[16] = {};
int32_t* pair_table(int device) {
    if (device < 0 || device >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    if (!g_pair_table[device]) {
        void* p = nullptr;
        if (hipMalloc(&p, misti::PAIR_TABLE * sizeof(int32_t)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, misti::PAIR_TABLE * sizeof(int32_t)) != hipSuccess) { (void)hipFree(p); return nullptr; }
        g_pair_table[device] = (int32_t*)p;
    }
    return g_pair_table[device];
}

int other_contexts_busy(const misti_ctx* self) {
    int busy = 0;
    std::lock_guard<std::mutex> lock(g_ctx_mu);
    for (misti_ctx* o : g_ctxs)
        if (o != self && o->device == self->device && o->last_ev_set && hipEventQuery(o->last_ev) == hipErrorNotReady) ++busy;
    (void)hipGetLastError();                       // "not ready" is an answer, not an error to be found by the next launch check
    return busy;
}

// Greedy runs of (numerically) constant lh, exactly as SmoothConst scans them
// (MigrationInference.py:387-405): a run starts at k and takes every j with
// |lh[j] - lh[k]| < 1e-10 while j < numT-1.
void smoothing_runs(const double* lh, int numT, int k, std::vector<int>& rs, std::vector<int>& re) {
    rs.assign(numT, 0);
    re.assign(numT, 0);
    int i = 0;
    while (i < numT) {
        int j = i;
        double lam = lh[2 * i + k];
        while (j < numT - 1 && std::fabs(lh[2 * j + k] - lam) < 1e-10) ++j;
        if (j == i) j = i + 1;                 // index numT-1: a run of its own (never smoothed: t < split <= numT-1)
        for (int t = i; t < j; ++t) { rs[t] = i; re[t] = j; }
        i = j;
    }
}

int validate_model(const misti_model_t* m) {
    if (!m) return fail(MISTI_E_ARG, "model is NULL");
    if (m->numT < 2) return fail(MISTI_E_ARG, "numT must be >= 2 (got %d)", m->numT);
    if (m->numT > MISTI_MAX_NUMT) return fail(MISTI_E_LIMIT, "numT %d exceeds MISTI_MAX_NUMT %d", m->numT, MISTI_MAX_NUMT);
    if (!m->times || !m->lh) return fail(MISTI_E_ARG, "times / lh is NULL");
    if (m->n_band < 0 || m->n_band > MISTI_MAX_BANDS) return fail(MISTI_E_LIMIT, "n_band %d out of range", m->n_band);
    if (m->n_pulse < 0 || m->n_pulse > MISTI_MAX_PULSES) return fail(MISTI_E_LIMIT, "n_pulse %d out of range", m->n_pulse);
    if (m->n_param < 0 || m->n_param > MISTI_MAX_PARAMS) return fail(MISTI_E_LIMIT, "n_param %d out of range", m->n_param);
    if ((m->n_band && !m->bands) || (m->n_pulse && !m->pulses)) return fail(MISTI_E_ARG, "bands / pulses is NULL");
    if (m->sample_date < 0 || m->sample_date >= m->numT) return fail(MISTI_E_ARG, "sample_date %d out of range", m->sample_date);
    for (int t = 0; t < m->numT - 1; ++t)
        if (!(m->times[t] >= 0) || !std::isfinite(m->times[t])) return fail(MISTI_E_ARG, "times[%d] is negative or not finite", t);
    for (int t = 0; t < 2 * m->numT; ++t)
        if (!(m->lh[t] > 0) || !std::isfinite(m->lh[t])) return fail(MISTI_E_ARG, "lh[%d] must be positive and finite", t);
    for (int b = 0; b < m->n_band; ++b) {
        const misti_band_t& B = m->bands[b];
        // SetModel, MigrationInference.py:237-247
        if (B.pop != 0 && B.pop != 1) return fail(MISTI_E_ARG, "band %d: population index should be 1 or 2", b);
        if (B.start < m->sample_date) return fail(MISTI_E_ARG, "band %d: migration start (%d) should be >= sample date (%d)", b, B.start, m->sample_date);
        if (B.end != -1 && B.end <= B.start) return fail(MISTI_E_ARG, "band %d: migration start (%d) should be strictly less than migration end (%d)", b, B.start, B.end);
        if (B.end > m->numT + 1) return fail(MISTI_E_ARG, "band %d: end %d beyond the grid", b, B.end);
        if (B.param < -1 || B.param >= m->n_param) return fail(MISTI_E_ARG, "band %d: param index %d out of range", b, B.param);
        if (B.param < 0 && !(B.value >= 0)) return fail(MISTI_E_ARG, "band %d: fixed rate must be >= 0", b);
        for (int c = 0; c < b; ++c) {                                   // :254-255 overlap
            const misti_band_t& C = m->bands[c];
            if (C.pop != B.pop) continue;
            int be = B.end < 0 ? INT32_MAX : B.end, ce = C.end < 0 ? INT32_MAX : C.end;
            if (B.start < ce && C.start < be) return fail(MISTI_E_ARG, "bands %d and %d: migration rate intervals should not overlap", c, b);
        }
    }
    for (int p = 0; p < m->n_pulse; ++p) {
        const misti_pulse_t& P = m->pulses[p];
        // :260-277
        if (P.pop != 0 && P.pop != 1) return fail(MISTI_E_ARG, "pulse %d: population index should be 1 or 2", p);
        if (P.time < m->sample_date) return fail(MISTI_E_ARG, "pulse %d: time (%d) should be >= sample date (%d)", p, P.time, m->sample_date);
        if (P.time >= m->numT + 1) return fail(MISTI_E_ARG, "pulse %d: time %d beyond the grid", p, P.time);
        if (P.param < -1 || P.param >= m->n_param) return fail(MISTI_E_ARG, "pulse %d: param index %d out of range", p, P.param);
        if (P.param < 0 && !(P.value >= 0 && P.value <= 1)) return fail(MISTI_E_ARG, "pulse %d: pulse migration rate should be between 0 and 1", p);
        for (int c = 0; c < p; ++c)
            if (m->pulses[c].time == P.time) return fail(MISTI_E_ARG, "pulses %d and %d: only single-direction pulse migration at a time", c, p);
    }
    return 0;
}

// Fold the finished event pairs of one stage into its totals (all of them when `all`, else only while more than
// `keep` are pending: a caller that enables timing and never reads it must not accumulate events without bound).
int drain_pending(misti_ctx* c, int which, bool all, size_t keep = 256) {
    auto& v = c->pending[which];
    size_t done = 0;
    while (done < v.size() && (all || v.size() - done > keep)) {
        HIP_TRY(hipEventSynchronize(v[done].second));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, v[done].first, v[done].second));
        c->ms[which] += t;
        (void)hipEventDestroy(v[done].first);
        (void)hipEventDestroy(v[done].second);
        ++done;
    }
    v.erase(v.begin(), v.begin() + (long)done);
    return 0;
}
int record_begin(misti_ctx* c, int which, hipEvent_t* a, hipEvent_t* b) {
    *a = *b = nullptr;
    if (!c->timing) return 0;
    if (int r = drain_pending(c, which, false)) return r;
    hipError_t e = hipEventCreate(a);
    if (e == hipSuccess) e = hipEventCreate(b);
    if (e == hipSuccess) e = hipEventRecord(*a, c->stream);
    if (e != hipSuccess) {
        if (*a) (void)hipEventDestroy(*a);
        if (*b) (void)hipEventDestroy(*b);
        *a = *b = nullptr;
        return fail(MISTI_E_HIP, "timing events: %s", hipGetErrorString(e));
    }
    return 0;
}
// which < 0: the stage failed - drop the pair
int record_end(misti_ctx* c, int which, hipEvent_t a, hipEvent_t b) {
    if (!a) return 0;
    hipError_t e = which >= 0 ? hipEventRecord(b, c->stream) : hipErrorUnknown;
    if (e != hipSuccess) {
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        return which >= 0 ? fail(MISTI_E_HIP, "hipEventRecord: %s", hipGetErrorString(e)) : 0;
    }
    c->pending[which].push_back({a, b});
    return 0;
}
// a launch between record_begin and record_end failed: release the pair, keep the error
#define HIP_TRY_EV(expr, a, b)                                                            \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { (void)record_end(c, -1, a, b); return fail(MISTI_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)

// hints (what an internal caller knows about its own batch; 0 from the ABI's entry points, whose split times live on the device):
//   RUN_INTEGER_SPLITS  no split time has a fractional part (or is negative: an empty slot) - no tail for the post launch
//   RUN_UNSHARED        every candidate has its own parameter vector: no trunk for a batch too large for one chain per wave
//   RUN_ONE_LENGTH      every chain has the same number of intervals: the chains need no sorting by length
enum : unsigned { RUN_INTEGER_SPLITS = 1u, RUN_UNSHARED = 2u, RUN_ONE_LENGTH = 4u };
int run_dev_impl(misti_ctx* c, int64_t n_cand, const double* d_split, const double* d_params, const int32_t* d_bounds, int64_t n_rep, const double* d_jsfs,
                 double* d_llk, double* d_jafs, double* d_lc, double* d_pr, int32_t* d_status, unsigned hints) {
    if (n_cand < 0 || n_rep < 0) return fail(MISTI_E_ARG, "negative batch size");
    if (n_cand == 0) return 0;
    if (!d_split) return fail(MISTI_E_ARG, "split_time is NULL");
    if (c->hints & MISTI_HINT_INTEGER_SPLITS) hints |= RUN_INTEGER_SPLITS;           // the caller's word (verified on the device: spectrum_kernel)
    if (c->dm.n_param > 0 && !d_params) return fail(MISTI_E_ARG, "params is NULL but the model has %d parameters", c->dm.n_param);
    if (n_rep > 0 && (!d_jsfs || !d_llk)) return fail(MISTI_E_ARG, "jsfs / llk is NULL with n_rep > 0");
    HIP_TRY(hipSetDevice(c->device));
    if (!d_jafs) { HIP_TRY(c->ws_jafs.reserve((size_t)n_cand * 7 * sizeof(double))); d_jafs = c->ws_jafs.as<double>(); }
    if (!d_status) { HIP_TRY(c->ws_status.reserve((size_t)n_cand * sizeof(int32_t))); d_status = c->ws_status.as<int32_t>(); }
    if (n_cand > INT32_MAX / 8) return fail(MISTI_E_LIMIT, "n_cand too large for one call");
    const size_t nc = (size_t)n_cand, numT = (size_t)c->dm.numT;
    // chain machinery: one allocation per type, carved below
    const size_t tsize = misti::chain_table_size(n_cand);
    const int64_t follow_max_h = c->tune.follow_max > 0 ? c->tune.follow_max : misti::FOLLOW_MAX_CHAINS;
    const size_t ntr = ((hints & RUN_UNSHARED) && n_cand > follow_max_h) ? 0 : (size_t)misti::trunk_capacity(n_cand, c->tune);
    const size_t f64_n = nc * numT * 2 + nc * (numT + 1) * 6 + nc * 6 + nc * 2 + nc * 6;
    // chain tables: TWO sets {n_chains[2], table, slot_chain, slot_len} used alternately - a batch clears the other set for its successor
    const size_t set_n = 4 + 4 * tsize;                     // counters [4] | table | slot_chain | slot_len | slot_keep
    const size_t i32_n = 2 * set_n + 10 * nc + ntr;         // two table sets | slot_of, of, chain_slot, rep, fail_t, fail_status, tail_status, chain_order, resume_t, resume_list | trunk_ok
    HIP_TRY(c->ws_chain_f64.reserve(f64_n * sizeof(double)));
    {
        void* before = c->ws_chain_i32.p;
        HIP_TRY(c->ws_chain_i32.reserve(i32_n * sizeof(int32_t)));
        if (c->ws_chain_i32.p != before) c->table_clean[0] = c->table_clean[1] = 0;      // a new allocation: nothing is clean
    }
    HIP_TRY(c->ws_order.reserve(nc * sizeof(int32_t)));
    if (ntr) HIP_TRY(c->ws_trunk.reserve(ntr * numT * misti::TRUNK_REC * sizeof(double)));
    if (n_rep > 0) HIP_TRY(c->consts.reserve((size_t)n_rep * sizeof(double)));
    misti::ChainBufs cb;
    {
        double* d = c->ws_chain_f64.as<double>();
        cb.lc = d; d += nc * numT * 2;
        cb.trace = d; d += nc * (numT + 1) * 6;
        cb.work = d; d += nc * 6;
        cb.tail_lc = d; d += nc * 2;
        cb.tail_state = d;
        int32_t* q = c->ws_chain_i32.as<int32_t>();
        const int cur = c->table_cur;
        int32_t* set[2] = {q, q + set_n};
        q += 2 * set_n;
        // a set is clean for this batch if the previous batch cleared it for exactly this table size (same layout)
        if (c->table_clean[cur] != tsize) HIP_TRY(hipMemsetAsync(set[cur], 0, set_n * sizeof(int32_t), c->stream));
        c->table_clean[cur] = 0;                    // about to be used
        c->table_clean[cur ^ 1] = 0;                // until the setup launch below is in the stream
        cb.n_chains = set[cur]; cb.table = set[cur] + 4; cb.slot_chain = cb.table + tsize; cb.slot_len = cb.slot_chain + tsize;
        cb.slot_keep = cb.slot_len + tsize;
        cb.z_n_chains = set[cur ^ 1]; cb.z_table = set[cur ^ 1] + 4; cb.z_slot_len = cb.z_table + 2 * tsize; cb.z_slot_keep = cb.z_slot_len + tsize;
        cb.slot_of = q; q += nc; cb.of = q; q += nc; cb.chain_slot = q; q += nc; cb.rep = q; q += nc;
        cb.fail_t = q; q += nc; cb.fail_status = q; q += nc; cb.tail_status = q; q += nc; cb.chain_order = q; q += nc; cb.resume_t = q; q += nc; cb.resume_list = q; q += nc;
        cb.trunk_ok = q;
        cb.tmask = (uint32_t)(tsize - 1);
        cb.trunk = ntr ? c->ws_trunk.as<double>() : nullptr;
        cb.trunk_cap = (int64_t)ntr;
        cb.hint = c->hint_dev;
        cb.simd_load = c->tune.pairing ? pair_table(c->device) : nullptr;
        cb.bounds = c->dm.n_band > 0 ? d_bounds : nullptr;
        cb.integer_splits = (hints & RUN_INTEGER_SPLITS) ? 1 : 0;
        cb.post_lam = nullptr;
        cb.post_word = nullptr;
        if (!(c->dm.flags & MISTI_CPFIT)) {
            HIP_TRY(c->ws_post.reserve(nc * (numT + 1) * (sizeof(double) + sizeof(int32_t))));
            cb.post_lam = c->ws_post.as<double>();
            if (c->trace) cb.post_word = reinterpret_cast<int32_t*>(cb.post_lam + nc * (numT + 1));
        }
        cb.solver = cb.tail_solver = cb.cand_solver = nullptr;
        cb.iters = nullptr;
        cb.iter_cap = 0;
        c->trace_n = 0;
        c->trace_iter_cap = 0;
        if (c->trace) {
            HIP_TRY(c->ws_solver.reserve((nc * numT + nc + nc * (numT + 1)) * sizeof(int32_t)));
            HIP_TRY(hipMemsetAsync(c->ws_solver.p, 0, (nc * numT + nc + nc * (numT + 1)) * sizeof(int32_t), c->stream));
            cb.solver = c->ws_solver.as<int32_t>();
            cb.tail_solver = cb.solver + nc * numT;
            cb.cand_solver = cb.tail_solver + nc;
            c->trace_n = n_cand;
            c->trace_of = cb.of;
            if (n_cand <= MISTI_TRACE_MAX_CAND) {
                const size_t it_n = nc * numT * MISTI_TRACE_MAX_ITER * 2;
                HIP_TRY(c->ws_iters.reserve(it_n * sizeof(double)));
                HIP_TRY(hipMemsetAsync(c->ws_iters.p, 0xff, it_n * sizeof(double), c->stream));      // all-ones = NaN: "no such iterate"
                cb.iters = c->ws_iters.as<double>();
                cb.iter_cap = n_cand;
                c->trace_iter_cap = n_cand;
            }
        }
    }
    // Launch shape of kernel 1 depends on the number of chains, which lives on the device: setup_kernel drops
    // {chains, candidates, batch tag} into pinned memory.  A batch of the same size as the previous one on this
    // context uses that one's count (no waiting; a stale value costs speed only); otherwise the host waits for
    // this batch's own count - bounded, a few microseconds after the launch.
    c->batch_seq += 1;
    cb.seq = c->batch_seq;
    cb.unsorted = (hints & RUN_ONE_LENGTH) ? 3 : 0;          // bit 0: candidates in their own order, bit 1: chains in order of arrival
    int64_t est_chains = -1;
    const volatile int32_t* hint = c->hint_host;
    if (hint && hint[1] == (int32_t)n_cand && hint[0] > 0 && hint[0] <= n_cand) est_chains = hint[0];
    // chains per wavefront: packed (up to 8) when the batch is large - fewer instructions in total, which is what
    // counts when batches overlap - unless it is known to collapse into a few long chains (pure latency: one chain
    // per wave and the trunk following it)
    int busy_seen = -1;                                      // other contexts with a batch in flight (looked up at most once per batch)
    auto busy = [&]() { if (busy_seen < 0) busy_seen = other_contexts_busy(c); return busy_seen; };
    auto shape = [&](int64_t est, int& cpw, bool& follow) {
        int64_t follow_max = c->tune.follow_max > 0 ? c->tune.follow_max : misti::FOLLOW_MAX_CHAINS;
        // mid-sized batches: one chain per wave is the latency shape (a 1 024-chain batch then occupies every wave slot of the chip);
        // when other contexts have batches in flight the caller is after throughput and the packed shape carries ten chains per
        // instruction stream (up to 1.9 x the rate).  Looked at only where it matters; the result never depends on it.
        const int busy_from = c->tune.busy_contexts >= 0 ? c->tune.busy_contexts : misti::FOLLOW_BUSY_CONTEXTS;
        if (c->tune.follow_max <= 0 && busy_from > 0 && est > misti::FOLLOW_BUSY_CHAINS && est <= follow_max && busy() >= busy_from)
            follow_max = misti::FOLLOW_BUSY_CHAINS;
        cpw = (est >= 0 && est <= follow_max) ? 1 : misti::correct_cands_per_wave(n_cand, c->tune);
        // The default fit's one-chain-per-wave kernel holds ONE wave per SIMD (512 registers): with its trunk wave beside it a 64-chain batch
        // takes 128 of the chip's 1 024 wave slots and eight batches fill it, where --cpfit fits sixteen.  With other contexts' batches in
        // flight a small default-fit batch therefore packs two chains per wave (trunks in the next launch): measured on the headline grid
        // with 20 batches in flight 2.03 -> 2.62e7 evals/s (4 per wave 2.39, 8: 1.94, 10: 2.11; trunks after the chains but one chain
        // per wave 2.46); alone it keeps the latency shape (1.73 ms against 2.54).  The result never depends on it.
        if (!(c->dm.flags & MISTI_CPFIT) && cpw == 1 && est >= 0 && est <= misti::FOLLOW_BUSY_CHAINS && busy_from > 0 && c->tune.follow_max <= 0 &&
            busy() >= busy_from)
            cpw = 2;
        const int f = c->tune.chains_per_wave;                       // diagnostic override (scratch experiments, tests)
        if (f == 1 || f == 2 || f == 4 || f == 8 || f == 10) cpw = f;
        follow = misti::trunk_follows(cpw, (int64_t)ntr, c->tune);
    };
    // a batch is four launches (+1 for the default fit, +1 with more than LLK_INLINE_MAX replicates): setup | chains |
    // trunks + tails | candidates (+ replicate epilogue).  Few launches matter when many batches are in flight.
    int32_t* d_order = c->ws_order.as<int32_t>();
    double* d_consts = n_rep > 0 ? c->consts.as<double>() : nullptr;
    const bool llk_inline = n_rep > 0 && n_rep <= misti::LLK_INLINE_MAX;
    hipEvent_t a = nullptr, b = nullptr;
    if (int r = record_begin(c, 0, &a, &b)) return r;
    HIP_TRY_EV(misti::launch_setup(c->dm, n_cand, d_params, d_split, cb, d_order, n_rep, d_jsfs, d_consts, c->unfolded, c->stream), a, b);
    c->table_clean[c->table_cur ^ 1] = tsize;       // setup_kernel clears the other set (table, slot_len, counters) for this table size
    c->table_cur ^= 1;
    if (est_chains < 0 && hint && misti::correct_cands_per_wave(n_cand, c->tune) > 1) {
        const auto t0 = std::chrono::steady_clock::now();
        while (hint[2] != cb.seq && std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(500)) std::this_thread::yield();
        if (hint[2] == cb.seq && hint[1] == (int32_t)n_cand) est_chains = hint[0];
    }
    int cpw_chains = 1;
    bool follow = false;
    shape(est_chains, cpw_chains, follow);
    // Packed launches let a chain whose solve runs away yield to a second launch, one chain per wave (latency: a packed launch ends
    // with its few long chains) - unless other contexts keep the device busy, where the packed shape's throughput is what counts.
    // --cpfit only by default: the default fit's one-per-wave kernel holds one wave per SIMD, half the resume launch's parallelism, and
    // its solves are short (bounded 2-D fits) - measured on 16 384 chains 3.59 ms without yielding, 4.39 with (MISTI_YIELD_NFEV forces it).
    int yield_nfev = 0;
    if (cpw_chains > 1) {
        yield_nfev = c->tune.yield_nfev >= 0 ? c->tune.yield_nfev : ((c->dm.flags & MISTI_CPFIT) ? misti::YIELD_NFEV : 0);
        const int busy_from = c->tune.busy_contexts >= 0 ? c->tune.busy_contexts : misti::FOLLOW_BUSY_CONTEXTS;
        if (yield_nfev > 0 && busy_from > 0 && busy() >= busy_from) yield_nfev = 0;
    }
    // Two phases behind a packed launch that lets chains yield (--cpfit; misti_kernels.hip: wrong_phase): what waits for the chains that launch
    // completed - trunks, tails, the candidate kernel of their members: 91 % of BASELINE config 3 - runs on a second stream BESIDE the resume
    // launch, which is one chain latency long and leaves most of the chip idle; the members of the chains that yielded follow behind it on the
    // batch's own stream, which then waits for the second one.  MISTI_TWO_PHASE=0 keeps everything on one stream.
    // Only while NO other context of the process has a batch in flight: the second stream is one more hardware queue (a process gets 24, and
    // twenty contexts with a second stream each collapse the overlapped rate: config 5 4.5e7 -> 2.9e7 evals/s measured), and beside other contexts'
    // batches there is no idle chip to fill.
    const bool two_phase = yield_nfev > 0 && (c->dm.flags & MISTI_CPFIT) && c->tune.two_phase && busy() == 0;
    if (two_phase) {
        if (!c->side_stream) HIP_TRY_EV(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking), a, b);
        if (!c->packed_ev) HIP_TRY_EV(hipEventCreateWithFlags(&c->packed_ev, hipEventDisableTiming), a, b);
        if (!c->side_ev) HIP_TRY_EV(hipEventCreateWithFlags(&c->side_ev, hipEventDisableTiming), a, b);
    }
    HIP_TRY_EV(misti::launch_correct(c->dm, n_cand, cb, d_split, d_params, cpw_chains, follow, est_chains, c->tune, yield_nfev, c->stream,
                                     two_phase ? c->packed_ev : nullptr), a, b);
    // Kernel 2 in single-wave workgroups (they slip into any free wave slot while other batches' chain kernels are resident: +21 % on
    // the headline grid with 20 batches in flight) - except behind a chip-filling one-chain-per-wave launch of this context's own:
    // there the next such launch was measured 18 % slower after single-wave workgroups (1.66 -> 1.96 ms on 1 024 chains; the placement
    // of its chain and trunk waves on the SIMDs follows where the previous kernel's workgroups ended), and such a batch runs alone -
    // with other contexts busy it would have taken the packed shape.
    const bool single_waves = c->tune.k2_single_waves >= 0 ? c->tune.k2_single_waves != 0 : !(follow && est_chains > 256);
    if (int r = record_end(c, 0, a, b)) return r;
    if (c->timing) c->launches[0] += 1;
    HIP_TRY(c->ws_diag.reserve(nc * sizeof(double)));
    c->diag_n = n_cand;
    if (int r = record_begin(c, 1, &a, &b)) return r;
    const bool skip_post = (hints & RUN_INTEGER_SPLITS) && (follow || ntr == 0);
    if (two_phase) {
        HIP_TRY_EV(hipStreamWaitEvent(c->side_stream, c->packed_ev, 0), a, b);
        HIP_TRY_EV(misti::launch_spectrum(c->dm, n_cand, d_order, d_split, d_params, cb, d_lc, d_pr, d_jafs, d_status, c->ws_diag.as<double>(),
                                          n_rep, d_jsfs, d_consts, d_llk, follow, skip_post, single_waves, c->tune, c->side_stream, 1), a, b);
        HIP_TRY_EV(hipEventRecord(c->side_ev, c->side_stream), a, b);
    }
    HIP_TRY_EV(misti::launch_spectrum(c->dm, n_cand, d_order, d_split, d_params, cb, d_lc, d_pr, d_jafs, d_status, c->ws_diag.as<double>(),
                                      n_rep, d_jsfs, d_consts, d_llk, follow, skip_post, single_waves, c->tune, c->stream, two_phase ? 2 : 0), a, b);
    if (two_phase) HIP_TRY_EV(hipStreamWaitEvent(c->stream, c->side_ev, 0), a, b);            // the batch's stream is behind both phases from here on
    if (int r = record_end(c, 1, a, b)) return r;
    if (c->timing) c->launches[1] += 1;
    if (n_rep > 0 && !llk_inline) {
        if (int r = record_begin(c, 2, &a, &b)) return r;
        HIP_TRY_EV(misti::launch_llk(n_cand, d_jafs, d_status, n_rep, d_jsfs, d_consts, d_llk, c->unfolded, c->stream), a, b);
        if (int r = record_end(c, 2, a, b)) return r;
        if (c->timing) c->launches[2] += 1;
    }
    if (c->last_ev) {                              // behind the batch: other contexts read it (other_contexts_busy)
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        HIP_TRY(hipEventRecord(c->last_ev, c->stream));
        c->last_ev_set = true;
    }
    return 0;
}

// A two-phase batch joins its side stream to the batch's stream with its LAST wait; a failure anywhere between the phase-1 launch and
// that wait would return with phase-1 kernels in flight on a stream nothing else waits for (misti_sync and the host-buffer drain
// know the batch's stream) while the next batch's set-up rewrites the chain buffers they read.  So: on any failure the side stream is
// drained before the error is reported.
int run_dev(misti_ctx* c, int64_t n_cand, const double* d_split, const double* d_params, const int32_t* d_bounds, int64_t n_rep, const double* d_jsfs,
            double* d_llk, double* d_jafs, double* d_lc, double* d_pr, int32_t* d_status, unsigned hints = 0) {
    const int r = run_dev_impl(c, n_cand, d_split, d_params, d_bounds, n_rep, d_jsfs, d_llk, d_jafs, d_lc, d_pr, d_status, hints);
    if (r != 0 && c->side_stream) {
        const std::string why = g_err;
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipGetLastError();
        g_err = why;
    }
    return r;
}

}  // namespace

extern "C" {

int misti_abi_version(void) { return MISTI_ABI_VERSION; }
#ifndef MISTI_BUILD_ID
#define MISTI_BUILD_ID "unknown"
#endif
const char* misti_build_id(void) { return MISTI_BUILD_ID; }

const char* misti_last_error(void) { return g_err.c_str(); }

// internal (misti_multi.cpp): make `msg` the calling thread's last error; returns `code`
int misti_set_error_(int code, const char* msg) { return fail(code, "%s", msg ? msg : ""); }

int misti_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(MISTI_E_NODEV, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return n;
}

int misti_tables(int32_t* gen, int32_t* jaf) {
    try {
        const misti::HostTables& t = host_tables();
        if (gen)
            for (int k = 0; k < 4; ++k) for (int d = 0; d < misti::NS2; ++d) for (int s = 0; s < misti::NS2; ++s)
                gen[(k * misti::NS2 + d) * misti::NS2 + s] = t.gen[k][d][s];
        if (jaf)
            for (int s = 0; s < misti::NS2; ++s) for (int c = 0; c < 7; ++c) jaf[s * 7 + c] = t.jaf[s][c];
    } catch (const std::exception& e) {
        return fail(MISTI_E_ARG, "table construction failed: %s", e.what());
    }
    return 0;
}

int misti_create(const misti_model_t* model, int device, misti_ctx** out) {
    if (!out) return fail(MISTI_E_ARG, "out is NULL");
    *out = nullptr;
    if (int r = validate_model(model)) return r;
    int ndev = misti_device_count();
    if (ndev <= 0) return fail(MISTI_E_NODEV, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(MISTI_E_ARG, "device %d out of range (0..%d)", device, ndev - 1);
    misti_ctx* c = nullptr;
    // every failure after `new` releases what the context already owns (stream, pinned hint, device buffers)
    struct Guard { misti_ctx*& c; bool keep = false; ~Guard() { if (!keep && c) { misti_destroy(c); c = nullptr; } } } guard{c};
    try {
        c = new misti_ctx();
        c->tune = misti::read_tuning();
        const misti::HostTables& t = host_tables();
        c->device = device;
        HIP_TRY(hipSetDevice(device));
        // constant tables (per device; idempotent)
        misti::DevTables dt{};
        std::memcpy(dt.src, t.src, sizeof dt.src);
        std::memcpy(dt.kind, t.kind, sizeof dt.kind);
        std::memcpy(dt.mult, t.mult, sizeof dt.mult);
        std::memcpy(dt.dcnt, t.dcnt, sizeof dt.dcnt);
        for (int cl = 0; cl < 7; ++cl) {
            for (int s = 0; s < 64; ++s) dt.jaf[cl][s] = s < misti::NS2 ? t.jaf[s][cl] : 0;
            for (int s = 0; s < misti::NS1; ++s) dt.jaf1[cl][s] = t.jaf1[s][cl];
            for (int s = 0; s < misti::NS2; ++s) {
                const int wgt = t.jaf[s][cl];
                if (wgt < 0 || wgt > 7) throw std::runtime_error("spectrum weight out of range");
                for (int b = 0; b < 3; ++b) if ((wgt >> b) & 1) dt.jaf_bits[cl][b] |= 1ull << s;
            }
            for (int s = 0; s < misti::NS1; ++s) {
                const int wgt = t.jaf1[s][cl];
                if (wgt < 0 || wgt > 7) throw std::runtime_error("spectrum weight out of range");
                dt.jaf1_bits[cl] |= (unsigned)wgt << (3 * s);
            }
        }
        std::memcpy(dt.grp_lo, t.grp_lo, sizeof dt.grp_lo);
        std::memcpy(dt.grp_hi, t.grp_hi, sizeof dt.grp_hi);
        std::memcpy(dt.anc_n, t.anc_n, sizeof dt.anc_n);
        std::memcpy(dt.anc_dst, t.anc_dst, sizeof dt.anc_dst);
        std::memcpy(dt.anc_src, t.anc_src, sizeof dt.anc_src);
        std::memcpy(dt.pulse_n, t.pulse_n, sizeof dt.pulse_n);
        std::memcpy(dt.pulse_src, t.pulse_src, sizeof dt.pulse_src);
        std::memcpy(dt.pulse_ab, t.pulse_ab, sizeof dt.pulse_ab);
        {   // modified Talbot contour z(th) = N[-0.6122 + 0.5017 th cot(0.6407 th) + 0.2645 i th]
            // (Trefethen, Weideman, Schmelzer 2006), midpoint nodes th_k = -pi + (2k-1) pi/N, upper half
            const int N = misti::TALBOT_N;
            const double PI = 3.14159265358979323846;
            for (int j = 0; j < misti::TALBOT_HALF; ++j) {
                int k = N / 2 + 1 + j;
                double th = -PI + (2.0 * k - 1.0) * PI / N;
                double a = 0.6407 * th;
                std::complex<double> z(N * (-0.6122 + 0.5017 * th / std::tan(a)), N * 0.2645 * th);
                std::complex<double> dz(N * (0.5017 / std::tan(a) - 0.5017 * 0.6407 * th / (std::sin(a) * std::sin(a))), N * 0.2645);
                std::complex<double> cw = std::complex<double>(0.0, 1.0 / N) * std::exp(z) * dz;
                dt.tal_zr[j] = z.real(); dt.tal_zi[j] = z.imag(); dt.tal_cr[j] = cw.real(); dt.tal_ci[j] = cw.imag();
            }
        }
        HIP_TRY(misti::upload_tables(dt));
        // the closed form after the split assumes this one-population generator
        static const int want1[8][8] = {{-6, 0, 0, 0, 0, 0, 0, 0}, {1, -3, 0, 0, 0, 0, 0, 0}, {4, 0, -3, 0, 0, 0, 0, 0}, {1, 0, 0, -3, 0, 0, 0, 0},
                                        {0, 2, 1, 0, -1, 0, 0, 0}, {0, 0, 1, 2, 0, -1, 0, 0}, {0, 1, 0, 1, 0, 0, -1, 0}, {0, 0, 1, 0, 0, 0, 0, -1}};
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j)
            if (t.gen1[i][j] != want1[i][j]) return fail(MISTI_E_ARG, "one-population generator mismatch at (%d,%d)", i, j);

        const int numT = model->numT;
        HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
        if (hipHostMalloc((void**)&c->hint_host, 4 * sizeof(int32_t), hipHostMallocMapped) == hipSuccess) {
            c->hint_host[0] = c->hint_host[1] = c->hint_host[2] = 0;
            if (hipHostGetDevicePointer((void**)&c->hint_dev, c->hint_host, 0) != hipSuccess) c->hint_dev = nullptr;
        } else { (void)hipGetLastError(); c->hint_host = nullptr; }
        c->stream = c->own_stream;
        std::vector<double> f64((size_t)(numT - 1) + 2 * (size_t)numT);
        std::memcpy(f64.data(), model->times, sizeof(double) * (numT - 1));
        std::memcpy(f64.data() + (numT - 1), model->lh, sizeof(double) * 2 * numT);
        HIP_TRY(c->model_f64.reserve(f64.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(c->model_f64.p, f64.data(), f64.size() * sizeof(double), hipMemcpyHostToDevice));
        std::vector<int> runs(4 * (size_t)numT), rs, re;
        for (int k = 0; k < 2; ++k) {
            smoothing_runs(model->lh, numT, k, rs, re);
            std::memcpy(runs.data() + k * numT, rs.data(), sizeof(int) * numT);
            std::memcpy(runs.data() + (2 + k) * numT, re.data(), sizeof(int) * numT);
        }
        {
            // Where can a candidate leave its chain's trunk?  (trunk_leave in misti_kernels.hip, for every split an integer or a
            // fractional split time can produce.)  A function of the smoothing runs alone: the trunk stores records only there.
            const int* rs0 = runs.data(), *rs1 = runs.data() + numT, *re0 = runs.data() + 2 * numT, *re1 = runs.data() + 3 * numT;
            std::vector<int> ok((size_t)numT, 0);
            const bool smooth = (model->flags & MISTI_SMOOTH) != 0;
            for (int s = 0; s < numT; ++s) {
                int frac_own = s, int_own = s;
                if (smooth) {
                    frac_own = rs0[s] < rs1[s] ? rs0[s] : rs1[s];               // the interval s is split in two
                    if (s > 0) {
                        if (re0[s - 1] > s && rs0[s - 1] < int_own) int_own = rs0[s - 1];
                        if (re1[s - 1] > s && rs1[s - 1] < int_own) int_own = rs1[s - 1];
                    }
                }
                ok[(size_t)frac_own] = 1;
                ok[(size_t)int_own] = 1;
            }
            runs.insert(runs.end(), ok.begin(), ok.end());
        }
        HIP_TRY(c->model_i32.reserve(runs.size() * sizeof(int)));
        HIP_TRY(hipMemcpy(c->model_i32.p, runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice));
        misti::DevModel& d = c->dm;
        d.numT = numT;
        d.sample_date = model->sample_date;
        d.flags = model->flags;
        d.n_band = model->n_band; d.n_pulse = model->n_pulse; d.n_param = model->n_param;
        d.mixture_th = model->mixture_th;
        d.times = c->model_f64.as<double>();
        d.lh = d.times + (numT - 1);
        d.run_start = c->model_i32.as<int>();
        d.run_end = d.run_start + 2 * numT;
        d.leave_ok = d.run_start + 4 * numT;
        for (int b = 0; b < model->n_band; ++b) d.bands[b] = model->bands[b];
        for (int p = 0; p < model->n_pulse; ++p) d.pulses[p] = model->pulses[p];
        c->unfolded = (model->flags & MISTI_UNFOLDED) ? 1 : 0;
    } catch (const std::exception& e) {
        return fail(MISTI_E_ARG, "misti_create: %s", e.what());
    }
    if (hipEventCreateWithFlags(&c->last_ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->last_ev = nullptr; }
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        g_ctxs.push_back(c);
    }
    guard.keep = true;
    *out = c;
    return 0;
}

int misti_destroy(misti_ctx* c) {
    if (!c) return 0;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mu);
        for (size_t i = 0; i < g_ctxs.size(); ++i) if (g_ctxs[i] == c) { g_ctxs.erase(g_ctxs.begin() + (long)i); break; }
    }
    (void)hipSetDevice(c->device);
    // EVERYTHING the context ever issued is finished before anything it owns is released: the stream batches are issued on (the
    // caller's, after misti_set_stream), the context's own stream (work issued before a misti_set_stream is ordered before its
    // successor by an event, which this wait does not rely on) and the side stream of two-phase batches.  Only then buffers, the
    // mapped hint word the set-up kernel writes, events, and last the streams themselves.
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream && c->own_stream != c->stream) (void)hipStreamSynchronize(c->own_stream);
    if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    (void)hipGetLastError();
    for (auto* b : {&c->model_f64, &c->model_i32, &c->consts, &c->ws_jafs, &c->ws_status, &c->ws_chain_f64, &c->ws_chain_i32, &c->ws_order, &c->ws_diag, &c->ws_trunk, &c->ws_solver, &c->ws_iters, &c->ws_post,
                    &c->st_split, &c->st_params, &c->st_bounds, &c->st_jsfs, &c->st_llk, &c->st_jafs, &c->st_lc, &c->st_pr, &c->st_status, &c->nm_f64, &c->nm_i32})
        b->release();
    c->pin_in.release();
    c->pin_out.release();
    if (c->nm_live_host) (void)hipHostFree(c->nm_live_host);
    if (c->hint_host) (void)hipHostFree(c->hint_host);
    for (int w = 0; w < 3; ++w)
        for (auto& pr : c->pending[w]) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t e : {c->order_ev, c->last_ev, c->packed_ev, c->side_ev}) if (e) (void)hipEventDestroy(e);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    (void)hipGetLastError();
    delete c;
    return 0;
}

int misti_set_stream(misti_ctx* c, void* s) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    hipStream_t next = s ? static_cast<hipStream_t>(s) : c->own_stream;
    if (next == c->stream) return 0;
    // All batches of a context share its workspaces (chain table, rates, trunk records, the pinned hint) and the
    // *_dev calls are asynchronous: what was issued on the old stream must finish before anything issued on the new
    // one starts, or batch N+1 overwrites what batch N is still reading.
    HIP_TRY(hipSetDevice(c->device));
    if (!c->order_ev) HIP_TRY(hipEventCreateWithFlags(&c->order_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->order_ev, c->stream));
    HIP_TRY(hipStreamWaitEvent(next, c->order_ev, 0));
    c->stream = next;
    return 0;
}

int misti_set_hints(misti_ctx* c, uint32_t hints) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (hints & ~MISTI_HINT_INTEGER_SPLITS) return fail(MISTI_E_ARG, "unknown hint bits 0x%x", (unsigned)(hints & ~MISTI_HINT_INTEGER_SPLITS));
    c->hints = hints;
    return 0;
}

int misti_get_stream(misti_ctx* c, void** s) {
    if (!c || !s) return fail(MISTI_E_ARG, "ctx / output is NULL");
    *s = static_cast<void*>(c->stream);
    return 0;
}

int misti_sync(misti_ctx* c) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->side_stream) HIP_TRY(hipStreamSynchronize(c->side_stream));      // joined to the stream by every batch that used it; a failed one may not have got that far
    return 0;
}

int misti_eval_batch_dev(misti_ctx* c, int64_t n_cand, const double* d_split, const double* d_params, const int32_t* d_bounds, int64_t n_rep,
                         const double* d_jsfs, double* d_llk, double* d_jafs, double* d_lc, double* d_pr, int32_t* d_status) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    return run_dev(c, n_cand, d_split, d_params, d_bounds, n_rep, d_jsfs, d_llk, d_jafs, d_lc, d_pr, d_status);
}

int misti_llk_dev(misti_ctx* c, int64_t n_cand, const double* d_jafs, const int32_t* d_status, int64_t n_rep, const double* d_jsfs, double* d_llk) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_cand < 0 || n_rep < 0) return fail(MISTI_E_ARG, "negative batch size");
    if (n_cand == 0 || n_rep == 0) return 0;
    if (!d_jafs || !d_jsfs || !d_llk) return fail(MISTI_E_ARG, "jafs / jsfs / llk is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(c->consts.reserve((size_t)n_rep * sizeof(double)));
    HIP_TRY(misti::launch_llh_const(n_rep, d_jsfs, c->consts.as<double>(), c->unfolded, c->stream));
    hipEvent_t a = nullptr, b = nullptr;
    if (int r = record_begin(c, 2, &a, &b)) return r;
    HIP_TRY_EV(misti::launch_llk(n_cand, d_jafs, d_status, n_rep, d_jsfs, c->consts.as<double>(), d_llk, c->unfolded, c->stream), a, b);
    if (int r = record_end(c, 2, a, b)) return r;
    if (c->timing) c->launches[2] += 1;
    return 0;
}

int misti_argmax_dev(misti_ctx* c, int64_t n_cand, int64_t n_rep, const double* d_llk, int32_t* d_best, double* d_best_llk) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_cand < 0 || n_rep < 0) return fail(MISTI_E_ARG, "negative batch size");
    if (n_rep == 0) return 0;
    if (!d_best || (n_cand > 0 && !d_llk)) return fail(MISTI_E_ARG, "llk / best is NULL");
    if (n_cand > INT32_MAX) return fail(MISTI_E_LIMIT, "n_cand too large");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(misti::launch_argmax(n_cand, n_rep, d_llk, d_best, d_best_llk, c->stream));
    return 0;
}

}  // extern "C"

namespace {
int eval_batch_host(misti_ctx* c, int64_t n_cand, const double* split, const double* params, const int32_t* band_bounds, int64_t n_rep, const double* jsfs,
                    double* llk, double* jafs, double* lc, double* pr, int32_t* status, const int64_t* idx = nullptr);
int eval_batch_drained(misti_ctx* c, int r) {
    // An error between the first asynchronous copy and the final wait leaves DMAs reading the context's pinned input block or
    // writing its pinned output block: the next call would overwrite what they read.  Drain the stream before reporting.
    if (r != 0 && c) {
        const std::string why = g_err;
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        if (c->side_stream) (void)hipStreamSynchronize(c->side_stream);
        (void)hipGetLastError();
        g_err = why;
    }
    return r;
}
}

extern "C" {

int misti_eval_batch(misti_ctx* c, int64_t n_cand, const double* split, const double* params, const int32_t* band_bounds, int64_t n_rep, const double* jsfs,
                     double* llk, double* jafs, double* lc, double* pr, int32_t* status) {
    return eval_batch_drained(c, eval_batch_host(c, n_cand, split, params, band_bounds, n_rep, jsfs, llk, jafs, lc, pr, status));
}

// misti_eval_batch on the rows idx[0 .. n) of the caller's arrays: inputs are GATHERED from those rows into the context's staging
// block and every output row i is SCATTERED to row idx[i] - what a context of misti_multi_eval_batch does with its shard of a
// batch (misti_multi.cpp), without a staging copy of the shard in between.  jsfs is shared (not indexed).  Not part of the public ABI.
int misti_eval_batch_indexed_(misti_ctx* c, int64_t n, const int64_t* idx, const double* split, const double* params, const int32_t* band_bounds,
                              int64_t n_rep, const double* jsfs, double* llk, double* jafs, double* lc, double* pr, int32_t* status) {
    if (!idx) return fail(MISTI_E_ARG, "idx is NULL");
    try {
        return eval_batch_drained(c, eval_batch_host(c, n, split, params, band_bounds, n_rep, jsfs, llk, jafs, lc, pr, status, idx));
    } catch (const std::bad_alloc&) {
        return eval_batch_drained(c, fail(MISTI_E_NOMEM, "out of host memory staging %lld candidates", (long long)n));
    } catch (const std::exception& e) {
        return eval_batch_drained(c, fail(MISTI_E_ARG, "%s", e.what()));
    }
}

}  // extern "C"

namespace {

int eval_batch_host(misti_ctx* c, int64_t n_cand, const double* split, const double* params, const int32_t* band_bounds, int64_t n_rep, const double* jsfs,
                    double* llk, double* jafs, double* lc, double* pr, int32_t* status, const int64_t* idx) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_cand < 0 || n_rep < 0) return fail(MISTI_E_ARG, "negative batch size");
    if (n_cand == 0) return 0;
    if (!split) return fail(MISTI_E_ARG, "split_time is NULL");
    const int P = c->dm.n_param, numT = c->dm.numT;
    if (P > 0 && !params) return fail(MISTI_E_ARG, "params is NULL but the model has %d parameters", P);
    if (n_rep > 0 && (!jsfs || !llk)) return fail(MISTI_E_ARG, "jsfs / llk is NULL with n_rep > 0");
    HIP_TRY(hipSetDevice(c->device));
    const size_t nc = (size_t)n_cand, nr = (size_t)n_rep;
    const bool with_bounds = band_bounds && c->dm.n_band > 0;
    const size_t b_split = nc * sizeof(double), b_par = P > 0 ? nc * P * sizeof(double) : 0;
    const size_t b_bounds = with_bounds ? nc * 2 * (size_t)c->dm.n_band * sizeof(int32_t) : 0, b_jsfs = nr * 8 * sizeof(double);
    const size_t lc_n = nc * (size_t)(numT + 1) * 2, pr_n = nc * (size_t)(numT + 2) * 6;
    const size_t b_llk = nr ? nc * nr * sizeof(double) : 0, b_jafs = jafs ? nc * 7 * sizeof(double) : 0, b_lc = lc ? lc_n * sizeof(double) : 0;
    const size_t b_pr = pr ? pr_n * sizeof(double) : 0, b_status = status ? nc * sizeof(int32_t) : 0;
    const size_t in_bytes = b_split + b_par + b_bounds + b_jsfs, out_bytes = b_llk + b_jafs + b_lc + b_pr + b_status;
    // The caller's buffers are pageable: inputs are packed into ONE pinned block (a CPU memcpy of tens of KB), copied by truly
    // asynchronous DMAs; outputs come back into one pinned block and are handed over after the single wait at the end.
    const bool pinned = in_bytes <= PIN_STAGE_MAX && out_bytes <= PIN_STAGE_MAX &&
                        c->pin_in.reserve(in_bytes) == hipSuccess && c->pin_out.reserve(out_bytes ? out_bytes : 8) == hipSuccess;
    if (!pinned) (void)hipGetLastError();
    // Indexed form (idx: rows of the caller's arrays, misti_eval_batch_indexed_): rows are gathered on the way in and scattered on the
    // way out; beyond the pinned block's limit the gather goes through pageable heap blocks (std::bad_alloc is the caller's to catch).
    // They belong to the context: an error return leaves copies in flight until misti_eval_batch's drain, which comes AFTER this
    // function's locals are gone.
    std::vector<char>& heap_in = c->heap_in;
    std::vector<char>& heap_out = c->heap_out;
    if (idx && !pinned) { heap_in.resize(in_bytes); heap_out.resize(out_bytes ? out_bytes : 8); }
    char* hin = pinned ? static_cast<char*>(c->pin_in.p) : (idx ? heap_in.data() : nullptr);
    const bool staged = pinned || idx;
    // row = bytes per candidate of this array (0: not per candidate - the replicate table)
    auto h2d = [&](DevBuf& dst, const void* src, size_t bytes, size_t row, size_t& off) -> int {
        if (!bytes) return 0;
        HIP_TRY(dst.reserve(bytes));
        const void* from = src;
        if (staged) {
            if (idx && row) for (size_t i = 0; i < nc; ++i) std::memcpy(hin + off + i * row, static_cast<const char*>(src) + (size_t)idx[i] * row, row);
            else std::memcpy(hin + off, src, bytes);
            from = hin + off; off += bytes;
        }
        HIP_TRY(hipMemcpyAsync(dst.p, from, bytes, hipMemcpyHostToDevice, c->stream));
        return 0;
    };
    size_t off = 0;
    if (int r = h2d(c->st_split, split, b_split, sizeof(double), off)) return r;
    if (int r = h2d(c->st_params, params, b_par, (size_t)P * sizeof(double), off)) return r;
    if (int r = h2d(c->st_bounds, band_bounds, b_bounds, 2 * (size_t)c->dm.n_band * sizeof(int32_t), off)) return r;
    if (int r = h2d(c->st_jsfs, jsfs, b_jsfs, 0, off)) return r;
    if (nr) HIP_TRY(c->st_llk.reserve(nc * nr * sizeof(double)));
    HIP_TRY(c->st_jafs.reserve(nc * 7 * sizeof(double)));
    HIP_TRY(c->st_status.reserve(nc * sizeof(int32_t)));
    if (lc) HIP_TRY(c->st_lc.reserve(lc_n * sizeof(double)));
    if (pr) {
        HIP_TRY(c->st_pr.reserve(pr_n * sizeof(double)));
        HIP_TRY(hipMemsetAsync(c->st_pr.p, 0, pr_n * sizeof(double), c->stream));
    }
    // the split times are in host memory here: whether any has a fractional part is a glance (the device-buffer form needs misti_set_hints)
    bool whole = true;
    for (size_t i = 0; i < nc && whole; ++i) { const double st = split[idx ? (size_t)idx[i] : i]; whole = st == std::floor(st); }
    int r = run_dev(c, n_cand, c->st_split.as<double>(), P > 0 ? c->st_params.as<double>() : nullptr,
                    with_bounds ? c->st_bounds.as<int32_t>() : nullptr, n_rep,
                    nr ? c->st_jsfs.as<double>() : nullptr, nr ? c->st_llk.as<double>() : nullptr, c->st_jafs.as<double>(),
                    lc ? c->st_lc.as<double>() : nullptr, pr ? c->st_pr.as<double>() : nullptr, c->st_status.as<int32_t>(), whole ? RUN_INTEGER_SPLITS : 0u);
    if (r) return r;
    char* hout = pinned ? static_cast<char*>(c->pin_out.p) : (idx ? heap_out.data() : nullptr);
    struct Back { void* user; size_t off, bytes, row; };
    Back back[5];
    int n_back = 0;
    size_t ooff = 0;
    auto d2h = [&](void* user, const void* src, size_t bytes, size_t row) -> int {
        if (!bytes) return 0;
        void* to = user;
        if (staged) { to = hout + ooff; back[n_back++] = {user, ooff, bytes, row}; ooff += bytes; }
        HIP_TRY(hipMemcpyAsync(to, src, bytes, hipMemcpyDeviceToHost, c->stream));
        return 0;
    };
    if (int q = d2h(llk, c->st_llk.p, b_llk, nr * sizeof(double))) return q;
    if (int q = d2h(jafs, c->st_jafs.p, b_jafs, 7 * sizeof(double))) return q;
    if (int q = d2h(lc, c->st_lc.p, b_lc, (size_t)(numT + 1) * 2 * sizeof(double))) return q;
    if (int q = d2h(pr, c->st_pr.p, b_pr, (size_t)(numT + 2) * 6 * sizeof(double))) return q;
    if (int q = d2h(status, c->st_status.p, b_status, sizeof(int32_t))) return q;
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n_back; ++i) {
        if (idx) for (size_t k = 0; k < nc; ++k) std::memcpy(static_cast<char*>(back[i].user) + (size_t)idx[k] * back[i].row, hout + back[i].off + k * back[i].row, back[i].row);
        else std::memcpy(back[i].user, hout + back[i].off, back[i].bytes);
    }
    return 0;
}

}  // namespace

extern "C" {

int misti_forward_rates_dev(misti_ctx* c, int64_t n_cand, const double* d_split, const double* d_params, int hold_mu,
                            double* d_lh, double* d_pr, int32_t* d_status) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_cand < 0) return fail(MISTI_E_ARG, "negative batch size");
    if (n_cand == 0) return 0;
    if (!d_split || !d_lh) return fail(MISTI_E_ARG, "split_time / lh is NULL");
    if (c->dm.n_param > 0 && !d_params) return fail(MISTI_E_ARG, "params is NULL but the model has %d parameters", c->dm.n_param);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(misti::launch_forward(c->dm, n_cand, d_split, d_params, hold_mu != 0, d_lh, d_pr, d_status, c->stream));
    return 0;
}

int misti_forward_rates(misti_ctx* c, int64_t n_cand, const double* split, const double* params, int hold_mu,
                        double* lh, double* pr, int32_t* status) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_cand < 0) return fail(MISTI_E_ARG, "negative batch size");
    if (n_cand == 0) return 0;
    if (!split || !lh) return fail(MISTI_E_ARG, "split_time / lh is NULL");
    const int P = c->dm.n_param, numT = c->dm.numT;
    if (P > 0 && !params) return fail(MISTI_E_ARG, "params is NULL but the model has %d parameters", P);
    HIP_TRY(hipSetDevice(c->device));
    const size_t nc = (size_t)n_cand;
    const size_t lh_n = nc * (size_t)(numT + 1) * 2, pr_n = nc * (size_t)(numT + 2) * 6;
    HIP_TRY(c->st_split.reserve(nc * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->st_split.p, split, nc * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (P > 0) {
        HIP_TRY(c->st_params.reserve(nc * P * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(c->st_params.p, params, nc * P * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c->st_lc.reserve(lh_n * sizeof(double)));
    HIP_TRY(c->st_status.reserve(nc * sizeof(int32_t)));
    if (pr) HIP_TRY(c->st_pr.reserve(pr_n * sizeof(double)));
    int r = misti_forward_rates_dev(c, n_cand, c->st_split.as<double>(), P > 0 ? c->st_params.as<double>() : nullptr, hold_mu,
                                    c->st_lc.as<double>(), pr ? c->st_pr.as<double>() : nullptr, c->st_status.as<int32_t>());
    if (r) return r;
    HIP_TRY(hipMemcpyAsync(lh, c->st_lc.p, lh_n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (pr) HIP_TRY(hipMemcpyAsync(pr, c->st_pr.p, pr_n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) HIP_TRY(hipMemcpyAsync(status, c->st_status.p, nc * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"

namespace {

// Device-resident state of one batched Nelder-Mead run, carved from the context's nm buffers.
struct NmWork {
    misti::NmState st{};
    double *llk0 = nullptr, *llk1 = nullptr, *llk2 = nullptr, *llk3 = nullptr, *llk_spec = nullptr;
    double *d_starts = nullptr;      // [S][N] in: start points; out: best vertices
    double *d_llh = nullptr;         // [S]    out: their log-likelihood
    double *d_row = nullptr;         // [8]    the data JSFS
    double *extra_f64 = nullptr;     // whatever the caller asked for beyond the minimiser's own state
    int32_t *idx[2] = {nullptr, nullptr}, *cnt = nullptr, *extra_i32 = nullptr;
};

// Live starts up to which an iteration is speculative (misti_nm.hip): all 4 + N points of a start in one batch, as long as the
// batch still runs one chain per wave (FOLLOW_MAX_CHAINS candidates, each its own chain) - there the iteration costs one
// chain latency instead of three.  MISTI_NM_SPEC=0 turns it off (tests compare the two paths).
int64_t nm_spec_cap(int N) {
    const char* e = getenv("MISTI_NM_SPEC");
    if (e && e[0] == '0') return 0;
    return misti::FOLLOW_MAX_CHAINS / (4 + N);
}

int nm_prepare(misti_ctx* c, int64_t n_start, NmWork& w, size_t extra_f64 = 0, size_t extra_i32 = 0) {
    const int N = c->dm.n_param;
    const size_t S = (size_t)n_start, V = (size_t)N + 1;
    const size_t cap = (size_t)nm_spec_cap(N), K = 4 + (size_t)N;
    // one allocation per type: simplices and points | counters and slot tables
    const size_t f64_n = S * V * N * 2 + S * V + S * N * 2 + S * N * N + S + S * V + 2 * S + S * N      // state + split arrays
                         + S * V + 2 * S + S * N                                                           // llk of the four batches
                         + S * N + S + 8                                                                   // inputs / outputs, jsfs row
                         + cap * K * (N + 2) + extra_f64;                                                  // speculative points, their splits and values
    const size_t i32_n = 7 * S + 4 + extra_i32;
    HIP_TRY(c->nm_f64.reserve(f64_n * sizeof(double)));
    HIP_TRY(c->nm_i32.reserve(i32_n * sizeof(int32_t)));
    if (!c->nm_live_host) {
        if (hipHostMalloc((void**)&c->nm_live_host, 4 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->nm_live_host = nullptr;
            return fail(MISTI_E_HIP, "pinned allocation for the live-start count failed");
        }
    }
    misti::NmState& st = w.st;
    st.S = n_start; st.N = N;
    double* d = c->nm_f64.as<double>();
    st.sim = d; d += S * V * N; st.scratch = d; d += S * V * N; st.fsim = d; d += S * V;
    st.p1 = d; d += S * N; st.p2 = d; d += S * N; st.p3 = d; d += S * N * N; st.fxr = d; d += S;
    st.split0 = d; d += S * V; st.split1 = d; d += S; st.split2 = d; d += S; st.split3 = d; d += S * N;
    w.llk0 = d; d += S * V;
    w.llk1 = d; d += S;
    w.llk2 = d; d += S;
    w.llk3 = d; d += S * N;
    w.d_starts = d; d += S * N;
    w.d_llh = d; d += S;
    w.d_row = d; d += 8;
    st.ps = d; d += cap * K * N; st.ps_split = d; d += cap * K; w.llk_spec = d; d += cap * K;
    st.spec_cap = (int64_t)cap;
    w.extra_f64 = d;
    int32_t* q = c->nm_i32.as<int32_t>();
    st.nit = q; q += S; st.nfev = q; q += S; st.done = q; q += S; st.kind = q; q += S; st.shrunk = q; q += S;
    w.idx[0] = q; w.idx[1] = q + S; q += 2 * S;
    w.cnt = q; q += 4;                       // [2] live starts of the iteration in progress / of the next one
    w.extra_i32 = q;
    return 0;
}

// One minimisation of every start from w.d_starts (device); leaves best vertices in w.d_starts, their log-likelihood in w.d_llh,
// SciPy's counters in st.nit / st.nfev and the termination status in st.shrunk (0 converged, 1 evaluation budget, 2 iteration
// budget).  Asynchronous except for the 4-byte live counts; results are complete when the stream is.
int nm_run(misti_ctx* c, NmWork& w, double split_time, double xatol, double fatol, int32_t maxiter, int64_t maxfun) {
    misti::NmState& st = w.st;
    const int N = st.N;
    const size_t S = (size_t)st.S, V = (size_t)N + 1;
    st.maxiter = maxiter; st.maxfun = maxfun; st.xatol = xatol; st.fatol = fatol; st.split = split_time;
    hipStream_t sm = c->stream;
    // what the search knows about its own batches: one split time for every point (empty slots carry -1), distinct points
    static const int hint_mask = [] { const char* e = getenv("MISTI_NM_HINTS"); return e ? atoi(e) : 7; }();       // diagnostic: which hints the search passes on
    const unsigned nm_hints = ((split_time == std::floor(split_time) ? RUN_INTEGER_SPLITS : 0u) | RUN_UNSHARED | RUN_ONE_LENGTH) & (unsigned)hint_mask;
    int32_t* cnt = w.cnt;
    HIP_TRY(hipMemsetAsync(cnt, 0, 4 * sizeof(int32_t), sm));
    HIP_TRY(hipMemsetAsync(st.split1, 0xBF, S * sizeof(double), sm));          // all-0xBF bytes: a negative double = "no point in this slot"
    HIP_TRY(misti::launch_nm_init(st, w.d_starts, sm));
    if (int r = run_dev(c, (int64_t)(S * V), st.split0, st.sim, nullptr, 1, w.d_row, w.llk0, nullptr, nullptr, nullptr, nullptr, nm_hints)) return r;
    int cur = 0;
    st.idx_next = w.idx[cur]; st.count_next = cnt + cur;
    HIP_TRY(misti::launch_nm_begin(st, w.llk0, sm));
    // The host runs at most two iterations ahead of the device: before issuing iteration k it waits for the event of
    // iteration k - 2 and reads the count of live starts that iteration left in pinned memory (4 bytes; nothing else of
    // the search leaves the device).  That count bounds the live starts of iteration k from above - the number only falls -
    // and sizes its batches; zero -> stop issuing.
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct EvGuard { hipEvent_t* e; ~EvGuard() { for (int i = 0; i < 2; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } evg{ev};
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    volatile int32_t* live_host = c->nm_live_host;
    live_host[0] = live_host[1] = -1;
    HIP_TRY(hipMemcpyAsync((void*)&live_host[0], cnt + cur, sizeof(int32_t), hipMemcpyDeviceToHost, sm));     // after nm_begin
    HIP_TRY(hipEventRecord(ev[0], sm));
    HIP_TRY(hipEventSynchronize(ev[0]));
    int64_t bound = live_host[0];
    int64_t issued = 0, slots = 0, spec_iters = 0;
    bool spec_primed = false;
    const int64_t K = 4 + N;
    for (int it = 0; bound > 0 && it < maxiter; ++it) {
        const int slot = it & 1;
        if (it >= 2) {
            HIP_TRY(hipEventSynchronize(ev[slot]));
            if (live_host[slot] < bound) bound = live_host[slot];
            if (bound <= 0) break;
        }
        st.idx_cur = w.idx[cur]; st.count_cur = cnt + cur;
        st.idx_next = w.idx[cur ^ 1]; st.count_next = cnt + (cur ^ 1);
        const bool spec = bound <= st.spec_cap;
        if (spec) {
            // speculative iteration: every point SciPy could ask for, one batch, one decision kernel - which also lays out the points
            // of the NEXT iteration and drops the count of live starts into the host's pinned word (nm_spec_step_kernel: the live
            // starts of a speculative iteration fit one workgroup); only the first speculative iteration launches a points kernel
            if (!spec_primed) { HIP_TRY(misti::launch_nm_spec_points(st, bound, sm)); spec_primed = true; }
            if (int r = run_dev(c, bound * K, st.ps_split, st.ps, nullptr, 1, w.d_row, w.llk_spec, nullptr, nullptr, nullptr, nullptr, nm_hints)) return r;
            // (no memsets here: the points step zeroes the next slot counter, and the reflection-split array is only read by the
            //  three-batch path, which a search never returns to - the number of live starts only falls)
            misti::NmState nx = st;
            nx.idx_cur = st.idx_next; nx.count_cur = st.count_next; nx.idx_next = w.idx[cur]; nx.count_next = cnt + cur;
            HIP_TRY(misti::launch_nm_spec_step(st, nx, bound, w.llk_spec, (int32_t*)&live_host[slot], sm));
            ++spec_iters;
        } else {
            if (int r = run_dev(c, bound, st.split1, st.p1, nullptr, 1, w.d_row, w.llk1, nullptr, nullptr, nullptr, nullptr, nm_hints)) return r;
            HIP_TRY(misti::launch_nm_reflect(st, bound, w.llk1, sm));
            if (int r = run_dev(c, bound, st.split2, st.p2, nullptr, 1, w.d_row, w.llk2, nullptr, nullptr, nullptr, nullptr, nm_hints)) return r;
            HIP_TRY(misti::launch_nm_accept(st, bound, w.llk2, sm));
            if (int r = run_dev(c, bound * N, st.split3, st.p3, nullptr, 1, w.d_row, w.llk3, nullptr, nullptr, nullptr, nullptr, nm_hints)) return r;
            HIP_TRY(hipMemsetAsync(cnt + (cur ^ 1), 0, sizeof(int32_t), sm));
            HIP_TRY(hipMemsetAsync(st.split1, 0xBF, (size_t)bound * sizeof(double), sm));
            HIP_TRY(misti::launch_nm_finish(st, bound, w.llk3, sm));
        }
        if (!spec) HIP_TRY(hipMemcpyAsync((void*)&live_host[slot], cnt + (cur ^ 1), sizeof(int32_t), hipMemcpyDeviceToHost, sm));
        HIP_TRY(hipEventRecord(ev[slot], sm));
        cur ^= 1;
        ++issued;
        slots += bound;
    }
    c->nm_iterations += issued;
    c->nm_slots += slots;
    c->nm_spec_iterations += spec_iters;
    // results (device buffers reused: best vertices over the starts)
    HIP_TRY(misti::launch_nm_result(st, w.d_starts, w.d_llh, st.shrunk, sm));
    return 0;
}

}  // namespace

extern "C" {

int misti_nm_solve(misti_ctx* c, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                   double xatol, double fatol, int32_t maxiter,
                   double* x, double* llh, int32_t* nit, int32_t* nfev, int32_t* status) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_start < 0) return fail(MISTI_E_ARG, "negative number of starts");
    if (n_start == 0) return 0;
    const int N = c->dm.n_param;
    if (N < 1) return fail(MISTI_E_ARG, "the model has no optimised parameter");
    if (!starts || !jsfs_row || !x || !llh) return fail(MISTI_E_ARG, "starts / jsfs_row / x / llh is NULL");
    if (maxiter < 1) return fail(MISTI_E_ARG, "maxiter must be >= 1");
    if (n_start > INT32_MAX / (8 * (N + 1))) return fail(MISTI_E_LIMIT, "too many starts for one call");
    HIP_TRY(hipSetDevice(c->device));
    const size_t S = (size_t)n_start;
    NmWork w;
    if (int r = nm_prepare(c, n_start, w)) return r;
    hipStream_t sm = c->stream;
    HIP_TRY(hipMemcpyAsync(w.d_starts, starts, S * N * sizeof(double), hipMemcpyHostToDevice, sm));
    HIP_TRY(hipMemcpyAsync(w.d_row, jsfs_row, 8 * sizeof(double), hipMemcpyHostToDevice, sm));
    c->nm_iterations = c->nm_slots = c->nm_spec_iterations = 0;
    if (int r = nm_run(c, w, split_time, xatol, fatol, maxiter, INT64_MAX)) return r;
    HIP_TRY(hipMemcpyAsync(x, w.d_starts, S * N * sizeof(double), hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipMemcpyAsync(llh, w.d_llh, S * sizeof(double), hipMemcpyDeviceToHost, sm));
    if (nit) HIP_TRY(hipMemcpyAsync(nit, w.st.nit, S * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
    if (nfev) HIP_TRY(hipMemcpyAsync(nfev, w.st.nfev, S * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
    if (status) HIP_TRY(hipMemcpyAsync(status, w.st.shrunk, S * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipStreamSynchronize(sm));
    return 0;
}

int misti_basinhopping(misti_ctx* c, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                       int32_t niter, double T, double stepsize, int32_t interval, double target_accept_rate, double stepwise_factor,
                       double xatol, double fatol, int32_t nm_maxiter, int64_t nm_maxfev, const double* uniforms,
                       double* x, double* llh, int32_t* nfev, int32_t* failures, int32_t* accepted) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (n_start < 0 || niter < 0) return fail(MISTI_E_ARG, "negative number of starts / hops");
    if (n_start == 0) return 0;
    const int N = c->dm.n_param;
    if (N < 1) return fail(MISTI_E_ARG, "the model has no optimised parameter");
    if (!starts || !jsfs_row || !x || !llh || (niter > 0 && !uniforms)) return fail(MISTI_E_ARG, "starts / jsfs_row / uniforms / x / llh is NULL");
    if (nm_maxiter < 1 || nm_maxfev < 1 || interval < 1) return fail(MISTI_E_ARG, "nm_maxiter, nm_maxfev and interval must be >= 1");
    if (n_start > INT32_MAX / (8 * (N + 1))) return fail(MISTI_E_LIMIT, "too many starts for one call");
    HIP_TRY(hipSetDevice(c->device));
    const size_t S = (size_t)n_start, U = S * (size_t)niter * (size_t)(N + 1);
    NmWork w;
    if (int r = nm_prepare(c, n_start, w, S * N * 3 + S * 3 + U, 7 * S)) return r;
    misti::BhState bh{};
    bh.S = n_start; bh.N = N;
    bh.beta = T != 0.0 ? 1.0 / T : INFINITY;
    bh.target = target_accept_rate; bh.factor = stepwise_factor; bh.interval = interval;
    double* d = w.extra_f64;
    bh.x_cur = d; d += S * N; bh.x_best = d; d += S * N;
    double* d_trial = d; d += S * N;
    bh.f_cur = d; d += S; bh.f_best = d; d += S; bh.stepsize = d; d += S;
    double* d_uni = d;
    int32_t* q = w.extra_i32;
    bh.ok_cur = q; q += S; bh.ok_best = q; q += S; bh.nstep = q; q += S; bh.naccept = q; q += S; bh.nfev = q; q += S; bh.failures = q; q += S; bh.accepted = q;
    hipStream_t sm = c->stream;
    HIP_TRY(hipMemcpyAsync(w.d_starts, starts, S * N * sizeof(double), hipMemcpyHostToDevice, sm));
    HIP_TRY(hipMemcpyAsync(w.d_row, jsfs_row, 8 * sizeof(double), hipMemcpyHostToDevice, sm));
    if (U) HIP_TRY(hipMemcpyAsync(d_uni, uniforms, U * sizeof(double), hipMemcpyHostToDevice, sm));
    {
        std::vector<double> st0(S, stepsize);
        HIP_TRY(hipMemcpyAsync(bh.stepsize, st0.data(), S * sizeof(double), hipMemcpyHostToDevice, sm));
        HIP_TRY(hipStreamSynchronize(sm));                  // st0 is a local
    }
    c->nm_iterations = c->nm_slots = c->nm_spec_iterations = 0;
    // BasinHoppingRunner.__init__: the initial minimisation from the start itself
    if (int r = nm_run(c, w, split_time, xatol, fatol, nm_maxiter, nm_maxfev)) return r;
    HIP_TRY(misti::launch_bh_update(bh, w.st, w.d_starts, w.d_llh, w.st.shrunk, -1, niter, d_uni, sm));
    for (int hop = 0; hop < niter; ++hop) {                 // one_cycle, all starts in step
        HIP_TRY(misti::launch_bh_step(bh, hop, niter, d_uni, d_trial, sm));
        HIP_TRY(hipMemcpyAsync(w.d_starts, d_trial, S * N * sizeof(double), hipMemcpyDeviceToDevice, sm));
        if (int r = nm_run(c, w, split_time, xatol, fatol, nm_maxiter, nm_maxfev)) return r;
        HIP_TRY(misti::launch_bh_update(bh, w.st, w.d_starts, w.d_llh, w.st.shrunk, hop, niter, d_uni, sm));
    }
    HIP_TRY(misti::launch_bh_result(bh, w.d_starts, w.d_llh, sm));
    HIP_TRY(hipMemcpyAsync(x, w.d_starts, S * N * sizeof(double), hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipMemcpyAsync(llh, w.d_llh, S * sizeof(double), hipMemcpyDeviceToHost, sm));
    if (nfev) HIP_TRY(hipMemcpyAsync(nfev, bh.nfev, S * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
    if (failures) HIP_TRY(hipMemcpyAsync(failures, bh.failures, S * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
    if (accepted) HIP_TRY(hipMemcpyAsync(accepted, bh.accepted, S * sizeof(int32_t), hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipStreamSynchronize(sm));
    return 0;
}

int misti_nm_last_stats(misti_ctx* c, int64_t stats[2]) {
    if (!c || !stats) return fail(MISTI_E_ARG, "ctx / output is NULL");
    stats[0] = c->nm_iterations;
    stats[1] = c->nm_slots;
    return 0;
}

int misti_nm_last_spec_iterations(misti_ctx* c, int64_t* n) {
    if (!c || !n) return fail(MISTI_E_ARG, "ctx / output is NULL");
    *n = c->nm_spec_iterations;
    return 0;
}

int misti_last_diag(misti_ctx* c, int64_t n_cand, double* max_rate_x_len) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (!max_rate_x_len) return fail(MISTI_E_ARG, "output is NULL");
    if (n_cand != c->diag_n) return fail(MISTI_E_ARG, "the last batch had %lld candidates, not %lld", (long long)c->diag_n, (long long)n_cand);
    if (n_cand == 0) return 0;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(max_rate_x_len, c->ws_diag.p, (size_t)n_cand * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int misti_enable_solver_trace(misti_ctx* c, int on) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    c->trace = on != 0;
    if (!c->trace) { c->trace_n = 0; c->trace_iter_cap = 0; }
    return 0;
}

int misti_last_solver_trace(misti_ctx* c, int64_t n_cand, int32_t* trace, int64_t cand, double* iterates) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    if (!c->trace || c->trace_n == 0) return fail(MISTI_E_ARG, "no solver trace: enable it before the batch (misti_enable_solver_trace)");
    if (n_cand != c->trace_n) return fail(MISTI_E_ARG, "the traced batch had %lld candidates, not %lld", (long long)c->trace_n, (long long)n_cand);
    HIP_TRY(hipSetDevice(c->device));
    const size_t nc = (size_t)n_cand, numT = (size_t)c->dm.numT;
    if (trace) {
        const int32_t* src = c->ws_solver.as<int32_t>() + nc * numT + nc;
        HIP_TRY(hipMemcpyAsync(trace, src, nc * (numT + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    if (iterates) {
        if (c->trace_iter_cap == 0) return fail(MISTI_E_LIMIT, "iterates are recorded for batches of at most %d candidates", MISTI_TRACE_MAX_CAND);
        if (cand < 0 || cand >= n_cand) return fail(MISTI_E_ARG, "candidate %lld out of range", (long long)cand);
        int32_t ch = -1;
        HIP_TRY(hipMemcpyAsync(&ch, c->trace_of + cand, sizeof ch, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (ch < 0 || ch >= c->trace_iter_cap) return fail(MISTI_E_ARG, "candidate %lld has no chain", (long long)cand);
        const size_t per = numT * MISTI_TRACE_MAX_ITER * 2;
        HIP_TRY(hipMemcpyAsync(iterates, c->ws_iters.as<double>() + (size_t)ch * per, per * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int misti_enable_timing(misti_ctx* c, int on) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    c->timing = on != 0;
    return 0;
}

int misti_kernel_times(misti_ctx* c, double ms[3], int64_t launches[3], int reset) {
    if (!c) return fail(MISTI_E_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    for (int w = 0; w < 3; ++w) {
        if (int r = drain_pending(c, w, true)) return r;
        if (ms) ms[w] = c->ms[w];
        if (launches) launches[w] = c->launches[w];
        if (reset) { c->ms[w] = 0; c->launches[w] = 0; }
    }
    return 0;
}

}  // extern "C"

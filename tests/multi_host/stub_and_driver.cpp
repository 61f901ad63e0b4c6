// TEST INFRASTRUCTURE (tests/test_multi_host_cpu.py): misti_amd/csrc/misti_multi.cpp built HOST-ONLY with g++ against this file, which
// stands in for the single-device entry points of misti_api.cpp and for the three HIP runtime calls misti_multi.cpp makes itself, and
// drives the multi-device object the way its callers do - under -fsanitize=thread and -fsanitize=address,undefined.
// What is exercised is exactly the code the GPU cannot help with: the persistent-worker dispatch (job / generation / pending), the chain
// dealing, the per-object call lock, the error hand-over from a worker that fails or throws, create / destroy cycles.
//
// The stand-in "evaluates" a candidate as a pure function of its inputs, so the driver can check every row of a dealt batch against the
// single-context answer (bit for bit, as the real library promises) and that no row was written twice or not at all.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/misti_hip.h"

// ---- stand-ins for misti_api.cpp -----------------------------------------------------------------------------------------------------
struct misti_ctx {
    int device;
    int n_param, n_band, numT;
    std::atomic<int> in_call{0};          // a context is used by one host thread at a time: the stub CHECKS it
    int64_t calls = 0;
};

static thread_local std::string g_err;
static std::atomic<int> g_live_ctx{0};
static std::atomic<int> g_overlap{0};     // times two threads were inside one context at once (must stay 0)
static std::atomic<int> g_fail_device_call{-1};   // >= 0: the n-th indexed evaluation fails with MISTI_E_HIP

extern "C" {

int misti_set_error_(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
const char* misti_last_error(void) { return g_err.c_str(); }
int misti_device_count(void) { return 4; }

int misti_create(const misti_model_t* m, int device, misti_ctx** out) {
    misti_ctx* c = new misti_ctx;
    c->device = device; c->n_param = m->n_param; c->n_band = m->n_band; c->numT = m->numT;
    g_live_ctx.fetch_add(1);
    *out = c;
    return 0;
}
int misti_destroy(misti_ctx* c) { if (c) { g_live_ctx.fetch_sub(1); delete c; } return 0; }
int misti_get_stream(misti_ctx* c, void** s) { *s = c; return 0; }
int misti_sync(misti_ctx*) { return 0; }

static double value_of(double split, const double* par, int P, const int32_t* bb, int B, const double* row) {
    double v = std::sin(split) * 1000.0 + row[0] * 1e-3;
    for (int i = 0; i < P; ++i) v += (i + 1) * par[i];
    for (int i = 0; i < 2 * B; ++i) v += 0.25 * bb[i];
    return v;
}

struct InCall {
    misti_ctx* c;
    explicit InCall(misti_ctx* c_) : c(c_) { if (c->in_call.fetch_add(1) != 0) g_overlap.fetch_add(1); }
    ~InCall() { c->in_call.fetch_sub(1); }
};

int misti_eval_batch_indexed_(misti_ctx* c, int64_t n, const int64_t* idx, const double* split, const double* params, const int32_t* bb,
                              int64_t n_rep, const double* jsfs, double* llk, double* jafs, double* lc, double* pr, int32_t* status) {
    InCall guard(c);
    c->calls += 1;
    int expected = g_fail_device_call.load();
    if (expected >= 0 && g_fail_device_call.compare_exchange_strong(expected, -1)) return misti_set_error_(MISTI_E_HIP, "stub: simulated HIP failure");
    std::this_thread::sleep_for(std::chrono::microseconds(50 + 13 * (c->calls % 7)));    // give the other workers time to overlap
    const int P = c->n_param, B = bb ? c->n_band : 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t r = idx ? idx[i] : i;
        for (int64_t k = 0; k < n_rep; ++k) llk[r * n_rep + k] += value_of(split[r], params ? params + r * P : nullptr, params ? P : 0, bb ? bb + r * 2 * B : nullptr, B, jsfs + 8 * k);
        if (jafs) for (int q = 0; q < 7; ++q) jafs[r * 7 + q] = (double)q + split[r];
        if (status) status[r] += 1;                            // the driver starts from -1: exactly one writer per row leaves 0
        (void)lc; (void)pr;
    }
    return 0;
}
int misti_eval_batch(misti_ctx* c, int64_t n, const double* split, const double* params, const int32_t* bb, int64_t n_rep, const double* jsfs,
                     double* llk, double* jafs, double* lc, double* pr, int32_t* status) {
    return misti_eval_batch_indexed_(c, n, nullptr, split, params, bb, n_rep, jsfs, llk, jafs, lc, pr, status);
}
int misti_eval_batch_dev(misti_ctx* c, int64_t n, const double* split, const double* params, const int32_t* bb, int64_t n_rep, const double* jsfs,
                         double* llk, double*, double*, double*, int32_t* status) {
    InCall guard(c);
    std::this_thread::sleep_for(std::chrono::microseconds(40));
    const int P = c->n_param, B = bb ? c->n_band : 0;
    for (int64_t r = 0; r < n; ++r) {
        for (int64_t k = 0; k < n_rep; ++k) llk[r * n_rep + k] = value_of(split[r], params ? params + r * P : nullptr, params ? P : 0, bb ? bb + r * 2 * B : nullptr, B, jsfs + 8 * k);
        if (status) status[r] = 0;
    }
    return 0;
}
int misti_nm_solve(misti_ctx* c, int64_t n_start, const double* starts, double, const double*, double, double, int32_t,
                   double* x, double* llh, int32_t* nit, int32_t* nfev, int32_t* status) {
    InCall guard(c);
    for (int64_t s = 0; s < n_start; ++s) {
        for (int i = 0; i < c->n_param; ++i) x[s * c->n_param + i] = 2.0 * starts[s * c->n_param + i];
        llh[s] = -starts[s * c->n_param];
        if (nit) nit[s] = 7;
        if (nfev) nfev[s] = 11;
        if (status) status[s] = 0;
    }
    return 0;
}
int misti_basinhopping(misti_ctx* c, int64_t n_start, const double* starts, double, const double*, int32_t niter, double, double, int32_t, double, double,
                       double, double, int32_t, int64_t, const double* uniforms, double* x, double* llh, int32_t* nfev, int32_t* failures, int32_t* accepted) {
    InCall guard(c);
    const int N = c->n_param;
    for (int64_t s = 0; s < n_start; ++s) {
        double u = 0;
        for (int64_t k = 0; k < (int64_t)niter * (N + 1); ++k) u += uniforms[s * niter * (N + 1) + k];
        for (int i = 0; i < N; ++i) x[s * N + i] = starts[s * N + i] + u;
        llh[s] = u;
        if (nfev) nfev[s] = 1;
        if (failures) failures[s] = 0;
        if (accepted) accepted[s] = 1;
    }
    return 0;
}

// ---- the three HIP runtime calls misti_multi.cpp makes itself (device-resident form only) ---------------------------------------------
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }

int misti_multi_test_throw_in_worker_(misti_multi* m, int d);

}  // extern "C"

// ---- the driver ------------------------------------------------------------------------------------------------------------------------
static int g_bad = 0;
#define CHECK(cond, ...)                                                   \
    do {                                                                   \
        if (!(cond)) { ++g_bad; std::fprintf(stderr, "CHECK failed at line %d: %s : ", __LINE__, #cond); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } \
    } while (0)

struct Batch {
    int64_t n;
    int P, B;
    std::vector<double> split, params, rows;
    std::vector<int32_t> bounds;
    int64_t n_rep;
};

static Batch make_batch(int64_t n, int P, int B, int64_t n_rep, unsigned seed, int n_chains) {
    Batch b;
    b.n = n; b.P = P; b.B = B; b.n_rep = n_rep;
    b.split.resize(n); b.params.resize((size_t)n * P); b.bounds.resize((size_t)n * 2 * B); b.rows.resize((size_t)n_rep * 8);
    unsigned s = seed * 2654435761u + 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) & 0xffff; };
    for (int64_t i = 0; i < n; ++i) {
        const int chain = n_chains > 0 ? (int)(rnd() % (unsigned)n_chains) : (int)i;
        b.split[i] = 2.0 + (double)(rnd() % 20) + ((rnd() & 3) == 0 ? 0.5 : 0.0);
        for (int p = 0; p < P; ++p) b.params[i * P + p] = 0.001 * (chain * 7 + p + 1);
        for (int q = 0; q < 2 * B; ++q) b.bounds[i * 2 * B + q] = (chain + q) % 5;
    }
    for (int64_t k = 0; k < n_rep * 8; ++k) b.rows[k] = 100.0 + (double)(rnd() % 1000);
    return b;
}

static void expect(const Batch& b, std::vector<double>& llk) {
    llk.assign((size_t)b.n * b.n_rep, 0.0);
    for (int64_t r = 0; r < b.n; ++r)
        for (int64_t k = 0; k < b.n_rep; ++k)
            llk[r * b.n_rep + k] = value_of(b.split[r], b.P ? &b.params[r * b.P] : nullptr, b.P, b.B ? &b.bounds[r * 2 * b.B] : nullptr, b.B, &b.rows[8 * k]);
}

static int run_batch(misti_multi* m, const Batch& b, bool want_ok = true, bool alone = true) {
    std::vector<double> llk((size_t)b.n * b.n_rep, 0.0), jafs((size_t)b.n * 7, -1.0), want;
    std::vector<int32_t> status(b.n, -1);
    const int r = misti_multi_eval_batch(m, b.n, b.split.data(), b.P ? b.params.data() : nullptr, b.B ? b.bounds.data() : nullptr, b.n_rep, b.rows.data(),
                                         llk.data(), jafs.data(), nullptr, nullptr, status.data());
    if (!want_ok) return r;
    CHECK(r == 0, "misti_multi_eval_batch: %d %s", r, misti_last_error());
    expect(b, want);
    int64_t wrong = 0, not_once = 0;
    for (size_t i = 0; i < want.size(); ++i) if (std::memcmp(&llk[i], &want[i], 8) != 0) ++wrong;
    for (int64_t i = 0; i < b.n; ++i) if (status[i] != 0) ++not_once;
    CHECK(wrong == 0, "%lld values differ from the single-context answer", (long long)wrong);
    CHECK(not_once == 0, "%lld rows were not written exactly once", (long long)not_once);
    if (!alone) return r;                 // "the last call's shards" means nothing while another thread calls into the same object
    // whole chains per context: the shard sizes add up
    const int D = misti_multi_size(m);
    std::vector<int64_t> nc(D), nch(D);
    CHECK(misti_multi_last_shards(m, nc.data(), nch.data()) == 0, "last_shards");
    int64_t total = 0;
    for (int d = 0; d < D; ++d) total += nc[d];
    CHECK(total == b.n, "shards hold %lld of %lld candidates", (long long)total, (long long)b.n);
    return r;
}

int main(int argc, char** argv) {
    const int cycles = argc > 1 ? std::atoi(argv[1]) : 200;
    std::vector<double> times(15, 0.1), lh(32, 1.0);
    misti_band_t bands[2] = {{0, 0, -1, 0, 0.0}, {1, 0, -1, 1, 0.0}};
    misti_model_t model{};
    model.numT = 16; model.sample_date = 0; model.flags = MISTI_CPFIT; model.n_band = 2; model.n_pulse = 0; model.n_param = 2;
    model.mixture_th = 0.0; model.times = times.data(); model.lh = lh.data(); model.bands = bands; model.pulses = nullptr;

    // 1. create / evaluate / destroy cycles on {0, 0, 0}: ragged shards, fewer chains than contexts (empty shards), one candidate, none
    for (int i = 0; i < cycles; ++i) {
        const int devs[3] = {0, 0, 0};
        misti_multi* m = nullptr;
        CHECK(misti_create_multi(&model, 3, devs, &m) == 0 && m, "create: %s", misti_last_error());
        if (!m) return 2;
        run_batch(m, make_batch(37 + i % 11, 2, 2, 3, (unsigned)i, 5));
        if (i % 4 == 0) run_batch(m, make_batch(64, 2, 0, 1, (unsigned)i + 1000, 2));       // two chains on three contexts: one shard empty
        if (i % 5 == 0) run_batch(m, make_batch(1, 2, 2, 2, (unsigned)i + 2000, 1));
        if (i % 7 == 0) { Batch e = make_batch(1, 2, 2, 2, 1, 1); e.n = 0; CHECK(run_batch(m, e, false) == 0, "an empty batch is not an error"); }
        CHECK(misti_destroy_multi(m) == 0, "destroy");
    }
    CHECK(g_live_ctx.load() == 0, "%d contexts leaked", g_live_ctx.load());

    // 2. one object, many calls; a worker that throws, a worker whose call fails: the call fails with that context's message, the next one works
    {
        const int devs[4] = {0, 1, 2, 3};
        misti_multi* m = nullptr;
        CHECK(misti_create_multi(&model, 4, devs, &m) == 0, "create");
        for (int i = 0; i < cycles; ++i) run_batch(m, make_batch(50 + i % 13, 2, 2, 2, (unsigned)i + 3000, 9));
        for (int d = 0; d < 4; ++d) {
            CHECK(misti_multi_test_throw_in_worker_(m, d) == 0, "hook");
            const int r = run_batch(m, make_batch(40, 2, 2, 1, 7, 0), false);
            CHECK(r == MISTI_E_ARG && std::strstr(misti_last_error(), "misti_multi_test_throw_in_worker_"), "throw in worker %d: %d %s", d, r, misti_last_error());
            CHECK(misti_multi_test_throw_in_worker_(m, -1) == 0, "hook off");
            run_batch(m, make_batch(40, 2, 2, 1, 7, 0));
        }
        g_fail_device_call.store(0);
        const int r = run_batch(m, make_batch(40, 2, 2, 1, 8, 0), false);
        CHECK(r == MISTI_E_HIP && std::strstr(misti_last_error(), "simulated HIP failure"), "failing context: %d %s", r, misti_last_error());
        run_batch(m, make_batch(40, 2, 2, 1, 8, 0));

        // 3. two caller threads on ONE object: serialised by the per-object call lock (include/misti_hip.h), every result still right
        std::thread other([&] { for (int i = 0; i < cycles; ++i) run_batch(m, make_batch(33 + i % 5, 2, 2, 2, (unsigned)i + 5000, 6), true, false); });
        for (int i = 0; i < cycles; ++i) run_batch(m, make_batch(41 + i % 3, 2, 0, 1, (unsigned)i + 6000, 4), true, false);
        other.join();
        CHECK(g_overlap.load() == 0, "%d times two threads were inside one context at once", g_overlap.load());

        // 4. blocks of starts (Nelder-Mead / basin hopping): every start exactly once, in order
        const int64_t S = 101;
        std::vector<double> starts(S * 2), x(S * 2), llh(S), row(8, 1.0), uni(S * 3 * 3, 0.5);
        std::vector<int32_t> nit(S), nfev(S), st(S), fl(S), acc(S);
        for (int64_t s = 0; s < S; ++s) { starts[2 * s] = (double)s; starts[2 * s + 1] = 0.5 * s; }
        CHECK(misti_multi_nm_solve(m, S, starts.data(), 10.0, row.data(), 1e-4, 1e-4, 100, x.data(), llh.data(), nit.data(), nfev.data(), st.data()) == 0, "nm_solve");
        for (int64_t s = 0; s < S; ++s) CHECK(x[2 * s] == 2.0 * s && llh[s] == -(double)s && nit[s] == 7, "nm start %lld", (long long)s);
        CHECK(misti_multi_basinhopping(m, S, starts.data(), 10.0, row.data(), 3, 0.5, 0.5, 50, 0.5, 0.9, 1e-4, 1e-4, 400, 400, uni.data(), x.data(), llh.data(),
                                       nfev.data(), fl.data(), acc.data()) == 0, "basinhopping");
        for (int64_t s = 0; s < S; ++s) CHECK(llh[s] == 4.5 && x[2 * s] == (double)s + 4.5, "bh start %lld", (long long)s);
        CHECK(misti_destroy_multi(m) == 0, "destroy");
    }
    // 5. the gathered device-resident form with D = 3 contexts on {0, 0, 0} through the RCCL double (argv[2]: tests/multi_host/fake_rccl.cpp built
    //    with -DFAKE_RCCL_HOST; "device memory" is host memory here): ragged and empty shards, NaN / -1 padding, every context's table complete
    if (argc > 2) {
        setenv("MISTI_RCCL_LIB", argv[2], 1);
        const int devs[3] = {0, 0, 0};
        misti_multi* m = nullptr;
        CHECK(misti_create_multi(&model, 3, devs, &m) == 0, "create");
        const int64_t rows = 6, R = 2;
        for (int rep = 0; rep < cycles; ++rep) {
            const int64_t n_cand[3] = {(rep % 7), (rep % 3 == 0) ? 0 : 6, 1 + rep % 5};
            std::vector<Batch> shard;
            for (int d = 0; d < 3; ++d) shard.push_back(make_batch(n_cand[d] ? n_cand[d] : 1, 2, 2, R, (unsigned)(rep * 3 + d + 9000), 0));
            for (int d = 1; d < 3; ++d) shard[d].rows = shard[0].rows;                 // the replicate table is the same on every device
            std::vector<std::vector<double>> table(3, std::vector<double>(3 * rows * R, 12345.0));
            std::vector<std::vector<int32_t>> stat(3, std::vector<int32_t>(3 * rows, 77));
            const double* sp[3]; const double* pa[3]; const int32_t* bb[3]; const double* js[3]; double* tl[3]; int32_t* ts[3];
            for (int d = 0; d < 3; ++d) { sp[d] = shard[d].split.data(); pa[d] = shard[d].params.data(); bb[d] = shard[d].bounds.data(); js[d] = shard[d].rows.data(); tl[d] = table[d].data(); ts[d] = stat[d].data(); }
            const int r = misti_multi_eval_batch_dev(m, n_cand, rows, sp, pa, bb, R, js, tl, ts);
            CHECK(r == 0, "misti_multi_eval_batch_dev: %d %s", r, misti_last_error());
            CHECK(misti_multi_sync(m) == 0, "sync");
            int64_t wrong = 0;
            for (int dev = 0; dev < 3; ++dev)
                for (int d = 0; d < 3; ++d) {
                    std::vector<double> want;
                    Batch b = shard[d]; b.n = n_cand[d];
                    expect(b, want);
                    for (int64_t i = 0; i < rows; ++i) {
                        for (int64_t k = 0; k < R; ++k) {
                            const double got = table[dev][(d * rows + i) * R + k];
                            if (i < n_cand[d]) { if (std::memcmp(&got, &want[i * R + k], 8) != 0) ++wrong; }
                            else { uint64_t bits; std::memcpy(&bits, &got, 8); if (bits != ~0ull) ++wrong; }
                        }
                        if (stat[dev][d * rows + i] != (i < n_cand[d] ? 0 : -1)) ++wrong;
                    }
                }
            CHECK(wrong == 0, "gathered tables: %lld wrong entries (repeat %d)", (long long)wrong, rep);
            if (rep == 3) {
                CHECK(misti_multi_test_throw_in_worker_(m, 2) == 0, "hook");
                const int q = misti_multi_eval_batch_dev(m, n_cand, rows, sp, pa, bb, R, js, tl, ts);
                CHECK(q == MISTI_E_ARG && std::strstr(misti_last_error(), "context 2 of 3"), "throw in the gathered form: %d %s", q, misti_last_error());
                CHECK(misti_multi_test_throw_in_worker_(m, -1) == 0, "hook off");
            }
        }
        CHECK(misti_destroy_multi(m) == 0, "destroy");
    }
    CHECK(g_live_ctx.load() == 0, "%d contexts leaked", g_live_ctx.load());
    std::printf("multi host driver: %d cycles, %d failed checks\n", cycles, g_bad);
    return g_bad ? 1 : 0;
}

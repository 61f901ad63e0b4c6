// TEST INFRASTRUCTURE (tests/test_multi_host_cpu.py): misti_amd/csrc/misti_lanes.cpp built HOST-ONLY with g++ against stand-ins for the single-context
// entry points and for the HIP event calls it makes, driven the way a C caller drives it, under AddressSanitizer + UBSan.  What is checked is the pool's own
// logic: round-robin and "an idle lane first" under MISTI_LANE_ANY, lane bounds, the borrowed contexts, failure of a context in the middle of creation
// (everything already created is released), hints reaching every lane, create / destroy cycles without a leak.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/misti_hip.h"

struct misti_ctx { int id; int batches = 0; uint32_t hints = 0; int busy_for = 0; };     // busy_for: event queries that still answer "not ready"
struct FakeEvent { misti_ctx* owner = nullptr; };

static std::string g_err;
static int g_live_ctx = 0, g_live_ev = 0, g_fail_create_at = -1, g_created = 0;

extern "C" {
int misti_set_error_(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
const char* misti_last_error(void) { return g_err.c_str(); }
int misti_create(const misti_model_t*, int, misti_ctx** out) {
    if (g_created == g_fail_create_at) { ++g_created; return misti_set_error_(MISTI_E_HIP, "stub: create failed"); }
    misti_ctx* c = new misti_ctx;
    c->id = g_created++;
    ++g_live_ctx;
    *out = c;
    return 0;
}
int misti_destroy(misti_ctx* c) { if (c) { --g_live_ctx; delete c; } return 0; }
int misti_get_stream(misti_ctx* c, void** s) { *s = c; return 0; }        // the "stream" is the context: the event stub finds its owner through it
int misti_sync(misti_ctx* c) { c->busy_for = 0; return 0; }
int misti_set_hints(misti_ctx* c, uint32_t h) { c->hints = h; return 0; }
int misti_eval_batch_dev(misti_ctx* c, int64_t n, const double*, const double*, const int32_t*, int64_t, const double*, double* llk, double*, double*, double*, int32_t*) {
    if (n < 0) return misti_set_error_(MISTI_E_ARG, "negative batch size");
    c->batches += 1;
    c->busy_for = 3;                       // the next three queries of this lane's event say "not ready"
    if (llk) llk[0] = (double)c->id;
    return 0;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new FakeEvent); ++g_live_ev; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<FakeEvent*>(e); --g_live_ev; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { reinterpret_cast<FakeEvent*>(e)->owner = reinterpret_cast<misti_ctx*>(s); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) {
    misti_ctx* c = reinterpret_cast<FakeEvent*>(e)->owner;
    if (c && c->busy_for > 0) { c->busy_for -= 1; return hipErrorNotReady; }
    return hipSuccess;
}
// waiting for a lane's event (misti_lanes_wait / _sync wait for it before the stream): the batch it was recorded behind is then complete
hipError_t hipEventSynchronize(hipEvent_t e) {
    misti_ctx* c = reinterpret_cast<FakeEvent*>(e)->owner;
    if (c) c->busy_for = 0;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub error"; }
}

static int g_bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++g_bad; std::fprintf(stderr, "CHECK failed at line %d: %s : ", __LINE__, #cond); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } } while (0)

int main() {
    misti_model_t model;
    std::memset(&model, 0, sizeof model);
    for (int cycle = 0; cycle < 50; ++cycle) {
        misti_lanes* L = nullptr;
        const int n = 1 + cycle % 7;
        CHECK(misti_create_lanes(&model, 0, n, &L) == 0 && L, "create: %s", misti_last_error());
        CHECK(misti_lanes_size(L) == n, "size");
        CHECK(misti_lanes_set_hints(L, MISTI_HINT_INTEGER_SPLITS) == 0, "hints");
        for (int i = 0; i < n; ++i) { misti_ctx* c = nullptr; CHECK(misti_lanes_context(L, i, &c) == 0 && c && c->hints == MISTI_HINT_INTEGER_SPLITS, "context %d", i); }
        misti_ctx* none = nullptr;
        CHECK(misti_lanes_context(L, n, &none) == MISTI_E_ARG, "context out of range");
        double out = -1.0;
        int used = -1;
        // explicit lanes
        for (int k = 0; k < 3 * n; ++k) {
            CHECK(misti_lanes_eval_batch_dev(L, k % n, 1, nullptr, nullptr, nullptr, 0, nullptr, &out, nullptr, nullptr, nullptr, nullptr, &used) == 0 && used == k % n, "explicit lane %d", k % n);
        }
        CHECK(misti_lanes_eval_batch_dev(L, n, 1, nullptr, nullptr, nullptr, 0, nullptr, &out, nullptr, nullptr, nullptr, nullptr, &used) == MISTI_E_ARG, "lane out of range");
        CHECK(misti_lanes_eval_batch_dev(L, -2, 1, nullptr, nullptr, nullptr, 0, nullptr, &out, nullptr, nullptr, nullptr, nullptr, &used) == MISTI_E_ARG, "lane -2");
        CHECK(misti_lanes_sync(L) == 0, "sync");
        for (int i = 0; i < n; ++i) CHECK(misti_lanes_busy(L, i) == 0, "idle after sync");
        // MISTI_LANE_ANY: nothing in flight -> lanes in round-robin order from the current position; a busy lane is skipped while an idle one exists
        std::vector<int> seen;
        for (int k = 0; k < n; ++k) {
            CHECK(misti_lanes_eval_batch_dev(L, MISTI_LANE_ANY, 1, nullptr, nullptr, nullptr, 0, nullptr, &out, nullptr, nullptr, nullptr, nullptr, &used) == 0, "any");
            for (int s : seen) CHECK(s != used, "an idle lane exists but busy lane %d was taken again", used);
            seen.push_back(used);
        }
        // every lane is busy now: the next one is the round-robin lane, and the call still succeeds
        CHECK(misti_lanes_eval_batch_dev(L, MISTI_LANE_ANY, 1, nullptr, nullptr, nullptr, 0, nullptr, &out, nullptr, nullptr, nullptr, nullptr, &used) == 0 && used >= 0 && used < n, "any, all busy");
        CHECK(misti_lanes_wait(L, used) == 0 && misti_lanes_busy(L, used) == 0, "wait");
        // a failing batch reports the context's error and records nothing
        CHECK(misti_lanes_eval_batch_dev(L, 0, -1, nullptr, nullptr, nullptr, 0, nullptr, &out, nullptr, nullptr, nullptr, nullptr, &used) == MISTI_E_ARG, "failing batch");
        CHECK(misti_destroy_lanes(L) == 0, "destroy");
        CHECK(g_live_ctx == 0 && g_live_ev == 0, "leak: %d contexts, %d events", g_live_ctx, g_live_ev);
    }
    // creation fails at the fourth context: the three already made (and their events) are released, the message is the context's
    g_fail_create_at = g_created + 3;
    misti_lanes* L = reinterpret_cast<misti_lanes*>(1);
    CHECK(misti_create_lanes(&model, 0, 6, &L) == MISTI_E_HIP && L == nullptr && std::strstr(misti_last_error(), "create failed"), "failed creation: %s", misti_last_error());
    CHECK(g_live_ctx == 0 && g_live_ev == 0, "leak after a failed creation: %d contexts, %d events", g_live_ctx, g_live_ev);
    CHECK(misti_create_lanes(&model, 0, 0, &L) == MISTI_E_LIMIT && misti_create_lanes(&model, 0, MISTI_MAX_LANES + 1, &L) == MISTI_E_LIMIT, "limits");
    CHECK(misti_destroy_lanes(nullptr) == 0 && misti_lanes_size(nullptr) == 0, "NULL object");
    std::printf("lanes host driver: %d failed checks\n", g_bad);
    return g_bad ? 1 : 0;
}

/* The overlapped rate of the headline benchmark from plain C (ABI 6): misti_create_lanes + misti_lanes_eval_batch_dev + misti_lanes_sync.
 * One batch of the 4 096-point grid is latency-bound (its longest lambda-correction chain: 1.4 ms with 64 of 1 024 SIMDs busy); twenty
 * batches in flight on twenty lanes - contexts with a stream each, inside the library - reach the rate bench.py reports, and every one of
 * them returns the bits of a single context's misti_eval_batch.  The reference's counterpart is one MigrationInference object per process
 * and as many processes as cores (/root/reference/MiSTI.py:213-214 under `parallel -j 20`, README.md:110-115).
 *
 *   python -c "from misti_amd import workloads; workloads.dump_text('config2', '/tmp/config2.txt')"     # the grid as a text file
 *   gcc -std=c99 -O2 -I include examples/lanes_throughput.c -L misti_amd/csrc -lmisti_hip -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/misti_amd/csrc -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/lanes_throughput
 *   /tmp/lanes_throughput /tmp/config2.txt [lanes = 20] [steps = 400]
 *
 * File format (whitespace separated; doubles with 17 significant digits): numT sample_date flags n_band n_pulse n_param mixture_th |
 * times[numT-1] | lh[numT][2] | bands: pop start end param value | pulses: pop time param value | n_cand n_rep | split[n_cand] |
 * params[n_cand][n_param] | jsfs[n_rep][8].
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "misti_hip.h"

/* hip_runtime_api.h, the calls used (hipError_t 0 = success; hipMemcpyKind: 1 host to device, 2 device to host) */
extern int hipSetDevice(int device);
extern int hipMalloc(void** ptr, size_t bytes);
extern int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
extern int hipFree(void* ptr);

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int rd(FILE* f, double* v) { return fscanf(f, "%lf", v) == 1; }
static int ri(FILE* f, int* v) { return fscanf(f, "%d", v) == 1; }

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)
#define CHECK(call) do { int r_ = (call); if (r_ != 0) { fprintf(stderr, "%s: %d %s\n", #call, r_, misti_last_error()); return r_ == MISTI_E_NODEV ? 2 : 1; } } while (0)

int main(int argc, char** argv) {
    misti_model_t m;
    misti_band_t bands[8];
    misti_pulse_t pulses[8];
    double *times, *lh, *split, *params, *jsfs, *llk_one, *llk_lane;
    int32_t *st_one, *st_lane;
    int n_cand, n_rep, i, k, flags, whole = 1, identical = 1, n_lanes, steps;
    FILE* f;
    misti_ctx* one = NULL;
    misti_lanes* L = NULL;
    void *d_split = NULL, *d_par = NULL, *d_jsfs = NULL;
    void **d_llk, **d_jafs, **d_st;
    double t0, dt, finite = 0;

    if (argc < 2) DIE("usage: %s WORKLOAD.txt [lanes] [steps]", argv[0]);
    n_lanes = argc > 2 ? atoi(argv[2]) : 20;
    steps = argc > 3 ? atoi(argv[3]) : 400;
    if (misti_device_count() <= 0) { fprintf(stderr, "no HIP device: %s\n", misti_last_error()); return 2; }
    f = fopen(argv[1], "r");
    if (!f) DIE("cannot open %s", argv[1]);
    if (!ri(f, &m.numT) || !ri(f, &m.sample_date) || !ri(f, &flags) || !ri(f, &m.n_band) || !ri(f, &m.n_pulse) || !ri(f, &m.n_param) || !rd(f, &m.mixture_th)) DIE("bad header");
    m.flags = (uint32_t)flags;
    if (m.numT < 2 || m.numT > 255 || m.n_band > 8 || m.n_pulse > 8 || m.n_band < 0 || m.n_pulse < 0) DIE("model out of range");
    times = (double*)malloc(sizeof(double) * (size_t)(m.numT - 1));
    lh = (double*)malloc(sizeof(double) * 2 * (size_t)m.numT);
    for (i = 0; i < m.numT - 1; ++i) if (!rd(f, &times[i])) DIE("bad times");
    for (i = 0; i < 2 * m.numT; ++i) if (!rd(f, &lh[i])) DIE("bad lh");
    for (i = 0; i < m.n_band; ++i)
        if (!ri(f, &bands[i].pop) || !ri(f, &bands[i].start) || !ri(f, &bands[i].end) || !ri(f, &bands[i].param) || !rd(f, &bands[i].value)) DIE("bad band");
    for (i = 0; i < m.n_pulse; ++i) {
        pulses[i]._pad = 0;
        if (!ri(f, &pulses[i].pop) || !ri(f, &pulses[i].time) || !ri(f, &pulses[i].param) || !rd(f, &pulses[i].value)) DIE("bad pulse");
    }
    m.times = times; m.lh = lh; m.bands = bands; m.pulses = pulses;
    if (!ri(f, &n_cand) || !ri(f, &n_rep) || n_cand < 1 || n_rep < 1) DIE("bad sizes");
    split = (double*)malloc(sizeof(double) * (size_t)n_cand);
    params = (double*)malloc(sizeof(double) * (size_t)n_cand * (size_t)(m.n_param > 0 ? m.n_param : 1));
    jsfs = (double*)malloc(sizeof(double) * 8 * (size_t)n_rep);
    for (i = 0; i < n_cand; ++i) { if (!rd(f, &split[i])) DIE("bad split"); if (split[i] != floor(split[i])) whole = 0; }
    for (i = 0; i < n_cand * m.n_param; ++i) if (!rd(f, &params[i])) DIE("bad params");
    for (i = 0; i < 8 * n_rep; ++i) if (!rd(f, &jsfs[i])) DIE("bad jsfs");
    fclose(f);

    /* one context, host buffers: what every lane's batch must reproduce bit for bit */
    llk_one = (double*)malloc(sizeof(double) * (size_t)n_cand * (size_t)n_rep);
    llk_lane = (double*)malloc(sizeof(double) * (size_t)n_cand * (size_t)n_rep);
    st_one = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_cand);
    st_lane = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_cand);
    CHECK(misti_create(&m, 0, &one));
    CHECK(misti_eval_batch(one, n_cand, split, m.n_param ? params : NULL, NULL, n_rep, jsfs, llk_one, NULL, NULL, NULL, st_one));
    CHECK(misti_destroy(one));
    for (i = 0; i < n_cand * n_rep; ++i) if (isfinite(llk_one[i])) finite += 1;

    /* the lanes: inputs resident once, one set of output buffers per lane */
    CHECK(misti_create_lanes(&m, 0, n_lanes, &L));
    if (whole) CHECK(misti_lanes_set_hints(L, MISTI_HINT_INTEGER_SPLITS));
    if (hipSetDevice(0) != 0) DIE("hipSetDevice");
    if (hipMalloc(&d_split, sizeof(double) * (size_t)n_cand) != 0 || hipMemcpy(d_split, split, sizeof(double) * (size_t)n_cand, 1) != 0) DIE("device memory");
    if (m.n_param && (hipMalloc(&d_par, sizeof(double) * (size_t)n_cand * (size_t)m.n_param) != 0 ||
                      hipMemcpy(d_par, params, sizeof(double) * (size_t)n_cand * (size_t)m.n_param, 1) != 0)) DIE("device memory");
    if (hipMalloc(&d_jsfs, sizeof(double) * 8 * (size_t)n_rep) != 0 || hipMemcpy(d_jsfs, jsfs, sizeof(double) * 8 * (size_t)n_rep, 1) != 0) DIE("device memory");
    d_llk = (void**)calloc((size_t)n_lanes, sizeof(void*));
    d_jafs = (void**)calloc((size_t)n_lanes, sizeof(void*));
    d_st = (void**)calloc((size_t)n_lanes, sizeof(void*));
    for (k = 0; k < n_lanes; ++k)
        if (hipMalloc(&d_llk[k], sizeof(double) * (size_t)n_cand * (size_t)n_rep) != 0 || hipMalloc(&d_jafs[k], sizeof(double) * 7 * (size_t)n_cand) != 0 ||
            hipMalloc(&d_st[k], sizeof(int32_t) * (size_t)n_cand) != 0) DIE("device memory");
    /* a context's first batches allocate its workspaces and learn the launch shape: twice round the lanes, untimed */
    for (k = 0; k < 2 * n_lanes; ++k)
        CHECK(misti_lanes_eval_batch_dev(L, k % n_lanes, n_cand, (const double*)d_split, (const double*)d_par, NULL, n_rep, (const double*)d_jsfs,
                                         (double*)d_llk[k % n_lanes], (double*)d_jafs[k % n_lanes], NULL, NULL, (int32_t*)d_st[k % n_lanes], NULL));
    CHECK(misti_lanes_sync(L));
    t0 = now();
    for (k = 0; k < steps; ++k)
        CHECK(misti_lanes_eval_batch_dev(L, k % n_lanes, n_cand, (const double*)d_split, (const double*)d_par, NULL, n_rep, (const double*)d_jsfs,
                                         (double*)d_llk[k % n_lanes], (double*)d_jafs[k % n_lanes], NULL, NULL, (int32_t*)d_st[k % n_lanes], NULL));
    CHECK(misti_lanes_sync(L));
    dt = now() - t0;
    for (k = 0; k < n_lanes; ++k) {
        if (hipMemcpy(llk_lane, d_llk[k], sizeof(double) * (size_t)n_cand * (size_t)n_rep, 2) != 0 || hipMemcpy(st_lane, d_st[k], sizeof(int32_t) * (size_t)n_cand, 2) != 0) DIE("copy back");
        if (memcmp(llk_lane, llk_one, sizeof(double) * (size_t)n_cand * (size_t)n_rep) != 0 || memcmp(st_lane, st_one, sizeof(int32_t) * (size_t)n_cand) != 0) identical = 0;
    }
    /* MISTI_LANE_ANY: the library picks an idle lane (none is busy now: the round-robin position) */
    { int used = -1;
      CHECK(misti_lanes_eval_batch_dev(L, MISTI_LANE_ANY, n_cand, (const double*)d_split, (const double*)d_par, NULL, n_rep, (const double*)d_jsfs,
                                       (double*)d_llk[0], (double*)d_jafs[0], NULL, NULL, (int32_t*)d_st[0], &used));
      CHECK(misti_lanes_wait(L, used));
      printf("lane_any = %d\n", used); }
    printf("candidates = %d\nreplicates = %d\nlanes = %d\nsteps = %d\n", n_cand, n_rep, misti_lanes_size(L), steps);
    printf("finite = %.0f\n", finite);
    printf("ms_per_step = %.4f\n", 1e3 * dt / steps);
    printf("evals_per_s = %.4g\n", (double)n_cand * (double)n_rep * (double)steps / dt);
    printf("identical = %d\n", identical);
    CHECK(misti_destroy_lanes(L));
    for (k = 0; k < n_lanes; ++k) { hipFree(d_llk[k]); hipFree(d_jafs[k]); hipFree(d_st[k]); }
    hipFree(d_split); if (d_par) hipFree(d_par); hipFree(d_jsfs);
    return identical ? 0 : 1;
}

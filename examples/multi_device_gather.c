/* The device-resident multi-device form of the C ABI without Python (ABI 5): every listed device evaluates ITS shard of the candidates
 * from device memory and the log-likelihoods are gathered on the devices by RCCL inside the library - misti_multi_eval_batch_dev +
 * misti_multi_sync - so that every device ends up with the whole table; what the reference does by concatenating the stdout of its
 * processes (`parallel -j N ./MiSTI.py ... >> res.out`, /root/reference/README.md:110-115).  SURVEY appendix-A anchor A3's model, 12
 * candidates in 4 chains, dealt to the devices of the command line in contiguous blocks (default: device 0 alone; on a node:
 * ./multi_device_gather 0 1 2 3 4 5 6 7), checked against one context's misti_eval_batch bit for bit.
 * Device memory through the HIP runtime's C entry points (declared here: the HIP headers are C++).
 *
 *   gcc -std=c99 -I include examples/multi_device_gather.c -L misti_amd/csrc -lmisti_hip -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/misti_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/multi_device_gather && /tmp/multi_device_gather 0
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "misti_hip.h"

/* hip_runtime_api.h, the four calls used (hipError_t 0 = success; hipMemcpyKind: 1 host to device, 2 device to host) */
extern int hipSetDevice(int device);
extern int hipMalloc(void** ptr, size_t bytes);
extern int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);
extern int hipFree(void* ptr);

#define N_SPLIT 3
#define N_VEC 4
#define N_CAND (N_SPLIT * N_VEC)
#define MAX_DEV 8

static void* to_device(int dev, const void* host, size_t bytes) {
    void* d = NULL;
    if (hipSetDevice(dev) != 0 || hipMalloc(&d, bytes ? bytes : 8) != 0) return NULL;
    if (host && bytes && hipMemcpy(d, host, bytes, 1) != 0) return NULL;
    return d;
}

int main(int argc, char** argv) {
    const double times[7] = {0.01, 0.02, 0.04, 0.08, 0.16, 0.32, 0.64};
    const double lh[8][2] = {{1, 2}, {1, 2}, {0.8, 1.5}, {0.8, 1.5}, {1.2, 1.0}, {1.2, 1.0}, {0.9, 0.9}, {0.7, 0.7}};
    const double jsfs[8] = {100000, 900, 250, 1000, 600, 400, 260, 410};
    const misti_band_t bands[2] = {{0, 1, -1, 0, 0.3}, {1, 1, -1, 1, 0.1}};
    const double vecs[N_VEC][2] = {{0.3, 0.1}, {0.05, 0.7}, {1.5, 0.02}, {0.2, 0.2}};
    const double splits[N_SPLIT] = {5.0, 4.0, 6.0};
    misti_model_t m;
    int devices[MAX_DEV], n_dev = 0, i, c, ok = 1;
    double split[N_CAND], params[N_CAND][2], llk_one[N_CAND];
    int32_t st_one[N_CAND];
    int64_t n[MAX_DEV], lo[MAX_DEV + 1], per = 0;
    const double *d_split[MAX_DEV], *d_par[MAX_DEV], *d_jsfs[MAX_DEV];
    double* d_all[MAX_DEV];
    int32_t* d_st[MAX_DEV];
    misti_multi* mm = NULL;
    misti_ctx* one = NULL;

    m.numT = 8; m.sample_date = 0; m.flags = MISTI_CPFIT | MISTI_SMOOTH | MISTI_UNFOLDED;
    m.n_band = 2; m.n_pulse = 0; m.n_param = 2; m.mixture_th = 0.0;
    m.times = times; m.lh = &lh[0][0]; m.bands = bands; m.pulses = NULL;
    for (i = 1; i < argc && n_dev < MAX_DEV; ++i) devices[n_dev++] = atoi(argv[i]);
    if (n_dev == 0) { devices[0] = 0; n_dev = 1; }
    for (c = 0; c < N_CAND; ++c) {                       /* vector-major: a shard holds whole chains (all splits of a parameter vector) */
        split[c] = splits[c % N_SPLIT];
        params[c][0] = vecs[c / N_SPLIT][0];
        params[c][1] = vecs[c / N_SPLIT][1];
    }
    if (misti_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", misti_last_error());
        return 2;
    }
    if (misti_create_multi(&m, n_dev, devices, &mm) != 0) {
        fprintf(stderr, "misti_create_multi: %s\n", misti_last_error());
        return 1;
    }
    /* whole chains per device: N_VEC chains of N_SPLIT members, chains in contiguous blocks */
    for (i = 0; i <= n_dev; ++i) lo[i] = (int64_t)N_SPLIT * ((int64_t)N_VEC * i / n_dev);
    for (i = 0; i < n_dev; ++i) { n[i] = lo[i + 1] - lo[i]; if (n[i] > per) per = n[i]; }
    for (i = 0; i < n_dev; ++i) {
        d_split[i] = (const double*)to_device(devices[i], split + lo[i], (size_t)n[i] * sizeof(double));
        d_par[i] = (const double*)to_device(devices[i], &params[lo[i]][0], (size_t)n[i] * 2 * sizeof(double));
        d_jsfs[i] = (const double*)to_device(devices[i], jsfs, sizeof jsfs);
        d_all[i] = (double*)to_device(devices[i], NULL, (size_t)n_dev * (size_t)per * sizeof(double));
        d_st[i] = (int32_t*)to_device(devices[i], NULL, (size_t)n_dev * (size_t)per * sizeof(int32_t));
        if (!d_split[i] || !d_par[i] || !d_jsfs[i] || !d_all[i] || !d_st[i]) { fprintf(stderr, "device memory on device %d\n", devices[i]); return 1; }
    }
    if (misti_multi_eval_batch_dev(mm, n, per, d_split, d_par, NULL, 1, d_jsfs, d_all, d_st) != 0 || misti_multi_sync(mm) != 0) {
        fprintf(stderr, "misti_multi_eval_batch_dev: %s\n", misti_last_error());
        misti_destroy_multi(mm);
        return 1;
    }
    if (misti_create(&m, devices[0], &one) != 0 ||
        misti_eval_batch(one, N_CAND, split, &params[0][0], NULL, 1, jsfs, llk_one, NULL, NULL, NULL, st_one) != 0) {
        fprintf(stderr, "single device: %s\n", misti_last_error());
        misti_destroy_multi(mm);
        return 1;
    }
    /* EVERY device holds the whole table: block r = shard r's rows, padded to `per` rows with NaN / -1 */
    for (i = 0; i < n_dev; ++i) {
        double table[MAX_DEV * N_CAND];
        int32_t status[MAX_DEV * N_CAND];
        int r;
        hipSetDevice(devices[i]);
        if (hipMemcpy(table, d_all[i], (size_t)n_dev * (size_t)per * sizeof(double), 2) != 0 ||
            hipMemcpy(status, d_st[i], (size_t)n_dev * (size_t)per * sizeof(int32_t), 2) != 0) { fprintf(stderr, "copy back\n"); return 1; }
        for (r = 0; r < n_dev; ++r)
            for (c = 0; c < per; ++c) {
                const double v = table[(size_t)r * per + c];
                if (c < n[r]) ok = ok && memcmp(&v, &llk_one[lo[r] + c], sizeof v) == 0 && status[(size_t)r * per + c] == st_one[lo[r] + c];
                else ok = ok && isnan(v) && status[(size_t)r * per + c] == -1;
            }
    }
    printf("contexts = %d\nrows_per_shard = %lld\nidentical = %d\nstatus = %d\nllh = %.15g\n", misti_multi_size(mm), (long long)per, ok, (int)st_one[0], llk_one[0]);
    for (i = 0; i < n_dev; ++i) { hipSetDevice(devices[i]); hipFree((void*)d_split[i]); hipFree((void*)d_par[i]); hipFree((void*)d_jsfs[i]); hipFree(d_all[i]); hipFree(d_st[i]); }
    misti_destroy(one);
    misti_destroy_multi(mm);
    return ok ? 0 : 3;
}

/* The multi-device form of the C ABI without Python: SURVEY appendix-A anchor A3's model (two optimised bands, smooth + cpfit +
 * unfolded) swept over 3 split values x 4 parameter vectors = 12 candidates in 4 chains, on a LIST of devices -
 * misti_create_multi / misti_multi_eval_batch - and, for comparison, on one context with misti_eval_batch: the two must agree
 * bit for bit.  The device list comes from the command line (default "0 0": two contexts on device 0, which is how the sharding is
 * exercised on a one-GPU machine; on a node: ./multi_device 0 1 2 3 4 5 6 7).  Replaces `parallel -j N ./MiSTI.py ...`
 * (/root/reference/README.md:110-115).
 *
 *   gcc -I include examples/multi_device.c -L misti_amd/csrc -lmisti_hip -Wl,-rpath,$PWD/misti_amd/csrc -o /tmp/multi_device
 *   /tmp/multi_device 0 0     # prints the llh of candidate 0 (split 5, rates 0.3 / 0.1; reference A3: -211.9189044185307)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "misti_hip.h"

#define N_SPLIT 3
#define N_VEC 4
#define N_CAND (N_SPLIT * N_VEC)

int main(int argc, char** argv) {
    const double times[7] = {0.01, 0.02, 0.04, 0.08, 0.16, 0.32, 0.64};
    const double lh[8][2] = {{1, 2}, {1, 2}, {0.8, 1.5}, {0.8, 1.5}, {1.2, 1.0}, {1.2, 1.0}, {0.9, 0.9}, {0.7, 0.7}};
    const double jsfs[8] = {100000, 900, 250, 1000, 600, 400, 260, 410};
    /* -mi 1 1 {st} r0 1  -mi 2 1 {st} r1 1 : end = -1 follows the candidate's split */
    const misti_band_t bands[2] = {{0, 1, -1, 0, 0.3}, {1, 1, -1, 1, 0.1}};
    const double vecs[N_VEC][2] = {{0.3, 0.1}, {0.05, 0.7}, {1.5, 0.02}, {0.2, 0.2}};
    const double splits[N_SPLIT] = {5.0, 4.0, 6.0};
    misti_model_t m;
    int devices[16], n_dev = 0, i, c;
    double split[N_CAND], params[N_CAND][2], llk_multi[N_CAND], llk_one[N_CAND], jafs_multi[N_CAND][7], jafs_one[N_CAND][7];
    int32_t st_multi[N_CAND], st_one[N_CAND];
    int64_t n_cand_dev[16], n_chain_dev[16];
    misti_multi* mm = NULL;
    misti_ctx* one = NULL;

    m.numT = 8; m.sample_date = 0; m.flags = MISTI_CPFIT | MISTI_SMOOTH | MISTI_UNFOLDED;
    m.n_band = 2; m.n_pulse = 0; m.n_param = 2; m.mixture_th = 0.0;
    m.times = times; m.lh = &lh[0][0]; m.bands = bands; m.pulses = NULL;
    for (i = 1; i < argc && n_dev < 16; ++i) devices[n_dev++] = atoi(argv[i]);
    if (n_dev == 0) { devices[0] = 0; devices[1] = 0; n_dev = 2; }
    for (c = 0; c < N_CAND; ++c) {                       /* split-major, as a sweep is written */
        split[c] = splits[c / N_VEC];
        params[c][0] = vecs[c % N_VEC][0];
        params[c][1] = vecs[c % N_VEC][1];
    }
    if (misti_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", misti_last_error());
        return 2;
    }
    if (misti_create_multi(&m, n_dev, devices, &mm) != 0) {
        fprintf(stderr, "misti_create_multi: %s\n", misti_last_error());
        return 1;
    }
    if (misti_multi_eval_batch(mm, N_CAND, split, &params[0][0], NULL, 1, jsfs, llk_multi, &jafs_multi[0][0], NULL, NULL, st_multi) != 0) {
        fprintf(stderr, "misti_multi_eval_batch: %s\n", misti_last_error());
        misti_destroy_multi(mm);
        return 1;
    }
    misti_multi_last_shards(mm, n_cand_dev, n_chain_dev);
    printf("contexts = %d\n", misti_multi_size(mm));
    for (i = 0; i < n_dev; ++i) printf("context %d: device %d, %lld candidates in %lld chains\n", i, devices[i], (long long)n_cand_dev[i], (long long)n_chain_dev[i]);
    if (misti_create(&m, devices[0], &one) != 0 ||
        misti_eval_batch(one, N_CAND, split, &params[0][0], NULL, 1, jsfs, llk_one, &jafs_one[0][0], NULL, NULL, st_one) != 0) {
        fprintf(stderr, "single device: %s\n", misti_last_error());
        misti_destroy_multi(mm);
        return 1;
    }
    i = memcmp(llk_multi, llk_one, sizeof llk_one) == 0 && memcmp(jafs_multi, jafs_one, sizeof jafs_one) == 0 && memcmp(st_multi, st_one, sizeof st_one) == 0;
    printf("identical = %d\nstatus = %d\nllh = %.15g\n", i, (int)st_multi[0], llk_multi[0]);
    misti_destroy(one);
    misti_destroy_multi(mm);
    return i ? 0 : 3;
}

/* The C ABI without Python: SURVEY appendix-A anchor A2 (split 5, no migration, smooth + cpfit) through
 * misti_create / misti_eval_batch, plain C, caller-allocated buffers.
 *
 *   gcc -I include examples/anchor_a2.c -L misti_amd/csrc -lmisti_hip -Wl,-rpath,$PWD/misti_amd/csrc -o /tmp/anchor_a2
 *   /tmp/anchor_a2          # prints  llh = -183.19954045052...   (reference: -183.1995404505269)
 */
#include <stdio.h>
#include <stdlib.h>

#include "misti_hip.h"

int main(void) {
    const double times[7] = {0.01, 0.02, 0.04, 0.08, 0.16, 0.32, 0.64};
    const double lh[8][2] = {{1, 2}, {1, 2}, {0.8, 1.5}, {0.8, 1.5}, {1.2, 1.0}, {1.2, 1.0}, {0.9, 0.9}, {0.7, 0.7}};
    const double jsfs[8] = {100000, 900, 250, 1000, 600, 400, 260, 410};
    misti_model_t m;
    m.numT = 8;
    m.sample_date = 0;
    m.flags = MISTI_CPFIT | MISTI_SMOOTH;
    m.n_band = 0;
    m.n_pulse = 0;
    m.n_param = 0;
    m.mixture_th = 0.0;
    m.times = times;
    m.lh = &lh[0][0];
    m.bands = NULL;
    m.pulses = NULL;

    if (misti_device_count() <= 0) {
        fprintf(stderr, "no HIP device: %s\n", misti_last_error());
        return 2;
    }
    misti_ctx* ctx = NULL;
    if (misti_create(&m, 0, &ctx) != 0) {
        fprintf(stderr, "misti_create: %s\n", misti_last_error());
        return 1;
    }
    const double split = 5.0;
    double llk = 0.0, jafs[7];
    int32_t status = -1;
    if (misti_eval_batch(ctx, 1, &split, NULL, NULL, 1, jsfs, &llk, jafs, NULL, NULL, &status) != 0) {
        fprintf(stderr, "misti_eval_batch: %s\n", misti_last_error());
        misti_destroy(ctx);
        return 1;
    }
    printf("status = %d\nllh = %.15g\n", (int)status, llk);
    printf("JAFS =");
    for (int i = 0; i < 7; ++i) printf(" %.15g", jafs[i]);
    printf("\n");
    misti_destroy(ctx);
    return 0;
}

// Does gfx950 skip the 16-lane passes of a wave64 fp64 instruction whose EXEC bits are all zero?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, int active, int iters) {
    const int lane = threadIdx.x & 63;
    double x0 = 1.0 + lane, x1 = 2.0 + lane, x2 = 3.0 + lane, x3 = 4.0 + lane, x4 = 5.0, x5 = 6.0, x6 = 7.0, x7 = 8.0;
    const double a = 0.999999, b = 1e-9;
    long long t0 = 0, t1 = 0;
    if (lane < active) {
        t0 = clock64();
        for (int i = 0; i < iters; ++i) {
            x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
            x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
        }
        t1 = clock64();
    }
    out[blockIdx.x * 64 + lane] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 8 * 1024);
    const int iters = 20000;
    for (int active : {64, 48, 32, 16, 8, 1}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, cyc, active, iters);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("active lanes %2d: %.2f cycles per fp64 FMA wave-instruction (one wave alone, 8 independent chains)\n", active, (double)c / (8.0 * iters));
    }
    return 0;
}

// What a WRITE-ONLY stream reaches on this chip (the replicate epilogue llk_kernel writes 8 bytes per value and reads almost nothing): 524 MB written by
//   (a) plain 16-byte stores, (b) nontemporal 16-byte stores, (c) hipMemsetAsync, and for scale (d) a float4 copy of the same size (read + write).
// hipcc --offload-arch=gfx950 -O3 -o fill_bw fill_bw.hip && ./fill_bw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double2v __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_kernel(double2v* __restrict__ out, size_t n2, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        double2v x = {v + (double)i, v};
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}

// the replicate epilogue's two candidate write patterns on a [rows][1000] table of doubles, 256 threads x (2 + 2) values = one 8 000-byte row per block and step:
//   CHUNKED: block b writes rows b*chunk .. b*chunk + chunk - 1 one after the other (at any moment the resident blocks write pieces `chunk` rows apart)
//   STRIDED: block b writes rows b, b + G, b + 2 G, ... (at any moment the resident blocks write one contiguous window of G rows that moves through the table)
template <bool STRIDED>
__global__ __launch_bounds__(256) void rows_kernel(double* __restrict__ out, long rows, int chunk, double v) {
    const long G = gridDim.x;
    const int t = threadIdx.x;
    for (long k = 0;; ++k) {
        const long r = STRIDED ? (long)blockIdx.x + k * G : (long)blockIdx.x * chunk + k;
        if (r >= rows || (!STRIDED && k >= chunk)) break;
        double2v* o = (double2v*)(out + r * 1000);
        double2v x = {v + (double)r, v + t};
        o[t] = x;                                       // values 2 t, 2 t + 1
        if (512 + 2 * t < 1000) o[256 + t] = x;         // values 512 + 2 t, + 1
    }
}

int main() {
    const size_t bytes = (size_t)65536 * 1000 * 8;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 0, bytes); (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto&& launch, double moved) {
        for (int i = 0; i < 3; ++i) launch();
        (void)hipEventRecord(e0, 0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %.4f ms per launch  %.2f TB/s\n", name, ms / reps, moved / (ms / reps * 1e-3) / 1e12);
    };
    for (int blocks : {2048, 8192, 32768}) {
        printf("grid %d x 256\n", blocks);
        timeit("  plain 16-byte stores (write only)", [&] { fill_kernel<false><<<blocks, 256>>>((double2v*)a, bytes / 16, 1.0); }, (double)bytes);
        timeit("  nontemporal 16-byte stores (write only)", [&] { fill_kernel<true><<<blocks, 256>>>((double2v*)a, bytes / 16, 1.0); }, (double)bytes);
        timeit("  float4 copy (read + write, both counted)", [&] { copy_kernel<<<blocks, 256>>>((const float4*)a, (float4*)b, bytes / 16); }, 2.0 * bytes);
    }
    const long rows = 65536;
    for (int chunk : {64, 32, 8, 1}) {
        char name[96]; snprintf(name, sizeof name, "rows: block = %d consecutive rows (grid %ld)", chunk, rows / chunk);
        timeit(name, [&] { rows_kernel<false><<<(unsigned)(rows / chunk), 256>>>((double*)a, rows, chunk, 1.0); }, (double)bytes);
    }
    for (int G : {1024, 2048, 4096, 8192}) {
        char name[96]; snprintf(name, sizeof name, "rows: block b = rows b, b + %d, ... (grid %d)", G, G);
        timeit(name, [&] { rows_kernel<true><<<G, 256>>>((double*)a, rows, 0, 1.0); }, (double)bytes);
    }
    timeit("hipMemsetAsync (write only)", [&] { (void)hipMemsetAsync(a, 0, bytes, 0); }, (double)bytes);
    return 0;
}

// How a burst of batches on many streams starts from an idle device: N streams, each given `pre` short kernels, one long kernel (spins
// `long_us`, 20 workgroups of 128 threads - the shape of the follow kernel on config 2) and `post` short dependent kernels.
// Prints, per (N, pre, post): the burst's wall time and when each stream's long kernel started and ended (device clock, relative to the first).
// hipcc --offload-arch=gfx950 -O2 -o queue_burst queue_burst.hip && ./queue_burst
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void spin_kernel(long long ticks, long long* stamp) {
    const long long t0 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp) stamp[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (blockIdx.x == 0 && threadIdx.x == 0 && stamp) stamp[1] = wall_clock64();
}

__global__ void short_kernel(double* x) { x[blockIdx.x * blockDim.x + threadIdx.x] += 1.0; }

int main(int argc, char** argv) {
    const double long_us = argc > 1 ? atof(argv[1]) : 1400.0;
    const long long ticks = (long long)(long_us * 100.0);        // wall_clock64: 100 MHz
    double* x;
    hipMalloc(&x, 256 * 128 * sizeof(double));
    hipMemset(x, 0, 256 * 128 * sizeof(double));
    long long* stamps;
    hipHostMalloc(&stamps, 64 * 2 * sizeof(long long));
    const int shapes[][2] = {{0, 0}, {1, 1}, {1, 2}};
    // how the burst is waited for: 0 hipDeviceSynchronize; 1 poll hipStreamQuery on every stream, then hipDeviceSynchronize;
    // 2 an event per stream joined on one stream (device-side waits), hipStreamSynchronize of that one, then hipDeviceSynchronize
    // 3 an event recorded behind each stream's launches AT ISSUE, hipEventSynchronize of every one, then hipDeviceSynchronize
    // 4 the same events, polled with hipEventQuery;  5 hipStreamSynchronize of every stream, then hipDeviceSynchronize
    const int wait_mode = argc > 2 ? atoi(argv[2]) : 0;
    if (argc > 3) hipSetDeviceFlags(atoi(argv[3]) ? hipDeviceScheduleSpin : hipDeviceScheduleBlockingSync);
    const int use_graph = argc > 4 ? atoi(argv[4]) : 0;      // 1: the launches of a stream are ONE captured graph (hipGraphLaunch per stream)
    hipStream_t join;
    hipStreamCreateWithFlags(&join, hipStreamNonBlocking);
    std::vector<hipEvent_t> evs(32);
    for (auto& e : evs) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int N : {4, 12, 20}) {
        std::vector<hipStream_t> st(N);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (auto& sh : shapes) {
            std::vector<hipGraphExec_t> gx(N, nullptr);
            if (use_graph) {
                for (int i = 0; i < N; ++i) {
                    hipGraph_t g;
                    hipStreamBeginCapture(st[i], hipStreamCaptureModeThreadLocal);
                    for (int k = 0; k < sh[0]; ++k) short_kernel<<<256, 128, 0, st[i]>>>(x);
                    spin_kernel<<<20, 128, 0, st[i]>>>(ticks, stamps + 2 * i);
                    for (int k = 0; k < sh[1]; ++k) short_kernel<<<256, 128, 0, st[i]>>>(x);
                    hipStreamEndCapture(st[i], &g);
                    hipGraphInstantiate(&gx[i], g, nullptr, nullptr, 0);
                    hipGraphDestroy(g);
                    hipGraphLaunch(gx[i], st[i]);            // first launch uploads
                }
                hipDeviceSynchronize();
            }
            double best = 1e30;
            std::vector<double> starts;
            for (int rep = 0; rep < 5; ++rep) {
                hipDeviceSynchronize();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) {
                    if (use_graph) { hipGraphLaunch(gx[i], st[i]); continue; }
                    for (int k = 0; k < sh[0]; ++k) short_kernel<<<256, 128, 0, st[i]>>>(x);
                    spin_kernel<<<20, 128, 0, st[i]>>>(ticks, stamps + 2 * i);
                    for (int k = 0; k < sh[1]; ++k) short_kernel<<<256, 128, 0, st[i]>>>(x);
                    if (wait_mode == 3 || wait_mode == 4) hipEventRecord(evs[i], st[i]);
                }
                const auto t1 = std::chrono::steady_clock::now();
                if (wait_mode == 3) for (int i = 0; i < N; ++i) hipEventSynchronize(evs[i]);
                if (wait_mode == 4) for (int i = 0; i < N; ++i) while (hipEventQuery(evs[i]) == hipErrorNotReady) {}
                if (wait_mode == 5) for (int i = 0; i < N; ++i) hipStreamSynchronize(st[i]);
                if (wait_mode == 1) {
                    for (int i = 0; i < N; ++i) while (hipStreamQuery(st[i]) == hipErrorNotReady) {}
                } else if (wait_mode == 2) {
                    for (int i = 0; i < N; ++i) { hipEventRecord(evs[i], st[i]); hipStreamWaitEvent(join, evs[i], 0); }
                    hipStreamSynchronize(join);
                }
                const auto t1b = std::chrono::steady_clock::now();
                hipDeviceSynchronize();
                const auto t2 = std::chrono::steady_clock::now();
                const double ms = std::chrono::duration<double, std::milli>(t2 - t0).count();
                if (ms < best) {
                    best = ms;
                    starts.clear();
                    long long first = stamps[0];
                    for (int i = 0; i < N; ++i) first = std::min(first, stamps[2 * i]);
                    for (int i = 0; i < N; ++i) starts.push_back((stamps[2 * i] - first) / 100.0);
                    long long last = 0;
                    for (int i = 0; i < N; ++i) last = std::max(last, stamps[2 * i + 1]);
                    starts.push_back((last - first) / 100.0);
                    starts.push_back(std::chrono::duration<double, std::micro>(t2 - t1b).count());
                    starts.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
                }
            }
            for (auto g : gx) if (g) hipGraphExecDestroy(g);
            printf("N=%2d pre=%d post=%d  burst %.3f ms (issue %.0f us; last long kernel ends at %.0f us; final hipDeviceSynchronize %.0f us)",
                   N, sh[0], sh[1], best, starts[N + 2], starts[N], starts[N + 1]);
            if (argc > 5) { printf("  long-kernel starts [us]:"); for (int i = 0; i < N; ++i) printf(" %.0f", starts[i]); }
            printf("\n");
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}

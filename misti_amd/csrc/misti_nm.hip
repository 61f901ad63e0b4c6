// Batched Nelder-Mead with device-resident simplices (MI355X / gfx950).
//
// Replaces MigrationInference.Solve (MigrationInference.py:718-733: scipy.optimize.minimize(method='Nelder-Mead',
// xatol = fatol = tol, maxiter = 1000) on -JAFSLikelihood) for MANY starts at once (BASELINE config 3: 16 384 random
// starts).  Every start follows SciPy's iteration exactly (scipy/optimize/_optimize.py, _minimize_neldermead,
// adaptive=False, SciPy 1.15.3): same initial simplex, same reflection / expansion / contraction / shrink decisions,
// same termination test, same evaluation count - so a start's trajectory equals scipy's on the same objective.
// Simplices, function values and every accept / shrink decision live in HBM: one thread per start in four small
// kernels per iteration, the objective in between as ordinary engine batches (misti_eval_batch_dev's path) on the
// same stream.  Live starts are compacted into the leading slots of each iteration's batches (the host sizes the
// batches from a count of live starts that is at most two iterations old: an upper bound, the number only falls);
// a slot without a start, or a start that needs no point in a phase, hands the engine a candidate with a negative
// split time: it leaves the spectrum kernel at once with -inf and costs nothing.  Nothing of the search crosses PCIe
// inside the loop except that 4-byte count.
//
// Arithmetic: SciPy evaluates  (1 + rho) * xbar - rho * worst  etc. with one rounding per operation.  The library is
// built with -ffp-contract=fast (which ignores contraction pragmas, and HIP's __dmul_rn / __dadd_rn are plain operators
// in a header), so every product that feeds a sum passes through rn(): an empty asm the compiler cannot see through,
// hence cannot fuse into an fma.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "misti_device.h"

namespace misti {

namespace {

constexpr double NM_RHO = 1.0, NM_CHI = 2.0, NM_PSI = 0.5, NM_SIGMA = 0.5;
constexpr double NM_NONZDELT = 0.05, NM_ZDELT = 0.00025;

__device__ __forceinline__ double objective(double llk) { return (llk == llk && llk > -INFINITY) ? -llk : INFINITY; }   // -JAFSLikelihood; no value = +inf

// the value as rounded so far: opaque to the optimiser, so a product passed through it is never contracted into an fma
__device__ __forceinline__ double rn(double v) { __asm__ volatile("" : "+v"(v)); return v; }

// a*x - b*y with a rounding after each product and after the difference (NumPy elementwise semantics)
__device__ __forceinline__ double lin2(double a, double x, double b, double y) { return rn(a * x) - rn(b * y); }

// numpy.argsort on <= 17 values: insertion sort (stable), NaN last
__device__ __forceinline__ bool before(double a, double b) { return a < b || (b != b && a == a); }

__device__ void sort_simplex(const NmState& st, int64_t s) {
    const int N = st.N, V = N + 1;
    double* f = st.fsim + s * V;
    double* x = st.sim + s * (int64_t)V * N;
    double* tmp = st.scratch + s * (int64_t)V * N;
    int idx[MISTI_MAX_PARAMS + 1];
    double fv[MISTI_MAX_PARAMS + 1];
    for (int i = 0; i < V; ++i) { idx[i] = i; fv[i] = f[i]; }
    bool moved = false;
    for (int i = 1; i < V; ++i) {
        const double key = fv[i];
        const int ki = idx[i];
        int j = i - 1;
        while (j >= 0 && before(key, fv[j])) { fv[j + 1] = fv[j]; idx[j + 1] = idx[j]; --j; moved = true; }
        fv[j + 1] = key; idx[j + 1] = ki;
    }
    if (!moved) return;
    for (int i = 0; i < V; ++i) for (int k = 0; k < N; ++k) tmp[i * N + k] = x[idx[i] * N + k];
    for (int i = 0; i < V; ++i) { f[i] = fv[i]; for (int k = 0; k < N; ++k) x[i * N + k] = tmp[i * N + k]; }
}

// xbar[k] = (sim[0][k] + sim[1][k] + ... + sim[N-1][k]) / N   (np.add.reduce(sim[:-1], 0) / N: rows added in order)
__device__ __forceinline__ double centroid(const double* x, int N, int k) {
    double a = x[k];
    for (int i = 1; i < N; ++i) a = rn(a + x[i * N + k]);
    return rn(a / (double)N);
}

// Top of SciPy's while loop for start s: may it iterate, has it converged; a start that goes on takes the next free
// SLOT of the coming iteration's batches and leaves its reflection point there.  Live starts are thereby compacted:
// the batches of an iteration are as long as the number of starts still running (the host sizes them from a count that
// is at most two iterations old - an upper bound, since the number only falls), not as long as the number of starts.
__device__ void next_reflection(const NmState& st, int64_t s) {
    const int N = st.N, V = N + 1;
    const double* f = st.fsim + s * V;
    const double* x = st.sim + s * (int64_t)V * N;
    bool go = (int64_t)st.nfev[s] < st.maxfun && st.nit[s] < st.maxiter;
    int status = go ? 0 : ((int64_t)st.nfev[s] >= st.maxfun ? 1 : 2);        // warnflag: 1 maxfev, 2 maxiter
    if (go) {
        double dx = 0.0, df = 0.0;
        bool dx_nan = false, df_nan = false;
        for (int i = 1; i < V; ++i) {
            for (int k = 0; k < N; ++k) { const double d = fabs(x[i * N + k] - x[k]); if (d != d) dx_nan = true; if (d > dx) dx = d; }
            const double d = fabs(f[0] - f[i]);
            if (d != d) df_nan = true;
            if (d > df) df = d;
        }
        // np.max propagates NaN, and NaN <= tol is False
        if (!dx_nan && !df_nan && dx <= st.xatol && df <= st.fatol) go = false;
    }
    if (!go) {
        if (st.done[s] < 0) st.done[s] = status;
        return;
    }
    const int slot = atomicAdd(st.count_next, 1);           // which slot a start gets never matters: a point's value does not depend on the batch
    st.idx_next[slot] = (int32_t)s;
    double* p1 = st.p1 + (int64_t)slot * N;
    for (int k = 0; k < N; ++k) p1[k] = lin2(1.0 + NM_RHO, centroid(x, N, k), NM_RHO, x[N * N + k]);     // xr
    st.split1[slot] = st.split;
}

}  // namespace

// initial simplices -> the (N + 1) points of every start, in start-major order
__global__ __launch_bounds__(256)
void nm_init_kernel(NmState st, const double* __restrict__ starts) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= st.S) return;
    const int N = st.N, V = N + 1;
    double* x = st.sim + s * (int64_t)V * N;
    for (int i = 0; i < V; ++i) {
        for (int k = 0; k < N; ++k) {
            double y = starts[s * N + k];
            if (i == k + 1) y = (y != 0.0) ? rn((1.0 + NM_NONZDELT) * y) : NM_ZDELT;
            x[i * N + k] = y;
        }
        st.split0[s * V + i] = st.split;
    }
    st.nit[s] = 1; st.nfev[s] = 0; st.done[s] = -1;
}

// values of the initial simplex are in: sort, then the first reflection point
__global__ __launch_bounds__(256)
void nm_begin_kernel(NmState st, const double* __restrict__ llk0) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= st.S) return;
    const int V = st.N + 1;
    for (int i = 0; i < V; ++i) st.fsim[s * V + i] = objective(llk0[s * V + i]);
    st.nfev[s] = V;
    sort_simplex(st, s);
    next_reflection(st, s);
}

// the reflection value is in: which second point does SciPy evaluate, if any.  Thread = slot of this iteration.
__global__ __launch_bounds__(256)
void nm_reflect_kernel(NmState st, int64_t bound, const double* __restrict__ llk1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound) return;
    const int N = st.N, V = N + 1;
    if (i >= st.count_cur[0]) { st.split2[i] = -1.0; return; }         // no start in this slot
    const int64_t s = st.idx_cur[i];
    double* p2 = st.p2 + i * N;
    const double* f = st.fsim + s * V;
    const double* x = st.sim + s * (int64_t)V * N;
    const double fxr = objective(llk1[i]);
    st.fxr[s] = fxr;
    st.nfev[s] += 1;
    int kind;
    if (fxr < f[0]) {
        kind = NM_EXPAND;
        for (int k = 0; k < N; ++k) p2[k] = lin2(1.0 + NM_RHO * NM_CHI, centroid(x, N, k), NM_RHO * NM_CHI, x[N * N + k]);
    } else if (fxr < f[N - 1]) {
        kind = NM_REFLECT;                                  // fsim[-2]: accepted as it is
    } else if (fxr < f[N]) {
        kind = NM_CONTRACT;
        for (int k = 0; k < N; ++k) p2[k] = lin2(1.0 + NM_PSI * NM_RHO, centroid(x, N, k), NM_PSI * NM_RHO, x[N * N + k]);
    } else {
        kind = NM_INSIDE;                                   // (1 - psi) * xbar + psi * sim[-1]
        for (int k = 0; k < N; ++k) p2[k] = rn((1.0 - NM_PSI) * centroid(x, N, k)) + rn(NM_PSI * x[N * N + k]);
    }
    st.kind[s] = kind;
    if (kind == NM_REFLECT) for (int k = 0; k < N; ++k) p2[k] = 0.0;
    st.split2[i] = kind == NM_REFLECT ? -1.0 : st.split;
}

// the second value is in: replace the worst vertex, or shrink (then the N shrunk vertices are the third batch)
__global__ __launch_bounds__(256)
void nm_accept_kernel(NmState st, int64_t bound, const double* __restrict__ llk2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound) return;
    const int N = st.N, V = N + 1;
    double* p3 = st.p3 + i * (int64_t)N * N;
    if (i >= st.count_cur[0]) { for (int j = 0; j < N; ++j) st.split3[i * N + j] = -1.0; return; }
    const int64_t s = st.idx_cur[i];
    double* f = st.fsim + s * V;
    double* x = st.sim + s * (int64_t)V * N;
    const double* p1 = st.p1 + i * N;
    const double* p2 = st.p2 + i * N;
    const int kind = st.kind[s];
    bool shrink = false;
    const double fxr = st.fxr[s];
    const double* take = p1;
    double ftake = fxr;
    if (kind != NM_REFLECT) {
        const double f2 = objective(llk2[i]);
        st.nfev[s] += 1;
        if (kind == NM_EXPAND) { if (f2 < fxr) { take = p2; ftake = f2; } }
        else if (kind == NM_CONTRACT) { if (f2 <= fxr) { take = p2; ftake = f2; } else shrink = true; }
        else { if (f2 < f[N]) { take = p2; ftake = f2; } else shrink = true; }
    }
    if (!shrink) { for (int k = 0; k < N; ++k) x[N * N + k] = take[k]; f[N] = ftake; }
    else
        for (int j = 1; j < V; ++j)
            for (int k = 0; k < N; ++k) {
                const double v = x[k] + rn(NM_SIGMA * rn(x[j * N + k] - x[k]));   // sim[0] + sigma (sim[j] - sim[0])
                x[j * N + k] = v;
                p3[(j - 1) * N + k] = v;
            }
    st.shrunk[s] = shrink ? 1 : 0;
    for (int j = 0; j < N; ++j) {
        st.split3[i * N + j] = shrink ? st.split : -1.0;
        if (!shrink) for (int k = 0; k < N; ++k) p3[j * N + k] = 0.0;
    }
}

// shrink values are in: end of the iteration (count, sort) and the top of the next one
__global__ __launch_bounds__(256)
void nm_finish_kernel(NmState st, int64_t bound, const double* __restrict__ llk3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound || i >= st.count_cur[0]) return;
    const int N = st.N, V = N + 1;
    const int64_t s = st.idx_cur[i];
    if (st.shrunk[s]) {
        for (int j = 1; j < V; ++j) st.fsim[s * V + j] = objective(llk3[i * N + (j - 1)]);
        st.nfev[s] += N;
    }
    st.nit[s] += 1;
    sort_simplex(st, s);
    next_reflection(st, s);
}

// results: best vertex, its value (as a log-likelihood), counters
__global__ __launch_bounds__(256)
void nm_result_kernel(NmState st, double* __restrict__ x_out, double* __restrict__ llh_out, int32_t* __restrict__ status) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= st.S) return;
    const int N = st.N, V = N + 1;
    for (int k = 0; k < N; ++k) x_out[s * N + k] = st.sim[s * (int64_t)V * N + k];
    llh_out[s] = -st.fsim[s * V];
    if (status) status[s] = st.done[s] < 0 ? 2 : st.done[s];          // still live when the host stopped issuing: iteration cap
}

static dim3 nm_grid(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

hipError_t launch_nm_init(const NmState& st, const double* starts, hipStream_t stream) {
    hipLaunchKernelGGL(nm_init_kernel, nm_grid(st.S), dim3(256), 0, stream, st, starts);
    return hipGetLastError();
}
hipError_t launch_nm_begin(const NmState& st, const double* llk0, hipStream_t stream) {
    hipLaunchKernelGGL(nm_begin_kernel, nm_grid(st.S), dim3(256), 0, stream, st, llk0);
    return hipGetLastError();
}
hipError_t launch_nm_reflect(const NmState& st, int64_t bound, const double* llk1, hipStream_t stream) {
    hipLaunchKernelGGL(nm_reflect_kernel, nm_grid(bound), dim3(256), 0, stream, st, bound, llk1);
    return hipGetLastError();
}
hipError_t launch_nm_accept(const NmState& st, int64_t bound, const double* llk2, hipStream_t stream) {
    hipLaunchKernelGGL(nm_accept_kernel, nm_grid(bound), dim3(256), 0, stream, st, bound, llk2);
    return hipGetLastError();
}
hipError_t launch_nm_finish(const NmState& st, int64_t bound, const double* llk3, hipStream_t stream) {
    hipLaunchKernelGGL(nm_finish_kernel, nm_grid(bound), dim3(256), 0, stream, st, bound, llk3);
    return hipGetLastError();
}
hipError_t launch_nm_result(const NmState& st, double* x, double* llh, int32_t* status, hipStream_t stream) {
    hipLaunchKernelGGL(nm_result_kernel, nm_grid(st.S), dim3(256), 0, stream, st, x, llh, status);
    return hipGetLastError();
}

}  // namespace misti

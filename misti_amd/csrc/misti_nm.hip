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
// Evaluation budget (maxfev; basin hopping's minimisations run with SciPy's default maxiter = maxfev = 200 N): SciPy's wrapper
// (_wrap_scalar_function_maxfun_validation) raises BEFORE the call that would exceed it, the iteration in progress is abandoned -
// nothing accepted, `iterations` not incremented, a shrink applied up to and including the vertex whose evaluation was refused
// (which keeps its old value) - the simplex is sorted and the loop ends with warnflag 1.  Restated here per evaluation
// (NM_CUT, NmState::shrunk): nfev never exceeds maxfev.
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
    const int n0 = (int64_t)V <= st.maxfun ? V : (int)st.maxfun;             // fsim starts as +inf; evaluations beyond the budget are refused
    for (int i = 0; i < V; ++i) st.fsim[s * V + i] = i < n0 ? objective(llk0[s * V + i]) : INFINITY;
    st.nfev[s] = n0;
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
    if (kind != NM_REFLECT && (int64_t)st.nfev[s] >= st.maxfun) kind = NM_CUT;      // the second point's evaluation is refused
    st.kind[s] = kind;
    const bool second = kind != NM_REFLECT && kind != NM_CUT;
    if (!second) for (int k = 0; k < N; ++k) p2[k] = 0.0;
    st.split2[i] = second ? st.split : -1.0;
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
    if (kind != NM_REFLECT && kind != NM_CUT) {
        const double f2 = objective(llk2[i]);
        st.nfev[s] += 1;
        if (kind == NM_EXPAND) { if (f2 < fxr) { take = p2; ftake = f2; } }
        else if (kind == NM_CONTRACT) { if (f2 <= fxr) { take = p2; ftake = f2; } else shrink = true; }
        else { if (f2 < f[N]) { take = p2; ftake = f2; } else shrink = true; }
    }
    int n_eval = 0;                // shrunk vertices the budget still pays for
    if (kind == NM_CUT) {
        // iteration abandoned before anything was accepted
    } else if (!shrink) { for (int k = 0; k < N; ++k) x[N * N + k] = take[k]; f[N] = ftake; }
    else {
        const int64_t left = st.maxfun - (int64_t)st.nfev[s];
        n_eval = left >= N ? N : (left > 0 ? (int)left : 0);
        const int n_move = n_eval < N ? n_eval + 1 : N;      // SciPy moves sim[j] before the call that is refused
        for (int j = 1; j <= n_move; ++j)
            for (int k = 0; k < N; ++k) {
                const double v = x[k] + rn(NM_SIGMA * rn(x[j * N + k] - x[k]));   // sim[0] + sigma (sim[j] - sim[0])
                x[j * N + k] = v;
                if (j <= n_eval) p3[(j - 1) * N + k] = v;
            }
    }
    st.shrunk[s] = shrink ? 1 + n_eval : 0;
    for (int j = 0; j < N; ++j) {
        const bool live = shrink && j < n_eval;
        st.split3[i * N + j] = live ? st.split : -1.0;
        if (!live) for (int k = 0; k < N; ++k) p3[j * N + k] = 0.0;
    }
}

// shrink values are in: end of the iteration (count, sort) and the top of the next one
__global__ __launch_bounds__(256)
void nm_finish_kernel(NmState st, int64_t bound, const double* __restrict__ llk3) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound || i >= st.count_cur[0]) return;
    const int N = st.N, V = N + 1;
    const int64_t s = st.idx_cur[i];
    bool cut = st.kind[s] == NM_CUT;
    if (st.shrunk[s]) {
        const int n_eval = st.shrunk[s] - 1;
        for (int j = 1; j <= n_eval; ++j) st.fsim[s * V + j] = objective(llk3[i * N + (j - 1)]);
        st.nfev[s] += n_eval;
        cut = cut || n_eval < N;
    }
    if (!cut) st.nit[s] += 1;                  // an abandoned iteration does not count (the exception skips `iterations += 1`)
    sort_simplex(st, s);
    next_reflection(st, s);
}

// ---- speculative iteration (few live starts) ---------------------------------------------------------------------------------
// With a handful of starts still running an iteration is three dependent engine batches of a few candidates each - three
// times the latency of one lambda-correction chain while the chip idles (BASELINE config 3: 64 of 16 384 starts never meet
// fatol and run to maxiter; that tail was 70 % of the search's wall time).  Every point SciPy COULD evaluate in an iteration is
// known at its top - reflection, expansion, outside and inside contraction, the N shrunk vertices - so below spec_cap live
// starts all 4 + N of them go out as ONE batch and one kernel takes SciPy's decisions from the values it would have asked
// for, counting only those (nfev is SciPy's).  Same expressions as the three-batch path: same bits.
__device__ __forceinline__ void spec_points(const NmState& st, int64_t bound, int64_t i) {
    if (i == 0) st.count_next[0] = 0;          // the slot counter the next finish fills (later in stream order): no memset launch for it
    if (i >= bound) return;
    const int N = st.N, V = N + 1, K = 4 + N;
    double* pt = st.ps + i * (int64_t)K * N;
    double* sp = st.ps_split + i * K;
    if (i >= st.count_cur[0]) { for (int j = 0; j < K; ++j) { sp[j] = -1.0; for (int k = 0; k < N; ++k) pt[j * N + k] = 0.0; } return; }
    const int64_t s = st.idx_cur[i];
    const double* x = st.sim + s * (int64_t)V * N;
    const double* p1 = st.p1 + i * N;
    for (int k = 0; k < N; ++k) {
        const double xb = centroid(x, N, k), w = x[N * N + k];
        pt[0 * N + k] = p1[k];                                                            // xr (next_reflection)
        pt[1 * N + k] = lin2(1.0 + NM_RHO * NM_CHI, xb, NM_RHO * NM_CHI, w);              // xe
        pt[2 * N + k] = lin2(1.0 + NM_PSI * NM_RHO, xb, NM_PSI * NM_RHO, w);              // xc
        pt[3 * N + k] = rn((1.0 - NM_PSI) * xb) + rn(NM_PSI * w);                         // xcc
        for (int j = 1; j < V; ++j) pt[(3 + j) * N + k] = x[k] + rn(NM_SIGMA * rn(x[j * N + k] - x[k]));   // shrunk vertex j
    }
    for (int j = 0; j < K; ++j) sp[j] = st.split;
}

__global__ __launch_bounds__(256)
void nm_spec_points_kernel(NmState st, int64_t bound) { spec_points(st, bound, (int64_t)blockIdx.x * blockDim.x + threadIdx.x); }

__device__ __forceinline__ void spec_finish(const NmState& st, int64_t bound, const double* __restrict__ llk, int64_t i) {
    if (i >= bound || i >= st.count_cur[0]) return;
    const int N = st.N, V = N + 1, K = 4 + N;
    const int64_t s = st.idx_cur[i];
    double* f = st.fsim + s * V;
    double* x = st.sim + s * (int64_t)V * N;
    const double* pt = st.ps + i * (int64_t)K * N;
    const double* v = llk + i * K;
    const double fxr = objective(v[0]);
    int nfev = st.nfev[s] + 1;
    int take = 0;                 // which point replaces the worst vertex; -1: shrink; -2: budget ran out before the second point
    double ftake = fxr;
    const bool broke = (int64_t)nfev >= st.maxfun;      // a second evaluation would be refused
    if (fxr < f[0]) {             // expansion
        if (broke) take = -2;
        else { const double f2 = objective(v[1]); ++nfev; if (f2 < fxr) { take = 1; ftake = f2; } }
    } else if (fxr < f[N - 1]) {
        // accepted as it is
    } else if (fxr < f[N]) {      // outside contraction
        if (broke) take = -2;
        else { const double f2 = objective(v[2]); ++nfev; if (f2 <= fxr) { take = 2; ftake = f2; } else take = -1; }
    } else {                      // inside contraction
        if (broke) take = -2;
        else { const double f2 = objective(v[3]); ++nfev; if (f2 < f[N]) { take = 3; ftake = f2; } else take = -1; }
    }
    bool cut = take == -2;
    int n_eval = 0;
    if (take >= 0) { for (int k = 0; k < N; ++k) x[N * N + k] = pt[take * N + k]; f[N] = ftake; }
    else if (take == -1) {
        const int64_t left = st.maxfun - (int64_t)nfev;
        n_eval = left >= N ? N : (left > 0 ? (int)left : 0);
        const int n_move = n_eval < N ? n_eval + 1 : N;
        for (int j = 1; j <= n_move; ++j) { for (int k = 0; k < N; ++k) x[j * N + k] = pt[(3 + j) * N + k]; if (j <= n_eval) f[j] = objective(v[3 + j]); }
        nfev += n_eval;
        cut = n_eval < N;
    }
    st.nfev[s] = nfev;
    st.shrunk[s] = take == -1 ? 1 + n_eval : 0;
    if (!cut) st.nit[s] += 1;
    sort_simplex(st, s);
    next_reflection(st, s);
}

__global__ __launch_bounds__(256)
void nm_spec_finish_kernel(NmState st, int64_t bound, const double* __restrict__ llk) {
    spec_finish(st, bound, llk, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// The decision kernel of speculative iteration k and the points kernel of iteration k + 1 in ONE launch, and the count of live starts
// straight into the host's pinned word: possible because a speculative iteration has at most 1 024 / (4 + N) <= 204 live starts - one
// workgroup, whose barrier stands for the kernel boundary between the two (the compaction of the live starts is an atomic counter over
// the threads).  Per iteration of the search's long tail that is two launches and a 4-byte copy less: ~25 of ~560 microseconds.
// `nx` is the state as the NEXT iteration sees it (the lists swapped).
__global__ __launch_bounds__(256)
void nm_spec_step_kernel(NmState st, NmState nx, int64_t bound, const double* __restrict__ llk, volatile int32_t* live_host) {
    const int64_t i = threadIdx.x;
    spec_finish(st, bound, llk, i);
    __threadfence_block();
    __syncthreads();
    if (i == 0 && live_host) *live_host = st.count_next[0];
    spec_points(nx, bound, i);
}

// ---- basin hopping around the batched minimiser ---------------------------------------------------------------------------
// scipy.optimize.basinhopping (SciPy 1.15.3, _basinhopping.py) per start: BasinHoppingRunner.__init__ / one_cycle,
// AdaptiveStepsize.take_step (:245-250, adjustment every `interval` steps :230-243), RandomDisplacement (:275-278:
// x += rng.uniform(-stepsize, stepsize)), Metropolis.accept_reject (:316-336), Storage.update (:29-35).  The uniforms are
// drawn on the host by the caller's own numpy Generator, in SciPy's order (N for the displacement, then 1 for the acceptance
// test, per hop) - their NUMBER does not depend on the data, only their use does - so a start reproduces
// scipy.optimize.basinhopping(rng=that generator) on the same objective.
__global__ __launch_bounds__(256)
void bh_update_kernel(BhState bh, NmState nm, const double* __restrict__ x_min, const double* __restrict__ llh_min, const int32_t* __restrict__ nm_status,
                      int hop, int niter, const double* __restrict__ uniforms) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bh.S) return;
    const int N = bh.N;
    const double f_new = -llh_min[s];                       // OptimizeResult.fun of the minimisation (+inf: no vertex has a value)
    const bool ok_new = nm_status[s] == 0;
    if (hop < 0) {                                          // BasinHoppingRunner.__init__ (:80-97)
        for (int k = 0; k < N; ++k) { bh.x_cur[s * N + k] = x_min[s * N + k]; bh.x_best[s * N + k] = x_min[s * N + k]; }
        bh.f_cur[s] = f_new; bh.f_best[s] = f_new;
        bh.ok_cur[s] = ok_new; bh.ok_best[s] = ok_new;
        bh.nfev[s] = nm.nfev[s];
        bh.failures[s] = ok_new ? 0 : 1;
        bh.accepted[s] = 0;
        bh.nstep[s] = 0; bh.naccept[s] = 0;
        return;
    }
    bh.nfev[s] += nm.nfev[s];
    if (!ok_new) bh.failures[s] += 1;
    // Metropolis: prod = -(f_new - f_old) * beta; w = exp(min(0, prod)) - Python's min(0, nan) is 0
    const double prod = -(f_new - bh.f_cur[s]) * bh.beta;
    const double w = exp(prod < 0.0 ? prod : 0.0);
    const double rnd = uniforms[(s * (int64_t)niter + hop) * (N + 1) + N];
    const bool accept = (w >= rnd) && (ok_new || !bh.ok_cur[s]);
    if (accept) {
        bh.naccept[s] += 1;                                 // AdaptiveStepsize.report
        bh.accepted[s] += 1;
        for (int k = 0; k < N; ++k) bh.x_cur[s * N + k] = x_min[s * N + k];
        bh.f_cur[s] = f_new;
        bh.ok_cur[s] = ok_new;
        if (ok_new && (f_new < bh.f_best[s] || !bh.ok_best[s])) {     // Storage.update
            for (int k = 0; k < N; ++k) bh.x_best[s * N + k] = x_min[s * N + k];
            bh.f_best[s] = f_new;
            bh.ok_best[s] = 1;
        }
    }
}

__global__ __launch_bounds__(256)
void bh_step_kernel(BhState bh, int hop, int niter, const double* __restrict__ uniforms, double* __restrict__ trial) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bh.S) return;
    const int N = bh.N;
    int nstep = bh.nstep[s] + 1;
    double step = bh.stepsize[s];
    if (nstep % bh.interval == 0) {                         // _adjust_step_size
        const double rate = (double)bh.naccept[s] / (double)nstep;
        step = rate > bh.target ? step / bh.factor : step * bh.factor;
        bh.stepsize[s] = step;
        bh.naccept[s] = 0;
        nstep = 0;
    }
    bh.nstep[s] = nstep;
    const double lo = -step, range = step - lo;             // Generator.uniform(low, high): low + (high - low) * next_double
    const double* u = uniforms + (s * (int64_t)niter + hop) * (N + 1);
    for (int k = 0; k < N; ++k) trial[s * N + k] = bh.x_cur[s * N + k] + (lo + rn(range * u[k]));
}

__global__ __launch_bounds__(256)
void bh_result_kernel(BhState bh, double* __restrict__ x, double* __restrict__ llh) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= bh.S) return;
    for (int k = 0; k < bh.N; ++k) x[s * bh.N + k] = bh.x_best[s * bh.N + k];
    llh[s] = -bh.f_best[s];
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
hipError_t launch_nm_spec_points(const NmState& st, int64_t bound, hipStream_t stream) {
    hipLaunchKernelGGL(nm_spec_points_kernel, nm_grid(bound), dim3(256), 0, stream, st, bound);
    return hipGetLastError();
}
hipError_t launch_nm_spec_step(const NmState& st, const NmState& nx, int64_t bound, const double* llk, int32_t* live_host, hipStream_t stream) {
    if (bound > 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nm_spec_step_kernel, dim3(1), dim3(256), 0, stream, st, nx, bound, llk, live_host);
    return hipGetLastError();
}
hipError_t launch_nm_spec_finish(const NmState& st, int64_t bound, const double* llk, hipStream_t stream) {
    hipLaunchKernelGGL(nm_spec_finish_kernel, nm_grid(bound), dim3(256), 0, stream, st, bound, llk);
    return hipGetLastError();
}
hipError_t launch_bh_update(const BhState& bh, const NmState& nm, const double* x_min, const double* llh_min, const int32_t* nm_status,
                            int hop, int niter, const double* uniforms, hipStream_t stream) {
    hipLaunchKernelGGL(bh_update_kernel, nm_grid(bh.S), dim3(256), 0, stream, bh, nm, x_min, llh_min, nm_status, hop, niter, uniforms);
    return hipGetLastError();
}
hipError_t launch_bh_step(const BhState& bh, int hop, int niter, const double* uniforms, double* trial, hipStream_t stream) {
    hipLaunchKernelGGL(bh_step_kernel, nm_grid(bh.S), dim3(256), 0, stream, bh, hop, niter, uniforms, trial);
    return hipGetLastError();
}
hipError_t launch_bh_result(const BhState& bh, double* x, double* llh, hipStream_t stream) {
    hipLaunchKernelGGL(bh_result_kernel, nm_grid(bh.S), dim3(256), 0, stream, bh, x, llh);
    return hipGetLastError();
}

}  // namespace misti

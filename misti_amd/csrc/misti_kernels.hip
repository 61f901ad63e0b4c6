// MI355X (gfx950) kernels of the MiSTI composite-likelihood engine.
//
// A batch of candidates (split time, band rates, pulse rates) x bootstrap JSFS replicates is four
// launches:
//   setup_kernel            candidates with identical parameters share a CHAIN (hash table); dispatch order; llh_const of the
//                           replicates; the other (double-buffered) chain table cleared for the next batch
//   correct_[follow_]kernel lambda-correction of the chains, a resumable state machine per chain;
//                           with one chain per wave a second wave of the workgroup builds the
//                           chain's TRUNK (44-state propagation shared by its candidates) behind it
//   post_kernel             trunks that did not follow + tails (fractional splits)
//   spectrum_kernel         one wavefront per candidate, the 44-state vector one state per lane:
//                           own intervals after the trunk, collapse, one-population closed form,
//                           normalisation, llk of up to 8 replicates  (llk_kernel beyond that)
// fp64 throughout, no MFMA (the per-interval matrices are 3x3 / 44x44 with ~200 non-zeros).
//
// Reference path restated here (cites: /root/reference):
//   JAFSLikelihood  MigrationInference.py:566-614     driver, status codes
//   CorrectLambdas  MigrationInference.py:305-378     -> correct_body(), post-split rates in spectrum_kernel
//   CorrectLambda   CorrectLambda.py:29-317           -> pair chain: pair_expv(), pair_batch(), next_step(), trf_bounded()
//   Smooth          MigrationInference.py:380-405     -> smooth_rates(), run_mean()
//   JAFSpectrum     MigrationInference.py:467-540     -> twopop_interval(), trunk_body()/trunk_follow(), spectrum_kernel
//   TwoPopulations / OnePopulation                     -> tables (misti_tables.hpp) + closed form
//   CoalescentRates MigrationInference.py:542-564     -> forward_kernel
//
// Numerical method (differs from the reference's dense Pade expm + inverse, same
// mathematics): exp(M T) P0 and the occupation integral  int_0^T exp(M t) P0 dt
// (= M^-1 (P1 - P0), MigrationInference.py:538-540) are evaluated together as an
// action on the vector by uniformisation: M T = N - q I with N >= 0, and the
// augmented series  p_{k+1} = N p_k/(k+1),  i_{k+1} = (T p_k + q i_k)/(k+1)  whose
// sums are P1 and the integral.  All terms are non-negative, so there is no
// cancellation, no inverse, and the mu = 0 generator (singular: the reference
// deletes 7 stationary states and restores their mass, TwoPopulations.py:231-309)
// needs no special case.  After the split the chain is the Kingman coalescent in
// rescaled time, whose 8x8 generator has eigenvalues -6,-3,-1: closed form.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>


#include "misti_device.h"

namespace misti {

__constant__ DevTables c_tab;
__constant__ double c_inv[INV_TABLE];   // c_inv[k] = 1/k: series terms divide by k (an fp64 division costs ~40 instructions)
// Terms of the uniformisation series as a function of q = (largest exit rate) x (interval length): the series stops after term K,
// the first k with e^-q q^(k-1) / (k-1)! < 1e-19 and k > q.  c_qmax[K] = the largest q for which K terms suffice (increasing in K;
// c_qmax[1] < 0: never one term), so K(q) = 1 + #{ j >= 1 : c_qmax[j] < q } - a compare, a ballot and a population count per 64
// entries instead of a running bound, a conversion and two compares inside every term (twopop_interval).
constexpr int QMAX_TABLE = 320;         // Q_SWITCH = 96 needs K <= 240
__constant__ double c_qmax[QMAX_TABLE];

// exp(M T) P0 and the occupation integral per interval, two methods:
//  * q = (largest exit rate) x (interval length) <= Q_SWITCH: uniformisation series
//    (cost ~ q + 8 sqrt(q) + 10 sparse mat-vecs, all terms non-negative);
//  * q > Q_SWITCH (a runaway corrected rate; the reference's dense Pade expm handles it by
//    squaring): Talbot contour quadrature  exp(A) v = sum_k -c_k (z_k - A)^-1 v  on the
//    "modified Talbot" cotangent contour of Trefethen, Weideman & Schmelzer (BIT 46, 2006),
//    N = 28 nodes (14 conjugate pairs; truncation 3.89^-N, measured error < 1e-15 here),
//    each shifted system solved by Jacobi sweeps with the same sparse row gather (the
//    coalescence part of the generator is nilpotent, so the sweeps terminate in <= 7 steps
//    unless migration is strong in both directions).  Cost is independent of q.
constexpr double Q_SWITCH = 96.0;
constexpr int JACOBI_MAX = 600;
constexpr long long FOLLOW_SPIN_LIMIT = 1LL << 22;   // polls (~1 us each) before a following trunk wave gives up

// ---------------------------------------------------------------- helpers ----
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// broadcast from a WAVE-UNIFORM source lane: scalar readlane, no LDS crossbar round trip
__device__ __forceinline__ double bcast(double v, int src_lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// The correction kernel packs GROUP (8, 16, 32 or 64) lanes per candidate, 6 of which carry the
// residual evaluations (3 forward-difference points x 2 genomes): 8 ... 1 candidates per wavefront,
// chosen at launch from the batch size (few candidates per wave while the chip is not full).
// Everything "uniform" is uniform within a group; cross-lane traffic never leaves a group.
template <int GROUP>
__device__ __forceinline__ double gbcast(double v, int j) {
    if (GROUP == 64) {                       // one candidate per wave: scalar broadcast (j wave-uniform), no LDS crossbar round trip
        const int ju = __builtin_amdgcn_readfirstlane(j);
        int lo = __builtin_amdgcn_readlane(__double2loint(v), ju);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), ju);
        return __hiloint2double(hi, lo);
    }
    return __shfl(v, (lane_id() - lane_id() % GROUP) + j, 64);     // GROUP = 6: ten items per wave, lanes 60-63 idle
}

// Hand-over words in LDS between the two waves of a workgroup (correct_follow_kernel).  The pointers reach the
// device functions as generic pointers and a volatile access through a generic pointer is a FLAT instruction with
// system-scope cache bits followed by a wait for every outstanding global store; through an LDS-typed pointer it is a
// plain ds_read / ds_write.  Ordering between the two waves ("data, then count" on the writer's side, "count, then data" on the
// reader's) is a workgroup-scope release / acquire fence restricted to the LDS address space: it compiles to s_waitcnt lgkmcnt(0)
// - no wait for the outstanding global stores of the chain - and is what the AMDGPU memory model asks for between waves (a
// wavefront-scope fence orders nothing across waves and merely happened to work; the build never uses -mtgsplit).
typedef __attribute__((address_space(3))) int lds_i32_t;
__device__ __forceinline__ void lds_put(volatile int* p, int v) { *(volatile lds_i32_t*)(lds_i32_t*)p = v; }
__device__ __forceinline__ int lds_get(const volatile int* p) { return *(const volatile lds_i32_t*)(lds_i32_t*)p; }
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// A branch condition that is the same in every lane - everything of a chain is, when the chain has the wave to itself
// (GROUP == 64) - handed to the compiler as a SCALAR: it then branches with s_cbranch_scc instead of masking lanes off
// (s_and_saveexec / s_or exec around every block, and copies of all loop-carried state at every divergent loop edge:
// that plumbing was about a fifth of the correction kernel's instructions).  With several chains per wave the condition
// really differs between lanes and is left alone.
template <int GROUP>
__device__ __forceinline__ bool uni(bool c) { return GROUP == 64 ? (bool)__builtin_amdgcn_readfirstlane((int)c) : c; }

__device__ __forceinline__ void lds_fence() {
    // one wave owns its LDS slice: ordering only has to be kept by the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-candidate view of the interval grid.  A fractional split time splits
// interval s = floor(st) in two and moves the split index to s+1
// (MigrationInference.__init__, MigrationInference.py:89-99).
struct Grid {
    const double* times;   // [numT0-1]
    const double* lh;      // [numT0][2]
    int numT0;             // intervals of the shared grid
    int numT;              // intervals of this candidate (numT0 or numT0+1)
    int split;             // split index of this candidate
    int ins;               // index of the interval that was split in two, or -1
    double frac;

    __device__ __forceinline__ int src(int t) const { return (ins >= 0 && t > ins) ? t - 1 : t; }
    __device__ __forceinline__ double T(int t) const {
        // t in [0, numT-1)
        if (ins < 0) return times[t];
        if (t < ins) return times[t];
        double whole = times[ins];
        double t1 = frac * whole;
        if (t == ins) return t1;
        if (t == ins + 1) return whole - t1;
        return times[t - 1];
    }
    __device__ __forceinline__ double lhk(int t, int k) const { return lh[2 * src(t) + k]; }
};

// Migration rates / pulse of interval t for this candidate
// (SetModel + MapParameters, MigrationInference.py:229-298).
struct Model {
    const DevModel* m;
    const double* par;     // [n_param] of this candidate
    int split;
    double pv[4];          // the first parameters, cached in registers (a sweep has 1-3)
    const int32_t* bb = nullptr;   // [n_band][2] this candidate's (start, end) of every band, or NULL: the model's
    __device__ __forceinline__ void cache() { for (int i = 0; i < 4; ++i) pv[i] = (i < m->n_param) ? par[i] : 0.0; }
    __device__ __forceinline__ double param(int i) const {
        return i == 0 ? pv[0] : i == 1 ? pv[1] : i == 2 ? pv[2] : i == 3 ? pv[3] : par[i];
    }
    __device__ __forceinline__ void mig(int t, double& mu0, double& mu1) const {
        mu0 = 0.0; mu1 = 0.0;
        for (int b = 0; b < m->n_band; ++b) {
            const misti_band_t& B = m->bands[b];
            const int start = bb ? bb[2 * b] : B.start;
            int end = bb ? bb[2 * b + 1] : B.end;
            if (end < 0) end = split;
            if (t >= start && t < end) {
                double v = B.param >= 0 ? param(B.param) : B.value;
                if (B.pop == 0) mu0 = v; else mu1 = v;
            }
        }
    }
    __device__ __forceinline__ void pulse(int t, double& pu0, double& pu1) const {
        pu0 = 0.0; pu1 = 0.0;
        for (int b = 0; b < m->n_pulse; ++b) {
            const misti_pulse_t& P = m->pulses[b];
            if (t == P.time) {
                double v = P.param >= 0 ? param(P.param) : P.value;
                if (P.pop == 0) pu0 = v; else pu1 = v;
            }
        }
    }
};

// ------------------------------------------------------------ pair chain ----
// Three states of one genome's two lineages: both in pop 0, both in pop 1, one
// in each (CorrectLambda.SetMatrix, CorrectLambda.py:55-56):
//     [ -2mu0-l0     0       mu1     ]
//     [    0      -2mu1-l1   mu0     ]
//     [  2mu0      2mu1    -mu0-mu1  ]
// v <- exp(M) v by uniformisation.  Every lane carries its own (l, v); the trip
// count is wave-uniform (bound from the largest q in the wave).
// Per-candidate diagnostics of the correction: overflow guard and work counters.
// Work counters of a chain (last row of the `pr` output).  evals, max_nfev, lm and spec are wave-uniform (scalar registers) and always
// counted; dense, terms and squarings are counted INSIDE the per-lane evaluation and cost three vector registers carried through the
// whole solver loop - exactly what pushed correct_follow_kernel<true> into scratch (10 spilled VGPRs, 44 B; round 4) - so they are
// counted in the diagnostic build only (-DMISTI_WORK_COUNTERS=1: tools/stamp_run.py) and read 0 otherwise.
#ifndef MISTI_WORK_COUNTERS
#define MISTI_WORK_COUNTERS 0
#endif
struct Diag { bool guard = false; int evals = 0, dense = 0, terms = 0, squarings = 0, max_nfev = 0, lm = 0, spec = 0; };
// K terms of the Taylor series of exp(M) v for the pair generator, fully unrolled.
// INT: also  vint = sum_k (M^k v / k!) / (k + 2)  =  int_0^1 u exp(u M) v du  - the default fit's expected coalescence
// time needs  M^-1 exp(M) v - M^-2 (exp(M) - I) v  (CorrectLambda.py:99-107), which is exactly that integral: the series
// has no inverse and no cancellation, where the reference's formula loses ~1/|M|^2 digits, and like the exponential it
// acts on the vector only, so a decoupled component still sees identical arithmetic in every forward-difference lane
// (the reference's finite-difference Jacobian has an exactly zero entry there; a 3x3 solve by cofactors does not keep it).
#ifndef MISTI_TAYLOR_LEAN
#define MISTI_TAYLOR_LEAN 0       // 1: EXPERIMENT (round 6; never the shipped build): Horner form of the --cpfit series, see taylor3_horner
#endif
#if MISTI_TAYLOR_LEAN
// exp(M) v by Horner's rule, w <- v + (M w) / k for k = K .. 1: ten fp64 operations per term where the term-by-term form below needs thirteen (no running sum, the
// reciprocal folded into the update).  Same accuracy class (the truncation bound is the same; rounding a few ulps either way) - and every bit of every evaluation
// different, which re-draws the reference's coin flips of DESIGN.md section 2: measured as a variant build against the whole suite, not shipped.
template <int K>
__device__ __forceinline__ void taylor3_horner(double d0, double d1, double d2, double mu0, double mu1, double v[3]) {
    const double v0 = v[0], v1 = v[1], v2 = v[2];
    const double twomu0 = 2.0 * mu0, twomu1 = 2.0 * mu1;
    double w0 = v0, w1 = v1, w2 = v2;
#pragma unroll
    for (int k = K; k >= 1; --k) {
        const double inv = 1.0 / (double)k;          // a literal after unrolling
        const double r0 = mu1 * w2 - d0 * w0;
        const double r1 = mu0 * w2 - d1 * w1;
        const double r2 = (twomu0 * w0 + twomu1 * w1) - d2 * w2;
        w0 = fma(r0, inv, v0); w1 = fma(r1, inv, v1); w2 = fma(r2, inv, v2);
    }
    v[0] = w0; v[1] = w1; v[2] = w2;
}
#endif
template <int K, bool INT>
__device__ __forceinline__ void taylor3(double d0, double d1, double d2, double mu0, double mu1, double v[3], Diag& dg, double vint[3]) {
#if MISTI_TAYLOR_LEAN
    if (!INT) { taylor3_horner<K>(d0, d1, d2, mu0, mu1, v); return; }
#endif
    double p0 = v[0], p1 = v[1], p2 = v[2];
    double a0 = p0, a1 = p1, a2 = p2;
    double b0 = 0.5 * p0, b1 = 0.5 * p1, b2 = 0.5 * p2;
    const double twomu0 = 2.0 * mu0, twomu1 = 2.0 * mu1;
#pragma unroll
    for (int k = 1; k <= K; ++k) {
        const double inv = 1.0 / (double)k;          // a literal after unrolling
        const double t0 = (mu1 * p2 - d0 * p0) * inv;
        const double t1 = (mu0 * p2 - d1 * p1) * inv;
        const double t2 = ((twomu0 * p0 + twomu1 * p1) - d2 * p2) * inv;
        p0 = t0; p1 = t1; p2 = t2;
        a0 += p0; a1 += p1; a2 += p2;
        if (INT) { const double w2 = 1.0 / (double)(k + 2); b0 = fma(p0, w2, b0); b1 = fma(p1, w2, b1); b2 = fma(p2, w2, b2); }
    }
#if MISTI_WORK_COUNTERS
    dg.terms += K;
#endif
    v[0] = a0; v[1] = a1; v[2] = a2;
    if (INT) { vint[0] = b0; vint[1] = b1; vint[2] = b2; }
}

// Reciprocal / square root for the solver's bookkeeping: hardware seed + Newton steps, ~1 ulp,
// a third of the instructions of the correctly rounded forms (no denormal/overflow scaling:
// every operand here is a normal-range quantity or the result is tested for finiteness anyway).
__device__ __forceinline__ double rcp64(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double sqrt64(double x) {
    // straight-line: the solver's bookkeeping is one dependent instruction stream, every branch costs it a handful of
    // scalar instructions.  rsq(0) = inf and rsq(inf) = 0 make the Newton steps NaN: those two inputs are passed through;
    // negative and NaN inputs give NaN by themselves (rsq), as sqrt() does.
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double d = fma(-g, g, x);
    const double s = fma(d, h, g);
    return (x == 0.0 || x == INFINITY) ? x : s;
}

// Structural reduction of the pair chain.  With migration in one direction only (mu1 == 0: nothing enters "both in
// population 0") and that state empty in both genomes - it is, exactly, from the first interval in which its rate ran
// away: exp(-rate x length) underflows or falls below PAIR_EMPTY of the total - the three-state chain is the two-state one on (both in population 1, one in each), whose
// generator [[-d1, mu0], [0, -d2]] is triangular:
//     w1 = e^{-d1} v1 + mu0 phi v2,   w2 = e^{-d2} v2,   phi = (e^{-d1} - e^{-d2}) / (d2 - d1),   w0 = 0,
// three exponentials instead of a 30-45 term series or a dense scaling-and-squaring on a matrix made stiff by a rate
// (l0) that no longer enters the result.  This is exactly the solver's rank-one regime (correct_body): the Jacobian
// column of l0 is zero because l0 is not read.  Mirror case (which = 2): mu0 == 0 and "both in population 1" empty.
constexpr double PAIR_EMPTY = 1e-30;
__device__ __forceinline__ void pair_reduced(int which, double l0, double l1, double mu0, double mu1, double v[3], bool ok) {
    if (!ok) { l0 = 0.0; l1 = 0.0; }
    const double dk = which == 1 ? 2.0 * mu1 + l1 : 2.0 * mu0 + l0;      // exit rate of the state that is left
    const double d2 = mu0 + mu1;
    const double mu = which == 1 ? mu0 : mu1;                             // "one in each" -> that state
    const double vk = which == 1 ? v[1] : v[0];
    const double a = -dk, c = -d2;
    const double ax = fabs(a - c);
    const double phi = exp(fmax(a, c)) * (ax == 0.0 ? 1.0 : -expm1(-ax) / ax);
    const double wk = exp(a) * vk + mu * phi * v[2];
    const double w2 = exp(c) * v[2];
    v[0] = which == 1 ? 0.0 : wk;
    v[1] = which == 1 ? wk : 0.0;
    v[2] = w2;
    if (!ok) { v[0] = v[1] = v[2] = NAN; }
}

// exp(M) v for migration in one direction only, in closed form.  which = 1: mu1 == 0 - nothing enters "both in population 0"
// (state 0); it empties at a = 2 mu0 + l0 into "one in each" (state 2, at 2 mu0), which empties at b = mu0 into "both in
// population 1" (state 1, at mu0), which coalesces at c = l1:
//     w0 = e^-a v0,    w2 = e^-b v2 + 2 mu0 D(a, b) v0,    w1 = e^-c v1 + mu0 D(b, c) v2 + 2 mu0^2 D(a, b, c) v0
// with D the divided differences of e^{-x}:  D(x, y) = (e^-x - e^-y) / (y - x),  D(x, y, z) = (D(x, y) - D(y, z)) / (z - x).
// which = 2 is the mirror image (mu0 == 0).  Used where the generator is stiff (a rate has run away), so the largest gap
// among the three rates is large and the second difference is taken across it; D(x, y) itself goes through expm1.
__device__ __forceinline__ double dd_exp(double x, double y) {
    const double g = fabs(x - y);
    return exp(-fmin(x, y)) * (g == 0.0 ? 1.0 : -expm1(-g) / g);
}
__device__ __forceinline__ void pair_cascade(int which, double l0, double l1, double mu0, double mu1, double v[3]) {
    const double mu = which == 1 ? mu0 : mu1;
    const double a = which == 1 ? 2.0 * mu0 + l0 : 2.0 * mu1 + l1;      // exit rate of the state that is left (S)
    const double c = which == 1 ? 2.0 * mu1 + l1 : 2.0 * mu0 + l0;      // ... of the state that is entered (K)
    const double b = mu0 + mu1;                                          // ... of "one in each"
    const double vS = which == 1 ? v[0] : v[1], vK = which == 1 ? v[1] : v[0], v2 = v[2];
    // second divided difference across the largest gap: sort the three rates
    double s0 = a, s1 = b, s2 = c;
    if (s0 > s1) { const double t = s0; s0 = s1; s1 = t; }
    if (s1 > s2) { const double t = s1; s1 = s2; s2 = t; }
    if (s0 > s1) { const double t = s0; s0 = s1; s1 = t; }
    const double gap = s2 - s0;
    const double D3 = gap > 0.0 ? (dd_exp(s0, s1) - dd_exp(s1, s2)) / gap : 0.5 * exp(-s0);
    const double wS = exp(-a) * vS;
    const double w2 = exp(-b) * v2 + (2.0 * mu) * dd_exp(a, b) * vS;
    const double wK = exp(-c) * vK + mu * dd_exp(b, c) * v2 + (2.0 * mu * mu) * D3 * vS;
    v[0] = which == 1 ? wS : wK;
    v[1] = which == 1 ? wK : wS;
    v[2] = w2;
}

// exp(M) v for migration in BOTH directions, in closed form (round 5; until then the stiff two-way generator went through a dense
// degree-12 Taylor kernel and up to 17 squarings: ~2 000 instructions per evaluation, a systematic relative error of ~1e-11 at
// |M| = 1e5, and the floor of every packed launch - the resumed chains of BASELINE configs 3 and 5 spend their time there).
// In the order (both in 0, one in each, both in 1) the generator is tridiagonal with positive off-diagonal products, hence similar to
// the symmetric  [[-d0, s, 0], [s, -d2, s], [0, s, -d1]],  s^2 = 2 mu0 mu1:  three real eigenvalues -x_i, the roots of the secular
// equation   (d2 - x) = s^2 / (d0 - x) + s^2 / (d1 - x),   one below both poles d0, d1, one between them, one above, and
//     exp(M) v = sum_i e^{-x_i} r_i (l_i . v) / (l_i . r_i),     r_i = (mu1 / g0, mu0 / g1, 1),  l_i = (2 mu0 / g0, 2 mu1 / g1, 1),
//     g0 = d0 - x_i,  g1 = d1 - x_i,   l_i . r_i = 1 + s^2 / g0^2 + s^2 / g1^2
// (rows / columns 0 and 1 of (M + x) r = 0; no square root, no matrix product).  What decides the accuracy is the GAPS g0, g1, so
// every root is found in the variable "distance to the nearest pole" (as LAPACK's dlaed4 does for its secular equation): for the
// outer roots  F(t) = c + t - s^2/t - s^2/(G + t),  for the middle one  F(t) = c + t - s^2/t + s^2/(G - t)  on (0, G/2]
// (G = |d0 - d1|; c = the distance of d2 from that pole, signed) - increasing and concave, so Newton's iteration from a point left of the
// root converges monotonically, without safeguards, and two such points come from quadratics (a function above F with a computable
// root: the far pole dropped, or the near pole moved to the far one).  The smallest root when d2 lies below both poles is found as
// d2 - e instead (K(e) = e - s^2/(A0 + e) - s^2/(A1 + e), A = d - d2 > 0): the runaway rates make the poles huge and the root stays at
// d2, where a distance to a pole of 1e5 would lose eleven digits of e^{-x}.  Newton's iteration reaches rounding from the better of the two
// starts in 1 - 3 steps per root over 6 000 random generators spanning rates 1e-2 ... 3e5 and migration 1e-6 ... 3 (scratch prototype against 60-digit
// arithmetic: 7e-16 of the norm of the result in the stiff regime; the eigenvector scaling costs a factor sqrt(mu1 / mu0) where the two
// migration rates differ by orders of magnitude).  Smooth in the varied rate to a few ulps, like pair_cascade.
__device__ __forceinline__ double pe_qroot(double a, double c, double s2) {          // positive root of a t^2 + c t - s2 = 0
    const double r = sqrt64(c * c + 4.0 * a * s2);
    return c >= 0.0 ? 2.0 * s2 * rcp64(c + r) : (r - c) * rcp64(2.0 * a);
}
// Newton's steps stop for the whole wave once every lane's last correction is below PE_TOL of its root: the iteration converges
// quadratically from the left, so the correction after that one is below rounding (prototype: 1e-14 of the result's norm at worst,
// 4.4 Newton steps per evaluation for all three roots together where a fixed count needs 18).
constexpr int PE_NEWTON_MAX = 12;
constexpr double PE_TOL = 1e-9;
__device__ __forceinline__ double pe_outer(double c, double G, double s2) {
    // left starts: the far pole dropped (c + t - s2/t), or the near pole moved onto the far one (c + t - 2 s2/(G + t): t^2 + (c + G) t = 2 s2 - c G)
    const double cg = c + G, r = sqrt64((c - G) * (c - G) + 8.0 * s2);
    const double t2 = cg >= 0.0 ? 2.0 * (2.0 * s2 - c * G) * rcp64(cg + r) : 0.5 * (r - cg);
    double t = fmax(pe_qroot(1.0, c, s2), t2);
    bool done = false;                    // a lane's root is frozen by ITS OWN test: what else sits in the wave changes no bit of it
    for (int it = 0; it < PE_NEWTON_MAX; ++it) {
        const double a = rcp64(t), b = rcp64(G + t);
        const double F = ((c + t) - s2 * a) - s2 * b, dF = 1.0 + s2 * (a * a + b * b);
        const double d = F * rcp64(dF);
        t = done ? t : t - d;
        done = done || !(fabs(d) > PE_TOL * t);
        if (!__any(!done)) break;
    }
    return t;
}
__device__ __forceinline__ double pe_middle(double c, double G, double s2) {
    const double iG = rcp64(G);
    double t = fmax(pe_qroot(1.0, c + 2.0 * s2 * iG, s2), pe_qroot(1.0 + 2.0 * s2 * iG * iG, c + s2 * iG, s2));
    bool done = false;
    for (int it = 0; it < PE_NEWTON_MAX; ++it) {
        const double a = rcp64(t), b = rcp64(G - t);
        const double F = ((c + t) - s2 * a) + s2 * b, dF = 1.0 + s2 * (a * a + b * b);
        const double d = F * rcp64(dF);
        t = done ? t : t - d;
        done = done || !(fabs(d) > PE_TOL * t);
        if (!__any(!done)) break;
    }
    return t;
}
__device__ __forceinline__ double pe_below(double A0, double A1, double s2) {
    double e = fmax(pe_qroot(1.0, fmin(A0, A1), s2), pe_qroot(1.0, fmax(A0, A1), 2.0 * s2));
    bool done = false;
    for (int it = 0; it < PE_NEWTON_MAX; ++it) {
        const double a = rcp64(A0 + e), b = rcp64(A1 + e);
        const double K = (e - s2 * a) - s2 * b, dK = 1.0 + s2 * (a * a + b * b);
        const double d = K * rcp64(dK);
        e = done ? e : e - d;
        done = done || !(fabs(d) > PE_TOL * e);
        if (!__any(!done)) break;
    }
    return e;
}
__device__ __forceinline__ void pair_eigen(double l0, double l1, double mu0, double mu1, double v[3]) {
    const double d0 = 2.0 * mu0 + l0, d1 = 2.0 * mu1 + l1, d2 = mu0 + mu1, s2 = 2.0 * mu0 * mu1;
    const bool low0 = d0 <= d1;                          // which pole is the lower one
    const double p = low0 ? d0 : d1, P = low0 ? d1 : d0, G = P - p;
    // per root: x and the gaps (p - x, P - x)
    double x[3], gp[3], gP[3];
    if (d2 < p) { const double e = pe_below(p - d2, P - d2, s2); x[0] = d2 - e; gp[0] = (p - d2) + e; gP[0] = (P - d2) + e; }
    else { const double t = pe_outer(d2 - p, G, s2); x[0] = p - t; gp[0] = t; gP[0] = G + t; }
    const bool hi = d2 > p + 0.5 * G;                    // the middle root lies in the half of (p, P) on d2's side of the midpoint
    {
        const double t = G > 0.0 ? pe_middle(hi ? d2 - P : p - d2, G, s2) : 0.0;
        x[1] = hi ? P - t : p + t; gp[1] = hi ? -(G - t) : -t; gP[1] = hi ? t : G - t;
    }
    // the largest root lies above the upper pole: with a runaway rate its e^{-x} is zero beside the smallest root's - not computed then
    // (a per-lane decision: lanes that need it compute it, and a lane's result never depends on its company in the wave)
    const bool third = P - x[0] <= 745.0;
    x[2] = 0.0; gp[2] = -1.0; gP[2] = -1.0;        // placeholders of a lane without a third term (its weight below is an exact zero)
    if (third) { const double t = pe_outer(P - d2, G, s2); x[2] = P + t; gp[2] = -(G + t); gP[2] = -t; }
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i == 2 && !__any(third)) break;
        const double ex = (i == 2 && !third) ? 0.0 : exp(-x[i]);
        const double i0 = rcp64(low0 ? gp[i] : gP[i]), i1 = rcp64(low0 ? gP[i] : gp[i]);        // 1 / (d0 - x), 1 / (d1 - x)
        double cf = (((2.0 * mu0) * v[0]) * i0 + ((2.0 * mu1) * v[1]) * i1 + v[2]) * rcp64(1.0 + s2 * (i0 * i0 + i1 * i1));
        double r0 = mu1 * i0, r1 = mu0 * i1, r2 = 1.0;
        if (i == 1 && !(G > 0.0)) {
            // coinciding poles: the middle eigenvalue IS the pole, right vector (mu1, -mu0, 0), left (mu0, -mu1, 0), l . r = s^2
            cf = (mu0 * v[0] - mu1 * v[1]) * rcp64(s2); r0 = mu1; r1 = -mu0; r2 = 0.0;
        }
        const double z = ex * cf;
        w0 = fma(z, r0, w0); w1 = fma(z, r1, w1); w2 = fma(z, r2, w2);
    }
    v[0] = w0; v[1] = w1; v[2] = w2;
}

// q, neg: the SAME for every lane of the candidate's group (computed by the caller from the
// base point and both forward-difference points).  M = N - q I with N >= 0.  When a state is
// numerically decoupled (e.g. no mass and no inflow in "both in pop 0" after a runaway rate
// with one-directional migration) the forward-difference lanes then perform bit-identical
// arithmetic on the remaining components, so the Jacobian column is exactly zero - as it is in
// the reference, whose solver leaves that rate untouched.
// INT: with the series path taken, vint receives int_0^1 u exp(u M) v du and have_int is set (see taylor3); the stiff paths
// leave it unset (there |M| > 1 and the reference's inverse-based formula is well conditioned: the caller uses that).
template <bool INT = false>
__device__ __forceinline__ void pair_expv(double l0, double l1, double mu0, double mu1, double v[3], double q, double neg, bool ok, Diag& dg, bool& guard,
                                          double* vint = nullptr, bool* have_int = nullptr) {
    if (!ok) { l0 = 0.0; l1 = 0.0; }
    double d0 = 2.0 * mu0 + l0, d1 = 2.0 * mu1 + l1, d2 = mu0 + mu1;
    double nbmax = q + neg;                                  // >= ||N||_1 (column sums q - l0, q - l1, q)
    if (!(nbmax < 1e300)) {                                  // overflowing iterate: report non-finite (TRF shrinks the step)
        guard = true;
        v[0] = v[1] = v[2] = NAN;
        return;
    }
    if (nbmax > 6.0) {
        // stiff iterate (the unbounded solver can run l up to ~1e5 when the gradient
        // vanishes): dense scaling and squaring of the 3x3 matrix, degree-12 Taylor kernel
        int sq = 0;
        { double nrm = 2.0 * nbmax; while (nrm > 0.25) { nrm *= 0.5; ++sq; } }
#if MISTI_WORK_COUNTERS
        dg.dense += 1; dg.squarings += sq;
#endif
        double scl = ldexp(1.0, -sq);
        if (mu1 == 0.0 || mu0 == 0.0) {
            // Migration in one direction only: the pair chain is a cascade S -> "one in each" -> K (S = both in the population that
            // is left, K = both in the other) and exp(M) is its closed form in divided differences of e^{-x} over the three exit
            // rates (pair_cascade).  A runaway rate makes M stiff but not this form: a few ulps at any norm, and smooth in the rate
            // that is varied, so the forward-difference Jacobian of a saturated residual keeps its digits - the trust region's gain
            // ratio there sits within 5e-4 of SciPy's 0.75 threshold step after step (x / (x + p) for a residual ~ 1/x), and the
            // scaled-and-squared Taylor kernel used here before (13 squarings at rate x length 700) put it on the wrong side on the
            // headline grid: one chain of 64, 2e-6 ... 4e-6 in the likelihood where the reference holds 1e-10.
            pair_cascade(mu1 == 0.0 ? 1 : 2, l0, l1, mu0, mu1, v);
#if MISTI_WORK_COUNTERS
            dg.dense += 1;
#endif
            if (!ok) { v[0] = v[1] = v[2] = NAN; }
            return;
        }
#ifndef MISTI_PAIR_EIGEN
#define MISTI_PAIR_EIGEN 1
#endif
#if MISTI_PAIR_EIGEN
        // migration in both directions: the closed form over the three eigenvalues of the (symmetrisable) generator (pair_eigen)
        pair_eigen(l0, l1, mu0, mu1, v);
        if (!ok) { v[0] = v[1] = v[2] = NAN; }
        return;
#endif
        double B[3][3] = {{-d0 * scl, 0.0, mu1 * scl}, {0.0, -d1 * scl, mu0 * scl}, {2.0 * mu0 * scl, 2.0 * mu1 * scl, -d2 * scl}};
        // Horner, E = I + B E / k for k = 12 .. 1.  The first step (E = I) is B / 12 + I; B[0][1] = B[1][0] = 0 are left out
        // of the products - with the fused forms spelt out both give the bits of the full 3x3 product for finite entries.
        double E[3][3];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { const double b12 = B[r][c] * c_inv[12]; E[r][c] = b12 + (r == c ? 1.0 : 0.0); }   // two roundings, as in the loop
        for (int k = 11; k >= 1; --k) {
            double inv = c_inv[k];
            double Tm[3][3];
            for (int c = 0; c < 3; ++c) {
                Tm[0][c] = fma(B[0][2], E[2][c], B[0][0] * E[0][c]) * inv;
                Tm[1][c] = fma(B[1][2], E[2][c], B[1][1] * E[1][c]) * inv;
                Tm[2][c] = ((B[2][0] * E[0][c] + B[2][1] * E[1][c]) + B[2][2] * E[2][c]) * inv;
            }
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) E[r][c] = Tm[r][c] + (r == c ? 1.0 : 0.0);
        }
        for (int i = 0; i < sq; ++i) {
            double Tm[3][3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c)
                Tm[r][c] = (E[r][0] * E[0][c] + E[r][1] * E[1][c]) + E[r][2] * E[2][c];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) E[r][c] = Tm[r][c];
        }
        double w0 = (E[0][0] * v[0] + E[0][1] * v[1]) + E[0][2] * v[2];
        double w1 = (E[1][0] * v[0] + E[1][1] * v[1]) + E[1][2] * v[2];
        double w2 = (E[2][0] * v[0] + E[2][1] * v[1]) + E[2][2] * v[2];
        v[0] = w0; v[1] = w1; v[2] = w2;
        if (!ok) { v[0] = v[1] = v[2] = NAN; }
        return;
    }
    if (2.0 * nbmax <= 2.0) {
        // small norm (the usual case: rate x interval length << 1): plain Taylor series of exp(M) v,
        // straight-line code with a degree fixed by the norm class ((2 nb)^K / K! < 1e-19) and literal
        // reciprocals.  No shift, no exp(); cancellation is bounded by e^2 ulp, and a decoupled
        // component again sees identical arithmetic in every forward-difference lane.
        const double nn = 2.0 * nbmax;                // >= ||M||_1
#if MISTI_TAYLOR_LEAN == 2
        // ... and degrees for nn^K / K! < 1e-17 (one ulp of a state vector of order one is 1.1e-16) instead of 1e-19: one or two terms fewer per class
        if (nn <= 0.01) taylor3<7, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.04) taylor3<9, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.12) taylor3<11, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.25) taylor3<13, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.5) taylor3<15, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 1.0) taylor3<19, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else taylor3<25, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
#else
        if (nn <= 0.01) taylor3<8, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.04) taylor3<10, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.12) taylor3<12, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.25) taylor3<14, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 0.5) taylor3<17, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else if (nn <= 1.0) taylor3<21, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
        else taylor3<27, INT>(d0, d1, d2, mu0, mu1, v, dg, vint);
#endif
        if (INT) *have_int = true;
        if (!ok) { v[0] = v[1] = v[2] = NAN; }
        return;
    }
    const double qs = q, nbs = nbmax;
    double n00 = qs - d0, n11 = qs - d1, n22 = qs - d2;
    double n02 = mu1, n12 = mu0, n20 = 2.0 * mu0, n21 = 2.0 * mu1;
    double eq = exp(-qs);
    {
        double p0 = eq * v[0], p1 = eq * v[1], p2 = eq * v[2];
        double a0 = p0, a1 = p1, a2 = p2;
        double b = 1.0;                    // nbs^k / k! bounds the k-th term for every lane
        double inv_next = c_inv[1];
        for (int k = 1; k < 200; ++k) {
            const double inv = inv_next;
            inv_next = c_inv[k + 1];
            double t0 = (n00 * p0 + n02 * p2) * inv;
            double t1 = (n11 * p1 + n12 * p2) * inv;
            double t2 = (n20 * p0 + n21 * p1 + n22 * p2) * inv;
            p0 = t0; p1 = t1; p2 = t2;
            a0 += p0; a1 += p1; a2 += p2;
            b *= nbs * inv;
#if MISTI_WORK_COUNTERS
            dg.terms += 1;
#endif
            if (b < 1e-19 && (double)k > nbs) break;
        }
        v[0] = a0; v[1] = a1; v[2] = a2;
    }
    if (!ok) { v[0] = v[1] = v[2] = NAN; }
}

// 3x3 inverse times vector (for the default-fit residual, CorrectLambda.py:99-107)
__device__ __forceinline__ void solve3(const double M[3][3], const double b[3], double x[3]) {
    double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1];
    double c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2];
    double c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
    double det = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
    double id = 1.0 / det;
    double c10 = M[0][2] * M[2][1] - M[0][1] * M[2][2];
    double c11 = M[0][0] * M[2][2] - M[0][2] * M[2][0];
    double c12 = M[0][1] * M[2][0] - M[0][0] * M[2][1];
    double c20 = M[0][1] * M[1][2] - M[0][2] * M[1][1];
    double c21 = M[0][2] * M[1][0] - M[0][0] * M[1][2];
    double c22 = M[0][0] * M[1][1] - M[0][1] * M[1][0];
    x[0] = (c00 * b[0] + c10 * b[1] + c20 * b[2]) * id;
    x[1] = (c01 * b[0] + c11 * b[1] + c21 * b[2]) * id;
    x[2] = (c02 * b[0] + c12 * b[1] + c22 * b[2]) * id;
}

// 3x3 solve by Gaussian elimination with partial pivoting (the order of LAPACK's getrf/getrs, which is what
// scipy.linalg.inv does): unlike the cofactor form it never multiplies a structurally zero entry into a live one, so a
// component of the solution that does not depend on some matrix entry mathematically does not depend on it bitwise
// either - the reference's finite-difference Jacobian has exact zeros there and so must ours.
__device__ __forceinline__ void solve3_ge(const double Min[3][3], const double b[3], double x[3]) {
    double A[3][3], r[3];
    for (int i = 0; i < 3; ++i) { r[i] = b[i]; for (int j = 0; j < 3; ++j) A[i][j] = Min[i][j]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        // pivot: largest magnitude in column c at or below the diagonal (first one on ties, as idamax)
        int pv = c;
#pragma unroll
        for (int i = c + 1; i < 3; ++i) if (fabs(A[i][c]) > fabs(A[pv][c])) pv = i;
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            if (i == pv) {                                   // swap rows c and pv (selects: pv is data)
                for (int j = 0; j < 3; ++j) { const double t = A[c][j]; A[c][j] = A[i][j]; A[i][j] = t; }
                const double t = r[c]; r[c] = r[i]; r[i] = t;
            }
        }
        const double piv = 1.0 / A[c][c];                    // getf2 scales the column by the reciprocal of the pivot
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            const double l = A[i][c] * piv;
            for (int j = c + 1; j < 3; ++j) A[i][j] = A[i][j] - l * A[c][j];
            r[i] = r[i] - l * r[c];
        }
    }
    x[2] = r[2] / A[2][2];
    x[1] = (r[1] - A[1][2] * x[2]) / A[1][1];
    x[0] = ((r[0] - A[0][1] * x[1]) - A[0][2] * x[2]) / A[0][0];
}

// ---- the REFERENCE's form of the default fit's expected coalescence time, imitated operation by operation ----
// ExpectedCoalTimeTwoPop (CorrectLambda.py:94-110, T = 1 after the stretch):
//     MET = expm(M);  Minv = inv(M);  vec1 = Minv (Minv ((MET - I) pn));  vec2 = Minv (MET pn);  ect = l . (vec2 - vec1) / (1 - sum(MET pn))
// with expm as scipy computes it for small matrices - the Pade approximant of Al-Mohy & Higham (2009), order 3 / 5 / 7 / 9 by
// the 1-norm, (V - U) X = (V + U) solved by LU - and the inverse as an explicit matrix by LU with partial pivoting (LAPACK's
// getrf order).  NOT used for any value or step: its difference to the integral series is a draw of the rounding noise the
// reference's residual carries (ect_noise_continues).  What matters is the STRUCTURE - matrix functions first, vectors last: an
// entry of exp(M) that does not depend on the rates being varied keeps its rounding error from one forward-difference point to
// the next, as in the reference, so only the error's random part reaches the Jacobian; solving with the vector as right-hand
// side instead re-draws all of it (measured on 481 solves of the random campaign against the reference's own formula: noise
// width 2.2 x the reference's at the median, 1.0 ... 5 x between the deciles; vector solves 3 x, 1.2 ... 20 x).
__device__ __forceinline__ void mat3_mul(const double X[3][3], const double Y[3][3], double Z[3][3]) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Z[r][c] = (X[r][0] * Y[0][c] + X[r][1] * Y[1][c]) + X[r][2] * Y[2][c];
}
__device__ __forceinline__ void mat3_vec(const double X[3][3], const double v[3], double o[3]) {
    for (int r = 0; r < 3; ++r) o[r] = (X[r][0] * v[0] + X[r][1] * v[1]) + X[r][2] * v[2];
}
// X = A^-1 B for a 3 x 3 right-hand side: Gaussian elimination with partial pivoting (rows swapped by selects: the pivot is data)
__device__ __forceinline__ void mat3_solve(const double Ain[3][3], const double Bin[3][3], double X[3][3]) {
    double A[3][3], R[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { A[i][j] = Ain[i][j]; R[i][j] = Bin[i][j]; }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        int pv = c;
#pragma unroll
        for (int i = c + 1; i < 3; ++i) if (fabs(A[i][c]) > fabs(A[pv][c])) pv = i;
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            const bool sw = i == pv;
            for (int j = 0; j < 3; ++j) {
                const double a = A[c][j], b = A[i][j]; A[c][j] = sw ? b : a; A[i][j] = sw ? a : b;
                const double u = R[c][j], v = R[i][j]; R[c][j] = sw ? v : u; R[i][j] = sw ? u : v;
            }
        }
        const double piv = rcp64(A[c][c]);                    // getf2 scales the column by the reciprocal of the pivot
#pragma unroll
        for (int i = c + 1; i < 3; ++i) {
            const double l = A[i][c] * piv;
            for (int j = c + 1; j < 3; ++j) A[i][j] = A[i][j] - l * A[c][j];
            for (int j = 0; j < 3; ++j) R[i][j] = R[i][j] - l * R[c][j];
        }
    }
    // back substitution with the reciprocals of the diagonal (an fp64 division is ~40 instructions, and there would be nine)
    const double r0 = rcp64(A[0][0]), r1 = rcp64(A[1][1]), r2 = rcp64(A[2][2]);
    for (int j = 0; j < 3; ++j) {
        X[2][j] = R[2][j] * r2;
        X[1][j] = (R[1][j] - A[1][2] * X[2][j]) * r1;
        X[0][j] = ((R[0][j] - A[0][1] * X[1][j]) - A[0][2] * X[2][j]) * r0;
    }
}
__constant__ double c_pade[4][10] = {{120., 60., 12., 1., 0., 0., 0., 0., 0., 0.},
                                     {30240., 15120., 3360., 420., 30., 1., 0., 0., 0., 0.},
                                     {17297280., 8648640., 1995840., 277200., 25200., 1512., 56., 1., 0., 0.},
                                     {17643225600., 8821612800., 2075673600., 302702400., 30270240., 2162160., 110880., 3960., 90., 1.}};
__device__ __forceinline__ double ect_reference_form(double mu0, double mu1, double l0, double l1, const double pn[3]) {
    const double M[3][3] = {{-2 * mu0 - l0, 0.0, mu1}, {0.0, -2 * mu1 - l1, mu0}, {2 * mu0, 2 * mu1, -mu0 - mu1}};
    // Pade coefficients b_0 .. b_m of orders 3, 5, 7, 9 and the 1-norm up to which each order is used (Al-Mohy & Higham 2009, table 2.3)
    double n1 = 0.0;
    for (int c = 0; c < 3; ++c) n1 = fmax(n1, (fabs(M[0][c]) + fabs(M[1][c])) + fabs(M[2][c]));
    const int m = n1 <= 1.495585217958292e-002 ? 3 : n1 <= 2.539398330063230e-001 ? 5 : n1 <= 9.504178996162932e-001 ? 7 : 9;
    const double* bm = c_pade[(m - 3) >> 1];
    auto coef = [&](int i) { return bm[i]; };
    double A2[3][3], P[3][3], Wm[3][3], V[3][3], T3[3][3];
    mat3_mul(M, M, A2);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { const double id = r == c ? 1.0 : 0.0; P[r][c] = id; Wm[r][c] = coef(1) * id; V[r][c] = coef(0) * id; }
#pragma unroll 1
    for (int j = 1; j <= m / 2; ++j) {
        mat3_mul(P, A2, T3);
        const double bo = coef(2 * j + 1), be = coef(2 * j);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { P[r][c] = T3[r][c]; Wm[r][c] = Wm[r][c] + bo * P[r][c]; V[r][c] = V[r][c] + be * P[r][c]; }
    }
    double U[3][3], Q[3][3], N[3][3], E[3][3], Minv[3][3];
    mat3_mul(M, Wm, U);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Q[r][c] = V[r][c] - U[r][c]; N[r][c] = V[r][c] + U[r][c]; }
    mat3_solve(Q, N, E);
    const double I3[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
    mat3_solve(M, I3, Minv);
    double EmI[3][3], t1[3], t2[3], vec1[3], w[3], vec2[3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) EmI[r][c] = E[r][c] - (r == c ? 1.0 : 0.0);
    mat3_vec(EmI, pn, t1);
    mat3_vec(Minv, t1, t2);
    mat3_vec(Minv, t2, vec1);
    mat3_vec(E, pn, w);
    const double pnc = (w[0] + w[1]) + w[2];
    mat3_vec(Minv, w, vec2);
    return (l0 * (vec2[0] - vec1[0]) + l1 * (vec2[1] - vec1[1])) / (1.0 - pnc);
}

// ------------------------------------------------------- least squares -------
// SciPy's trust-region-reflective solver for the 2x2 systems of the
// lambda-correction, restated so that the ITERATION SEQUENCE matches
// scipy.optimize.least_squares(method='trf', x_scale=1, loss='linear',
// tr_solver='exact', jac='2-point', ftol=1e-8, xtol=gtol=1e-10, max_nfev=100*n)
// as called at CorrectLambda.py:85,260,303,305 (SciPy 1.15.3:
// optimize/_lsq/trf.py trf_no_bounds :401-560 / trf_bounds :205-400,
// common.py solve_lsq_trust_region :57, update_tr_radius :222, check_termination
// :705, CL_scaling_vector :467; _numdiff.py _compute_absolute_step :146).
// The reference's results are defined by where that iteration stops (SURVEY.md
// section 7, hard part 1), not by the exact root.
constexpr double LSQ_EPS = 2.220446049250313e-16;
constexpr double LSQ_FTOL = 1e-8, LSQ_XTOL = 1e-10, LSQ_GTOL = 1e-10;
constexpr double SQRT_EPS = 1.4901161193847656e-08;

// thin SVD of an m x 2 matrix (m = 2 or 4) by one one-sided Jacobi rotation
struct Svd2 {
    double s[2];        // singular values, descending
    double V[2][2];     // V[:, i] = right singular vector i  (V[r][i])
    double uf[2];       // U^T f
};
template <int MROWS>
__device__ __forceinline__ Svd2 svd_mx2(const double A[MROWS][2], const double f[MROWS]) {
    double al = 0, be = 0, ga = 0;
    for (int r = 0; r < MROWS; ++r) { al += A[r][0] * A[r][0]; be += A[r][1] * A[r][1]; ga += A[r][0] * A[r][1]; }
    double c = 1.0, sn = 0.0;
    if (ga != 0.0) {
        double zeta = (be - al) * rcp64(2.0 * ga);
        double t = copysign(1.0, zeta) * rcp64(fabs(zeta) + sqrt64(1.0 + zeta * zeta));
        c = rcp64(sqrt64(1.0 + t * t));
        sn = c * t;
    }
    double n1 = 0, n2 = 0, f1 = 0, f2 = 0;
    for (int r = 0; r < MROWS; ++r) {
        double a1 = c * A[r][0] - sn * A[r][1];
        double a2 = sn * A[r][0] + c * A[r][1];
        n1 += a1 * a1; n2 += a2 * a2; f1 += a1 * f[r]; f2 += a2 * f[r];
    }
    Svd2 o;
    double s1 = sqrt64(n1), s2 = sqrt64(n2);
    double v1[2] = {c, -sn}, v2[2] = {sn, c};
    double u1 = s1 > 0 ? f1 * rcp64(s1) : 0.0, u2 = s2 > 0 ? f2 * rcp64(s2) : 0.0;
    if (s1 >= s2) { o.s[0] = s1; o.s[1] = s2; o.V[0][0] = v1[0]; o.V[1][0] = v1[1]; o.V[0][1] = v2[0]; o.V[1][1] = v2[1]; o.uf[0] = u1; o.uf[1] = u2; }
    else          { o.s[0] = s2; o.s[1] = s1; o.V[0][0] = v2[0]; o.V[1][0] = v2[1]; o.V[0][1] = v1[0]; o.V[1][1] = v1[1]; o.uf[0] = u2; o.uf[1] = u1; }
    return o;
}

// common.py solve_lsq_trust_region :57-166 (n = 2)
__device__ __forceinline__ void solve_tr(const Svd2& d, int m, double Delta, double& alpha, double p[2]) {
    double suf0 = d.s[0] * d.uf[0], suf1 = d.s[1] * d.uf[1];
    bool full_rank = (m >= 2) && (d.s[1] > LSQ_EPS * m * d.s[0]);
    if (full_rank) {
        double a = d.uf[0] * rcp64(d.s[0]), b = d.uf[1] * rcp64(d.s[1]);
        p[0] = -(d.V[0][0] * a + d.V[0][1] * b);
        p[1] = -(d.V[1][0] * a + d.V[1][1] * b);
        if (p[0] * p[0] + p[1] * p[1] <= Delta * Delta) { alpha = 0.0; return; }
    }
    const double rDelta = rcp64(Delta);
    double alpha_upper = sqrt64(suf0 * suf0 + suf1 * suf1) * rDelta;
    double s0s = d.s[0] * d.s[0], s1s = d.s[1] * d.s[1];
    auto phi_fn = [&](double al, double& phi, double& dphi) {
        double r0 = rcp64(s0s + al), r1 = rcp64(s1s + al);
        double a = suf0 * r0, b = suf1 * r1;
        double pn = sqrt64(a * a + b * b);
        phi = pn - Delta;
        dphi = -(a * a * r0 + b * b * r1) * rcp64(pn);
    };
    double alpha_lower = 0.0;
    if (full_rank) { double ph, dp; phi_fn(0.0, ph, dp); alpha_lower = -ph * rcp64(dp); }
    double al = alpha;
    if (!full_rank && al == 0.0) al = fmax(0.001 * alpha_upper, sqrt64(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (al < alpha_lower || al > alpha_upper) al = fmax(0.001 * alpha_upper, sqrt64(alpha_lower * alpha_upper));
        double ph, dp; phi_fn(al, ph, dp);
        if (ph < 0) alpha_upper = al;
        double ratio = ph * rcp64(dp);
        alpha_lower = fmax(alpha_lower, al - ratio);
        al -= (ph + Delta) * ratio * rDelta;
        if (fabs(ph) < 0.01 * Delta) break;
    }
    double a = suf0 * rcp64(s0s + al), b = suf1 * rcp64(s1s + al);
    p[0] = -(d.V[0][0] * a + d.V[0][1] * b);
    p[1] = -(d.V[1][0] * a + d.V[1][1] * b);
    double sc = Delta * rcp64(sqrt64(p[0] * p[0] + p[1] * p[1]));
    p[0] *= sc; p[1] *= sc;
    alpha = al;
}

// common.py update_tr_radius :222-245
__device__ __forceinline__ double update_radius(double Delta, double actual, double predicted, double step_norm, bool bound_hit, double& ratio) {
    if (predicted > 0) ratio = actual / predicted;
    else if (predicted == 0 && actual == 0) ratio = 1.0;
    else ratio = 0.0;
    if (ratio < 0.25) Delta = 0.25 * step_norm;
    else if (ratio > 0.75 && bound_hit) Delta *= 2.0;
    return Delta;
}
// common.py check_termination :705-717 (0 = continue)
__device__ __forceinline__ int check_term(double dF, double F, double dx, double xn, double ratio) {
    bool f_ok = dF < LSQ_FTOL * F && ratio > 0.25;
    bool x_ok = dx < LSQ_XTOL * (LSQ_XTOL + xn);
    return (f_ok && x_ok) ? 4 : f_ok ? 2 : x_ok ? 3 : 0;
}
// _numdiff._compute_absolute_step :173-178 (rel_step=None, '2-point')
__device__ __forceinline__ double fd_step(double x) { return SQRT_EPS * (x >= 0 ? 1.0 : -1.0) * fmax(1.0, fabs(x)); }

// ---- bounded variant (trf_bounds, trf.py:205-400) for N = 1, 2 unknowns, lower bound
// lb on every variable, no upper bound (CorrectLambda.py:82-86, :253-260).  The
// residual functor is evaluated per lane (no cross-lane traffic), so different
// lanes may run different problems (post-split refits: one interval per lane).
template <int N> struct SvdN { double s[N]; double V[N][N]; double uf[N]; };

__device__ __forceinline__ SvdN<1> svd_aug(const double A[2][1], const double f[2]) {
    SvdN<1> o;
    double nn = A[0][0] * A[0][0] + A[1][0] * A[1][0];
    o.s[0] = sqrt(nn);
    o.V[0][0] = 1.0;
    o.uf[0] = o.s[0] > 0 ? (A[0][0] * f[0] + A[1][0] * f[1]) / o.s[0] : 0.0;
    return o;
}
__device__ __forceinline__ SvdN<2> svd_aug(const double A[4][2], const double f[4]) {
    Svd2 t = svd_mx2<4>(A, f);
    SvdN<2> o;
    for (int i = 0; i < 2; ++i) { o.s[i] = t.s[i]; o.uf[i] = t.uf[i]; for (int r = 0; r < 2; ++r) o.V[r][i] = t.V[r][i]; }
    return o;
}

template <int N> __device__ __forceinline__ double vnorm(const double v[N]) {
    double a = 0; for (int i = 0; i < N; ++i) a += v[i] * v[i]; return sqrt(a);
}

// common.py solve_lsq_trust_region :57-166, general n
template <int N> __device__ __forceinline__ void solve_tr_n(const SvdN<N>& d, int m, double Delta, double& alpha, double p[N]) {
    double suf[N];
    for (int i = 0; i < N; ++i) suf[i] = d.s[i] * d.uf[i];
    bool full_rank = (m >= N) && (d.s[N - 1] > LSQ_EPS * m * d.s[0]);
    if (full_rank) {
        for (int r = 0; r < N; ++r) { double a = 0; for (int i = 0; i < N; ++i) a += d.V[r][i] * (d.uf[i] / d.s[i]); p[r] = -a; }
        if (vnorm<N>(p) <= Delta) { alpha = 0.0; return; }
    }
    double alpha_upper = vnorm<N>(suf) / Delta;
    auto phi_fn = [&](double al, double& phi, double& dphi) {
        double q[N], acc = 0;
        for (int i = 0; i < N; ++i) { double e = d.s[i] * d.s[i] + al; q[i] = suf[i] / e; acc += suf[i] * suf[i] / (e * e * e); }
        double pn = vnorm<N>(q);
        phi = pn - Delta; dphi = -acc / pn;
    };
    double alpha_lower = 0.0;
    if (full_rank) { double ph, dp; phi_fn(0.0, ph, dp); alpha_lower = -ph / dp; }
    double al = alpha;
    if (!full_rank && al == 0.0) al = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
    for (int it = 0; it < 10; ++it) {
        if (al < alpha_lower || al > alpha_upper) al = fmax(0.001 * alpha_upper, sqrt(alpha_lower * alpha_upper));
        double ph, dp; phi_fn(al, ph, dp);
        if (ph < 0) alpha_upper = al;
        double ratio = ph / dp;
        alpha_lower = fmax(alpha_lower, al - ratio);
        al -= (ph + Delta) * ratio / Delta;
        if (fabs(ph) < 0.01 * Delta) break;
    }
    for (int r = 0; r < N; ++r) { double a = 0; for (int i = 0; i < N; ++i) a += d.V[r][i] * (suf[i] / (d.s[i] * d.s[i] + al)); p[r] = -a; }
    double sc = Delta / vnorm<N>(p);
    for (int r = 0; r < N; ++r) p[r] *= sc;
    alpha = al;
}

// common.py step_size_to_bound :372-398 with ub = +inf
template <int N> __device__ __forceinline__ double step_to_bound(const double x[N], const double s[N], double lb, int hits[N]) {
    double steps[N], mn = INFINITY;
    for (int i = 0; i < N; ++i) {
        steps[i] = INFINITY;
        if (s[i] != 0) steps[i] = fmax((lb - x[i]) / s[i], s[i] > 0 ? INFINITY : -INFINITY);
        mn = fmin(mn, steps[i]);
    }
    for (int i = 0; i < N; ++i) hits[i] = (steps[i] == mn) ? (s[i] > 0 ? 1 : (s[i] < 0 ? -1 : 0)) : 0;
    return mn;
}
// common.py evaluate_quadratic :325-361 / build_quadratic_1d :248-299 / minimize_quadratic_1d :302-322
template <int N> __device__ __forceinline__ double eval_quad(const double Jh[N][N], const double gh[N], const double s[N], const double diag[N]) {
    double q = 0, l = 0;
    for (int r = 0; r < N; ++r) { double js = 0; for (int c = 0; c < N; ++c) js += Jh[r][c] * s[c]; q += js * js; }
    for (int i = 0; i < N; ++i) { q += s[i] * diag[i] * s[i]; l += s[i] * gh[i]; }
    return 0.5 * q + l;
}
template <int N> __device__ __forceinline__ void build_quad(const double Jh[N][N], const double gh[N], const double s[N], const double diag[N],
                                                            const double* s0, double& a, double& b, double& c) {
    double v[N];
    a = 0; b = 0; c = 0;
    for (int r = 0; r < N; ++r) { v[r] = 0; for (int k = 0; k < N; ++k) v[r] += Jh[r][k] * s[k]; a += v[r] * v[r]; }
    for (int i = 0; i < N; ++i) { a += s[i] * diag[i] * s[i]; b += gh[i] * s[i]; }
    a *= 0.5;
    if (s0) {
        double uu = 0;
        for (int r = 0; r < N; ++r) { double u = 0; for (int k = 0; k < N; ++k) u += Jh[r][k] * s0[k]; b += u * v[r]; uu += u * u; }
        c = 0.5 * uu;
        for (int i = 0; i < N; ++i) { c += gh[i] * s0[i]; b += s0[i] * diag[i] * s[i]; }
        for (int i = 0; i < N; ++i) c += 0.5 * s0[i] * diag[i] * s0[i];
    }
}
__device__ __forceinline__ double min_quad(double a, double b, double lo, double hi, double c, double& y) {
    double t = lo; y = lo * (a * lo + b) + c;
    double yh = hi * (a * hi + b) + c;
    if (yh < y) { y = yh; t = hi; }
    if (a != 0) { double e = -0.5 * b / a; if (lo < e && e < hi) { double ye = e * (a * e + b) + c; if (ye < y) { y = ye; t = e; } } }
    return t;
}
// trf.py select_step :128-203
template <int N> __device__ __forceinline__ double select_step(const double x[N], const double Jh[N][N], const double diag[N], const double gh[N],
                                                              double p[N], double ph[N], const double d[N], double Delta, double lb, double theta,
                                                              double step[N], double steph[N]) {
    bool inb = true;
    for (int i = 0; i < N; ++i) inb = inb && (x[i] + p[i] >= lb);
    if (inb) { for (int i = 0; i < N; ++i) { step[i] = p[i]; steph[i] = ph[i]; } return -eval_quad<N>(Jh, gh, ph, diag); }
    int hits[N];
    double pstride = step_to_bound<N>(x, p, lb, hits);
    double rh[N], r[N], xb[N];
    for (int i = 0; i < N; ++i) { rh[i] = hits[i] != 0 ? -ph[i] : ph[i]; r[i] = d[i] * rh[i]; }
    for (int i = 0; i < N; ++i) { p[i] *= pstride; ph[i] *= pstride; xb[i] = x[i] + p[i]; }
    double to_tr;
    {   // intersect_trust_region(p_h, r_h, Delta), common.py:17-54: positive root
        double a = 0, b = 0, c = -Delta * Delta;
        for (int i = 0; i < N; ++i) { a += rh[i] * rh[i]; b += ph[i] * rh[i]; c += ph[i] * ph[i]; }
        double dd = sqrt(b * b - a * c);
        double q = -(b + copysign(dd, b));
        double t1 = q / a, t2 = c / q;
        to_tr = fmax(t1, t2);
    }
    int h2[N];
    double to_bound = step_to_bound<N>(xb, r, lb, h2);
    double rs = fmin(to_bound, to_tr), rl, ru;
    if (rs > 0) { rl = (1 - theta) * pstride / rs; ru = (rs == to_bound) ? theta * to_bound : to_tr; }
    else { rl = 0; ru = -1; }
    double rval = INFINITY;
    if (rl <= ru) {
        double a, b, c;
        build_quad<N>(Jh, gh, rh, diag, ph, a, b, c);
        double t = min_quad(a, b, rl, ru, c, rval);
        for (int i = 0; i < N; ++i) { rh[i] = rh[i] * t + ph[i]; r[i] = rh[i] * d[i]; }
    }
    for (int i = 0; i < N; ++i) { p[i] *= theta; ph[i] *= theta; }
    double pval = eval_quad<N>(Jh, gh, ph, diag);
    double agh[N], ag[N];
    for (int i = 0; i < N; ++i) { agh[i] = -gh[i]; ag[i] = d[i] * agh[i]; }
    double to_tr2 = Delta / vnorm<N>(agh);
    double to_b2 = step_to_bound<N>(x, ag, lb, h2);
    double ags = to_b2 < to_tr2 ? theta * to_b2 : to_tr2;
    double a, b, c, agval;
    build_quad<N>(Jh, gh, agh, diag, nullptr, a, b, c);
    ags = min_quad(a, b, 0.0, ags, 0.0, agval);
    for (int i = 0; i < N; ++i) { agh[i] *= ags; ag[i] *= ags; }
    if (pval < rval && pval < agval) { for (int i = 0; i < N; ++i) { step[i] = p[i]; steph[i] = ph[i]; } return -pval; }
    if (rval < pval && rval < agval) { for (int i = 0; i < N; ++i) { step[i] = r[i]; steph[i] = rh[i]; } return -rval; }
    for (int i = 0; i < N; ++i) { step[i] = ag[i]; steph[i] = agh[i]; }
    return -agval;
}

// fun(x, f): N residuals of N unknowns.  x is updated in place.
// Returns the solver word (nfev | SciPy status << 16 | kind 2 << 20, see include/misti_hip.h).
template <int N, class Fun> __device__ __forceinline__ int32_t trf_bounded(Fun fun, double x[N], double lb) {
    auto eval = [&](const double xx[N], double f[N], double J[N][N]) {
        fun(xx, f);
        for (int j = 0; j < N; ++j) {
            double h = fd_step(xx[j]);
            if (xx[j] + h < lb) h = -h;                       // _adjust_scheme_to_bounds, 1-sided
            double x1[N], f1[N];
            for (int i = 0; i < N; ++i) x1[i] = xx[i];
            x1[j] = xx[j] + h;
            double dx = x1[j] - xx[j];
            fun(x1, f1);
            for (int r = 0; r < N; ++r) J[r][j] = (f1[r] - f[r]) / dx;
        }
    };
    for (int i = 0; i < N; ++i) {                              // make_strictly_feasible(x0), least_squares.py:828
        double th = 1e-10 * fmax(1.0, fabs(lb));
        if (x[i] - lb <= th) x[i] = lb + th;
    }
    double f[N], J[N][N], g[N];
    eval(x, f, J);
    int nfev = 1;
    const int max_nfev = 100 * N;
    double cost = 0;
    for (int i = 0; i < N; ++i) cost += f[i] * f[i];
    cost *= 0.5;
    auto grad = [&]() { for (int c = 0; c < N; ++c) { g[c] = 0; for (int r = 0; r < N; ++r) g[c] += J[r][c] * f[r]; } };
    grad();
    double v[N], dv[N];
    auto CL = [&]() { for (int i = 0; i < N; ++i) { if (g[i] > 0) { v[i] = x[i] - lb; dv[i] = 1.0; } else { v[i] = 1.0; dv[i] = 0.0; } } };
    CL();
    double Delta;
    { double t[N]; for (int i = 0; i < N; ++i) t[i] = x[i] / sqrt(v[i]); Delta = vnorm<N>(t); if (Delta == 0) Delta = 1.0; }
    double alpha = 0.0;
    int term = 0;
    for (;;) {
        CL();
        double g_norm = 0;
        for (int i = 0; i < N; ++i) g_norm = fmax(g_norm, fabs(g[i] * v[i]));
        if (g_norm < LSQ_GTOL) term = 1;
        if (term != 0 || nfev >= max_nfev) break;
        if (!(g_norm < INFINITY)) break;
        double d[N], diag[N], gh[N], Jh[N][N], A[2 * N][N], fa[2 * N];
        for (int i = 0; i < N; ++i) { d[i] = sqrt(v[i]); diag[i] = g[i] * dv[i]; gh[i] = d[i] * g[i]; }
        for (int r = 0; r < N; ++r) { fa[r] = f[r]; fa[N + r] = 0.0; for (int c = 0; c < N; ++c) { Jh[r][c] = J[r][c] * d[c]; A[r][c] = Jh[r][c]; A[N + r][c] = (r == c) ? sqrt(diag[r]) : 0.0; } }
        SvdN<N> sv = svd_aug(A, fa);
        double theta = fmax(0.995, 1.0 - g_norm);
        double actual = -1.0;
        double xn[N], fn[N], Jn[N][N], cost_new = cost;
        while (actual <= 0 && nfev < max_nfev) {
            double ph[N], p[N], step[N], steph[N];
            solve_tr_n<N>(sv, N, Delta, alpha, ph);
            for (int i = 0; i < N; ++i) p[i] = d[i] * ph[i];
            double predicted = select_step<N>(x, Jh, diag, gh, p, ph, d, Delta, lb, theta, step, steph);
            for (int i = 0; i < N; ++i) { xn[i] = x[i] + step[i]; if (xn[i] <= lb) xn[i] = nextafter(lb, INFINITY); }   // make_strictly_feasible(rstep=0)
            eval(xn, fn, Jn);
            ++nfev;
            double shn = vnorm<N>(steph);
            bool fin = true;
            for (int i = 0; i < N; ++i) fin = fin && isfinite(fn[i]);
            if (!fin) { Delta = 0.25 * shn; continue; }
            cost_new = 0;
            for (int i = 0; i < N; ++i) cost_new += fn[i] * fn[i];
            cost_new *= 0.5;
            actual = cost - cost_new;
            double ratio;
            double Delta_new = update_radius(Delta, actual, predicted, shn, shn > 0.95 * Delta, ratio);
            term = check_term(actual, cost, vnorm<N>(step), vnorm<N>(x), ratio);
            if (term != 0) break;
            alpha *= Delta / Delta_new;
            Delta = Delta_new;
        }
        if (actual > 0) {
            for (int i = 0; i < N; ++i) { x[i] = xn[i]; f[i] = fn[i]; for (int c = 0; c < N; ++c) J[i][c] = Jn[i][c]; }
            cost = cost_new;
            grad();
        }
    }
    return solver_word(nfev, term, 2);
}

// ExpectedCoalTimeOnePop, CorrectLambda.py:67-72 (note the lam > 100 clamp)
__device__ __forceinline__ double ect_one_pop(double lam, double T) {
    double r = lam > 100.0 ? 0.0 : T / (exp(lam * T) - 1.0);
    return 1.0 / lam - r;
}
// ExpectedCoalTimeOnePopNonConditional, CorrectLambda.py:79-80
__device__ __forceinline__ double ect_noncond(double lam, double T) { return (1.0 - exp(-lam * T) * (1.0 + lam * T)) / lam; }

// Residuals of the migrating two-population interval, evaluated for the base
// point and both forward-difference points in one pass of the candidate's lane group.
// lane (within the group) = 2*e + k: e = 0 base, 1 = x + h0 e0, 2 = x + h1 e1; k = genome.
struct PairProblem {
    double mu0, mu1;       // stretched to unit interval (CorrectLambda.py:293-298)
    // of THIS lane's genome k = role & 1 (a lane evaluates one genome for as long as it lives: keeping only its own row
    // halves what the solver loop carries)
    double Pk[3];          // pair-state vector at the start of the interval
    double sk;             // its sum
    double tgtk;           // cpfit: exp(-lh_k) * s_k (:141); default fit: one-population expected coalescence time (:74-77)
    int red;               // structural reduction of the pair generator for this interval (pair_reduced): 0 none, 1 / 2 state 0 / 1 empty and unfed
};

// One residual evaluation per lane: this lane's role r (= 2 e + k, see PairProblem) at ITS point (x0, x1).  The six
// lanes of a slot share the point; different slots of a wave may hold different points (the needed one and guesses of
// its successors, see correct_body).  A pure function of (pb, point, role): what else sits in the wave changes no bit.
// PROBE (default fit only): *probe receives, for this lane's evaluation, the difference between the REFERENCE's form of the
// expected coalescence time (ect_reference_form) and the integral series used for the value: one draw of the rounding noise the
// reference's residual carries at this point (the series is accurate to ~4e-15; the formula loses ~1/|M|^2 digits).  See
// ect_noise_continues.
template <bool CPFIT, bool PROBE = false>
__device__ __forceinline__ void pair_eval(const PairProblem& pb, double x0, double x1, int role, Diag& dg, double& res, double w[3], bool& guard,
                                          double* probe = nullptr) {
    const double h0 = fd_step(x0), h1 = fd_step(x1);
    const double xa = x0 + h0, xb = x1 + h1;
    const int e = role >> 1;
    const double l0 = (e == 1) ? xa : x0;
    const double l1 = (e == 2) ? xb : x1;
    const bool ok = isfinite(x0) && isfinite(x1);
    // slot-uniform uniformisation rate: the largest over the three evaluation points
    const double m0 = fmax(x0, xa), m1 = fmax(x1, xb);
    const double q = fmax(fmax(2.0 * pb.mu0 + m0, 2.0 * pb.mu1 + m1), fmax(pb.mu0 + pb.mu1, 0.0));
    const double neg = fmax(0.0, fmax(-fmin(x0, xa), -fmin(x1, xb)));
    const double sk = pb.sk;
    if (CPFIT) {
        // LambdaSystem1 / LambdaEquation, CorrectLambda.py:135-144,169-173
        for (int i = 0; i < 3; ++i) w[i] = pb.Pk[i];
        if (pb.red != 0 && q + neg < 1e300) pair_reduced(pb.red, l0, l1, pb.mu0, pb.mu1, w, ok);
        else pair_expv(l0, l1, pb.mu0, pb.mu1, w, q, neg, ok, dg, guard);
        // summed in the reference's order (:141-144).  (Tried: the pieces that do not depend on a runaway rate first, so that its
        // forward difference keeps more digits than the reference's own - campaign seeds 1 and 2 then lose 2 statuses and 10
        // candidates, config 5 sixteen: the reference's decisions carry the rounding of ITS sum.)
        res = ((w[0] + w[1]) + w[2]) - pb.tgtk;
    } else {
        // LambdaSystem / ExpectedCoalTimeTwoPop, CorrectLambda.py:94-110,151-157
        double pn[3], vint[3] = {0.0, 0.0, 0.0};
        bool have_int = false;
        for (int i = 0; i < 3; ++i) { pn[i] = pb.Pk[i] / sk; w[i] = pn[i]; }
        pair_expv<true>(l0, l1, pb.mu0, pb.mu1, w, q, neg, ok, dg, guard, vint, &have_int);
        double pnc = (w[0] + w[1]) + w[2];
        double v0 = vint[0], v1 = vint[1];          // (vec2 - vec1) of the reference, T = 1 after the stretch
#ifdef MISTI_ECT_FORMULA
        have_int = false;
#endif
        if (!have_int) {
            // stiff iterate (|M| > 1): the reference's own formula, M^-1 exp(M) pn - M^-2 (exp(M) - I) pn
            double M[3][3] = {{-2 * pb.mu0 - l0, 0.0, pb.mu1}, {0.0, -2 * pb.mu1 - l1, pb.mu0}, {2 * pb.mu0, 2 * pb.mu1, -pb.mu0 - pb.mu1}};
            double dd[3] = {w[0] - pn[0], w[1] - pn[1], w[2] - pn[2]};
            double y[3], vec1[3], vec2[3];
#ifdef MISTI_ECT_FORMULA
            solve3_ge(M, dd, y);
            solve3_ge(M, y, vec1);
            solve3_ge(M, w, vec2);
#else
            solve3(M, dd, y);
            solve3(M, y, vec1);
            solve3(M, w, vec2);
#endif
            v0 = vec2[0] - vec1[0]; v1 = vec2[1] - vec1[1];
        }
        double ect = (l0 * v0 + l1 * v1) / (1.0 - pnc);
        res = ect - pb.tgtk;
        if (PROBE) {
            double d = 0.0;
            if (have_int) d = ect_reference_form(pb.mu0, pb.mu1, l0, l1, pn) - ect;
            *probe = (d == d && fabs(d) < 1e300) ? d : 0.0;
        }
        // the state vector handed on is exp(M) applied to the unnormalised vector
        w[0] *= sk; w[1] *= sk; w[2] *= sk;
    }
}

// Residuals and forward-difference Jacobian at a point from the six residuals evaluated there (base point, first rate
// stepped, second rate stepped; both genomes).
__device__ __forceinline__ void collect_core(double x0, double x1, double fb0, double fb1, double fa0, double fa1, double fc0, double fc1,
                                             double f[2], double J[2][2], bool& finite) {
    const double h0 = fd_step(x0), h1 = fd_step(x1);
    const double xa = x0 + h0, xb = x1 + h1;
    const double dx0 = xa - x0, dx1 = xb - x1;      // recomputed as exactly representable (_numdiff.py)
    const double r0 = rcp64(dx0), r1 = rcp64(dx1);
    f[0] = fb0; f[1] = fb1;
    J[0][0] = (fa0 - fb0) * r0; J[1][0] = (fa1 - fb1) * r0;
    J[0][1] = (fc0 - fb0) * r1; J[1][1] = (fc1 - fb1) * r1;
    finite = isfinite(fb0) && isfinite(fb1);
}
// ... at the point xe from the six lanes base .. base + 5 that evaluated it (wave- or group-uniform result)
template <int GROUP>
__device__ __forceinline__ void pair_collect(double res, const double xe[2], int base, double f[2], double J[2][2], bool& finite) {
    const double fb0 = gbcast<GROUP>(res, base + 0), fb1 = gbcast<GROUP>(res, base + 1);
    const double fa0 = gbcast<GROUP>(res, base + 2), fa1 = gbcast<GROUP>(res, base + 3);
    const double fc0 = gbcast<GROUP>(res, base + 4), fc1 = gbcast<GROUP>(res, base + 5);
    collect_core(xe[0], xe[1], fb0, fb1, fa0, fa1, fc0, fc1, f, J, finite);
}
// ... in every slot at once: each lane gets the result of ITS slot (lanes sbase .. sbase + 5) at its slot's point
__device__ __forceinline__ void slot_collect(double res, double x0, double x1, int sbase, double f[2], double J[2][2], bool& finite) {
    const double fb0 = __shfl(res, sbase + 0, 64), fb1 = __shfl(res, sbase + 1, 64);
    const double fa0 = __shfl(res, sbase + 2, 64), fa1 = __shfl(res, sbase + 3, 64);
    const double fc0 = __shfl(res, sbase + 4, 64), fc1 = __shfl(res, sbase + 5, 64);
    collect_core(x0, x1, fb0, fb1, fa0, fa1, fc0, fc1, f, J, finite);
}

// Trust-region bookkeeping of one evaluated trial of trf_no_bounds (trf.py:488-521; update_tr_radius,
// common.py:222-245; check_termination, common.py:705-717).  Straight-line (selects, no branches): run with
// wave-uniform operands by the serial consume loop and with one hypothesis per slot by the tree consume of
// correct_body - the same expressions, so the same bits.
__device__ __forceinline__ void tr_update(bool finite, const double p[2], const double x[2], double cost, double cost_new, double predicted,
                                          double& Delta, double& dq, int& term, bool& accept) {
    const double sn2 = p[0] * p[0] + p[1] * p[1];
    dq = 0.25 * sqrt64(sn2);                                         // radius after a poor or non-finite step
    const double actual = cost - cost_new;
    const double rq = actual * rcp64(predicted);
    const double ratio = predicted > 0 ? rq : ((predicted == 0 && actual == 0) ? 1.0 : 0.0);
    const double Delta_new = ratio < 0.25 ? dq                       // update_tr_radius
                             : ((ratio > 0.75 && sn2 > 0.9025 * Delta * Delta) ? 2.0 * Delta : Delta);
    const bool f_ok = actual < LSQ_FTOL * cost && ratio > 0.25;      // check_termination
    const double lim = LSQ_XTOL * (LSQ_XTOL + sqrt64(x[0] * x[0] + x[1] * x[1]));
    const bool x_ok = sn2 < lim * lim;
    term = !finite ? 0 : ((f_ok && x_ok) ? 4 : f_ok ? 2 : x_ok ? 3 : 0);
    Delta = !finite ? dq : (term == 0 ? Delta_new : Delta);
    accept = finite && actual > 0;
}
// predicted reduction -(0.5 |J p|^2 + p . g) of a step (evaluate_quadratic, common.py:276)
__device__ __forceinline__ double predicted_of(const double J[2][2], const double p[2], const double g[2]) {
    const double Js0 = J[0][0] * p[0] + J[0][1] * p[1], Js1 = J[1][0] * p[0] + J[1][1] * p[1];
    return -(0.5 * (Js0 * Js0 + Js1 * Js1) + (p[0] * g[0] + p[1] * g[1]));
}

// Next trial step of trf_no_bounds (trf.py:469-486): Gauss-Newton step when the Jacobian has
// full rank and the step lies in the trust region (solve_lsq_trust_region, common.py:116-125),
// otherwise the regularised step from the SVD.  Returns the predicted reduction.
template <int GROUP>
__device__ __forceinline__ double next_step(const double J[2][2], const double f[2], const double g[2], double Delta, double& alpha,
                                            double p[2], int& lm, int& axis) {
    const double det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
    const double F = (J[0][0] * J[0][0] + J[0][1] * J[0][1]) + (J[1][0] * J[1][0] + J[1][1] * J[1][1]);
    // A structurally decoupled rate (Jacobian column exactly zero, see pair_expv): J has rank one,
    // solve_lsq_trust_region takes its regularised branch and ALWAYS rescales the step to the
    // trust radius (common.py:160-164), so the step is (0, -sign(g1) Delta) whatever alpha the
    // secular iteration ends with - and alpha is only a starting guess for the next call.
    const bool z0 = J[0][0] == 0.0 && J[1][0] == 0.0, z1 = J[0][1] == 0.0 && J[1][1] == 0.0;
    const bool rank1 = z0 != z1;
    const int k = z0 ? 1 : 0;
    axis = rank1 ? k : -1;
    const double gk = z0 ? g[1] : g[0];
    const double pk = gk > 0 ? -Delta : Delta;
    // Gauss-Newton step (computed unconditionally - straight-line code - and used when J certainly has full rank and the
    // step lies inside the trust region): s_max^2 <= F <= 2 s_max^2 and s_min = |det| / s_max, so full rank
    // (s_min > 2 eps s_max) is certain if |det| > 2 eps F
    const double rd = rcp64(det);
    const double p0 = -(J[1][1] * f[0] - J[0][1] * f[1]) * rd;
    const double p1 = -(J[0][0] * f[1] - J[1][0] * f[0]) * rd;
    const bool gn = !rank1 && fabs(det) > 2.0 * LSQ_EPS * 2.0 * F && p0 * p0 + p1 * p1 <= Delta * Delta;
    if (uni<GROUP>(rank1 || gn)) {
        p[0] = rank1 ? (z0 ? 0.0 : pk) : p0;
        p[1] = rank1 ? (z0 ? pk : 0.0) : p1;
        alpha = 0.0;
    } else {
        // the decomposition is recomputed on every regularised step (rejected steps repeat it for the same J, f: ~50 such
        // steps per chain) rather than carried through the solver loop: 16 registers of loop-carried state less
        const Svd2 d = svd_mx2<2>(J, f);
        solve_tr(d, 2, Delta, alpha, p);
        lm += 1;
    }
    return predicted_of(J, p, g);
}

// ---- default fit: would the REFERENCE stop here? ----------------------------------------------------------------------------
// SciPy stops a solve when |J^T f|_inf < gtol = 1e-10 (trf.py:452).  The reference's residual of the default fit is
// M^-1 e^M p - M^-2 (e^M - I) p (CorrectLambda.py:94-110), which carries ~eps/|M|^2 = 1e-14 ... 1e-9 of rounding noise; its
// forward-difference Jacobian (h = 1.5e-8) therefore ~1e-6 ... 1e-1, so its Gauss-Newton step from x to x + p lands
// -J^-1 dJ p off the step an exact Jacobian gives, where the residual is still f - dJ p: the reference's gradient test sees
// J^T (f - dJ p) and, wherever |J^T dJ p| >~ gtol, goes on for one more evaluation - which converges to the root - where the
// noise-free integral series used here already satisfies gtol and would stop 1e-8 ... 1e-7 short in the rate (round 2's open
// parity defect: campaign seed 2 model 35, 1.2e-8 ... 1.7e-8 in the likelihood where the reference's own spread is 6e-12).
// The reference's noise cannot be reproduced bit for bit (its expm is scipy's Pade approximant through BLAS, its inverse
// LAPACK's getri), but its SIZE can be measured: six draws per genome of (reference formula - series) at points within 3 h of
// the step's base point give the width W of the error distribution; the median |dJ p| of a forward difference of two such
// errors is 0.29 W |p| / h.  The solve goes on iff  sqrt(g_i^2 + sum_k (J_ki 0.29 W_k max_j |p_j / h_j|)^2) >= gtol  for some i
// (W: the measured width scaled by the calibrated ratio between the imitation's noise and the reference's own).
// Where that flips within the noise the reference itself flips under perturbation (its measured spread covers either
// decision: e.g. model 35's other chain, 1.9e-8); where the noise term dominates (model 35) or vanishes (a converged step:
// |p| ~ 1e-8) the decision is the reference's.  Steps and values always come from the series: a wrong guess moves the rate by
// at most what gtol leaves open anyway.
template <int GROUP>
__device__ __forceinline__ bool ect_noise_continues(const PairProblem& pb, double xo0, double xo1, const double p[2], const double J[2][2],
                                                    const double g[2], int role) {
    const double h0 = fd_step(xo0), h1 = fd_step(xo1);
    double lo0 = INFINITY, hi0 = -INFINITY, lo1 = INFINITY, hi1 = -INFINITY;
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
        const double bx0 = xo0 + (2.0 * s) * h0, bx1 = xo1 + (2.0 * s) * h1;
        Diag dgp;
        double resp, wp[3], d = 0.0;
        bool guardp = false;
        pair_eval<false, true>(pb, bx0, bx1, role, dgp, resp, wp, guardp, &d);
        for (int e = 0; e < 3; ++e) {
            const double d0 = gbcast<GROUP>(d, 2 * e), d1 = gbcast<GROUP>(d, 2 * e + 1);
            lo0 = fmin(lo0, d0); hi0 = fmax(hi0, d0); lo1 = fmin(lo1, d1); hi1 = fmax(hi1, d1);
        }
    }
    // range of N = 6 draws -> width of the distribution: x (N + 1) / (N - 1); the imitation's width -> the reference's: / 2.2 (median
    // ratio over 481 sampled solves, see ect_reference_form)
    const double W0 = (hi0 - lo0) * (1.4 / 2.2), W1 = (hi1 - lo1) * (1.4 / 2.2);
    const double ph = fmax(fabs(p[0] / h0), fabs(p[1] / h1));
    const double A0 = 0.29 * W0 * ph, A1 = 0.29 * W1 * ph;
    const double n0 = (J[0][0] * A0) * (J[0][0] * A0) + (J[1][0] * A1) * (J[1][0] * A1);
    const double n1 = (J[0][1] * A0) * (J[0][1] * A0) + (J[1][1] * A1) * (J[1][1] * A1);
    const double e0 = g[0] * g[0] + n0, e1 = g[1] * g[1] + n1;
    return fmax(e0, e1) >= LSQ_GTOL * LSQ_GTOL;
}

// ------------------------------------------------ two-population correction --
// MigrationInference.CorrectLambdas, loop t < splitT (:311-354).  All values are
// wave-uniform except inside pair_eval.  Returns false on "correction failed".
struct PairState { double p[2][3]; };
__device__ __forceinline__ bool wrong_phase(const ChainBufs& cb, int64_t ch, int phase);

template <int GROUP = 1>
__device__ __forceinline__ void pulse_pairs(PairState& ps, double pu0, double pu1) {
    double r = pu0 + pu1;                                        // :315-323
    if (uni<GROUP>(!(r > 0))) return;
    int a = pu0 > 0 ? 0 : 1, b = 1 - a;
    for (int k = 0; k < 2; ++k) {
        double pa = ps.p[k][a], pb = ps.p[k][b], pc = ps.p[k][2];
        double omr = 1.0 - r;
        ps.p[k][a] = pa * (omr * omr);
        ps.p[k][b] = pa * (r * r) + pb + pc * r;
        ps.p[k][2] = pa * 2 * omr * r + pc * omr;
    }
}

// ----------------------------------------------------------- the kernels ----
// Candidate structure shared by both kernels: fractional split (:89-99), negative
// parameter guard (:569-572), split beyond the grid.
// SetModel's checks on the bands as this candidate sees them (MigrationInference.py:237-255): start >= sample
// date, start < end (end == -1: the candidate's split index), no overlap within a population.
__device__ __forceinline__ bool bands_valid(const DevModel& m, const int32_t* bb, int split) {
    bool ok = true;
    for (int b = 0; b < m.n_band; ++b) {
        const int sb = bb ? bb[2 * b] : m.bands[b].start;
        int eb = bb ? bb[2 * b + 1] : m.bands[b].end;
        if (eb < -1 || eb > m.numT + 1) ok = false;
        if (eb < 0) eb = split;
        if (sb < m.sample_date || eb <= sb) ok = false;
        for (int c = 0; c < b; ++c) {
            if (m.bands[c].pop != m.bands[b].pop) continue;
            const int sc = bb ? bb[2 * c] : m.bands[c].start;
            int ec = bb ? bb[2 * c + 1] : m.bands[c].end;
            if (ec < 0) ec = split;
            if (sb < ec && sc < eb) ok = false;
        }
    }
    return ok;
}

__device__ __forceinline__ int setup_candidate(const DevModel& m, double st, const double* par, Grid& G, const int32_t* bb = nullptr) {
    int status = MISTI_OK;
    G.times = m.times; G.lh = m.lh; G.numT0 = m.numT;
    double fl = floor(st);
    G.frac = st - fl;
    int s = (int)fl;
    G.ins = -1; G.split = s; G.numT = m.numT;
    if (!(st >= 0) || !(st >= (double)m.sample_date) || s > m.numT) status = MISTI_BAD_STRUCTURE;
    else if (G.frac != 0.0) {
        if (s > m.numT - 2) status = MISTI_BAD_STRUCTURE;
        else { G.ins = s; G.split = s + 1; G.numT = m.numT + 1; }
    }
    if (status == MISTI_OK && m.n_band > 0 && !bands_valid(m, bb, G.split)) status = MISTI_BAD_STRUCTURE;
    // construction errors (PrintError + exit in the reference's __init__/SetModel) come before the guard of JAFSLikelihood (:569-572)
    if (status == MISTI_OK) for (int i = 0; i < m.n_param; ++i) if (par[i] < 0) status = MISTI_NEG_PARAM;
    if (status == MISTI_OK && G.split >= G.numT) status = MISTI_INF_COAL;
    return status;
}

// First interval whose rates depend on THIS candidate's split - where it leaves the trunk of its chain: the start of the
// smoothing run (of either genome) that the split cuts, or the shortened interval of a fractional split.  Used by the candidate
// kernel (where to start from) and by setup_kernel (from which interval on the chain's trunk has to store records).
__device__ __forceinline__ int trunk_leave(const DevModel& m, const Grid& G) {
    int t_own = (G.ins >= 0) ? G.ins : G.split;
    if (m.flags & MISTI_SMOOTH) {
        if (G.ins >= 0) {
            t_own = min(m.run_start[G.ins], m.run_start[m.numT + G.ins]);
        } else if (G.split > 0) {
            for (int k = 0; k < 2; ++k)
                if (m.run_end[k * m.numT + G.split - 1] > G.split) t_own = min(t_own, m.run_start[k * m.numT + G.split - 1]);
        }
    }
    return t_own;
}

// ------------------------------------------------------------ forward map ----
// MigrationInference.CoalescentRates (MigrationInference.py:542-564) with CorrectLambda.CoalRates
// (CorrectLambda.py:112-122): the model's own rates lh are taken as the TRUE per-population rates;
// for every two-population interval the rate a single-genome (PSMC) analysis would see is
// -log(P[no coalescence in the interval]) / T under the pair chain started from that genome's
// pair-state distribution.  No solver: one 3-state exponential action per genome and interval, so
// one thread per candidate.  Rates after the split are returned unchanged (the loop at :563 is empty).
// hold_mu: the reference's CoalescentRates never sets the migration rates of its CorrectLambda
// object, so every interval is evaluated with what the preceding CorrectLambdas loop left there
// (:324) - the rates of the LAST two-population interval.  hold_mu = 1 reproduces that; 0 applies
// each interval's own rates (the model as specified; what data generation wants).
__global__ __launch_bounds__(64)
void forward_kernel(DevModel m, int64_t n_cand, const double* __restrict__ split_time, const double* __restrict__ params, int hold_mu,
                    double* __restrict__ lh_out, double* __restrict__ pr_out, int32_t* __restrict__ status_out) {
    const int64_t cand = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (cand >= n_cand) return;
    Grid G;
    const double* par = params ? params + cand * m.n_param : nullptr;
    const int status = setup_candidate(m, split_time[cand], par, G);
    const int rows = m.numT + 1;
    double* lh_o = lh_out + cand * (int64_t)rows * 2;
    double* pr_o = pr_out ? pr_out + cand * (int64_t)(m.numT + 2) * 6 : nullptr;
    if (status_out) status_out[cand] = status;
    if (pr_o) for (int i = 0; i < (m.numT + 2) * 6; ++i) pr_o[i] = 0.0;
    if (status != MISTI_OK && status != MISTI_INF_COAL) {
        for (int i = 0; i < 2 * rows; ++i) lh_o[i] = NAN;
        return;
    }
    Model M; M.m = &m; M.par = par; M.split = G.split; M.cache();
    PairState ps = {{{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}}};
    Diag dg;
    double held0 = 0.0, held1 = 0.0;
    if (hold_mu && G.split >= 1) M.mig((G.split < G.numT ? G.split : G.numT - 1) - 1, held0, held1);
    for (int t = 0; t < rows; ++t) {
        double l0 = 0.0, l1 = 0.0;
        if (t < G.numT) { l0 = G.lhk(t, 0); l1 = G.lhk(t, 1); }
        if (t < G.split && t < G.numT - 1) {
            double pu0, pu1, mu0, mu1;
            M.pulse(t, pu0, pu1);
            pulse_pairs(ps, pu0, pu1);
            if (t == 0 && pr_o)
                for (int j = 0; j < 6; ++j) pr_o[j] = ps.p[j & 1][j >> 1];
            M.mig(t, mu0, mu1);
            if (hold_mu) { mu0 = held0; mu1 = held1; }
            const double T = G.T(t);
            const double a0 = l0 * T, a1 = l1 * T, b0 = mu0 * T, b1 = mu1 * T;
            const double q = fmax(fmax(2.0 * b0 + a0, 2.0 * b1 + a1), b0 + b1);
            for (int k = 0; k < 2; ++k) {
                double v[3] = {ps.p[k][0], ps.p[k][1], ps.p[k][2]};
                const double before = (v[0] + v[1]) + v[2];
                bool guard_unused = false;
                pair_expv(a0, a1, b0, b1, v, q, 0.0, true, dg, guard_unused);
                const double nc = (v[0] + v[1]) + v[2];
                const double seen = -log(nc / before) / T;
                if (k == 0) l0 = seen; else l1 = seen;
                ps.p[k][0] = v[0]; ps.p[k][1] = v[1]; ps.p[k][2] = v[2];
            }
            if (pr_o)
                for (int j = 0; j < 6; ++j) pr_o[6 * (t + 1) + j] = ps.p[j & 1][j >> 1];
        }
        lh_o[2 * t] = l0; lh_o[2 * t + 1] = l1;
    }
}

// Kernel 1: lambda correction of the two-population intervals (CorrectLambdas loop t < splitT,
// MigrationInference.py:307-354; SolveLambdaSystem, CorrectLambda.py:266-317).
//
// The recursion over intervals does not depend on the split time except for where it stops:
// candidates with the same parameter vector share one CHAIN (discover_kernel), computed
// once up to the largest split index of its members - in a split x rate sweep 64 chains instead of
// 4 096.  A fractional split shortens the candidate's last two-population interval; that one
// interval is a TAIL, run per candidate from the chain's state (TAIL = true, in post_kernel).
//
// GROUP lanes per work item (64, 32, 16, 8 or 6; six carry the residual evaluations), no LDS beyond the staged grid.
// Each item runs its own resumable state machine: one pass of the loop = one residual batch
// (trial point + its forward-difference points) + the trust-region bookkeeping of
// trf_no_bounds (trf.py:401-560), so the items of a wavefront never wait for each other
// interval by interval.
template <bool CPFIT, int GROUP, bool TAIL, bool PRE = false>
__device__ __forceinline__
void correct_body(const DevModel& m, int64_t n_items, const ChainBufs& cb, const double* __restrict__ split_time, const double* __restrict__ params,
                  int64_t block, double* lds, double* lc_sh = nullptr, volatile int* flags = nullptr, double* pre = nullptr,
                  const int32_t* chain_map = nullptr, int yield_nfev = 0, int t_start = 0, int phase = 0) {
    const int lane = lane_id();
    const int sub = lane % GROUP;
    const int64_t n_live = TAIL ? n_items : (int64_t)cb.n_chains[0];
    // GROUP = 6 packs ten items into a wave; its last four lanes belong to no item
    const int64_t pos = (lane / GROUP < 64 / GROUP) ? (block * (64 / GROUP)) + (lane / GROUP) : n_live;
    if (TAIL) {
        // a wave none of whose items has a fractional split has nothing to do: leave before staging
        bool mine = false;
        if (pos < n_live) { const double st = split_time[pos]; mine = st != floor(st); }
        if (!__any(mine)) {
            if (pos < n_live && sub == 0) cb.tail_status[pos] = MISTI_OK;
            return;
        }
    } else if (block * (64 / GROUP) >= n_live) return;
    // chains of a packed launch (chain_map): position -> chain by descending length, so that the longest start first and the
    // chains sharing a wave are of similar length; a chain's bits do not depend on where it runs
    const int64_t slot = (!TAIL && chain_map && pos < n_live) ? (int64_t)chain_map[pos] : pos;
    // the shared grid (interval lengths, PSMC rates) staged in LDS: every pass of the state
    // machine of some item reads it, and an L2 round trip per read dominated the kernel
    {
        const int nt = m.numT - 1, nl = 2 * m.numT;
        for (int i = lane; i < nt; i += 64) lds[i] = m.times[i];
        for (int i = lane; i < nl; i += 64) lds[nt + i] = m.lh[i];
    }
    lds_fence();                                   // staged and read by this wavefront only
    if (pos >= n_live) return;
    const int64_t cand = TAIL ? slot : (int64_t)cb.rep[slot];      // whose parameters (chains: slot IS the chain id, see correct_kernel)
    const double* par = params ? params + cand * m.n_param : nullptr;
    Grid G;
    int status;
    int t = 0;
    double* lc_w;     // lc_w[2 t + k]        corrected rates of interval t
    double* tr_w;     // tr_w[6 (t + 1) + j]  pair state after interval t
    PairState ps;
    const int32_t* bb = cb.bounds ? cb.bounds + cand * 2 * m.n_band : nullptr;
    int32_t* sv_w = nullptr;   // sv_w[t]  solver word of interval t (trace on)
    double* it_w = nullptr;    // it_w[(t * MISTI_TRACE_MAX_ITER + i) * 2 + k]  trial points of the unbounded solve of interval t
    if (!TAIL) {
        status = MISTI_OK;
        for (int i = 0; i < m.n_param; ++i) if (par[i] < 0) status = MISTI_NEG_PARAM;
        G.numT0 = m.numT; G.numT = m.numT; G.split = chain_len(cb, slot); G.ins = -1; G.frac = 0.0;
        if (cb.solver) sv_w = cb.solver + slot * (int64_t)m.numT;
        if (cb.iters && slot < cb.iter_cap) it_w = cb.iters + slot * (int64_t)m.numT * MISTI_TRACE_MAX_ITER * 2;
        lc_w = cb.lc + slot * (int64_t)m.numT * 2;
        tr_w = cb.trace + slot * (int64_t)(m.numT + 1) * 6;
        ps.p[0][0] = 1; ps.p[0][1] = 0; ps.p[0][2] = 0;
        ps.p[1][0] = 0; ps.p[1][1] = 1; ps.p[1][2] = 0;
        if (t_start > 0) {
            // a chain that YIELDED in the packed launch (see below) resumes at the interval it was in: the pair state before
            // interval t_start is in its trace, and a solve restarted from the same state is the same solve
            t = t_start;
            const double* r = tr_w + 6 * t_start;
            ps.p[0][0] = r[0]; ps.p[1][0] = r[1]; ps.p[0][1] = r[2]; ps.p[1][1] = r[3]; ps.p[0][2] = r[4]; ps.p[1][2] = r[5];
        } else if (sub == 0) { tr_w[0] = 1; tr_w[1] = 0; tr_w[2] = 0; tr_w[3] = 1; tr_w[4] = 0; tr_w[5] = 0; }
    } else {
        status = setup_candidate(m, split_time[cand], par, G, bb);
        if (sub == 0) cb.tail_status[cand] = MISTI_OK;
        if (status != MISTI_OK || G.ins < 0) return;               // nothing to do: no fractional split
        if (cb.tail_solver) sv_w = cb.tail_solver + cand - G.ins;
        const int64_t ch = chain_of(cb, cand);
        if (wrong_phase(cb, ch, phase)) return;                    // this chain's members belong to the other phase's launch
        if (cb.fail_t[ch] < G.ins) return;                         // the chain failed before this candidate's tail
        t = G.ins;
        const double* r = cb.trace + (ch * (int64_t)(m.numT + 1) + t) * 6;
        ps.p[0][0] = r[0]; ps.p[1][0] = r[1]; ps.p[0][1] = r[2]; ps.p[1][1] = r[3]; ps.p[0][2] = r[4]; ps.p[1][2] = r[5];
        lc_w = cb.tail_lc + 2 * cand - 2 * (int64_t)t;
        tr_w = cb.tail_state + 6 * cand - 6 * (int64_t)(t + 1);
    }
    G.times = lds; G.lh = lds + (m.numT - 1);
    Model mod{&m, par, G.split, {0, 0, 0, 0}};
    mod.bb = bb;
    mod.cache();
    const bool correct = !(m.flags & MISTI_TRUE_EPS);
    const int max_nfev = 200;                    // 100 * n (least_squares.py)
    Diag dg;
#ifdef MISTI_TREE_STATS
    int ts_tree = 0, ts_miss = 0, ts_cons = 0, ts_miss_at[6] = {0, 0, 0, 0, 0, 0};
#endif

    // solver state of the interval in progress
    PairProblem pb;
    double x[2] = {0, 0}, f[2] = {0, 0}, J[2][2] = {{0, 0}, {0, 0}}, g[2] = {0, 0}, xe[2] = {0, 0}, p[2] = {0, 0}, vk[3] = {0, 0, 0};
    double cost = 0.0, Delta = 0.0, alpha = 0.0, predicted = 0.0;
    int nfev = 0;
    bool first = false, in_solve = false;
    bool yielded = false;      // packed launch: this chain left at interval t for correct_resume_kernel
    bool noise_go = false;     // default fit: the solve went on past a gradient test its noise-free residual satisfied (ect_noise_continues)
    bool stalled = false;      // default fit: the stall rule returned the starting point (trace bit MISTI_TRACE_STALL_BIT; nfev stays what was evaluated)
    // speculative slots (one chain per wave only): SPEC_SLOTS x 6 lanes, slot 0 = the point the solver asked for
#ifndef MISTI_SPEC
#define MISTI_SPEC 1
#endif
    constexpr bool SPEC = MISTI_SPEC && (GROUP == 64) && !TAIL;
    constexpr int SPEC_SLOTS = 10;
    const int slot_of_lane = SPEC ? (lane < 6 * SPEC_SLOTS ? lane / 6 : 0) : 0;
    const int role = SPEC ? (lane < 6 * SPEC_SLOTS ? lane % 6 : (lane - 6 * SPEC_SLOTS) % 6) : sub;   // lanes beyond the slots repeat slot 0
    int spec_axis = -1;        // >= 0: the solver is in the exact rank-one regime and moves along this axis
    int vk_base = 0;           // first lane of the slot whose state vectors vk belong to the accepted point

    auto finish_interval = [&](double lc0, double lc1, int32_t word) -> bool {       // :345-350; false = correction failed
        if (sub == 0) { lc_w[2 * t] = lc0; lc_w[2 * t + 1] = lc1; if (sv_w) sv_w[t] = word; }
        if (uni<GROUP>(!(lc0 > 0) || !(lc1 > 0))) {
            status = (isnan(lc0) || isnan(lc1)) ? MISTI_NUMERIC : MISTI_CORR_FAILED;
            return false;
        }
        if (sub == 0) {
            double* r = tr_w + 6 * (t + 1);
            r[0] = ps.p[0][0]; r[1] = ps.p[1][0]; r[2] = ps.p[0][1]; r[3] = ps.p[1][1]; r[4] = ps.p[0][2]; r[5] = ps.p[1][2];
        }
        if (!TAIL && lc_sh) {
            // hand the interval to the trunk wave of this workgroup (correct_follow_kernel): rates, then the count
            if (sub == 0) { lc_sh[2 * t] = lc0; lc_sh[2 * t + 1] = lc1; }
            lds_order();
            if (sub == 0) lds_put(flags, t + 1);
        }
        ++t;
        return true;
    };

#ifdef MISTI_STAMP
    long long c_adv = 0, c_batch = 0, c_book = 0, c_t0 = 0, c_collect = 0, c_update = 0, c_next = 0, c_tree = 0;
#define STAMP(acc) { long long now_ = clock64(); acc += now_ - c_t0; c_t0 = now_; }
#else
#define STAMP(acc)
#endif
    bool active = status == MISTI_OK;
    if (PRE) {
        // What an interval needs that does not depend on the recursion - migration rates, pulse, exp(-lh T) of both
        // genomes - for all intervals at once, one interval per lane, instead of a band-table walk and two exponentials on
        // the critical path of every interval.  Same expressions as below, so the same bits.
        if (active)
            for (int tt = lane; tt < G.split; tt += 64) {
                double a0, a1, b0, b1;
                mod.mig(tt, a0, a1);
                mod.pulse(tt, b0, b1);
                const double Tt = G.T(tt);
                double* q = pre + 6 * tt;
                q[0] = a0; q[1] = a1; q[2] = b0; q[3] = b1; q[4] = exp(-G.lhk(tt, 0) * Tt); q[5] = exp(-G.lhk(tt, 1) * Tt);
            }
        lds_fence();
    }
#ifdef MISTI_STAMP
    c_t0 = clock64();
#endif
    while (uni<GROUP>(active)) {
        if (uni<GROUP>(!in_solve)) {
            // ---- advance over intervals until one needs the iterative solver ----------
            while (uni<GROUP>(t < G.split)) {
                double pu0, pu1, mu0, mu1, eh0 = 0.0, eh1 = 0.0;
                if (PRE) { const double* q = pre + 6 * t; mu0 = q[0]; mu1 = q[1]; pu0 = q[2]; pu1 = q[3]; eh0 = q[4]; eh1 = q[5]; }
                else { mod.pulse(t, pu0, pu1); mod.mig(t, mu0, mu1); }
                pulse_pairs<GROUP>(ps, pu0, pu1);                                       // :315-323
                double lh0 = G.lhk(t, 0), lh1 = G.lhk(t, 1);
                if (!correct) { if (uni<GROUP>(!finish_interval(lh0, lh1, 0))) break; continue; }    // :325-326
                const double T = G.T(t);
                const double s0 = (ps.p[0][0] + ps.p[0][1]) + ps.p[0][2];
                const double s1 = (ps.p[1][0] + ps.p[1][1]) + ps.p[1][2];
                if (m.mixture_th > 0.0) {                                               // CorrectLambda.py:267-272 (threshold 0 never fires)
                    double mix = 0.0;
                    for (int i = 0; i < 3; ++i) { double d = ps.p[0][i] / s0 - ps.p[1][i] / s1; mix += d * d; }
                    if (sqrt(mix) < m.mixture_th) { finish_interval(-1.0, -1.0, 0); break; }
                }
                if (uni<GROUP>(mu0 + mu1 < 1e-10)) {
                    double lc0, lc1;
                    int32_t word = solver_word(0, 0, 1);
                    if (CPFIT) {
                        // SolveNoMigration1 :213-235
                        double A1 = ps.p[0][0] / s0, A2 = ps.p[0][1] / s0, A3 = ps.p[1][0] / s1, A4 = ps.p[1][1] / s1;
                        double C1 = ps.p[0][2] / s0, C2 = ps.p[1][2] / s1;
                        double D = A1 * A4 - A2 * A3;
                        double B1 = A4 / D, B2 = -A2 / D, B3 = -A3 / D, B4 = A1 / D;
                        if (!PRE) { eh0 = exp(-lh0 * T); eh1 = exp(-lh1 * T); }
                        double X1 = eh0 - C1, X2 = eh1 - C2;
                        double y0 = B1 * X1 + B2 * X2, y1 = B3 * X1 + B4 * X2;
                        if (uni<GROUP>(y0 > 0 && y1 > 0)) { lc0 = -log(y0) / T; lc1 = -log(y1) / T; }
                        else { lc0 = lc1 = -1.0; }
                    } else {
                        // SolveNoMigration :253-264: bounded 2-D fit of the conditional expected coalescence time
                        double pr[2][3];
                        for (int i = 0; i < 3; ++i) { pr[0][i] = ps.p[0][i] / s0; pr[1][i] = ps.p[1][i] / s1; }
                        double tgt0 = ect_one_pop(lh0, T), tgt1 = ect_one_pop(lh1, T);
                        auto resid = [&](const double l[2], double ff[2]) {          // LambdaSystemNoMigration :237-251
                            double e0 = exp(-l[0] * T), e1 = exp(-l[1] * T);
                            double n0 = ect_noncond(l[0], T), n1 = ect_noncond(l[1], T);
                            for (int k = 0; k < 2; ++k) {
                                double pnc = pr[k][0] * e0 + pr[k][1] * e1 + pr[k][2];
                                double ct = (pr[k][0] * n0 + pr[k][1] * n1) / (1.0 - pnc);
                                ff[k] = ct - (k ? tgt1 : tgt0);
                            }
                        };
                        double xx[2] = {lh0, lh1};
                        word = trf_bounded<2>(resid, xx, 0.01 * fmin(lh0, lh1));
                        lc0 = xx[0]; lc1 = xx[1];
                    }
                    double e0 = exp(-lc0 * T), e1 = exp(-lc1 * T);
                    for (int k = 0; k < 2; ++k) { ps.p[k][0] *= e0; ps.p[k][1] *= e1; }
                    if (uni<GROUP>(!finish_interval(lc0, lc1, word))) break;
                    continue;
                }
                // migrating interval: set the 2x2 problem up (:278-305) and leave the advance loop
                double n0 = 0, n1 = 0, nd = 0;
                for (int i = 0; i < 3; ++i) { n0 += ps.p[0][i] * ps.p[0][i]; n1 += ps.p[1][i] * ps.p[1][i]; double d = ps.p[0][i] - ps.p[1][i]; nd += d * d; }
                bool averaged = false;
                if (uni<GROUP>(nd < 0.0004 * fmin(n0, n1))) { double mean = (lh0 + lh1) / 2.0; lh0 = lh1 = mean; averaged = true; }   // normD < 0.02 min(norms), squared
                pb.mu0 = mu0 * T; pb.mu1 = mu1 * T;                                      // stretch :293-298
                const double lhs0 = lh0 * T, lhs1 = lh1 * T;
                const bool g1 = role & 1;                                                // this lane's genome
                for (int i = 0; i < 3; ++i) pb.Pk[i] = g1 ? ps.p[1][i] : ps.p[0][i];
                pb.sk = g1 ? s1 : s0;
                // "empty": below 1e-30 of the genome's total - what one interval with rate x length > 70 leaves; far below
                // the rounding of every sum the state enters (it is dropped, not carried)
                pb.red = !CPFIT ? 0 : (pb.mu1 == 0.0 && ps.p[0][0] <= PAIR_EMPTY * s0 && ps.p[1][0] <= PAIR_EMPTY * s1) ? 1
                                    : (pb.mu0 == 0.0 && ps.p[0][1] <= PAIR_EMPTY * s0 && ps.p[1][1] <= PAIR_EMPTY * s1) ? 2 : 0;
                if (uni<GROUP>(!PRE || averaged)) { eh0 = exp(-lhs0); eh1 = exp(-lhs1); }
                if (CPFIT) { const double t0_ = eh0 * s0, t1_ = eh1 * s1; pb.tgtk = g1 ? t1_ : t0_; }
                else {
                    double pa = eh0, pbb = eh1;                                          // ExpectedCoalTimeOnePopTmp, T = 1
                    const double t0_ = 1.0 / lhs0 - 1.0 / (1.0 / pa - 1.0);
                    const double t1_ = 1.0 / lhs1 - 1.0 / (1.0 / pbb - 1.0);
                    pb.tgtk = g1 ? t1_ : t0_;
                }
                xe[0] = lhs0; xe[1] = lhs1;
                first = true; in_solve = true; spec_axis = -1;
                break;
            }
            if (uni<GROUP>(!in_solve)) { active = false; break; }  // reached the split, or failed
        }
        STAMP(c_adv)
        // ---- yield (packed launches only) ----------------------------------------------------------------------------------
        // A solve that has not converged after yield_nfev evaluations is a rate running away: tens to hundreds of evaluations,
        // interval after interval, each of them alone in its six lanes - a packed launch ends with its few such chains (16 384
        // random starts: 9 % of the chains, 3 x the mean evaluation count; the average wave lives a third of the kernel).  The chain
        // leaves here and is RESUMED at this interval, one chain per wave with the speculation tree, by correct_resume_kernel.
        if (!TAIL && yield_nfev > 0 && in_solve && !first && nfev >= yield_nfev) { yielded = true; active = false; break; }
        // ---- one residual batch: the point the solver needs in slot 0, guesses of its successors in the other slots ----
        // One chain per wave leaves 58 of 64 lanes idle during an evaluation.  In the exact rank-one regime (a rate
        // decoupled after a runaway: its Jacobian column is exactly zero and the solver moves along the other axis by
        // +-Delta, see next_step) the iteration is a walk on a line whose next points are computable before the trial's
        // residual is known: the trial is accepted and the radius doubled (next point xe +- 2 Delta) or it is rejected
        // and the radius quartered (next point x + sign Delta/4) - 98 % of the steps of the reference's traces
        // (tests/golden/golden_traces.json.gz) - and the same again one level down.  The tree of those successors
        //        slot 0  the trial                     1, 2  0 accepted, doubled, then + / -        3  0 rejected
        //        4, 5    1 / 2 rejected                6, 7  3 accepted, doubled, then + / -        8  3 rejected
        //        9       6 rejected (the likeliest third step: rejected, accepted +, rejected)
        // is evaluated in the idle lanes (slot s = lanes 6 s .. 6 s + 5) by the same pure function of the point, and the
        // bookkeeping of all of them is then done at once, one hypothesis per slot (tree consume below): up to four
        // solver steps per pass, bit for bit the same iteration.  A guess never consumed costs nothing.
        double px0 = xe[0], px1 = xe[1];
        const bool tree = SPEC && uni<GROUP>(spec_axis >= 0 && !first);       // wave-uniform
        double Dpre_l = Delta, ps_l = 0.0;                                    // per slot: radius and signed step of its hypothesis
        if (tree) {
            const int k = __builtin_amdgcn_readfirstlane(spec_axis);
            const double pk = k == 0 ? p[0] : p[1];
            const bool pos = pk > 0;
            const double D2 = 2.0 * Delta;
            const double dq0 = 0.25 * sqrt64(p[0] * p[0] + p[1] * p[1]);      // radius after rejecting the trial (tr_update)
            const double dq1 = 0.25 * sqrt64(D2 * D2);                        // ... after accepting it, doubling, and rejecting that
            const double dq3 = 0.25 * sqrt64(dq0 * dq0);                      // ... after rejecting twice
            const double xk = k == 0 ? x[0] : x[1], xek = k == 0 ? xe[0] : xe[1];
            const double u = xk + (pos ? dq0 : -dq0);                         // slot 3's point
            const int sl = slot_of_lane;
            const double dq0x2 = 2.0 * dq0;
            const double dq6 = 0.25 * sqrt64(dq0x2 * dq0x2);                  // ... after rejecting, accepting the retry, doubling, rejecting
            const bool root = sl == 0 || sl == 3 || sl == 8;                  // hypotheses that keep the current point x
            Dpre_l = sl == 0 ? Delta : sl <= 2 ? D2 : sl == 3 ? dq0 : sl <= 5 ? dq1 : sl <= 7 ? dq0x2 : sl == 8 ? dq3 : dq6;
            const bool pos_l = root ? pos : (sl == 1 || sl == 4 || sl == 6 || sl == 9);
            ps_l = sl == 0 ? pk : (pos_l ? Dpre_l : -Dpre_l);
            const double from = root ? xk : (sl == 6 || sl == 7 || sl == 9) ? u : xek;
            const double at = from + ps_l;
            if (sl >= 1 && sl <= 9) { px0 = k == 0 ? at : xe[0]; px1 = k == 1 ? at : xe[1]; }
        }
        double res, w[3];
        bool guard_l = false;
        pair_eval<CPFIT>(pb, px0, px1, role, dg, res, w, guard_l);
        dg.evals += 1;
        STAMP(c_batch)
        int slot = 0;
        bool stop = false;
        bool look = false;
        if (tree) {
            // ---- tree consume: the bookkeeping of every slot under its own hypothesis, all slots at once ------------
            const int k = __builtin_amdgcn_readfirstlane(spec_axis);
            const int sl = slot_of_lane;
            double fnl[2], Jnl[2][2];
            bool finl;
            slot_collect(res, px0, px1, 6 * sl, fnl, Jnl, finl);
            const double cnl = 0.5 * (fnl[0] * fnl[0] + fnl[1] * fnl[1]);
            // the state a hypothesis starts from: the current one (slots 0, 3, 8), the trial accepted (1, 2, 4, 5: slot 0's
            // evaluation) or the retry accepted (6, 7: slot 3's evaluation).  One gather per quantity: lane 60 (idle, it
            // repeats slot 0) is lent the current state, so every slot pulls from lane 60, 0 or 18.
            double xp[2], fp[2], Jp[2][2], cp;
            {
                const bool lend = lane == 60;
                const int src = (sl == 1 || sl == 2 || sl == 4 || sl == 5) ? 0 : (sl == 6 || sl == 7 || sl == 9) ? 18 : 60;
                fp[0] = __shfl(lend ? f[0] : fnl[0], src, 64); fp[1] = __shfl(lend ? f[1] : fnl[1], src, 64);
                for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) Jp[r][c] = __shfl(lend ? J[r][c] : Jnl[r][c], src, 64);
                cp = __shfl(lend ? cost : cnl, src, 64);
                xp[0] = __shfl(lend ? x[0] : px0, src, 64); xp[1] = __shfl(lend ? x[1] : px1, src, 64);
            }
            double gp[2], pl[2];
            gp[0] = Jp[0][0] * fp[0] + Jp[1][0] * fp[1];
            gp[1] = Jp[0][1] * fp[0] + Jp[1][1] * fp[1];
            pl[0] = k == 0 ? ps_l : 0.0; pl[1] = k == 1 ? ps_l : 0.0;
            const double predl = predicted_of(Jp, pl, gp);
            double Dl = Dpre_l, dql;
            int terml;
            bool accl;
            tr_update(finl, pl, xp, cp, cnl, predl, Dl, dql, terml, accl);
            // the state after the step and what the solver does next (the tests of the serial loop below and of next_step)
            double xs[2], fs[2], Js[2][2], gs[2];
            xs[0] = accl ? px0 : xp[0]; xs[1] = accl ? px1 : xp[1];
            fs[0] = accl ? fnl[0] : fp[0]; fs[1] = accl ? fnl[1] : fp[1];
            const double cs = accl ? cnl : cp;
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) Js[r][c] = accl ? Jnl[r][c] : Jp[r][c];
            gs[0] = Js[0][0] * fs[0] + Js[1][0] * fs[1];
            gs[1] = Js[0][1] * fs[0] + Js[1][1] * fs[1];
            const double g_norm = fmax(fabs(gs[0]), fabs(gs[1]));
            const bool donel = terml != 0 || (accl && (g_norm < LSQ_GTOL || !(g_norm < INFINITY)));
            const bool z0 = Js[0][0] == 0.0 && Js[1][0] == 0.0, z1 = Js[0][1] == 0.0 && Js[1][1] == 0.0;
            const bool same_axis = (z0 != z1) && (z0 ? 1 : 0) == k;
            const bool plus = !((k == 0 ? gs[0] : gs[1]) > 0);                 // next_step: pk = gk > 0 ? -Delta : Delta
            // outcome: 1 / 2 accepted, radius doubled, next step + / -; 3 rejected, radius quartered, same direction again;
            // 0 anything else (termination, a kept radius, another regime): the serial loop below takes it from there
            int codel = 0;
            if (!donel && same_axis) {
                if (accl) codel = Dl == 2.0 * Dpre_l ? (plus ? 1 : 2) : 0;
                else codel = (Dl == dql && plus == (ps_l > 0)) ? 3 : 0;
            }
            // ---- walk the tree along the outcomes that came true (wave-uniform) ----
            int cur = 0, last = -1, last_acc = -1, consumed = 0;
            for (;;) {
                if (nfev + 1 >= max_nfev) break;                               // the budget test belongs to the serial loop
                const int at = 6 * cur;
                const int code = __builtin_amdgcn_readlane(codel, at);
                if (code == 0) break;
                if (it_w && sub == 0 && nfev < MISTI_TRACE_MAX_ITER) {
                    double* r = it_w + ((int64_t)t * MISTI_TRACE_MAX_ITER + nfev) * 2;
                    r[0] = bcast(px0, at); r[1] = bcast(px1, at);
                }
                ++nfev; ++consumed; last = cur;
                if (__builtin_amdgcn_readlane((int)guard_l, at)) dg.guard = true;
                if (__builtin_amdgcn_readlane((int)accl, at)) last_acc = cur;
                int child = -1;
                if (cur == 0) child = code;
                else if (cur == 1) child = code == 3 ? 4 : -1;
                else if (cur == 2) child = code == 3 ? 5 : -1;
                else if (cur == 3) child = code == 1 ? 6 : code == 2 ? 7 : 8;
                else if (cur == 6) child = code == 3 ? 9 : -1;
#ifdef MISTI_TREE_STATS
                if (child < 0) { ts_miss += 1; ts_miss_at[cur == 1 || cur == 2 ? 0 : cur == 4 || cur == 5 ? 1 : cur == 7 ? 2 : cur == 8 ? 3 : cur == 9 ? 4 : 5] += 1; }
#endif
                if (child < 0) break;
                cur = child;
            }
#ifdef MISTI_TREE_STATS
            ts_tree += 1; ts_cons += consumed;
#endif
            if (consumed > 0) {
                const int at = 6 * last;
                x[0] = bcast(xs[0], at); x[1] = bcast(xs[1], at);
                f[0] = bcast(fs[0], at); f[1] = bcast(fs[1], at);
                for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) J[r][c] = bcast(Js[r][c], at);
                cost = bcast(cs, at);
                Delta = bcast(Dl, at);
                if (last_acc >= 0) { vk[0] = w[0]; vk[1] = w[1]; vk[2] = w[2]; vk_base = 6 * last_acc; }
                g[0] = J[0][0] * f[0] + J[1][0] * f[1];
                g[1] = J[0][1] * f[0] + J[1][1] * f[1];
                // the step after a consumed slot is again the rank-one one along k (that is what its outcome code says):
                // next_step's rank-one branch, without the rest of it
                const double pk = (k == 0 ? g[0] : g[1]) > 0 ? -Delta : Delta;
                p[0] = k == 0 ? pk : 0.0; p[1] = k == 1 ? pk : 0.0;
                alpha = 0.0;
                predicted = predicted_of(J, p, g);
                xe[0] = x[0] + p[0]; xe[1] = x[1] + p[1];
                dg.spec += consumed - 1;
                look = true;
            }
            STAMP(c_tree)
        }
        // ---- consume: trust-region bookkeeping (trf_no_bounds) for the needed point, and again while the next point
        //      asked for is one of those evaluated in this pass ---------------------------------------------------
        for (;; look = true) {
            if (look) {
                // was the point now needed evaluated in this pass?  (lanes of role 0 carry their slot's point)
                const unsigned long long hit = __ballot(role == 0 && lane < 6 * SPEC_SLOTS && px0 == xe[0] && px1 == xe[1]);
                if (hit == 0) break;
                slot = (__ffsll((long long)hit) - 1) / 6;
                dg.spec += 1;
            }
            double fn[2], Jn[2][2];
            bool finite;
            if (it_w && sub == 0) {
                const int i = first ? 0 : nfev;
                if (i < MISTI_TRACE_MAX_ITER) { double* r = it_w + ((int64_t)t * MISTI_TRACE_MAX_ITER + i) * 2; r[0] = xe[0]; r[1] = xe[1]; }
            }
            const int base = SPEC ? 6 * slot : 0;
            pair_collect<GROUP>(res, xe, base, fn, Jn, finite);
            STAMP(c_collect)
            if (SPEC) { if (__builtin_amdgcn_readlane((int)guard_l, __builtin_amdgcn_readfirstlane(base))) dg.guard = true; }
            else if (guard_l) dg.guard = true;
            bool accept = false, done = false;
            int term = 0;
            double cost_new = 0.5 * (fn[0] * fn[0] + fn[1] * fn[1]);
            if (uni<GROUP>(first)) {
                first = false;
                nfev = 1;
                if (uni<GROUP>(!finite)) { status = MISTI_NUMERIC; active = false; stop = true; break; }   // SciPy raises on a non-finite start
                Delta = sqrt64(xe[0] * xe[0] + xe[1] * xe[1]);
                if (Delta == 0) Delta = 1.0;
                alpha = 0.0;
                accept = true;
#ifndef MISTI_NO_STALL_RULE
                if (!CPFIT) {
                    // ---- default fit: would the REFERENCE's solve get anywhere from here? ---------------------------------------------
                    // Its residual T M^-1 e^{MT} p - M^-2 (e^{MT} - I) p carries ~eps / |M T|^2 of rounding noise (see ect_noise_continues); on a
                    // very short interval (rate x length ~1e-4: two nearly coincident time points of the merged grid) that noise, divided by the
                    // forward-difference step h = 1.5e-8, is LARGER than the Jacobian itself.  SciPy's iteration then takes steps a tenth of the
                    // needed size in random directions, rejects most of them, quarters its radius until xtol fires (status 3 after 14 - 23
                    // evaluations) and returns a point within ~0.3 % of where it started - 4 - 10 % from the root the noise-free series finds in
                    // three evaluations.  In the reference's own traces (tests/golden/*_traces.json.gz: 16 200 two-population default-fit solves
                    // of BASELINE's grids and the held-out grid config2b) EVERY one of the 205 solves with 0.29 W / h >= 2 |J| ended that way, and
                    // none of the 15 000 with 0.29 W / h <= 0.04 |J| (in between: 10 % of 239 at 0.15).  The starting point is what the reference
                    // returns there, so it is what is returned here: status 3 (the reference's xtol), with the trace word saying so - bit
                    // MISTI_TRACE_STALL_BIT - and nfev = 1, the evaluations actually made (the reference's own count there is 14 - 23; a
                    // made-up 19 used to stand here: ADVICE r5).  Its model of W is the one of the
                    // noise rule below (eps / (0.7 x 2 min(d0, d1)^2), calibrated in round 3).
                    // (Tried: no stall where one Gauss-Newton step of the noise-free residual leaves the positive quadrant - the reference fails
                    // there on config 3's start 4908 in all of its runs - : config 3 under the default fit then has 41 instead of 40 status
                    // mismatches and 2 instead of 0 candidates outside in the first pass; not kept.)
                    const double dmin = fmin(2.0 * pb.mu0 + xe[0], 2.0 * pb.mu1 + xe[1]);
                    const double wm = (0.5 * LSQ_EPS / 0.7) / (dmin * dmin);
                    const double jmax = fmax(fmax(fabs(Jn[0][0]), fabs(Jn[0][1])), fmax(fabs(Jn[1][0]), fabs(Jn[1][1])));
                    const double hh = fmin(fabs(fd_step(xe[0])), fabs(fd_step(xe[1])));
                    if (dmin > 0.0 && 0.29 * wm > jmax * hh) { term = 3; stalled = true; }
                }
#endif
            } else {
                ++nfev;
                const double Delta_old = Delta;
                double dq;
                tr_update(finite, p, x, cost, cost_new, predicted, Delta, dq, term, accept);
                if (uni<GROUP>(finite && term == 0 && alpha != 0.0)) alpha *= Delta_old * rcp64(Delta);   // trf.py:515
            }
            // accepted: the trial becomes the current point (selects; g recomputed from whatever (J, f) is current)
            const double xo0 = x[0], xo1 = x[1];         // base point of the step p
            x[0] = accept ? xe[0] : x[0]; x[1] = accept ? xe[1] : x[1];
            f[0] = accept ? fn[0] : f[0]; f[1] = accept ? fn[1] : f[1];
            cost = accept ? cost_new : cost;
            for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) J[r][c] = accept ? Jn[r][c] : J[r][c];
            vk[0] = accept ? w[0] : vk[0]; vk[1] = accept ? w[1] : vk[1]; vk[2] = accept ? w[2] : vk[2];
            vk_base = accept ? base : vk_base;           // the lanes whose state vectors belong to the accepted point
            g[0] = J[0][0] * f[0] + J[1][0] * f[1];
            g[1] = J[0][1] * f[0] + J[1][1] * f[1];
            STAMP(c_update)
            if (term != 0) done = true;
            else if (accept) {
                const double g_norm = fmax(fabs(g[0]), fabs(g[1]));
                bool g_stop = g_norm < LSQ_GTOL;
#ifndef MISTI_NOISE_PROBE
#define MISTI_NOISE_PROBE 1
#endif
#ifndef MISTI_NO_NOISE_RULE           // diagnostic build switch: the solver exactly as SciPy's tests see the noise-free residual
                if (!CPFIT) {
                    // the gradient test as the reference's noisy residual would see it (ect_noise_continues): only after a real step
                    if (uni<GROUP>(g_stop && nfev > 1 && nfev < max_nfev)) {
                        // Measuring the noise costs two evaluations of both forms (behind every solve it made the default fit's chains
                        // half again as slow, and with several chains per wave each measurement runs with one chain's lanes).  The
                        // width follows eps / min(d0, d1)^2 - the 1 / |M|^2 lost digits - within a factor 10 either way on 97 % of 1 020
                        // sampled solves (median 0.7): where the test's outcome is the same at a tenth and at ten times that model, it is
                        // taken from the model; the measurement decides the band in between.
                        const double ph = fmax(fabs(p[0] / fd_step(xo0)), fabs(p[1] / fd_step(xo1)));
                        const double dmin = fmin(2.0 * pb.mu0 + xo0, 2.0 * pb.mu1 + xo1);
                        const double wm = (0.5 * LSQ_EPS / 0.7) / (dmin * dmin);
                        const double nb = 0.29 * wm * ph;
                        const double jf2 = (J[0][0] * J[0][0] + J[0][1] * J[0][1]) + (J[1][0] * J[1][0] + J[1][1] * J[1][1]);
                        const double g2 = g_norm * g_norm, t2 = LSQ_GTOL * LSQ_GTOL;
                        const bool surely_on = dmin > 0.0 && g2 + 0.01 * (jf2 > 0 ? fmin(fmin(J[0][0] * J[0][0] + J[1][0] * J[1][0], J[0][1] * J[0][1] + J[1][1] * J[1][1]), jf2) : 0.0) * nb * nb >= t2;
                        const bool maybe_on = !(dmin > 0.0) || g2 + 100.0 * jf2 * nb * nb >= t2;
                        if (uni<GROUP>(surely_on)) { g_stop = false; noise_go = true; }
                        else if (uni<GROUP>(maybe_on)) {
#if MISTI_NOISE_PROBE
                            g_stop = !ect_noise_continues<GROUP>(pb, xo0, xo1, p, J, g, role);
#else
                            g_stop = !(g2 + jf2 * nb * nb >= t2);        // diagnostic build: the model alone
#endif
                            if (!g_stop) noise_go = true;
                        }
                    }
                }
#endif
                if (g_stop || nfev >= max_nfev || !(g_norm < INFINITY)) done = true;
            } else if (nfev >= max_nfev) done = true;
            if (uni<GROUP>(done)) {
                dg.max_nfev = nfev > dg.max_nfev ? nfev : dg.max_nfev;
                for (int i = 0; i < 3; ++i) { ps.p[0][i] = gbcast<GROUP>(vk[i], vk_base + 0); ps.p[1][i] = gbcast<GROUP>(vk[i], vk_base + 1); }   // :313-317
                in_solve = false;
                // OptimizeResult.status: the termination test that fired, 1 = gtol, 0 = evaluation budget (trf.py:452-456,556-558)
                const int code = term != 0 ? term : ((accept && fmax(fabs(g[0]), fabs(g[1])) < LSQ_GTOL) ? 1 : 0);
                const int32_t went_on = (noise_go ? MISTI_TRACE_NOISE_BIT : 0) | (stalled ? MISTI_TRACE_STALL_BIT : 0);
                noise_go = false;
                stalled = false;
                const double T = G.T(t);               // re-read (LDS) rather than carried through the solver loop
                if (uni<GROUP>(!finish_interval(x[0] / T, x[1] / T, solver_word(nfev, code, 3) | went_on))) { active = false; stop = true; }   // :312, :346-348
                break;
            }
            predicted = next_step<GROUP>(J, f, g, Delta, alpha, p, dg.lm, spec_axis);
            xe[0] = x[0] + p[0]; xe[1] = x[1] + p[1];
            STAMP(c_next)
            if (!SPEC) break;
        }
        if (uni<GROUP>(stop)) break;
        STAMP(c_book)
    }
    if (dg.guard) { status = MISTI_NUMERIC; if (!TAIL) t = 0; yielded = false; }       // an overflowing iterate was cut off somewhere
    if (sub == 0) {
        if (!TAIL) {
            double* r = cb.work + slot * 6;
            if (yielded) {
                cb.resume_t[slot] = t;
                cb.resume_list[atomicAdd(&cb.n_chains[3], 1)] = (int32_t)slot;
            } else {
                if (yield_nfev > 0) cb.resume_t[slot] = -1;               // a packed launch that lets chains yield: this one is complete (see chain_phase)
                cb.fail_t[slot] = (status == MISTI_OK) ? 0x7fffffff : t;   // t = the interval that failed
                cb.fail_status[slot] = status;
            }
            if (t_start > 0) { dg.evals += (int)r[0]; dg.dense += (int)r[1]; dg.terms += (int)r[2]; dg.spec += (int)r[3]; dg.lm += (int)r[5];
                               dg.max_nfev = dg.max_nfev > (int)r[4] ? dg.max_nfev : (int)r[4]; }     // counters of the part before the yield
            r[0] = dg.evals; r[1] = dg.dense; r[2] = dg.terms; r[3] = dg.spec; r[4] = dg.max_nfev; r[5] = dg.lm;
#ifdef MISTI_TREE_STATS
            r[1] = ts_tree; r[2] = ts_miss; r[3] = ts_cons; r[4] = ts_miss_at[0] * 1e6 + ts_miss_at[1] * 1e3 + ts_miss_at[2]; r[5] = ts_miss_at[3] * 1e6 + ts_miss_at[4] * 1e3 + ts_miss_at[5];
#endif
#ifdef MISTI_STAMP
            r[0] = (double)c_tree; r[1] = (double)(c_collect + c_update); r[2] = (double)c_next;
            r[3] = (double)c_adv; r[4] = (double)c_batch; r[5] = (double)c_book;
#endif
        } else {
            cb.tail_status[cand] = status;
        }
    }
}

#ifndef MISTI_DEFAULT_FIT_WAVES
#define MISTI_DEFAULT_FIT_WAVES 1
#endif
#ifndef MISTI_FOLLOW_WAVES
#define MISTI_FOLLOW_WAVES 2
#endif
template <bool CPFIT, int GROUP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CPFIT ? 2 : MISTI_DEFAULT_FIT_WAVES, CPFIT ? 2 : MISTI_DEFAULT_FIT_WAVES)))
void correct_kernel(DevModel m, int64_t n_items, ChainBufs cb, const double* __restrict__ split_time, const double* __restrict__ params, int yield_nfev) {
    extern __shared__ double lds[];
    {   // candidate -> chain, once, by blocks that mostly have nothing else to do (the launch has one block per
        // 64/GROUP candidates, the chains occupy the first few): later kernels read it without the slot indirection
        const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
        if (i < n_items) cb.of[i] = cb.slot_chain[cb.slot_of[i]];
    }
    correct_body<CPFIT, GROUP, false>(m, n_items, cb, split_time, params, (int64_t)blockIdx.x, lds, nullptr, nullptr, nullptr, cb.chain_order,
                                      GROUP == 64 ? 0 : yield_nfev);
}

// The chains that yielded in a packed launch, one per wave (speculation tree, per-interval constants in LDS), pulled from the
// list the packed launch left; workgroups beyond its length leave at once.  LDS: grid staging [3 numT] | constants [6 numT].
template <bool CPFIT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CPFIT ? MISTI_FOLLOW_WAVES : 1, CPFIT ? MISTI_FOLLOW_WAVES : 1)))
void correct_resume_kernel(DevModel m, int64_t n_items, ChainBufs cb, const double* __restrict__ split_time, const double* __restrict__ params) {
    extern __shared__ double lds[];
    __shared__ int next;
    const int n_res = cb.n_chains[3];
    if ((int)blockIdx.x >= n_res) return;
    // the launch waits for its longest chain, and the phase-1 candidate waves of the same batch run beside it (launch_spectrum): the
    // chains get the issue priority on the SIMDs they share
    __builtin_amdgcn_s_setprio(3);
    double* pre = lds + 3 * (size_t)m.numT;
    for (;;) {
        if (threadIdx.x == 0) next = atomicAdd(&cb.n_chains[2], 1);
        __syncthreads();
        const int pos = next;
        __syncthreads();
        if (pos >= n_res) break;
        const int64_t ch = cb.resume_list[pos];
        correct_body<CPFIT, 64, false, true>(m, n_items, cb, split_time, params, ch, lds, nullptr, nullptr, pre, nullptr, 0, cb.resume_t[ch]);
        __syncthreads();
    }
}

// A packed launch whose long chains YIELD to the resume launch finishes most chains itself (BASELINE config 3: 91 %).  What waits for a
// chain - its trunk, the tails and the candidate kernel of its members - therefore runs in two PHASES: phase 1 (the chains the packed
// launch completed) on a second stream BESIDE the resume launch, phase 2 (the chains that yielded) behind it; phase 0 = no split (every
// other launch shape).  resume_t[ch] is -1 for a completed chain, the interval it resumes at otherwise.
__device__ __forceinline__ bool wrong_phase(const ChainBufs& cb, int64_t ch, int phase) {
    return phase != 0 && ((cb.resume_t[ch] >= 0) != (phase == 2));
}

// ---- replicate epilogue pieces ----
// llh_const of SetJAFS (MigrationInference.py:217-227) for one replicate
__device__ __forceinline__ double llh_const_of(const double* __restrict__ row, int unfolded) {
    const double* d = row + 1;
    double snps = 0.0;
    for (int i = 0; i < 7; ++i) snps += d[i];
    double c = lgamma(snps + 1.0);
    if (unfolded) { for (int i = 0; i < 7; ++i) c -= lgamma(d[i] + 1.0); }
    else {
        c -= lgamma(d[0] + d[6] + 1.0);
        c -= lgamma(d[1] + d[5] + 1.0);
        c -= lgamma(d[2] + d[4] + 1.0);
        c -= lgamma(d[3] + 1.0);
    }
    return c;
}

// Multinomial log-likelihood of one replicate (MigrationInference.py:600-609) from the logs of the
// spectrum classes (folded: classes 0+6, 1+5, 2+4, 3).  Shared by llk_kernel and the epilogue of
// the spectrum kernel so that both give the same bits.
__device__ __forceinline__ double log_class(const double* J, int i, int unfolded) {
    if (unfolded) return log(J[i]);
    return (i < 3) ? log(J[i] + J[6 - i]) : (i == 3 ? log(J[3]) : 0.0);
}
__device__ __forceinline__ double llk_of(const double* __restrict__ row, double cst, const double* lj, int unfolded) {
    const double* d = row + 1;
    double a = cst;
    if (unfolded) { for (int i = 0; i < 7; ++i) a = fma(d[i], lj[i], a); }
    else {
        a = fma(d[0] + d[6], lj[0], a);
        a = fma(d[1] + d[5], lj[1], a);
        a = fma(d[2] + d[4], lj[2], a);
        a = fma(d[3], lj[3], a);
    }
    return a;
}

// ---- two-population propagation shared by the trunk and the candidate kernel ----
// Row view of the 44-state generator for state `lane`, kept in registers.
struct TwoPopRow {
    int srcl[MAXNZ], knd[MAXNZ];
    double mlt[MAXNZ];
    double dc0, dc1, dc2, dc3;
    bool live;
    __device__ __forceinline__ void load(int lane) {
        for (int n = 0; n < MAXNZ; ++n) { srcl[n] = c_tab.src[n][lane]; knd[n] = c_tab.kind[n][lane]; mlt[n] = (double)c_tab.mult[n][lane]; }
        dc0 = c_tab.dcnt[0][lane]; dc1 = c_tab.dcnt[1][lane]; dc2 = c_tab.dcnt[2][lane]; dc3 = c_tab.dcnt[3][lane];
        live = lane < NS2;
    }
};

// AncientSampleP0 (TwoPopulations.py:246-262)
__device__ __forceinline__ void ancient_project(double* xbuf, int lane, double& x) {
    xbuf[lane] = x; lds_fence();
    double nx = 0.0;
    for (int a = 0; a < 2; ++a) if (lane == c_tab.anc_dst[a]) for (int j = 0; j < c_tab.anc_n[a]; ++j) nx += xbuf[c_tab.anc_src[a][j]];
    lds_fence();
    x = nx;
}

// One interval of the two-population loop (JAFSpectrum :483-502): pulse at the start of the
// interval, then x <- exp(M T) x and the occupation integral added to w_pre / w_post.
// lcb: this wave's (smoothed) rates in LDS.  Returns MISTI_OK / MISTI_NUMERIC / MISTI_STIFF.
__device__ __forceinline__ int twopop_interval(const TwoPopRow& R, const DevModel& m, const Model& mod, const Grid& G, const double* lcb,
                                               double* xbuf, int lane, int t, double& x, double& w_pre, double& w_post) {
    double pu0, pu1, mu0, mu1;
    mod.pulse(t, pu0, pu1);
    mod.mig(t, mu0, mu1);
    double pr = pu0 + pu1;
    if (pr > 0) {
        // PulseMigration (TwoPopulations.py:361-377)
        int from = pu0 > 0 ? 0 : 1;
        xbuf[lane] = x; lds_fence();
        double pw_s[5], pw_m[5];
        pw_s[0] = pw_m[0] = 1.0;
        for (int i = 1; i < 5; ++i) { pw_s[i] = pw_s[i - 1] * (1.0 - pr); pw_m[i] = pw_m[i - 1] * pr; }
        double nx = 0.0;
        int n = R.live ? c_tab.pulse_n[from][lane] : 0;
        for (int j = 0; j < n; ++j) {
            int ab = c_tab.pulse_ab[from][lane][j];
            int mul = ab >> 8, a = (ab >> 4) & 15, b = ab & 15;
            nx += xbuf[c_tab.pulse_src[from][lane][j]] * ((double)mul * pw_s[a] * pw_m[b]);
        }
        lds_fence();
        x = nx;
    }
    double la0 = lcb[2 * t], la1 = lcb[2 * t + 1];
    double T = G.T(t);
    // largest total exit rate over the 44 states (4 lineages dominate)
    double r40 = 6 * la0 + 4 * mu0, r04 = 6 * la1 + 4 * mu1;
    double r31 = 3 * la0 + 3 * mu0 + mu1, r13 = 3 * la1 + 3 * mu1 + mu0;
    double r22 = la0 + la1 + 2 * mu0 + 2 * mu1;
    double q = T * fmax(fmax(r40, r04), fmax(fmax(r31, r13), r22));
    if (!(q < 1e300)) return MISTI_NUMERIC;
    double rate[4] = {la0, la1, mu0, mu1};
    double cf[MAXNZ];
    for (int n = 0; n < MAXNZ; ++n) cf[n] = R.mlt[n] * rate[R.knd[n]] * T;
    const double dT = (R.dc0 * la0 + R.dc1 * la1 + R.dc2 * mu0 + R.dc3 * mu1) * T;   // exit rate x T of this state
    double wint = 0.0;
    if (q <= Q_SWITCH) {
        // uniformisation: M T = N - q I, p_{k+1} = N p_k/(k+1), i_{k+1} = (T p_k + q i_k)/(k+1)
        const double dg = q - dT;
        const double eq = exp(-q);
        double p = eq * x, ii = 0.0;
        double accp = p, acci = 0.0;
        // the number of terms from q alone (q is the same in every lane): see c_qmax
        int K = 1;
        for (int c = 0; c < QMAX_TABLE / 64; ++c) {
            const unsigned long long below = __ballot(c_qmax[1 + 64 * c + lane] < q);
            K += __popcll(below);
            if (below != ~0ull) break;
        }
        double inv_next = c_inv[1];
        for (int k = 1; k <= K; ++k) {
            const double inv = inv_next;
            inv_next = c_inv[k + 1];
            // (the four source states: an LDS write and four reads; the same gather through ds_bpermute was measured slower - config 3's
            // kernel 2 1.27 -> 1.62 ms)
            xbuf[lane] = p;
            lds_fence();
            double r0 = xbuf[R.srcl[0]], r1 = xbuf[R.srcl[1]], r2 = xbuf[R.srcl[2]], r3 = xbuf[R.srcl[3]];
            lds_fence();
            double pn = (dg * p + ((cf[0] * r0 + cf[1] * r1) + (cf[2] * r2 + cf[3] * r3))) * inv;
            ii = (T * p + q * ii) * inv;
            p = pn;
            accp += p; acci += ii;
        }
        x = accp; wint = acci;
    } else {
        // Talbot contour, conjugate pairs folded: f = 2 Re sum_{k upper} (-c_k) (z_k - M T)^-1 x
        double accp = 0.0, acci = 0.0;
        bool stalled = false;
        for (int nd = 0; nd < TALBOT_HALF; ++nd) {
            const double zr = c_tab.tal_zr[nd], zi = c_tab.tal_zi[nd], cr = c_tab.tal_cr[nd], ci = c_tab.tal_ci[nd];
            const double ar = zr + dT, ai = zi;
            const double den = 1.0 / (ar * ar + ai * ai);
            const double ir = ar * den, im = -ai * den;              // 1 / (z + D_i)
            double xr = x * ir, xi = x * im;
            int it = 0;
            for (; it < JACOBI_MAX; ++it) {
                xbuf[lane] = xr; xbuf[64 + lane] = xi;
                lds_fence();
                double sr = x + ((cf[0] * xbuf[R.srcl[0]] + cf[1] * xbuf[R.srcl[1]]) + (cf[2] * xbuf[R.srcl[2]] + cf[3] * xbuf[R.srcl[3]]));
                double si = (cf[0] * xbuf[64 + R.srcl[0]] + cf[1] * xbuf[64 + R.srcl[1]]) + (cf[2] * xbuf[64 + R.srcl[2]] + cf[3] * xbuf[64 + R.srcl[3]]);
                lds_fence();
                double nr = sr * ir - si * im, ni = sr * im + si * ir;
                bool moving = (fabs(nr - xr) + fabs(ni - xi)) > 4e-16 * (fabs(nr) + fabs(ni)) + 1e-300;
                xr = nr; xi = ni;
                if (!__any(moving)) break;
            }
            if (it >= JACOBI_MAX) stalled = true;
            const double wr = -(cr * xr - ci * xi), wi = -(cr * xi + ci * xr);
            const double zz = 1.0 / (zr * zr + zi * zi);
            const double gr = T * zr * zz, gi = -T * zi * zz;        // T / z
            accp += 2.0 * wr;
            acci += 2.0 * (wr * gr - wi * gi);
        }
        if (stalled) return MISTI_STIFF;
        x = accp; wint = acci;
    }
    if (t < m.sample_date) w_pre += wint; else w_post += wint;
    return MISTI_OK;
}

// time-weighted mean of genome k's rates over intervals [a, b) (one smoothing run, :392-403)
__device__ __forceinline__ double run_mean(const double* lc, int k, int a, int b, const Grid& G) {
    double acc = 0.0, tt = 0.0;
    for (int j = a; j < b; ++j) { double Tj = G.T(j); acc += lc[2 * j + k] * Tj; tt += Tj; }
    return acc / tt;
}

// Smooth (:380-405) on a wave's rates in LDS: time-weighted mean of lc over runs of constant lh,
// t < bound; a run is cut at `bound` (the candidate's split index).
__device__ __forceinline__ void smooth_rates(const DevModel& m, const Grid& G, double* lcb, int lane, int lo, int bound) {
    // intervals lo <= t < bound are smoothed (lo > 0: the caller does not need the others); wave-uniform skip of
    // the 64-interval slices outside that range
    double sm[SMOOTH_REPS][2];   // intervals rep*64+lane, both genomes
#pragma unroll
    for (int rep = 0; rep < SMOOTH_REPS; ++rep) {
        sm[rep][0] = sm[rep][1] = 0.0;
        if (rep * 64 >= bound || rep * 64 + 63 < lo) continue;
        int t = rep * 64 + lane;
        for (int k = 0; k < 2; ++k) {
            double v = 0.0;
            if (t >= lo && t < bound) {
                int a = m.run_start[k * m.numT + t];
                int b = m.run_end[k * m.numT + t];
                if (b > bound) b = bound;
                v = run_mean(lcb, k, a, b, G);
            }
            sm[rep][k] = v;
        }
    }
    lds_fence();
#pragma unroll
    for (int rep = 0; rep < SMOOTH_REPS; ++rep) {
        int t = rep * 64 + lane;
        if (t >= lo && t < bound) { lcb[2 * t] = sm[rep][0]; lcb[2 * t + 1] = sm[rep][1]; }
    }
    lds_fence();
}

// Trunk kernel.  Candidates of one chain (same parameters, different split) propagate the 44-state
// chain through the SAME intervals with the SAME rates up to the smoothing run their split cuts:
// the rates of interval t, the migration rates and the pulses do not depend on the split as long
// as every smoothing run touching [0, t] ends at or before it.  One wavefront per chain walks the
// chain's intervals once and stores, before each interval t, the state vector and both occupation
// integrals (3 x 44 doubles); a candidate then starts from the record of the first interval whose
// run its split cuts (t_own in spectrum_kernel) and adds only its own 0-5 intervals - on a
// split x rate grid ~14x less propagation work.  The arithmetic per interval is the same code
// (twopop_interval) on the same inputs, so a candidate's result is bit-identical with or without
// the trunk (tests/test_gpu_trunk.py).
// Active only when sharing pays: n_chains * TRUNK_MIN_SHARE <= n_cand (decided on the device, the
// chain count never visits the host).
__device__ __forceinline__ bool trunk_active(const ChainBufs& cb, int64_t n_cand) {
    const int64_t nch = cb.n_chains[0];
    return cb.trunk_cap > 0 && nch <= cb.trunk_cap && nch * TRUNK_MIN_SHARE <= n_cand;
}

__device__ __forceinline__
void trunk_body(const DevModel& m, int64_t n_cand, const double* __restrict__ params, const ChainBufs& cb, int64_t ch, double* lds, int phase = 0) {
    const int lane = lane_id();
    if (!trunk_active(cb, n_cand) || ch >= cb.n_chains[0]) return;
    if (wrong_phase(cb, ch, phase)) return;
    double* xbuf = lds;
    double* lcb = lds + 128;
    const int ft = cb.fail_t[ch];
    int Lt = chain_len(cb, ch);
    if (ft < Lt) Lt = ft;                                   // members beyond the failing interval have no value anyway
    const double* par = params ? params + (int64_t)cb.rep[ch] * m.n_param : nullptr;
    Grid G;
    G.times = m.times; G.lh = m.lh; G.numT0 = m.numT; G.numT = m.numT; G.split = Lt; G.ins = -1; G.frac = 0.0;
    Model mod{&m, par, Lt, {0, 0, 0, 0}};
    mod.bb = cb.bounds ? cb.bounds + (int64_t)cb.rep[ch] * 2 * m.n_band : nullptr;
    mod.cache();
    const double* lc_ch = cb.lc + ch * (int64_t)m.numT * 2;
    for (int i = lane; i < 2 * (m.numT + 1); i += 64) lcb[i] = ((i >> 1) < Lt) ? lc_ch[i] : 0.0;
    lds_fence();
    if (m.flags & MISTI_SMOOTH) smooth_rates(m, G, lcb, lane, 0, Lt);
    TwoPopRow R;
    R.load(lane);
    double x = (lane == 2) ? 1.0 : 0.0;
    double w_pre = 0.0, w_post = 0.0;
    double* rec = cb.trunk + ch * (int64_t)m.numT * TRUNK_REC;
    // Records are read from the first interval at which a member leaves the trunk (setup_kernel: slot_keep), only at intervals
    // where SOME split can leave it (m.leave_ok: with smoothing, starts of the runs a split can cut - a function of the model,
    // computed in misti_create) - and at the interval the trunk ends on, wherever that is.  On the headline grid a third of the
    // records lies before the first split and three quarters of the rest inside runs: never read, no longer written.
    const int t_first = m.numT - cb.slot_keep[cb.chain_slot[ch]];
    auto store = [&](int t, double xs, double ws0, double ws1) {
        if (R.live) { double* r = rec + (int64_t)t * TRUNK_REC; r[lane] = xs; r[NS2 + lane] = ws0; r[2 * NS2 + lane] = ws1; }
    };
    int ok = 0;
    bool stored = false;
    double xs = x, ws0 = w_pre, ws1 = w_post;
    for (int t = 0; t < m.numT; ++t) {
        xs = x; ws0 = w_pre; ws1 = w_post;
        stored = t >= t_first && m.leave_ok[t] != 0;
        if (stored) store(t, xs, ws0, ws1);
        ok = t;
        if (t >= Lt) break;
        if (t == m.sample_date) ancient_project(xbuf, lane, x);
        if (twopop_interval(R, m, mod, G, lcb, xbuf, lane, t, x, w_pre, w_post) != MISTI_OK) break;
    }
    if (!stored) store(ok, xs, ws0, ws1);
    if (lane == 0) cb.trunk_ok[ch] = ok;
}

// The trunk as a FOLLOWER of its chain: second wavefront of the chain's workgroup
// (correct_follow_kernel).  The chain wave hands over each corrected interval through LDS (rates,
// then a count); this wave propagates interval t as soon as the smoothing runs containing t are
// complete, so the trunk is finished almost when the chain is - its ~0.3 ms leave the critical
// path of a batch.  Same records, same arithmetic as trunk_body (run_mean + twopop_interval).  Both
// waves are resident by construction (one workgroup), and the wait is bounded: a follower that gives up
// publishes the prefix it has, which is all the candidate kernel relies on (trunk_ok).
__device__ __forceinline__
void trunk_follow(const DevModel& m, int64_t n_cand, const double* __restrict__ params, const ChainBufs& cb, int64_t ch, double* lds,
                  const double* lc_sh, volatile int* flags) {
    const int lane = lane_id();
    if (!trunk_active(cb, n_cand) || ch >= cb.n_chains[0]) return;
    double* xbuf = lds;
    double* lcb = lds + 128;
    const int len = chain_len(cb, ch);
    const double* par = params ? params + (int64_t)cb.rep[ch] * m.n_param : nullptr;
    Grid G;
    G.times = m.times; G.lh = m.lh; G.numT0 = m.numT; G.numT = m.numT; G.split = len; G.ins = -1; G.frac = 0.0;
    Model mod{&m, par, len, {0, 0, 0, 0}};
    mod.bb = cb.bounds ? cb.bounds + (int64_t)cb.rep[ch] * 2 * m.n_band : nullptr;
    mod.cache();
    for (int i = lane; i < 2 * (m.numT + 1); i += 64) lcb[i] = 0.0;
    lds_fence();
    TwoPopRow R;
    R.load(lane);
    double x = (lane == 2) ? 1.0 : 0.0;
    double w_pre = 0.0, w_post = 0.0;
    double* rec = cb.trunk + ch * (int64_t)m.numT * TRUNK_REC;
    const bool smooth = m.flags & MISTI_SMOOTH;
    const int t_first = m.numT - cb.slot_keep[cb.chain_slot[ch]];         // see trunk_body
    auto store = [&](int t, double xs, double ws0, double ws1) {
        if (R.live) { double* r = rec + (int64_t)t * TRUNK_REC; r[lane] = xs; r[NS2 + lane] = ws0; r[2 * NS2 + lane] = ws1; }
    };
    int ok = 0, have = 0, done = 0;
    bool stored = false;
    double xs = x, ws0 = w_pre, ws1 = w_post;
    long long spins = 0;
    for (int t = 0; t < m.numT; ++t) {
        xs = x; ws0 = w_pre; ws1 = w_post;
        stored = t >= t_first && m.leave_ok[t] != 0;
        if (stored) store(t, xs, ws0, ws1);
        ok = t;
        if (t >= len) break;
        int a0 = t, b0 = t + 1, a1 = t, b1 = t + 1;
        if (smooth) { a0 = m.run_start[t]; b0 = m.run_end[t]; a1 = m.run_start[m.numT + t]; b1 = m.run_end[m.numT + t]; }
        int need = b0 > b1 ? b0 : b1;
        if (need > len) need = len;
        while (have < need && !done) {
            done = lds_get(flags + 1);             // read `done` first: the count read after it is then final
            have = lds_get(flags);
            if (have >= need || done) break;
            __builtin_amdgcn_s_sleep(16);
            if (++spins > FOLLOW_SPIN_LIMIT) break;
        }
        lds_order();                               // the rates read below were written before the count read above
        if (have < need && !done) break;           // gave up waiting: keep the prefix
        const int cut = have >= need ? len : have; // the chain ended early (failure): runs are cut where it ended
        if (t >= cut) break;
        if (b0 > cut) b0 = cut;
        if (b1 > cut) b1 = cut;
        const double l0 = smooth ? run_mean(lc_sh, 0, a0, b0, G) : lc_sh[2 * t];
        const double l1 = smooth ? run_mean(lc_sh, 1, a1, b1, G) : lc_sh[2 * t + 1];
        if (lane == 0) { lcb[2 * t] = l0; lcb[2 * t + 1] = l1; }
        lds_fence();
        if (t == m.sample_date) ancient_project(xbuf, lane, x);
        if (twopop_interval(R, m, mod, G, lcb, xbuf, lane, t, x, w_pre, w_post) != MISTI_OK) break;
    }
    if (!stored) store(ok, xs, ws0, ws1);
    if (lane == 0) cb.trunk_ok[ch] = ok;
}

// Kernel 1 with the trunk following: 128-thread workgroups, wave 0 = the chain (one chain per wave),
// wave 1 = its trunk.  LDS (doubles): kernel-1 staging [3 numT] | trunk xbuf [128] + rates [2 (numT+1)] |
// hand-over rates [2 numT] | count, done flag [2] | per-interval constants of the chain (mu, pulse, exp(-lh T)) [6 numT].
// Two waves per SIMD for the --cpfit instantiation: the chain wave is bound by the latency of its own dependent
// instruction stream (one instruction per ~4.7 cycles), so a second wave on the SIMD costs it nothing (single batch
// 1.884 -> 1.888 ms) and with 20 batches in flight the chip holds twice as many chains (2.38e7 -> 2.62e7 evals/s).
// The register allocator gives up the 12 AGPRs for 28 bytes of scratch.  The default fit (256 VGPRs + 108 AGPRs) stays
// at one wave.
template <bool CPFIT>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(CPFIT ? MISTI_FOLLOW_WAVES : 1, CPFIT ? MISTI_FOLLOW_WAVES : 1)))
void correct_follow_kernel(DevModel m, int64_t n_items, ChainBufs cb, const double* __restrict__ split_time, const double* __restrict__ params) {
    extern __shared__ double lds[];
    double* tk = lds + 3 * (size_t)m.numT;
    double* lc_sh = tk + 128 + 2 * (size_t)(m.numT + 1);
    volatile int* flags = (volatile int*)(lc_sh + 2 * (size_t)m.numT);          // count, done | queue position of the chain in progress
    double* pre = lc_sh + 2 * (size_t)m.numT + 4;                              // per-interval constants of the chain in progress (after 8 ints: flags, queue position, role exchange)
    // candidate -> chain (see correct_kernel)
    for (int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 128) cb.of[i] = cb.slot_chain[cb.slot_of[i]];
    // Workgroups PULL chains from a queue, the longest first (cb.chain_order, sorted by setup_kernel; head = n_chains[2]): any
    // grid is correct, and with more chains than resident workgroups (1 024 two-wave workgroups at two waves per SIMD) a
    // workgroup that finishes a short chain takes the next one instead of waiting for a fixed stride partner.
    const int64_t n_live = cb.n_chains[0];
    if ((int64_t)blockIdx.x >= n_live) return;                                // more workgroups than chains (a generous grid): leave at once
    int* next = (int*)(flags + 2);
    // Which of the two waves runs the chain: the one on the SIMD that hosts fewer chain waves right now (of any context: a device-wide
    // table of counters per compute unit and SIMD).  A chain wave beside a trunk wave runs at full speed, beside another chain
    // wave it does not (1.66 against 1.96 ms on 1 024 chains), and which SIMDs a workgroup's waves land on is the dispatcher's choice.
    // Measured with 20 batches in flight on the headline grid: 3.60 -> 3.76e7 evals/s; alone nothing changes (the dispatcher's own
    // placement on an idle chip is already the good one).  The counters only steer - a stale or shared one costs speed, never a result.
    bool chain_role = threadIdx.x < 64;
    int32_t* my_load = nullptr;
    if (cb.simd_load) {
        int* ex = next + 1;                              // [4]: SIMD of wave 0, of wave 1, table index of the compute unit, role swap
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID: wave [3:0] simd [5:4] cu [11:8] sh [12] se [15:13]
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);         // HW_REG_XCC_ID [3:0]
        const int simd = (hw >> 4) & 3;
        if ((threadIdx.x & 63) == 0) ex[threadIdx.x >> 6] = simd;
        if (threadIdx.x == 0) ex[2] = (int)((((xcc & 7) * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)) * 4;
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t* t = cb.simd_load + ex[2];
            const int c0 = __hip_atomic_load(&t[ex[0]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int c1 = __hip_atomic_load(&t[ex[1]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int sw = c1 < c0 ? 1 : 0;              // wave 1 takes the chain
            ex[3] = sw;
            atomicAdd(&t[ex[sw]], 1);
        }
        __syncthreads();
        chain_role = ((int)(threadIdx.x >> 6) ^ ex[3]) == 0;
        my_load = cb.simd_load + ex[2] + ex[ex[3]];
        __syncthreads();
    }
    for (;;) {
        if (threadIdx.x == 0) { lds_put(flags, 0); lds_put(flags + 1, 0); *next = atomicAdd(&cb.n_chains[2], 1); }
        __syncthreads();
        const int64_t pos = *next;
        if (pos >= n_live) break;
        const int64_t ch = cb.chain_order[pos];
        if (chain_role) {
            // a launch that fills the chip: the longest chains - the ones the launch waits for - get the issue priority on the SIMD
            // they share with another wave (s_setprio; measured on 1 024 chains: 2.04 -> 1.92 ms per call; on the headline grid's 64
            // chains it changes nothing alone and costs 1 % with 20 batches in flight, so not there)
            if (n_live > 256) {
                if (4 * pos < n_live) __builtin_amdgcn_s_setprio(3);
                else if (2 * pos < n_live) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(1);
            }
            correct_body<CPFIT, 64, false, true>(m, n_items, cb, split_time, params, ch, lds, lc_sh, flags, pre);
            lds_order();
            if (lane_id() == 0) lds_put(flags + 1, 1);                       // whatever way the chain ended
        } else {
            trunk_follow(m, n_items, params, cb, ch, tk, lc_sh, flags);
        }
        __syncthreads();                                                     // both waves are done with the hand-over area
    }
    if (my_load && threadIdx.x == 0) atomicSub(my_load, 1);
}

// Everything that waits for the chains and that the candidate kernel waits for, in ONE launch: the
// trunks (blocks below trunk_cap, dispatched first: they are the long ones) and the tails of the
// lambda-correction (the shortened last interval of candidates with a fractional split).  A launch
// costs a queue round trip, which is what limits the rate when many batches are in flight.
template <bool CPFIT, int GROUP>
__global__ __launch_bounds__(64)
void post_kernel(DevModel m, int64_t n_cand, ChainBufs cb, int64_t trunk_blocks, const double* __restrict__ split_time, const double* __restrict__ params, int phase) {
    extern __shared__ double lds[];
    if ((int64_t)blockIdx.x < trunk_blocks) trunk_body(m, n_cand, params, cb, (int64_t)blockIdx.x, lds, phase);
    else correct_body<CPFIT, GROUP, true>(m, n_cand, cb, split_time, params, (int64_t)blockIdx.x - trunk_blocks, lds, nullptr, nullptr, nullptr, nullptr, 0, 0, phase);
}

// Default fit only: the rates after the split (FitSinglePop, CorrectLambda.py:82-92, called at MigrationInference.py:361-364).
// The weights of the two genomes depend only on the difference of their log-survival at the split (both decay by the
// same rate afterwards), so every interval is an independent bounded 1-D solve: one THREAD per (candidate, interval).
// Kept out of the spectrum kernel: the bounded trust-region code needs ~100 registers of its own, which pushed that
// kernel (128 VGPRs for four waves per SIMD) into scratch.
__global__ __launch_bounds__(256)
void postsplit_kernel(DevModel m, int64_t n_cand, const double* __restrict__ split_time, const double* __restrict__ params, ChainBufs cb) {
    const int rows = m.numT + 1;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_cand * rows) return;
    const int64_t cand = gid / rows;
    const int t = (int)(gid - cand * rows);
    cb.post_lam[gid] = 1.0;
    if (cb.post_word) cb.post_word[gid] = 0;
    Grid G;
    const double* par = params ? params + cand * m.n_param : nullptr;
    const int32_t* bb = cb.bounds ? cb.bounds + cand * 2 * m.n_band : nullptr;
    if (setup_candidate(m, split_time[cand], par, G, bb) != MISTI_OK) return;
    if (t < G.split || t >= G.numT - 1) return;
    const int64_t ch = chain_of(cb, cand);
    const int nfull = (G.ins >= 0) ? G.ins : G.split;
    if (cb.fail_t[ch] < nfull || (G.ins >= 0 && cb.tail_status[cand] != MISTI_OK)) return;     // no value for this candidate
    const double T = G.T(t);
    if (T == 0) return;
    const double* stt = (G.ins >= 0) ? cb.tail_state + 6 * cand : cb.trace + (ch * (int64_t)(m.numT + 1) + nfull) * 6;   // pair state at the split
    const double nc0 = (stt[0] + stt[2]) + stt[4], nc1 = (stt[1] + stt[3]) + stt[5];     // :353-354 (a probability used as a log)
    const double lh0 = G.lhk(t, 0), lh1 = G.lhk(t, 1);
    // FitSinglePop :88-92 with P0 = [[exp(nc0),0,0],[exp(nc1),0,0]] (:361)
    const double pa = exp(nc0), pb = exp(nc1);
    const double w0 = pa / (pa + pb), w1 = pb / (pa + pb);
    const double Te = w0 * ect_one_pop(lh0, T) + w1 * ect_one_pop(lh1, T);
    double x[1] = {w0 * lh0 + w1 * lh1};
    auto resid = [&](const double l[1], double f[1]) { f[0] = ect_one_pop(l[0], T) - Te; };
    const int32_t word = trf_bounded<1>(resid, x, 0.01 * fmin(lh0, lh1));
    cb.post_lam[gid] = x[0];
    if (cb.post_word) cb.post_word[gid] = word;
}

// Kernel 2: post-split rates (:355-376), Smooth (:380-405) and the expected joint spectrum
// (JAFSpectrum, :467-540).  One wavefront per candidate; lane = interval in the prologue,
// lane = state of the 44-state chain afterwards.
// LDS per wave (doubles): xbuf[128] (re | im) | lc[2*(numT0+1)]
// WPB: candidates (waves) per workgroup - 4, or 1 (launch_spectrum: `single_waves`).  A four-wave workgroup needs four free wave slots
// on one compute unit at the same moment; with other contexts' chain kernels resident everywhere those seldom come free together, and
// single-wave workgroups slip into every slot as it opens: measured with 20 batches in flight, the headline grid 2.96 -> 3.59e7 evals/s,
// config2x16 8.2 -> 9.1e7.  Alone on the device the kernel itself runs the same either way.
template <bool CPFIT, int WPB = WAVES_PER_BLOCK>
__global__ __launch_bounds__(WPB * 64, 4)      // 4 waves per SIMD: a 4 096-candidate batch is resident in one round
void spectrum_kernel(DevModel m, int64_t n_cand, const int32_t* __restrict__ order, const double* __restrict__ split_time, const double* __restrict__ params,
                     ChainBufs cb, double* __restrict__ lc_out, double* __restrict__ pr_out,
                     double* __restrict__ jafs_out, int32_t* __restrict__ status_out, double* __restrict__ diag_out,
                     int n_inline, const double* __restrict__ jsfs, const double* __restrict__ consts, double* __restrict__ llk_out, int phase) {
    extern __shared__ double lds[];
    const int lane = lane_id();
    // everything per candidate is wave-uniform: keep it in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t slot = (int64_t)blockIdx.x * WPB + wave;
    if (slot >= n_cand) return;
    if (phase != 0) {                              // two-phase launch (wrong_phase): a candidate of the other phase's chains leaves at once
        const int64_t cd = (int64_t)__builtin_amdgcn_readfirstlane(order[slot]);
        if (wrong_phase(cb, (int64_t)__builtin_amdgcn_readfirstlane((int)chain_of(cb, cd)), phase)) return;
    }
#ifdef MISTI_STAMP2
    long long s_[8];
#define KSTAMP(i) s_[i] = clock64();
#else
#define KSTAMP(i)
#endif
    KSTAMP(0)
    const int64_t cand = (int64_t)__builtin_amdgcn_readfirstlane(order[slot]);
    const int lc_rows = m.numT + 1;
    double* xbuf = lds + (size_t)wave * (128 + 2 * lc_rows);
    double* lcb = xbuf + 128;
    const double* par = params ? params + cand * m.n_param : nullptr;
    Grid G;
    const int32_t* bb = cb.bounds ? cb.bounds + cand * 2 * m.n_band : nullptr;
    int status = setup_candidate(m, split_time[cand], par, G, bb);
    // a batch issued under "no fractional split times" (misti_set_hints; no tail launch was made) that has one after all: refused, never wrong
    if (status == MISTI_OK && cb.integer_splits && G.ins >= 0) status = MISTI_BAD_STRUCTURE;
    Model mod{&m, par, G.split, {0, 0, 0, 0}};
    mod.bb = bb;
    mod.cache();
    // ---- this candidate's share of its chain (+ its own tail interval after a fractional split)
    const int64_t ch = (int64_t)__builtin_amdgcn_readfirstlane((int)chain_of(cb, cand));
    const int nfull = (G.ins >= 0) ? G.ins : G.split;          // intervals taken from the chain
    const double* lc_ch = cb.lc + ch * (int64_t)m.numT * 2;
    const double* tr_ch = cb.trace + ch * (int64_t)(m.numT + 1) * 6;
    int have = 0;                                               // intervals with a corrected rate (for diagnostics)
    if (status == MISTI_OK) {
        const int ft = cb.fail_t[ch];
        if (ft < nfull) { status = cb.fail_status[ch]; have = ft + 1; }
        else {
            have = nfull;
            if (G.ins >= 0) { const int ts = cb.tail_status[cand]; have = nfull + 1; if (ts != MISTI_OK) status = ts; }
        }
    }
    double* lc_o = lc_out ? lc_out + cand * (int64_t)lc_rows * 2 : nullptr;
    double* pr_o = pr_out ? pr_out + cand * (int64_t)(m.numT + 2) * 6 : nullptr;
    int32_t* sv_o = cb.cand_solver ? cb.cand_solver + cand * (int64_t)lc_rows : nullptr;
    if (sv_o) {
        // solver trace of this candidate: its share of the chain (+ the shortened interval of a fractional split);
        // the post-split intervals are added below
        for (int i = lane; i < lc_rows; i += 64) {
            int32_t wd = 0;
            if (i < nfull && i < have) wd = cb.solver[ch * (int64_t)m.numT + i];
            else if (i == nfull && G.ins >= 0 && have > nfull) wd = cb.tail_solver[cand];
            sv_o[i] = wd;
        }
    }
    if (pr_o) {
        // .Pr trace (MigrationInference.py:309,350): rows 0..split from the chain (+ tail); last row = work counters
        const int rows = (status == MISTI_OK) ? G.split + 1 : (have > nfull ? nfull + 1 : (have > 0 ? have : (status == MISTI_OK ? 1 : 0)));
        for (int i = lane; i < (m.numT + 2) * 6; i += 64) {
            const int r = i / 6, j = i - 6 * r;
            double v = 0.0;
            if (r == m.numT + 1) v = cb.work[ch * 6 + j];
            else if (r < rows) v = (G.ins >= 0 && r == nfull + 1) ? cb.tail_state[cand * 6 + j] : tr_ch[6 * r + j];
            pr_o[i] = v;
        }
    }
    if (status != MISTI_OK) {
        if (lane == 0) { status_out[cand] = status; if (diag_out) diag_out[cand] = NAN; }
        if (lane < 7) jafs_out[cand * 7 + lane] = NAN;
        if (lane < n_inline) llk_out[cand * n_inline + lane] = -INFINITY;
        if (lc_o)                            // partial rates (up to the failing interval) for diagnostics
            for (int i = lane; i < 2 * lc_rows; i += 64) {
                const int t = i >> 1;
                lc_o[i] = (t < have && t < nfull) ? lc_ch[i] : (t < have && t == nfull && G.ins >= 0) ? cb.tail_lc[2 * cand + (i & 1)] : 0.0;
            }
        return;
    }
    KSTAMP(1)
    for (int i = lane; i < 2 * lc_rows; i += 64) {
        const int t = i >> 1;
        lcb[i] = (t < nfull) ? lc_ch[i] : (t == nfull && G.ins >= 0) ? cb.tail_lc[2 * cand + (i & 1)] : 0.0;
    }
    lds_fence();
    {
        // diagnostic: largest corrected rate x interval length before smoothing.  From ~5 upwards the
        // correction's residual is nearly flat in that rate and the reference's own value is not
        // determined to 1e-9; callers can tell such candidates apart.
        double mx = 0.0;
        for (int t = lane; t < G.split; t += 64) { double T = G.T(t); mx = fmax(mx, fmax(lcb[2 * t], lcb[2 * t + 1]) * T); }
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
        if (diag_out && lane == 0) diag_out[cand] = (m.flags & MISTI_TRUE_EPS) ? 0.0 : mx;
    }
    // ---- where this candidate leaves the trunk of its chain --------------------------------
    int t0 = 0;
    const bool use_trunk = trunk_active(cb, n_cand);
    int smooth_from = 0;
    if (use_trunk) {
        // first interval whose rates depend on this candidate's split (trunk_leave); the trunk stores records from the
        // smallest such interval of the chain's members on, and its last one where it ended early
        const int t_own = trunk_leave(m, G);
        t0 = min(t_own, cb.trunk_ok[ch]);
        // smoothed rates below t0 are only needed for the optional lc output: skip them otherwise (a run that
        // reaches back below t_own still reads the unsmoothed values there: smoothing writes after all reads)
        if (t0 == t_own && !lc_o) smooth_from = t_own;
    }
    KSTAMP(2)
    {
        // ---- post-split rates (:355-376); nc is a probability used as a log ----
        const double* stt = (G.ins >= 0) ? cb.tail_state + 6 * cand : tr_ch + 6 * nfull;    // pair state at the split
        const double nc0 = (stt[0] + stt[2]) + stt[4], nc1 = (stt[1] + stt[3]) + stt[5];     // :353-354
        const double delta = nc1 - nc0;
        const double ed = exp(delta);
        const int last = G.numT - 1;
        for (int base = G.split; base < G.numT; base += 64) {
            int t = base + lane;
            if (t < last) {
                double T = G.T(t);
                double lam = 1.0;
                int32_t word = 0;
                if (T != 0) {
                    double lh0 = G.lhk(t, 0), lh1 = G.lhk(t, 1);
                    if (CPFIT) {
                        double pnc = (exp(-T * lh0) + exp(delta - T * lh1)) / (1.0 + ed);   // :366
                        lam = -log(pnc) / T;
                        word = solver_word(0, 0, 1);
                    } else {
                        // FitSinglePop (default fit): solved by postsplit_kernel, one thread per interval
                        lam = cb.post_lam[cand * lc_rows + t];
                        word = cb.post_word ? cb.post_word[cand * lc_rows + t] : 0;
                    }
                }
                lcb[2 * t] = lam; lcb[2 * t + 1] = lam;
                if (sv_o) sv_o[t] = word;
            } else if (t == last) {
                double lh0 = G.lhk(t, 0), lh1 = G.lhk(t, 1);
                double lam = (1.0 + ed) / (1.0 / lh0 + ed / lh1);                            // :372-376
                lcb[2 * t] = lam; lcb[2 * t + 1] = lam;
                if (sv_o) sv_o[t] = solver_word(0, 0, 1);
            }
        }
        lds_fence();
        KSTAMP(3)
        // ---- Smooth (:380-405): time-weighted mean of lc over runs of constant lh, t < split
        if (m.flags & MISTI_SMOOTH) smooth_rates(m, G, lcb, lane, smooth_from, G.split);
        for (int i = lane; i < 2 * G.numT; i += 64) { double v = lcb[i]; if (!(v == v)) status = MISTI_NUMERIC; }
        status = __any(status != MISTI_OK) ? MISTI_NUMERIC : MISTI_OK;
        if (lc_o) for (int i = lane; i < 2 * lc_rows; i += 64) lc_o[i] = (i < 2 * G.numT) ? lcb[i] : 0.0;
    }
    double jn = NAN;                      // lane c < 7: normalised expected spectrum, class c
    KSTAMP(4)
    KSTAMP(5) KSTAMP(6) KSTAMP(7)

    if (status == MISTI_OK) {
        // ---- expected spectrum, two-population part (:467-506) ----------------
        TwoPopRow R;
        R.load(lane);
        const bool live = R.live;
        double x = (lane == 2) ? 1.0 : 0.0;
        double w_pre = 0.0, w_post = 0.0;          // occupation integrals before / from the sample date
        if (use_trunk) {
            const double* r = cb.trunk + (ch * (int64_t)m.numT + t0) * TRUNK_REC;
            if (live) { x = r[lane]; w_pre = r[NS2 + lane]; w_post = r[2 * NS2 + lane]; }
        }
        for (int t = t0; t <= G.split && t < G.numT; ++t) {
            if (t == m.sample_date) ancient_project(xbuf, lane, x);
            if (t == G.split) break;
            const int st = twopop_interval(R, m, mod, G, lcb, xbuf, lane, t, x, w_pre, w_post);
            if (st != MISTI_OK) { status = st; break; }
        }
        KSTAMP(5)
        if (status == MISTI_OK) {
            // two-population share of the spectrum: class c = sum over the states of weight_c(state) x occupation integral (w_post for
            // every class, w_pre for the two classes only the ancient sample's pre-date part feeds).  Every state lane forms its
            // seven products; the seven sums over the wave are taken together - halves of the wave swap half of the classes at
            // each of the first three butterfly steps (8 -> 4 -> 2 -> 1 classes per lane), three plain steps finish: 7 shuffles for
            // all classes instead of a 44-step loop on seven lanes (a sixth of this kernel's instructions on a shared grid), and a
            // pairwise sum instead of a running one.  Lanes 8c .. 8c + 7 end up with class c.
            double jp;
            {
                double a4[4], a2[2], z;
                {
                    double a[8];
#pragma unroll
                    for (int c = 0; c < 7; ++c) {
                        const double wg = (double)c_tab.jaf[c][lane];
                        a[c] = live ? (c < 2 ? wg * w_post + wg * w_pre : wg * w_post) : 0.0;
                    }
                    a[7] = 0.0;
                    const bool hi = lane & 32;
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const double keep = hi ? a[c + 4] : a[c], give = hi ? a[c] : a[c + 4]; a4[c] = keep + __shfl_xor(give, 32, 64); }
                }
                {
                    const bool hi = lane & 16;
#pragma unroll
                    for (int c = 0; c < 2; ++c) { const double keep = hi ? a4[c + 2] : a4[c], give = hi ? a4[c] : a4[c + 2]; a2[c] = keep + __shfl_xor(give, 16, 64); }
                }
                {
                    const bool hi = lane & 8;
                    const double keep = hi ? a2[1] : a2[0], give = hi ? a2[0] : a2[1];
                    z = keep + __shfl_xor(give, 8, 64);
                }
                z += __shfl_xor(z, 4, 64); z += __shfl_xor(z, 2, 64); z += __shfl_xor(z, 1, 64);
                jp = __shfl(z, 8 * (lane & 7), 64);                                   // lane c < 7: class c
            }
            // CollapsePops (:518-528)
            xbuf[lane] = live ? x : 0.0; lds_fence();
            double c8 = 0.0;
            if (lane < NS1) for (int i = c_tab.grp_lo[lane]; i < c_tab.grp_hi[lane]; ++i) c8 += xbuf[i];
            lds_fence();
            double P8[NS1];
            for (int i = 0; i < NS1; ++i) P8[i] = bcast(c8, i);
            KSTAMP(6)
            // ---- one-population part: Kingman coalescent in rescaled time --------
            // P(s) = V1 e^-s + V3 e^-3s + V6 e^-6s, s = int lc dt; occupation integral of
            // interval t is (1/lc_t) * int_{S_t}^{S_t+tau_t} P(s) ds (OnePopulation.py:153-178,
            // SolveDifEq :530-540 incl. the last, infinite interval).
            double G1 = 0, G3 = 0, G6 = 0;           // sum_t e^{-a S_t} (1 - e^{-a tau_t}) / (a lc_t)
            double carry = 0.0;
            const int last = G.numT - 1;
            for (int base = G.split; base < G.numT; base += 64) {
                int t = base + lane;
                double lam = 1.0, tau = 0.0;
                if (t < last) { lam = lcb[2 * t]; tau = lam * G.T(t); }
                else if (t == last) { lam = lcb[2 * t]; }
                double inc = tau;                      // inclusive prefix sum over the wave
                for (int o = 1; o < 64; o <<= 1) { double u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
                double S = carry + (inc - tau);
                // e^-3S and e^-6S as powers of e^-S, expm1(-3 tau) and expm1(-6 tau) from a = expm1(-tau) by (1 + a)^3 - 1 = a (3 + a (3 + a))
                // and (1 + b)^2 - 1 = b (2 + b): two transcendentals per interval instead of six, a few ulps each way
                // (one division per interval: 1 / lam, shared by the three sums)
                const double e1 = exp(-S), e3 = e1 * e1 * e1, e6 = e3 * e3;
                const double rl = 1.0 / lam, rl3 = rl * (1.0 / 3.0), rl6 = rl * (1.0 / 6.0);
                if (t < last) {
                    const double a1 = expm1(-tau), a3 = a1 * (3.0 + a1 * (3.0 + a1)), a6 = a3 * (2.0 + a3);
                    G1 += e1 * (-a1) * rl;
                    G3 += e3 * (-a3) * rl3;
                    G6 += e6 * (-a6) * rl6;
                } else if (t == last) {
                    G1 += e1 * rl; G3 += e3 * rl3; G6 += e6 * rl6;
                }
                carry += bcast(inc, 63);
            }
            {
                // the three sums over the wave together (see the class sums above): 7 shuffles instead of 18; lanes 0-15 end up
                // with G1, 16-31 with G3, 32-47 with G6
                const bool h5 = lane & 32, h4 = lane & 16;
                const double k0 = (h5 ? G6 : G1) + __shfl_xor(h5 ? G1 : G6, 32, 64);     // G1 | G6
                const double k1 = (h5 ? 0.0 : G3) + __shfl_xor(h5 ? G3 : 0.0, 32, 64);    // G3 | 0
                double z = (h4 ? k1 : k0) + __shfl_xor(h4 ? k0 : k1, 16, 64);             // lanes 0-15: G1, 16-31: G3, 32-47: G6, 48-63: 0
                z += __shfl_xor(z, 8, 64); z += __shfl_xor(z, 4, 64); z += __shfl_xor(z, 2, 64); z += __shfl_xor(z, 1, 64);
                G1 = bcast(z, 0); G3 = bcast(z, 16); G6 = bcast(z, 32);
            }
            // coefficient vectors of the three exponentials (after the sums: they are not live across the loop)
            double x0 = P8[0];
            const double a3[3] = {1.0, 4.0, 1.0};
            double X[3], V1[NS1], V3[NS1], V6[NS1];
            for (int i = 0; i < 3; ++i) X[i] = P8[1 + i] + a3[i] * x0 / 3.0;
            // b[j][i]: 3-lineage state i -> 2-lineage state 4+j (one-population generator / la)
            const double bm[4][3] = {{2, 1, 0}, {0, 1, 2}, {1, 0, 1}, {0, 1, 0}};
            V6[0] = x0; V3[0] = 0; V1[0] = 0;
            for (int i = 0; i < 3; ++i) { V6[1 + i] = -a3[i] * x0 / 3.0; V3[1 + i] = X[i]; V1[1 + i] = 0; }
            for (int j = 0; j < 4; ++j) {
                double sx = 0, sa = 0;
                for (int i = 0; i < 3; ++i) { sx += bm[j][i] * X[i]; sa += bm[j][i] * a3[i]; }
                V6[4 + j] = sa * x0 / 15.0;
                V3[4 + j] = -sx / 2.0;
                V1[4 + j] = P8[4 + j] + sx / 2.0 - sa * x0 / 15.0;
            }
            // assemble: lane c < 7 holds class c
            double j1 = 0.0;
            const unsigned w1 = lane < 7 ? c_tab.jaf1_bits[lane] : 0u;
            if (lane < 7) for (int i = 0; i < NS1; ++i) j1 += (double)(int)((w1 >> (3 * i)) & 7u) * (V1[i] * G1 + V3[i] * G3 + V6[i] * G6);
            double jc = jp + j1;
            double tot = 0.0;
            for (int c = 0; c < 7; ++c) tot += bcast(jc, c);                         // :583-584
            jn = jc / tot;                                                           // lane c < 7: class c, normalised
            if (__any(lane < 7 && !(jn == jn))) status = MISTI_NUMERIC;
        }
    }
    KSTAMP(7)
    // ---- outputs ------------------------------------------------------------
    if (lane == 0) status_out[cand] = status;
    if (lane < 7) {
        jafs_out[cand * 7 + lane] = (status == MISTI_OK) ? jn : NAN;
#ifdef MISTI_STAMP2
        jafs_out[cand * 7 + lane] = (double)(s_[lane + 1] - s_[lane]);
#endif
    }
    // ---- replicate epilogue for small replicate counts (:600-609): lane r = replicate r ----
    if (n_inline > 0) {
        const int unfolded = (m.flags & MISTI_UNFOLDED) ? 1 : 0;
        // log of class `lane` (folded: classes 0+6, 1+5, 2+4, 3), the same expression as log_class()
        const double mirror = __shfl(jn, lane < 7 ? 6 - lane : lane, 64);
        double lcl = 0.0;
        if (lane < 7 && status == MISTI_OK) lcl = unfolded ? log(jn) : (lane < 3 ? log(jn + mirror) : (lane == 3 ? log(jn) : 0.0));
        double lj[7];
        for (int i = 0; i < 7; ++i) lj[i] = bcast(lcl, i);
        if (lane < n_inline)
            llk_out[cand * n_inline + lane] = (status == MISTI_OK) ? llk_of(jsfs + (int64_t)lane * 8, consts[lane], lj, unfolded) : -INFINITY;
    }
}

// ---- chain discovery: candidates with bitwise identical parameter vectors share a chain ----
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

// Every candidate is inserted into an open-addressing table keyed by the bits of its parameter
// vector (linear probing; the table was cleared during the previous batch, see setup_kernel).  The first candidate in a slot
// owns the chain and draws the chain id; equal hashes are verified against the owner's parameters
// (read-only input), so a collision costs a probe, never correctness.  Chain ids depend on the
// order of arrival - they only name buffers; no result depends on them.
// First launch of a batch, everything that depends on the inputs alone, in ONE kernel (a launch behind a long kernel
// costs its queue a scheduler round trip, which is what limits the rate with many batches in flight):
//   blocks [0, cand_blocks)   chain discovery, one thread per candidate; on the way every thread clears a slice of the
//                             OTHER chain table - the tables are double-buffered, the next batch of this context finds
//                             its table clean without a launch of its own (nothing of the previous batch is running:
//                             same stream);
//   block cand_blocks         dispatch order: candidates sorted by descending split index (counting sort).  Work per
//                             candidate grows with the number of two-population intervals and the dispatcher hands
//                             blocks out in index order, so the longest start first.  The order never affects a result;
//   the blocks after it       llh_const of every replicate.
__global__ __launch_bounds__(256)
void setup_kernel(DevModel m, int64_t n, const double* __restrict__ params, const double* __restrict__ split_time, ChainBufs cb,
                  int cand_blocks, int32_t* __restrict__ order, int64_t n_rep, const double* __restrict__ jsfs, double* __restrict__ consts, int unfolded) {
    const int numT = m.numT, P = m.n_param, NB2 = cb.bounds ? 2 * m.n_band : 0;
    if ((int)blockIdx.x > cand_blocks) {
        const int64_t r = (int64_t)((int)blockIdx.x - cand_blocks - 1) * blockDim.x + threadIdx.x;
        if (r < n_rep) consts[r] = llh_const_of(jsfs + r * 8, unfolded);
        return;
    }
    constexpr int SORT_SUB = 32;       // copies of every bin of the candidate sort (lane mod 32 picks one: fewer conflicts, and - measured - a dispatch order within a split that kernel 2 likes better: 0.39 -> 0.36 ms on config2x16)
    __shared__ int hist[(MISTI_MAX_NUMT + 4) * SORT_SUB];
    if ((int)blockIdx.x == cand_blocks) {
        if (cb.unsorted & 1) return;          // one split for all (the caller says): candidates are dispatched in their own order (below)
        const int nb = numT + 3;
        for (int i = threadIdx.x; i < nb * SORT_SUB; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        const int sub = threadIdx.x & (SORT_SUB - 1);
        auto key = [&](double st) { int k = (st >= 0 && st < (double)(numT + 1)) ? (int)st : 0; return (numT + 1 - k) * SORT_SUB + sub; };
        // One block for the whole batch (a counting sort by descending split index).  Two things bound it, both measured on 65 536
        // candidates (0.095 ms of this kernel's 0.12): the latency of the global load of a trip - hence SORT_ILP loads in flight per
        // thread - and the shared-memory atomics: a sweep is ordered split-major, so the 64 lanes of a wave hit ONE bin, a 64-way
        // conflict per instruction - hence SORT_SUB copies of every bin (a 2-way conflict), summed in the prefix step.
        constexpr int SORT_ILP = 8;
        const int64_t step = SORT_ILP * (int64_t)blockDim.x;
        for (int64_t i0 = threadIdx.x; i0 < n; i0 += step) {
            double st[SORT_ILP];
#pragma unroll
            for (int u = 0; u < SORT_ILP; ++u) { const int64_t i = i0 + u * (int64_t)blockDim.x; st[u] = i < n ? split_time[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < SORT_ILP; ++u) if (i0 + u * (int64_t)blockDim.x < n) atomicAdd(&hist[key(st[u])], 1);
        }
        __syncthreads();
        // exclusive prefix over (bin, copy): within a bin by its thread, over the bins by thread 0
        __shared__ int bin_base[MISTI_MAX_NUMT + 4];
        for (int b = threadIdx.x; b < nb; b += blockDim.x) {
            int acc = 0;
            for (int c = 0; c < SORT_SUB; ++c) { const int v = hist[b * SORT_SUB + c]; hist[b * SORT_SUB + c] = acc; acc += v; }
            bin_base[b] = acc;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int acc = 0; for (int b = 0; b < nb; ++b) { const int c = bin_base[b]; bin_base[b] = acc; acc += c; } }
        __syncthreads();
        for (int b = threadIdx.x; b < nb; b += blockDim.x) for (int c = 0; c < SORT_SUB; ++c) hist[b * SORT_SUB + c] += bin_base[b];
        __syncthreads();
        for (int64_t i0 = threadIdx.x; i0 < n; i0 += step) {
            double st[SORT_ILP];
            int pos[SORT_ILP];
#pragma unroll
            for (int u = 0; u < SORT_ILP; ++u) { const int64_t i = i0 + u * (int64_t)blockDim.x; st[u] = i < n ? split_time[i] : 0.0; }
#pragma unroll
            for (int u = 0; u < SORT_ILP; ++u) pos[u] = i0 + u * (int64_t)blockDim.x < n ? atomicAdd(&hist[key(st[u])], 1) : 0;
#pragma unroll
            for (int u = 0; u < SORT_ILP; ++u) { const int64_t i = i0 + u * (int64_t)blockDim.x; if (i < n) order[pos[u]] = (int32_t)i; }
        }
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    {   // clear the other table for the next batch
        const int64_t tsize = (int64_t)cb.tmask + 1, nthr = (int64_t)cand_blocks * blockDim.x;
        for (int64_t k = i; k < tsize; k += nthr) { cb.z_table[k] = 0; cb.z_slot_len[k] = 0; cb.z_slot_keep[k] = 0; }
        if (i == 0) { cb.z_n_chains[0] = 0; cb.z_n_chains[1] = 0; cb.z_n_chains[2] = 0; cb.z_n_chains[3] = 0; }
    }
    if (i < n) {
        if (cb.unsorted & 1) order[i] = (int32_t)i;
        // the key: the parameter bits and, with per-candidate band bounds, the (start, end) pairs as given (end == -1
        // stays symbolic: members of a chain may differ in their split, never in where a band starts or ends)
        uint64_t h = 0x243f6a8885a308d3ull;
        const double* a = params + i * P;
        const int32_t* ab = cb.bounds ? cb.bounds + i * NB2 : nullptr;
        for (int k = 0; k < P; ++k) h = mix64(h ^ (uint64_t)__double_as_longlong(a[k]));
        if (ab) for (int k = 0; k < NB2; ++k) h = mix64(h ^ (uint64_t)(uint32_t)ab[k]);
        const double st = split_time[i];
        // a candidate without a valid split time (an EMPTY SLOT of a batched search: negative) starts probing at a slot of its own:
        // thousands of them carry one and the same parameter vector, and their inserts would queue on one table entry (measured:
        // the 32 768 empty slots of a Nelder-Mead shrink batch made this kernel 0.72 ms).  Where they end up sharing a chain it is
        // a chain of no intervals.
        if (!(st >= 0)) h = mix64(h ^ (uint64_t)i);
        uint32_t sl = (uint32_t)(h >> 20) & cb.tmask;
        for (;;) {
            const int prev = atomicCAS(&cb.table[sl], 0, (int)i + 1);
            if (prev == 0) {
                const int ch = atomicAdd(cb.n_chains, 1);
                cb.chain_slot[ch] = (int32_t)sl;
                cb.rep[ch] = (int32_t)i;
                cb.slot_chain[sl] = ch;
                if (cb.unsorted & 2) cb.chain_order[ch] = ch;
                break;
            }
            const double* b = params + (int64_t)(prev - 1) * P;
            bool same = true;
            for (int k = 0; k < P; ++k) if (__double_as_longlong(a[k]) != __double_as_longlong(b[k])) same = false;
            if (ab) { const int32_t* bbp = cb.bounds + (int64_t)(prev - 1) * NB2; for (int k = 0; k < NB2; ++k) if (ab[k] != bbp[k]) same = false; }
            if (same) break;
            sl = (sl + 1) & cb.tmask;
        }
        cb.slot_of[i] = (int32_t)sl;
        int need = 0;
        if (st >= 0 && st <= (double)numT) { need = (int)st; if (need > numT - 1) need = numT - 1; }   // full intervals before the (fractional) split
        atomicMax(&cb.slot_len[sl], need);
        // from which interval on this candidate reads its chain's trunk (candidates without a value read nothing)
        Grid G;
        if (setup_candidate(m, st, P ? a : nullptr, G, ab) == MISTI_OK) {
            int t_own = trunk_leave(m, G);
            t_own = t_own < 0 ? 0 : (t_own > numT - 1 ? numT - 1 : t_own);
            atomicMax(&cb.slot_keep[sl], numT - t_own);
        }
    }
    // The last candidate block to finish (a) sorts the chains by descending length - the dispatch order of the chain launch:
    // workgroups pull chains from a queue, the longest first - and (b) tells the host how many chains there are: pinned memory,
    // {chains, candidates, batch tag}; the host uses it for the launch shape of kernel 1 (this batch if it cares to wait a few
    // microseconds, else the next).  Lengths and owners were written by other blocks: read at device scope.
    __shared__ int is_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        is_last = atomicAdd(&cb.n_chains[1], 1) == cand_blocks - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    const int nch = __hip_atomic_load(cb.n_chains, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cb.unsorted & 2) {                                   // one length for all chains (the caller says): the order of arrival will do
        if (cb.hint && threadIdx.x == 0) { cb.hint[0] = nch; cb.hint[1] = (int32_t)n; __threadfence_system(); cb.hint[2] = cb.seq; }
        return;
    }
    auto len_of = [&](int ch) {
        const int sl = __hip_atomic_load(&cb.chain_slot[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int L = __hip_atomic_load(&cb.slot_len[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return numT - (L < 0 ? 0 : (L > numT ? numT : L));                                   // key 0 = the longest
    };
    for (int k = threadIdx.x; k < numT + 2; k += blockDim.x) hist[k] = 0;
    __syncthreads();
    for (int ch = threadIdx.x; ch < nch; ch += blockDim.x) atomicAdd(&hist[len_of(ch)], 1);
    __syncthreads();
    if (threadIdx.x == 0) { int acc = 0; for (int b = 0; b < numT + 2; ++b) { int c = hist[b]; hist[b] = acc; acc += c; } }
    __syncthreads();
    for (int ch = threadIdx.x; ch < nch; ch += blockDim.x) { const int pos = atomicAdd(&hist[len_of(ch)], 1); cb.chain_order[pos] = ch; }
    if (cb.hint && threadIdx.x == 0) {
        cb.hint[0] = nch;
        cb.hint[1] = (int32_t)n;
        __threadfence_system();
        cb.hint[2] = cb.seq;
    }
}

uint32_t chain_table_size(int64_t n_cand) {
    uint32_t t = 64;
    while ((int64_t)t < 2 * n_cand) t <<= 1;
    return t;
}

hipError_t launch_setup(const DevModel& m, int64_t n, const double* params, const double* split, const ChainBufs& cb, int32_t* order,
                        int64_t n_rep, const double* jsfs, double* consts, int unfolded, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const int cand_blocks = (int)((n + 255) / 256);
    const int rep_blocks = (int)((n_rep + 255) / 256);
    hipLaunchKernelGGL(setup_kernel, dim3((unsigned)(cand_blocks + 1 + rep_blocks)), dim3(256), 0, stream, m, n, params, split, cb,
                       cand_blocks, order, n_rep, jsfs, consts, unfolded);
    return hipGetLastError();
}

// ------------------------------------------------------- replicate epilogue --
// llh_const of SetJAFS (MigrationInference.py:217-227): one thread per replicate (misti_llk_dev).
__global__ __launch_bounds__(256) void llh_const_kernel(int64_t n_rep, const double* __restrict__ jsfs, double* __restrict__ consts, int unfolded) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rep) return;
    consts[r] = llh_const_of(jsfs + r * 8, unfolded);
}

// llk[c][r] = const[r] + sum_i data[r][i] log JAFS[c][i]  (folded: pairs 0+6, 1+5, 2+4, 3)
// MigrationInference.py:600-609.  The one HBM-WRITE-bound kernel of the path: 8 bytes out per value, 56 bytes of
// spectrum in per CANDIDATE and 72 bytes of data in per REPLICATE.  So the replicate is what a thread keeps: thread = two adjacent
// replicates (their class counts and constants in registers, read once), block = 512 replicates x a CHUNK of candidates whose
// class logs the block computes once into LDS; per value that leaves 4 (folded) or 7 fused multiply-adds and one half of a
// 16-byte store, 1 KB contiguous per wave-instruction, streamed past the caches (nontemporal: nothing reads llk in this kernel).
// Round 4's kernel gave every value its own 64-byte read of the replicate row - 4.7 GB through L2 for 0.5 GB written.
// Same expressions as llk_of / log_class (the inline epilogue of spectrum_kernel): the same bits.
constexpr int LLK_CHUNK_MAX = 64;      // candidates per block at most (class logs in LDS: 64 x 7 doubles)
template <bool UNFOLDED>
__global__ __launch_bounds__(256)
void llk_kernel(int64_t n_cand, int chunk, const double* __restrict__ jafs, const int32_t* __restrict__ status,
                int64_t n_rep, const double* __restrict__ jsfs, const double* __restrict__ consts,
                double* __restrict__ llk) {
    constexpr int NF = UNFOLDED ? 7 : 4;
    __shared__ double lj[LLK_CHUNK_MAX][8];           // [7] = 1.0 where the candidate has no value (status != OK), else 0.0
    const int64_t c0 = (int64_t)blockIdx.y * chunk;
    const int nc = (int)(n_cand - c0 < chunk ? n_cand - c0 : chunk);
    for (int i = threadIdx.x; i < nc * 8; i += blockDim.x) {
        const int c = i >> 3, k = i & 7;
        double v;
        if (k < 7) v = log_class(jafs + (c0 + c) * 7, k, UNFOLDED ? 1 : 0);
        else v = (status && status[c0 + c] != MISTI_OK) ? 1.0 : 0.0;
        lj[c][k] = v;
    }
    // this thread's two replicates: class counts (folded: the four sums) and constants
    const int64_t r0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    double f[2][NF], cst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t r = r0 + u < n_rep ? r0 + u : (n_rep - 1);         // a lane beyond the table repeats the last row (never stored)
        const double* d = jsfs + r * 8 + 1;
        cst[u] = consts[r];
        if (UNFOLDED) { for (int i = 0; i < 7; ++i) f[u][i] = d[i]; }
        else { f[u][0] = d[0] + d[6]; f[u][1] = d[1] + d[5]; f[u][2] = d[2] + d[4]; f[u][3] = d[3]; }
    }
    __syncthreads();
    if (r0 >= n_rep) return;
    const bool pair = r0 + 1 < n_rep && (n_rep & 1) == 0;                  // both in range and every row 16-byte aligned
    double* out = llk + c0 * n_rep + r0;
    for (int c = 0; c < nc; ++c, out += n_rep) {
        double a0 = cst[0], a1 = cst[1];
#pragma unroll
        for (int i = 0; i < NF; ++i) { const double l = lj[c][i]; a0 = fma(f[0][i], l, a0); a1 = fma(f[1][i], l, a1); }
        if (lj[c][7] != 0.0) { a0 = -INFINITY; a1 = -INFINITY; }
        if (pair) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 v; v.x = a0; v.y = a1;
            __builtin_nontemporal_store(v, (d2*)out);
        } else {
            __builtin_nontemporal_store(a0, out);
            if (r0 + 1 < n_rep) __builtin_nontemporal_store(a1, out + 1);
        }
    }
}

// Bootstrap reduction (test.bs/bs_conf_int.ipynb: per replicate the split value with the largest
// likelihood): thread = replicate, coalesced reads along the replicate axis; -inf / NaN are skipped,
// ties go to the lowest candidate index (numpy.argmax), -1 when no candidate has a value.
__global__ __launch_bounds__(256)
void argmax_kernel(int64_t n_cand, int64_t n_rep, const double* __restrict__ llk, int32_t* __restrict__ best, double* __restrict__ best_llk) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rep) return;
    double bv = -INFINITY;
    int32_t bi = -1;
    for (int64_t c = 0; c < n_cand; ++c) {
        const double v = llk[c * n_rep + r];
        if (v > bv) { bv = v; bi = (int32_t)c; }           // false for NaN and for -inf
    }
    best[r] = bi;
    if (best_llk) best_llk[r] = bv;
}

hipError_t launch_argmax(int64_t n_cand, int64_t n_rep, const double* llk, int32_t* best, double* best_llk, hipStream_t stream) {
    if (n_rep <= 0) return hipSuccess;
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((n_rep + 255) / 256)), dim3(256), 0, stream, n_cand, n_rep, llk, best, best_llk);
    return hipGetLastError();
}

// ----------------------------------------------------------- launchers -------
hipError_t upload_tables(const DevTables& t) {
    static double inv[INV_TABLE];
    inv[0] = 0.0;
    for (int k = 1; k < INV_TABLE; ++k) inv[k] = 1.0 / (double)k;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_inv), inv, sizeof inv);
    if (e != hipSuccess) return e;
    {
        // c_qmax[K]: the root in (0, K - 1) of e^-q q^(K-1) / (K-1)! = 1e-19 (the Poisson weight grows with q below its mode), by bisection
        static double qmax[QMAX_TABLE];
        qmax[0] = -1.0; qmax[1] = -1.0;
        const long double tol = logl(1e-19L);
        for (int K = 2; K < QMAX_TABLE; ++K) {
            long double lo = 0.0L, hi = (long double)(K - 1);
            for (int it = 0; it < 200; ++it) {
                const long double mid = 0.5L * (lo + hi);
                const long double lw = -mid + (long double)(K - 1) * logl(mid) - lgammal((long double)K);
                if (lw < tol) lo = mid; else hi = mid;
            }
            qmax[K] = (double)lo;
        }
        qmax[QMAX_TABLE - 1] = 1e300;           // never reached: Q_SWITCH bounds q
        e = hipMemcpyToSymbol(HIP_SYMBOL(c_qmax), qmax, sizeof qmax);
        if (e != hipSuccess) return e;
    }
    return hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &t, sizeof(DevTables));
}

size_t spectrum_lds_bytes(int numT, int wpb) { return (size_t)wpb * (128 + 2 * (numT + 1)) * sizeof(double); }

// Work items per wavefront.  Kernel 1 holds two wavefronts per SIMD (208 VGPRs), 2 048 on the chip: as few
// items per wave as keep the launch within one resident round - 1, 2, 4 - and beyond that ten (six lanes per
// item, none spare).  (Measured, 16 384 chains: 10 per wave 4.5 ms, 8 per wave 4.5 ms, 4 per wave 4.9 ms,
// 2 per wave 5.4 ms; packing also executes fewer instructions in total, which is what counts when batches
// overlap: 6.8 against 6.2 and 4.8 million evaluations/s for 10, 8 and 4 per wave.)  A chain's bits do not
// depend on the packing (tests/test_gpu_trunk.py).
int correct_cands_per_wave(int64_t n_items, const Tuning& tn) {
    const int forced = tn.cands_per_wave;
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 10) return forced;
    int cpw = 1;
    while (cpw < 8 && n_items / cpw > 2048) cpw *= 2;
    return cpw == 8 ? 10 : cpw;
}

Tuning read_tuning() {
    Tuning t;
    auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
    auto flag = [](const char* name) { const char* e = getenv(name); return e && e[0] && e[0] != '0'; };
    t.chains_per_wave = num("MISTI_CHAINS_PER_WAVE");
    t.cands_per_wave = num("MISTI_CANDS_PER_WAVE");
    t.no_follow = flag("MISTI_NO_FOLLOW");
    t.no_trunk = flag("MISTI_NO_TRUNK");
    t.follow_max = num("MISTI_FOLLOW_MAX_CHAINS");
    t.min_blocks = num("MISTI_FOLLOW_MIN_BLOCKS");
    t.busy_contexts = getenv("MISTI_FOLLOW_BUSY_CONTEXTS") ? num("MISTI_FOLLOW_BUSY_CONTEXTS") : -1;
    t.yield_nfev = getenv("MISTI_YIELD_NFEV") ? num("MISTI_YIELD_NFEV") : -1;
    t.k2_single_waves = getenv("MISTI_K2_SINGLE_WAVES") ? num("MISTI_K2_SINGLE_WAVES") : -1;
    t.pairing = !(getenv("MISTI_FOLLOW_PAIRING") && getenv("MISTI_FOLLOW_PAIRING")[0] == '0');
    t.two_phase = !(getenv("MISTI_TWO_PHASE") && getenv("MISTI_TWO_PHASE")[0] == '0');
    return t;
}

static size_t correct_lds_bytes(int numT) { return (size_t)(3 * numT - 1) * sizeof(double); }
static size_t trunk_lds_bytes(int numT) { return (128 + 2 * (size_t)(numT + 1)) * sizeof(double); }

template <bool CPFIT, int GROUP>
static void launch_chains_t(const DevModel& m, int64_t n_items, const ChainBufs& cb, const double* split, const double* params, int yield_nfev,
                            hipStream_t stream, hipEvent_t after_packed, hipError_t& event_error) {
    const int per_wave = 64 / GROUP;
    dim3 grid((unsigned)((n_items + per_wave - 1) / per_wave));
    hipLaunchKernelGGL((correct_kernel<CPFIT, GROUP>), grid, dim3(64), correct_lds_bytes(m.numT), stream, m, n_items, cb, split, params, yield_nfev);
    // phase 1 of what follows the chains starts here, beside the resume launch.  A record that fails must fail the batch: the side
    // stream's wait on a never-recorded (or stale) event is a no-op, and phase 1 would race this launch.
    if (after_packed) event_error = hipEventRecord(after_packed, stream);
    if (yield_nfev > 0 && GROUP != 64) {
        // the chains that yielded, one per wave: as many workgroups as are resident (two waves per SIMD for --cpfit, one for the
        // default fit); those beyond the list's length leave at once
        const int64_t resident = CPFIT ? 2048 : 1024;
        const int64_t blocks = n_items < resident ? n_items : resident;
        const size_t lds = (9 * (size_t)m.numT) * sizeof(double);
        hipLaunchKernelGGL((correct_resume_kernel<CPFIT>), dim3((unsigned)blocks), dim3(64), lds, stream, m, n_items, cb, split, params);
    }
}

// One chain per wavefront and a trunk to build: the trunk follows its chain inside the chain launch
// (correct_follow_kernel) instead of running after it.  MISTI_NO_FOLLOW=1 keeps it in the post launch.
bool trunk_follows(int cpw_chains, int64_t trunk_cap, const Tuning& tn) {
    return !tn.no_follow && trunk_cap > 0 && cpw_chains == 1;
}

template <bool CPFIT, int GROUP>
static void launch_post_t(const DevModel& m, int64_t n_cand, const ChainBufs& cb, const double* split, const double* params, bool follow, hipStream_t stream, int phase) {
    const int per_wave = 64 / GROUP;
    const int64_t trunk_blocks = follow ? 0 : cb.trunk_cap;
    dim3 grid((unsigned)(trunk_blocks + (n_cand + per_wave - 1) / per_wave));
    size_t lds = correct_lds_bytes(m.numT);
    if (trunk_lds_bytes(m.numT) > lds) lds = trunk_lds_bytes(m.numT);
    hipLaunchKernelGGL((post_kernel<CPFIT, GROUP>), grid, dim3(64), lds, stream, m, n_cand, cb, trunk_blocks, split, params, phase);
}

#define MISTI_DISPATCH_GROUP(FN, ...)                                                                   \
    switch (cpw) {                                                                                       \
        case 10: cp ? FN<true, 6>(__VA_ARGS__) : FN<false, 6>(__VA_ARGS__); break;                       \
        case 8: cp ? FN<true, 8>(__VA_ARGS__) : FN<false, 8>(__VA_ARGS__); break;                        \
        case 4: cp ? FN<true, 16>(__VA_ARGS__) : FN<false, 16>(__VA_ARGS__); break;                      \
        case 2: cp ? FN<true, 32>(__VA_ARGS__) : FN<false, 32>(__VA_ARGS__); break;                      \
        default: cp ? FN<true, 64>(__VA_ARGS__) : FN<false, 64>(__VA_ARGS__); break;                     \
    }

// the chains (the number of live chains is read on the device: slots beyond it exit at once)
// cpw: chains per wavefront, chosen by the caller from the expected number of chains
// est_chains: chains of the previous batch of this size on the context, or < 0 when unknown
hipError_t launch_correct(const DevModel& m, int64_t n_cand, const ChainBufs& cb, const double* split, const double* params,
                          int cpw, bool follow, int64_t est_chains, const Tuning& tn, int yield_nfev, hipStream_t stream, hipEvent_t after_packed) {
    if (n_cand <= 0) return hipSuccess;
    const bool cp = m.flags & MISTI_CPFIT;
    if (follow) {
        const size_t lds = (3 * (size_t)m.numT + 128 + 2 * (size_t)(m.numT + 1) + 2 * (size_t)m.numT + 4 + 6 * (size_t)m.numT) * sizeof(double);
        // one workgroup per chain expected, at most the resident 1 024 (two waves per SIMD; 512 for the default fit at one): the
        // workgroups pull chains from a queue, so any grid is correct and a stale hint costs at most idle or missing workgroups
        int64_t blocks = (est_chains > 0 && est_chains < n_cand) ? est_chains : n_cand;
        const int64_t resident = cp ? 1024 : 512;
        if (blocks > resident) blocks = resident;
        // the hint may be stale (same batch size, other parameters): never fewer than that many workgroups for a larger batch
        // (idle workgroups find the queue empty and leave at once)
        const int64_t min_blocks = tn.min_blocks > 0 ? tn.min_blocks : FOLLOW_MIN_BLOCKS;
        const int64_t floor_blocks = n_cand < min_blocks ? n_cand : min_blocks;
        if (blocks < floor_blocks) blocks = floor_blocks;
        if (cp) hipLaunchKernelGGL(correct_follow_kernel<true>, dim3((unsigned)blocks), dim3(128), lds, stream, m, n_cand, cb, split, params);
        else hipLaunchKernelGGL(correct_follow_kernel<false>, dim3((unsigned)blocks), dim3(128), lds, stream, m, n_cand, cb, split, params);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && after_packed) e = hipEventRecord(after_packed, stream);      // never left unrecorded for a caller that waits on it
        return e;
    }
    hipError_t event_error = hipSuccess;
    MISTI_DISPATCH_GROUP(launch_chains_t, m, n_cand, cb, split, params, yield_nfev, stream, after_packed, event_error)
    const hipError_t e = hipGetLastError();
    return e != hipSuccess ? e : event_error;
}

// chains the trunk buffer must hold for a batch of n_cand (the trunk runs only when
// n_chains * TRUNK_MIN_SHARE <= n_cand); MISTI_NO_TRUNK=1 in the environment disables it
int64_t trunk_capacity(int64_t n_cand, const Tuning& tn) {
    if (tn.no_trunk) return 0;
    const int64_t cap = n_cand / TRUNK_MIN_SHARE;
    return cap < TRUNK_MAX_CHAINS ? cap : TRUNK_MAX_CHAINS;   // beyond that many chains candidates walk their own intervals
}

// trunks + tails in one launch, then the candidates (with the replicate epilogue when n_rep is small)
hipError_t launch_spectrum(const DevModel& m, int64_t n_cand, const int32_t* order, const double* split, const double* params,
                           const ChainBufs& cb, double* lc_out, double* pr_out, double* jafs, int32_t* status, double* diag,
                           int64_t n_rep, const double* jsfs, const double* consts, double* llk, bool follow, bool skip_post, bool single_waves,
                           const Tuning& tn, hipStream_t stream, int phase) {
    if (n_cand <= 0) return hipSuccess;
    const bool cp = m.flags & MISTI_CPFIT;
    const int cpw = correct_cands_per_wave(n_cand, tn);          // tails: one item per candidate
    // skip_post: the caller knows there is nothing for the post launch to do (no trunk left for it and no fractional split: run_dev)
    if (!skip_post) { MISTI_DISPATCH_GROUP(launch_post_t, m, n_cand, cb, split, params, follow, stream, phase) }
    if (!cp) {
        const int64_t threads = n_cand * (int64_t)(m.numT + 1);
        hipLaunchKernelGGL(postsplit_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, m, n_cand, split, params, cb);
    }
    const int n_inline = (n_rep > 0 && n_rep <= LLK_INLINE_MAX) ? (int)n_rep : 0;
    const int wpb = single_waves ? 1 : WAVES_PER_BLOCK;
    dim3 grid((unsigned)((n_cand + wpb - 1) / wpb));
#define MISTI_SPECTRUM_LAUNCH(CP, W)                                                                                                   \
    hipLaunchKernelGGL((spectrum_kernel<CP, W>), grid, dim3(W * 64), spectrum_lds_bytes(m.numT, W), stream,                            \
                       m, n_cand, order, split, params, cb, lc_out, pr_out, jafs, status, diag, n_inline, jsfs, consts, llk, phase)
    if (cp) { if (single_waves) MISTI_SPECTRUM_LAUNCH(true, 1); else MISTI_SPECTRUM_LAUNCH(true, WAVES_PER_BLOCK); }
    else { if (single_waves) MISTI_SPECTRUM_LAUNCH(false, 1); else MISTI_SPECTRUM_LAUNCH(false, WAVES_PER_BLOCK); }
#undef MISTI_SPECTRUM_LAUNCH
    return hipGetLastError();
}

hipError_t launch_forward(const DevModel& m, int64_t n_cand, const double* split, const double* params, int hold_mu, double* lh_out, double* pr_out,
                          int32_t* status, hipStream_t stream) {
    if (n_cand <= 0) return hipSuccess;
    hipLaunchKernelGGL(forward_kernel, dim3((unsigned)((n_cand + 63) / 64)), dim3(64), 0, stream, m, n_cand, split, params, hold_mu, lh_out, pr_out, status);
    return hipGetLastError();
}

hipError_t launch_llh_const(int64_t n_rep, const double* jsfs, double* consts, int unfolded, hipStream_t stream) {
    if (n_rep <= 0) return hipSuccess;
    hipLaunchKernelGGL(llh_const_kernel, dim3((unsigned)((n_rep + 255) / 256)), dim3(256), 0, stream, n_rep, jsfs, consts, unfolded);
    return hipGetLastError();
}

hipError_t launch_llk(int64_t n_cand, const double* jafs, const int32_t* status, int64_t n_rep, const double* jsfs,
                      const double* consts, double* llk, int unfolded, hipStream_t stream) {
    if (n_cand <= 0 || n_rep <= 0) return hipSuccess;
    // block = 512 replicates x `chunk` candidates: as large a chunk as still leaves the chip a few thousand blocks (a block's
    // set-up - the class logs of its chunk, the rows of its replicates - is paid once per chunk)
    const int64_t tiles = (n_rep + 511) / 512;
    int64_t chunk = n_cand * tiles / 4096;
    chunk = chunk < 4 ? 4 : (chunk > LLK_CHUNK_MAX ? LLK_CHUNK_MAX : chunk);
    const int64_t per_launch = 65535 * chunk;                               // gridDim.y limit
    for (int64_t c0 = 0; c0 < n_cand; c0 += per_launch) {
        const int64_t nc = n_cand - c0 < per_launch ? n_cand - c0 : per_launch;
        const dim3 grid((unsigned)tiles, (unsigned)((nc + chunk - 1) / chunk));
        if (unfolded) hipLaunchKernelGGL(llk_kernel<true>, grid, dim3(256), 0, stream, nc, (int)chunk, jafs + c0 * 7, status ? status + c0 : nullptr, n_rep, jsfs, consts, llk + c0 * n_rep);
        else hipLaunchKernelGGL(llk_kernel<false>, grid, dim3(256), 0, stream, nc, (int)chunk, jafs + c0 * 7, status ? status + c0 : nullptr, n_rep, jsfs, consts, llk + c0 * n_rep);
    }
    return hipGetLastError();
}

}  // namespace misti

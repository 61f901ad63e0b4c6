// Shared between the kernels (misti_kernels.hip) and the C-ABI host layer (misti_api.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/misti_hip.h"

#include "misti_consts.h"

namespace misti {

// Per-row view of the constant chain structure, in __constant__ memory.
struct DevTables {
    int src[MAXNZ][64];      // source state of the n-th off-diagonal entry of row `lane`
    int kind[MAXNZ][64];     // 0 la0, 1 la1, 2 mu0, 3 mu1
    int mult[MAXNZ][64];     // integer multiplicity
    int dcnt[4][64];         // exit-rate multiplicities of the state (diagonal = -sum dcnt*rate)
    int jaf[7][64];          // class-major StateToJAF
    int jaf1[7][NS1];
    unsigned long long jaf_bits[7][3];   // the same weights (0..7) as bit planes over the 44 states: one 24-byte load per class
    unsigned int jaf1_bits[7];           // one-population weights, 3 bits per state
    int grp_lo[NS1], grp_hi[NS1];
    int anc_n[2], anc_dst[2], anc_src[2][8];
    int pulse_n[2][64];
    int pulse_src[2][64][MAXPULSE];
    int pulse_ab[2][64][MAXPULSE];
    double tal_zr[TALBOT_HALF], tal_zi[TALBOT_HALF];   // Talbot nodes z_k (upper half plane)
    double tal_cr[TALBOT_HALF], tal_ci[TALBOT_HALF];   // weights c_k = (i/N) exp(z_k) z'(theta_k)
};

// Kernel-argument copy of the model (device pointers).
struct DevModel {
    int numT;
    int sample_date;
    unsigned flags;
    int n_band, n_pulse, n_param;
    double mixture_th;
    const double* times;     // [numT-1]
    const double* lh;        // [numT][2]
    const int* run_start;    // [2][numT]  smoothing runs of constant lh (MigrationInference.py:387-405)
    const int* run_end;      // [2][numT]
    const int* leave_ok;     // [numT]     1 where SOME split (integer or fractional) leaves a chain's trunk (trunk_leave): the only intervals
                             //            whose trunk record a candidate can ever read
    misti_band_t bands[MISTI_MAX_BANDS];
    misti_pulse_t pulses[MISTI_MAX_PULSES];
};

// Device buffers of the chain machinery (see correct_kernel).  Candidates with bitwise identical
// parameter vectors share a chain; chains are found by inserting every candidate into an open-addressing
// table keyed by its parameters (discover_kernel): the first one in a slot owns the chain.
struct ChainBufs {
    int32_t* n_chains;      // [4]: chains; candidate blocks of setup_kernel that have finished; head of the chain queue; spare
    int32_t* z_n_chains;    // the OTHER set of {n_chains, table, slot_len, slot_keep}: cleared by this batch for the next one
    int32_t* z_table;
    int32_t* z_slot_len;
    int32_t* z_slot_keep;
    int32_t* table;         // [tsize] slot -> owner candidate + 1, 0 = empty
    int32_t* slot_chain;    // [tsize] slot -> chain
    int32_t* slot_len;      // [tsize] slot -> number of full intervals needed (max over members)
    int32_t* slot_keep;     // [tsize] slot -> numT - (first interval at which a member leaves the trunk), max over members: the trunk
                            //         stores its records from interval numT - slot_keep on (0: no member reads any)
    int32_t* resume_t;      // [n] per chain that yielded in a packed launch: the interval it resumes at (correct_resume_kernel)
    int32_t* resume_list;   // [n] those chains; length n_chains[3], head n_chains[2]
    int32_t* chain_order;   // [n] chains by descending length (the last candidate block of setup_kernel sorts them): dispatch order
    int32_t* slot_of;       // [n] candidate -> slot
    int32_t* of;            // [n] candidate -> chain, resolved by the idle blocks of the chain launch
    int32_t* chain_slot;    // [n] chain -> slot
    int32_t* rep;           // [n] chain -> a member candidate (its parameters)
    uint32_t tmask;         // tsize - 1 (tsize a power of two >= 2 n)
    double* lc;             // [n][numT][2]     per chain: corrected rates, unsmoothed
    double* trace;          // [n][numT+1][6]   per chain: pair state before interval t (.Pr layout)
    int32_t* fail_t;        // [n] per chain: first failing interval, INT_MAX if none
    int32_t* fail_status;   // [n]
    double* work;           // [n][6] per chain: work counters
    double* tail_lc;        // [n][2] per candidate with a fractional split
    double* tail_state;     // [n][6]
    int32_t* tail_status;   // [n]
    double* trunk;          // [trunk_cap][numT][TRUNK_REC] per chain: 44-state vector + occupation integrals before interval t
    int32_t* trunk_ok;      // [trunk_cap] per chain: last valid record
    int64_t trunk_cap;      // chains the trunk buffer holds (0: no trunk)
    int32_t* simd_load;     // [PAIR_TABLE] chain waves resident per (compute unit, SIMD) - all contexts of the device share it - or NULL (correct_follow_kernel)
    int32_t* hint;          // host-pinned [3] or NULL: {chains, candidates, batch tag}, written by the last block of discover_kernel
    int32_t seq;            // this batch's tag
    int32_t unsorted;       // the caller knows its batch has one split time: bit 0 - candidates are dispatched in their own order, bit 1 - chains in
                            // order of arrival (setup_kernel sorts neither)
    int32_t integer_splits; // the caller (or the host-buffer entry point, which sees them) says no split time has a fractional part: no tail launch was
                            // made, and a candidate that has one after all is refused (spectrum_kernel)
    const int32_t* bounds;  // [n][n_band][2] per-candidate (start, end) of every band, or NULL: the model's
    double* post_lam;       // [n][numT+1] default fit: rates after the split (postsplit_kernel -> spectrum kernel)
    int32_t* post_word;     // [n][numT+1] their solver words, or NULL (trace off)
    // solver trace (misti_enable_solver_trace), all NULL when off
    int32_t* solver;        // [n][numT] per chain: packed word of interval t (nfev | status << 16 | kind << 20)
    int32_t* tail_solver;   // [n] per candidate with a fractional split: the shortened interval
    int32_t* cand_solver;   // [n][numT+1] per candidate, assembled by the spectrum kernel (+ post-split solves)
    double* iters;          // [iter_cap][numT][MISTI_TRACE_MAX_ITER][2] trial points of the unbounded solves, or NULL
    int64_t iter_cap;       // chains the iterate buffer holds
};

__host__ __device__ __forceinline__ int32_t solver_word(int nfev, int status, int kind) {
    return (int32_t)((nfev & 0xffff) | ((status & 15) << 16) | ((kind & 15) << 20));
}

__device__ __forceinline__ int64_t chain_of(const ChainBufs& cb, int64_t cand) { return cb.of[cand]; }
__device__ __forceinline__ int chain_len(const ChainBufs& cb, int64_t ch) { return cb.slot_len[cb.chain_slot[ch]]; }

// Batched Nelder-Mead (misti_nm.hip): everything a start owns, in HBM.  V = N + 1 vertices.
enum { NM_NONE = 0, NM_REFLECT = 1, NM_EXPAND = 2, NM_CONTRACT = 3, NM_INSIDE = 4,
       NM_CUT = 5 };        // the evaluation budget (maxfev) ran out before the iteration's second point: SciPy abandons the iteration there
struct NmState {
    int64_t S;              // starts
    int N;                  // parameters
    int maxiter;
    int64_t maxfun;
    double xatol, fatol;
    double split;           // the split time every point is evaluated at
    // per start
    double* sim;            // [S][V][N] simplices, best vertex first after every sort
    double* fsim;           // [S][V]    objective (-llk, +inf where the engine has no value)
    double* scratch;        // [S][V][N] row permutation buffer of the sort
    double* fxr;            // [S]
    int32_t* nit;           // [S] SciPy's `iterations`
    int32_t* nfev;          // [S] SciPy's fcalls
    int32_t* done;          // [S] -1 while running; 0 converged, 1 evaluation budget, 2 iteration budget
    int32_t* kind;          // [S] NM_* of the iteration in progress
    int32_t* shrunk;        // [S] 0: no shrink in the iteration in progress; 1 + n: a shrink of which n vertices were evaluated (n < N: the budget
                            //     ran out inside it - vertex n + 1 is moved but keeps its old value, the rest is untouched, as in SciPy)
    // per slot of the iteration's batches (live starts compacted)
    double* p1;             // [S][N]    reflection points
    double* p2;             // [S][N]    expansion / contraction points
    double* p3;             // [S][N][N] shrunk vertices
    double* split0;         // [S * V]   split time per engine candidate of the initial batch
    double* split1;         // [S]       ... of the reflection batch (negative: no point in this slot)
    double* split2;         // [S]
    double* split3;         // [S * N]
    // speculative iterations (few live starts: latency-bound): every point SciPy COULD ask for in the iteration, one batch
    double* ps;             // [spec_cap][4 + N][N]  reflection, expansion, outside / inside contraction, the N shrunk vertices
    double* ps_split;       // [spec_cap * (4 + N)]
    int64_t spec_cap;       // live starts up to which an iteration is speculative
    const int32_t* idx_cur; // [S] slot -> start of the iteration in progress
    const int32_t* count_cur;   // [1] its number of live starts
    int32_t* idx_next;      // [S] slot -> start of the next iteration (filled by the kernel that ends this one)
    int32_t* count_next;    // [1]
};
hipError_t launch_nm_init(const NmState& st, const double* starts, hipStream_t stream);
hipError_t launch_nm_begin(const NmState& st, const double* llk0, hipStream_t stream);
hipError_t launch_nm_reflect(const NmState& st, int64_t bound, const double* llk1, hipStream_t stream);
hipError_t launch_nm_accept(const NmState& st, int64_t bound, const double* llk2, hipStream_t stream);
hipError_t launch_nm_finish(const NmState& st, int64_t bound, const double* llk3, hipStream_t stream);
hipError_t launch_nm_result(const NmState& st, double* x, double* llh, int32_t* status, hipStream_t stream);
hipError_t launch_nm_spec_points(const NmState& st, int64_t bound, hipStream_t stream);
hipError_t launch_nm_spec_step(const NmState& st, const NmState& nx, int64_t bound, const double* llk, int32_t* live_host, hipStream_t stream);
hipError_t launch_nm_spec_finish(const NmState& st, int64_t bound, const double* llk, hipStream_t stream);

// Batched basin hopping (scipy.optimize.basinhopping with Nelder-Mead as the local minimiser; reference semantics
// MigrationInference.Solve(globalOpt=True), /root/reference/MigrationInference.py:723-725): everything a start owns besides its simplex.
struct BhState {
    int64_t S;
    int N;
    double beta;            // 1 / T (inf for T == 0), Metropolis
    double target, factor;  // AdaptiveStepsize: target acceptance rate, stepwise factor
    int interval;
    double* x_cur;          // [S][N] incumbent
    double* f_cur;          // [S]
    int32_t* ok_cur;        // [S]    success of the incumbent's minimisation
    double* x_best;         // [S][N] Storage: lowest successful result
    double* f_best;         // [S]
    int32_t* ok_best;       // [S]
    double* stepsize;       // [S]
    int32_t* nstep;         // [S]    AdaptiveStepsize.nstep (reset at every adjustment)
    int32_t* naccept;       // [S]
    int32_t* nfev;          // [S]    res.nfev: evaluations of all minimisations
    int32_t* failures;      // [S]    res.minimization_failures
    int32_t* accepted;      // [S]    hops accepted (not a SciPy field; diagnostics)
};
// after a minimisation from `trial` (results in x_min / llh_min / nm.nfev / nm status): initial one (hop < 0) or hop `hop` with its Metropolis draw
hipError_t launch_bh_update(const BhState& bh, const NmState& nm, const double* x_min, const double* llh_min, const int32_t* nm_status,
                            int hop, int niter, const double* uniforms, hipStream_t stream);
// next trial points: AdaptiveStepsize.take_step + RandomDisplacement with the pre-drawn uniforms of hop `hop`
hipError_t launch_bh_step(const BhState& bh, int hop, int niter, const double* uniforms, double* trial, hipStream_t stream);
hipError_t launch_bh_result(const BhState& bh, double* x, double* llh, hipStream_t stream);

// Replicate epilogue fused into the spectrum kernel up to this many replicates (one launch less per batch).
constexpr int LLK_INLINE_MAX = 8;

hipError_t upload_tables(const DevTables& t);
size_t spectrum_lds_bytes(int numT, int wpb);
// Diagnostic overrides of the launch shape, read from the environment ONCE per context (misti_create) - never on the batch path.
struct Tuning {
    int chains_per_wave = 0;   // MISTI_CHAINS_PER_WAVE: 1 | 2 | 4 | 8 | 10 forces the packing of the chain launch
    int cands_per_wave = 0;    // MISTI_CANDS_PER_WAVE: ... of the packed kernels (chains and tails)
    bool no_follow = false;    // MISTI_NO_FOLLOW=1: trunks in the launch after the chains
    bool no_trunk = false;     // MISTI_NO_TRUNK=1: every candidate walks all its intervals
    int follow_max = 0;        // MISTI_FOLLOW_MAX_CHAINS: chains up to which a batch runs one chain per wave (0: FOLLOW_MAX_CHAINS)
    int min_blocks = 0;        // MISTI_FOLLOW_MIN_BLOCKS: least workgroups of that launch (0: FOLLOW_MIN_BLOCKS)
    int yield_nfev = -1;       // MISTI_YIELD_NFEV: evaluations after which a solve of a PACKED launch yields its chain to correct_resume_kernel
                               // (-1: YIELD_NFEV; 0: never)
    bool two_phase = true;     // MISTI_TWO_PHASE=0: everything behind a yielding packed launch on one stream (no phase-1 launch beside the resume launch)
    bool pairing = true;       // MISTI_FOLLOW_PAIRING=0: the chain always on the first wave of its workgroup (default: placement-aware, correct_follow_kernel)
    int k2_single_waves = -1;  // MISTI_K2_SINGLE_WAVES: 1 / 0 forces kernel 2's workgroups to one / four waves (-1: chosen per batch, run_dev)
    int busy_contexts = -1;    // MISTI_FOLLOW_BUSY_CONTEXTS: other contexts with a batch in flight from which on a batch of more than
                               // FOLLOW_BUSY_CHAINS chains is packed (-1: FOLLOW_BUSY_CONTEXTS; 0: never look, always the latency shape)
};
Tuning read_tuning();
int64_t trunk_capacity(int64_t n_cand, const Tuning& tn);
uint32_t chain_table_size(int64_t n_cand);
hipError_t launch_setup(const DevModel& m, int64_t n, const double* params, const double* split, const ChainBufs& cb, int32_t* order,
                        int64_t n_rep, const double* jsfs, double* consts, int unfolded, hipStream_t stream);
int correct_cands_per_wave(int64_t n_items, const Tuning& tn);
bool trunk_follows(int cpw_chains, int64_t trunk_cap, const Tuning& tn);
hipError_t launch_correct(const DevModel& m, int64_t n_cand, const ChainBufs& cb, const double* split, const double* params,
                          int cpw, bool follow, int64_t est_chains, const Tuning& tn, int yield_nfev, hipStream_t stream, hipEvent_t after_packed = nullptr);
hipError_t launch_spectrum(const DevModel& m, int64_t n_cand, const int32_t* order, const double* split, const double* params,
                           const ChainBufs& cb, double* lc_out, double* pr_out, double* jafs, int32_t* status, double* diag,
                           int64_t n_rep, const double* jsfs, const double* consts, double* llk, bool follow, bool skip_post, bool single_waves, const Tuning& tn, hipStream_t stream,
                           int phase = 0);
hipError_t launch_forward(const DevModel& m, int64_t n_cand, const double* split, const double* params, int hold_mu, double* lh_out, double* pr_out,
                          int32_t* status, hipStream_t stream);
hipError_t launch_argmax(int64_t n_cand, int64_t n_rep, const double* llk, int32_t* best, double* best_llk, hipStream_t stream);
hipError_t launch_llh_const(int64_t n_rep, const double* jsfs, double* consts, int unfolded, hipStream_t stream);
hipError_t launch_llk(int64_t n_cand, const double* jafs, const int32_t* status, int64_t n_rep, const double* jsfs,
                      const double* consts, double* llk, int unfolded, hipStream_t stream);

}  // namespace misti

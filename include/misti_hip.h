/*
 * misti_hip.h - C ABI of the MI355X (gfx950) composite-likelihood engine for MiSTI.
 *
 * Drop-in boundary.  The reference (Genomics-HSE/MiSTI, pure Python) has no FFI;
 * its seam for this path is the Python method
 *
 *     MigrationInference.JAFSLikelihood(mu) -> float      MigrationInference.py:566-614
 *
 * together with the constructor (MigrationInference.py:41-200), SetModel /
 * MapParameters (:229-298) and SetJAFS (:202-227).  The entry points below are
 * what a ctypes binding for that seam binds (see INTEGRATION.md for the stub a
 * maintainer would add to MigrationInference.py).  One call evaluates a BATCH of
 * candidates (split time, migration-band rates, pulse rates) x bootstrap JSFS
 * replicates; the reference evaluates one (candidate, replicate) per call and
 * fans out over OS processes (README.md:110-115, test.bs/ scripts).
 *
 * Conventions
 *   - plain C types only; every buffer is caller-allocated, C-contiguous;
 *     the library copies what it needs and keeps no caller pointer after return;
 *   - every function returns 0 on success or a negative MISTI_E_* code and never
 *     calls exit() or lets a C++ exception cross the ABI; misti_last_error()
 *     gives the message of the last failure on the calling thread;
 *   - a context is used by one host thread at a time; work is issued on one HIP
 *     stream per context (replaceable with misti_set_stream).  Every batch of a
 *     context uses the same device workspaces, so batches of one context are
 *     ordered: misti_set_stream makes the new stream wait for everything already
 *     issued on the old one.  Use several contexts for concurrent batches;
 *   - there is NO CPU fallback: without a HIP device misti_create fails.
 */
#ifndef MISTI_HIP_H
#define MISTI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISTI_ABI_VERSION 6

/* model flags = keyword arguments of MigrationInference.__init__ (:53-74) */
#define MISTI_CPFIT     1u   /* cpfit=True    (MiSTI.py --cpfit)            */
#define MISTI_TRUE_EPS  2u   /* trueEPS=True  (MiSTI.py --trueEPS)          */
#define MISTI_SMOOTH    4u   /* smooth=True   (MiSTI.py default; --nosmooth clears) */
#define MISTI_UNFOLDED  8u   /* unfolded=True (MiSTI.py -uf)                */

/* error codes */
#define MISTI_E_ARG      (-1)  /* invalid argument / model                  */
#define MISTI_E_HIP      (-2)  /* HIP runtime error                         */
#define MISTI_E_NODEV    (-3)  /* no usable HIP device                      */
#define MISTI_E_LIMIT    (-4)  /* size beyond a compiled-in limit           */
#define MISTI_E_NOMEM    (-5)  /* out of host memory (a C++ allocation failed; never thrown across the ABI) */

/* per-candidate status (the reference prints a line and returns -inf, or exits) */
#define MISTI_OK             0
#define MISTI_NEG_PARAM      1  /* "Hit negative value of migration rate"   :569-572 */
#define MISTI_CORR_FAILED    2  /* "Lambda correction failed"               :346-348,576-578 */
#define MISTI_INF_COAL       3  /* two-population last interval (:475-476 and splitT == numT) */
#define MISTI_BAD_STRUCTURE  4  /* band/pulse/split inconsistent for this candidate (reference: PrintError + exit) */
#define MISTI_NUMERIC        5  /* non-finite intermediate / iteration cap  */
#define MISTI_STIFF          6  /* the contour solver for a stiff interval did not converge (very strong
                                   two-way migration together with a runaway rate) */

#define MISTI_MAX_BANDS   8
#define MISTI_MAX_PULSES  8
#define MISTI_MAX_PARAMS  16
#define MISTI_MAX_NUMT    255   /* numT + 1 <= 256 (four 64-lane passes of the smoothing step) */

/* One -mi option: MiSTI.py:63, MigrationInference.SetModel :236-258.
 * start/end are indices into the candidate's interval grid (after the extra
 * interval of a fractional split has been inserted, as in the reference).
 * end == -1 means "the candidate's split index" (test.bs/san_sar.bs.sh:36:
 * `-mi 1 4 ${st} ...`).  param >= 0 selects params[param] of the candidate
 * (an optimised band, MapParameters :294-296); param == -1 uses `value`. */
typedef struct {
    int32_t pop;      /* 0 or 1  (= reference index 1 or 2, source population) */
    int32_t start;
    int32_t end;
    int32_t param;
    double  value;
} misti_band_t;

/* One -pu option: MiSTI.py:65, SetModel :259-279, MapParameters :297-298. */
typedef struct {
    int32_t pop;
    int32_t time;     /* interval index */
    int32_t param;
    int32_t _pad;
    double  value;
} misti_pulse_t;

/* Everything MigrationInference.__init__ receives apart from the per-candidate
 * split time and the data JSFS. */
typedef struct {
    int32_t numT;          /* number of rate intervals; `times` has numT-1 entries (:102-107) */
    int32_t sample_date;   /* sampleDate kwarg: grid index of the second sample (:81-83)       */
    uint32_t flags;        /* MISTI_CPFIT | MISTI_TRUE_EPS | MISTI_SMOOTH | MISTI_UNFOLDED       */
    int32_t n_band;
    int32_t n_pulse;
    int32_t n_param;       /* length of a candidate's parameter vector (optMis + optPus, :288-293) */
    double  mixture_th;    /* mixtureTH kwarg (:184-185), CorrectLambda.py:267-272             */
    const double* times;   /* [numT-1] interval lengths                                         */
    const double* lh;      /* [numT][2] PSMC coalescence rates of genome 1 / genome 2           */
    const misti_band_t*  bands;   /* [n_band]  */
    const misti_pulse_t* pulses;  /* [n_pulse] */
} misti_model_t;

typedef struct misti_ctx misti_ctx;

/* ---- library ------------------------------------------------------------- */
int         misti_abi_version(void);
/* Hash of the sources and switches this library was built from (misti_amd/build.py: source_hash): measurements stored beside the
 * code (profiles/pmc_latest.json) name the build they were taken on, and bench.py prices a run only with counters of ITS build. */
const char* misti_build_id(void);
const char* misti_last_error(void);
int         misti_device_count(void);            /* HIP devices visible; <0 on error */

/* ---- context = one MigrationInference "model" on one device --------------- */
/* Replaces MigrationInference.__init__ + SetModel (:41-200, :229-289).       */
int misti_create(const misti_model_t* model, int device, misti_ctx** out);
int misti_destroy(misti_ctx* ctx);

/* Issue work on an existing hipStream_t (e.g. PyTorch's current stream) instead
 * of the context's own; pass NULL to go back.  The stream must belong to the
 * context's device.  When the stream actually changes, an event recorded on the old
 * stream is waited for on the new one (hipStreamWaitEvent): work already issued on
 * this context completes before anything issued later starts, because all batches of
 * a context share its workspaces (chain table, rates, trunk records). */
int misti_set_stream(misti_ctx* ctx, void* hip_stream);
/* The hipStream_t the context currently issues on (its own non-blocking stream unless replaced):
 * lets a caller order its own work or events against the batch (bench.py wraps it for RCCL). */
int misti_get_stream(misti_ctx* ctx, void** hip_stream);
int misti_sync(misti_ctx* ctx);

/* ---- batched JAFSLikelihood ----------------------------------------------- */
/* Host-buffer form.  Replaces a loop of
 *     m = MigrationInference(times, lh, jsfs[r], split_time[c], mi, pu, ...)   :41
 *     llk[c][r] = m.JAFSLikelihood(params[c])                                   :566
 * n_cand >= 0, n_rep >= 0 (n_rep == 0: spectrum only; llk may be NULL).
 *   split_time [n_cand]            fractional allowed (:89-99)
 *   params     [n_cand][n_param]   NULL allowed when n_param == 0
 *   band_bounds [n_cand][n_band][2] or NULL   per-candidate (start, end) of every -mi band, replacing
 *                                  the model's; end == -1 = the candidate's split index.  This is the
 *                                  reference's own recommended sweep (README.md:110-115: `-mi 1 0 {mc} ..
 *                                  -mi 1 {mc} {st} .. ::: st 20 21 .. ::: mc 8 9 ..`), where a band boundary
 *                                  varies independently of the split.  A candidate whose bounds violate
 *                                  SetModel's checks (:237-255: start >= sample date, start < end, no
 *                                  overlap within a population) gets status MISTI_BAD_STRUCTURE.
 *                                  NULL = the model's bounds for every candidate.
 *   jsfs       [n_rep][8]          rows "total + 7 classes" (SetJAFS :208-211);
 *                                  llh_const (:217-227) is computed inside
 *   llk        [n_cand][n_rep]     -inf on a soft failure (:572,:578)
 *   jafs       [n_cand][7]  or NULL   normalised expected spectrum (.JAFS, :583-584)
 *   lc         [n_cand][numT+1][2] or NULL   corrected rates (.lc); row numT is used
 *                                  only by a fractional split; unused rows = 0
 *   pr         [n_cand][numT+2][6] or NULL   pair-state trace (.Pr, :309,:350):
 *                                  row t = p11 g1,g2, p22 g1,g2, p12 g1,g2; the last row (numT+1)
 *                                  carries work counters of the correction: [0] residual batches, [3] solver steps taken
 *                                  from speculative slots, [4] max nfev, [5] regularised (SVD) steps; [1] dense (stiff)
 *                                  exponentials and [2] series terms only in a -DMISTI_WORK_COUNTERS=1 build (0 otherwise)
 *   status     [n_cand]     or NULL   MISTI_OK / MISTI_NEG_PARAM / ...
 */
int misti_eval_batch(misti_ctx* ctx, int64_t n_cand,
                     const double* split_time, const double* params, const int32_t* band_bounds,
                     int64_t n_rep, const double* jsfs,
                     double* llk, double* jafs, double* lc, double* pr, int32_t* status);

/* Device-buffer form: same arguments, every pointer is DEVICE memory on the
 * context's device; asynchronous on the context's stream (call misti_sync or
 * synchronise the stream yourself).  This is the form bench.py times. */
int misti_eval_batch_dev(misti_ctx* ctx, int64_t n_cand,
                         const double* d_split_time, const double* d_params, const int32_t* d_band_bounds,
                         int64_t n_rep, const double* d_jsfs,
                         double* d_llk, double* d_jafs, double* d_lc, double* d_pr, int32_t* d_status);

/* What the CALLER knows about the batches it issues on this context from now on (0 clears).  A batch is four launches on one stream and,
 * with many contexts' batches in flight, every launch boundary costs a round of the queue scheduler (~0.4 ms measured with 20 busy
 * queues): the launch that only exists for fractional split times is not made when the caller says there are none.
 *   MISTI_HINT_INTEGER_SPLITS   no split time of any candidate has a fractional part (the usual sweep: `::: st 20 21 22`, README.md:113).
 * The host-buffer form misti_eval_batch looks at its split times itself and needs no hint; the device-buffer form cannot (they live in
 * HBM).  The hint is VERIFIED on the device: a candidate with a fractional split in a batch issued under it gets status
 * MISTI_BAD_STRUCTURE and -inf - never a wrong value. */
#define MISTI_HINT_INTEGER_SPLITS 1u
int misti_set_hints(misti_ctx* ctx, uint32_t hints);

/* Replicate epilogue alone: llk[c][r] from already computed spectra (device
 * pointers).  status may be NULL (all OK).  MigrationInference.py:600-609 + :217-227. */
int misti_llk_dev(misti_ctx* ctx, int64_t n_cand, const double* d_jafs, const int32_t* d_status,
                  int64_t n_rep, const double* d_jsfs, double* d_llk);

/* Bootstrap reduction on device buffers: per replicate r the candidate with the largest
 * llk[c][r] (what test.bs/bs_conf_int.ipynb computes from the printed "llh =" lines before its
 * Student-t interval).  -inf and NaN never win; ties go to the lowest index; best[r] = -1 when no
 * candidate has a value.  d_best_llk may be NULL.  Asynchronous on the context's stream. */
int misti_argmax_dev(misti_ctx* ctx, int64_t n_cand, int64_t n_rep, const double* d_llk, int32_t* d_best, double* d_best_llk);

/* Diagnostic of the last batch evaluated on this context (either form): per candidate the
 * largest corrected rate x interval length before smoothing (NaN where the candidate has no
 * value, 0 with MISTI_TRUE_EPS).  From ~5 upwards the correction's residual is nearly flat in
 * that rate: SciPy's solver in the reference then stops at a noise-dependent "runaway" rate and
 * the reference's own log-likelihood is not determined to 1e-9.
 * Copies n_cand doubles to HOST memory and synchronises the stream. */
int misti_last_diag(misti_ctx* ctx, int64_t n_cand, double* max_rate_x_len);

/* ---- batched optimiser ------------------------------------------------------------ */
/* Replaces MigrationInference.Solve (MigrationInference.py:718-733: SciPy Nelder-Mead on -JAFSLikelihood, xatol =
 * fatol = tol, maxiter = 1000, started from the -mi / -pu initial values) for n_start starts at once - BASELINE
 * config 3 runs it from 16 384 random starts.  Every start follows scipy.optimize.minimize(method='Nelder-Mead')
 * decision for decision (same simplex, same evaluation count), so its result equals SciPy's on the same objective.
 * Simplices, function values and decisions stay in HBM; per iteration the reflection points of all starts are one
 * engine batch, their expansion / contraction points a second, shrunk vertices a third; finished starts cost nothing;
 * once few starts are left, iterations become speculative (misti_nm_last_spec_iterations).
 * Host buffers; synchronous.
 *   starts      [n_start][n_param]   (n_param >= 1)
 *   split_time  the split time of every evaluation (fractional allowed)
 *   jsfs_row    [8]                  the data JSFS (one replicate)
 *   maxiter     SciPy's maxiter (the reference passes 1000); the evaluation budget is unlimited, as there
 *   x           [n_start][n_param]   best vertex per start
 *   llh         [n_start]            its log-likelihood (-inf if no vertex has a value)
 *   nit, nfev   [n_start] or NULL    SciPy's OptimizeResult.nit / .nfev
 *   status      [n_start] or NULL    0 converged (both tolerances), 2 iteration budget (1: evaluation budget - only inside
 *                                    misti_basinhopping, whose minimisations run with SciPy's default maxfev = 200 x n_param; the
 *                                    budget is cut per evaluation as SciPy does: nfev never exceeds it) */
int misti_nm_solve(misti_ctx* ctx, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                   double xatol, double fatol, int32_t maxiter,
                   double* x, double* llh, int32_t* nit, int32_t* nfev, int32_t* status);

/* Batched basin hopping: scipy.optimize.basinhopping(func, x0, niter, T, stepsize, minimizer_kwargs=dict(method='Nelder-Mead'),
 * interval, target_accept_rate, stepwise_factor, rng=...) for n_start starts at once - the reference's global search,
 * MigrationInference.Solve(globalOpt=True) (MigrationInference.py:723-725: T = 0.5, Nelder-Mead with SciPy's defaults, i.e.
 * xatol = fatol = 1e-4, maxiter = maxfev = 200 x n_param), which BASELINE config 3 runs from 16 384 random starts.
 * Every start follows SciPy's runner step for step (_basinhopping.py: initial minimisation, then per hop
 * AdaptiveStepsize.take_step, RandomDisplacement, the local minimisation - misti_nm_solve's machinery, all starts in one set of
 * engine batches - Metropolis.accept_reject, Storage.update).  Incumbents, step sizes and decisions stay in HBM.
 * The random numbers are the CALLER's: SciPy draws, per hop, n_param uniforms for the displacement and then one for the
 * acceptance test from one generator; their number does not depend on the data, so the caller draws them up front -
 *   uniforms   [n_start][niter][n_param + 1]   numpy.random.Generator.random() values (in [0, 1)) in exactly that order,
 *                                              start s from the generator SciPy would be given for start s -
 * and start s then reproduces scipy.optimize.basinhopping(rng = that generator) on the same objective (x, fun, nfev,
 * minimization_failures).  Host buffers; synchronous.
 *   x, llh     [n_start][n_param], [n_start]   lowest successful minimum found (res.x, -res.fun)
 *   nfev, failures, accepted [n_start] or NULL  res.nfev, res.minimization_failures, hops accepted */
int misti_basinhopping(misti_ctx* ctx, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                       int32_t niter, double T, double stepsize, int32_t interval, double target_accept_rate, double stepwise_factor,
                       double xatol, double fatol, int32_t nm_maxiter, int64_t nm_maxfev, const double* uniforms,
                       double* x, double* llh, int32_t* nfev, int32_t* failures, int32_t* accepted);

/* Work counters of the last misti_nm_solve on this context: [0] iterations issued, [1] batch slots over all iterations
 * (live starts plus the slack of the two-iterations-old count that sizes the batches; x (2 + n_param) = candidates
 * handed to the engine after the initial simplices). */
int misti_nm_last_stats(misti_ctx* ctx, int64_t stats[2]);
/* ... and how many of those iterations were SPECULATIVE: with few starts still running (at most 1 024 / (4 + n_param)) all
 * 4 + n_param points SciPy could ask for in an iteration go out as one engine batch and one kernel takes its decisions
 * from the values it would have asked for - one chain latency per iteration instead of three; nfev stays SciPy's count. */
int misti_nm_last_spec_iterations(misti_ctx* ctx, int64_t* n);

/* ---- lanes: many batches in flight on ONE device ------------------------------------------------ */
/* One batch is latency-bound: its longest lambda-correction chain is as sequential as the reference's solver (the 4 096-point headline
 * grid keeps 64 of the chip's 1 024 SIMDs busy for 1.4 ms), so THROUGHPUT comes from independent batches in flight - the grids of
 * several data sets or models, the replicates of a bootstrap, the vertices of several optimisers: 3.1e7 evaluations/s on the headline
 * grid with twenty batches in flight against 2.9e6 one at a time.  A lane is one engine context with its own
 * non-blocking HIP stream; a misti_lanes object is `n_lanes` of them for one model on one device, so that a caller of the C ABI reaches
 * the overlapped rate without building the pool itself.  The reference's counterpart is one MigrationInference object per process and
 * as many processes as cores (MiSTI.py:213-214 under `parallel -j`, README.md:110-115).
 *   - batches of one lane run in submission order, batches of different lanes overlap on the device;
 *   - results are bit for bit those of a single context (a batch never depends on what else is in flight);
 *   - own streams map one to one onto hardware queues, of which the HIP runtime opens GPU_MAX_HW_QUEUES per process (default 4):
 *     unless the variable is already set, loading this library sets it to 22 - the runtime reads it when it initialises, i.e. at the
 *     process's first HIP call.  22 because the device runs 23 queues beside each other and time-slices them from the 24th ACTIVE one
 *     on (a burst of twenty batches then takes 10 ms instead of 2.7); under the cap, streams beyond it share queues instead - slower
 *     (their batches serialise), never the cliff.  A process whose runtime was initialised earlier with fewer queues still gets
 *     correct results, at a lower overlapped rate.
 * Threading: like a context, a misti_lanes object is used by one host thread at a time. */
typedef struct misti_lanes misti_lanes;
int misti_create_lanes(const misti_model_t* model, int device, int n_lanes, misti_lanes** out);   /* 1 <= n_lanes <= MISTI_MAX_LANES */
int misti_destroy_lanes(misti_lanes* lanes);            /* waits for every lane, then releases everything */
int misti_lanes_size(misti_lanes* lanes);
/* The i-th lane's context (borrowed: never misti_destroy it): misti_enable_timing, misti_get_stream, misti_last_diag ... per lane. */
int misti_lanes_context(misti_lanes* lanes, int i, misti_ctx** ctx);
int misti_lanes_set_hints(misti_lanes* lanes, uint32_t hints);     /* misti_set_hints on every lane */
/* misti_eval_batch_dev on one lane; returns when the batch is ISSUED.  `lane` >= 0 names the lane; MISTI_LANE_ANY takes a lane that has
 * nothing in flight if there is one, else the next in round-robin order.  *lane_used (may be NULL) receives the lane the batch went to.
 * Every pointer is device memory on the object's device, complete before the call (the lanes' streams are non-blocking: they do not
 * wait for the null stream); output buffers belong to the batch until its lane has been waited for, and a lane's NEXT batch may reuse
 * them only if the caller is done with the previous results (batches of a lane are ordered, so the device side is safe either way). */
#define MISTI_LANE_ANY (-1)
#define MISTI_MAX_LANES 64
int misti_lanes_eval_batch_dev(misti_lanes* lanes, int lane, int64_t n_cand,
                               const double* d_split_time, const double* d_params, const int32_t* d_band_bounds,
                               int64_t n_rep, const double* d_jsfs,
                               double* d_llk, double* d_jafs, double* d_lc, double* d_pr, int32_t* d_status, int* lane_used);
int misti_lanes_wait(misti_lanes* lanes, int lane);     /* everything issued on that lane has finished */
int misti_lanes_sync(misti_lanes* lanes);               /* ... on every lane */
/* 1 if the lane has work in flight, 0 if not (never blocks); < 0 on error */
int misti_lanes_busy(misti_lanes* lanes, int lane);

/* ---- several devices --------------------------------------------------------------------- */
/* One model on a LIST of devices - 1, 2, 4 or 8 GPUs of a node from ONE process: one engine context and one host thread per
 * entry (a device may be listed more than once: two contexts overlap their batches on it).  Replaces what the reference does
 * with more than one processor: `parallel -j 20 ./MiSTI.py ... ::: st ... ::: mc ... >> res.out` (README.md:110-115) and the
 * bash loops of test.bs/ (san_sar.bs.sh:29-36) - one OS process per grid point, results concatenated from stdout.
 * Candidates are independent, so there is no exchange between devices during evaluation; what must not be split is a CHAIN:
 * candidates with bitwise identical parameter vectors and band bounds share one lambda-correction chain,
 * computed once per context that holds any of them.  misti_multi_eval_batch therefore deals whole chains to the contexts - the
 * costliest first, each to the context with the least work so far (a chain costs its length: the corrected two-population
 * intervals up to the largest split index of its members, + 1/64 per member; chains of equal cost end up round-robin in order of
 * first appearance; a batch without parameters is one chain and is interleaved instead) -, runs the batch on every context at the
 * same time (one persistent host thread per context) and gathers / scatters every candidate's rows straight from / into the
 * caller's buffers: the result is bit for bit that of misti_eval_batch on one device.  Same arguments and conventions as
 * misti_eval_batch.  A failure on any context - a C++ exception in its worker thread included - fails the call with that
 * context's message; it never terminates the process.
 * Threading: like a misti_ctx, a misti_multi is used by ONE host thread at a time - its worker dispatch, shards and "last" records are
 * per object.  The evaluating entry points take a per-object lock, so two threads calling into one object are serialised, never
 * interleaved; callers that want concurrent batches create one object per thread.  misti_destroy_multi must not race a call. */
typedef struct misti_multi misti_multi;
int misti_create_multi(const misti_model_t* model, int n_dev, const int* devices, misti_multi** out);
int misti_destroy_multi(misti_multi* m);
int misti_multi_size(misti_multi* m);                                        /* contexts (= entries of the device list) */
int misti_multi_context(misti_multi* m, int i, misti_ctx** ctx, int* device); /* the i-th context (borrowed) and its device; either may be NULL */
int misti_multi_eval_batch(misti_multi* m, int64_t n_cand,
                           const double* split_time, const double* params, const int32_t* band_bounds,
                           int64_t n_rep, const double* jsfs,
                           double* llk, double* jafs, double* lc, double* pr, int32_t* status);
/* Shards of the last misti_multi_eval_batch: candidates and chains per context ([misti_multi_size] each; either may be NULL);
 * misti_multi_last_cost: the summed chain cost per context (what the dealing balances: they differ by less than one chain). */
int misti_multi_last_shards(misti_multi* m, int64_t* n_cand, int64_t* n_chain);
int misti_multi_last_cost(misti_multi* m, double* cost);
/* Device-resident form with the gather INSIDE the library (RCCL over xGMI).  The caller has sharded its candidates: context i
 * (device i of the list) evaluates n_cand[i] candidates from DEVICE pointers on its own device, exactly as misti_eval_batch_dev,
 * and the log-likelihoods are then all-gathered on the devices - ncclAllGather, in place, on a single-process communicator over
 * the device list (ncclCommInitAll at the first call; every device may be listed once), issued on each context's stream behind
 * its batch - so that EVERY device ends up with the whole table.  This is what the reference does by concatenating the stdout
 * of its processes (README.md:113-114), kept in HBM; the rank-per-GPU counterpart is misti_amd/dist.py (torch.distributed).
 *   n_cand          [D]   candidates of shard i (0 allowed), each <= rows_per_shard
 *   d_split_time, d_params, d_band_bounds, d_jsfs   [D] host arrays of device pointers (arrays as in misti_eval_batch_dev; d_band_bounds
 *                         or any of its entries may be NULL; the replicate table d_jsfs[i] [n_rep][8] must be resident on every device)
 *   d_llk_all       [D]   device i's table [D][rows_per_shard][n_rep]: block r = shard r's rows, rows beyond n_cand[r] NaN
 *   d_status_all    [D] or NULL   device i's table [D][rows_per_shard] of per-candidate status (-1 beyond n_cand[r])
 * Asynchronous: returns when everything is issued; misti_multi_sync waits for every context's stream.  librccl.so.1 is bound
 * at the first call (dlopen by soname: a process that already maps an RCCL - PyTorch-ROCm does - uses that copy; MISTI_RCCL_LIB=<path>
 * in the environment names another build to bind instead - read once, at the first gathered call of the process). */
int misti_multi_eval_batch_dev(misti_multi* m, const int64_t* n_cand, int64_t rows_per_shard,
                               const double* const* d_split_time, const double* const* d_params, const int32_t* const* d_band_bounds,
                               int64_t n_rep, const double* const* d_jsfs, double* const* d_llk_all, int32_t* const* d_status_all);
int misti_multi_sync(misti_multi* m);
/* misti_nm_solve / misti_basinhopping with the starts dealt to the contexts in contiguous blocks, all contexts searching at the
 * same time (BASELINE config 3: 16 384 starts -> 2 048 per GPU on a node).  Starts are independent searches and a start's
 * trajectory does not depend on what else travels in its batches: results equal the single-device call's, start for start.
 * Arguments as misti_nm_solve / misti_basinhopping (`uniforms` is indexed by start, so a block takes its slice). */
int misti_multi_nm_solve(misti_multi* m, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                         double xatol, double fatol, int32_t maxiter,
                         double* x, double* llh, int32_t* nit, int32_t* nfev, int32_t* status);
int misti_multi_basinhopping(misti_multi* m, int64_t n_start, const double* starts, double split_time, const double* jsfs_row,
                             int32_t niter, double T, double stepsize, int32_t interval, double target_accept_rate, double stepwise_factor,
                             double xatol, double fatol, int32_t nm_maxiter, int64_t nm_maxfev, const double* uniforms,
                             double* x, double* llh, int32_t* nfev, int32_t* failures, int32_t* accepted);

/* ---- solver trace (parity diagnostics) ---------------------------------------- */
/* The reference's corrected rates are DEFINED by where SciPy's trust-region iteration stops
 * (CorrectLambda.py:85,260,303,305 -> scipy.optimize.least_squares); tests compare that iteration
 * itself.  When enabled, every batch of this context records per candidate and interval one word
 *     bits 0-15  nfev        residual evaluations SciPy would count (OptimizeResult.nfev)
 *     bits 16-19 status      SciPy's termination code: 0 max_nfev, 1 gtol, 2 ftol, 3 xtol, 4 ftol+xtol
 *     bits 20-23 kind        0 no solve (trueEPS / T == 0), 1 closed form (SolveNoMigration1 :213-235,
 *                            cpfit post-split :366), 2 bounded TRF (SolveNoMigration :253-264, FitSinglePop
 *                            :82-92), 3 unbounded TRF (SolveLambdaSystem :299-305)
 *     bit 24     noise       default fit: the solve went on past a gradient test (gtol) that its noise-free residual
 *                            satisfied, because the reference's own residual - whose rounding noise is measured on the
 *                            spot - would typically not have satisfied it
 *     bit 25     stall       default fit: the solve was PREDICTED to stall and the starting point was returned with status 3 - what the
 *                            reference's noisy iteration does on very short intervals after 14 - 23 evaluations (misti_kernels.hip: the stall rule);
 *                            nfev is then 1, the evaluations the device actually made
 * for intervals 0..numT (row numT is used only by a fractional split); and, for batches of at most
 * MISTI_TRACE_MAX_CAND candidates, the trial points of the unbounded solves (stretched to the unit
 * interval as the reference does, :293-298), at most MISTI_TRACE_MAX_ITER per interval.
 * misti_last_solver_trace copies to HOST memory and synchronises the stream:
 *   trace    [n_cand][numT+1]                              (n_cand = size of the last batch)
 *   iterates [numT][MISTI_TRACE_MAX_ITER][2] or NULL       trial points of candidate `cand`'s chain
 *                                                          (NaN beyond nfev); row = interval on the
 *                                                          shared grid; the interval shortened by a
 *                                                          fractional split is not recorded */
#define MISTI_TRACE_NOISE_BIT (1 << 24)
#define MISTI_TRACE_STALL_BIT (1 << 25)
#define MISTI_TRACE_MAX_CAND 64
#define MISTI_TRACE_MAX_ITER 200
int misti_enable_solver_trace(misti_ctx* ctx, int on);
int misti_last_solver_trace(misti_ctx* ctx, int64_t n_cand, int32_t* trace, int64_t cand, double* iterates);

/* ---- forward map (data generation; TestModel route) --------------------------- */
/* Replaces MigrationInference.CoalescentRates (MigrationInference.py:542-564) and
 * CorrectLambda.CoalRates (CorrectLambda.py:112-122) for a batch of candidates: the
 * model's rates are taken as the TRUE per-population rates and the rates a single-genome
 * (PSMC) analysis would infer under the candidate's migration model are returned.
 *   lh   [n_cand][numT+1][2]   rows below the candidate's split: -log(P[no coalescence])/T
 *                              of that genome's pair chain; other rows: the model's rates
 *                              (row numT is used only by a fractional split, else 0);
 *                              NaN rows when status is not MISTI_OK / MISTI_INF_COAL
 *   pr   [n_cand][numT+2][6] or NULL   pair-state trace as in misti_eval_batch (rows 0..split)
 *   status [n_cand] or NULL
 *   hold_mu  0: every interval is evaluated with its own migration rates (the model as specified;
 *               what generating self-consistent data wants);
 *            1: bit-for-bit the reference's behaviour - CoalescentRates never sets the migration
 *               rates of its CorrectLambda object, so ALL intervals see what the preceding
 *               CorrectLambdas loop left there (:324): the rates of the candidate's last
 *               two-population interval.  Identical to 0 when migration is constant up to the split.
 * Host-buffer and device-buffer (asynchronous on the context's stream) forms. */
int misti_forward_rates(misti_ctx* ctx, int64_t n_cand, const double* split_time, const double* params, int hold_mu,
                        double* lh, double* pr, int32_t* status);
int misti_forward_rates_dev(misti_ctx* ctx, int64_t n_cand, const double* d_split_time, const double* d_params, int hold_mu,
                            double* d_lh, double* d_pr, int32_t* d_status);

/* ---- measurement ----------------------------------------------------------- */
/* When enabled, the stages of every batch of this context are bracketed by HIP events
 * on its stream.  misti_kernel_times returns the accumulated device time (ms)
 * and batch counts since the last reset: [0] prepare + chain discovery + lambda-correction
 * of the chains, [1] trunks/tails + spectrum kernel (incl. the replicate epilogue for <= 8
 * replicates), [2] separate replicate (llk) kernel (more than 8 replicates, misti_llk_dev). */
int misti_enable_timing(misti_ctx* ctx, int on);
int misti_kernel_times(misti_ctx* ctx, double ms[3], int64_t launches[3], int reset);

/* ---- introspection (tests) ------------------------------------------------- */
/* Constant structure of the 44-state chain as the library derived it:
 *   gen[4][44][44]  integer coefficient patterns A0, A1 (coalescence in pop 0/1),
 *                   B0, B1 (migration out of pop 0/1): M = la0*A0+la1*A1+mu0*B0+mu1*B1
 *                   (TwoPopulations.UpdateMatrixCol :336-359), row = destination
 *   jaf[44][7]      StateToJAF (:188-219)
 * Either pointer may be NULL. */
int misti_tables(int32_t* gen, int32_t* jaf);

#ifdef __cplusplus
}
#endif
#endif /* MISTI_HIP_H */

// Compile-time sizes shared by the table builder, the kernels and the C-ABI layer.
#pragma once
namespace misti {
constexpr int NS2 = 44;               // two-population states
constexpr int NS1 = 8;                // one-population states
constexpr int MAXNZ = 4;              // off-diagonal entries per generator row
constexpr int MAXPULSE = 12;          // entries per row of the pulse operator
#ifndef MISTI_WAVES_PER_BLOCK
#define MISTI_WAVES_PER_BLOCK 4
#endif
constexpr int WAVES_PER_BLOCK = MISTI_WAVES_PER_BLOCK;    // candidates per workgroup of kernel 2 (one wavefront each)
constexpr int TALBOT_N = 28;           // nodes of the Talbot contour (conjugate pairs folded: TALBOT_HALF solves)
constexpr int TALBOT_HALF = TALBOT_N / 2;
constexpr int INV_TABLE = 1024;         // reciprocal table for the series (also the cap on terms per series)
constexpr int TRUNK_REC = 3 * NS2;    // doubles per trunk record: state vector | occupation integral before | from the sample date
constexpr int TRUNK_MAX_CHAINS = 8192; // trunk buffer bound: 8 192 chains x numT records x 1 056 B (1.1 GB at numT = 128)
constexpr int TRUNK_MIN_SHARE = 8;    // the trunk runs when a chain has on average at least this many candidates
constexpr int FOLLOW_MAX_CHAINS = 1024; // up to this many chains a batch runs one chain per wave (speculation, trunk wave following), workgroups
                                        // pulling chains longest-first from a queue; beyond it chains are packed 10 per wave.  1 024 two-wave
                                        // workgroups are resident at 2 waves per SIMD: one round.  Measured (MI355X): 1 024 chains 1.68 ms
                                        // against 2.71 ms packed; 2 048 chains (config 5) 5.07 against 5.50 ms alone, and with 20 batches in
                                        // flight 1.7e7 against 4.1e7 evals/s - a packed wave carries ten chains per instruction stream
constexpr int FOLLOW_BUSY_CHAINS = 128; // ... but only up to this many while the device is busy with OTHER contexts' batches (FOLLOW_BUSY_CONTEXTS of them or
                                        // more have a batch in flight): one chain per wave buys latency with the whole chip, and a caller who
                                        // overlaps batches wants throughput.  Measured, 20 batches in flight: 128 chains 3.3e7 evals/s either way;
                                        // 256 chains 3.7e7 one per wave, 4.7e7 packed; 512: 4.0e7 / 6.3e7; 1 024 (config2x16): 4.2e7 / 7.7e7
constexpr int FOLLOW_BUSY_CONTEXTS = 3;
constexpr int YIELD_NFEV = 8;           // packed launches: a solve still running after this many evaluations hands its chain to the resume launch
constexpr int PAIR_TABLE = 8 * 8 * 2 * 16 * 4;   // XCC x SE x SH x CU x SIMD counters of the placement-aware role choice (correct_follow_kernel)
constexpr int FOLLOW_MIN_BLOCKS = 256;   // workgroups of the one-chain-per-wave launch whatever the (possibly stale) chain-count hint says
constexpr int SMOOTH_REPS = 4;        // numT <= 64 * SMOOTH_REPS (smoothing pass keeps runs in registers)
}  // namespace misti

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
constexpr int SMOOTH_REPS = 4;        // numT <= 64 * SMOOTH_REPS (smoothing pass keeps runs in registers)
}  // namespace misti

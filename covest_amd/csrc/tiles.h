// tiles.h -- the tile table of the pmf recurrence (see streams.h), shared by the
// host builder (capi.cpp: build_tiles) and the fast kernels.
#pragma once
#include <stdint.h>

namespace covest {

constexpr int kTileBins = 32;
constexpr int kScaleBits = 540;
constexpr double kScaleLn = 540.0 * 0.693147180559945309417232121458; // ln 2^SC
constexpr double kWindowLn = -760.0; // terms below e^-760 are 0 in double

// Tiles of <= 32 consecutive keys over the evaluated bins (built on the host,
// capi.cpp: build_tiles).  All arrays live in one device buffer; everything
// indexed by tile is wave-uniform and read through the scalar cache.
struct TileView {
    int32_t n_tiles;
    const double *first_key;   // [n_tiles] k0 as a double
    const int32_t *n_bins;     // [n_tiles] keys in the tile (1..32)
    const int32_t *run_start;  // [n_tiles] 1: keys are not contiguous with the previous tile -> re-anchor
    const double *lgam_prev;   // [n_tiles] lgamma(k0)       = ln (k0-1)!
    const double *lgam_last;   // [n_tiles] lgamma(k0 + nb)  = ln (k0+nb-1)!
    const double *renorm;      // [n_tiles] (k0-1)! / (k0+nb-1)!   carries v into the next tile
    const double *scal;        // [n_tiles][32] 2^-SC (k0-1)!/(k0+b)!
    const double *cnt;         // [n_tiles][32] h_j (0 for padding and filler keys)
    const double *in_sp;       // [n_tiles][32] 1.0 if the key is a histogram key (counts in sp_j), else 0.0
};

} // namespace covest

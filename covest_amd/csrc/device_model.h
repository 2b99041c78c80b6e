// device_model.h -- plain-data views shared by the host side (host.h and the *_host.cpp, abi_*.cpp files) and the
// gfx950 kernels.  Everything here is passed to kernels BY VALUE as a kernel
// argument (well under the 4 KB kernarg limit), so the scalar unit serves the
// per-model constants from SGPRs / the scalar cache.
#pragma once
#include <stdint.h>

namespace covest {

constexpr int kMaxParams = 5;
constexpr int kMaxErr = 64;

// Histogram bins as the kernels see them.  Layout in HBM: structure of arrays,
// one double per bin per array, contiguous, so that a wavefront reading bins
// [b, b+64) issues one coalesced 512-B load per array.  The three arrays of a
// 10k-bin histogram are 240 KB: L2-resident on every XCD after first touch.
struct BinView {
    int64_t n;          // number of bins in this view
    const double *key;  // j as a double (max(j, 0): c_src/covest_poissonmodule.c:22 runs no iteration for j <= 0)
    const double *lgam; // lgamma(j + 1), rounded from long double
    const double *cnt;  // h_j
};

struct DevModel {
    int32_t kind;  // 0 basic, 1 repeats
    int32_t k, r, n_err;
    double comb[kMaxErr];    // self.comb[s]                       covest/models.py:25
    double pow3neg[kMaxErr]; // 3 ** -s, libm pow on the host      covest/models.py:77
    double ln_comb[kMaxErr]; // ln comb[s] (host libm; -inf for the padding classes): streams.h StreamSet::init
    double lo[kMaxParams];   // self.bounds, NaN = None            covest/models.py:23,179
    double hi[kMaxParams];
    double tail;
    BinView bins; // bins that must be evaluated (see abi_model.cpp: all keys iff tail != 0)
};

// Where the grid points of one launch come from.
struct PointSource {
    int32_t is_grid;
    // list mode: params[n][P] and, for the repeats model, threshold_o per point
    const double *params;
    const int32_t *t_list;
    // grid mode: axes in itertools.product order (last axis fastest)
    const double *axis[kMaxParams];
    int64_t len[kMaxParams];
    int64_t flat_begin;       // first flat index of this block
    const int32_t *t_table;   // threshold_o over the (q1, q2, q) sub-grid
};

} // namespace covest

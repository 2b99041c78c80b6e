// tiles_host.cpp -- the host side of the recurrence kernels' tables: the bin views and the tile table (tiles.h).
#include "host.h"

using namespace covest;

namespace covest {

// One buffer, one copy: [key | lgam | cnt].
int upload_bins(DevBuf &buf, BinView &view, const std::vector<double> &key,
                const std::vector<double> &lgam, const std::vector<double> &cnt)
{
    const size_t n = key.size();
    view.n = (int64_t)n;
    view.key = view.lgam = view.cnt = nullptr;
    if (n == 0)
        return COVEST_OK;
    HIP_TRY(buf.reserve(3 * n * sizeof(double)));
    double *base = buf.as<double>();
    {
        SharedStage &ss = shared_stage();
        std::lock_guard<std::mutex> hold(ss.mu);
        HIP_TRY(ss.buf.reserve(3 * n * sizeof(double)));
        double *stage = ss.buf.as<double>();
        std::copy(key.begin(), key.end(), stage);
        std::copy(lgam.begin(), lgam.end(), stage + n);
        std::copy(cnt.begin(), cnt.end(), stage + 2 * n);
        HIP_TRY(hipMemcpy(base, stage, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    }
    view.key = base;
    view.lgam = base + n;
    view.cnt = base + 2 * n;
    return COVEST_OK;
}

// Tile table of streams.h over the evaluated bins: keys sorted ascending, split
// into runs of consecutive keys (gaps of up to kGapFill keys are bridged with
// filler keys that are stepped over but neither logged nor summed), each run cut
// into tiles of <= 32 keys.  Returns false when the fast kernels do not apply.
int build_tiles(covest_model *m, std::vector<HostBin> bins)
{
    m->has_tiles = false;
    if (m->dm.n_err > 32 || bins.empty()) // (the recurrence kernels hold max_error <= 32 error classes)
        return COVEST_OK;
    std::sort(bins.begin(), bins.end(), [](const HostBin &a, const HostBin &b) { return a.key < b.key; });
    if (bins.front().key < 1 || bins.back().key > kMaxFastKey)
        return COVEST_OK;
    struct Tile {
        int k0, nb, run_start;
    };
    std::vector<Tile> tiles;
    std::vector<double> scal, cnt; // (scal WITHOUT the 2^kBasicShift the device table carries: applied at the upload)
    std::vector<int32_t> row_bin;
    size_t i = 0;
    while (i < bins.size()) {
        // one run: keys bins[i..j) with gaps <= kGapFill
        size_t j = i + 1;
        while (j < bins.size() && bins[j].key - bins[j - 1].key <= kGapFill + 1)
            ++j;
        const int first = bins[i].key, last = bins[j - 1].key;
        size_t cur = i;
        for (int k0 = first; k0 <= last; k0 += kTileBins) {
            const int nb = std::min(kTileBins, last - k0 + 1);
            tiles.push_back({k0, nb, k0 == first ? 1 : 0});
            long double sc = ldexpl(1.0L, -kScaleBits);
            for (int b = 0; b < kTileBins; ++b) {
                double sv = 0.0, cv = 0.0;
                int32_t which = -1;
                if (b < nb) {
                    const int key = k0 + b;
                    sc /= (long double)key;
                    if (cur < j && bins[cur].key == key) {
                        sv = (double)sc;
                        cv = bins[cur].cnt;
                        which = bins[cur].index;
                        ++cur;
                    } // else a FILLER key (a gap of the histogram the recurrence walks through): scale 0, so
                      // that its p_j is exactly 0 -- it is no key of the reference's p_j dict, and must add
                      // nothing to sp_j (covest/models.py:103) and take no log
                }
                scal.push_back(sv);
                cnt.push_back(cv);
                row_bin.push_back(which);
            }
        }
        i = j;
    }
    const size_t nt = tiles.size();
    // layout: [first_key | lgam_prev | lgam_last | renorm] doubles, then scal/cnt, then int32 n_bins/run_start
    std::vector<double> dbl(4 * nt);
    std::vector<int32_t> ints(2 * nt);
    for (size_t t = 0; t < nt; ++t) {
        const Tile &tl = tiles[t];
        dbl[t] = (double)tl.k0;
        dbl[nt + t] = lgamma_of_factorial((int64_t)tl.k0 - 1);
        dbl[2 * nt + t] = lgamma_of_factorial((int64_t)(tl.k0 + tl.nb) - 1);
        long double rn = 1.0L;
        for (int b = 0; b < tl.nb; ++b)
            rn /= (long double)(tl.k0 + b);
        dbl[3 * nt + t] = (double)rn;
        ints[t] = tl.nb;
        ints[nt + t] = tl.run_start;
    }
    // items (tiles.h): runs of all-zero-count tiles (they exist only with a tail) are grouped, up to 32 per item
    std::vector<int32_t> item_first, item_ntiles, item_sum;
    std::vector<double> item_cnt;
#ifdef COVEST_DIAG
    const bool no_sum_items = std::getenv("COVEST_NO_SUM_ITEMS") != nullptr; // diagnostic builds: every tile a plain item
#else
    const bool no_sum_items = false;
#endif
    for (size_t t = 0; t < nt;) {
        auto all_zero = [&](size_t tt) {
            for (int b = 0; b < kTileBins; ++b)
                if (cnt[tt * kTileBins + (size_t)b] != 0.0)
                    return false;
            return true;
        };
        if (!all_zero(t) || no_sum_items) {
            item_first.push_back((int32_t)t);
            item_ntiles.push_back(1);
            item_sum.push_back(0);
            item_cnt.insert(item_cnt.end(), cnt.begin() + (std::ptrdiff_t)(t * kTileBins),
                            cnt.begin() + (std::ptrdiff_t)((t + 1) * kTileBins));
            ++t;
            continue;
        }
        size_t e = t + 1;
        while (e < nt && e - t < (size_t)kTileBins && all_zero(e))
            ++e;
        item_first.push_back((int32_t)t);
        item_ntiles.push_back((int32_t)(e - t));
        item_sum.push_back(1);
        item_cnt.insert(item_cnt.end(), (size_t)kTileBins, 0.0);
        t = e;
    }
    const size_t ni = item_first.size();
    m->rows_contracted = (double)ni * kTileBins;
    {
        size_t low = 0;
        for (const Tile &tl : tiles)
            low += tl.k0 <= kLowKeyTile ? 1 : 0;
        m->low_tile_share = nt ? (double)low / (double)nt : 0.0;
    }
    m->keys_logged = 0.0;
    for (double c : cnt)
        m->keys_logged += c != 0.0 ? 1.0 : 0.0;
    // K-factored's view of the rows (tiles.h): the scale of a plain item's rows as a factor (and its reciprocal, for
    // the clamp in the row's units) and as the constant it adds to the item's sum of h_j log p_j
    std::vector<double> item_scal(ni * kTileBins, 0.0), item_iscal(ni * kTileBins, 0.0), item_lconst(ni, 0.0);
    for (size_t i2 = 0; i2 < ni; ++i2) {
        if (item_sum[i2]) {
            for (int b = 0; b < kTileBins; ++b)
                item_scal[i2 * kTileBins + (size_t)b] = 1.0;
            continue;
        }
        const size_t t = (size_t)item_first[i2];
        long double lc = 0.0L, lratio = 0.0L; // ln((k0-1)!/(k0+b)!) = -sum_{i=k0}^{k0+b} ln i
        for (int b = 0; b < tiles[t].nb; ++b) {
            lratio -= logl((long double)(tiles[t].k0 + b));
            const double sv = scal[t * kTileBins + (size_t)b];
            if (sv == 0.0)
                continue; // filler key
            item_scal[i2 * kTileBins + (size_t)b] = sv;
            if (cnt[t * kTileBins + (size_t)b] != 0.0) // (a row without a count takes no log and is never "low": 0)
                item_iscal[i2 * kTileBins + (size_t)b] = 1.0 / sv;
            lc += (long double)cnt[t * kTileBins + (size_t)b] * lratio;
        }
        item_lconst[i2] = (double)lc;
    }
    // K-basic's closed form (ll_basic.hip): suffix sums over the counted keys of the tiles t .. nt - 1
    std::vector<double> suf(5 * (nt + 1) + 2, 0.0);
    {
        long double s_h = 0.0L, s_jh = 0.0L, s_lgh = 0.0L;
        double first_key = 0.0, first_lg = 0.0;
        for (size_t t = nt; t-- > 0;) {
            for (int b = tiles[t].nb - 1; b >= 0; --b) {
                const double h = cnt[t * kTileBins + (size_t)b];
                if (h == 0.0)
                    continue;
                const int key = tiles[t].k0 + b;
                const double lg = lgamma_at(key);
                s_h += (long double)h;
                s_jh += (long double)h * (long double)key;
                s_lgh += (long double)h * (long double)lg;
                first_key = (double)key;
                first_lg = lg;
                if (suf[5 * (nt + 1)] == 0.0) { // the first one met from the end: the last counted key
                    suf[5 * (nt + 1)] = (double)key;
                    suf[5 * (nt + 1) + 1] = lg;
                }
            }
            suf[t] = (double)s_h;
            suf[(nt + 1) + t] = (double)s_jh;
            suf[2 * (nt + 1) + t] = (double)s_lgh;
            suf[3 * (nt + 1) + t] = first_key;
            suf[4 * (nt + 1) + t] = first_lg;
        }
    }
    const size_t n_arrays = 4 * nt + 2 * nt * kTileBins + 3 * ni * kTileBins + ni + suf.size();
    const size_t n_dbl = (size_t)tile_dbl_count((int32_t)nt, (int32_t)ni); // (the arrays, padded to a cache line, + the records)
    if ((size_t)tile_arrays_dbl((int32_t)nt, (int32_t)ni) < n_arrays)
        return fail(COVEST_E_INVALID, "tile table layout (internal)");
    std::vector<int32_t> tile_zero(nt, 0);
    for (size_t i2 = 0; i2 < ni; ++i2)
        if (item_sum[i2])
            for (int32_t r = 0; r < item_ntiles[i2]; ++r)
                tile_zero[(size_t)(item_first[i2] + r)] = 1;
    std::vector<int32_t> tile_filler(nt, 0); // a row inside the tile's keys that is no key of the histogram
    for (size_t t = 0; t < nt; ++t)
        for (int b = 0; b < tiles[t].nb; ++b)
            if (row_bin[t * kTileBins + (size_t)b] < 0)
                tile_filler[t] = 1;
    const size_t bytes = n_dbl * sizeof(double) + (4 * nt + 3 * ni + nt * kTileBins) * sizeof(int32_t);
    HIP_TRY(m->tiles_buf.reserve(bytes));
    double *base = m->tiles_buf.as<double>();
    int32_t *ibase = reinterpret_cast<int32_t *>(base + n_dbl);
    SharedStage &ss = shared_stage();
    std::lock_guard<std::mutex> hold(ss.mu);
    HIP_TRY(ss.buf.reserve(bytes)); // one copy instead of five
    char *const stage = ss.buf.as<char>();
    char *sp = stage;
    auto put = [&](const void *src, size_t n) {
        std::memcpy(sp, src, n);
        sp += n;
    };
    put(dbl.data(), 4 * nt * sizeof(double));
    {
        std::vector<double> scal_dev(scal);
        for (double &v : scal_dev)
            v *= kBasicScale; // tiles.h kBasicShift (exact: a power of two, and 2^-540 (k0-1)!/(k0+b)! >= 1e-303)
        put(scal_dev.data(), nt * kTileBins * sizeof(double));
    }
    put(cnt.data(), nt * kTileBins * sizeof(double));
    put(item_cnt.data(), ni * kTileBins * sizeof(double));
    put(item_scal.data(), ni * kTileBins * sizeof(double));
    put(item_iscal.data(), ni * kTileBins * sizeof(double));
    put(item_lconst.data(), ni * sizeof(double));
    put(suf.data(), suf.size() * sizeof(double));
    {
        std::vector<double> gap((size_t)tile_arrays_dbl((int32_t)nt, (int32_t)ni) - n_arrays, 0.0);
        put(gap.data(), gap.size() * sizeof(double));
        std::vector<TileRec> recs(nt);
        for (size_t t = 0; t < nt; ++t) {
            TileRec &r = recs[t];
            r.k0 = dbl[t];
            r.lgam_prev = dbl[nt + t];
            r.lgam_last = dbl[2 * nt + t];
            r.renorm = dbl[3 * nt + t];
            r.nb = ints[t];
            r.run_start = ints[nt + t];
            r.all_zero = tile_zero[t];
            r.has_filler = tile_filler[t];
            r.pad[0] = r.pad[1] = 0.0;
        }
        put(recs.data(), nt * sizeof(TileRec));
    }
    put(ints.data(), 2 * nt * sizeof(int32_t));
    put(tile_zero.data(), nt * sizeof(int32_t));
    put(tile_filler.data(), nt * sizeof(int32_t));
    put(item_first.data(), ni * sizeof(int32_t));
    put(item_ntiles.data(), ni * sizeof(int32_t));
    put(item_sum.data(), ni * sizeof(int32_t));
    put(row_bin.data(), nt * kTileBins * sizeof(int32_t));
    HIP_TRY(hipMemcpy(base, stage, bytes, hipMemcpyHostToDevice));
    m->tv = tile_view_from((int32_t)nt, (int32_t)ni, base, ibase);
    m->has_tiles = true;
    return COVEST_OK;
}

// p_clamp of direct_point.h for a launch whose largest threshold_o is t_max.
double clamp_for(const covest_model *m, int t_max)
{
    return (double)(m->dm.n_err + std::max(t_max, 2)) * kClampPerTerm;
}


} // namespace covest

// host.h -- what the host-side translation units of libcovest_amd.so share (round 4: capi.cpp, 2 900 lines, cut into
//   host_common.cpp   error state, device selection, ln j! table, threshold_o (host, libm)
//   tiles_host.cpp    bins and the tile table of the recurrence kernels (tiles.h)
//   plan_factored.cpp K-factored's work descriptions: dense-grid parts, point lists (tiles.h FactoredPlan)
//   abi_model.cpp     covest_model_*, covest_eval_points, covest_probabilities, kernel dispatch
//   abi_grid.cpp      covest_grid_*
//   kmer_host.cpp     covest_kmer_*
//   thin_host.cpp     covest_thin_histogram*
// ).  Nothing here is part of the C ABI (include/covest_amd.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/covest_amd.h"
#include "device_model.h"
#include "direct_point.h"
#include "kernels.h"

namespace covest {

// the message covest_last_error returns (thread-local), and the status codes of a failed HIP call
int set_error(int code, const std::string &msg);
inline int fail(int code, const std::string &msg) { return set_error(code, msg); }
int fail_hip(hipError_t e, const char *what);

#define HIP_TRY(expr)                                \
    do {                                             \
        hipError_t e__ = (expr);                     \
        if (e__ != hipSuccess)                       \
            return fail_hip(e__, #expr);             \
    } while (0)

// Small device buffers that a handle lets go are KEPT for the next handle of the process (round 4): hipMalloc costs
// 20-50 us and hipFree waits for the device, and a search that creates a model handle and a grid handle pays four of each
// -- a quarter of a millisecond of a 1.1 ms time-to-argmin.  Per device, buffers of up to 8 MB, 64 MB at most; a buffer
// is given only by an owner whose work on it is done (the destroy entry points wait for the device first), and taken by
// best fit.  host_common.cpp.
bool dev_cache_take(size_t bytes, void **ptr, size_t *cap, int *device);
bool dev_cache_give(void *ptr, size_t cap, int device);
// The cache protects itself (round 5): a buffer is handed on only once nothing can still be working on it.
// An owner that has just waited for the device says so with a DeviceIdleScope around its releases (the destroy, clear
// and grow paths); every other release -- a move-assignment, a destructor on an error path -- waits for the buffer's
// device itself before it gives the buffer away, as the hipFree it replaces did.
struct DeviceIdleScope {
    DeviceIdleScope();
    ~DeviceIdleScope();
    static bool active();
};
// ... and the page-locked, device-mapped block (kPinnedBlockBytes) a grid handle's arg-min writes to: the winner in the
// first 64 bytes, the records of the selection scan (kernels.h ScanRecords) behind them
constexpr size_t kPinnedBlockBytes = 2048;
void *pinned_block_take();
void pinned_block_give(void *p);

// A device allocation that grows on demand and is released with its owner.
struct DevBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    int device = -1; // the device the allocation lives on (the cache hands it to that device's handles only)
    DevBuf() = default;
    ~DevBuf() { release(); } // (round 4: an owner that goes away -- or an error path that returns -- frees what it holds)
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : ptr(o.ptr), cap(o.cap), device(o.device) { o.ptr = nullptr, o.cap = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) {
            release();
            ptr = o.ptr, cap = o.cap, device = o.device;
            o.ptr = nullptr, o.cap = 0;
        }
        return *this;
    }
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        release(); // (growing: whatever still works on the old buffer finishes first -- release waits, as hipFree did)
        if (dev_cache_take(bytes, &ptr, &cap, &device))
            return hipSuccess;
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e == hipSuccess) {
            cap = bytes;
            if (hipGetDevice(&device) != hipSuccess)
                device = -1;
        }
        return e;
    }
    void release()
    {
        if (ptr) {
            bool kept = false;
            if (cap <= ((size_t)8 << 20) && device >= 0) { // (what the cache takes at all: host_common.cpp)
                if (!DeviceIdleScope::active()) {
                    int cur = -1;
                    const bool other = hipGetDevice(&cur) == hipSuccess && cur != device;
                    if (other)
                        (void)hipSetDevice(device);
                    (void)hipDeviceSynchronize();
                    if (other)
                        (void)hipSetDevice(cur);
                }
                kept = dev_cache_give(ptr, cap, device);
            }
            if (!kept)
                (void)hipFree(ptr);
        }
        ptr = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

// Page-locked host memory that is kept from call to call: what a call stages for the device goes through here.  (A
// pageable source makes hipMemcpy pin and unpin it, or bounce it, per call -- from a few hundred KB on that is most
// of a point-list evaluation's time, and how much depends on the machine: 64 points took 214 us on one box of the pool
// and 430 on another, 128 points 0.4 and 27 ms.)
struct HostBuf {
    void *ptr = nullptr;
    size_t cap = 0;
    unsigned flags = hipHostMallocPortable | hipHostMallocMapped; // (any device of the process may copy from it -- or read
                                                                  // and write it in place: small point lists, abi_model.cpp)
    HostBuf() = default;
    ~HostBuf() { release(); }
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    HostBuf(HostBuf &&o) noexcept : ptr(o.ptr), cap(o.cap), flags(o.flags) { o.ptr = nullptr, o.cap = 0; }
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap)
            return hipSuccess;
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 2, 1 << 16);
        hipError_t e = hipHostMalloc(&ptr, want, flags);
        if (e == hipSuccess)
            cap = want;
        return e;
    }
    void release()
    {
        if (ptr)
            (void)hipHostFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return static_cast<T *>(ptr); }
};

// ONE page-locked staging buffer for the uploads that handles make when they are created or re-configured (bins, tile
// tables, axes, plans: tens to hundreds of KB, each copy blocking): kept for the life of the process, so that a
// handle's creation pays neither a pageable copy nor a page-locked allocation of its own.
struct SharedStage {
    std::mutex mu;
    HostBuf buf;
};
SharedStage &shared_stage();


// Every entry point works on ITS handle's device and leaves the calling thread's current device as it found
// it: the caller (torch, another library, a rank bound to another GPU) never sees its device change underneath.
class DeviceGuard {
  public:
    explicit DeviceGuard(int device)
    {
        // (HIP keeps the last error of the thread until somebody reads it: whatever an earlier, unrelated call left
        // behind must not be taken for a failure of the launches this entry point is about to make)
        (void)hipGetLastError();
        had_prev_ = hipGetDevice(&prev_) == hipSuccess;
        if (had_prev_ && prev_ == device)
            return; // nothing to switch, nothing to restore
        const hipError_t e = hipSetDevice(device);
        if (e != hipSuccess)
            status_ = fail_hip(e, "hipSetDevice");
        else
            switched_ = true;
    }
    ~DeviceGuard()
    {
        if (switched_ && had_prev_)
            (void)hipSetDevice(prev_);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
    int status() const { return status_; }

  private:
    int prev_ = 0;
    bool had_prev_ = false, switched_ = false;
    int status_ = COVEST_OK;
};


int resolve_device(int device, const char *who, int *out);

} // namespace covest

using namespace covest; // (an internal header of the library's own host files; the handle types are global: the C ABI names them)

struct covest_model {
    int device = 0;
    int n_par = 2;
    DevModel dm{};       // bins = the evaluated view
    BinView all_bins{};  // every key, in dict order (compute_probabilities); uploaded on first use
    std::vector<double> host_all_key, host_all_lgam, host_all_cnt;
    bool all_bins_ready = false;
    int64_t n_keys = 0;
    int hist_max = 0;    // max(self.hist)
    int64_t key_max = 0; // the largest key the reference evaluates a pmf term for (0 if none is positive)
    // work accounting of K-factored over the item table (tiles.h): rows that are contracted (32 per item: a sum item
    // stands for up to 1024 keys) and keys that take a log
    double rows_contracted = 0.0, keys_logged = 0.0;
    double low_tile_share = 0.0; // share of the tile table's tiles that start at a key <= kLowKeyTile (plan_factored.cpp)
    bool tail_is_zero = true;
    double threshold = 0.0;
    bool has_threshold = true;
    // device storage of the two bin views
    DevBuf bins_eval, bins_all;
    // tile table of the pmf recurrence (fast kernels); has_tiles == false -> direct kernel only
    DevBuf tiles_buf;
    TileView tv{};
    bool has_tiles = false;
    // scratch for covest_eval_points / covest_probabilities
    DevBuf ws_params, ws_t, ws_out, ws_p, ws_plan, ws_plan2, ws_partial, ws_items;
    HostBuf ws_stage; // staging of a point list's tables (build_list_plan)
    HostBuf ws_result; // page-locked, device-mapped: what a point-list launch leaves for the host (list mode 1's parts)
    DevBuf ws_sub_index, ws_sub_word, ws_sub_ctl; // the queue of handed-back points of a point-list launch (direct_point.h)
    std::mutex lock;
};

struct covest_grid {
    covest_model *model = nullptr;
    int64_t len[kMaxParams] = {1, 1, 1, 1, 1};
    int64_t flat_begin = 0, flat_end = 0;
    PointSource src{};
    // one allocation (grown on demand, kept across covest_grid_reset) behind the fixed-purpose views below
    DevBuf arena, plan_buf;
    struct View {
        void *ptr = nullptr;
        template <class T> T *as() const { return static_cast<T *>(ptr); }
        void release() { ptr = nullptr; }
    };
    View axes, t_table, ll, sub_index, sub_word, sub_ctl, partial_val, partial_idx, result;
    FactoredPlan plan{};        // K-factored work description (repeats model, dense grid): the weight vectors whose
                                //   threshold_o fits a workgroup's lanes (build_factored_plan)
    bool has_plan = false;
    bool has_short_part = false;
    // the weight vectors beyond that: one part per chunk of copy numbers, p_j summed in HBM, logs by ll_finish_dense
    struct Part {
        DevBuf buf;
        FactoredPlan plan{};
    };
    std::vector<Part> long_parts;
    int32_t n_long_tiles = 0;
    std::vector<int32_t> long_q_orig_host;
    DevBuf long_q_orig, long_partial;
    int t_max = 2; // largest threshold_o of the (q1, q2, q) product
    double q_sum_t_minus_1 = 0.0; // sum over the Q weight vectors of (threshold_o - 1)
    double contract_flops_per_row = 0.0; // K-factored: useful flops of the contraction per row (build_factored_plan)
    double sum_t_minus_1 = 0.0; // sum over the block's points of (threshold_o - 1)
    const char *last_kernel = "none";
    int last_kernel_id = 0;
    hipStream_t last_stream = nullptr;
    ArgminResult *result_host = nullptr; // page-locked mirror of `result` (+ the scan's records: scan_host())
    ScanRecords *scan_host() const { return reinterpret_cast<ScanRecords *>(reinterpret_cast<char *>(result_host) + 64); }
    bool scan_valid = false; // the last evaluation left the records of the selection scan (covest_grid_eval_scan)
    bool counter_clean = false; // the hand-back queue's counter (sub_ctl) is known to be 0 where it lies
    // covest_grid_reset stages what it uploads in page-locked memory OF THE HANDLE and copies asynchronously (the null
    // stream; an evaluation on another stream waits for upload_ev): the kernels queue up behind the copies and an
    // optimize_grid iteration waits for the device once, when it reads its result.  The staging memory of one reset
    // stays untouched until the next (stage_off only grows; a buffer that must grow waits for the copies first).
    HostBuf stage;
    std::vector<HostBuf> stage_retired; // blocks the staging outgrew during a reset: what they hold may still be read
                                        // (copies in flight, tables read in place) -- freed by the next reset
    size_t stage_off = 0;
    bool async_uploads = false;
    bool upload_pending = false;
    hipEvent_t upload_ev = nullptr;
    bool evaluated = false;
    bool configured = false; // false while (and after) a covest_grid_reset failed half way: the views may dangle
    // optional hipEvent bracketing of the likelihood kernel
    bool profiling = false;
    std::vector<hipEvent_t> ev_begin, ev_end;
    size_t ev_used = 0;
};


struct covest_kmer {
    int device = 0;
    int k = 20;
    int canonical = 0;
    int wide = 0;          // 0: k <= 31, `table`; else the words per key of `wtable` (kmer_wide.hip)
    KmerWideTable wtable{};
    KmerTable table{};
    DevBuf slots, flag, stats, hist, ws_bases, ws_offsets;
    // the partitioned path (kmer_bulk.hip): its buffers, kept from call to call, and what it found
    bool bulk = false; // the counter holds the result of covest_kmer_count_reads_device (until covest_kmer_clear)
    DevBuf bulk_sampled, bulk_cursor, bulk_fill, bulk_tile_reads, bulk_later, bulk_partial, bulk_recs, bulk_ovf, bulk_ctl, bulk_hist, bulk_big;
    unsigned long long bulk_stats[4] = {0, 0, 0, 0};
    int64_t bulk_info[5] = {0, 0, 0, 0, 0}; // buckets, m, sample, records there was room for, records that found none
    unsigned bulk_later_n = 0;              // buckets a workgroup (not a wave) counted
    unsigned long long bulk_to_table_n = 0; // buckets counted through the table in HBM
    bool bulk_table_used = false;           // ... and whether the table holds anything of the result
    int64_t bulk_mem_limit = 0; // covest_kmer_memory_limit: bytes the buckets' records may take (0: what the device has free)
    hipEvent_t bulk_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // start, placed, scattered, counted, done
    float bulk_ms[4] = {0, 0, 0, 0};
    std::mutex lock;
};

constexpr unsigned long long kBulkHistLen = 1ull << 20; // dense count-of-counts bins of the partitioned path
constexpr unsigned long long kBulkBigCap = 4096;        // counts beyond them, listed one by one

namespace covest {

// ---- host_common.cpp
int threshold_o_host(double q1, double q2, double q, double thr, bool has_thr, int hist_max);
void threshold_table(const covest_model *m, const double *a1, int64_t n1, const double *a2, int64_t n2, const double *a3,
                     int64_t n3, int32_t *out); // threshold_o over the product of three (q1, q2, q) axes, last fastest
double clamp_one(const DevModel &dm, int d, double v);
int threshold_for_point(const covest_model *m, const double *par);
void lgamma_ensure(int64_t j_max);
double lgamma_at(int64_t j); // (after lgamma_ensure(j) or larger)
double lgamma_of_factorial(int64_t j);

// ---- tiles_host.cpp
constexpr int64_t kListModeMaxPoints = 4096; // longer point lists are throughput work: K-direct
constexpr int kGapFill = 12;
constexpr int kMaxFastKey = 16384;

struct HostBin {
    int key;
    double cnt;
    int32_t index; // in the evaluated bin view (DevModel::bins)
};
int upload_bins(DevBuf &buf, BinView &view, const std::vector<double> &key, const std::vector<double> &lgam,
                const std::vector<double> &cnt);
int build_tiles(covest_model *m, std::vector<HostBin> bins);
double clamp_for(const covest_model *m, int t_max); // p_clamp of direct_point.h for a launch whose largest threshold_o is t_max

// ---- abi_grid.cpp: staging of a grid handle's uploads (see covest_grid::stage)
struct StageSlot {
    char *ptr = nullptr;
    std::unique_lock<std::mutex> shared_hold; // held while the process-wide staging buffer is in use (blocking path)
};
int grid_stage_begin(covest_grid *g, size_t bytes, StageSlot &slot);
int grid_stage_commit(covest_grid *g, StageSlot &slot, void *dst, size_t bytes);

// ---- plan_factored.cpp
double copy_number_weight_host(double q1, double q2, double q, int o);
int build_factored_plan(covest_grid *g, const double *const *axes, const int64_t *axis_len,
                        const std::vector<int32_t> &t_table);
int build_list_plan(covest_model *m, int64_t n, const double *params, const std::vector<int32_t> &t_list,
                    const std::vector<int32_t> *o_base_list, DevBuf &buf, FactoredPlan &pl, bool in_place = false);

// ---- abi_model.cpp: kernel dispatch shared with the grid entry points
int resolve_kernel(const covest_model *m, int32_t kernel, const covest_grid *g);
SubList sub_list_of(const covest_model *m, int t_max, void *index, void *word, void *ctl);
hipError_t launch_ll(const covest_model *m, int kernel, const PointSource &src, int64_t n, double *out, const SubList &sub,
                     hipStream_t st, const char **name, const covest_grid *g = nullptr);

} // namespace covest

// groupby_handle.hpp -- the opaque aqg_groupby handle (shared by groupby.hip and postproc.hip)
#pragma once
#include <cstddef>
#include <cstdint>

struct aqg_ctx;
constexpr int MAXKEYS = 8, MAXACC = 8, MAXAGG = 8;

struct aqg_groupby {
    aqg_ctx* ctx = nullptr;
    uint32_t n = 0, ngroups = 0;
    int nkeys = 0;
    int key_dt[MAXKEYS] = {0};
    int key_esz[MAXKEYS] = {0};                 // bytes per key element when it is not aqg_dtype_size(key_dt) (merged tables of typed keys: dates 4, times 8, timestamps 12)
    bool has_counts = false, has_reversemap = false;
    // device buffers owned by the handle (grow-only)
    void* keys_out[MAXKEYS] = {nullptr};
    uint32_t* first_rows = nullptr;
    uint32_t* counts = nullptr;
    uint32_t* reversemap = nullptr;
    void* results[MAXAGG] = {nullptr};
    int nagg = 0;
    int res_dt[MAXAGG] = {0};
    // allocated BYTES of every group-sized buffer, each tracked by itself: a reused handle may gain a key column (or come from
    // the small merge path, which keeps one) while others are already large
    size_t cap_keys[MAXKEYS] = {0}, cap_first = 0, cap_counts = 0, cap_rows = 0, cap_results[MAXAGG] = {0};
    uint32_t hint_used = 0;
    uint32_t plan_bits = 0;                     // AQG_PLAN_*: the plan the last call through this handle took (diagnostic: aqg_groupby_plan)
    // aqg_groupby_agg_sharded: rows of a sharded table are numbered globally (64 bits); the 32-bit first rows are not filled
    int64_t* first_rows64 = nullptr;
    size_t cap_first64 = 0;
    bool sharded = false;
    // key columns that are not plain integers (dates, times, 128-bit integers, floating columns): grouped through NORMALISED integer
    // columns held here; their key values are fetched from the caller's column through the first rows (aqg_groupby_keys)
    int nuser = 0;                              // 0: every key column is a plain integer column (keys_out[k] is key k)
    int user_dt[MAXKEYS] = {0};
    const void* user_col[MAXKEYS] = {nullptr};
    int user_norm[MAXKEYS] = {0};               // index of the plain column among the normalised ones, or -1: fetch through first rows
    void* norm_buf[2 * MAXKEYS] = {nullptr};
    size_t cap_norm[2 * MAXKEYS] = {0};
    bool build_assigned = false;                // the last build took a partition plan: reversemap and counts are already written
    bool no_lookup_build = false;               // a key outside the sampled domain met the look-up build: the routed form from now on
    bool no_sorted_tail = false;                // the ordering tail met a partition outside its plan: the bitmap tail from now on
    bool no_pack = false;                       // a value column did not keep to the sampled range of its field in the key word: unpacked planes from now on
    bool no_wide_part = false;                  // a wide-tuple partition overflowed its LDS capacity (a dominant tuple, or twice by chance): HBM table from now on
    uint32_t wide_rows = 0;                     // rows one partition of the last wide-tuple plan could hold (packed tuples: more than the unpacked plan's)
    uint32_t wide_seed = 0;                     // seed of the partition hash: bumped once when a partition overflowed by a little (chance, not a dominant tuple)
    bool dense_exact = false;         // a sampled key range missed values once: take exact ranges from now on
    // sampled key ranges of the last dense plan made through this handle: a call over the same columns takes them without the
    // sampling pass and its host round trip (the kernels verify every row against the ranges anyway; a miss drops the cache)
    bool range_valid = false; int range_nkeys = 0; uint32_t range_n = 0;
    const void* range_col[MAXKEYS] = {}; int range_dt[MAXKEYS] = {}; long long range_min[MAXKEYS] = {}, range_max[MAXKEYS] = {};
    aqg_groupby* scratch = nullptr;   // reusable handle for aqg_grouped_reduce
    aqg_groupby* scratch2 = nullptr;  // second one (aqg_grouped_corr: two passes whose results are read together)
    // the flat row-list layout of a build (segscan.hip): group g owns positions [flat_off[g], flat_off[g+1]) -- ht_postproc's offsets --
    // and `flat_heads` is a bitmap over positions, bit p set = a group starts at p (bit n is set too: the end).  Made on first use.
    uint32_t* flat_off = nullptr; uint32_t* flat_heads = nullptr; uint32_t* flat_short = nullptr; uint32_t* flat_gid = nullptr;
    size_t cap_flat_off = 0, cap_flat_heads = 0, cap_flat_short = 0, cap_flat_gid = 0;
    bool flat_valid = false, flat_gid_valid = false;
    uint32_t flat_short_w = 0;        // flat_short marks the heads of groups with at most this many rows (ratiow's degenerate window); 0: not made
    // aqg_groupby_merge_packed: the concatenated shard tables (keys / values), owned by the merged handle
    void* xkeys = nullptr; void* xvals = nullptr;
    size_t cap_xkeys = 0, cap_xvals = 0;
};


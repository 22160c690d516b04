// dense.hpp -- direct-indexed group tables over dense key domains (dense.hip)
#pragma once
#include "groupby_dev.hpp"

constexpr size_t DENSE_LDS_BYTES = 150 * 1024;   // one workgroup's table (gfx950: 160 KB of LDS per workgroup)
constexpr uint32_t DENSE_MAX_PASSES = 4;

struct DenseSpec {
    long long kmin[MAXKEYS];     // minimum of every key column
    uint32_t mult[MAXKEYS];      // mixed-radix weight: product of the ranges of the columns before it
    uint32_t range[MAXKEYS];     // max_j - min_j + 1
    int sampled;                 // the ranges come from a sample of the rows: kernels verify every row and flag a miss
    uint32_t D;                  // domain size = product of the ranges
    uint32_t per_pass, npass;    // idx in [p * per_pass, (p + 1) * per_pass) belongs to pass p
};

int aqg_key_ranges(aqg_ctx* ctx, const KeySpec& ks, uint32_t n, long long* mins, long long* maxs, bool* ok, uint32_t total = 0);   // over the first n rows; total != 0: over n rows spread evenly (1024 blocks) over a column of `total` rows
size_t aqg_dense_slot_bytes(const AccSpec& as, int need_count);
bool aqg_dense_plan(const KeySpec& ks, const long long* mins, const long long* maxs, const AccSpec& as, int need_count, DenseSpec* ds);
int aqg_dense_assign(aqg_ctx* ctx, const KeySpec& ks, const DenseSpec& ds, const uint32_t* slot_gid, uint32_t n, uint32_t G, uint32_t* reversemap, uint32_t* counts);
int aqg_dense_aggregate(aqg_ctx* ctx, const KeySpec& ks, const DenseSpec& ds, const AccSpec& as, uint32_t n, int need_count, GTable gt);

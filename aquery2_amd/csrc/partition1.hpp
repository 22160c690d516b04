// partition1.hpp -- host entry points of the partition plans of partition1.hip, called by run_agg (groupby.hip)
#pragma once
#include "groupby_dev.hpp"

constexpr uint32_t AQG_P1_MAXBINS = 3584;    // one level: scatter LDS = 128 KB of staging + 8 B per bin
constexpr uint32_t AQG_P2_MAXPARTS = 4096;   // two levels: 64 x 64 bins
// partitions needed for `hint` expected groups (mean + 5 sigma of a partition's groups fit its LDS tables); 0: none
enum : int { AQG_P1_LAYOUT_DENSE_IDS = 0, AQG_P1_LAYOUT_SLOT = 1 };   // where a partition's accumulators sit in LDS (p1_agg_kernel / p1_agg_slot_kernel)
uint32_t aqg_partition_parts(int ksz, const AccSpec& as, int need_count, uint32_t hint, int* layout = nullptr);
// where a partition plan left the rows {key word, row id} and which records every partition wrote: what the build's id pass needs
struct PartRows { const void* keys; const uint32_t* rows; const uint32_t* pstart; uint32_t pstride, nparts, ntotal; int ksz; uint32_t* part_base; uint32_t cap; bool valid; };
size_t aqg_partition1_ws_bytes(const aqg_ctx* ctx, const KeySpec& ks, uint32_t n, const AccSpec& as, uint32_t nbins);
int aqg_partition1_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t nbins, int need_count, GTable out, uint32_t out_cap, PartRows* pr = nullptr, int layout = 0, int* ranged = nullptr);   // *ranged: in, packing / range partitions allowed; out, bit 0 value columns inside the key word, bit 1 range partitions
size_t aqg_partition2_ws_bytes(const aqg_ctx* ctx, const KeySpec& ks, uint32_t n, const AccSpec& as, uint32_t parts);
int aqg_partition2_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t parts, int need_count, GTable out, uint32_t out_cap, PartRows* pr = nullptr, int* pack = nullptr, int layout = 0);   // *pack: in, packing allowed; out, bit 0 value columns travelled inside the key word, bit 1 range partitions
// the build's id pass over the partitioned rows: reversemap[row] = dense id of the row's key (slot_gid: record -> dense id)
int aqg_partition_assign(aqg_ctx* ctx, const PartRows& pr, GTable gt, const uint32_t* slot_gid, uint32_t* reversemap);
size_t aqg_partition_assign_ws_bytes(uint32_t n);
// tuples wider than 8 bytes: hash-partitioned in up to three levels, every partition grouped inside LDS (sized by ROWS)
bool aqg_partitionw_applies(const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t hint);
uint32_t aqg_partitionw_rows(const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t hint);      // rows one partition may hold
size_t aqg_partitionw_ws_bytes(const aqg_ctx* ctx, const KeySpec& ks, uint32_t n, const AccSpec& as, uint32_t hint);
int aqg_partitionw_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, int need_count, GTable out, uint32_t out_cap, uint32_t seed, uint32_t hint, int* pack = nullptr, uint32_t* rows_out = nullptr, bool may_defer = false);   // *pack: in, key packing allowed; out, the key columns travelled packed; *rows_out: rows one partition may hold
// ordering a huge group table (more than ~1.6e7 groups): the record planes partitioned by first row with the tile scatter (order-preserving
// bins, up to three levels) until a partition's row interval fits LDS; sorted_emit_kernel (groupby.hip) ranks and emits from there
struct SortedPlan { uint32_t levels, bits, M, cap; size_t lds; };    // cap: rows (so records at most) of one partition
struct SortedParts {
    const uint32_t* first; const uint32_t* count; const uint64_t* key; const uint64_t* acc[MAXACC];
    const uint32_t* pstart;                                           // [nparts + 1] first group id of every partition
    uint32_t nparts, M, cap; size_t lds;                              // partition p: first rows in [ceil(p 2^32 / M), ceil((p + 1) 2^32 / M))
};
bool aqg_sorted_tail_plan(uint32_t n_rows, int nacc, bool wide, SortedPlan* out);
size_t aqg_sorted_tail_ws_bytes(uint32_t gcap, uint32_t n_rows, int nacc, bool wide);
int aqg_sorted_tail(aqg_ctx* ctx, const GTable& gt, uint32_t G, uint32_t n_rows, int nacc, bool wide, SortedParts* out);
// grouped reductions keyed by DENSE group ids with known group sizes (aqg_grouped_reduce beyond the LDS tables): the rows {id, value}
// partitioned on the id itself (order-preserving bins, no histogram passes: the cursors come from the offsets), direct-indexed LDS
// accumulators, results written in id order.  AQG_ERR_DTYPE: not served here (the caller takes the hashed plans)
int aqg_gid_reduce(aqg_ctx* ctx, const uint32_t* gid, const uint32_t* offsets, const uint32_t* counts, uint32_t n, uint32_t G, int op, int t, const void* x, void* out_dev);

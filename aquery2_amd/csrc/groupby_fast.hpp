// groupby_fast.hpp -- the fast LDS group-by kernel (groupby_fast.hip) as seen by the planner in groupby.hip
#pragma once
#include "groupby_dev.hpp"

// vkind: 0 int32, 1 uint32, 2 float, 3 int64, 4 uint64, 5 double, 6 int8, 7 uint8 / bool, 8 int16, 9 uint16; kind: ACC_*; square: accumulate x*x; part: 1 / 2 = the low /
// high 32-bit half of an int64 sum
struct FastVals { const void* col[4]; int vkind[4]; int kind[4]; int square[4]; int part[4]; };
constexpr uint32_t OCCUPIED = 0xFFFFFFFEu;   // first_row mark: "group exists, first row not yet known"

// one pass over the rows: keys (+ keys_hi: the second 4-byte key column; k64 with keys_hi == nullptr: `keys` is one 8-byte
// column), `nacc` accumulators described by fv (v8: some value column is not 4 bytes wide), per-workgroup LDS tables of lcap slots
// merged into gt.  Records the kernel's duration in the context's kernel timer.
int aqg_fast_aggregate(aqg_ctx* ctx, const uint32_t* keys, const uint32_t* keys_hi, bool k64, bool v8, int nacc, bool need_count,
                       const FastVals& fv, GTable gt, uint32_t n, uint32_t lcap, size_t lds, unsigned grid, unsigned block);

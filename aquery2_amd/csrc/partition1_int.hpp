// partition1_int.hpp -- what partition1.hip (histograms, scatters, the plans' hosts) and partition1_agg.hip (the per-partition aggregation
// kernels and their launchers) share.  Two translation units only because the aggregation kernels are ~90 large instantiations: hipcc takes
// three minutes for them, and the build runs them beside the scatters instead of behind them.
#pragma once
#include <cmath>
#include "groupby_dev.hpp"
#include "partition1.hpp"

namespace {

constexpr int SB = 1024;          // threads per workgroup
template <bool K64> struct KeyWord { using type = uint32_t; };
template <> struct KeyWord<true> { using type = uint64_t; };
template <bool K64> using key_t_ = typename KeyWord<K64>::type;
template <bool K64> __device__ inline uint32_t key_hash(key_t_<K64> k) { return lds_h1<!K64>((uint64_t)k); }
template <bool K64> __device__ inline key_t_<K64> empty_key() { if constexpr (K64) return EMPTY64; else return EMPTY32; }

__device__ inline uint64_t val_operand_bits(int dt, uint64_t bits, int kind, int square, int part) {
    switch (dt) {
    case AQG_INT8: return val_operand_t((int8_t)bits, kind, square);
    case AQG_INT16: return val_operand_t((int16_t)bits, kind, square);
    case AQG_INT32: return val_operand_t((int32_t)bits, kind, square);
    case AQG_INT64: return val_operand_t((int64_t)bits, kind, square, part);
    case AQG_UINT8: case AQG_BOOL: return val_operand_t((uint8_t)bits, kind, square);
    case AQG_UINT16: return val_operand_t((uint16_t)bits, kind, square);
    case AQG_UINT32: return val_operand_t((uint32_t)bits, kind, square);
    case AQG_UINT64: return val_operand_t((uint64_t)bits, kind, square, part);
    case AQG_FLOAT: return val_operand_t(__uint_as_float((uint32_t)bits), kind, square);
    default: return val_operand_t(__builtin_bit_cast(double, bits), kind, square);
    }
}
// What an accumulator does with a row, decided once per call on the host: the common (kind, dtype) pairs get straight-line code,
// everything else (squares, halves of 8-byte integers, 1- / 2-byte and 8-byte integer inputs) the generic operand switch.  The row
// loop was VALU- and branch-bound with that switch evaluated per row and accumulator (h2o Q5, 1e9 rows: 9.1 ms for 20 GB).
enum : int { OPC_ADDI_I32 = 0, OPC_ADDI_U32, OPC_ADDF_F32, OPC_ADDF_F64, OPC_MIN_I32, OPC_MAX_I32, OPC_MIN_U32, OPC_MAX_U32, OPC_MIN_F32, OPC_MAX_F32, OPC_GENERIC };
struct AggOps { int opc[MAXACC]; };

constexpr size_t AGG_LDS = 150 * 1024;
constexpr uint32_t LF1000 = 500;    // load factor of the key table

} // namespace

// the distinct value columns of the accumulators (an accumulator over the row index has none)
struct ValCols { int n; const void* col[MAXACC]; int dt[MAXACC]; int of_acc[MAXACC]; };
// which value columns travel inside the key word (PackSpec, partition1.hip) and the sampled range of the key column
struct PackPlan { int n; const void* col[2]; uint32_t min[2], shift[2], fmask[2]; uint32_t kmax, kclear; bool have_range, exact; long long key_lo, key_hi; };
static inline int pack_field_of(const PackPlan& pp, const void* col) { for (int f = 0; f < pp.n; ++f) if (pp.col[f] == col) return f; return -1; }
// range partitions over a dense key domain (p1_agg_direct_kernel)
struct RangePlan { bool on; uint32_t kmin, D, P, M, W; };

static inline size_t part_val_bytes(int dt) { return aqg_dtype_size(dt) <= 4 ? 4 : 8; }   // narrow values travel widened to one dword
uint32_t p1_direct_capacity(const AccSpec& as, int need_count);
// aggregate the partitions [pstart[p * pstride], pstart[(p + 1) * pstride]) (the last one ends at n) of the partitioned planes (partition1_agg.hip)
int p1_launch_agg(aqg_ctx* ctx, int ksz, const AccSpec& as, const ValCols& vc, const void* pkeys, const void* prows, void* const* pvals,
                  const uint32_t* pstart, uint32_t pstride, uint32_t nparts, uint32_t n, int need_count, GTable out, uint32_t out_cap, PartRows* pr = nullptr,
                  const PackPlan* pp = nullptr, int layout = AQG_P1_LAYOUT_DENSE_IDS);
int p1_launch_agg_direct(aqg_ctx* ctx, const AccSpec& as, const ValCols& vc, const void* pkeys, const void* prows, void* const* pvals,
                         const uint32_t* pstart, uint32_t pstride, uint32_t n, int need_count, GTable out, uint32_t out_cap, const PackPlan* pp, const RangePlan& rp);

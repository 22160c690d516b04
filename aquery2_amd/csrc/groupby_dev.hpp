// groupby_dev.hpp -- device-side vocabulary shared by groupby.hip and partition.hip: key packing, accumulator operands,
// the open-addressing table record layout and its find/insert primitives.
#pragma once
#include "aqg_internal.hpp"
#include "dev_common.hpp"
#include "groupby_handle.hpp"

namespace aqgdev {

constexpr uint64_t EMPTY64 = ~0ull;
constexpr uint32_t EMPTY32 = 0x80000000u;   // nullval<int> (server/types.h:458) doubles as the LDS empty mark
constexpr uint32_t NOROW = 0xFFFFFFFFu;
constexpr uint32_t FAIL = 0xFFFFFFFFu;

enum : int { ACC_ADD_I = 0, ACC_ADD_F = 1, ACC_MIN = 2, ACC_MAX = 3 };
enum : int { VC_I = 0, VC_U = 1, VC_F = 2 };   // value class of a column: signed / unsigned / floating

// wide != 0: the tuple does not fit 64 bits; the table then stores a REPRESENTATIVE ROW per slot and compares key columns
struct KeySpec {
    int nkeys; int dt[MAXKEYS]; const void* col[MAXKEYS]; int shift[MAXKEYS]; int total_bytes; int wide;
    // the exact value range of a single key column when the caller knows it (the dense group ids of a build: 0 .. G-1).  The plans that
    // otherwise sample the first 2^20 rows take it as it is: ids are numbered by first occurrence, so a sample of the first rows sees only
    // the low ones and every such call paid a failed attempt
    int range_known; long long range_lo, range_hi;
};
// square: accumulate x*x (in the promoted type).  part: 0 whole value; 1 / 2 = low / high 32 bits of an 8-byte
// integer, so that sums of 8-byte integers stay exact (the two 64-bit accumulators cannot overflow for n < 2^32)
struct AccSpec { int nacc; int kind[MAXACC]; int dt[MAXACC]; const void* col[MAXACC]; int square[MAXACC]; int part[MAXACC]; };
// Table storage is addressed through per-field bases and strides so that two layouts share all kernels:
//   records (AoS)  {key u64 | first_row u32 | count u32 | acc u64 x nacc} in one power-of-two sized record: every atomic of
//                  a row lands in ONE cache line -- best when the table is far larger than the caches (h2o Q3/Q5: -15..25 %)
//   columns (SoA)  separate arrays: the read-mostly key words stay cacheable while the accumulators take the atomics --
//                  best for small hot tables (h2o Q2, 1e4 groups: 54 ms vs 140 ms per 1e9 rows as records)
struct GTable {
    unsigned char *kb, *fb, *cb, *ab;   // key / first_row / count / accumulator bases
    uint32_t kst, fst, cst, ast;        // byte strides per slot
    uint64_t astep;                     // byte step between accumulators of one slot
    uint32_t cap;                       // power of two; slot `cap` holds the group whose packed key equals EMPTY64
    uint32_t* flags;                    // [0] overflow, [1] number of occupied slots (after collect)
    int has_count;
    __device__ inline uint64_t* key_p(uint32_t s) const { return reinterpret_cast<uint64_t*>(kb + (size_t)s * kst); }
    __device__ inline uint32_t* first_p(uint32_t s) const { return reinterpret_cast<uint32_t*>(fb + (size_t)s * fst); }
    __device__ inline uint32_t* count_p(uint32_t s) const { return reinterpret_cast<uint32_t*>(cb + (size_t)s * cst); }
    __device__ inline uint64_t* acc_p(int a, uint32_t s) const { return reinterpret_cast<uint64_t*>(ab + (size_t)s * ast + (size_t)a * astep); }
};

__host__ __device__ inline int aqg_dtype_size_dev(int dt) {
    switch (dt) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: return 1;
    case AQG_INT16: case AQG_UINT16: return 2;
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: return 4;
    default: return 8;
    }
}
// element `i` of a column of 1 / 2 / 4 / 8-byte integers := the low bytes of `bits`.  Deliberately NOT inlined: with the size switch
// inlined into unpack_kernel's copy loop, hipcc (ROCm 7.2, -O3) left the register holding the row index undefined on the 8-byte
// path and the store of the value column that followed the switch went to a wild address (found by a GPU memory fault in the
// merge of int64-keyed shard tables; the ISA showed `implicit-def $vgpr10_vgpr11` on that path).
__device__ __noinline__ static void store_sized(void* __restrict__ col, size_t i, int size, unsigned long long bits) {
    if (size == 1) static_cast<uint8_t*>(col)[i] = (uint8_t)bits;
    else if (size == 2) static_cast<uint16_t*>(col)[i] = (uint16_t)bits;
    else if (size == 4) static_cast<uint32_t*>(col)[i] = (uint32_t)bits;
    else static_cast<unsigned long long*>(col)[i] = bits;
}
// slot of a 4-byte key in a power-of-two table of 2^bits slots: the TOP bits of a Fibonacci hash.  They depend on every key
// bit (keys that are all multiples of 1024 made the low bits of the product useless: 350 ms instead of 1.4 ms per 1e9 rows),
// and runs of consecutive keys -- dictionary ids -- land evenly spaced (three-distance theorem): 100 consecutive keys never share
// one of 256 slots.
// (A random-looking pre-mix was measured too: it makes every key pattern cost what random collisions cost -- 4-5 ms per 1e9 rows
// at 1000 groups -- while the plain form stays at 1.4-2.9 ms for most strides and 4-7 ms for a few unlucky ones.)
__device__ inline uint32_t fib_slot(uint32_t k, uint32_t bits) { return (k * 0x9E3779B1u) >> (32 - bits); }
__device__ inline uint32_t hash64(uint64_t k) { k *= 0x9E3779B97F4A7C15ull; return (uint32_t)(k >> 32) ^ (uint32_t)k; }

// LDS-table hashing in 32-bit multiplies only (the row loop is VALU-bound: a 64-bit multiply costs four quarter-rate ones).
// h1's top bits pick the pass, a second mix of h1 picks the slot, so the two are independent.
template <bool K32> __device__ inline uint32_t lds_h1(uint64_t key) {
    uint32_t k = (uint32_t)key;
    if constexpr (!K32) k ^= (uint32_t)(key >> 32) * 0x85EBCA6Bu;
    return k * 0x9E3779B1u;
}
__device__ inline uint32_t lds_h2(uint32_t h1) { return (h1 ^ (h1 >> 15)) * 0x2C1B3C6Du; }

// order-preserving maps into uint64 so that MIN/MAX of every class are unsigned integer atomics
__device__ inline uint64_t map_i(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
__device__ inline int64_t unmap_i(uint64_t u) { return (int64_t)(u ^ 0x8000000000000000ull); }
__device__ inline uint64_t map_f(double d) { uint64_t b = __builtin_bit_cast(uint64_t, d); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
__device__ inline double unmap_f(uint64_t u) { uint64_t b = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u; return __builtin_bit_cast(double, b); }

__host__ __device__ inline int vclass(int dt) {
    switch (dt) {
    case AQG_FLOAT: case AQG_DOUBLE: return VC_F;
    case AQG_UINT8: case AQG_UINT16: case AQG_UINT32: case AQG_UINT64: case AQG_BOOL: return VC_U;
    default: return VC_I;
    }
}

// raw bits of element i, zero-extended (key packing)
__device__ inline uint64_t load_bits(int dt, const void* col, size_t i) {
    switch (dt) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: return static_cast<const uint8_t*>(col)[i];
    case AQG_INT16: case AQG_UINT16: return static_cast<const uint16_t*>(col)[i];
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: return static_cast<const uint32_t*>(col)[i];
    default: return static_cast<const uint64_t*>(col)[i];
    }
}
__device__ inline uint64_t pack_key(const KeySpec& ks, size_t i) {
    uint64_t k = load_bits(ks.dt[0], ks.col[0], i);
    for (int j = 1; j < ks.nkeys; ++j) k |= load_bits(ks.dt[j], ks.col[j], i) << ks.shift[j];
    return k;
}

// four consecutive rows at once: one vector load per key column instead of four scalar ones
template <class T> __device__ inline void load_bits4_t(const void* col, size_t base, uint64_t (&o)[4]) {
    pack<T, 4> v = *reinterpret_cast<const pack<T, 4>*>(static_cast<const T*>(col) + base);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = v.v[j];
}
__device__ inline void load_bits4(int dt, const void* col, size_t base, uint64_t (&o)[4]) {
    switch (dt) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: load_bits4_t<uint8_t>(col, base, o); break;
    case AQG_INT16: case AQG_UINT16: load_bits4_t<uint16_t>(col, base, o); break;
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: load_bits4_t<uint32_t>(col, base, o); break;
    default: load_bits4_t<uint64_t>(col, base, o); break;
    }
}
__device__ inline void pack_key4(const KeySpec& ks, size_t base, uint64_t (&key)[4]) {
    load_bits4(ks.dt[0], ks.col[0], base, key);
    for (int j = 1; j < ks.nkeys; ++j) {
        uint64_t b[4];
        load_bits4(ks.dt[j], ks.col[j], base, b);
#pragma unroll
        for (int r = 0; r < 4; ++r) key[r] |= b[r] << ks.shift[j];
    }
}

// value of element i as the 64-bit operand of its accumulator
//   ADD_I: two's complement int64 (unsigned inputs zero-extended), `x*x` in the promoted type if square
//   ADD_F: double bits; MIN/MAX: order-preserving map
template <class T> __device__ inline uint64_t val_operand_t(T v, int kind, int square, int part = 0) {
    if constexpr (sizeof(T) == 8 && std::is_integral_v<T>) {
        if (kind == ACC_ADD_I && part) {
            uint64_t b = square ? (uint64_t)v * (uint64_t)v : (uint64_t)v;
            if (part == 1) return b & 0xFFFFFFFFull;
            if constexpr (std::is_unsigned_v<T>) return b >> 32; else return (uint64_t)((int64_t)b >> 32);
        }
    }
    if constexpr (std::is_floating_point_v<T>) {
        double d = square ? (double)(v * v) : (double)v;
        return kind == ACC_ADD_F ? __builtin_bit_cast(uint64_t, d) : map_f(d);
    } else {
        if (kind == ACC_ADD_I) {
            if (square) {
                using P = decltype(v * v);
                using UP = std::make_unsigned_t<P>;
                P p = (P)((UP)(P)v * (UP)(P)v);
                if constexpr (std::is_unsigned_v<P>) return (uint64_t)p; else return (uint64_t)(int64_t)p;
            }
            if constexpr (std::is_unsigned_v<T>) return (uint64_t)v; else return (uint64_t)(int64_t)v;
        }
        if constexpr (std::is_unsigned_v<T>) return (uint64_t)v; else return map_i((int64_t)v);
    }
}
// dt == AQG_NONE: the operand is the row index itself (arg-max of the row id for FIRST; see aqg_grouped_reduce)
__device__ inline uint64_t val_operand(int dt, const void* col, size_t i, int kind, int square, int part) {
    if (dt == AQG_NONE) return (uint64_t)i;
    switch (dt) {
    case AQG_INT8: return val_operand_t(static_cast<const int8_t*>(col)[i], kind, square);
    case AQG_INT16: return val_operand_t(static_cast<const int16_t*>(col)[i], kind, square);
    case AQG_INT32: return val_operand_t(static_cast<const int32_t*>(col)[i], kind, square);
    case AQG_INT64: return val_operand_t(static_cast<const int64_t*>(col)[i], kind, square, part);
    case AQG_UINT8: case AQG_BOOL: return val_operand_t(static_cast<const uint8_t*>(col)[i], kind, square);
    case AQG_UINT16: return val_operand_t(static_cast<const uint16_t*>(col)[i], kind, square);
    case AQG_UINT32: return val_operand_t(static_cast<const uint32_t*>(col)[i], kind, square);
    case AQG_UINT64: return val_operand_t(static_cast<const uint64_t*>(col)[i], kind, square, part);
    case AQG_FLOAT: return val_operand_t(static_cast<const float*>(col)[i], kind, square);
    default: return val_operand_t(static_cast<const double*>(col)[i], kind, square);
    }
}
// four consecutive rows of a 4-byte column with one 16-byte load
template <class T> __device__ inline void val_operand4_t(const void* col, size_t base, int kind, int square, int part, uint64_t (&o)[4]) {
    pack<T, 4> v = *reinterpret_cast<const pack<T, 4>*>(static_cast<const T*>(col) + base);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = val_operand_t(v.v[j], kind, square, part);
}
__device__ inline void val_operand4(int dt, const void* col, size_t base, int kind, int square, int part, uint64_t (&o)[4]) {
    if (dt == AQG_NONE) { o[0] = base; o[1] = base + 1; o[2] = base + 2; o[3] = base + 3; return; }
    switch (dt) {
    case AQG_INT8: val_operand4_t<int8_t>(col, base, kind, square, part, o); break;
    case AQG_INT16: val_operand4_t<int16_t>(col, base, kind, square, part, o); break;
    case AQG_INT32: val_operand4_t<int32_t>(col, base, kind, square, part, o); break;
    case AQG_INT64: val_operand4_t<int64_t>(col, base, kind, square, part, o); break;
    case AQG_UINT8: case AQG_BOOL: val_operand4_t<uint8_t>(col, base, kind, square, part, o); break;
    case AQG_UINT16: val_operand4_t<uint16_t>(col, base, kind, square, part, o); break;
    case AQG_UINT32: val_operand4_t<uint32_t>(col, base, kind, square, part, o); break;
    case AQG_UINT64: val_operand4_t<uint64_t>(col, base, kind, square, part, o); break;
    case AQG_FLOAT: val_operand4_t<float>(col, base, kind, square, part, o); break;
    default: val_operand4_t<double>(col, base, kind, square, part, o); break;
    }
}

__device__ inline void acc_apply(uint64_t* p, int kind, uint64_t v) {
    switch (kind) {
    case ACC_ADD_I: atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); break;
    case ACC_ADD_F: atomicAdd(reinterpret_cast<double*>(p), __builtin_bit_cast(double, v)); break;
    case ACC_MIN: atomicMin(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); break;
    default: atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v); break;
    }
}
__host__ __device__ inline uint64_t acc_init(int kind) { return kind == ACC_MIN ? ~0ull : 0ull; }

// ---- global table ---------------------------------------------------------------------------
// Slots only ever change EMPTY -> key, so a plain (possibly stale) load is safe: a stale EMPTY is
// corrected by the device-scope compare-and-swap that follows.
// Once some row has found the table full the call is going to be re-planned with a larger table: every later row gives up at
// once instead of walking the whole (full) table first (1e6 distinct keys against a 2048-slot table: 300 ms -> 1 ms).
__device__ inline bool gt_gave_up(const GTable& gt) { return __hip_atomic_load(&gt.flags[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
__device__ inline uint32_t gt_find_or_insert(const GTable& gt, uint64_t key) {
    if (key == EMPTY64) return gt.cap;
    if (gt_gave_up(gt)) return FAIL;
    uint32_t mask = gt.cap - 1, s = hash64(key) & mask;
    for (uint32_t p = 0; p < gt.cap; ++p) {
        uint64_t cur = (*gt.key_p(s));
        if (cur == key) return s;
        if (cur == EMPTY64) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(gt.key_p(s)), EMPTY64, key);
            if (old == EMPTY64 || old == key) return s;
        }
        s = (s + 1) & mask;
    }
    gt.flags[0] = 1;   // table full
    return FAIL;
}
__device__ inline uint32_t gt_find(const GTable& gt, uint64_t key) {
    if (key == EMPTY64) return gt.cap;
    uint32_t mask = gt.cap - 1, s = hash64(key) & mask;
    for (uint32_t p = 0; p < gt.cap; ++p) {
        uint64_t cur = (*gt.key_p(s));
        if (cur == key) return s;
        if (cur == EMPTY64) return FAIL;
        s = (s + 1) & mask;
    }
    return FAIL;
}
// ---- wide tuples: slot word = representative row, equality = column-wise compare against that row ----------
__device__ inline uint32_t hash_wide(const KeySpec& ks, size_t row) {
    uint64_t h = 0x7c5f3e9a1b2d4c6bull;
    for (int j = 0; j < ks.nkeys; ++j) h = (h ^ load_bits(ks.dt[j], ks.col[j], row)) * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(h >> 32) ^ (uint32_t)h;
}
__device__ inline bool rows_equal(const KeySpec& ks, size_t a, size_t b) {
    for (int j = 0; j < ks.nkeys; ++j) if (load_bits(ks.dt[j], ks.col[j], a) != load_bits(ks.dt[j], ks.col[j], b)) return false;
    return true;
}
__device__ inline uint32_t gt_find_or_insert_wide(const GTable& gt, const KeySpec& ks, uint32_t row) {
    if (gt_gave_up(gt)) return FAIL;
    uint32_t mask = gt.cap - 1, s = hash_wide(ks, row) & mask;
    for (uint32_t p = 0; p < gt.cap; ++p) {
        uint64_t cur = (*gt.key_p(s));
        if (cur == EMPTY64) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(gt.key_p(s)), EMPTY64, (unsigned long long)row);
            if (old == EMPTY64) return s;
            cur = old;
        }
        if (rows_equal(ks, (uint32_t)cur, row)) return s;
        s = (s + 1) & mask;
    }
    gt.flags[0] = 1;
    return FAIL;
}
__device__ inline uint32_t gt_find_wide(const GTable& gt, const KeySpec& ks, uint32_t row) {
    uint32_t mask = gt.cap - 1, s = hash_wide(ks, row) & mask;
    for (uint32_t p = 0; p < gt.cap; ++p) {
        uint64_t cur = (*gt.key_p(s));
        if (cur == EMPTY64) return FAIL;
        if (rows_equal(ks, (uint32_t)cur, row)) return s;
        s = (s + 1) & mask;
    }
    return FAIL;
}
__device__ inline void gt_touch_first(const GTable& gt, uint32_t s, uint32_t row) {
    // (*gt.first_p(s)) only decreases: a stale (larger) value just costs one redundant atomic
    if (row < (*gt.first_p(s))) atomicMin(gt.first_p(s), row);
}

} // namespace aqgdev
using namespace aqgdev;

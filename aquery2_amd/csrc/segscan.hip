// segscan.hip -- per-group scans, windows and shifts in ONE launch set for all groups: the device form of the generated loop
//     for g: out[g] = scan(w, col[vecs[g]])          (engine/ast.py:696-790; frozen sample mem_opt.cpp:53-63: avgw(10, sales[vecs[i]], col[i]))
// whose results the generated code lays out in one flat buffer sliced by the group offsets (`buf + g.offsets[i]`).
//
// The FLAT LAYOUT of a grouping: position offsets[g] + i holds element i of group g's row list vecs[g] (ht_postproc order,
// server/hasher.h:181-198: descending row id inside a group).  aqg_grouped_flatten brings a column into that layout with the
// value-carrying radix passes of postproc.hip (no row ids, no gather); a bitmap of group starts (`heads`) then makes every kernel of
// scan.hip segmented:
//   prefix scans (sums / avgs / mins / maxs / vars / stddevs): reduce-then-scan over 2048-position tiles with the carry
//       {value, last group start, groups so far} -- a group start resets the value;
//   windows (sumw / avgw / varw / stddevw, minw / maxw): tile + halo in LDS as in scan.hip, every window clamped at its group's start
//       (no carry between tiles: a window never reaches back further than the halo);
//   shifts (deltas / prev / aggnext / ratiow): neighbour loads guarded by the start bits.
// Integer results are exact; floating sums follow the tile order (tolerance as for the whole-column scans).
// HBM-bound: sizeof(T) + sizeof(out) bytes per row for the scan proper, 12 (one radix pass, <= 256 groups) to 20 x passes for flatten.
#include "aqg_internal.hpp"
#include "dev_common.hpp"
#include "scan_dev.hpp"
#include "groupby_handle.hpp"

namespace {
using namespace aqgscan;

// ---- the carry of a segmented scan -------------------------------------------------------------------------------------------
// v: fold of the values behind the last group start of the range (of the whole range when it has none); s: position + 1 of that
// start (0: none); c: group starts in the range
template <class A> struct SegCarry { A v; uint32_t s; uint32_t c; };
template <class ALG> struct seg_alg {
    using A = SegCarry<typename ALG::A>;
    __device__ static A identity() { A r; r.v = ALG::identity(); r.s = 0; r.c = 0; return r; }
    __device__ static A op(A a, A b) { A r; r.s = b.s ? b.s : a.s; r.c = a.c + b.c; r.v = b.s ? b.v : ALG::op(a.v, b.v); return r; }
};
struct dpair { double s, q; };
template <class T> struct var2_alg {          // running sum and sum of squares in double (vars / stddevs, as vars_kernel of scan.hip)
    using A = dpair;
    __device__ static dpair identity() { return {0.0, 0.0}; }
    __device__ static dpair lift(T v) { const double d = (double)v; return {d, d * d}; }
    __device__ static dpair op(dpair a, dpair b) { return {a.s + b.s, a.q + b.q}; }
};
struct ipair { int64_t s, q; };
// sum and sum of squares of a group for var / stddev (aggregations.h:332-348): exact for integer columns of up to four bytes -- x * x in the
// C++ type of the operands, like the reference's `arr[i] * arr[i]` -- and in double for floating columns (the square in the column's type)
template <class T> struct varred_alg {
    using A = std::conditional_t<std::is_floating_point_v<T>, dpair, ipair>;
    __device__ static A identity() { A r; r.s = 0; r.q = 0; return r; }
    __device__ static A lift(T v) {
        A r;
        if constexpr (std::is_floating_point_v<T>) { r.s = (double)v; r.q = (double)(T)(v * v); }
        else {
            using P = decltype(v * v);
            using UP = std::make_unsigned_t<P>;
            const P sq = (P)((UP)(P)v * (UP)(P)v);
            r.s = (int64_t)v;
            r.q = (int64_t)sq;
        }
        return r;
    }
    __device__ static A op(A a, A b) { A r; r.s = a.s + b.s; r.q = a.q + b.q; return r; }
};
struct none_alg {                              // position-only scans (group index / distance to the group start of every position)
    using A = uint32_t;
    __device__ static uint32_t identity() { return 0; }
    __device__ static uint32_t lift(uint8_t) { return 0; }
    __device__ static uint32_t op(uint32_t a, uint32_t) { return a; }
};

// ---- group starts ----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) heads_kernel(const uint32_t* __restrict__ off, const uint32_t* __restrict__ counts, uint32_t G,
                                                    uint32_t* __restrict__ heads, uint32_t* __restrict__ shorts, uint32_t short_w) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g <= G; g += gridDim.x * blockDim.x) {
        const uint32_t p = off[g];                                       // g == G: the end of the table, a start nobody owns
        if (heads) atomicOr(&heads[p >> 5], 1u << (p & 31));
        if (shorts && g < G && counts[g] <= short_w) atomicOr(&shorts[p >> 5], 1u << (p & 31));
    }
}
__device__ inline bool head_bit(const uint32_t* __restrict__ heads, uint32_t p) { return (heads[p >> 5] >> (p & 31)) & 1u; }
// number of predecessors of position p inside its group, capped at maxd (walks the bitmap backwards; position 0 always starts a group)
__device__ inline uint32_t dist_to_head(const uint32_t* __restrict__ heads, uint32_t p, uint32_t maxd) {
    uint32_t wi = p >> 5;
    const uint32_t b = p & 31;
    uint32_t m = heads[wi] & (0xFFFFFFFFu >> (31 - b));
    if (m) { const uint32_t d = b - (31 - __clz((int)m)); return d < maxd ? d : maxd; }
    uint32_t d = b + 1;
    while (d <= maxd && wi > 0) {
        m = heads[--wi];
        if (m) { d += __clz((int)m); return d < maxd ? d : maxd; }
        d += 32;
    }
    return maxd;
}

// ---- prefix family: K1 tile carries, K2 launch_agg_scan<seg_alg>, K3 results ----------------------------------------------------------
template <class T, class ALG>
__global__ void __launch_bounds__(SB) seg_tile_reduce_kernel(const T* __restrict__ x, uint32_t n, const uint8_t* __restrict__ heads8,
                                                             SegCarry<typename ALG::A>* __restrict__ tile_carry) {
    using SA = seg_alg<ALG>;
    using C = typename SA::A;
    __shared__ C lds_w[8];
    uint32_t base = blockIdx.x * TS + threadIdx.x * IT, cnt;
    T v[IT];
    if constexpr (std::is_same_v<ALG, none_alg>) cnt = base >= n ? 0 : (n - base < (uint32_t)IT ? n - base : IT);
    else load_tile_items(x, n, base, v, cnt);
    const uint32_t hb = cnt ? heads8[base >> 3] : 0;
    C a = SA::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        if ((uint32_t)j < cnt) {
            if ((hb >> j) & 1) { a.v = ALG::identity(); a.s = base + j + 1; ++a.c; }
            if constexpr (!std::is_same_v<ALG, none_alg>) a.v = ALG::op(a.v, ALG::lift(v[j]));
        }
    }
    C total;
    block_scan_excl<SA>(a, lds_w, total);
    if (threadIdx.x == 0) tile_carry[blockIdx.x] = total;
}

enum : int { SW_SUMS = 0, SW_AVGS, SW_MINS, SW_MAXS, SW_MAXP, SW_VARS, SW_STDDEVS, SW_RAW, SW_DIST, SW_GID, SW_RED_SUM, SW_RED_AVG, SW_RED_MIN, SW_RED_MAX, SW_RED_VAR, SW_RED_STDDEV };
template <class T, int WR> struct seg_out {
    using type = std::conditional_t<WR == SW_SUMS || WR == SW_RED_SUM, std::conditional_t<std::is_floating_point_v<T>, double, aqg_i128>,
                 std::conditional_t<WR == SW_AVGS || WR == SW_VARS || WR == SW_STDDEVS || WR == SW_RED_AVG || WR == SW_RED_VAR || WR == SW_RED_STDDEV, double,
                 std::conditional_t<WR == SW_RAW, typename sum_alg<T>::A,
                 std::conditional_t<WR == SW_DIST || WR == SW_GID, uint32_t, T>>>>;
};
template <class T, class ALG, int WR>
__global__ void __launch_bounds__(SB) seg_tile_scan_kernel(const T* __restrict__ x, uint32_t n, const uint8_t* __restrict__ heads8,
                                                           const SegCarry<typename ALG::A>* __restrict__ tile_prefix, void* __restrict__ out) {
    using A = typename ALG::A;
    using SA = seg_alg<ALG>;
    using C = typename SA::A;
    using O = typename seg_out<T, WR>::type;
    constexpr bool RED = WR >= SW_RED_SUM;
    __shared__ C lds_w[8];
    extern __shared__ __align__(16) unsigned char stage_raw[];
    uint32_t base = blockIdx.x * TS + threadIdx.x * IT, cnt;
    T v[IT];
    if constexpr (std::is_same_v<ALG, none_alg>) cnt = base >= n ? 0 : (n - base < (uint32_t)IT ? n - base : IT);
    else load_tile_items(x, n, base, v, cnt);
    const uint32_t hb = cnt ? heads8[base >> 3] : 0;
    uint32_t hnext = 0;                                                  // start bit of the position behind this lane's block
    if constexpr (RED) hnext = cnt ? (heads8[(base >> 3) + 1] & 1u) : 0u;
    C a = SA::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        if ((uint32_t)j < cnt) {
            if ((hb >> j) & 1) { a.v = ALG::identity(); a.s = base + j + 1; ++a.c; }
            if constexpr (!std::is_same_v<ALG, none_alg>) a.v = ALG::op(a.v, ALG::lift(v[j]));
        }
    }
    C total;
    const C excl = block_scan_excl<SA>(a, lds_w, total);
    const C pre = SA::op(tile_prefix[blockIdx.x], excl);
    A run = pre.v;
    uint32_t s = pre.s, c = pre.c;
    O o[IT];
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        if ((uint32_t)j < cnt) {
            if ((hb >> j) & 1) { run = ALG::identity(); s = base + j + 1; ++c; }
            if constexpr (!std::is_same_v<ALG, none_alg>) run = ALG::op(run, ALG::lift(v[j]));
        }
        const uint32_t p = base + j;
        const double rows = (double)(p - s + 2);                        // rows of the group up to and including p (s = start + 1)
        if constexpr (WR == SW_SUMS || WR == SW_RED_SUM) {
            if constexpr (std::is_floating_point_v<T>) o[j] = run; else o[j] = sum_alg<T>::to_i128(run);
        } else if constexpr (WR == SW_AVGS || WR == SW_RED_AVG) o[j] = sum_alg<T>::to_double(run) / rows;       // (s += arr[i]) / (double)(i + 1)
        else if constexpr (WR == SW_MINS || WR == SW_MAXP || WR == SW_RED_MIN) o[j] = run;
        else if constexpr (WR == SW_MAXS || WR == SW_RED_MAX) { T seed = dlimits<T>::min(); o[j] = seed > run ? seed : run; }    // max / maxs seed with numeric_limits<T>::min() (D8)
        else if constexpr (WR == SW_VARS || WR == SW_STDDEVS) {
            double var = (run.q - run.s * run.s / rows) / rows;
            if (var < 0) var = 0;
            o[j] = WR == SW_STDDEVS ? sqrt(var) : var;
        } else if constexpr (WR == SW_RED_VAR || WR == SW_RED_STDDEV) {          // (ssq - s * s / (FPType)(len + 1)) / (FPType)(len + 1): D9 kept
            const double np1 = (double)(uint32_t)(p - s + 3);
            double d;
            if constexpr (std::is_floating_point_v<T>) d = (run.q - run.s * run.s / np1) / np1;
            else {
                const __int128 ss = (__int128)run.s * (__int128)run.s;                   // the reference's 128-bit LongType product (|s| < 2^63: no wrap)
                d = ((double)run.q - i128_to_double(aqg_i128{(uint64_t)ss, (uint64_t)(ss >> 64)}) / np1) / np1;
            }
            o[j] = WR == SW_RED_STDDEV ? sqrt(d) : d;
        } else if constexpr (WR == SW_RAW) o[j] = run;
        else if constexpr (WR == SW_DIST) o[j] = p - (s - 1);
        else o[j] = c - 1;
        if constexpr (RED) {
            const uint32_t last = j + 1 < IT ? ((hb >> (j + 1)) & 1u) : hnext;   // the next position starts a group (bit n is set): p ends its group
            if ((uint32_t)j < cnt && last) static_cast<O*>(out)[c - 1] = o[j];
        }
    }
    if constexpr (!RED) store_tile_striped(static_cast<O*>(out), blockIdx.x * TS, o, n, reinterpret_cast<O*>(stage_raw));
}

// ---- shifts / ratios --------------------------------------------------------------------------------------------------------------
template <class T, int OP>
__global__ void __launch_bounds__(SB) seg_shift_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, const uint32_t* __restrict__ heads,
                                                       const uint32_t* __restrict__ shorts, void* __restrict__ out) {
    using FP = std::conditional_t<sizeof(T) == 4, float, double>;           // GetFPType
    using O = std::conditional_t<OP == AQG_SCAN_RATIOW, FP, T>;
    uint32_t lo, hi;
    wg_span(n, lo, hi, 256);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const T cur = x[i];
        O r;
        if constexpr (OP == AQG_SCAN_DELTAS) r = head_bit(heads, i) ? (T)0 : (T)(cur - x[i - 1]);          // ret[0] = 0 (aggregations.h:441)
        else if constexpr (OP == AQG_SCAN_PREV) r = head_bit(heads, i) ? cur : x[i - 1];                     // ret[0] = arr[0] (:457)
        else if constexpr (OP == AQG_SCAN_NEXT) r = head_bit(heads, i + 1) ? cur : x[i + 1];                 // the last row repeats (:476)
        else {
            // ratiow (:169-183): a group of at most w rows degrades to w = min(w, 1); the first w rows divide by the group's first row
            const uint32_t d = dist_to_head(heads, i, w);
            uint32_t we = w;
            if (shorts && d < w && head_bit(shorts, i - d)) we = w ? 1u : 0u;
            const T prv = d < we ? x[i - d] : x[i - we];
            r = (FP)(cur / (FP)prv);
        }
        static_cast<O*>(out)[i] = r;
    }
}

// four consecutive positions per lane (vector load / store, the start bits of the four as one nibble); ratios = ratiow(1) included.
// Needs 16-byte aligned columns (sub-views of columns take the scalar kernel above).  1e9 rows: deltas 2.7 -> see DESIGN.md
template <class T, int OP>
__global__ void __launch_bounds__(SB) seg_shift4_kernel(const T* __restrict__ x, uint32_t n, const uint32_t* __restrict__ heads, void* __restrict__ out) {
    using FP = std::conditional_t<sizeof(T) == 4, float, double>;
    using O = std::conditional_t<OP == AQG_SCAN_RATIOW, FP, T>;
    constexpr int V = 4;
    const uint32_t nv = n / V;
    const uint32_t q = blockIdx.x * SB + threadIdx.x;
    if (q < nv) {
        const uint32_t p = q * V;
        const pack<T, V> c = *reinterpret_cast<const pack<T, V>*>(x + p);
        const uint32_t hw = heads[p >> 5] >> (p & 31);                 // bits of p .. p+3 (p is a multiple of 4: they share a word)
        pack<O, V> o;
        if constexpr (OP == AQG_SCAN_NEXT) {
            const T right = head_bit(heads, p + V) ? c.v[V - 1] : x[p + V];   // (bit n is set: no read beyond the column)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const bool last = j + 1 < V ? ((hw >> (j + 1)) & 1u) : head_bit(heads, p + V);
                o.v[j] = last ? c.v[j] : (j + 1 < V ? c.v[j + 1] : right);
            }
        } else {
            const T left = (hw & 1u) ? c.v[0] : x[p - 1];               // (position 0 starts a group: no read in front of the column)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const bool h = (hw >> j) & 1u;
                const T prv = j ? c.v[j - 1] : left;
                if constexpr (OP == AQG_SCAN_DELTAS) o.v[j] = h ? (T)0 : (T)(c.v[j] - prv);
                else if constexpr (OP == AQG_SCAN_PREV) o.v[j] = h ? c.v[j] : prv;
                else o.v[j] = (FP)(c.v[j] / (FP)(h ? c.v[j] : prv));
            }
        }
        *reinterpret_cast<pack<O, V>*>(static_cast<O*>(out) + p) = o;
    }
    if (q == 0) {                                                       // the last n % 4 positions
        for (uint32_t i = nv * V; i < n; ++i) {
            const T cur = x[i];
            const bool h = head_bit(heads, i);
            O r;
            if constexpr (OP == AQG_SCAN_DELTAS) r = h ? (T)0 : (T)(cur - x[i - 1]);
            else if constexpr (OP == AQG_SCAN_PREV) r = h ? cur : x[i - 1];
            else if constexpr (OP == AQG_SCAN_NEXT) r = head_bit(heads, i + 1) ? cur : x[i + 1];
            else r = (FP)(cur / (FP)(h ? cur : x[i - 1]));
            static_cast<O*>(out)[i] = r;
        }
    }
}

// ---- sliding sums (sumw / avgw / varw / stddevw): window_sum_kernel of scan.hip with every window clamped at its group's start ----------
template <class T, int MODE>
__global__ void __launch_bounds__(SB) seg_window_sum_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, const uint8_t* __restrict__ heads8, void* __restrict__ out) {
    using ALG = std::conditional_t<(MODE >= 2), dsum_alg<T>, sum_alg<T>>;
    using A = typename ALG::A;
    using MX = max_alg<uint32_t>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ A lds_w[8];
    __shared__ A lds_w2[8];
    __shared__ uint32_t lds_m[8];
    const uint32_t tile_start = blockIdx.x * TS, tile_end = tile_start + TS < n ? tile_start + TS : n;
    const uint32_t H = (w - 1 + IT - 1) / IT * IT;                // LDS position p <-> row tile_start - H + p
    const uint32_t L = H + TS, nblk = L / IT;
    A* S = reinterpret_cast<A*>(smem_raw);
    A* Q = S + (MODE >= 2 ? L : 0);
    uint32_t* LH = reinterpret_cast<uint32_t*>(Q + L);            // per block of IT positions: {position + 1 of the last group start BEFORE the block (0: none in this tile), the block's start bits : 8}
    A carry = ALG::identity(), carry2 = ALG::identity();
    uint32_t carry_m = 0;
    for (uint32_t blk0 = 0; blk0 < nblk; blk0 += SB) {
        const uint32_t blk = blk0 + threadIdx.x;
        const int64_t g0 = (int64_t)tile_start - (int64_t)H + (int64_t)blk * IT;
        T v[IT];
        if (blk < nblk && g0 >= 0 && g0 + IT <= (int64_t)n && (((uintptr_t)(x + g0)) & (sizeof(T) * IT > 16 ? 15 : sizeof(T) * IT - 1)) == 0) {
            pack<T, IT> pk = *reinterpret_cast<const pack<T, IT>*>(x + g0);
#pragma unroll
            for (int j = 0; j < IT; ++j) v[j] = pk.v[j];
        } else {
#pragma unroll
            for (int j = 0; j < IT; ++j) { const int64_t g = g0 + j; v[j] = (blk < nblk && g >= 0 && g < (int64_t)n) ? x[g] : (T)0; }
        }
        const uint32_t hb = (blk < nblk && g0 >= 0 && g0 < (int64_t)n) ? heads8[g0 >> 3] : 0u;
        const uint32_t lh = hb ? blk * IT + (31 - __clz((int)hb)) + 1 : 0u;
        A loc[IT], loc2[IT];
        A a = ALG::identity(), q = ALG::identity();
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            a = ALG::op(a, ALG::lift(v[j])); loc[j] = a;
            if constexpr (MODE >= 2) { q = q + (double)v[j] * (double)v[j]; loc2[j] = q; }
        }
        A tot, tot2;
        A excl = ALG::op(carry, block_scan_excl<ALG>(a, lds_w, tot));
        A excl2 = ALG::identity();
        if constexpr (MODE >= 2) excl2 = ALG::op(carry2, block_scan_excl<ALG>(q, lds_w2, tot2));
        uint32_t totm;
        const uint32_t before = MX::op(carry_m, block_scan_excl<MX>(lh, lds_m, totm));
        if (blk < nblk) {
#pragma unroll
            for (int j = 0; j < IT; ++j) { S[blk * IT + j] = ALG::op(excl, loc[j]); if constexpr (MODE >= 2) Q[blk * IT + j] = ALG::op(excl2, loc2[j]); }
            LH[blk] = (before << 8) | hb;
        }
        carry = ALG::op(carry, tot);
        if constexpr (MODE >= 2) carry2 = ALG::op(carry2, tot2);
        carry_m = MX::op(carry_m, totm);
    }
    __syncthreads();
    for (uint32_t i = tile_start + threadIdx.x; i < tile_end; i += SB) {
        const uint32_t idx = i - tile_start + H, blk = idx >> 3, j = idx & 7;
        const uint32_t lhb = LH[blk], m = lhb & ((2u << j) - 1u);
        const uint32_t st = m ? blk * IT + (31 - __clz((int)m)) + 1 : (lhb >> 8);  // position + 1 of the group's start (0: further back than the halo)
        uint32_t lower = idx + 1 - w;                             // idx >= H >= w - 1
        if (st && st - 1 > lower) lower = st - 1;
        const uint32_t len = idx - lower + 1;
        A s = lower ? ALG::sub(S[idx], S[lower - 1]) : S[idx];
        if constexpr (MODE == 0) {
            if constexpr (std::is_floating_point_v<T>) static_cast<double*>(out)[i] = s;
            else static_cast<aqg_i128*>(out)[i] = ALG::to_i128(s);
        } else if constexpr (MODE == 1) {
            static_cast<double*>(out)[i] = ALG::to_double(s) / (double)len;
        } else {
            A sq = lower ? ALG::sub(Q[idx], Q[lower - 1]) : Q[idx];
            double mean = ALG::to_double(s) / (double)len;
            double var = ALG::to_double(sq) / (double)len - mean * mean;
            if (var < 0) var = 0;
            static_cast<double*>(out)[i] = MODE == 3 ? sqrt(var) : var;
        }
    }
}
// floating inputs, short windows: add the window's elements directly (oldest first), as window_direct_kernel of scan.hip
template <class T, int MODE>
__global__ void __launch_bounds__(SB) seg_window_direct_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, const uint32_t* __restrict__ heads, double* __restrict__ out) {
    uint32_t lo, hi;
    wg_span(n, lo, hi, 256);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const uint32_t len = dist_to_head(heads, i, w - 1) + 1;
        double s = 0;
        for (uint32_t j = i + 1 - len; j <= i; ++j) s += (double)x[j];
        out[i] = MODE == 0 ? s : s / (double)len;
    }
}
// windows wider than the LDS halo: a segmented inclusive prefix S (SW_RAW) and the distance D of every position to its group's start
template <class T, int MODE>
__global__ void __launch_bounds__(SB) seg_prefix_diff_kernel(const typename sum_alg<T>::A* __restrict__ S, const uint32_t* __restrict__ D, uint32_t n, uint32_t w, void* __restrict__ out) {
    using ALG = sum_alg<T>;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t d = D[i], len = d + 1 < w ? d + 1 : w;
        auto s = len == d + 1 ? S[i] : ALG::sub(S[i], S[i - len]);
        if constexpr (MODE == 0) {
            if constexpr (std::is_floating_point_v<T>) static_cast<double*>(out)[i] = s; else static_cast<aqg_i128*>(out)[i] = ALG::to_i128(s);
        } else static_cast<double*>(out)[i] = ALG::to_double(s) / (double)len;
    }
}

// ---- sliding min / max: window_minmax_kernel of scan.hip (doubling, eight positions per lane), a level taken only where the position
// 2^k back still belongs to the group: DS[p] = predecessors of p inside its group (capped) ----------------------------------------------
template <class T, bool IS_MAX>
__global__ void __launch_bounds__(SB) seg_window_minmax_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, const uint8_t* __restrict__ heads8, T* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int E = 8;
    struct alignas(E * sizeof(T) > 16 ? 16 : E * sizeof(T)) blk_t { T v[E]; };
    struct alignas(16) dblk_t { uint16_t d[E]; };
    __shared__ uint32_t lds_m[8];
    using MX = max_alg<uint32_t>;
    const uint32_t tile_start = blockIdx.x * TS;
    const uint32_t H = (w - 1 + E - 1) / E * E;
    const uint32_t L = H + TS, nblk = L / E;
    T* M0 = reinterpret_cast<T*>(smem_raw);
    T* M1 = M0 + L;
    uint16_t* DS = reinterpret_cast<uint16_t*>(M1 + L);
    T ident;
    if constexpr (std::is_floating_point_v<T>) ident = IS_MAX ? -(T)INFINITY : (T)INFINITY;
    else ident = IS_MAX ? dlimits<T>::min() : dlimits<T>::max();
    auto better = [](T a, T b) { if constexpr (IS_MAX) return b > a ? b : a; else return b < a ? b : a; };
    for (uint32_t p = threadIdx.x; p < H; p += SB) {
        const int64_t g = (int64_t)tile_start - (int64_t)H + p;
        M0[p] = g >= 0 ? x[g] : ident;
    }
    {
        const uint32_t g0 = tile_start + threadIdx.x * E;
        blk_t b;
        if (g0 + E <= n && (reinterpret_cast<uintptr_t>(x + g0) & (alignof(blk_t) - 1)) == 0) b = *reinterpret_cast<const blk_t*>(x + g0);
        else {
#pragma unroll
            for (int q = 0; q < E; ++q) b.v[q] = g0 + q < n ? x[g0 + q] : ident;
        }
        *reinterpret_cast<blk_t*>(M0 + H + threadIdx.x * E) = b;
    }
    uint32_t carry_m = 0;
    for (uint32_t blk0 = 0; blk0 < nblk; blk0 += SB) {             // distances to the group starts (a start further back than the halo: "far")
        const uint32_t blk = blk0 + threadIdx.x;
        const int64_t g0 = (int64_t)tile_start - (int64_t)H + (int64_t)blk * E;
        const uint32_t hb = (blk < nblk && g0 >= 0 && g0 < (int64_t)n) ? heads8[g0 >> 3] : 0u;
        const uint32_t lh = hb ? blk * E + (31 - __clz((int)hb)) + 1 : 0u;
        uint32_t totm;
        uint32_t cur = MX::op(carry_m, block_scan_excl<MX>(lh, lds_m, totm));
        if (blk < nblk) {
            dblk_t dd;
#pragma unroll
            for (int q = 0; q < E; ++q) {
                if ((hb >> q) & 1) cur = blk * E + q + 1;
                const uint32_t dist = cur ? blk * E + q - (cur - 1) : 0xFFFFu;
                dd.d[q] = (uint16_t)(dist < 0xFFFFu ? dist : 0xFFFFu);
            }
            *reinterpret_cast<dblk_t*>(DS + blk * E) = dd;
        }
        carry_m = MX::op(carry_m, totm);
    }
    __syncthreads();
    uint32_t K = 0;
    while ((2u << K) <= w) ++K;                                    // 2^K <= w < 2^(K+1)
    const uint32_t KA = K < 3 ? K : 3;
    T* cur = M0; T* nxt = M1;
    if (KA) {
        for (uint32_t blk = threadIdx.x; blk < nblk; blk += SB) {
            T a[2 * E];
            uint32_t dist[2 * E];
            const blk_t own = *reinterpret_cast<const blk_t*>(cur + blk * E);
            const dblk_t downd = *reinterpret_cast<const dblk_t*>(DS + blk * E);
            blk_t prev;
            dblk_t dprev;
            if (blk) { prev = *reinterpret_cast<const blk_t*>(cur + (blk - 1) * E); dprev = *reinterpret_cast<const dblk_t*>(DS + (blk - 1) * E); }
#pragma unroll
            for (int q = 0; q < E; ++q) { a[q] = blk ? prev.v[q] : ident; a[E + q] = own.v[q]; dist[q] = blk ? dprev.d[q] : 0u; dist[E + q] = downd.d[q]; }
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                if (k < KA) {
                    const int d = 1 << k;
#pragma unroll
                    for (int j = 2 * E - 1; j >= d; --j) if (dist[j] >= (uint32_t)d) a[j] = better(a[j], a[j - d]);
                }
            }
            blk_t o;
#pragma unroll
            for (int q = 0; q < E; ++q) o.v[q] = a[E + q];
            *reinterpret_cast<blk_t*>(nxt + blk * E) = o;
        }
        __syncthreads();
        T* t = cur; cur = nxt; nxt = t;
    }
    for (uint32_t k = KA; k < K; ++k) {
        const uint32_t db = (1u << k) / E, dk = 1u << k;
        for (uint32_t blk = threadIdx.x; blk < nblk; blk += SB) {
            blk_t a = *reinterpret_cast<const blk_t*>(cur + blk * E);
            if (blk >= db) {
                const blk_t b = *reinterpret_cast<const blk_t*>(cur + (blk - db) * E);
                const dblk_t dd = *reinterpret_cast<const dblk_t*>(DS + blk * E);
#pragma unroll
                for (int q = 0; q < E; ++q) if (dd.d[q] >= dk) a.v[q] = better(a.v[q], b.v[q]);
            }
            *reinterpret_cast<blk_t*>(nxt + blk * E) = a;
        }
        __syncthreads();
        T* t = cur; cur = nxt; nxt = t;
    }
    const uint32_t off = w - (1u << K);
    const uint32_t p0 = H + threadIdx.x * E, g0 = tile_start + threadIdx.x * E;
    if (g0 < n) {
        blk_t a = *reinterpret_cast<const blk_t*>(cur + p0);
        if (off) {
            const dblk_t dd = *reinterpret_cast<const dblk_t*>(DS + p0);
#pragma unroll
            for (int q = 0; q < E; ++q) if (dd.d[q] >= off) a.v[q] = better(a.v[q], cur[p0 + q - off]);
        }
        if (g0 + E <= n && (reinterpret_cast<uintptr_t>(out + g0) & (alignof(blk_t) - 1)) == 0) *reinterpret_cast<blk_t*>(out + g0) = a;
        else {
#pragma unroll
            for (int q = 0; q < E; ++q) if (g0 + q < n) out[g0 + q] = a.v[q];
        }
    }
}
// windows wider than the LDS halo: doubling passes through HBM guarded by the distances D
template <class T, bool IS_MAX>
__global__ void __launch_bounds__(SB) seg_doubling_pass_kernel(const T* __restrict__ src, T* __restrict__ dst, const uint32_t* __restrict__ D, uint32_t n, uint32_t d) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        T a = src[i];
        if (D[i] >= d) { T b = src[i - d]; if constexpr (IS_MAX) a = b > a ? b : a; else a = b < a ? b : a; }
        dst[i] = a;
    }
}
template <class T, bool IS_MAX>
__global__ void __launch_bounds__(SB) seg_doubling_final_kernel(const T* __restrict__ m, T* __restrict__ out, const uint32_t* __restrict__ D, uint32_t n, uint32_t w, uint32_t span) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t len = D[i] + 1 < w ? D[i] + 1 : w;
        T a = m[i];
        if (len > span) { T b = m[i - (len - span)]; if constexpr (IS_MAX) a = b > a ? b : a; else a = b < a ? b : a; }
        out[i] = a;
    }
}

__global__ void __launch_bounds__(256) ends_kernel(const uint32_t* __restrict__ off, uint32_t G, uint32_t* __restrict__ last_pos) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) last_pos[g] = off[g + 1] - 1;
}
__global__ void __launch_bounds__(256) counts64_kernel(const uint32_t* __restrict__ counts, uint32_t G, uint64_t* __restrict__ out) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) out[g] = counts[g];
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------
int seg_realloc(aqg_ctx* ctx, uint32_t** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return AQG_OK;
    if (*p) { aqg_pool_give(ctx, *p, *cap); *p = nullptr; *cap = 0; }
    size_t want = need < 256 ? 256 : need;
    if (void* q = aqg_pool_take(ctx, want, cap)) { *p = static_cast<uint32_t*>(q); return AQG_OK; }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, want);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); (void)hipGetLastError(); return AQG_ERR_NOMEM; }
    *p = static_cast<uint32_t*>(q);
    *cap = want;
    return AQG_OK;
}
size_t heads_bytes(uint32_t n) { return (((size_t)n + 32) / 32 + 8) * 4; }      // bit n included, two words of padding (the byte behind a block is read)

// offsets + start bitmap of the flat layout, made once per build (uses the workspace: call before any sub-allocation of a call)
int ensure_flat(aqg_ctx* ctx, aqg_groupby* g) {
    if (g->flat_valid) return AQG_OK;
    const uint32_t n = g->n, G = g->ngroups;
    AQG_TRY(seg_realloc(ctx, &g->flat_off, &g->cap_flat_off, ((size_t)G + 2) * 4));
    AQG_TRY(seg_realloc(ctx, &g->flat_heads, &g->cap_flat_heads, heads_bytes(n)));
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, ((size_t)(G + 2048) / 2048 + 16) * 4 + 4096));
    uint32_t* bsum;
    AQG_TRY(aqg_ws_get(ctx, (G + 2048) / 2048 + 16, &bsum));
    AQG_TRY(aqg_group_offsets(ctx, g, g->flat_off, bsum));
    AQG_HIP(ctx, hipMemsetAsync(g->flat_heads, 0, heads_bytes(n), ctx->stream));
    hipLaunchKernelGGL(heads_kernel, dim3(aqg_grid(ctx, (uint64_t)G + 1, 256, 1, 8)), dim3(256), 0, ctx->stream, g->flat_off, g->counts, G, g->flat_heads, (uint32_t*)nullptr, 0u);
    AQG_TRY(aqg_check_launch(ctx, "heads_kernel"));
    g->flat_valid = true;
    g->flat_gid_valid = false;
    g->flat_short_w = 0;
    return AQG_OK;
}
int ensure_short(aqg_ctx* ctx, aqg_groupby* g, uint32_t w) {
    if (g->flat_short_w == w && g->flat_short) return AQG_OK;
    AQG_TRY(seg_realloc(ctx, &g->flat_short, &g->cap_flat_short, heads_bytes(g->n)));
    AQG_HIP(ctx, hipMemsetAsync(g->flat_short, 0, heads_bytes(g->n), ctx->stream));
    hipLaunchKernelGGL(heads_kernel, dim3(aqg_grid(ctx, (uint64_t)g->ngroups + 1, 256, 1, 8)), dim3(256), 0, ctx->stream, g->flat_off, g->counts, g->ngroups,
                       (uint32_t*)nullptr, g->flat_short, w);
    AQG_TRY(aqg_check_launch(ctx, "heads_kernel"));
    g->flat_short_w = w;
    return AQG_OK;
}

size_t carry_ws_bytes(uint32_t n) {
    const size_t ntiles = aqg_ceil_div(n, TS);
    return ntiles * 32 + (ntiles / CH + 2) * 32 + 4096;
}
// one segmented prefix pass (workspace already sized): carries -> their scan -> results
template <class T, class ALG, int WR>
int seg_prefix(aqg_ctx* ctx, aqg_groupby* g, const T* x, uint32_t n, void* out) {
    using C = SegCarry<typename ALG::A>;
    using O = typename seg_out<T, WR>::type;
    static_assert(sizeof(C) <= 32, "carry_ws_bytes");
    const uint32_t ntiles = aqg_ceil_div(n, TS);
    const uint8_t* heads8 = reinterpret_cast<const uint8_t*>(g->flat_heads);
    C *carry, *chunk_tot;
    AQG_TRY(aqg_ws_get(ctx, ntiles, &carry));
    AQG_TRY(aqg_ws_get(ctx, (size_t)ntiles / CH + 2, &chunk_tot));
    hipLaunchKernelGGL((seg_tile_reduce_kernel<T, ALG>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, heads8, carry);
    launch_agg_scan<seg_alg<ALG>>(ctx, carry, ntiles, chunk_tot);
    aqg_kernel_timer_begin(ctx);
    hipLaunchKernelGGL((seg_tile_scan_kernel<T, ALG, WR>), dim3(ntiles), dim3(SB), WR >= SW_RED_SUM ? 0 : (size_t)TS * sizeof(O), ctx->stream, x, n, heads8, carry, out);
    aqg_kernel_timer_end(ctx);
    return aqg_check_launch(ctx, "segmented prefix scan");
}
int dist_column(aqg_ctx* ctx, aqg_groupby* g, uint32_t n, uint32_t* D) { return seg_prefix<uint8_t, none_alg, SW_DIST>(ctx, g, nullptr, n, D); }

size_t scan_ws_bytes(int op, int t, uint32_t n, uint32_t w) {
    size_t need = carry_ws_bytes(n) * 2 + 65536;
    const size_t esz = aqg_dtype_size(t);
    switch (op) {
    case AQG_SCAN_SUMW: case AQG_SCAN_AVGW: need += (size_t)n * (4 + 16) + 8192; break;          // (only the wide-window path uses them)
    case AQG_SCAN_MINW: case AQG_SCAN_MAXW: need += (size_t)n * (4 + 2 * esz) + 8192; break;
    default: break;
    }
    (void)w;
    return need;
}

// the scan of a column already in the flat layout (workspace sized by scan_ws_bytes and not reset in here)
int scan_flat(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* xv, uint32_t w, void* out) {
    const uint32_t n = g->n;
    const uint8_t* heads8 = reinterpret_cast<const uint8_t*>(g->flat_heads);
    const uint32_t* heads = g->flat_heads;
    if (op == AQG_SCAN_RATIOW && w >= 2) AQG_TRY(ensure_short(ctx, g, w));
    const uint32_t* shorts = (op == AQG_SCAN_RATIOW && w >= 2) ? g->flat_short : nullptr;
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        const T* x = static_cast<const T*>(xv);
        const uint32_t ntiles = aqg_ceil_div(n, TS);
        const unsigned egrid = aqg_grid(ctx, n, SB, 4, 16);
        auto shift = [&](auto kern, const char* what) -> int {
            aqg_kernel_timer_begin(ctx);
            hipLaunchKernelGGL(kern, dim3(egrid), dim3(SB), 0, ctx->stream, x, n, w, heads, shorts, out);
            aqg_kernel_timer_end(ctx);
            return aqg_check_launch(ctx, what);
        };
        const bool al16 = ((reinterpret_cast<uintptr_t>(xv) | reinterpret_cast<uintptr_t>(out)) & 15) == 0 && n >= 4;
        auto shift4 = [&](auto kern, const char* what) -> int {
            aqg_kernel_timer_begin(ctx);
            hipLaunchKernelGGL(kern, dim3(aqg_ceil_div(n / 4, SB)), dim3(SB), 0, ctx->stream, x, n, heads, out);
            aqg_kernel_timer_end(ctx);
            return aqg_check_launch(ctx, what);
        };
        switch (op) {
        case AQG_SCAN_SUMS: return seg_prefix<T, sum_alg<T>, SW_SUMS>(ctx, g, x, n, out);
        case AQG_SCAN_AVGS: return seg_prefix<T, sum_alg<T>, SW_AVGS>(ctx, g, x, n, out);
        case AQG_SCAN_MINS: return seg_prefix<T, min_alg<T>, SW_MINS>(ctx, g, x, n, out);
        case AQG_SCAN_MAXS: return seg_prefix<T, max_alg<T>, SW_MAXS>(ctx, g, x, n, out);
        case AQG_SCAN_VARS: return seg_prefix<T, var2_alg<T>, SW_VARS>(ctx, g, x, n, out);
        case AQG_SCAN_STDDEVS: return seg_prefix<T, var2_alg<T>, SW_STDDEVS>(ctx, g, x, n, out);
        case AQG_SCAN_DELTAS: return al16 ? shift4(&seg_shift4_kernel<T, AQG_SCAN_DELTAS>, "deltas (grouped)") : shift(&seg_shift_kernel<T, AQG_SCAN_DELTAS>, "deltas (grouped)");
        case AQG_SCAN_PREV: return al16 ? shift4(&seg_shift4_kernel<T, AQG_SCAN_PREV>, "prev (grouped)") : shift(&seg_shift_kernel<T, AQG_SCAN_PREV>, "prev (grouped)");
        case AQG_SCAN_NEXT: return al16 ? shift4(&seg_shift4_kernel<T, AQG_SCAN_NEXT>, "aggnext (grouped)") : shift(&seg_shift_kernel<T, AQG_SCAN_NEXT>, "aggnext (grouped)");
        case AQG_SCAN_RATIOW: return (al16 && w == 1) ? shift4(&seg_shift4_kernel<T, AQG_SCAN_RATIOW>, "ratios (grouped)") : shift(&seg_shift_kernel<T, AQG_SCAN_RATIOW>, "ratiow (grouped)");
        case AQG_SCAN_SUMW: case AQG_SCAN_AVGW: case AQG_SCAN_VARW: case AQG_SCAN_STDDEVW: {
            using A = typename sum_alg<T>::A;
            const uint32_t ww = w > n ? n : w;                                          // (a window is clamped by its group anyway)
            const bool var = op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW;
            if constexpr (std::is_floating_point_v<T>) {
                if (!var && ww <= 64) {
                    aqg_kernel_timer_begin(ctx);
                    if (op == AQG_SCAN_SUMW) hipLaunchKernelGGL((seg_window_direct_kernel<T, 0>), dim3(egrid), dim3(SB), 0, ctx->stream, x, n, ww, heads, static_cast<double*>(out));
                    else hipLaunchKernelGGL((seg_window_direct_kernel<T, 1>), dim3(egrid), dim3(SB), 0, ctx->stream, x, n, ww, heads, static_cast<double*>(out));
                    aqg_kernel_timer_end(ctx);
                    return aqg_check_launch(ctx, "seg_window_direct_kernel");
                }
            }
            const size_t ext = (size_t)TS + (ww - 1 + IT - 1) / IT * IT;
            const size_t lds = (var ? ext * sizeof(double) * 2 : ext * sizeof(A)) + ext / IT * 4 + 16;
            if (lds <= HALO_MAX_BYTES) {
                auto go = [&](auto kern) -> int {
                    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
                    aqg_kernel_timer_begin(ctx);
                    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(SB), lds, ctx->stream, x, n, ww, heads8, out);
                    aqg_kernel_timer_end(ctx);
                    return aqg_check_launch(ctx, "seg_window_sum_kernel");
                };
                switch (op) {
                case AQG_SCAN_SUMW: return go(&seg_window_sum_kernel<T, 0>);
                case AQG_SCAN_AVGW: return go(&seg_window_sum_kernel<T, 1>);
                case AQG_SCAN_VARW: return go(&seg_window_sum_kernel<T, 2>);
                default: return go(&seg_window_sum_kernel<T, 3>);
                }
            }
            if (var) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_grouped_scan: varw/stddevw window too large for the LDS halo");
            A* S; uint32_t* D;
            AQG_TRY(aqg_ws_get(ctx, n, &S));
            AQG_TRY(aqg_ws_get(ctx, n, &D));
            AQG_TRY((seg_prefix<T, sum_alg<T>, SW_RAW>(ctx, g, x, n, S)));
            AQG_TRY(dist_column(ctx, g, n, D));
            if (op == AQG_SCAN_SUMW) hipLaunchKernelGGL((seg_prefix_diff_kernel<T, 0>), dim3(egrid), dim3(SB), 0, ctx->stream, S, D, n, ww, out);
            else hipLaunchKernelGGL((seg_prefix_diff_kernel<T, 1>), dim3(egrid), dim3(SB), 0, ctx->stream, S, D, n, ww, out);
            return aqg_check_launch(ctx, "wide window sum (grouped)");
        }
        case AQG_SCAN_MINW: case AQG_SCAN_MAXW: {
            const bool is_max = op == AQG_SCAN_MAXW;
            // the deque never expires anything when w == 0 or w >= n: the running min / max of the group (no seed)
            if (w == 0 || w >= n) return is_max ? seg_prefix<T, max_alg<T>, SW_MAXP>(ctx, g, x, n, out) : seg_prefix<T, min_alg<T>, SW_MINS>(ctx, g, x, n, out);
            const size_t ext = (size_t)TS + (w - 1 + 7) / 8 * 8;
            const size_t lds = ext * sizeof(T) * 2 + ext * 2 + 16;
            if (lds <= HALO_MAX_BYTES) {
                auto go = [&](auto kern) -> int {
                    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
                    aqg_kernel_timer_begin(ctx);
                    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(SB), lds, ctx->stream, x, n, w, heads8, static_cast<T*>(out));
                    aqg_kernel_timer_end(ctx);
                    return aqg_check_launch(ctx, "seg_window_minmax_kernel");
                };
                return is_max ? go(&seg_window_minmax_kernel<T, true>) : go(&seg_window_minmax_kernel<T, false>);
            }
            T *b0, *b1; uint32_t* D;
            AQG_TRY(aqg_ws_get(ctx, n, &b0));
            AQG_TRY(aqg_ws_get(ctx, n, &b1));
            AQG_TRY(aqg_ws_get(ctx, n, &D));
            AQG_TRY(dist_column(ctx, g, n, D));
            uint32_t K = 0;
            while ((2u << K) <= w && K < 31) ++K;
            const T* src = x;
            T* dst = b0;
            for (uint32_t k = 0; k < K; ++k) {
                if (is_max) hipLaunchKernelGGL((seg_doubling_pass_kernel<T, true>), dim3(egrid), dim3(SB), 0, ctx->stream, src, dst, D, n, 1u << k);
                else hipLaunchKernelGGL((seg_doubling_pass_kernel<T, false>), dim3(egrid), dim3(SB), 0, ctx->stream, src, dst, D, n, 1u << k);
                src = dst;
                dst = dst == b0 ? b1 : b0;
            }
            if (is_max) hipLaunchKernelGGL((seg_doubling_final_kernel<T, true>), dim3(egrid), dim3(SB), 0, ctx->stream, src, static_cast<T*>(out), D, n, w, 1u << K);
            else hipLaunchKernelGGL((seg_doubling_final_kernel<T, false>), dim3(egrid), dim3(SB), 0, ctx->stream, src, static_cast<T*>(out), D, n, w, 1u << K);
            return aqg_check_launch(ctx, "wide window min/max (grouped)");
        }
        }
        return AQG_ERR_ARG;
    });
}

int check_build(aqg_ctx* ctx, const aqg_groupby* g, const char* what) {
    if (!ctx || !g) return aqg_fail(ctx, AQG_ERR_ARG, what);
    if (!g->has_reversemap || !g->has_counts) return aqg_fail(ctx, AQG_ERR_ARG, "grouped scan: the handle was not made by aqg_groupby_build");
    return AQG_OK;
}
int flat_esz(int t) {
    switch (t) {
    case AQG_INT8: case AQG_UINT8: case AQG_BOOL: case AQG_CHAR: return 1;
    case AQG_INT16: case AQG_UINT16: return 2;
    case AQG_INT32: case AQG_UINT32: case AQG_FLOAT: case AQG_DATE: return 4;
    case AQG_INT64: case AQG_UINT64: case AQG_DOUBLE: case AQG_TIME: return 8;
    }
    return 0;
}

} // namespace

extern "C" {

const uint32_t* aqg_groupby_offsets(aqg_groupby* g) {
    if (!g || !g->has_counts) return nullptr;
    return ensure_flat(g->ctx, g) == AQG_OK ? g->flat_off : nullptr;
}

int aqg_grouped_flatten(aqg_ctx* ctx, aqg_groupby* g, int t, const void* x, void* out_flat) {
    AQG_TRY(check_build(ctx, g, "aqg_grouped_flatten: bad argument"));
    if ((!x || !out_flat) && g->n) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_flatten: bad argument");
    const int esz = flat_esz(t);
    if (!esz) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_grouped_flatten: 1-, 2-, 4- and 8-byte elements");
    if (g->n == 0) return AQG_OK;
    return aqg_radix_by_group(ctx, g, nullptr, x, esz, out_flat, /*ws_managed=*/false);
}

int aqg_grouped_scan_flat(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* xflat, uint32_t w, void* out_flat) {
    AQG_TRY(check_build(ctx, g, "aqg_grouped_scan_flat: bad argument"));
    if ((!xflat || !out_flat) && g->n) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_scan_flat: bad argument");
    if (op < 0 || op > AQG_SCAN_STDDEVW) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_scan_flat: bad op");
    if (w == 0 && (op == AQG_SCAN_SUMW || op == AQG_SCAN_AVGW || op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW))
        return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_scan: window 0 is undefined for sumw/avgw/varw");
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "grouped scan: the column dtype is not numeric");
    if (g->n == 0) return AQG_OK;
    AQG_TRY(ensure_flat(ctx, g));
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, scan_ws_bytes(op, t, g->n, w)));
    return scan_flat(ctx, g, op, t, xflat, w, out_flat);
}

int aqg_grouped_scan(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* x, uint32_t w, void* out_flat) {
    AQG_TRY(check_build(ctx, g, "aqg_grouped_scan: bad argument"));
    if ((!x || !out_flat) && g->n) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_scan: bad argument");
    if (op < 0 || op > AQG_SCAN_STDDEVW) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_scan: bad op");
    if (w == 0 && (op == AQG_SCAN_SUMW || op == AQG_SCAN_AVGW || op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW))
        return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_scan: window 0 is undefined for sumw/avgw/varw");
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "grouped scan: the column dtype is not numeric");
    const uint32_t n = g->n;
    if (n == 0) return AQG_OK;
    AQG_TRY(ensure_flat(ctx, g));
    const int esz = flat_esz(t);
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, (size_t)n * esz + 4096 + aqg_postproc_ws_bytes(n, g->ngroups, esz) + scan_ws_bytes(op, t, n, w)));
    unsigned char* xs;
    AQG_TRY(aqg_ws_get(ctx, (size_t)n * esz + 64, &xs));
    AQG_TRY(aqg_radix_by_group(ctx, g, nullptr, x, esz, xs, /*ws_managed=*/true));
    return scan_flat(ctx, g, op, t, xs, w, out_flat);
}

// out[g] = op(flat[offsets[g] .. offsets[g+1])): reductions of per-group scan results (`max(ratios(x[vecs[g]]))`, tests/q4.a:23)
int aqg_grouped_reduce_flat(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* xflat, void* out_dev) {
    AQG_TRY(check_build(ctx, g, "aqg_grouped_reduce_flat: bad argument"));
    if ((!xflat && g->n) || !out_dev) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_reduce_flat: bad argument");
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_grouped_reduce_flat: value dtype");
    const uint32_t n = g->n, G = g->ngroups;
    if (G == 0) return AQG_OK;
    AQG_TRY(ensure_flat(ctx, g));
    if (op == AQG_RED_COUNT) {
        hipLaunchKernelGGL(counts64_kernel, dim3(aqg_grid(ctx, G, 256, 1, 8)), dim3(256), 0, ctx->stream, g->counts, G, static_cast<uint64_t*>(out_dev));
        return aqg_check_launch(ctx, "counts64_kernel");
    }
    if (op == AQG_RED_FIRST) return aqg_gather(ctx, t, xflat, g->flat_off, G, out_dev);
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, carry_ws_bytes(n) * 2 + (size_t)G * 4 + 65536));
    if (op == AQG_RED_LAST) {
        uint32_t* lastp;
        AQG_TRY(aqg_ws_get(ctx, G, &lastp));
        hipLaunchKernelGGL(ends_kernel, dim3(aqg_grid(ctx, G, 256, 1, 8)), dim3(256), 0, ctx->stream, g->flat_off, G, lastp);
        return aqg_gather(ctx, t, xflat, lastp, G, out_dev);
    }
    if (op == AQG_RED_SUM || op == AQG_RED_AVG || op == AQG_RED_MIN || op == AQG_RED_MAX) {
        return aqg_dispatch_num(t, [&](auto tt) -> int {
            using T = typename decltype(tt)::type;
            const T* x = static_cast<const T*>(xflat);
            switch (op) {
            case AQG_RED_SUM: return seg_prefix<T, sum_alg<T>, SW_RED_SUM>(ctx, g, x, n, out_dev);
            case AQG_RED_AVG: return seg_prefix<T, sum_alg<T>, SW_RED_AVG>(ctx, g, x, n, out_dev);
            case AQG_RED_MIN: return seg_prefix<T, min_alg<T>, SW_RED_MIN>(ctx, g, x, n, out_dev);
            default: return seg_prefix<T, max_alg<T>, SW_RED_MAX>(ctx, g, x, n, out_dev);
            }
        });
    }
    if ((op == AQG_RED_VAR || op == AQG_RED_STDDEV) && (aqg_dtype_size(t) <= 4 || dt_is_fp(t))) {
        return aqg_dispatch_num(t, [&](auto tt) -> int {
            using T = typename decltype(tt)::type;
            if constexpr (sizeof(T) <= 4 || std::is_floating_point_v<T>) {
                const T* x = static_cast<const T*>(xflat);
                return op == AQG_RED_VAR ? seg_prefix<T, varred_alg<T>, SW_RED_VAR>(ctx, g, x, n, out_dev) : seg_prefix<T, varred_alg<T>, SW_RED_STDDEV>(ctx, g, x, n, out_dev);
            } else return AQG_ERR_DTYPE;
        });
    }
    // VAR / STDDEV: through the group-by plans, keyed by the group index of every flat position
    if (!g->flat_gid_valid) {
        AQG_TRY(seg_realloc(ctx, &g->flat_gid, &g->cap_flat_gid, ((size_t)n + 4) * 4));
        AQG_TRY((seg_prefix<uint8_t, none_alg, SW_GID>(ctx, g, nullptr, n, g->flat_gid)));
        g->flat_gid_valid = true;
    }
    return aqg_grouped_reduce_keyed(ctx, g, g->flat_gid, op, t, xflat, out_dev);
}

} // extern "C"

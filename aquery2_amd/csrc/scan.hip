// scan.hip -- prefix scans, sliding windows and shifts of a column
// (reference server/aggregations.h: mins/maxs :89-125, minw/maxw :127-167, ratiow :169-201,
//  sums/avgs :203-236, sumw/avgw :238-281, vars/stddevs :350-381, varw/stddevw :283-330,
//  deltas/prev/aggnext :439-485).
//
// Prefix scans: reduce-then-scan over 2048-element tiles (tile reduce -> one-workgroup scan of the
// tile aggregates -> tile scan with carry-in); inside a tile: 8 consecutive elements per lane,
// wavefront inclusive scan (__shfl_up over 64 lanes), LDS combine of the 4 waves.
// Windows: every tile stages its elements plus a (w-1)-element halo in LDS; sums use the tile-local
// prefix difference S[i]-S[i-w] (exact for integers), min/max use log2(w) LDS doubling steps.
// Integer results are exact (bit-identical to the reference); floating sums follow a tree order.
// HBM-bound; algorithmic bytes per row = sizeof(T) + sizeof(out).
#include "aqg_internal.hpp"
#include "dev_common.hpp"
#include "chain_dev.hpp"
#include "scan_dev.hpp"

namespace {
using namespace aqgscan;


// K1: aggregate of each tile
template <class T, class ALG> __global__ void __launch_bounds__(SB) tile_reduce_kernel(const T* __restrict__ x, uint32_t n, typename ALG::A* __restrict__ tile_agg) {
    using A = typename ALG::A;
    __shared__ A lds_w[8];
    uint32_t base = blockIdx.x * TS + threadIdx.x * IT, cnt;
    T v[IT];
    load_tile_items(x, n, base, v, cnt);
    A a = ALG::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) if ((uint32_t)j < cnt) a = ALG::op(a, ALG::lift(v[j]));
    A total;
    block_scan_excl<ALG>(a, lds_w, total);
    if (threadIdx.x == 0) tile_agg[blockIdx.x] = total;
}
// K3: scan inside the tile with the carry-in; WRITER(out, i, inclusive_value)
enum : int { W_SUMS = 0, W_AVGS = 1, W_MINS = 2, W_MAXS = 3, W_MAXP = 4 /* running max without the reference's seed (maxw, w >= n) */ };

// a scan that resumes a column sharded by row range (aqg_scan_resume): the sum of every earlier row in the result's LongType
// and the number of earlier rows.  Kept apart from the tile accumulator, which is 64 bits wide for <= 4-byte integer columns
// while the carry of a table of more than 2^32 rows is not.
struct ScanSeed {
    aqg_i128 i;        // integer columns
    double d;          // floating columns (-0.0 when there is nothing to add: the only value that leaves every double unchanged)
    uint64_t row0;     // rows before this shard (avgs divide by row0 + i + 1)
};

// writes one tile's results: `run` = fold of everything before this lane's first element
template <class T, class ALG, int WR>
__device__ inline void write_tile(typename ALG::A run, const T (&v)[IT], uint32_t cnt, uint32_t base, uint32_t tile_base, uint32_t n, void* __restrict__ out,
                                  unsigned char* stage_raw, const ScanSeed& seed = ScanSeed{{0, 0}, -0.0, 0}) {
    if constexpr (WR == W_SUMS) {
        using O = std::conditional_t<std::is_floating_point_v<T>, double, aqg_i128>;
        O o[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            if ((uint32_t)j < cnt) run = ALG::op(run, ALG::lift(v[j]));
            if constexpr (std::is_floating_point_v<T>) o[j] = seed.d + run; else o[j] = i128_add(seed.i, sum_alg<T>::to_i128(run));
        }
        store_tile_striped(static_cast<O*>(out), tile_base, o, n, reinterpret_cast<O*>(stage_raw));
    } else if constexpr (WR == W_AVGS) {
        double o[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            if ((uint32_t)j < cnt) run = ALG::op(run, ALG::lift(v[j]));
            double sum;                                                            // (s += arr[i]) / (double)(i + 1)
            if constexpr (std::is_floating_point_v<T>) sum = seed.d + run;
            else {
                const aqg_i128 t = i128_add(seed.i, sum_alg<T>::to_i128(run));
                if constexpr (std::is_unsigned_v<T>) sum = u128_to_double(t.hi, t.lo); else sum = i128_to_double(t);
            }
            o[j] = sum / (double)(seed.row0 + base + j + 1);
        }
        store_tile_striped(static_cast<double*>(out), tile_base, o, n, reinterpret_cast<double*>(stage_raw));
    } else {
        T o[IT];
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            if ((uint32_t)j < cnt) run = ALG::op(run, ALG::lift(v[j]));
            T r = run;
            if constexpr (WR == W_MAXS) { T seed = dlimits<T>::min(); r = seed > r ? seed : r; }  // maxs seeds with numeric_limits<T>::min()
            o[j] = r;
        }
        store_tile_striped(static_cast<T*>(out), tile_base, o, n, reinterpret_cast<T*>(stage_raw));
    }
}

template <class T, class ALG, int WR>
__global__ void __launch_bounds__(SB) tile_scan_kernel(const T* __restrict__ x, uint32_t n, const typename ALG::A* __restrict__ tile_prefix,
                                                       void* __restrict__ out, ScanSeed seed) {
    using A = typename ALG::A;
    __shared__ A lds_w[8];
    extern __shared__ __align__(16) unsigned char stage_raw[];
    uint32_t base = blockIdx.x * TS + threadIdx.x * IT, cnt;
    T v[IT];
    load_tile_items(x, n, base, v, cnt);
    A a = ALG::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) if ((uint32_t)j < cnt) a = ALG::op(a, ALG::lift(v[j]));
    A total;
    A run = ALG::op(tile_prefix[blockIdx.x], block_scan_excl<ALG>(a, lds_w, total));
    write_tile<T, ALG, WR>(run, v, cnt, base, blockIdx.x * TS, n, out, stage_raw, seed);
}

// ---- single-pass scan: chained links with decoupled look-back (protocol: chain_dev.hpp) ----------------------------------
using namespace aqgchain;

// fold of one value per lane over the workgroup (every lane gets it); ALG::op must commute
template <class ALG, class A> __device__ inline A block_fold(A v, A* lds_r /* >= SB / 64 */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = ALG::op(v, shfl_xor_any(v, off));
    if (lane_id() == 0) lds_r[wave_id()] = v;
    __syncthreads();
    A t = ALG::identity();
#pragma unroll
    for (int w = 0; w < SB / 64; ++w) t = ALG::op(t, lds_r[w]);
    return t;
}

template <class T> constexpr int chain_m() { return sizeof(T) <= 4 ? 8 : 4; }   // 2048-element sub-tiles per chain link

template <class T, class ALG, int WR, int M>
__global__ void __launch_bounds__(SB) chained_scan_kernel(const T* __restrict__ x, uint32_t n, uint32_t* __restrict__ ctrl /* [0] link counter, [1] error */,
                                                          uint64_t* __restrict__ slots, void* __restrict__ out, typename ALG::A seed /* fold of the rows of earlier shards (identity: none) */) {
    using A = typename ALG::A;
    constexpr int NW = flagged_words<A>();
    __shared__ A lds_w[8];
    __shared__ A lds_r[SB / 64];
    __shared__ uint32_t s_tile;
    __shared__ A s_prefix;
    extern __shared__ __align__(16) unsigned char stage_raw[];
    if (threadIdx.x == 0) s_tile = atomicAdd(&ctrl[0], 1u);
    __syncthreads();
    const uint32_t tile = s_tile;
    const uint64_t link_base = (uint64_t)tile * M * TS;
    T v[M][IT];                                                 // one chain link = M sub-tiles kept in registers
    uint32_t cnt[M];
    A a[M], excl[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint64_t b64 = link_base + (uint64_t)m * TS + threadIdx.x * IT;
        const uint32_t base = b64 < n ? (uint32_t)b64 : n;
        load_tile_items(x, n, base, v[m], cnt[m]);
    }
    A mine = ALG::identity();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        a[m] = ALG::identity();
#pragma unroll
        for (int j = 0; j < IT; ++j) if ((uint32_t)j < cnt[m]) a[m] = ALG::op(a[m], ALG::lift(v[m][j]));
        mine = ALG::op(mine, a[m]);
    }
    // the link's aggregate goes out before the sub-tile scans: by the time a successor looks back it is there
    const A total = block_fold<ALG>(mine, lds_r);
    if (threadIdx.x == 0) publish_flagged<A>(slots + (size_t)tile * NW, tile == 0 ? ST_PREFIX : ST_AGG, tile == 0 ? ALG::op(seed, total) : total);
    A run = ALG::identity();
#pragma unroll
    for (int m = 0; m < M; ++m) {
        A sub_total;
        A e = block_scan_excl<ALG>(a[m], lds_w, sub_total);
        excl[m] = ALG::op(run, e);                              // prefix inside the link
        run = ALG::op(run, sub_total);
    }
    if (wave_id() == 0) {          // wave 0 looks back, 64 predecessors per step
        const A prefix = lookback<ALG, A>(slots, tile, total, &ctrl[1]);
        if (lane_id() == 0) s_prefix = prefix;
    }
    __syncthreads();
    const A link_prefix = tile == 0 ? seed : s_prefix;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        const uint64_t tb = link_base + (uint64_t)m * TS;
        if (tb >= n) break;
        write_tile<T, ALG, WR>(ALG::op(link_prefix, excl[m]), v[m], cnt[m], (uint32_t)tb + threadIdx.x * IT, (uint32_t)tb, n, out, stage_raw);
        __syncthreads();                                        // the staging buffer is reused by the next sub-tile
    }
}

// ---- shifts / ratios: neighbour element-wise ------------------------------------------------------
template <class T, int OP>
__global__ void __launch_bounds__(SB) shift_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, void* __restrict__ out) {
    using FP = std::conditional_t<sizeof(T) == 4, float, double>;           // GetFPType
    using O = std::conditional_t<OP == AQG_SCAN_RATIOW, FP, T>;
    constexpr int V = 16 / sizeof(T) < 16 / sizeof(O) ? 16 / sizeof(T) : 16 / sizeof(O);   // elements per lane per step
    auto one = [&](uint32_t i, T cur, T prv, T nxt) -> O {
        if constexpr (OP == AQG_SCAN_DELTAS) return i ? (T)(cur - prv) : (T)0;
        else if constexpr (OP == AQG_SCAN_PREV) return i ? prv : cur;
        else if constexpr (OP == AQG_SCAN_NEXT) return i + 1 < n ? nxt : cur;
        else return (FP)(cur / (FP)prv);                                    // ratiow: arr[i] / (FPType)arr[i-w] (prv = arr[0] for i < w)
    };
    const uint32_t nv = n / V;
    const bool aligned = (((uintptr_t)x | (uintptr_t)out) & 15) == 0;
    // the neighbour across a vector boundary comes from the adjacent lane's registers (wave shuffle); only the first / last
    // lane of a wavefront reads it from memory.  The loop bound is wavefront-uniform so that every lane takes part in the shuffle.
    const uint32_t nv_wave = (nv + 63) & ~63u;
    constexpr int U = 1;                                          // vectors per lane per step (4 measured slower: 1.76 vs 1.66 ms per 1e9 rows)
    const uint32_t stride = blockDim.x;
    uint32_t v_lo, v_hi;
    wg_span(nv_wave, v_lo, v_hi, 256);                            // spans of whole wavefronts
    for (uint32_t c0 = v_lo + threadIdx.x; c0 < v_hi && aligned; c0 += stride * U) {
        pack<T, V> cur[U];
        uint32_t base[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = c0 + u * stride;                   // c0 < nv_wave does not bound c: clamp
            live[u] = c < nv;
            base[u] = (live[u] ? c : nv - 1) * V;
            cur[u] = *reinterpret_cast<const pack<T, V>*>(x + base[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = c0 + u * stride;
            T left = shfl_up_t(cur[u].v[V - 1], 1), right = __shfl_down(cur[u].v[0], 1, 64);
            if constexpr (OP == AQG_SCAN_DELTAS || OP == AQG_SCAN_PREV) { if (lane_id() == 0) left = base[u] ? x[base[u] - 1] : cur[u].v[0]; }
            if constexpr (OP == AQG_SCAN_NEXT) { if (lane_id() == 63 || c + 1 >= nv) right = base[u] + V < n ? x[base[u] + V] : cur[u].v[V - 1]; }
            if (!live[u]) continue;
            pack<O, V> o;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const uint32_t i = base[u] + j;
                T prv, nxt = cur[u].v[j];
                if constexpr (OP == AQG_SCAN_RATIOW) prv = i < w ? x[0] : x[i - w];
                else prv = j ? cur[u].v[j - 1] : left;
                if constexpr (OP == AQG_SCAN_NEXT) nxt = j + 1 < V ? cur[u].v[j + 1] : right;
                o.v[j] = one(i, cur[u].v[j], prv, nxt);
            }
            *reinterpret_cast<pack<O, V>*>(static_cast<O*>(out) + base[u]) = o;
        }
    }
    // tail (and the whole column when the buffers are not 16-byte aligned)
    const uint32_t start = aligned ? nv * V : 0;
    for (uint32_t i = start + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        T cur = x[i];
        T prv = OP == AQG_SCAN_RATIOW ? (i < w ? x[0] : x[i - w]) : (i ? x[i - 1] : cur);
        T nxt = i + 1 < n ? x[i + 1] : cur;
        static_cast<O*>(out)[i] = one(i, cur, prv, nxt);
    }
}

// ---- sliding sums: tile + halo in LDS, prefix difference -------------------------------------------
// MODE 0 sumw (LongType out) / 1 avgw (double) / 2 varw / 3 stddevw (intended population variance; see D9)
template <class T, int MODE>
__global__ void __launch_bounds__(SB) window_sum_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, void* __restrict__ out) {
    using ALG = std::conditional_t<(MODE >= 2), dsum_alg<T>, sum_alg<T>>;
    using A = typename ALG::A;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ A lds_w[8];
    __shared__ A lds_w2[8];
    const uint32_t tile_start = blockIdx.x * TS, tile_end = tile_start + TS < n ? tile_start + TS : n;
    // LDS position p <-> row tile_start - H + p, with the halo H = w - 1 rounded up to whole blocks of IT rows; rows before
    // row 0 count as zeros, so the growing prefix of the first w rows needs no special case below
    const uint32_t H = (w - 1 + IT - 1) / IT * IT;
    const uint32_t L = H + TS, nblk = L / IT;
    A* S = reinterpret_cast<A*>(smem_raw);                       // inclusive prefix of x over the extended tile
    A* Q = S + (MODE >= 2 ? L : 0);                              // inclusive prefix of x*x (variance modes)
    // a lane takes blocks of IT consecutive rows straight from HBM (vector load), scans them in registers and writes the
    // prefixes to LDS once; blocks beyond the first SB (the halo's worth) take further rounds with a running carry
    A carry = ALG::identity(), carry2 = ALG::identity();
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
        if (blk < nblk) {
#pragma unroll
            for (int j = 0; j < IT; ++j) { S[blk * IT + j] = ALG::op(excl, loc[j]); if constexpr (MODE >= 2) Q[blk * IT + j] = ALG::op(excl2, loc2[j]); }
        }
        carry = ALG::op(carry, tot);
        if constexpr (MODE >= 2) carry2 = ALG::op(carry2, tot2);
    }
    __syncthreads();
    for (uint32_t i = tile_start + threadIdx.x; i < tile_end; i += SB) {
        const uint32_t idx = i - tile_start + H;
        const uint32_t len = i + 1 < w ? i + 1 : w;               // growing prefix for i < w
        A s = idx >= len ? ALG::sub(S[idx], S[idx - len]) : S[idx];
        if constexpr (MODE == 0) {
            if constexpr (std::is_floating_point_v<T>) static_cast<double*>(out)[i] = s;
            else static_cast<aqg_i128*>(out)[i] = ALG::to_i128(s);
        } else if constexpr (MODE == 1) {
            static_cast<double*>(out)[i] = ALG::to_double(s) / (double)len;
        } else {
            A sq = idx >= len ? ALG::sub(Q[idx], Q[idx - len]) : Q[idx];
            double m = ALG::to_double(s) / (double)len;
            double var = ALG::to_double(sq) / (double)len - m * m;
            if (var < 0) var = 0;
            static_cast<double*>(out)[i] = MODE == 3 ? sqrt(var) : var;
        }
    }
}

// floating inputs, short windows: add the window's elements directly (oldest first) -- no prefix cancellation
template <class T, int MODE>
__global__ void __launch_bounds__(SB) window_direct_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, double* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t len = i + 1 < w ? i + 1 : w;
        double s = 0;
        for (uint32_t j = i + 1 - len; j <= i; ++j) s += (double)x[j];
        out[i] = MODE == 0 ? s : s / (double)len;
    }
}

// ---- sliding min / max: tile + halo in LDS, doubling, eight elements per lane ---------------------------------------------
// M_k[p] = best of the 2^k elements ending at p; M_{k+1}[p] = better(M_k[p], M_k[p - 2^k]); the window of length w is
// better(M_K[p], M_K[p - (w - 2^K)]) with 2^K <= w < 2^(K+1).  A lane works on blocks of eight consecutive positions: the levels
// with 2^k < 8 happen in registers in one step (block + predecessor block), every later level reads its neighbour block with
// 16-byte LDS loads (positions are laid out so that blocks are 16-byte aligned).  Positions before row 0 hold the identity, so
// the growing prefix of the first w rows needs no special case.  (Element-at-a-time doubling: minw(100) ran at 34 % of the
// HBM roofline, bounded by LDS instructions.)
template <class T, bool IS_MAX>
__global__ void __launch_bounds__(SB) window_minmax_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, T* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int E = 8;
    struct alignas(E * sizeof(T) > 16 ? 16 : E * sizeof(T)) blk_t { T v[E]; };
    const uint32_t tile_start = blockIdx.x * TS;
    const uint32_t H = (w - 1 + E - 1) / E * E;                   // halo, rounded up to whole blocks
    const uint32_t L = H + TS;                                    // LDS position p <-> row tile_start - H + p
    T* M0 = reinterpret_cast<T*>(smem_raw);
    T* M1 = M0 + L;
    T ident;
    if constexpr (std::is_floating_point_v<T>) ident = IS_MAX ? -(T)INFINITY : (T)INFINITY;
    else ident = IS_MAX ? dlimits<T>::min() : dlimits<T>::max();
    auto better = [](T a, T b) { if constexpr (IS_MAX) return b > a ? b : a; else return b < a ? b : a; };
    for (uint32_t p = threadIdx.x; p < H; p += SB) {
        const int64_t g = (int64_t)tile_start - (int64_t)H + p;
        M0[p] = g >= 0 ? x[g] : ident;                            // g < tile_start <= n - 1
    }
    {
        const uint32_t g0 = tile_start + threadIdx.x * E;         // TS == SB * E: one block of the tile per lane
        blk_t b;
        if (g0 + E <= n && (reinterpret_cast<uintptr_t>(x + g0) & (alignof(blk_t) - 1)) == 0) b = *reinterpret_cast<const blk_t*>(x + g0);
        else {
#pragma unroll
            for (int q = 0; q < E; ++q) b.v[q] = g0 + q < n ? x[g0 + q] : ident;
        }
        *reinterpret_cast<blk_t*>(M0 + H + threadIdx.x * E) = b;
    }
    __syncthreads();
    uint32_t K = 0;
    while ((2u << K) <= w) ++K;                                    // 2^K <= w < 2^(K+1)
    const uint32_t KA = K < 3 ? K : 3;
    const uint32_t nblk = L / E;
    T* cur = M0; T* nxt = M1;
    if (KA) {                                                      // levels 0 .. KA-1 in registers
        for (uint32_t blk = threadIdx.x; blk < nblk; blk += SB) {
            T a[2 * E];
            const blk_t own = *reinterpret_cast<const blk_t*>(cur + blk * E);
            blk_t prev;
            if (blk) prev = *reinterpret_cast<const blk_t*>(cur + (blk - 1) * E);
#pragma unroll
            for (int q = 0; q < E; ++q) { a[q] = blk ? prev.v[q] : ident; a[E + q] = own.v[q]; }
#pragma unroll
            for (uint32_t k = 0; k < 3; ++k) {
                if (k < KA) {
                    const int d = 1 << k;
#pragma unroll
                    for (int j = 2 * E - 1; j >= d; --j) a[j] = better(a[j], a[j - d]);
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
    for (uint32_t k = KA; k < K; ++k) {                            // 2^k is a multiple of the block: aligned neighbour blocks
        const uint32_t db = (1u << k) / E;
        for (uint32_t blk = threadIdx.x; blk < nblk; blk += SB) {
            blk_t a = *reinterpret_cast<const blk_t*>(cur + blk * E);
            if (blk >= db) {
                const blk_t b = *reinterpret_cast<const blk_t*>(cur + (blk - db) * E);
#pragma unroll
                for (int q = 0; q < E; ++q) a.v[q] = better(a.v[q], b.v[q]);
            }
            *reinterpret_cast<blk_t*>(nxt + blk * E) = a;
        }
        __syncthreads();
        T* t = cur; cur = nxt; nxt = t;
    }
    const uint32_t off = w - (1u << K);                            // second span ends off positions earlier (0 <= off < 2^K, off <= H)
    const uint32_t p0 = H + threadIdx.x * E, g0 = tile_start + threadIdx.x * E;
    if (g0 < n) {
        blk_t a = *reinterpret_cast<const blk_t*>(cur + p0);
        if (off) {
#pragma unroll
            for (int q = 0; q < E; ++q) a.v[q] = better(a.v[q], cur[p0 + q - off]);
        }
        if (g0 + E <= n && (reinterpret_cast<uintptr_t>(out + g0) & (alignof(blk_t) - 1)) == 0) *reinterpret_cast<blk_t*>(out + g0) = a;
        else {
#pragma unroll
            for (int q = 0; q < E; ++q) if (g0 + q < n) out[g0 + q] = a.v[q];
        }
    }
}

// ---- sliding min / max, long windows (w >= 128): van Herk / Gil-Werman ------------------------------------------------------------
// The doubling kernel above pays log2(w) LDS passes over tile + halo (maxw(1000): nine, 34 % of the HBM roofline).  Here the LDS
// positions are cut into segments of w; with F[p] = best of [segment start, p] and B[p] = best of [p, segment end], the window of
// length w ending at p is better(B[p - w + 1], F[p]) -- three passes whatever w is.  A workgroup of 1024 lanes holds C x 1024 positions
// (halo rounded up to a 16-byte vector, then the tile); a lane owns C consecutive positions (C odd: the strided LDS accesses of a
// wavefront fall into distinct banks), scans them in registers in both directions, and the carries between lanes come from
// segmented scans (value + "a segment border lies inside" flag): DPP row shifts inside the 16-lane rows, the three row totals
// through v_readlane, the 16 wavefront totals through LDS and one more 16-lane row scan.  The kernel is bound by VALU issue
// (one wave64 instruction per cycle and CU: ~40 per element leave 70 % of the HBM roofline), which is why the scans avoid
// ds_bpermute shuffles and per-wavefront loops.
template <class T, bool IS_MAX> __device__ inline T vh_better(T a, T b) { if constexpr (IS_MAX) return b > a ? b : a; else return b < a ? b : a; }
template <int CTRL> __device__ inline uint32_t vh_dpp32(uint32_t old, uint32_t src) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false); }
// lane i <- lane i -/+ N of its 16-lane row (CTRL 0x110 + N: row_shr, 0x100 + N: row_shl); lanes without a source keep `old`
template <int CTRL, class T> __device__ inline T vh_dpp(T old, T src) {
    if constexpr (sizeof(T) == 8) {
        const uint64_t o = __builtin_bit_cast(uint64_t, old), v = __builtin_bit_cast(uint64_t, src);
        const uint64_t r = (uint64_t)vh_dpp32<CTRL>((uint32_t)o, (uint32_t)v) | ((uint64_t)vh_dpp32<CTRL>((uint32_t)(o >> 32), (uint32_t)(v >> 32)) << 32);
        return __builtin_bit_cast(T, r);
    } else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, vh_dpp32<CTRL>(__builtin_bit_cast(uint32_t, old), __builtin_bit_cast(uint32_t, src)));
    else return (T)vh_dpp32<CTRL>((uint32_t)(int32_t)old, (uint32_t)(int32_t)src);
}
template <class T> __device__ inline T vh_readlane(T v, int l) {
    if constexpr (sizeof(T) == 8) {
        const uint64_t x = __builtin_bit_cast(uint64_t, v);
        const uint64_t r = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l) << 32);
        return __builtin_bit_cast(T, r);
    } else if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, (uint32_t)__builtin_amdgcn_readlane((int)__builtin_bit_cast(uint32_t, v), l));
    else return (T)__builtin_amdgcn_readlane((int)v, l);
}
// inclusive segmented scan over the 16-lane rows: UP = towards higher lanes (prefix), else towards lower lanes (suffix)
template <class T, bool IS_MAX, bool UP> __device__ inline void vh_row_scan(T& v, uint32_t& f, T ident) {
#define AQG_VH_STEP(N) { const T ov = vh_dpp<(UP ? 0x110 : 0x100) + N>(ident, v); const uint32_t of = vh_dpp32<(UP ? 0x110 : 0x100) + N>(0u, f); \
                         if (!f) v = vh_better<T, IS_MAX>(ov, v); f |= of; }
    AQG_VH_STEP(1) AQG_VH_STEP(2) AQG_VH_STEP(4) AQG_VH_STEP(8)
#undef AQG_VH_STEP
}
template <class T, bool IS_MAX, int C, int WPE>
__global__ void __launch_bounds__(1024, WPE) window_minmax_vh_kernel(const T* __restrict__ x, uint32_t n, uint32_t w, uint32_t tile_rows, T* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr uint32_t NT = 1024, L = C * NT;
    constexpr int V = 16 / sizeof(T);                            // elements per 16-byte vector
    T* A = reinterpret_cast<T*>(smem_raw);                       // [L] inputs, later the results
    T* Bs = A + L;                                               // [L] suffix bests
    __shared__ T wv[2][16];
    __shared__ uint32_t wf[2][16];
    const uint32_t Hp = L - tile_rows;                            // halo (>= w - 1, a multiple of V like tile_rows)
    const uint64_t tile_start = (uint64_t)blockIdx.x * tile_rows;
    T ident;
    if constexpr (std::is_floating_point_v<T>) ident = IS_MAX ? -(T)INFINITY : (T)INFINITY;
    else ident = IS_MAX ? dlimits<T>::min() : dlimits<T>::max();
    struct alignas(16) vec_t { T v[V]; };
    const bool oal = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    // position p <-> row tile_start - Hp + p (x is 16-byte aligned: the host sends other columns to the doubling kernel)
    const int64_t base = (int64_t)tile_start - (int64_t)Hp;
    if (base >= 0 && base + (int64_t)L <= (int64_t)n) {
        for (uint32_t q = threadIdx.x; q < L / V; q += NT) *reinterpret_cast<vec_t*>(A + (size_t)q * V) = *reinterpret_cast<const vec_t*>(x + base + (int64_t)q * V);
    } else {
        for (uint32_t p = threadIdx.x; p < L; p += NT) { const int64_t r = base + p; A[p] = r >= 0 && r < (int64_t)n ? x[r] : ident; }
    }
    __syncthreads();
    const uint32_t p0 = threadIdx.x * C, lane = threadIdx.x & 63, row = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    T a[C], f[C], bk[C];
#pragma unroll
    for (int j = 0; j < C; ++j) a[j] = A[p0 + j];
    const uint32_t r0 = p0 % w;                                  // position inside its segment; w > C: at most one border in the chunk
    // forward: f[j] = best of [max(segment start, p0), p0 + j]; jf = first index that starts a segment (C: none)
    int jf = C;
    f[0] = a[0];
    if (r0 == 0) jf = 0;
#pragma unroll
    for (int j = 1; j < C; ++j) {
        const bool start = r0 + j == w;
        if (start) jf = j;
        f[j] = start ? a[j] : vh_better<T, IS_MAX>(f[j - 1], a[j]);
    }
    // backward: bk[j] = best of [p0 + j, min(segment end, p0 + C - 1)]; je = last index that ends a segment (-1: none)
    int je = -1;
    bk[C - 1] = a[C - 1];
    if (r0 + C == w) je = C - 1;
#pragma unroll
    for (int j = C - 2; j >= 0; --j) {
        const bool end = r0 + j + 1 == w;
        if (end && je < 0) je = j;
        bk[j] = end ? a[j] : vh_better<T, IS_MAX>(bk[j + 1], a[j]);
    }
    // ---- carries inside the wavefront: row scans, then the totals of the rows before (F) / behind (B) ----
    T fv = f[C - 1]; uint32_t ff = jf < C;
    T bv = bk[0]; uint32_t bf = je >= 0;
    vh_row_scan<T, IS_MAX, true>(fv, ff, ident);
    vh_row_scan<T, IS_MAX, false>(bv, bf, ident);
    T fcr = ident, bcr = ident; uint32_t fgr = 0, bgr = 0;          // what enters this lane's row from the other rows
    {
        const T t0 = vh_readlane(fv, 15), t1 = vh_readlane(fv, 31), t2 = vh_readlane(fv, 47);
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)ff, 15), g1 = (uint32_t)__builtin_amdgcn_readlane((int)ff, 31), g2 = (uint32_t)__builtin_amdgcn_readlane((int)ff, 47);
        const T c1 = t0, c2 = g1 ? t1 : vh_better<T, IS_MAX>(c1, t1), c3 = g2 ? t2 : vh_better<T, IS_MAX>(c2, t2);
        fcr = row == 0 ? ident : row == 1 ? c1 : row == 2 ? c2 : c3;
        fgr = row == 0 ? 0u : row == 1 ? g0 : row == 2 ? (g0 | g1) : (g0 | g1 | g2);
        const T u3 = vh_readlane(bv, 48), u2 = vh_readlane(bv, 32), u1 = vh_readlane(bv, 16);
        const uint32_t h3 = (uint32_t)__builtin_amdgcn_readlane((int)bf, 48), h2 = (uint32_t)__builtin_amdgcn_readlane((int)bf, 32), h1 = (uint32_t)__builtin_amdgcn_readlane((int)bf, 16);
        const T d2 = u3, d1 = h2 ? u2 : vh_better<T, IS_MAX>(d2, u2), d0 = h1 ? u1 : vh_better<T, IS_MAX>(d1, u1);
        bcr = row == 3 ? ident : row == 2 ? d2 : row == 1 ? d1 : d0;
        bgr = row == 3 ? 0u : row == 2 ? h3 : row == 1 ? (h3 | h2) : (h3 | h2 | h1);
    }
    // this lane's EXCLUSIVE value inside the wavefront (the lane before / behind; across a row border: what enters the row)
    T fe = vh_dpp<0x111>(fcr, fv); uint32_t fef = vh_dpp32<0x111>(fgr, ff);
    T be = vh_dpp<0x101>(bcr, bv); uint32_t bef = vh_dpp32<0x101>(bgr, bf);
    if ((lane & 15) != 0) { if (!fef) fe = vh_better<T, IS_MAX>(fcr, fe); fef |= fgr; }
    if ((lane & 15) != 15) { if (!bef) be = vh_better<T, IS_MAX>(bcr, be); bef |= bgr; }
    // wavefront totals
    if (lane == 63) { wv[0][wave] = ff ? fv : vh_better<T, IS_MAX>(fcr, fv); wf[0][wave] = ff | fgr; }
    if (lane == 0) { wv[1][wave] = bf ? bv : vh_better<T, IS_MAX>(bcr, bv); wf[1][wave] = bf | bgr; }
    __syncthreads();
    // ---- carries between the wavefronts: the 16 totals, scanned in a 16-lane row; wavefront W takes lane W - 1 (F) / W + 1 (B) ----
    T fc = ident, bc = ident;
    {
        T tv = wv[0][lane & 15]; uint32_t tf = wf[0][lane & 15];
        vh_row_scan<T, IS_MAX, true>(tv, tf, ident);
        const T got = vh_readlane(tv, wave > 0 ? wave - 1 : 0);
        if (wave > 0) fc = got;
        T sv = wv[1][lane & 15]; uint32_t sf = wf[1][lane & 15];
        vh_row_scan<T, IS_MAX, false>(sv, sf, ident);
        const T got2 = vh_readlane(sv, wave < 15 ? wave + 1 : 15);
        if (wave < 15) bc = got2;
    }
    const T fcar = fef ? fe : vh_better<T, IS_MAX>(fc, fe);
    const T bcar = bef ? be : vh_better<T, IS_MAX>(bc, be);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        if (j < jf) f[j] = vh_better<T, IS_MAX>(fcar, f[j]);
        if (j > je) bk[j] = vh_better<T, IS_MAX>(bcar, bk[j]);
        Bs[p0 + j] = bk[j];
    }
    __syncthreads();
    const uint32_t back = w - 1;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint32_t p = p0 + j;
        if (p >= Hp) A[p] = vh_better<T, IS_MAX>(Bs[p - back], f[j]);
    }
    __syncthreads();
    for (uint32_t q = threadIdx.x; q < tile_rows / V; q += NT) {
        const uint64_t g0 = tile_start + (uint64_t)q * V;
        if (g0 >= n) break;
        const vec_t b = *reinterpret_cast<const vec_t*>(A + Hp + (size_t)q * V);
        if (g0 + V <= n && oal) *reinterpret_cast<vec_t*>(out + g0) = b;
        else {
#pragma unroll
            for (int e = 0; e < V; ++e) if (g0 + e < n) out[g0 + e] = b.v[e];
        }
    }
}

// large-window fallback for min/max: doubling passes through HBM (ping-pong), then the two-span combine
template <class T, bool IS_MAX>
__global__ void __launch_bounds__(SB) doubling_pass_kernel(const T* __restrict__ src, T* __restrict__ dst, uint32_t n, uint32_t d) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        T a = src[i];
        if (i >= d) { T b = src[i - d]; if constexpr (IS_MAX) a = b > a ? b : a; else a = b < a ? b : a; }
        dst[i] = a;
    }
}
template <class T, bool IS_MAX>
__global__ void __launch_bounds__(SB) doubling_final_kernel(const T* __restrict__ m, T* __restrict__ out, uint32_t n, uint32_t w, uint32_t span) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t len = i + 1 < w ? i + 1 : w;
        T a = m[i];
        if (len > span) { T b = m[i - (len - span)]; if constexpr (IS_MAX) a = b > a ? b : a; else a = b < a ? b : a; }
        out[i] = a;
    }
}
// large-window fallback for sums: out[i] = S[i] - S[i-len] over a global inclusive prefix (8-byte accumulators)
template <class T, int MODE>
__global__ void __launch_bounds__(SB) prefix_diff_kernel(const typename sum_alg<T>::A* __restrict__ S, uint32_t n, uint32_t w, void* __restrict__ out) {
    using ALG = sum_alg<T>;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t len = i + 1 < w ? i + 1 : w;
        auto s = i >= len ? ALG::sub(S[i], S[i - len]) : S[i];
        if constexpr (MODE == 0) {
            if constexpr (std::is_floating_point_v<T>) static_cast<double*>(out)[i] = s; else static_cast<aqg_i128*>(out)[i] = ALG::to_i128(s);
        } else static_cast<double*>(out)[i] = ALG::to_double(s) / (double)len;
    }
}
// inclusive prefix in accumulator form (used by the fallback above)
template <class T>
__global__ void __launch_bounds__(SB) tile_scan_raw_kernel(const T* __restrict__ x, uint32_t n, const typename sum_alg<T>::A* __restrict__ tile_prefix,
                                                           typename sum_alg<T>::A* __restrict__ out) {
    using ALG = sum_alg<T>;
    using A = typename ALG::A;
    __shared__ A lds_w[8];
    uint32_t base = blockIdx.x * TS + threadIdx.x * IT, cnt;
    T v[IT];
    load_tile_items(x, n, base, v, cnt);
    A a = ALG::identity();
#pragma unroll
    for (int j = 0; j < IT; ++j) if ((uint32_t)j < cnt) a = ALG::op(a, ALG::lift(v[j]));
    A total;
    A run = ALG::op(tile_prefix[blockIdx.x], block_scan_excl<ALG>(a, lds_w, total));
    for (uint32_t j = 0; j < cnt; ++j) { run = ALG::op(run, ALG::lift(v[j])); out[base + j] = run; }
}

// running variance (vars / stddevs): MnX_i = ssq_i - s_i^2/(i+1), value = MnX_i/(i+1) -- from exact prefix sums of
// x and x*x (the reference updates MnX with a floating recurrence, aggregations.h:364-372)
template <class T, bool SD>
__global__ void __launch_bounds__(SB) vars_kernel(const T* __restrict__ x, uint32_t n, const double* __restrict__ tp_s, const double* __restrict__ tp_q,
                                                  double* __restrict__ out) {
    using dalg = dsum_alg<T>;
    __shared__ double lds_w[8];
    __shared__ double lds_w2[8];
    uint32_t base = blockIdx.x * TS + threadIdx.x * IT, cnt;
    T v[IT];
    load_tile_items(x, n, base, v, cnt);
    double a = 0, q = 0;
#pragma unroll
    for (int j = 0; j < IT; ++j) if ((uint32_t)j < cnt) { double d = (double)v[j]; a += d; q += d * d; }
    double t1, t2;
    double rs = tp_s[blockIdx.x] + block_scan_excl<dalg>(a, lds_w, t1);
    double rq = tp_q[blockIdx.x] + block_scan_excl<dalg>(q, lds_w2, t2);
    for (uint32_t j = 0; j < cnt; ++j) {
        double d = (double)v[j];
        rs += d; rq += d * d;
        double cntd = (double)(base + j + 1);
        double var = (rq - rs * rs / cntd) / cntd;
        if (var < 0) var = 0;
        out[base + j] = SD ? sqrt(var) : var;
    }
}
// out[i] = better(out[i], seed): the three-kernel fallback of a seeded running min / max
template <class T, bool IS_MAX> __global__ void __launch_bounds__(SB) apply_seed_kernel(T* __restrict__ out, uint32_t n, T seed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { const T v = out[i]; out[i] = IS_MAX ? (seed > v ? seed : v) : (seed < v ? seed : v); }
}
template <class T, class ALG, int WR>
int run_prefix(aqg_ctx* ctx, const T* x, uint32_t n, void* out, const ScanSeed& seed = ScanSeed{{0, 0}, -0.0, 0}, const T* mm_seed = nullptr) {
    using A = typename ALG::A;
    uint32_t ntiles = aqg_ceil_div(n, TS);
    constexpr int M = chain_m<T>();
    const uint32_t nlinks = (ntiles + M - 1) / M;
    constexpr size_t osz = WR == W_SUMS ? (std::is_floating_point_v<T> ? 8 : 16) : WR == W_AVGS ? 8 : sizeof(T);
    constexpr int PW = std::is_same_v<A, aqg_i128> ? 2 : 1;
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, (size_t)ntiles * (4 + 16 * PW + sizeof(A)) + ((size_t)ntiles / CH + 2) * sizeof(A) + 16384));
    // Measured at 1e9 int32 rows (whole call): mins 1.53 ms chained vs 3.17 ms three-kernel; sums 4.34 vs 4.09; avgs 2.94 vs 2.75.
    // A 4-byte aggregate is handed over in one flagged word; the 8-byte sum of an int32 column takes two words per link and
    // the look-back then costs 1.0 ms at 1e9 rows (the same kernel without any look-back: sums 3.35 ms, avgs 2.30 ms).
    // Packing that sum into one 62-bit word changed nothing (4.43 ms): at two workgroups per CU (the 16-byte results take the
    // registers) the look-back latency itself is exposed, not the number of words.
    constexpr bool use_chain = WR == W_MINS || WR == W_MAXS || WR == W_MAXP;
    if (!use_chain) {
        A *agg3, *chunk_tot;
        AQG_TRY(aqg_ws_get(ctx, ntiles, &agg3));
        AQG_TRY(aqg_ws_get(ctx, (size_t)ntiles / CH + 2, &chunk_tot));
        hipLaunchKernelGGL((tile_reduce_kernel<T, ALG>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, agg3);
        launch_agg_scan<ALG>(ctx, agg3, ntiles, chunk_tot);
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL((tile_scan_kernel<T, ALG, WR>), dim3(ntiles), dim3(SB), (size_t)TS * osz, ctx->stream, x, n, agg3, out, seed);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "prefix scan");
    }
    // ---- single pass (chained tiles) --------------------------------------------------------------------------------
    uint32_t* ctrl;
    uint64_t* slots;
    constexpr int NW = flagged_words<A>();
    AQG_TRY(aqg_ws_get(ctx, 16, &ctrl));
    AQG_TRY(aqg_ws_get(ctx, (size_t)nlinks * NW, &slots));
    AQG_HIP(ctx, hipMemsetAsync(ctrl, 0, 64, ctx->stream));
    AQG_HIP(ctx, hipMemsetAsync(slots, 0, (size_t)nlinks * NW * 8, ctx->stream));
    aqg_kernel_timer_begin(ctx);
    if constexpr (use_chain) {
        T none;                                              // the algebra's identity, spelled on the host
        if constexpr (WR == W_MINS) none = dlimits<T>::max();
        else if constexpr (std::is_floating_point_v<T>) none = -dlimits<T>::max();
        else none = dlimits<T>::min();
        hipLaunchKernelGGL((chained_scan_kernel<T, ALG, WR, M>), dim3(nlinks), dim3(SB), (size_t)TS * osz, ctx->stream, x, n, ctrl, slots, out, mm_seed ? *mm_seed : none);
    }
    aqg_kernel_timer_end(ctx);
    AQG_TRY(aqg_check_launch(ctx, "chained_scan_kernel"));
    uint32_t h[2] = {0, 0};
    AQG_HIP(ctx, hipMemcpyAsync(h, ctrl, 8, hipMemcpyDeviceToHost, ctx->stream));
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[1] == 0) return AQG_OK;
    // ---- a look-back timed out (should not happen): redo with the three-kernel scan -----------------------------------
    A* agg;
    AQG_TRY(aqg_ws_get(ctx, ntiles, &agg));
    hipLaunchKernelGGL((tile_reduce_kernel<T, ALG>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, agg);
    hipLaunchKernelGGL((agg_scan_kernel<ALG>), dim3(1), dim3(SB), 0, ctx->stream, agg, ntiles);
    hipLaunchKernelGGL((tile_scan_kernel<T, ALG, WR>), dim3(ntiles), dim3(SB), (size_t)TS * osz, ctx->stream, x, n, agg, out, seed);
    if constexpr (use_chain) {
        if (mm_seed) hipLaunchKernelGGL((apply_seed_kernel<T, WR != W_MINS>), dim3(aqg_grid(ctx, n, SB, 4, 8)), dim3(SB), 0, ctx->stream, static_cast<T*>(out), n, *mm_seed);
    }
    return aqg_check_launch(ctx, "prefix scan");
}

} // namespace

extern "C" {

int aqg_scan_out_dtype(int op, int t) {
    if (!dt_is_num(t)) return AQG_ERROR;
    switch (op) {
    case AQG_SCAN_SUMS: case AQG_SCAN_SUMW: return aqg_long_type(t);
    case AQG_SCAN_AVGS: case AQG_SCAN_AVGW: case AQG_SCAN_VARS: case AQG_SCAN_STDDEVS: case AQG_SCAN_VARW: case AQG_SCAN_STDDEVW: return AQG_DOUBLE;
    case AQG_SCAN_MINS: case AQG_SCAN_MAXS: case AQG_SCAN_MINW: case AQG_SCAN_MAXW: case AQG_SCAN_DELTAS: case AQG_SCAN_PREV: case AQG_SCAN_NEXT: return t;
    case AQG_SCAN_RATIOW: return aqg_fp_type(t);
    }
    return AQG_ERROR;
}

int aqg_scan_resume(aqg_ctx* ctx, int op, int t, const void* xv, uint32_t n, const void* carry_host16, uint64_t row_offset, void* out) {
    if (!ctx || (!xv && n) || (!out && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_resume: bad argument");
    if (op != AQG_SCAN_SUMS && op != AQG_SCAN_AVGS) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_resume: sums and avgs only (mins / maxs resume from a leading row)");
    AQG_CHECK_ROWS(ctx, n, "aqg_scan_resume");
    if (n == 0) return AQG_OK;
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "scan: the column dtype is not numeric (128-bit results are not inputs)");
    ScanSeed seed{{0, 0}, -0.0, row_offset};
    if (carry_host16) {
        if (t == AQG_FLOAT || t == AQG_DOUBLE) memcpy(&seed.d, carry_host16, 8);
        else memcpy(&seed.i, carry_host16, 16);
    }
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        const T* x = static_cast<const T*>(xv);
        if (op == AQG_SCAN_SUMS) return run_prefix<T, sum_alg<T>, W_SUMS>(ctx, x, n, out, seed);
        return run_prefix<T, sum_alg<T>, W_AVGS>(ctx, x, n, out, seed);
    });
}

// mins / maxs of one row-range shard (aqg_scan_sharded): `seed_host` = the fold (min / max, of the column's element type) of every row
// of the earlier shards, NULL when there is none; op AQG_SCAN_MINS / AQG_SCAN_MAXS, or AQG_SCAN_MINW / AQG_SCAN_MAXW for the running
// form a window at least as long as the whole column degrades to (no numeric_limits seed on the max)
int aqg_scan_minmax_seeded(aqg_ctx* ctx, int op, int t, const void* xv, uint32_t n, const void* seed_host, void* out) {
    if (!ctx || (!xv && n) || (!out && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan_minmax_seeded: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_scan_minmax_seeded");
    if (n == 0) return AQG_OK;
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "scan: the column dtype is not numeric");
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        const T* x = static_cast<const T*>(xv);
        T sd;
        if (seed_host) memcpy(&sd, seed_host, sizeof(T));
        const T* sp = seed_host ? &sd : nullptr;
        switch (op) {
        case AQG_SCAN_MINS: case AQG_SCAN_MINW: return run_prefix<T, min_alg<T>, W_MINS>(ctx, x, n, out, ScanSeed{{0, 0}, -0.0, 0}, sp);
        case AQG_SCAN_MAXS: return run_prefix<T, max_alg<T>, W_MAXS>(ctx, x, n, out, ScanSeed{{0, 0}, -0.0, 0}, sp);
        case AQG_SCAN_MAXW: return run_prefix<T, max_alg<T>, W_MAXP>(ctx, x, n, out, ScanSeed{{0, 0}, -0.0, 0}, sp);
        }
        return AQG_ERR_ARG;
    });
}

int aqg_scan(aqg_ctx* ctx, int op, int t, const void* xv, uint32_t n, uint32_t w, void* out) {
    if (!ctx || (!xv && n) || (!out && n)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan: bad argument");
    if (op < 0 || op > AQG_SCAN_STDDEVW) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan: bad op");
    AQG_CHECK_ROWS(ctx, n, "aqg_scan");
    // w == 0 is undefined in the reference for sumw/avgw/varw (reads ret[-1], divides by zero)
    if (w == 0 && (op == AQG_SCAN_SUMW || op == AQG_SCAN_AVGW || op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW))
        return aqg_fail(ctx, AQG_ERR_ARG, "aqg_scan: window 0 is undefined for sumw/avgw/varw");
    if (n == 0) return AQG_OK;
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "scan: the column dtype is not numeric (128-bit results are not inputs)");
    return aqg_dispatch_num(t, [&](auto tt) -> int {
        using T = typename decltype(tt)::type;
        const T* x = static_cast<const T*>(xv);
        const uint32_t ntiles = aqg_ceil_div(n, TS);
        // shifts: an exact grid, ONE 16-byte vector per lane and workgroup (1e9 rows: deltas 1.59 -> 1.25 ms against a capped
        // grid-stride launch; 4 / 8 / 16 / 32 / 64 vectors per lane: 1.37 / 1.45 / 1.44 / 1.51 / 1.48 ms)
        auto shift_grid = [&](size_t out_size, unsigned per = 1) -> unsigned {
            const size_t v = 16 / (sizeof(T) > out_size ? sizeof(T) : out_size);        // elements per vector, as in shift_kernel
            const uint64_t nv = ((uint64_t)n / v + 63) & ~63ull;
            const uint64_t g = (nv + (uint64_t)SB * per - 1) / ((uint64_t)SB * per);
            return (unsigned)(g < 1 ? 1 : g);
        };
        const unsigned sgrid = shift_grid(sizeof(T));
        switch (op) {
        case AQG_SCAN_SUMS: return run_prefix<T, sum_alg<T>, W_SUMS>(ctx, x, n, out);
        case AQG_SCAN_AVGS: return run_prefix<T, sum_alg<T>, W_AVGS>(ctx, x, n, out);
        case AQG_SCAN_MINS: return run_prefix<T, min_alg<T>, W_MINS>(ctx, x, n, out);
        case AQG_SCAN_MAXS: return run_prefix<T, max_alg<T>, W_MAXS>(ctx, x, n, out);
        case AQG_SCAN_DELTAS: aqg_kernel_timer_begin(ctx); hipLaunchKernelGGL((shift_kernel<T, AQG_SCAN_DELTAS>), dim3(sgrid), dim3(SB), 0, ctx->stream, x, n, w, out); aqg_kernel_timer_end(ctx); return aqg_check_launch(ctx, "deltas");
        case AQG_SCAN_PREV: aqg_kernel_timer_begin(ctx); hipLaunchKernelGGL((shift_kernel<T, AQG_SCAN_PREV>), dim3(sgrid), dim3(SB), 0, ctx->stream, x, n, w, out); aqg_kernel_timer_end(ctx); return aqg_check_launch(ctx, "prev");
        case AQG_SCAN_NEXT: aqg_kernel_timer_begin(ctx); hipLaunchKernelGGL((shift_kernel<T, AQG_SCAN_NEXT>), dim3(sgrid), dim3(SB), 0, ctx->stream, x, n, w, out); aqg_kernel_timer_end(ctx); return aqg_check_launch(ctx, "aggnext");
        case AQG_SCAN_RATIOW: {
            // aggregations.h:172-175: a window not smaller than the column degrades to w = 1
            uint32_t len = n, ww = w;
            if (n <= ww) len = 1;
            ww = ww > len ? len : ww;
            aqg_kernel_timer_begin(ctx);
            hipLaunchKernelGGL((shift_kernel<T, AQG_SCAN_RATIOW>), dim3(shift_grid(sizeof(T) == 4 ? 4 : 8, 4)), dim3(SB), 0, ctx->stream, x, n, ww, out);   // (one vector per lane: 1.59 ms, four: 1.43 ms)
            aqg_kernel_timer_end(ctx);
            return aqg_check_launch(ctx, "ratiow");
        }
        case AQG_SCAN_SUMW: case AQG_SCAN_AVGW: case AQG_SCAN_VARW: case AQG_SCAN_STDDEVW: {
            using A = typename sum_alg<T>::A;
            uint32_t ww = w > n ? n : w;                                        // w clamped to len (:241,264)
            const bool var = op == AQG_SCAN_VARW || op == AQG_SCAN_STDDEVW;
            const size_t ext = (size_t)TS + (ww - 1 + IT - 1) / IT * IT;            // tile + halo rounded up to whole blocks
            size_t lds = var ? ext * sizeof(double) * 2 : ext * sizeof(A);
            if constexpr (std::is_floating_point_v<T>) {
                if (!var && ww <= 64) {
                    aqg_kernel_timer_begin(ctx);
                    if (op == AQG_SCAN_SUMW) hipLaunchKernelGGL((window_direct_kernel<T, 0>), dim3(sgrid), dim3(SB), 0, ctx->stream, x, n, ww, static_cast<double*>(out));
                    else hipLaunchKernelGGL((window_direct_kernel<T, 1>), dim3(sgrid), dim3(SB), 0, ctx->stream, x, n, ww, static_cast<double*>(out));
                    aqg_kernel_timer_end(ctx);
                    return aqg_check_launch(ctx, "window_direct_kernel");
                }
            }
            if (lds <= HALO_MAX_BYTES) {
                auto go = [&](auto kern) -> int {
                    AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    aqg_kernel_timer_begin(ctx);
                    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(SB), lds, ctx->stream, x, n, ww, out);
                    aqg_kernel_timer_end(ctx);
                    return aqg_check_launch(ctx, "window_sum_kernel");
                };
                switch (op) {
                case AQG_SCAN_SUMW: return go(&window_sum_kernel<T, 0>);
                case AQG_SCAN_AVGW: return go(&window_sum_kernel<T, 1>);
                case AQG_SCAN_VARW: return go(&window_sum_kernel<T, 2>);
                default: return go(&window_sum_kernel<T, 3>);
                }
            }
            if (var) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_scan: varw/stddevw window too large for the LDS halo");
            // wide window: global inclusive prefix, then the difference
            AQG_TRY(aqg_ws_reset(ctx));
            AQG_TRY(aqg_ws_ensure(ctx, (size_t)ntiles * sizeof(A) + (size_t)n * sizeof(A) + 8192));
            A *agg, *S;
            AQG_TRY(aqg_ws_get(ctx, ntiles, &agg));
            AQG_TRY(aqg_ws_get(ctx, n, &S));
            hipLaunchKernelGGL((tile_reduce_kernel<T, sum_alg<T>>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, agg);
            hipLaunchKernelGGL((agg_scan_kernel<sum_alg<T>>), dim3(1), dim3(SB), 0, ctx->stream, agg, ntiles);
            hipLaunchKernelGGL((tile_scan_raw_kernel<T>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, agg, S);
            if (op == AQG_SCAN_SUMW) hipLaunchKernelGGL((prefix_diff_kernel<T, 0>), dim3(sgrid), dim3(SB), 0, ctx->stream, S, n, ww, out);
            else hipLaunchKernelGGL((prefix_diff_kernel<T, 1>), dim3(sgrid), dim3(SB), 0, ctx->stream, S, n, ww, out);
            return aqg_check_launch(ctx, "wide window sum");
        }
        case AQG_SCAN_MINW: case AQG_SCAN_MAXW: {
            const bool is_max = op == AQG_SCAN_MAXW;
            // the deque never expires anything when w == 0 or w >= n: plain running min / max (no seed)
            uint32_t ww = (w == 0 || w > n) ? n : w;
            if (ww == n) return is_max ? run_prefix<T, max_alg<T>, W_MAXP>(ctx, x, n, out) : run_prefix<T, min_alg<T>, W_MINS>(ctx, x, n, out);
            // long windows: van Herk / Gil-Werman over C x 1024 LDS positions (C odd; two arrays of them; two workgroups per CU while
            // they take <= 78 KB and the kernel keeps to 64 VGPRs).  Needs a 16-byte aligned column; others take the doubling kernel.
            static const bool vh_off = getenv("AQG_DISABLE_VANHERK") != nullptr;
            static const int vh_c = getenv("AQG_VANHERK_C") ? atoi(getenv("AQG_VANHERK_C")) : 0;
            if (ww >= 128 && !vh_off && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && n >= 64) {
                constexpr uint32_t V = 16 / sizeof(T);
                constexpr int CMAX2 = sizeof(T) <= 4 ? 9 : 3;                           // largest C still run with two workgroups per CU
                const uint32_t hp = (ww - 1 + V - 1) / V * V;
                int c = 0;
                for (int cand : {9, 7, 5, 3}) if (cand <= CMAX2 && (size_t)cand * 2048 * sizeof(T) <= 78 * 1024 && (size_t)cand * 1024 >= 2 * (size_t)hp + V) { c = cand; break; }
                if (!c) for (int cand : {5, 7, 9, 11}) if ((sizeof(T) <= 4 || cand <= 7) && (size_t)cand * 2048 * sizeof(T) <= 150 * 1024 && (size_t)cand * 1024 >= 2 * (size_t)hp + V) { c = cand; break; }
                if (vh_c && (size_t)vh_c * 1024 >= (size_t)hp + V && (size_t)vh_c * 2048 * sizeof(T) <= 150 * 1024 && (sizeof(T) <= 4 || vh_c <= 7)) c = vh_c;
                if (c) {
                    const uint32_t tile_rows = (uint32_t)c * 1024 - hp;
                    const size_t vlds = (size_t)c * 2048 * sizeof(T);
                    const bool two = c <= CMAX2 && vlds <= 78 * 1024;
                    const unsigned vtiles = (unsigned)(((uint64_t)n + tile_rows - 1) / tile_rows);
                    auto gov = [&](auto kern) -> int {
                        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlds));
                        aqg_kernel_timer_begin(ctx);
                        hipLaunchKernelGGL(kern, dim3(vtiles), dim3(1024), vlds, ctx->stream, x, n, ww, tile_rows, static_cast<T*>(out));
                        aqg_kernel_timer_end(ctx);
                        return aqg_check_launch(ctx, "window_minmax_vh_kernel");
                    };
                    auto byc = [&](auto mx) -> int {
                        constexpr bool MX = decltype(mx)::value;
                        if constexpr (sizeof(T) <= 4) {
                            switch (c) {
                            case 3: return gov(&window_minmax_vh_kernel<T, MX, 3, 8>);
                            case 5: return gov(&window_minmax_vh_kernel<T, MX, 5, 8>);
                            case 7: return two ? gov(&window_minmax_vh_kernel<T, MX, 7, 8>) : gov(&window_minmax_vh_kernel<T, MX, 7, 4>);
                            case 9: return two ? gov(&window_minmax_vh_kernel<T, MX, 9, 8>) : gov(&window_minmax_vh_kernel<T, MX, 9, 4>);
                            default: return gov(&window_minmax_vh_kernel<T, MX, 11, 4>);
                            }
                        } else {
                            switch (c) {
                            case 3: return gov(&window_minmax_vh_kernel<T, MX, 3, 8>);
                            case 5: return gov(&window_minmax_vh_kernel<T, MX, 5, 4>);
                            default: return gov(&window_minmax_vh_kernel<T, MX, 7, 4>);
                            }
                        }
                    };
                    return is_max ? byc(std::true_type{}) : byc(std::false_type{});
                }
            }
            size_t lds = (size_t)(TS + (ww - 1 + 7) / 8 * 8) * sizeof(T) * 2;
            if (lds <= HALO_MAX_BYTES) {
                auto go = [&](auto kern) -> int {
                    AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    aqg_kernel_timer_begin(ctx);
                    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(SB), lds, ctx->stream, x, n, ww, static_cast<T*>(out));
                    aqg_kernel_timer_end(ctx);
                    return aqg_check_launch(ctx, "window_minmax_kernel");
                };
                return is_max ? go(&window_minmax_kernel<T, true>) : go(&window_minmax_kernel<T, false>);
            }
            // wide window: doubling passes through HBM
            AQG_TRY(aqg_ws_reset(ctx));
            AQG_TRY(aqg_ws_ensure(ctx, (size_t)n * sizeof(T) * 2 + 8192));
            T *b0, *b1;
            AQG_TRY(aqg_ws_get(ctx, n, &b0));
            AQG_TRY(aqg_ws_get(ctx, n, &b1));
            uint32_t K = 0;
            while ((2u << K) <= ww && K < 31) ++K;
            const T* src = x;
            T* dst = b0;
            for (uint32_t k = 0; k < K; ++k) {
                if (is_max) hipLaunchKernelGGL((doubling_pass_kernel<T, true>), dim3(sgrid), dim3(SB), 0, ctx->stream, src, dst, n, 1u << k);
                else hipLaunchKernelGGL((doubling_pass_kernel<T, false>), dim3(sgrid), dim3(SB), 0, ctx->stream, src, dst, n, 1u << k);
                src = dst;
                dst = dst == b0 ? b1 : b0;
            }
            if (is_max) hipLaunchKernelGGL((doubling_final_kernel<T, true>), dim3(sgrid), dim3(SB), 0, ctx->stream, src, static_cast<T*>(out), n, ww, 1u << K);
            else hipLaunchKernelGGL((doubling_final_kernel<T, false>), dim3(sgrid), dim3(SB), 0, ctx->stream, src, static_cast<T*>(out), n, ww, 1u << K);
            return aqg_check_launch(ctx, "wide window min/max");
        }
        case AQG_SCAN_VARS: case AQG_SCAN_STDDEVS: {
            AQG_TRY(aqg_ws_reset(ctx));
            AQG_TRY(aqg_ws_ensure(ctx, (size_t)ntiles * 16 + 8192));
            double *a1, *a2;
            AQG_TRY(aqg_ws_get(ctx, ntiles, &a1));
            AQG_TRY(aqg_ws_get(ctx, ntiles, &a2));
            hipLaunchKernelGGL((tile_reduce_kernel<T, dsum_alg<T>>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, a1);
            hipLaunchKernelGGL((tile_reduce_kernel<T, sq_alg<T>>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, a2);
            hipLaunchKernelGGL((agg_scan_kernel<dsum_alg<T>>), dim3(1), dim3(SB), 0, ctx->stream, a1, ntiles);
            hipLaunchKernelGGL((agg_scan_kernel<dsum_alg<T>>), dim3(1), dim3(SB), 0, ctx->stream, a2, ntiles);
            if (op == AQG_SCAN_VARS) hipLaunchKernelGGL((vars_kernel<T, false>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, a1, a2, static_cast<double*>(out));
            else hipLaunchKernelGGL((vars_kernel<T, true>), dim3(ntiles), dim3(SB), 0, ctx->stream, x, n, a1, a2, static_cast<double*>(out));
            return aqg_check_launch(ctx, "vars");
        }
        }
        return AQG_ERR_ARG;
    });
}

} // extern "C"

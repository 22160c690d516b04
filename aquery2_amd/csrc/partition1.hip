// partition1.hip -- high-cardinality group-by in ONE partition level (h2o Q3 / Q5 / Q7 at 1e7 groups): replaces AQHashTable's
// robin-hood build (reference server/hasher.h:146-199, server/unordered_dense.h:1117-1147) and the generated per-group loop
// (engine/ast.py:722-789) where the groups do not fit one workgroup's LDS.
//
// Round 1 partitioned {key, row id, values} in two MSD levels of <= 8 bits: every level reads and writes all planes, ~100 B
// moved per row for 16 algorithmic (h2o Q5).  This plan moves them ONCE:
//   p1_hist     one workgroup per CHUNK of rows (a multiple of the 32768-row tile): bin counts of the chunk, [bin][chunk]
//   scan        exclusive scan of that matrix = start of every (bin, chunk) run
//   p1_scatter  one workgroup per chunk walks its tiles carrying the running bin cursors in LDS: rows are ranked inside their
//               bins with returning LDS atomics, each dword plane is staged bin-major in LDS (128 KB) and streamed out, so a
//               tile writes one run per bin and plane and consecutive tiles of a chunk continue each other's runs
//   p1_agg      one 1024-thread workgroup per partition: open-addressing KEY table {key, dense id} at a low load factor and
//               DENSE accumulator arrays indexed by the id -- the accumulators (28 B per group for Q5) are not multiplied by the
//               table's slack, which is what lets a partition hold ~3700 groups in 150 KB and 1e7 groups fit ~2900 bins
// Up to MAXBINS bins (LDS of the scatter: 128 KB of staging + 8 B per bin).  More groups than that: partition.hip (two levels).
#include "partition1_int.hpp"
#include "dense.hpp"

namespace {

constexpr int SR = 32;            // rows per thread and tile
constexpr int PT = SB * SR;       // rows per tile: 32768 (one staged dword plane = 128 KB)
constexpr int MAXPL = 4 + 2 * MAXACC;


// A lane owns SR / 4 groups of FOUR consecutive rows of a tile (one 16-byte load per 4-byte column and group).
__device__ inline uint32_t tile_row(int r) { return (uint32_t)(r >> 2) * (SB * 4) + threadIdx.x * 4 + (r & 3); }
// Loader of one BATCH of R rows of a lane: rows r0 .. r0 + R - 1 of the lane's SR rows of the tile.
// FULL: the tile has all its rows (every tile but the last one of the input): vector loads, no bounds.  Otherwise row by row
// with a clamped index (every load is issued, none sits behind a branch; the caller masks rows >= nrows).
template <bool FULL, class T, int R> __device__ inline void load_rows(const T* __restrict__ p, size_t tile_first, uint32_t nrows, int r0, T (&t)[R]) {
    const T* tp = p + tile_first;
    if constexpr (FULL) {
#pragma unroll
        for (int c = 0; c < R / 4; ++c) __builtin_memcpy(&t[4 * c], tp + tile_row(r0 + 4 * c), 4 * sizeof(T));
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) { const uint32_t o = tile_row(r0 + r); t[r] = tp[o < nrows ? o : nrows - 1]; }
    }
}

// Everything the scatter moves is a DWORD PLANE: one 32-bit word per row, read from a column of 4-byte elements (stride 1) or
// from one half of a column of 8-byte elements (stride 2 dwords), or made from the row index, and written at a dword stride.
// The key is one column of key words (4 or 8 bytes): tuples of several columns and 1- / 2-byte values are packed / widened into
// such columns first (p1_pack_keys_kernel, p1_widen_kernel).
enum : int { PL_LOAD = 0, PL_ROWIDX = 1, PL_PACK = 2 };
struct Plane {
    const uint32_t* src; int src_stride_dw; int src_off_dw;
    uint32_t* dst; int dst_stride_dw; int dst_off_dw;
    int kind;
};
// Narrow integer value columns travelling INSIDE the 4-byte key word (h2o: id6 < 2^24 leaves eight bits; v1 in 1..5 and v2 in 1..15 need
// seven): the first level's key plane is made as key | (v - min) << shift per field (PL_PACK), every later user of the word masks the
// fields off before hashing / comparing and the aggregation unpacks them -- Q5 moves three planes per level instead of five, Q7 two
// instead of four.  Ranges come from a sample of the rows; every row is verified while it is packed and a miss fails the call over to
// the unpacked plan (`flag`).
struct PackSpec { int n; const uint32_t* src[2]; uint32_t min[2], shift[2], fmask[2]; uint32_t kmax; uint32_t* flag; };
struct Planes { int n; Plane p[MAXPL]; PackSpec pk; };

// Chunks of whole tiles cover rows [0, nfull); the rows behind the last whole tile, if any, are one more chunk (the TAIL chunk,
// index nchunks - 1), scattered by its own small kernel so that the main kernel never sees a partial tile.
struct Chunks {
    uint32_t n, nfull, chunk_rows, nchunks, nbins, has_tail;
    // range partitions (the one-level form of BIN_RANGED below): bin = umulhi(min(key - kmin, xmax), rmul); rflag is set by a key outside the sampled domain
    uint32_t ranged, kmin, xmax, rmul; uint32_t* rflag;
};
template <bool K64> __device__ inline uint32_t key_hash(key_t_<K64> k);
template <bool K64, bool RANGED> __device__ inline uint32_t p1_bin(const Chunks& ch, key_t_<K64> key, uint32_t& outside) {
    if constexpr (RANGED && !K64) {
        uint32_t x = (uint32_t)key - ch.kmin;
        outside |= x > ch.xmax ? 1u : 0u;
        x = x < ch.xmax ? x : ch.xmax;
        return __umulhi(x, ch.rmul);
    } else return __umulhi(key_hash<K64>(key), ch.nbins);
}
__device__ inline void chunk_range(const Chunks& ch, uint32_t c, uint64_t& b, uint64_t& e) {
    if (ch.has_tail && c + 1 == ch.nchunks) { b = ch.nfull; e = ch.n; return; }
    b = (uint64_t)c * ch.chunk_rows;
    e = b + ch.chunk_rows < ch.nfull ? b + ch.chunk_rows : ch.nfull;
    if (b > e) b = e;
}

// ---- key tuples of several columns -> one column of key words; 1- / 2-byte values -> dwords -------------------------------------
template <bool K64>
__global__ void __launch_bounds__(256) p1_pack_keys_kernel(KeySpec ks, uint32_t n, key_t_<K64>* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (key_t_<K64>)pack_key(ks, i);
}
__global__ void __launch_bounds__(256) p1_widen_kernel(const void* __restrict__ col, int esz, uint32_t n, uint32_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = esz == 1 ? (uint32_t)static_cast<const uint8_t*>(col)[i] : (uint32_t)static_cast<const uint16_t*>(col)[i];
}

// ---- bin counts of every chunk -----------------------------------------------------------------------------------------------
constexpr int HR = 16;   // rows per thread and step of the histogram pass
template <bool K64, bool RANGED = false>
__global__ void __launch_bounds__(SB) p1_hist_kernel(const key_t_<K64>* __restrict__ keys, Chunks ch, uint32_t* __restrict__ hist /* [bin][chunk] */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);
    for (uint32_t b = threadIdx.x; b < ch.nbins; b += SB) cnt[b] = 0;
    __syncthreads();
    uint64_t cb, ce;
    chunk_range(ch, blockIdx.x, cb, ce);
    constexpr uint64_t STEP = (uint64_t)SB * HR;
    if (!(ch.has_tail && blockIdx.x + 1 == ch.nchunks)) {     // whole tiles, so whole steps: the next step's keys in flight while this one's are counted
        key_t_<K64> cur[HR];
        if (cb < ce) load_rows<true>(keys, cb, (uint32_t)STEP, 0, cur);
        for (uint64_t rb = cb; rb < ce; rb += STEP) {
            key_t_<K64> nxt[HR];
            load_rows<true>(keys, rb + STEP < ce ? rb + STEP : rb, (uint32_t)STEP, 0, nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < HR; ++r) { uint32_t o_ = 0; atomicAdd(&cnt[p1_bin<K64, RANGED>(ch, cur[r], o_)], 1u); }      // (a key outside the sampled domain: the scatter flags it)
#pragma unroll
            for (int r = 0; r < HR; ++r) cur[r] = nxt[r];
        }
    } else for (uint64_t rb = cb; rb < ce; rb += STEP) {      // the tail chunk (less than one tile)
        const uint32_t nrows = ce - rb < STEP ? (uint32_t)(ce - rb) : (uint32_t)STEP;
        key_t_<K64> key[HR];
        load_rows<false>(keys, rb, nrows, 0, key);
#pragma unroll
        for (int r = 0; r < HR; ++r)
            if (tile_row(r) < nrows) { uint32_t o_ = 0; atomicAdd(&cnt[p1_bin<K64, RANGED>(ch, key[r], o_)], 1u); }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < ch.nbins; b += SB) hist[(size_t)b * ch.nchunks + blockIdx.x] = cnt[b];
}

// ---- scatter -------------------------------------------------------------------------------------------------------------------
// One tile: rank the rows inside their bins, then stage and stream out plane after plane.  A lane's SR rows are handled in
// batches of H (all loads of a batch are issued before the first use: memory-level parallelism is what bounds these kernels,
// and 32 rows at once did not fit 128 registers); the keys are not kept: the key planes load them again (from L2).
constexpr int H = 16;
template <bool K64, bool FULL, bool RANGED = false>
__device__ inline void scatter_tile(const key_t_<K64>* __restrict__ keys, const Planes& pl, const Chunks& ch, uint64_t rb, uint32_t nrows, uint32_t* stage, uint32_t* lb, uint32_t* gd, uint32_t* wsum) {
    const uint32_t NB = ch.nbins;
    uint32_t outside = 0;
    for (uint32_t b = threadIdx.x; b <= NB; b += SB) lb[b] = 0;
    __syncthreads();
    uint32_t pos[SR];                                         // (bin << 15) | rank, later the staged position
#pragma unroll
    for (int h = 0; h < SR; h += H) {
        key_t_<K64> key[H];
        load_rows<FULL>(keys, rb, nrows, h, key);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < H; ++r) {
            const uint32_t d = p1_bin<K64, RANGED>(ch, key[r], outside);
            pos[h + r] = FULL || tile_row(h + r) < nrows ? (d << 15) | atomicAdd(&lb[d], 1u) : 0xFFFFFFFFu;
        }
    }
    if constexpr (RANGED) { if (outside) *ch.rflag = 1u; }      // (rows beyond a partial tile repeat its last row: no false alarm)
    __syncthreads();
    {   // exclusive scan of the bin counts: a thread owns bins 4 tid .. 4 tid + 3 (NB <= 4096)
        uint32_t c[4], s = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const uint32_t b = threadIdx.x * 4 + k; c[k] = b < NB ? lb[b] : 0; s += c[k]; }
        const uint32_t incl = wave_scan_incl(s, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t base = incl - s;
        for (int w = 0; w < wave_id(); ++w) base += wsum[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t b = threadIdx.x * 4 + k;
            if (b < NB) { lb[b] = base; gd[b] -= base; }
            base += c[k];
        }
        if (threadIdx.x == SB - 1) lb[NB] = base;             // == nrows
    }
    __syncthreads();
    // staged position of every row (two 15-bit positions per register); the bin of every staged position goes through the
    // stage buffer to the lane that will stream that position out (two 12-bit bin ids per register): the destination of
    // staged position j is j + gd[bin(j)]
    uint32_t ppos[SR / 2];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        uint32_t p = 0;
        if (FULL || pos[r] != 0xFFFFFFFFu) {
            const uint32_t d = pos[r] >> 15;
            p = lb[d] + (pos[r] & 0x7FFFu);
            stage[p] = d;
        }
        if (r & 1) ppos[r >> 1] |= p << 16; else ppos[r >> 1] = p;
    }
    __syncthreads();
    uint32_t pbin[SR / 2];
#pragma unroll
    for (int i = 0; i < SR; ++i) {
        const uint32_t j = i * SB + threadIdx.x;
        const uint32_t d = FULL || j < nrows ? stage[j] : 0;
        if (i & 1) pbin[i >> 1] |= d << 16; else pbin[i >> 1] = d;
    }
#pragma nounroll
    for (int ci = 0; ci < pl.n; ++ci) {
        const Plane& P = pl.p[ci];
        __syncthreads();                       // the previous plane (or the bin ids) has left `stage`
#pragma unroll
        for (int h = 0; h < SR; h += H) {
            uint32_t v[H];
            if (P.kind == PL_ROWIDX) {
#pragma unroll
                for (int r = 0; r < H; ++r) v[r] = (uint32_t)rb + tile_row(h + r);
            } else if (P.kind == PL_PACK) {            // the key word with the narrow value columns in its spare bits, every row verified (PackSpec)
                load_rows<FULL>(P.src, rb, nrows, h, v);
                uint32_t bad = 0;
#pragma unroll
                for (int r = 0; r < H; ++r) bad |= v[r] > pl.pk.kmax ? 1u : 0u;
                for (int f = 0; f < pl.pk.n; ++f) {
                    uint32_t x[H];
                    load_rows<FULL>(pl.pk.src[f], rb, nrows, h, x);
#pragma unroll
                    for (int r = 0; r < H; ++r) { const uint32_t y = x[r] - pl.pk.min[f]; bad |= y > pl.pk.fmask[f] ? 1u : 0u; v[r] |= (y & pl.pk.fmask[f]) << pl.pk.shift[f]; }
                }
                if (bad) *pl.pk.flag = 1u;             // (rows beyond a partial tile repeat its last row: no false alarm)
            } else if (P.src_stride_dw == 1) {
                load_rows<FULL>(P.src, rb, nrows, h, v);
            } else {                                   // one half of every element of an 8-byte column
                const uint32_t* tp = P.src + 2 * rb + P.src_off_dw;
#pragma unroll
                for (int r = 0; r < H; ++r) { const uint32_t o = tile_row(h + r); v[r] = tp[2 * (size_t)(FULL || o < nrows ? o : nrows - 1)]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < H; ++r) if (FULL || tile_row(h + r) < nrows) stage[(ppos[(h + r) >> 1] >> (((h + r) & 1) * 16)) & 0xFFFFu] = v[r];
        }
        __syncthreads();
        uint32_t* dst = P.dst + P.dst_off_dw;
        const uint32_t dstride = (uint32_t)P.dst_stride_dw;
#pragma unroll
        for (int i = 0; i < SR; ++i) {
            const uint32_t j = i * SB + threadIdx.x;
            if (FULL || j < nrows) dst[(size_t)(j + gd[(pbin[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu]) * dstride] = stage[j];
        }
    }
    __syncthreads();
    // cursors for the next tile: cursor + count = (cursor - lb[b]) + lb[b + 1]
    for (uint32_t b = threadIdx.x; b < NB; b += SB) gd[b] += lb[b + 1];
    __syncthreads();
}

template <bool K64, bool RANGED = false>
__global__ void __launch_bounds__(SB) p1_scatter_kernel(const key_t_<K64>* __restrict__ keys, Planes pl, Chunks ch, const uint32_t* __restrict__ hist_scanned) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw);      // [PT] one plane of the tile, bin-major
    uint32_t* lb = stage + PT;                                    // [nbins + 1] counts, then first staged position of every bin
    uint32_t* gd = lb + (ch.nbins + 1);                           // [nbins] between tiles: global cursor; inside: cursor - lb
    __shared__ uint32_t wsum[SB / 64];
    const uint32_t NB = ch.nbins;
    uint64_t cb, ce;
    chunk_range(ch, blockIdx.x, cb, ce);
    for (uint32_t b = threadIdx.x; b < NB; b += SB) gd[b] = hist_scanned[(size_t)b * ch.nchunks + blockIdx.x];
    for (uint64_t rb = cb; rb + PT <= ce; rb += PT) scatter_tile<K64, true, RANGED>(keys, pl, ch, rb, PT, stage, lb, gd, wsum);
}
// the tail chunk: fewer rows than a tile (one workgroup, once per call)
template <bool K64, bool RANGED = false>
__global__ void __launch_bounds__(SB) p1_scatter_tail_kernel(const key_t_<K64>* __restrict__ keys, Planes pl, Chunks ch, const uint32_t* __restrict__ hist_scanned) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw);
    uint32_t* lb = stage + PT;
    uint32_t* gd = lb + (ch.nbins + 1);
    __shared__ uint32_t wsum[SB / 64];
    const uint32_t NB = ch.nbins, c = ch.nchunks - 1;
    for (uint32_t b = threadIdx.x; b < NB; b += SB) gd[b] = hist_scanned[(size_t)b * ch.nchunks + c];
    scatter_tile<K64, false, RANGED>(keys, pl, ch, ch.nfull, ch.n - ch.nfull, stage, lb, gd, wsum);
}

// ==== two levels (more partitions than one level writes well) ==================================================================
// The run a tile writes per bin and plane is (LDS staging bytes / bins) long: 3000 bins leave 44-byte runs, and partial lines are
// what the memory system charges for (measured at 1e9 rows, 5 planes: the one-level scatter takes 8.0 ms at 64 bins, 10.6 at 256,
// 17 at 1024, 31 at 2900 -- the bytes at the L2 / fabric interface only grow from 41 to 53 GB).  Beyond ~1000 partitions the rows
// therefore move TWICE, through <= 64 coarse and then 64 fine bins per coarse one, in runs of a kilobyte:
//   p2_hist      sizes of the P fine partitions (P = 64 B1; LDS counters per workgroup, merged with global atomics)
//   p2_setup     exclusive scan -> partition starts; the write cursors of both levels
//   p2_scatter   level 1: user columns -> buffer set A by coarse bin; level 2: set A -> set B by fine bin inside each coarse
//                partition.  A tile is independent: it ranks its rows inside their bins (LDS atomics), RESERVES its run of every
//                bin with one global atomicAdd on that bin's cursor, stages each plane bin-major and streams it out.  No per-tile
//                histogram, no scan between the levels; rows inside a partition end up in arrival order (the aggregation does not
//                care: first rows come from the carried row ids).
//   p1_agg       as for one level, over the P fine partitions
struct P2Level {
    const uint32_t* seg_start;    // [nseg + 1] rows of every segment (level 1: the whole input; level 2: the coarse partitions)
    const uint32_t* tile_prefix;  // [nseg + 1] first tile of every segment
    uint32_t* cursor;             // write cursors: level 1 [B1], level 2 [P]
    uint32_t nseg, P, shift, mask, nbins, cursor_per_seg;
    uint32_t kclear;              // bits of the key word that are not key (packed value fields): cleared before hashing (0: none)
    uint32_t kmin, xmax;          // BIN_RANGED: the bin is umulhi(key - kmin, P) -- order-preserving bins over a dense key domain
    uint32_t* flag;               // BIN_RANGED: set when a key lies outside [kmin, kmin + xmax] (the range came from a sample)
    // XCD-local segments (null: off).  Workgroups go to the eight XCDs round-robin (blockIdx & 7), every XCD has its own L2, and a tile
    // writes one run per bin at an arbitrary alignment: the partial lines at the ends of neighbouring runs meet in ONE L2 -- and leave it
    // as whole lines -- only if the same XCD writes both.  With this map XCD x takes the segments x, x + 8, ... one after the other:
    // xtp[x * XTP_STRIDE + j] = tiles of its first j segments, xtp[8 * XTP_STRIDE] = 1 when the launch grid covers the fullest XCD.
    const uint32_t* xtp;
    uint32_t* xq;                 // the eight queue heads (one per 128-byte line, zeroed by the setup)
};
constexpr uint32_t XTP_STRIDE = 16;   // (<= 128 segments: up to 16 per XCD)
// how a key word becomes a bin: BIN_RAW umulhi(word, P) (dense group ids, row ids), BIN_HASHED umulhi(hash(word), P), BIN_RANGED
enum : int { BIN_RAW = 0, BIN_HASHED = 1, BIN_RANGED = 2 };

template <int TB> __device__ inline uint32_t trow(int r) { return (uint32_t)(r >> 2) * (TB * 4) + threadIdx.x * 4 + (r & 3); }
template <int TB, bool FULL, class T, int R> __device__ inline void load_rows_t(const T* __restrict__ p, size_t tile_first, uint32_t nrows, int r0, T (&t)[R]) {
    const T* tp = p + tile_first;
    if constexpr (FULL) {
#pragma unroll
        for (int c = 0; c < R / 4; ++c) __builtin_memcpy(&t[4 * c], tp + trow<TB>(r0 + 4 * c), 4 * sizeof(T));
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) { const uint32_t o = trow<TB>(r0 + r); t[r] = tp[o < nrows ? o : nrows - 1]; }
    }
}

constexpr int HB = 16;    // rows per thread and step of the fine histogram
template <bool K64, bool RANGED = false>
__global__ void __launch_bounds__(1024) p2_hist_kernel(const key_t_<K64>* __restrict__ keys, uint32_t n, uint32_t P, uint32_t* __restrict__ ftot, uint32_t kmin = 0, uint32_t xmax = 0) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);
    for (uint32_t b = threadIdx.x; b < (RANGED ? __umulhi(xmax, P) + 1 : P); b += 1024) cnt[b] = 0;
    __syncthreads();
    const uint64_t step = (uint64_t)1024 * HB;
    auto count = [&](const key_t_<K64> (&key)[HB], uint32_t nrows) {
#pragma unroll
        for (int r = 0; r < HB; ++r) {
            if (!(trow<1024>(r) < nrows)) continue;
            uint32_t h;
            if constexpr (RANGED) { h = (uint32_t)key[r] - kmin; h = h < xmax ? h : xmax; }       // (a key outside the sampled range: the scatter flags it)
            else h = key_hash<K64>(key[r]);
            atomicAdd(&cnt[__umulhi(h, P)], 1u);
        }
    };
    // whole steps, the next one's keys in flight while this one's are counted (one code path for the loads: a loader that may take the
    // clamped form is waited for right behind its loads); the partial step at the end of the column by itself
    const uint64_t nfull = (uint64_t)n / step, stride = gridDim.x;
    uint64_t t = blockIdx.x;
    key_t_<K64> cur[HB];
    if (t < nfull) load_rows_t<1024, true>(keys, t * step, (uint32_t)step, 0, cur);
    while (t < nfull) {
        const uint64_t tn = t + stride;
        key_t_<K64> nxt[HB];
        load_rows_t<1024, true>(keys, (tn < nfull ? tn : t) * step, (uint32_t)step, 0, nxt);
        __builtin_amdgcn_sched_barrier(0);
        count(cur, (uint32_t)step);
#pragma unroll
        for (int r = 0; r < HB; ++r) cur[r] = nxt[r];
        t = tn;
    }
    if (blockIdx.x == 0 && nfull * step < n) {
        key_t_<K64> last[HB];
        const uint32_t nrows = (uint32_t)(n - nfull * step);
        load_rows_t<1024, false>(keys, nfull * step, nrows, 0, last);
        count(last, nrows);
    }
    __syncthreads();
    const uint32_t nb = RANGED ? __umulhi(xmax, P) + 1 : P;                                        // (RANGED: P is the multiplier, not the bin count)
    for (uint32_t b = threadIdx.x; b < nb; b += 1024) { const uint32_t c = cnt[b]; if (c) atomicAdd(&ftot[b], c); }
}

// one workgroup: fstart = exclusive scan of the P partition sizes (P <= 4096); segments, tile counts and cursors of both levels
__global__ void __launch_bounds__(1024) p2_setup_kernel(const uint32_t* __restrict__ ftot, uint32_t P, uint32_t n, uint32_t tile_rows,
                                                        uint32_t* __restrict__ fstart /* [P + 1] */, uint32_t* __restrict__ cur2 /* [P] */,
                                                        uint32_t* __restrict__ seg1 /* [2] */, uint32_t* __restrict__ tp1 /* [2] */, uint32_t* __restrict__ cur1 /* [P / 64] */,
                                                        uint32_t* __restrict__ seg2 /* [P / 64 + 1] */, uint32_t* __restrict__ tp2 /* [P / 64 + 1] */,
                                                        uint32_t* __restrict__ xtp /* [8 * XTP_STRIDE + 1] or null */, uint32_t grid_per_xcd) {
    __shared__ uint32_t wsum[16], fs[4097], tcount[65], stile[64], xmax[8];
    uint32_t c[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint32_t b = threadIdx.x * 4 + k; c[k] = b < P ? ftot[b] : 0; s += c[k]; }
    const uint32_t incl = wave_scan_incl(s, OpAdd{}, lane_id());
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t base = incl - s;
    for (int w = 0; w < wave_id(); ++w) base += wsum[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = threadIdx.x * 4 + k;
        if (b < P) { fstart[b] = base; cur2[b] = base; fs[b] = base; }
        base += c[k];
    }
    if (threadIdx.x == 0) { fstart[P] = n; fs[P] = n; seg1[0] = 0; seg1[1] = n; tp1[0] = 0; tp1[1] = (uint32_t)(((uint64_t)n + tile_rows - 1) / tile_rows); }
    __syncthreads();
    const uint32_t B1 = P >> 6;
    if (threadIdx.x <= B1) { seg2[threadIdx.x] = fs[threadIdx.x << 6]; if (threadIdx.x < B1) cur1[threadIdx.x] = fs[threadIdx.x << 6]; }
    if (threadIdx.x < 64) {
        const uint32_t len = threadIdx.x < B1 ? fs[(threadIdx.x + 1) << 6] - fs[threadIdx.x << 6] : 0;
        const uint32_t t = (uint32_t)(((uint64_t)len + tile_rows - 1) / tile_rows);
        const uint32_t ti = wave_scan_incl(t, OpAdd{}, lane_id());
        tp2[threadIdx.x] = ti - t;
        if (threadIdx.x == 63) tcount[0] = ti;
        if (threadIdx.x + 1 == B1) tp2[B1] = ti;
        stile[threadIdx.x] = t;
    }
    if (xtp) {                                                   // level 2 by XCD: segments x, x + 8, ... and their tile prefixes
        __syncthreads();
        if (threadIdx.x < 8) {
            uint32_t run = 0, j = 0;
            for (uint32_t sgm = threadIdx.x; sgm < B1; sgm += 8, ++j) { xtp[threadIdx.x * XTP_STRIDE + j] = run; run += stile[sgm]; }
            xtp[threadIdx.x * XTP_STRIDE + j] = run;
            xmax[threadIdx.x] = run;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t m = 0;
            for (int x = 0; x < 8; ++x) m = xmax[x] > m ? xmax[x] : m;
            xtp[8 * XTP_STRIDE] = B1 <= 8 * (XTP_STRIDE - 1) && m <= grid_per_xcd ? 1u : 0u;     // (a skewed table: the plain walk)
            for (int x = 0; x < 8; ++x) xtp[8 * XTP_STRIDE + 32 + x * 32] = 0u;                   // the queue heads
        }
    }
}

// FULL: grid over all tiles of all segments, whole tiles only.  !FULL: one workgroup per segment takes its last, partial tile.
// NBMAX: bins per level, 128 (the levels of the two-level and wide plans) or 256 (the one-level range plan: FUSE0 keeps a bin in a byte)
template <int TB, int TR, bool K64, bool FULL, int MODE = BIN_HASHED, bool PACK = false, int NBMAX = 128>
__global__ void __launch_bounds__(TB, (FULL && TB * TR * 4 <= 65536) ? 8 : 1) p2_scatter_kernel(const key_t_<K64>* __restrict__ keys, Planes pl, P2Level lv) {
    constexpr int TPT = TB * TR;
    constexpr int HH = TR < 16 ? TR : 16;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw);      // [TPT]
    static_assert(NBMAX == 128 || NBMAX == 256, "bins per level");
    __shared__ uint32_t lb[NBMAX + 1], gd[NBMAX], wtot, wbin[NBMAX / 64];
    __shared__ uint8_t tb[(TPT + 127) / 128];                   // FUSE0: the bin at every 128th staged position
    uint32_t seg, rb, nrows;
    if constexpr (FULL) {
        uint64_t b;
        if (lv.xtp && lv.xtp[8 * XTP_STRIDE]) {                  // XCD x walks its own segments
            // Which XCD this workgroup runs on is read from the hardware (the dispatcher's round-robin holds on some boxes and runs and not on
            // others: with blockIdx & 7 as the XCD, the same binary took 14.9 or 16.9 ms for h2o Q5).  The workgroup pulls the next tile
            // of ITS XCD's list from that list's queue head (one device-scope atomic); a list that has run dry sends it to the next one,
            // so every tile is taken whatever the placement of the workgroups: placement changes the speed only.
            if (threadIdx.x == 0) {
                const uint32_t me = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7u;       // HW_REG_XCC_ID
                uint32_t got = 0xFFFFFFFFu, gx = 0;
                for (uint32_t a = 0; a < 8 && got == 0xFFFFFFFFu; ++a) {
                    const uint32_t xx = (me + a) & 7u;
                    if (xx >= lv.nseg) continue;
                    const uint32_t total = lv.xtp[xx * XTP_STRIDE + ((lv.nseg - xx + 7) >> 3)];
                    uint32_t* head = lv.xq + xx * 32;
                    if (__hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total) continue;
                    const uint32_t k = atomicAdd(head, 1u);
                    if (k < total) { got = k; gx = xx; }
                }
                wtot = got; lb[0] = gx;
            }
            __syncthreads();
            const uint32_t k = wtot, x = lb[0];
            __syncthreads();
            if (k == 0xFFFFFFFFu) return;
            const uint32_t* tp = lv.xtp + x * XTP_STRIDE;
            const uint32_t cnt = (lv.nseg - x + 7) >> 3;
            uint32_t lo = 0, hi = cnt;                           // largest j with tp[j] <= k
            while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (tp[mid] <= k) lo = mid; else hi = mid; }
            seg = x + 8 * lo;
            b = (uint64_t)lv.seg_start[seg] + (uint64_t)(k - tp[lo]) * TPT;
        } else {
        const uint32_t t = blockIdx.x;
        if (t >= lv.tile_prefix[lv.nseg]) return;
        uint32_t lo = 0, hi = lv.nseg;                           // largest segment with tile_prefix[seg] <= t
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (lv.tile_prefix[mid] <= t) lo = mid; else hi = mid; }
        seg = lo;
        b = (uint64_t)lv.seg_start[seg] + (uint64_t)(t - lv.tile_prefix[seg]) * TPT;
        }
        if (b + TPT > lv.seg_start[seg + 1]) return;             // the partial tile of the segment: the tail launch
        rb = (uint32_t)b; nrows = TPT;
    } else {
        seg = blockIdx.x;
        const uint32_t len = lv.seg_start[seg + 1] - lv.seg_start[seg];
        nrows = len % TPT;
        if (!nrows) return;
        rb = lv.seg_start[seg + 1] - nrows;
    }
    const uint32_t NB = lv.nbins;
    if (threadIdx.x <= NBMAX) lb[threadIdx.x] = 0;
    __syncthreads();
    // FUSE0 (4-byte key words; every caller's plane 0 IS the key column): the keys stay in registers from the ranking to the staging of
    // plane 0 -- COUNT the bins (non-returning LDS atomics), scan, then take every row's staged position from its bin's running cursor
    // and store the key word there.  The other form ranks rows while counting and re-reads the keys for their plane: 4 bytes per row
    // and level (h2o Q5 at 1e9 rows: 4 of a level's 28 GB).
    constexpr bool FUSE0 = !K64 && TR <= 16;
    uint32_t pos[TR];                                         // !FUSE0: (bin << 15) | rank; then the staged position
    uint32_t kw[FUSE0 ? TR : 1], dpk[FUSE0 ? (TR + 3) / 4 : 1];   // FUSE0: the key words and their bins (a byte each)
    uint32_t outside = 0;
    if constexpr (FUSE0) {
        load_rows_t<TB, FULL>(reinterpret_cast<const uint32_t*>(keys), rb, nrows, 0, kw);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < (TR + 3) / 4; ++q) dpk[q] = 0;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            uint32_t hw;
            if constexpr (MODE == BIN_HASHED) hw = key_hash<false>(kw[r] & ~lv.kclear);
            else if constexpr (MODE == BIN_RANGED) { hw = (kw[r] & ~lv.kclear) - lv.kmin; outside |= hw > lv.xmax ? 1u : 0u; hw = hw < lv.xmax ? hw : lv.xmax; }
            else hw = kw[r] & ~lv.kclear;
            const uint32_t d = (__umulhi(hw, lv.P) >> lv.shift) & lv.mask;
            dpk[r >> 2] |= d << (8 * (r & 3));
            if (FULL || trow<TB>(r) < nrows) atomicAdd(&lb[d], 1u);
        }
        if constexpr (PACK) {     // plane 0 = the key word with the narrow value columns in its spare bits, every row verified.  Here, while the
            uint32_t bad = 0;     // positions are not live yet: behind the scan the sixteen field values spilled (0.3 extra bytes moved per byte)
#pragma unroll
            for (int r = 0; r < TR; ++r) bad |= kw[r] > pl.pk.kmax ? 1u : 0u;
            for (int f = 0; f < pl.pk.n; ++f) {
                uint32_t x[TR];
                load_rows_t<TB, FULL>(pl.pk.src[f], rb, nrows, 0, x);
#pragma unroll
                for (int r = 0; r < TR; ++r) { const uint32_t y = x[r] - pl.pk.min[f]; bad |= y > pl.pk.fmask[f] ? 1u : 0u; kw[r] |= (y & pl.pk.fmask[f]) << pl.pk.shift[f]; }
            }
            if (bad) *pl.pk.flag = 1u;                        // (rows beyond a partial tile repeat its last row: no false alarm)
        }
    } else {
#pragma unroll
        for (int h = 0; h < TR; h += HH) {
            key_t_<K64> key[HH];
            load_rows_t<TB, FULL>(keys, rb, nrows, h, key);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < HH; ++r) {
                uint32_t hw;
                if constexpr (MODE == BIN_HASHED) hw = key_hash<K64>(K64 ? key[r] : (key_t_<K64>)((uint32_t)key[r] & ~lv.kclear));
                else if constexpr (MODE == BIN_RANGED) { hw = ((uint32_t)key[r] & ~lv.kclear) - lv.kmin; outside |= hw > lv.xmax ? 1u : 0u; hw = hw < lv.xmax ? hw : lv.xmax; }
                else hw = (uint32_t)key[r];
                const uint32_t d = (__umulhi(hw, lv.P) >> lv.shift) & lv.mask;
                pos[h + r] = FULL || trow<TB>(h + r) < nrows ? (d << 15) | atomicAdd(&lb[d], 1u) : 0xFFFFFFFFu;
            }
        }
    }
    if constexpr (MODE == BIN_RANGED) { if (outside) *lv.flag = 1u; }     // (rows beyond a partial tile repeat its last row: no false alarm)
    __syncthreads();
    {   // first wavefronts: exclusive scan of the bin counts; reserve this tile's run of every bin with one atomic per bin
        uint32_t c = 0, incl = 0;
        if (threadIdx.x < NBMAX) {
            c = threadIdx.x < NB ? lb[threadIdx.x] : 0;
            incl = wave_scan_incl(c, OpAdd{}, lane_id());
            if (lane_id() == 63) wbin[wave_id()] = incl;
        }
        __syncthreads();
        if (threadIdx.x < NBMAX) {
            uint32_t excl = incl - c;
            for (int w = 0; w < wave_id(); ++w) excl += wbin[w];
            const uint32_t base = c ? atomicAdd(&lv.cursor[(size_t)seg * lv.cursor_per_seg + threadIdx.x], c) : 0;
            lb[threadIdx.x] = excl;
            gd[threadIdx.x] = base - excl;
        }
    }
    __syncthreads();
    uint32_t dlt[TR];                                         // destination row minus staged position, per output position
    if constexpr (FUSE0) {
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (FULL || trow<TB>(r) < nrows) {
                const uint32_t p = atomicAdd(&lb[(dpk[r >> 2] >> (8 * (r & 3))) & 0xFFu], 1u);     // the bin's cursor: exclusive start -> end
                pos[r] = p;
                stage[p] = kw[r];
            } else pos[r] = 0xFFFFFFFFu;
        }
        __syncthreads();
        // lb[d] is now the END of bin d in the staged order: the bin of an output position = the first bin that ends behind it, found
        // from the bin at the start of the position's block of 128 (tb, written by the bins themselves) in a step or two
        if (threadIdx.x < NB) {
            const uint32_t s0 = threadIdx.x ? lb[threadIdx.x - 1] : 0, e0 = lb[threadIdx.x];
            for (uint32_t blk = (s0 + 127) >> 7; (blk << 7) < e0; ++blk) tb[blk] = (uint8_t)threadIdx.x;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            const uint32_t j = i * TB + threadIdx.x;
            uint32_t b = 0;
            if (FULL || j < nrows) { b = tb[j >> 7]; while (lb[b] <= j) ++b; }
            dlt[i] = FULL || j < nrows ? gd[b] : 0;
        }
        if (pl.p[0].dst) {                                     // (null: the key plane only ranks the rows -- nobody reads it behind this level)
            const Plane& Q = pl.p[0];
            uint32_t* dst = Q.dst + Q.dst_off_dw;
            const uint32_t dstride = (uint32_t)Q.dst_stride_dw;
#pragma unroll
            for (int i = 0; i < TR; ++i) {
                const uint32_t j = i * TB + threadIdx.x;
                if (FULL || j < nrows) dst[(size_t)(j + dlt[i]) * dstride] = stage[j];
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            if (FULL || pos[r] != 0xFFFFFFFFu) {
                const uint32_t d = pos[r] >> 15, p = lb[d] + (pos[r] & 0x7FFFu);
                pos[r] = p;
                stage[p] = d;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TR; ++i) { const uint32_t j = i * TB + threadIdx.x; dlt[i] = FULL || j < nrows ? gd[stage[j]] : 0; }
    }
#pragma nounroll
    for (int ci = FUSE0 ? 1 : 0; ci < pl.n; ++ci) {
        const Plane& Q = pl.p[ci];
        __syncthreads();                       // the previous plane (or the bin ids) has left `stage`
#pragma unroll
        for (int h = 0; h < TR; h += HH) {
            uint32_t v[HH];
            if (Q.kind == PL_ROWIDX) {
#pragma unroll
                for (int r = 0; r < HH; ++r) v[r] = rb + trow<TB>(h + r);
            } else if (Q.src_stride_dw == 1) {
                load_rows_t<TB, FULL>(Q.src, rb, nrows, h, v);
            } else {                                   // one dword of every element of a wider record (halves of 8-byte columns, fields of AoS records)
                const uint32_t* tp = Q.src + (size_t)Q.src_stride_dw * rb + Q.src_off_dw;
#pragma unroll
                for (int r = 0; r < HH; ++r) { const uint32_t o = trow<TB>(h + r); v[r] = tp[(size_t)Q.src_stride_dw * (FULL || o < nrows ? o : nrows - 1)]; }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < HH; ++r) if (FULL || pos[h + r] != 0xFFFFFFFFu) stage[pos[h + r]] = v[r];
        }
        __syncthreads();
        uint32_t* dst = Q.dst + Q.dst_off_dw;
        const uint32_t dstride = (uint32_t)Q.dst_stride_dw;
#pragma unroll
        for (int i = 0; i < TR; ++i) {
            const uint32_t j = i * TB + threadIdx.x;
            if (FULL || j < nrows) dst[(size_t)(j + dlt[i]) * dstride] = stage[j];
        }
    }
}


// ==== tuples wider than 8 bytes (h2o Q10: six int32 keys, nearly every row its own group) ==============================================
// Round 1 sent such rows straight to an HBM table with device-scope atomics (0.54 s per 1e9 rows).  Here the rows are partitioned on a
// 32-bit HASH of the tuple (pw_hash: one pass over the key columns), through up to three levels of the same tile scatter (a first level
// of <= 128 bins, lower levels of 64 or 128; every level: a per-segment histogram pass over the hash plane, a scan, the scatter), until
// a partition has ~1000 ROWS (pw_plan).  The key columns travel as ordinary dword planes.  pw_agg then loads a whole partition into LDS
// and groups it there: an open-addressing table of representative row indices, tuples compared LDS to LDS, accumulators indexed by
// the representative.  The record's key word is the group's first row: emit fetches the key columns through it (the wide-tuple
// convention of groupby.hip).  Sized by rows, not by groups: a tuple that dominates the input overflows its partition and the call
// falls back to the HBM table.
__device__ inline uint32_t pw_seeded(uint32_t h, uint32_t seed) { h ^= seed; h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; return h ^ (h >> 16); }
__global__ void __launch_bounds__(256) pw_hash_kernel(KeySpec ks, uint32_t n, uint32_t seed, uint32_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint32_t h = hash_wide(ks, i); out[i] = seed ? pw_seeded(h, seed) : h; }
}
// the same for key columns that are all 4 bytes wide and 16-byte aligned (h2o Q10): four rows per lane by vector loads, 32-bit
// multiplies only (murmur3's block mix and finaliser -- NOT the chain pw_agg hashes a partition's rows with: the slots inside a
// partition must not follow from the bits that chose the partition).  (the generic pw_hash_kernel: 9.2 ms per 1e9 rows of six columns)
struct Keys32 { const uint32_t* col[MAXKEYS]; int n; };
__device__ inline uint32_t pw_mix32(uint32_t h, uint32_t k, uint32_t c1, uint32_t c2) {
    k *= c1; k = (k << 15) | (k >> 17); k *= c2;
    h ^= k; h = (h << 13) | (h >> 19);
    return h * 5u + 0xE6546B64u;
}
__device__ inline uint32_t pw_fin32(uint32_t h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; return h ^ (h >> 16); }
// TWO 32-bit states with different multipliers: with one, the 1e11 distinct (id1, id2, id3) prefixes of h2o Q10 collide in the
// state after three columns and stay collided, and the partition sizes grow a tail (one partition of 1164 rows where 954 + 6.4
// sigma were allowed: the whole call fell back to the HBM table)
__device__ inline uint32_t pw_hash_row(const uint32_t* k, int nk, uint32_t seed) {
    uint32_t a = 0x2F0B4C9Du ^ seed, b = 0x8A91E5C3u + seed;
    for (int j = 0; j < nk; ++j) { a = pw_mix32(a, k[j], 0xCC9E2D51u, 0x1B873593u); b = pw_mix32(b, k[j], 0x9E3779B1u, 0x85EBCA77u); }
    return pw_fin32(a ^ pw_fin32(b));
}
// PACKW: the key columns also leave this pass PACKED -- column k as the field ((value - min[k]) & mask[k]) << shift[k] of dword plane
// word[k] -- under ranges sampled from the first 2^20 rows; every row is verified here (a miss sets *flag: the call repeats unpacked).
// The six id columns of h2o Q10 (7 + 7 + 24 + 7 + 7 + 24 bits) travel as three dword planes instead of six through every level and
// through pw_agg's LDS; tuple equality on the packed planes IS tuple equality (the map is injective on verified rows), and the result's
// key columns are fetched through the groups' first rows as before.
struct PackW { int nout; uint32_t min[MAXKEYS], mask[MAXKEYS]; int word[MAXKEYS], shift[MAXKEYS]; uint32_t* out[4]; uint32_t* flag; };
// The pass also counts the bins of the FIRST partition level (lc.cnt: <= 128 bins, counted in LDS, flushed with one atomic per bin and workgroup):
// the hash is in a register here, and pn_level_hist read the whole hash plane again for it (1.1 ms per 1e9 rows).
struct Level1Count { uint32_t* cnt; uint32_t P, shift; };
__device__ inline uint32_t pw_level1_bin(uint32_t h, const Level1Count& lc) { return __umulhi(key_hash<false>(h), lc.P) >> lc.shift; }
template <bool PACKW>
__global__ void __launch_bounds__(256) pw_hash32_kernel(Keys32 ks, uint32_t n, uint32_t seed, uint32_t* __restrict__ out, PackW pk, Level1Count lc) {
    const uint32_t nchunk = n >> 2;
    uint32_t bad = 0;
    __shared__ uint32_t lbin[128];
    if (threadIdx.x < 128) lbin[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < nchunk; c += gridDim.x * 256) {
        pack<uint32_t, 4> v[MAXKEYS];
#pragma unroll
        for (int k = 0; k < MAXKEYS; ++k) if (k < ks.n) v[k] = *reinterpret_cast<const pack<uint32_t, 4>*>(ks.col[k] + (size_t)c * 4);
        pack<uint32_t, 4> a, b, h;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a.v[j] = 0x2F0B4C9Du ^ seed; b.v[j] = 0x8A91E5C3u + seed; }
#pragma unroll
        for (int k = 0; k < MAXKEYS; ++k) {
            if (k < ks.n) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { a.v[j] = pw_mix32(a.v[j], v[k].v[j], 0xCC9E2D51u, 0x1B873593u); b.v[j] = pw_mix32(b.v[j], v[k].v[j], 0x9E3779B1u, 0x85EBCA77u); }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) h.v[j] = pw_fin32(a.v[j] ^ pw_fin32(b.v[j]));
        *reinterpret_cast<pack<uint32_t, 4>*>(out + (size_t)c * 4) = h;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(&lbin[pw_level1_bin(h.v[j], lc) & 127u], 1u);
        if constexpr (PACKW) {
#pragma unroll
            for (int k = 0; k < MAXKEYS; ++k) {
                if (k < ks.n) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[k].v[j] -= pk.min[k]; bad |= v[k].v[j] > pk.mask[k] ? 1u : 0u; v[k].v[j] = (v[k].v[j] & pk.mask[k]) << pk.shift[k]; }
                }
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < pk.nout) {
                    pack<uint32_t, 4> w;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w.v[j] = 0;
#pragma unroll
                    for (int k = 0; k < MAXKEYS; ++k) {
                        if (k < ks.n && pk.word[k] == o) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) w.v[j] |= v[k].v[j];
                        }
                    }
                    *reinterpret_cast<pack<uint32_t, 4>*>(pk.out[o] + (size_t)c * 4) = w;
                }
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = ((size_t)nchunk << 2) + threadIdx.x;
        uint32_t k[MAXKEYS];
        for (int j = 0; j < ks.n; ++j) k[j] = ks.col[j][i];
        out[i] = pw_hash_row(k, ks.n, seed);
        atomicAdd(&lbin[pw_level1_bin(out[i], lc) & 127u], 1u);
        if constexpr (PACKW) {
            for (int o = 0; o < pk.nout; ++o) {
                uint32_t w = 0;
                for (int j = 0; j < ks.n; ++j) if (pk.word[j] == o) { const uint32_t y = k[j] - pk.min[j]; bad |= y > pk.mask[j] ? 1u : 0u; w |= (y & pk.mask[j]) << pk.shift[j]; }
                pk.out[o][i] = w;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 128 && lbin[threadIdx.x]) atomicAdd(&lc.cnt[threadIdx.x], lbin[threadIdx.x]);
    if constexpr (PACKW) { if (bad) *pk.flag = 1u; }
}
__global__ void __launch_bounds__(256) pn_gather_strided_kernel(const uint32_t* __restrict__ src, uint32_t stride, uint32_t count, uint32_t* __restrict__ dst) {
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) dst[i] = src[(size_t)i * stride];
}
// tile_prefix[s] = number of tiles of the segments before s; segment s = rows [seg_start[s], seg_start[s + 1])
__global__ void __launch_bounds__(1024) pn_tiles_kernel(const uint32_t* __restrict__ seg_start, uint32_t nseg, uint32_t tile_rows, uint32_t* __restrict__ tile_prefix) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base <= nseg; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < nseg ? (uint32_t)(((uint64_t)(seg_start[i + 1] - seg_start[i]) + tile_rows - 1) / tile_rows) : 0;
        const uint32_t incl = wave_scan_incl(v, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave_id(); ++w) wbase += wsum[w];
        if (i <= nseg) tile_prefix[i] = wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + incl;
        __syncthreads();
    }
}
// bin counts of one level: a tile lies inside ONE segment, so its counts go to cnt[seg * nbins + bin] (LDS histogram, one atomic per bin)
// HASHED = false: the key is a row id and the bin ORDER-PRESERVING, f = umulhi(row, lv.P) with lv.P = floor(partitions * 2^32 / rows)
template <int TB, int TR, bool HASHED>
__global__ void __launch_bounds__(TB) pn_level_hist_kernel(const uint32_t* __restrict__ keys, P2Level lv, uint32_t* __restrict__ cnt) {
    constexpr uint32_t TPT = TB * TR;
    __shared__ uint32_t h[128];
    const uint32_t t = blockIdx.x;
    if (t >= lv.tile_prefix[lv.nseg]) return;
    uint32_t lo = 0, hi = lv.nseg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (lv.tile_prefix[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t seg = lo;
    const uint64_t b = (uint64_t)lv.seg_start[seg] + (uint64_t)(t - lv.tile_prefix[seg]) * TPT;
    const uint32_t e = lv.seg_start[seg + 1];
    const uint32_t nrows = b + TPT > e ? (uint32_t)(e - b) : TPT;
    if (threadIdx.x < 128) h[threadIdx.x] = 0;
    __syncthreads();
    {   // the tile's TR rows of a lane loaded together (a row at a time: TR memory latencies per tile)
        static_assert(TR % 4 == 0, "whole groups of four rows");
        uint32_t key[TR];
        if (nrows == TPT) load_rows_t<TB, true>(keys, (size_t)b, nrows, 0, key); else load_rows_t<TB, false>(keys, (size_t)b, nrows, 0, key);
#pragma unroll
        for (int r = 0; r < TR; ++r)
            if (trow<TB>(r) < nrows) atomicAdd(&h[(__umulhi(HASHED ? key_hash<false>(key[r]) : key[r], lv.P) >> lv.shift) & lv.mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x < lv.nbins && h[threadIdx.x]) atomicAdd(&cnt[(size_t)seg * lv.nbins + threadIdx.x], h[threadIdx.x]);
}

struct WideIn {
    int nkd;                                  // key dwords per row
    const uint32_t* kplane[2 * MAXKEYS];      // partitioned key planes
    const uint32_t* rows;                     // partitioned global row ids
    const void* vcol[MAXACC]; int vesz[MAXACC];   // partitioned value arrays per accumulator (null: the row id)
};
constexpr uint32_t WEMPTY = 0xFFFFu, WEMPTY32 = 0xFFFFFFFFu;
// one workgroup of NT threads per partition (grid-stride); R <= 3 NT = row capacity.  LDS: acc u64[NACC][R] | keys u32[nkd][R] |
// first u32[R] | count u32[R] | table u16[2R] | rep u16[R]; a row's id and values stay in the registers of the thread that loaded it.
// The phases are separated by barriers and each is a chain of LDS round trips, so the kernel lives on workgroups per CU: the plan
// sizes a partition for three workgroups of 512 threads where the level structure allows it (pw_plan).
template <int NACC, int NT>
__global__ void __launch_bounds__(NT, NT == 512 ? 6 : 4) pw_agg_kernel(WideIn in, AccSpec as, AggOps ops, const uint32_t* __restrict__ pstart, uint32_t nparts, uint32_t ntotal,
                                                                        uint32_t R, int need_count, GTable out, uint32_t out_cap, uint8_t* __restrict__ dmark, uint32_t* __restrict__ dcount, int mode, int lazy_vals) {
    constexpr int RPT = 3;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t* lacc = reinterpret_cast<uint64_t*>(smem_raw);                    // [NACC][R]
    uint32_t* lkey = reinterpret_cast<uint32_t*>(lacc + (size_t)NACC * R);     // [nkd][R]
    uint32_t* lfirst = lkey + (size_t)in.nkd * R;
    uint32_t* lcount = lfirst + R;
    uint32_t* table = lcount + R;                                              // [2R] slot -> the row that represents the slot's tuple
    uint16_t* rep = reinterpret_cast<uint16_t*>(table + 2 * R);                // [R]
    __shared__ uint32_t lemit, gbase, ngrp;
    const uint32_t T = 2 * R;
    for (uint32_t part = blockIdx.x; part < nparts; part += gridDim.x) {
        const uint32_t b = pstart[part], e = part + 1 < nparts ? pstart[part + 1] : ntotal;
        const uint32_t m = e - b;
        if (!m) continue;
        if (m > R) { if (threadIdx.x == 0) { out.flags[0] = 1; out.flags[4] = part; out.flags[5] = m; } continue; }   // a partition larger than LDS holds: the host falls back (flags 4, 5: which, how large)
        if (mode == 2 && !dmark[part]) continue;                                // second launch: only the partitions the first one put off
        const bool lazy = mode == 1 && lazy_vals;
        uint32_t myrow[RPT];
        uint64_t myval[NACC > 0 ? NACC : 1][RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = threadIdx.x + q * NT;
            myrow[q] = 0;
            if (i < m) {
                // all key dwords of the row in flight together (a loop over a run-time number of planes waits for every load before
                // the LDS store behind it: 6 planes x 3 rows = 18 memory latencies in a row per partition)
                uint32_t kv[2 * MAXKEYS];
                _Pragma("unroll") for (int k = 0; k < 2 * MAXKEYS; ++k) if (k < in.nkd) kv[k] = in.kplane[k][b + i];
                _Pragma("unroll") for (int k = 0; k < 2 * MAXKEYS; ++k) if (k < in.nkd) lkey[(size_t)k * R + i] = kv[k];
                if (!lazy) {
                    lfirst[i] = NOROW; lcount[i] = 0;
                    _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * R + i] = acc_init(as.kind[a]);
                    myrow[q] = in.rows[b + i];
                    _Pragma("unroll") for (int a = 0; a < NACC; ++a)
                        myval[a][q] = !in.vcol[a] ? (uint64_t)myrow[q] : in.vesz[a] == 4 ? (uint64_t)static_cast<const uint32_t*>(in.vcol[a])[b + i] : static_cast<const uint64_t*>(in.vcol[a])[b + i];
                }
            }
        }
        for (uint32_t s = threadIdx.x; s < T; s += NT) table[s] = WEMPTY32;
        if (threadIdx.x == 0) { lemit = 0; ngrp = 0; }
        __syncthreads();
        // representative of every row: the first row index that claimed the slot of an equal tuple
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < m; i += NT) {
            uint32_t h = 0x9E3779B1u;
            for (int k = 0; k < in.nkd; ++k) h = (h ^ lkey[(size_t)k * R + i]) * 0x85EBCA6Bu;
            uint32_t s = __umulhi(h ^ (h >> 15), T);
            uint32_t r = WEMPTY;
            for (uint32_t step = 0; step < T; ++step) {
                uint32_t cur = *reinterpret_cast<volatile uint32_t*>(&table[s]);
                if (cur == WEMPTY32) {                                          // claim the slot for this row (a plain 32-bit compare-and-swap: the 16-bit
                    const uint32_t got = atomicCAS(&table[s], WEMPTY32, i);     //  slots of rounds 2 - 3 took a read-modify-write loop on the pair holding them)
                    cur = got == WEMPTY32 ? i : got;
                }
                bool eq = cur == i;
                if (!eq) { eq = true; for (int k = 0; k < in.nkd && eq; ++k) eq = lkey[(size_t)k * R + cur] == lkey[(size_t)k * R + i]; }
                if (eq) { r = cur; break; }
                s = s + 1 == T ? 0 : s + 1;
            }
            rep[i] = (uint16_t)r;
            mine += r == i;                                                     // groups = rows that represent themselves
        }
        mine = wave_reduce(mine, OpAdd{});
        if (lane_id() == 0 && mine) atomicAdd(&ngrp, mine);
        __syncthreads();
        // mode 1 (the caller can emit straight from the input rows when EVERY row turns out to be its own group -- h2o Q10, any grouping by a
        // unique key): a partition of distinct rows is put off -- marked and counted, nothing accumulated, no records written (32 of the
        // 56 bytes per row this kernel moves).  All partitions put off: the records were never needed.  Otherwise the host launches mode 2
        // over the marked ones.
        if (mode == 1 && ngrp == m) {
            __syncthreads();                                                    // (everybody has read ngrp: the next partition may clear it)
            if (threadIdx.x == 0) { dmark[part] = 1; atomicAdd(dcount, m); }
            continue;
        }
        if (lazy) {                                                             // (the rows are expected to be distinct: ids and values only now, for the partition that has a duplicate)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const uint32_t i = threadIdx.x + q * NT;
                if (i >= m) continue;
                myrow[q] = in.rows[b + i];
                lfirst[i] = NOROW; lcount[i] = 0;
                _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                    lacc[(size_t)a * R + i] = acc_init(as.kind[a]);
                    myval[a][q] = !in.vcol[a] ? (uint64_t)myrow[q] : in.vesz[a] == 4 ? (uint64_t)static_cast<const uint32_t*>(in.vcol[a])[b + i] : static_cast<const uint64_t*>(in.vcol[a])[b + i];
                }
            }
            __syncthreads();                                                    // (the accumulators of all rows are initialised before the first is used)
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const uint32_t i = threadIdx.x + q * NT;
            if (i >= m) continue;
            const uint32_t r = rep[i];
            atomicMin(&lfirst[r], myrow[q]);
            if (need_count) atomicAdd(&lcount[r], 1u);
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                uint64_t* acc = lacc + (size_t)a * R + r;
                const uint64_t x = myval[a][q];
                switch (ops.opc[a]) {
                case OPC_ADDI_I32: atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)(long long)(int32_t)(uint32_t)x); break;
                case OPC_ADDI_U32: atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)(uint32_t)x); break;
                case OPC_ADDF_F32: atomicAdd(reinterpret_cast<double*>(acc), (double)__uint_as_float((uint32_t)x)); break;
                case OPC_ADDF_F64: atomicAdd(reinterpret_cast<double*>(acc), __builtin_bit_cast(double, x)); break;
                default: acc_apply(acc, as.kind[a], val_operand_bits(as.dt[a] == AQG_NONE ? AQG_UINT32 : as.dt[a], x, as.kind[a], as.square[a], as.part[a])); break;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) gbase = atomicAdd(&out.flags[1], ngrp);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < m; i += NT) {
            if (rep[i] != i) continue;
            const uint32_t g = gbase + atomicAdd(&lemit, 1u);
            if (g >= out_cap) { out.flags[0] = 1; continue; }
            if (out.kb) *out.key_p(g) = (uint64_t)lfirst[i];                   // wide tuples: the key word is a representative ROW (null: the ordering tail takes the first-row plane)
            *out.first_p(g) = lfirst[i];
            *out.count_p(g) = need_count ? lcount[i] : 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) *out.acc_p(a, g) = lacc[(size_t)a * R + i];
        }
        __syncthreads();
    }
}


// ==== ordering a huge group table (G ~ N: h2o Q10) ========================================================================================
// Group ids are ranks of first rows.  For <= 1e7 groups groupby.hip ranks through a bitmap over the rows and gathers the records in id
// order; at 1e9 groups those gathers fetch 600 GB.  Here the RECORDS are ordered instead, with the same tile scatter keyed on the first
// row through an ORDER-PRESERVING bin f = umulhi(first_row, M), M = floor(P * 2^32 / rows): up to three levels of <= 64 bins leave P
// partitions, partition p holding exactly the groups whose first rows fall into its row interval -- so its start is the id of its first
// group -- and few enough of them that groupby.hip's sorted_emit_kernel ranks a partition inside LDS (bitmap of the interval) and
// emits the final columns from there.  (host side: aqg_sorted_tail below)

} // namespace

constexpr int P2_TB = 1024, P2_TR = 16, P2_PT = P2_TB * P2_TR;     // 16384-row tiles: 64 KB of staging, two workgroups per CU

static void p1_geometry(const aqg_ctx* ctx, uint32_t n, Chunks* ch) {
    const uint64_t tiles = (uint64_t)n / PT;                                 // whole tiles
    const uint64_t target = (uint64_t)ctx->num_cu * 4;                       // chunks: four rounds of one workgroup per CU
    uint64_t tiles_per_chunk = (tiles + target - 1) / target;
    if (tiles_per_chunk < 1) tiles_per_chunk = 1;
    ch->n = n;
    ch->nfull = (uint32_t)(tiles * PT);
    ch->has_tail = ch->nfull != n;
    ch->chunk_rows = (uint32_t)(tiles_per_chunk * PT);
    ch->nchunks = (uint32_t)((tiles + tiles_per_chunk - 1) / tiles_per_chunk) + ch->has_tail;
    ch->nbins = 0;
}

// distinct value columns of `as` (an accumulator over the row index has none)
static void p1_val_cols(const AccSpec& as, ValCols* vc) {
    vc->n = 0;
    for (int a = 0; a < as.nacc; ++a) {
        vc->of_acc[a] = -1;
        if (as.dt[a] == AQG_NONE) continue;
        for (int u = 0; u < vc->n; ++u) if (vc->col[u] == as.col[a]) vc->of_acc[a] = u;
        if (vc->of_acc[a] < 0) { vc->col[vc->n] = as.col[a]; vc->dt[vc->n] = as.dt[a]; vc->of_acc[a] = vc->n++; }
    }
}
static bool p1_key_is_column(const KeySpec& ks, int ksz) { return ks.nkeys == 1 && (int)aqg_dtype_size(ks.dt[0]) == ksz; }

// which value columns travel inside the key word (PackSpec): one 4-byte integer key column whose sampled maximum leaves spare bits, 4-byte
// integer value columns whose sampled range fits them (at most two), every accumulator over such a column a plain sum / min / max / square
static int plan_pack(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, const ValCols& vc, PackPlan* pp) {
    memset(pp, 0, sizeof *pp);
    static const bool off = getenv("AQG_DISABLE_PACK") != nullptr;           // A/B measurements only
    if (n < (1u << 22) || !p1_key_is_column(ks, 4) || !(ks.dt[0] == AQG_INT32 || ks.dt[0] == AQG_UINT32)) return AQG_OK;
    KeySpec probe;
    memset(&probe, 0, sizeof probe);
    probe.nkeys = 1; probe.dt[0] = ks.dt[0]; probe.col[0] = ks.col[0];
    int cand[MAXACC], nc = 0;
    for (int u = 0; u < vc.n && probe.nkeys < MAXKEYS && !off; ++u) {
        if (!(vc.dt[u] == AQG_INT32 || vc.dt[u] == AQG_UINT32)) continue;
        bool ok = true;
        for (int a = 0; a < as.nacc; ++a) if (as.col[a] == vc.col[u] && as.dt[a] != AQG_NONE) ok = ok && as.part[a] == 0 && (as.kind[a] == ACC_ADD_I || as.kind[a] == ACC_MIN || as.kind[a] == ACC_MAX);
        if (!ok) continue;
        probe.dt[probe.nkeys] = vc.dt[u]; probe.col[probe.nkeys] = vc.col[u]; ++probe.nkeys;
        cand[nc++] = u;
    }
    long long mins[MAXKEYS], maxs[MAXKEYS];
    bool ok = false;
    AQG_TRY(aqg_key_ranges(ctx, probe, 1u << 20, mins, maxs, &ok, n));       // (a sample spread over the column: every row is verified while it is packed / binned)
    if (!ok) return AQG_OK;
    if (ks.range_known) { mins[0] = ks.range_lo; maxs[0] = ks.range_hi; }     // (the key's range as the caller knows it; the value columns' from the sample)
    pp->have_range = true; pp->key_lo = mins[0]; pp->key_hi = maxs[0]; pp->exact = ks.range_known != 0;
    if (!nc || mins[0] < 0 || maxs[0] >= (1ll << 31)) return AQG_OK;
    auto bits_of = [](unsigned long long v) { int b = 0; while (b < 33 && (1ull << b) <= v) ++b; return b; };
    int used = bits_of((unsigned long long)maxs[0]);
    if (used < 1) used = 1;
    pp->kmax = (uint32_t)((1ull << used) - 1);
    for (int i = 0; i < nc && pp->n < 2; ++i) {
        const unsigned long long range = (unsigned long long)(maxs[1 + i] - mins[1 + i]);
        const int fb = bits_of(range) < 1 ? 1 : bits_of(range);
        if (used + fb > 32) continue;
        const int f = pp->n++;
        pp->col[f] = vc.col[cand[i]]; pp->min[f] = (uint32_t)(long long)mins[1 + i]; pp->shift[f] = (uint32_t)used; pp->fmask[f] = (uint32_t)((1ull << fb) - 1);
        pp->kclear |= pp->fmask[f] << used;
        used += fb;
    }
    return AQG_OK;
}
// Range partitions over a dense key domain (p1_agg_direct_kernel): the sampled domain [key_lo, key_hi] cut into P pieces of at most W key
// values, P a multiple of 64 within the two-level plan's limits.  The slack on both ends (a sample of the first 2^20 rows rarely sees the
// extremes of the column) costs nothing: empty pieces of the domain are empty partitions.

// the direct-indexed aggregation over range partitions (p1_agg_direct_kernel)
static void plan_range(const PackPlan& pp, const AccSpec& as, int need_count, uint32_t parts_hashed, RangePlan* rp) {
    memset(rp, 0, sizeof *rp);
    static const bool off = getenv("AQG_DISABLE_RANGED") != nullptr;         // A/B measurements only
    if (off || !pp.have_range || pp.key_hi < pp.key_lo) return;
    const long long span = pp.key_hi - pp.key_lo + 1, slack = pp.exact ? 0 : span / 64 + 1024;
    // (the key column is int32 or uint32: its values as 64-bit integers; the bins work on the 32-bit difference to kmin, which wraps correctly)
    const long long lo = pp.key_lo - slack, hi = pp.key_hi + slack;
    const unsigned long long D = (unsigned long long)(hi - lo + 1);
    if (D >= (1ull << 32)) return;
    const uint32_t W = p1_direct_capacity(as, need_count);
    if (W < 64) return;
    uint64_t P = (D + (W - 4) - 1) / (W - 4);
    if (P < 256) P = 256;                                                     // every CU gets a partition
    P = (P + 63) & ~63ull;
    for (; P <= AQG_P2_MAXPARTS; P += 64) {
        if (D <= 8 * P) return;                                               // (a domain this small is not this plan's business)
        const uint64_t M = (P << 32) / D;                                     // umulhi(x, M) < P for every x < D
        if (M < 1 || M >= (1ull << 32)) return;
        if ((1ull << 32) / M + 2 <= W) { rp->on = true; rp->kmin = (uint32_t)lo; rp->D = (uint32_t)D; rp->P = (uint32_t)P; rp->M = (uint32_t)M; rp->W = W; return; }
    }
    (void)parts_hashed;
}
size_t aqg_partition1_ws_bytes(const aqg_ctx* ctx, const KeySpec& ks, uint32_t n, const AccSpec& as, uint32_t nbins) {
    const int ksz = ks.total_bytes <= 4 ? 4 : 8;
    ValCols vc;
    p1_val_cols(as, &vc);
    size_t per_row = (size_t)ksz + 4;
    if (!p1_key_is_column(ks, ksz)) per_row += ksz;                          // the packed key column
    for (int u = 0; u < vc.n; ++u) per_row += part_val_bytes(vc.dt[u]) + (aqg_dtype_size(vc.dt[u]) < 4 ? 4 : 0);   // + the widened copy
    Chunks ch;
    p1_geometry(ctx, n, &ch);
    const size_t hcount = (size_t)nbins * ch.nchunks;
    return ((size_t)n + 64) * per_row + 256 * (4 + 2 * MAXACC) + hcount * 4 + (hcount / 2048 + 64) * 4 + 65536;
}

// Partitioned aggregation of (ks, as) over n rows into the compact record table `out` (AoS records, `out_cap` slots,
// flags[1] = number of groups written, flags[0] = overflow).  Needs packed (<= 8 byte) keys and nbins from aqg_partition1_bins.
int aqg_partition1_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t nbins, int need_count, GTable out, uint32_t out_cap, PartRows* pr, int layout, int* ranged) {
    if (nbins < 1 || nbins > AQG_P1_MAXBINS) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "one-level partitioned group-by: 1..3584 bins");
    const int ksz = ks.total_bytes <= 4 ? 4 : 8;
    ValCols vc;
    p1_val_cols(as, &vc);
    // a dense 4-byte key domain: range partitions and the direct-indexed aggregation, as in the two-level plan (no more bins than the
    // hashed plan was given: the workspace is sized by them)
    RangePlan rp;
    PackPlan pp;
    memset(&rp, 0, sizeof rp);
    memset(&pp, 0, sizeof pp);
    if (ranged && *ranged && !pr) {
        AQG_TRY(plan_pack(ctx, ks, as, n, vc, &pp));          // narrow value columns inside the key word, and the sample of the key range
        plan_range(pp, as, need_count, nbins, &rp);
        if (rp.on && rp.P > nbins) rp.on = false;
    }
    if (ranged) *ranged = (rp.on ? 2 : 0) | (pp.n ? 1 : 0);
    if (rp.on) nbins = rp.P;
    Chunks ch;
    p1_geometry(ctx, n, &ch);
    ch.nbins = nbins;
    ch.ranged = rp.on ? 1u : 0u; ch.kmin = rp.kmin; ch.xmax = rp.on ? rp.D - 1 : 0u; ch.rmul = rp.M; ch.rflag = out.flags + 6;

    // the key as ONE column of 4- or 8-byte words: the user's column, or the packed tuple
    const void* keycol = ks.col[0];
    if (!p1_key_is_column(ks, ksz)) {
        void* packed;
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &packed));
        const unsigned g = aqg_grid(ctx, n, 256, 4, 16);
        if (ksz == 4) hipLaunchKernelGGL(p1_pack_keys_kernel<false>, dim3(g), dim3(256), 0, ctx->stream, ks, n, static_cast<uint32_t*>(packed));
        else hipLaunchKernelGGL(p1_pack_keys_kernel<true>, dim3(g), dim3(256), 0, ctx->stream, ks, n, static_cast<uint64_t*>(packed));
        keycol = packed;
    }
    void *pkeys, *prows, *pvals[MAXACC];
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &pkeys));
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &prows));
    const void* vsrc[MAXACC];
    for (int u = 0; u < vc.n; ++u) {
        pvals[u] = nullptr; vsrc[u] = nullptr;
        if (pack_field_of(pp, vc.col[u]) >= 0) continue;                     // travels in the key word
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * part_val_bytes(vc.dt[u]), &pvals[u]));
        vsrc[u] = vc.col[u];
        const int esz = (int)aqg_dtype_size(vc.dt[u]);
        if (esz < 4) {                                                       // 1- / 2-byte values travel as dwords
            void* wide;
            AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &wide));
            hipLaunchKernelGGL(p1_widen_kernel, dim3(aqg_grid(ctx, n, 256, 4, 16)), dim3(256), 0, ctx->stream, vc.col[u], esz, n, static_cast<uint32_t*>(wide));
            vsrc[u] = wide;
        }
    }
    Planes pl;
    memset(&pl, 0, sizeof pl);
    auto add = [&](int kind, const void* s, int sstride, int soff, void* d, int dstride, int doff) {
        Plane& P = pl.p[pl.n++];
        P.kind = kind; P.src = static_cast<const uint32_t*>(s); P.src_stride_dw = sstride; P.src_off_dw = soff;
        P.dst = static_cast<uint32_t*>(d); P.dst_stride_dw = dstride; P.dst_off_dw = doff;
    };
    if (ksz == 4 && pp.n) {
        add(PL_PACK, keycol, 1, 0, pkeys, 1, 0);
        pl.pk.n = pp.n; pl.pk.kmax = pp.kmax; pl.pk.flag = out.flags + 6;
        for (int f = 0; f < pp.n; ++f) { pl.pk.src[f] = static_cast<const uint32_t*>(pp.col[f]); pl.pk.min[f] = pp.min[f]; pl.pk.shift[f] = pp.shift[f]; pl.pk.fmask[f] = pp.fmask[f]; }
    }
    else if (ksz == 4) add(PL_LOAD, keycol, 1, 0, pkeys, 1, 0);
    else { add(PL_LOAD, keycol, 2, 0, pkeys, 2, 0); add(PL_LOAD, keycol, 2, 1, pkeys, 2, 1); }
    add(PL_ROWIDX, nullptr, 0, 0, prows, 1, 0);
    for (int u = 0; u < vc.n; ++u) {
        if (!pvals[u]) continue;                                             // (packed)
        if (part_val_bytes(vc.dt[u]) == 4) add(PL_LOAD, vsrc[u], 1, 0, pvals[u], 1, 0);
        else { add(PL_LOAD, vsrc[u], 2, 0, pvals[u], 2, 0); add(PL_LOAD, vsrc[u], 2, 1, pvals[u], 2, 1); }
    }
    // Range partitions of <= 256 bins go through the tile scatter of the two-level plan, as its only level: whole-column bin counts, write
    // cursors instead of per-chunk histograms, 16384-row tiles with the keys kept in registers, two workgroups per CU so that one tile's
    // loads run under the other's stores (h2o Q5 / Q7 at 1e9 rows, 1e6 groups: this kernel's own 32768-row tiles, one workgroup per CU,
    // moved 3.9 / 2.8 TB/s)
    static const bool cursor_off = getenv("AQG_DISABLE_P1_CURSORS") != nullptr;             // A/B measurements only
    if (rp.on && ksz == 4 && nbins <= 256 && !cursor_off) {
        const uint32_t P = nbins;
        uint32_t *ftot, *fstart, *cur, *seg1, *tp1, *cur1, *seg2, *tp2;
        AQG_TRY(aqg_ws_get(ctx, (size_t)P, &ftot));
        AQG_TRY(aqg_ws_get(ctx, (size_t)P + 1, &fstart));
        AQG_TRY(aqg_ws_get(ctx, (size_t)P, &cur));
        AQG_TRY(aqg_ws_get(ctx, 2, &seg1));
        AQG_TRY(aqg_ws_get(ctx, 2, &tp1));
        AQG_TRY(aqg_ws_get(ctx, 64, &cur1));
        AQG_TRY(aqg_ws_get(ctx, 65, &seg2));
        AQG_TRY(aqg_ws_get(ctx, 65, &tp2));
        AQG_HIP(ctx, hipMemsetAsync(ftot, 0, (size_t)P * 4, ctx->stream));
        const uint32_t* kc = static_cast<const uint32_t*>(keycol);
        hipLaunchKernelGGL((p2_hist_kernel<false, true>), dim3(aqg_grid(ctx, n, 1024, HB, 4)), dim3(1024), (size_t)P * 4, ctx->stream, kc, n, rp.M, ftot, rp.kmin, rp.D - 1);
        hipLaunchKernelGGL(p2_setup_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)ftot, P, n, (uint32_t)P2_PT, fstart, cur, seg1, tp1, cur1, seg2, tp2, (uint32_t*)nullptr, 0u);
        P2Level lv{seg1, tp1, cur, 1u, rp.M, 0u, 0xFFFFFFFFu, P, 0u, 0u, rp.kmin, rp.D - 1, out.flags + 6, nullptr, nullptr};
        const size_t lds = (size_t)P2_PT * 4;
        const unsigned tiles = (unsigned)(((uint64_t)n + P2_PT - 1) / P2_PT);
        auto go = [&](auto packing) -> int {
            constexpr bool PK = decltype(packing)::value;
            AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, true, BIN_RANGED, PK, 256>), lds));
            AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, false, BIN_RANGED, PK, 256>), lds));
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, true, BIN_RANGED, PK, 256>), dim3(tiles), dim3(P2_TB), lds, ctx->stream, kc, pl, lv);
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, false, BIN_RANGED, PK, 256>), dim3(1), dim3(P2_TB), lds, ctx->stream, kc, pl, lv);
            return aqg_check_launch(ctx, "one-level partition scatter (cursors)");
        };
        if (pp.n) AQG_TRY(go(std::true_type{})); else AQG_TRY(go(std::false_type{}));
        return p1_launch_agg_direct(ctx, as, vc, pkeys, prows, pvals, fstart, 1u, n, need_count, out, out_cap, pp.n ? &pp : nullptr, rp);
    }
    const size_t hcount = (size_t)nbins * ch.nchunks;
    uint32_t *hist, *bsum;
    AQG_TRY(aqg_ws_get(ctx, hcount, &hist));
    AQG_TRY(aqg_ws_get(ctx, hcount / 2048 + 64, &bsum));
    const size_t hist_lds = (size_t)nbins * 4;
    const size_t scat_lds = (size_t)PT * 4 + ((size_t)nbins * 2 + 1) * 4;
    const unsigned nmain = ch.nchunks - ch.has_tail;
    auto run = [&](auto k64, auto rng) -> int {
        constexpr bool K = decltype(k64)::value, R = decltype(rng)::value;
        const key_t_<K>* kc = static_cast<const key_t_<K>*>(keycol);
        hipLaunchKernelGGL((p1_hist_kernel<K, R>), dim3(ch.nchunks), dim3(SB), hist_lds, ctx->stream, kc, ch, hist);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, hist, hcount, bsum));
        if (nmain) {
            AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p1_scatter_kernel<K, R>), scat_lds));
            hipLaunchKernelGGL((p1_scatter_kernel<K, R>), dim3(nmain), dim3(SB), scat_lds, ctx->stream, kc, pl, ch, (const uint32_t*)hist);
        }
        if (ch.has_tail) {
            AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p1_scatter_tail_kernel<K, R>), scat_lds));
            hipLaunchKernelGGL((p1_scatter_tail_kernel<K, R>), dim3(1), dim3(SB), scat_lds, ctx->stream, kc, pl, ch, (const uint32_t*)hist);
        }
        return aqg_check_launch(ctx, "one-level partition scatter");
    };
    if (rp.on) AQG_TRY(run(std::false_type{}, std::true_type{}));
    else if (ksz == 4) AQG_TRY(run(std::false_type{}, std::false_type{}));
    else AQG_TRY(run(std::true_type{}, std::false_type{}));

    if (rp.on) return p1_launch_agg_direct(ctx, as, vc, pkeys, prows, pvals, hist, ch.nchunks, n, need_count, out, out_cap, pp.n ? &pp : nullptr, rp);
    return p1_launch_agg(ctx, ksz, as, vc, pkeys, prows, pvals, hist, ch.nchunks, nbins, n, need_count, out, out_cap, pr, pp.n ? &pp : nullptr, layout);
}


// ---- two levels: host ---------------------------------------------------------------------------------------------------------------
static uint32_t p2_round_parts(uint32_t parts) { return (parts + 63) & ~63u; }

size_t aqg_partition2_ws_bytes(const aqg_ctx* ctx, const KeySpec& ks, uint32_t n, const AccSpec& as, uint32_t parts) {
    const int ksz = ks.total_bytes <= 4 ? 4 : 8;
    ValCols vc;
    p1_val_cols(as, &vc);
    size_t per_row = 2 * ((size_t)ksz + 4);
    if (!p1_key_is_column(ks, ksz)) per_row += ksz;
    for (int u = 0; u < vc.n; ++u) per_row += 2 * part_val_bytes(vc.dt[u]) + (aqg_dtype_size(vc.dt[u]) < 4 ? 4 : 0);
    return ((size_t)n + 64) * per_row + 256 * (8 + 4 * MAXACC) + (size_t)p2_round_parts(parts) * 16 + 65536;
}

int aqg_partition2_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t parts, int need_count, GTable out, uint32_t out_cap, PartRows* pr, int* pack, int layout) {
    const int ksz = ks.total_bytes <= 4 ? 4 : 8;
    ValCols vc;
    p1_val_cols(as, &vc);
    PackPlan pp;
    RangePlan rp;
    memset(&pp, 0, sizeof pp);
    memset(&rp, 0, sizeof rp);
    if (pack && *pack && !pr) {
        AQG_TRY(plan_pack(ctx, ks, as, n, vc, &pp));                         // narrow value columns inside the key word (fewer planes per level)
        plan_range(pp, as, need_count, parts, &rp);                          // a dense key domain: range partitions, direct-indexed aggregation
    }
    if (pack) *pack = (pp.n ? 1 : 0) | (rp.on ? 2 : 0);
    const uint32_t P = rp.on ? rp.P : p2_round_parts(parts), B1 = P >> 6;
    if (P < 64 || P > AQG_P2_MAXPARTS) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "two-level partitioned group-by: 64..4096 partitions");
    const void* keycol = ks.col[0];
    if (!p1_key_is_column(ks, ksz)) {
        void* packed;
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &packed));
        const unsigned g = aqg_grid(ctx, n, 256, 4, 16);
        if (ksz == 4) hipLaunchKernelGGL(p1_pack_keys_kernel<false>, dim3(g), dim3(256), 0, ctx->stream, ks, n, static_cast<uint32_t*>(packed));
        else hipLaunchKernelGGL(p1_pack_keys_kernel<true>, dim3(g), dim3(256), 0, ctx->stream, ks, n, static_cast<uint64_t*>(packed));
        keycol = packed;
    }
    void *keysA, *rowsA, *valsA[MAXACC], *keysB, *rowsB, *valsB[MAXACC];
    const void* vsrc[MAXACC];
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &keysA));
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &rowsA));
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &keysB));
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &rowsB));
    for (int u = 0; u < vc.n; ++u) {
        valsA[u] = valsB[u] = nullptr; vsrc[u] = nullptr;
        if (pack_field_of(pp, vc.col[u]) >= 0) continue;                     // travels in the key word
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * part_val_bytes(vc.dt[u]), &valsA[u]));
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * part_val_bytes(vc.dt[u]), &valsB[u]));
        vsrc[u] = vc.col[u];
        const int esz = (int)aqg_dtype_size(vc.dt[u]);
        if (esz < 4) {
            void* wide;
            AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &wide));
            hipLaunchKernelGGL(p1_widen_kernel, dim3(aqg_grid(ctx, n, 256, 4, 16)), dim3(256), 0, ctx->stream, vc.col[u], esz, n, static_cast<uint32_t*>(wide));
            vsrc[u] = wide;
        }
    }
    uint32_t *ftot, *fstart, *cur2, *seg1, *tp1, *cur1, *seg2, *tp2;
    AQG_TRY(aqg_ws_get(ctx, (size_t)P, &ftot));
    AQG_TRY(aqg_ws_get(ctx, (size_t)P + 1, &fstart));
    AQG_TRY(aqg_ws_get(ctx, (size_t)P, &cur2));
    AQG_TRY(aqg_ws_get(ctx, 2, &seg1));
    AQG_TRY(aqg_ws_get(ctx, 2, &tp1));
    AQG_TRY(aqg_ws_get(ctx, 64, &cur1));
    AQG_TRY(aqg_ws_get(ctx, 65, &seg2));
    AQG_TRY(aqg_ws_get(ctx, 65, &tp2));
    uint32_t* xtp;
    AQG_TRY(aqg_ws_get(ctx, 8 * XTP_STRIDE + 32 + 8 * 32, &xtp));     // tile prefixes per XCD | flag | eight queue heads on lines of their own
    AQG_HIP(ctx, hipMemsetAsync(ftot, 0, (size_t)P * 4, ctx->stream));

    auto planes = [&](bool level1) {
        Planes pl;
        memset(&pl, 0, sizeof pl);
        auto add = [&](int kind, const void* s, int sstride, int soff, void* d, int dstride, int doff) {
            Plane& Q = pl.p[pl.n++];
            Q.kind = kind; Q.src = static_cast<const uint32_t*>(s); Q.src_stride_dw = sstride; Q.src_off_dw = soff;
            Q.dst = static_cast<uint32_t*>(d); Q.dst_stride_dw = dstride; Q.dst_off_dw = doff;
        };
        const void* ksrc = level1 ? keycol : keysA;
        void* kdst = level1 ? keysA : keysB;
        if (ksz == 4 && pp.n && level1) {
            add(PL_PACK, ksrc, 1, 0, kdst, 1, 0);
            pl.pk.n = pp.n; pl.pk.kmax = pp.kmax; pl.pk.flag = out.flags + 6;
            for (int f = 0; f < pp.n; ++f) { pl.pk.src[f] = static_cast<const uint32_t*>(pp.col[f]); pl.pk.min[f] = pp.min[f]; pl.pk.shift[f] = pp.shift[f]; pl.pk.fmask[f] = pp.fmask[f]; }
        }
        else if (ksz == 4) add(PL_LOAD, ksrc, 1, 0, kdst, 1, 0);
        else { add(PL_LOAD, ksrc, 2, 0, kdst, 2, 0); add(PL_LOAD, ksrc, 2, 1, kdst, 2, 1); }
        if (level1) add(PL_ROWIDX, nullptr, 0, 0, rowsA, 1, 0); else add(PL_LOAD, rowsA, 1, 0, rowsB, 1, 0);
        for (int u = 0; u < vc.n; ++u) {
            if (!valsA[u]) continue;                                          // (packed)
            const void* vs = level1 ? vsrc[u] : valsA[u];
            void* vd = level1 ? valsA[u] : valsB[u];
            if (part_val_bytes(vc.dt[u]) == 4) add(PL_LOAD, vs, 1, 0, vd, 1, 0);
            else { add(PL_LOAD, vs, 2, 0, vd, 2, 0); add(PL_LOAD, vs, 2, 1, vd, 2, 1); }
        }
        return pl;
    };
    const size_t scat_lds = (size_t)P2_PT * 4;
    const unsigned tiles1 = (unsigned)(((uint64_t)n + P2_PT - 1) / P2_PT), tiles2 = (unsigned)((uint64_t)n / P2_PT) + B1 + 1;
    auto run = [&](auto k64) -> int {
        constexpr bool K = decltype(k64)::value;
        const unsigned hgrid = aqg_grid(ctx, n, 1024, HB, 4);
        if (rp.on) { if constexpr (!K) hipLaunchKernelGGL((p2_hist_kernel<false, true>), dim3(hgrid), dim3(1024), (size_t)P * 4, ctx->stream, static_cast<const uint32_t*>(keycol), n, rp.M, ftot, rp.kmin, rp.D - 1); }
        else hipLaunchKernelGGL((p2_hist_kernel<K>), dim3(hgrid), dim3(1024), (size_t)P * 4, ctx->stream, static_cast<const key_t_<K>*>(keycol), n, P, ftot);
        static const bool xcd_off = getenv("AQG_DISABLE_XCD_MAP") != nullptr;                   // A/B measurements only
        const unsigned xgrid = xcd_off ? 0u : ((tiles2 + 7) / 8 * 5 / 4 + 8);                       // workgroups per XCD of the level-2 launch (a quarter of slack)
        hipLaunchKernelGGL(p2_setup_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)ftot, P, n, (uint32_t)P2_PT, fstart, cur2, seg1, tp1, cur1, seg2, tp2, xcd_off ? (uint32_t*)nullptr : xtp, xgrid);
        // the bin of a key word at both levels: the hash or (range partitions) the offset in the domain, scaled to P fine partitions;
        // level 1 takes the coarse partition (fine >> 6), level 2 the fine one inside it (fine & 63)
        const uint32_t scale = rp.on ? rp.M : P;
        P2Level l1{seg1, tp1, cur1, 1u, scale, 6u, 0xFFFFFFFFu, B1, 0u, 0u, rp.kmin, rp.on ? rp.D - 1 : 0u, out.flags + 6};
        P2Level l2{seg2, tp2, cur2, B1, scale, 0u, 63u, 64u, 64u, pp.kclear, rp.kmin, rp.on ? rp.D - 1 : 0u, out.flags + 6, xcd_off ? (const uint32_t*)nullptr : xtp, xcd_off ? (uint32_t*)nullptr : xtp + 8 * XTP_STRIDE + 32};
        const unsigned grid2 = xcd_off || 8 * xgrid < tiles2 ? tiles2 : 8 * xgrid;                  // (covers the plain walk too, should the setup decline the map)
        auto level = [&](auto mode, auto packing, const key_t_<K>* src, const Planes& pl, const P2Level& lv, unsigned tiles, unsigned tails) -> int {
            constexpr int MODE = decltype(mode)::value;
            constexpr bool PK = decltype(packing)::value;
            if constexpr (K && (MODE != BIN_HASHED || PK)) return AQG_ERR_ARG;               // (8-byte key words are hashed and never packed)
            else {
                AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, K, true, MODE, PK>), scat_lds));
                AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, K, false, MODE, PK>), scat_lds));
                hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, K, true, MODE, PK>), dim3(tiles), dim3(P2_TB), scat_lds, ctx->stream, src, pl, lv);
                hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, K, false, MODE, PK>), dim3(tails), dim3(P2_TB), scat_lds, ctx->stream, src, pl, lv);
                return AQG_OK;
            }
        };
        using Hashed = std::integral_constant<int, BIN_HASHED>;
        using Ranged = std::integral_constant<int, BIN_RANGED>;
        const key_t_<K>* k1 = static_cast<const key_t_<K>*>(keycol);
        const key_t_<K>* k2 = static_cast<const key_t_<K>*>(keysA);
        if (rp.on && pp.n) AQG_TRY(level(Ranged{}, std::true_type{}, k1, planes(true), l1, tiles1, 1u));
        else if (rp.on) AQG_TRY(level(Ranged{}, std::false_type{}, k1, planes(true), l1, tiles1, 1u));
        else if (pp.n) AQG_TRY(level(Hashed{}, std::true_type{}, k1, planes(true), l1, tiles1, 1u));
        else AQG_TRY(level(Hashed{}, std::false_type{}, k1, planes(true), l1, tiles1, 1u));
        if (rp.on) AQG_TRY(level(Ranged{}, std::false_type{}, k2, planes(false), l2, grid2, B1));
        else AQG_TRY(level(Hashed{}, std::false_type{}, k2, planes(false), l2, grid2, B1));
        return aqg_check_launch(ctx, "two-level partition scatter");
    };
    if (ksz == 4) AQG_TRY(run(std::false_type{})); else AQG_TRY(run(std::true_type{}));
    if (rp.on) return p1_launch_agg_direct(ctx, as, vc, keysB, rowsB, valsB, fstart, 1u, n, need_count, out, out_cap, pp.n ? &pp : nullptr, rp);
    return p1_launch_agg(ctx, ksz, as, vc, keysB, rowsB, valsB, fstart, 1u, P, n, need_count, out, out_cap, pr, pp.n ? &pp : nullptr, layout);
}

// ---- wide tuples: host ----------------------------------------------------------------------------------------------------------------
struct WidePlan { uint32_t R, P, B1; int L, nkd, low[3], nt; size_t lds; bool ok; };   // low[l]: bits of level l + 1 (the levels below the first)
static size_t pw_row_lds(int nkd, int nacc) { return 4 * (size_t)nkd + 4 + 4 + 8 * (size_t)nacc + 8 + 2; }   // key dwords | first | count | accumulators | two 4-byte slots | representative
// `hint` = the expected number of groups: with m = n / hint rows per tuple the rows of a partition are not independent -- the tuples are --
// and the spread of a partition's ROW count grows to sqrt(mean * m) (every tuple brings its m rows along); sizing by sqrt(mean) alone sent
// every table of multi-row tuples through two overflowing attempts to the HBM table (3.2e6 rows, 1.26e6 tuples: partitions at mean + 7 sigma)
static WidePlan pw_plan(const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t hint, int packed_nkd = 0 /* key dword planes when the tuple travels packed */) {
    WidePlan best;
    memset(&best, 0, sizeof best);
    int nkd = 0;
    for (int k = 0; k < ks.nkeys; ++k) nkd += aqg_dtype_size(ks.dt[k]) <= 4 ? 1 : 2;
    if (packed_nkd > 0 && packed_nkd < nkd) nkd = packed_nkd;
    if (nkd > 2 * MAXKEYS) return best;
    // workgroups per CU: three of 512 threads, two of 1024, one of 1024 -- the first that needs no more levels than the last
    // (four workgroups of 512 -- 39 KB each, 128 x 128 x 128 partitions of ~720 rows -- measured 39 ms against 24-26 for three)
    const struct { size_t budget; int nt; } shapes[3] = {{52 * 1024, 512}, {78 * 1024, 1024}, {AGG_LDS, 1024}};
    for (int si = 2; si >= 0; --si) {
        WidePlan w;
        memset(&w, 0, sizeof w);
        w.nkd = nkd; w.nt = shapes[si].nt;
        uint32_t R = (uint32_t)((shapes[si].budget - 64) / pw_row_lds(nkd, as.nacc));
        if (R > 3u * (uint32_t)w.nt) R = 3u * (uint32_t)w.nt;
        R &= ~7u;
        double mu = (double)R;
        static const double sigmas = getenv("AQG_PW_SIGMA") ? atof(getenv("AQG_PW_SIGMA")) : 6.0;     // (tests: a small value makes partitions overflow by chance)
        const double mult = hint && hint < n ? (double)n / (double)hint : 1.0;
        for (int it = 0; it < 8; ++it) mu = (double)R - sigmas * sqrt((mu > 1 ? mu : 1) * mult);   // (a million partitions: five sigma leave a quarter of the calls with one partition over)
        if (mu < 64) continue;
        const uint64_t P = (uint64_t)((double)n / mu) + 1;
        w.R = R;
        w.lds = (size_t)R * pw_row_lds(nkd, as.nacc) + 64;
        int lowsum = 0;
        if (P <= 128) w.L = 1;
        else if (P <= 128 * 128) { w.L = 2; w.low[0] = P <= 128 * 64 ? 6 : 7; lowsum = w.low[0]; }
        else if (P <= 128 * 128 * 128) { w.L = 3; w.low[0] = P <= 128 * 64 * 128 ? 6 : 7; w.low[1] = P <= 128 * 64 * 64 ? 6 : 7; lowsum = w.low[0] + w.low[1]; }
        else continue;
        w.B1 = (uint32_t)((P + ((uint64_t)1 << lowsum) - 1) >> lowsum);
        w.P = w.B1 << lowsum;
        w.ok = true;
        if (!best.ok || w.L <= best.L) best = w;
    }
    return best;
}
bool aqg_partitionw_applies(const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t hint) { return ks.wide && pw_plan(ks, as, n, hint).ok; }
uint32_t aqg_partitionw_rows(const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t hint) { return pw_plan(ks, as, n, hint).R; }

size_t aqg_partitionw_ws_bytes(const aqg_ctx* ctx, const KeySpec& ks, uint32_t n, const AccSpec& as, uint32_t hint) {
    const WidePlan w = pw_plan(ks, as, n, hint);
    ValCols vc;
    p1_val_cols(as, &vc);
    size_t per_row = 4 + 2 * (4 + 4 + 4 * (size_t)w.nkd);                    // the hash column; two sets of {hash, row, key dwords}
    for (int k = 0; k < ks.nkeys; ++k) if (aqg_dtype_size(ks.dt[k]) < 4) per_row += 4;       // widened key columns
    for (int u = 0; u < vc.n; ++u) per_row += 2 * part_val_bytes(vc.dt[u]) + (aqg_dtype_size(vc.dt[u]) < 4 ? 4 : 0);
    return ((size_t)n + 64) * per_row + 256 * (16 + 8 * MAXACC + 8 * MAXKEYS) + ((size_t)w.P + 4096) * 24 + ((size_t)n / P2_PT + (size_t)w.P + 64) * 8 + ((size_t)w.P + 8192) + 65536;   // (… + the put-off marks of pw_agg)
}

// the packing of wide tuples (PackW): fields by first fit, widest first; worth it when a third of the dword planes goes
static bool plan_packw(aqg_ctx* ctx, const KeySpec& ks, uint32_t n, PackW* pk, int* err) {
    memset(pk, 0, sizeof *pk);
    *err = AQG_OK;
    static const bool off = getenv("AQG_DISABLE_WIDE_PACK") != nullptr;      // A/B measurements only
    if (off || n < (1u << 22) || ks.nkeys < 3) return false;
    for (int k = 0; k < ks.nkeys; ++k) if (!(ks.dt[k] == AQG_INT32 || ks.dt[k] == AQG_UINT32) || ((uintptr_t)ks.col[k] & 15)) return false;
    long long mins[MAXKEYS], maxs[MAXKEYS];
    bool ok = false;
    *err = aqg_key_ranges(ctx, ks, 1u << 20, mins, maxs, &ok, n);           // (a sample spread over the column: every row is verified while it is packed)
    if (*err != AQG_OK || !ok) return false;
    int bits[MAXKEYS], order[MAXKEYS];
    for (int k = 0; k < ks.nkeys; ++k) {
        // the sample rarely holds a column's extremes (ids 1 .. 1e7: the first 2^20 rows start near 10): a little room on both sides, and a
        // non-negative column that starts near zero is measured from zero -- a value BELOW the offset would wrap into a miss
        const long long span = maxs[k] - mins[k], room = span / 64 + 8;
        mins[k] = mins[k] >= 0 && mins[k] <= span + room ? 0 : mins[k] - room;
        maxs[k] += room;
        const unsigned long long range = (unsigned long long)(maxs[k] - mins[k]);
        int b = 1;
        while (b < 32 && (1ull << b) <= range) ++b;
        bits[k] = b; order[k] = k;
    }
    for (int i = 1; i < ks.nkeys; ++i) for (int j = i; j > 0 && bits[order[j]] > bits[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    auto fit = [&](int extra, int* word, int* shift) -> int {                // dword planes needed with `extra` bits of slack per field
        int used[MAXKEYS] = {0}, nw = 0;
        for (int i = 0; i < ks.nkeys; ++i) {
            const int k = order[i], b = bits[k] + extra > 32 ? 32 : bits[k] + extra;
            int o = 0;
            while (o < nw && used[o] + b > 32) ++o;
            if (o == nw) ++nw;
            word[k] = o; shift[k] = used[o]; used[o] += b;
        }
        return nw;
    };
    int word[MAXKEYS], shift[MAXKEYS];
    const int tight = fit(0, word, shift);
    if (tight > 4 || tight * 3 > ks.nkeys * 2) return false;
    int extra = 0;
    if (fit(1, word, shift) == tight) extra = 1; else fit(0, word, shift);     // a bit of slack per field when it costs no plane
    pk->nout = tight;
    for (int k = 0; k < ks.nkeys; ++k) {
        const int b = bits[k] + extra > 32 ? 32 : bits[k] + extra;
        pk->min[k] = (uint32_t)mins[k]; pk->mask[k] = b >= 32 ? 0xFFFFFFFFu : (1u << b) - 1; pk->word[k] = word[k]; pk->shift[k] = shift[k];
    }
    return true;
}

int aqg_partitionw_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, int need_count, GTable out, uint32_t out_cap, uint32_t seed, uint32_t hint, int* pack, uint32_t* rows_out, bool may_defer) {
    WidePlan w = pw_plan(ks, as, n, hint);
    if (!w.ok) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "wide-tuple partitioned group-by: the input does not fit 128 x 128 x 128 partitions");
    ValCols vc;
    p1_val_cols(as, &vc);
    const unsigned g4 = aqg_grid(ctx, n, 256, 4, 16);
    // the partition key: a 32-bit hash of the tuple
    uint32_t* h32;
    AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &h32));
    PackW pk;
    bool packed = false, level1_counted = false;
    uint32_t *seg = nullptr, *tp = nullptr, *cnt = nullptr, *cur = nullptr, *bsum = nullptr;
    size_t maxseg = 0;
    {
        Keys32 k32;
        memset(&k32, 0, sizeof k32);
        bool all32 = true;
        for (int k = 0; k < ks.nkeys; ++k) {
            all32 = all32 && aqg_dtype_size(ks.dt[k]) == 4 && ((uintptr_t)ks.col[k] & 15) == 0;
            k32.col[k] = static_cast<const uint32_t*>(ks.col[k]);
        }
        k32.n = ks.nkeys;
        static const bool generic_hash = getenv("AQG_PW_GENERIC_HASH") != nullptr;
        memset(&pk, 0, sizeof pk);
        if (all32 && !generic_hash && pack && *pack) {
            int err = AQG_OK;
            packed = plan_packw(ctx, ks, n, &pk, &err);
            AQG_TRY(err);
        }
        if (packed) {                                     // fewer key dwords per row: more rows per partition, fewer partitions (the workspace was sized for the unpacked plan: more of each)
            const WidePlan wp = pw_plan(ks, as, n, hint, pk.nout);
            if (wp.ok && wp.P <= w.P) w = wp;
        }
        // level bookkeeping (segments of level l = the bins of level l - 1); the 32-bit hash passes count the first level's bins themselves
        maxseg = (size_t)w.P + 2;
        AQG_TRY(aqg_ws_get(ctx, maxseg, &seg));
        AQG_TRY(aqg_ws_get(ctx, maxseg, &tp));
        AQG_TRY(aqg_ws_get(ctx, maxseg, &cnt));
        AQG_TRY(aqg_ws_get(ctx, maxseg, &cur));
        AQG_TRY(aqg_ws_get(ctx, maxseg / 2048 + 64, &bsum));
        uint32_t shift1 = 0;
        for (int j = 1; j < w.L; ++j) shift1 += (uint32_t)w.low[j - 1];
        const Level1Count lc{cnt, w.P, shift1};
        level1_counted = all32 && !generic_hash && w.B1 <= 128;
        if (level1_counted) AQG_HIP(ctx, hipMemsetAsync(cnt, 0, ((size_t)w.B1 + 1) * 4, ctx->stream));
        if (packed) {
            for (int o = 0; o < pk.nout; ++o) AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &pk.out[o]));
            pk.flag = out.flags + 6;
            hipLaunchKernelGGL(pw_hash32_kernel<true>, dim3(aqg_grid(ctx, n / 4 + 1, 256, 2, 16)), dim3(256), 0, ctx->stream, k32, n, seed, h32, pk, lc);
        }
        else if (all32 && !generic_hash) hipLaunchKernelGGL(pw_hash32_kernel<false>, dim3(aqg_grid(ctx, n / 4 + 1, 256, 2, 16)), dim3(256), 0, ctx->stream, k32, n, seed, h32, pk, lc);
        else hipLaunchKernelGGL(pw_hash_kernel, dim3(g4), dim3(256), 0, ctx->stream, ks, n, seed, h32);
    }
    if (pack) *pack = packed ? 1 : 0;
    if (rows_out) *rows_out = w.R;
    const int nkd = packed ? pk.nout : w.nkd;             // key dword planes that travel
    // source planes: the key columns as dwords (1- / 2-byte ones widened, 8-byte ones as two planes), then the distinct value columns
    struct Src { const void* p; int stride, off, bytes; };
    std::vector<Src> ksrc, vsrc;
    if (packed) for (int o = 0; o < pk.nout; ++o) ksrc.push_back({pk.out[o], 1, 0, 4});
    for (int k = 0; k < ks.nkeys && !packed; ++k) {
        const int esz = (int)aqg_dtype_size(ks.dt[k]);
        const void* col = ks.col[k];
        if (esz < 4) {
            void* wide;
            AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &wide));
            hipLaunchKernelGGL(p1_widen_kernel, dim3(g4), dim3(256), 0, ctx->stream, col, esz, n, static_cast<uint32_t*>(wide));
            col = wide;
        }
        if (esz <= 4) ksrc.push_back({col, 1, 0, 4});
        else { ksrc.push_back({col, 2, 0, 4}); ksrc.push_back({col, 2, 1, 4}); }
    }
    for (int u = 0; u < vc.n; ++u) {
        const int esz = (int)aqg_dtype_size(vc.dt[u]);
        const void* col = vc.col[u];
        if (esz < 4) {
            void* wide;
            AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &wide));
            hipLaunchKernelGGL(p1_widen_kernel, dim3(g4), dim3(256), 0, ctx->stream, col, esz, n, static_cast<uint32_t*>(wide));
            col = wide;
        }
        vsrc.push_back({col, 1, 0, (int)part_val_bytes(vc.dt[u])});
    }
    // two buffer sets: hash | row | key dwords | values
    struct Set { uint32_t* hash; uint32_t* rows; uint32_t* kd[2 * MAXKEYS]; void* val[MAXACC]; } set[2];
    for (int i = 0; i < 2; ++i) {
        AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &set[i].hash));
        AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &set[i].rows));
        for (int k = 0; k < nkd; ++k) AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &set[i].kd[k]));
        for (int u = 0; u < vc.n; ++u) AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * vsrc[u].bytes, &set[i].val[u]));
    }
    auto planes = [&](int level, const Set* from, const Set& to) {
        Planes pl;
        memset(&pl, 0, sizeof pl);
        auto add = [&](int kind, const void* s_, int sstride, int soff, void* d, int dstride, int doff) {
            Plane& Q = pl.p[pl.n++];
            Q.kind = kind; Q.src = static_cast<const uint32_t*>(s_); Q.src_stride_dw = sstride; Q.src_off_dw = soff;
            Q.dst = static_cast<uint32_t*>(d); Q.dst_stride_dw = dstride; Q.dst_off_dw = doff;
        };
        add(PL_LOAD, level == 1 ? h32 : from->hash, 1, 0, level == w.L ? nullptr : to.hash, 1, 0);     // (nobody reads the hash behind the last level)
        if (level == 1) add(PL_ROWIDX, nullptr, 0, 0, to.rows, 1, 0); else add(PL_LOAD, from->rows, 1, 0, to.rows, 1, 0);
        for (int k = 0; k < nkd; ++k) {
            if (level == 1) add(PL_LOAD, ksrc[k].p, ksrc[k].stride, ksrc[k].off, to.kd[k], 1, 0);
            else add(PL_LOAD, from->kd[k], 1, 0, to.kd[k], 1, 0);
        }
        for (int u = 0; u < vc.n; ++u) {
            const void* src = level == 1 ? vsrc[u].p : from->val[u];
            if (vsrc[u].bytes == 4) add(PL_LOAD, src, 1, 0, to.val[u], 1, 0);
            else { add(PL_LOAD, src, 2, 0, to.val[u], 2, 0); add(PL_LOAD, src, 2, 1, to.val[u], 2, 1); }
        }
        return pl;
    };
    if (2 + nkd + 2 * vc.n > MAXPL) return aqg_fail(ctx, AQG_ERR_ARG, "wide-tuple partitioned group-by: too many planes");
    const uint32_t h0[2] = {0u, n};
    void* st = nullptr;
    AQG_TRY(aqg_host_stage(ctx, 16, &st));
    memcpy(st, h0, 8);
    AQG_HIP(ctx, hipMemcpyAsync(seg, st, 8, hipMemcpyHostToDevice, ctx->stream));
    const size_t scat_lds = (size_t)P2_PT * 4;
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, true>), scat_lds));
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, false>), scat_lds));
    uint32_t nseg = 1;
    const Set* from = nullptr;
    int to = 0;
    for (int l = 1; l <= w.L; ++l) {
        uint32_t shift = 0;
        for (int j = l; j < w.L; ++j) shift += (uint32_t)w.low[j - 1];          // bits of the levels below this one
        const uint32_t nb = l == 1 ? w.B1 : 1u << w.low[l - 2], mask = l == 1 ? 0xFFFFFFFFu : nb - 1;
        const uint32_t* keys = l == 1 ? h32 : from->hash;
        const unsigned tiles = (unsigned)((uint64_t)n / P2_PT) + nseg + 1;
        hipLaunchKernelGGL(pn_tiles_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg, nseg, (uint32_t)P2_PT, tp);
        P2Level lv{seg, tp, cur, nseg, w.P, shift, mask, nb, nb};
        if (!(l == 1 && level1_counted)) {
            AQG_HIP(ctx, hipMemsetAsync(cnt, 0, ((size_t)nseg * nb + 1) * 4, ctx->stream));
            hipLaunchKernelGGL((pn_level_hist_kernel<P2_TB, P2_TR, true>), dim3(tiles), dim3(P2_TB), 0, ctx->stream, keys, lv, cnt);
        }
        AQG_TRY(aqg_exclusive_scan_u32(ctx, cnt, (uint64_t)nseg * nb + 1, bsum));        // cnt[i] = start of (segment, bin) i; the last word = n
        AQG_HIP(ctx, hipMemcpyAsync(cur, cnt, (size_t)nseg * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
        const Planes pl = planes(l, from, set[to]);
        hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, true>), dim3(tiles), dim3(P2_TB), scat_lds, ctx->stream, keys, pl, lv);
        hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, false>), dim3(nseg), dim3(P2_TB), scat_lds, ctx->stream, keys, pl, lv);
        AQG_TRY(aqg_check_launch(ctx, "wide-tuple partition level"));
        // the bins of this level are the segments of the next (and, after the last level, the partitions)
        AQG_HIP(ctx, hipMemcpyAsync(seg, cnt, ((size_t)nseg * nb + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
        nseg *= nb;
        from = &set[to];
        to ^= 1;
    }
    // ---- aggregate every partition inside LDS -----------------------------------------------------------------------------------------
    WideIn in;
    memset(&in, 0, sizeof in);
    in.nkd = nkd;
    for (int k = 0; k < nkd; ++k) in.kplane[k] = from->kd[k];
    in.rows = from->rows;
    AggOps ops;
    memset(&ops, 0, sizeof ops);
    for (int a = 0; a < as.nacc; ++a) {
        if (vc.of_acc[a] >= 0) { in.vcol[a] = from->val[vc.of_acc[a]]; in.vesz[a] = vsrc[vc.of_acc[a]].bytes; }
        else { in.vcol[a] = nullptr; in.vesz[a] = 4; }
        const int dt = as.dt[a], kind = as.kind[a];
        int opc = OPC_GENERIC;
        if (!as.square[a] && !as.part[a]) {
            if (dt == AQG_INT32 && kind == ACC_ADD_I) opc = OPC_ADDI_I32;
            else if (dt == AQG_UINT32 && kind == ACC_ADD_I) opc = OPC_ADDI_U32;
            else if (dt == AQG_FLOAT && kind == ACC_ADD_F) opc = OPC_ADDF_F32;
            else if (dt == AQG_DOUBLE && kind == ACC_ADD_F) opc = OPC_ADDF_F64;
        }
        ops.opc[a] = opc;
    }
    const size_t lds = w.lds;
    unsigned per_cu = (unsigned)((160 * 1024) / (lds + 512));
    if (per_cu > 2048u / (unsigned)w.nt) per_cu = 2048u / (unsigned)w.nt;
    if (per_cu < 1) per_cu = 1;
    const unsigned grid = nseg < per_cu * (unsigned)ctx->num_cu ? nseg : per_cu * (unsigned)ctx->num_cu;
    static const bool defer_off = getenv("AQG_DISABLE_PW_DEFER") != nullptr;                // A/B measurements only
    if (defer_off) may_defer = false;
    uint32_t* dwords = nullptr;                                                            // [0] rows of the partitions put off | marks, a byte per partition
    if (may_defer) {
        AQG_TRY(aqg_ws_get(ctx, (size_t)nseg / 4 + 8, &dwords));
        AQG_HIP(ctx, hipMemsetAsync(dwords, 0, ((size_t)nseg / 4 + 8) * 4, ctx->stream));
    }
    uint8_t* dmark = may_defer ? reinterpret_cast<uint8_t*>(dwords + 4) : nullptr;
    const int lazy_vals = may_defer && (uint64_t)hint * 10 >= (uint64_t)n * 9 ? 1 : 0;     // nearly as many groups expected as rows: row ids and values are read only where a partition has a duplicate
    auto launch = [&](auto kern) -> int {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(w.nt), lds, ctx->stream, in, as, ops, (const uint32_t*)seg, nseg, n, w.R, need_count, out, out_cap, dmark, dwords, may_defer ? 1 : 0, lazy_vals);
        aqg_kernel_timer_end(ctx);
        AQG_TRY(aqg_check_launch(ctx, "pw_agg_kernel"));
        if (!may_defer) return AQG_OK;
        // every row its own group?  (one host round trip on a call of tens of milliseconds)
        uint32_t fl[2] = {0, 0}, put_off = 0;
        AQG_HIP(ctx, hipMemcpyAsync(fl, out.flags, 8, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipMemcpyAsync(&put_off, dwords, 4, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (fl[0] || !put_off) return AQG_OK;                                               // (an overflow is the caller's to judge)
        if ((uint64_t)fl[1] + put_off == n) {                                               // yes: the caller emits from the rows, the record table is not read
            AQG_HIP(ctx, hipMemcpyAsync(out.flags + 1, &n, 4, hipMemcpyHostToDevice, ctx->stream));
            AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));                                // (`n` lives on this stack frame)
            return AQG_OK;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(w.nt), lds, ctx->stream, in, as, ops, (const uint32_t*)seg, nseg, n, w.R, need_count, out, out_cap, dmark, dwords, 2, 0);
        return aqg_check_launch(ctx, "pw_agg_kernel (partitions put off)");
    };
    auto pick = [&](auto nacc) -> int {
        constexpr int N = decltype(nacc)::value;
        return w.nt == 512 ? launch(&pw_agg_kernel<N, 512>) : launch(&pw_agg_kernel<N, 1024>);
    };
    switch (as.nacc) {
    case 0: return pick(std::integral_constant<int, 0>{});
    case 1: return pick(std::integral_constant<int, 1>{});
    case 2: return pick(std::integral_constant<int, 2>{});
    case 3: return pick(std::integral_constant<int, 3>{});
    case 4: return pick(std::integral_constant<int, 4>{});
    default: return aqg_fail(ctx, AQG_ERR_OVERFLOW, "wide-tuple partitioned group-by: at most 4 accumulators");
    }
}

// ---- ordering a huge group table: host ------------------------------------------------------------------------------------------------
// LDS of sorted_emit_kernel per record: key (packed keys only) | first | count | accumulators, plus two bits per row of the interval
static size_t sorted_rec_bytes(int nacc, bool wide) { return (wide ? 0 : 8) + 8 + 8 * (size_t)nacc; }
bool aqg_sorted_tail_plan(uint32_t n_rows, int nacc, bool wide, SortedPlan* out) {
    if (n_rows <= 8192) return false;
    SortedPlan best;
    memset(&best, 0, sizeof best);
    for (size_t budget : {(size_t)64 * 1024, (size_t)150 * 1024}) {
        uint32_t iv = (uint32_t)((budget - 1024) / (sorted_rec_bytes(nacc, wide) + 1)) & ~31u;   // rows (= records at most) one partition may span
        if (iv > 16384) iv = 16384;                                   // one thread per bitmap word, 512 threads
        if (iv < 64) continue;
        uint32_t bits = 1;
        while (bits <= 18 && ((uint64_t)n_rows + ((uint64_t)1 << bits) - 1) / ((uint64_t)1 << bits) + 4 > iv) ++bits;
        if (bits > 18) continue;
        const uint32_t levels = (bits + 5) / 6;
        while (bits < 6 * levels && ((uint64_t)1 << (bits + 1)) <= n_rows / 64) ++bits;   // the levels are paid for: use their bins
        if (best.levels && best.levels <= levels) continue;
        best.levels = levels; best.bits = bits; best.cap = iv;
        best.M = (uint32_t)((((uint64_t)1 << bits) << 32) / n_rows);
        best.lds = (size_t)iv * sorted_rec_bytes(nacc, wide) + 2 * ((size_t)iv / 32 + 8) * 4 + 64;
    }
    if (!best.levels) return false;
    if (out) *out = best;
    return true;
}
size_t aqg_sorted_tail_ws_bytes(uint32_t gcap, uint32_t n_rows, int nacc, bool wide) {
    SortedPlan sp;
    if (!aqg_sorted_tail_plan(n_rows, nacc, wide, &sp)) return 0;
    const size_t per = 2 * (4 + 4 + (wide ? 0 : 8) + 8 * (size_t)nacc);                    // two plane sets
    return ((size_t)gcap + 64) * per + 256 * (16 + 4 * MAXACC) + (size_t)5 * (((size_t)1 << sp.bits) + 64) * 4 + 65536;
}
// records 0 .. G-1 of `gt` (column layout: keys | first rows | counts | accumulators) -> the same planes partitioned by first row
int aqg_sorted_tail(aqg_ctx* ctx, const GTable& gt, uint32_t G, uint32_t n_rows, int nacc, bool wide, SortedParts* out) {
    SortedPlan sp;
    if (!aqg_sorted_tail_plan(n_rows, nacc, wide, &sp)) return aqg_fail(ctx, AQG_ERR_ARG, "ordered group table: no plan for this shape");
    if (gt.fst != 4 || gt.cst != 4 || gt.kst != 8 || gt.ast != 8) return aqg_fail(ctx, AQG_ERR_ARG, "ordered group table: column layout expected");
    const uint32_t PP = 1u << sp.bits, M = sp.M;
    struct Set { uint32_t* first; uint32_t* count; uint64_t* key; uint64_t* acc[MAXACC]; } set[3];
    memset(set, 0, sizeof set);
    set[2].first = reinterpret_cast<uint32_t*>(gt.fb); set[2].count = reinterpret_cast<uint32_t*>(gt.cb); set[2].key = reinterpret_cast<uint64_t*>(gt.kb);
    for (int a = 0; a < nacc; ++a) set[2].acc[a] = reinterpret_cast<uint64_t*>(gt.ab + (size_t)a * gt.astep);
    for (int i = 0; i < 2; ++i) {
        AQG_TRY(aqg_ws_get(ctx, (size_t)G + 64, &set[i].first));
        AQG_TRY(aqg_ws_get(ctx, (size_t)G + 64, &set[i].count));
        if (!wide) AQG_TRY(aqg_ws_get(ctx, (size_t)G + 64, &set[i].key));
        for (int a = 0; a < nacc; ++a) AQG_TRY(aqg_ws_get(ctx, (size_t)G + 64, &set[i].acc[a]));
    }
    auto planes = [&](const Set& from, const Set& to) {
        Planes pl;
        memset(&pl, 0, sizeof pl);
        auto add = [&](const void* s_, int sstride, int soff, void* d, int dstride, int doff) {
            Plane& Q = pl.p[pl.n++];
            Q.kind = PL_LOAD; Q.src = static_cast<const uint32_t*>(s_); Q.src_stride_dw = sstride; Q.src_off_dw = soff;
            Q.dst = static_cast<uint32_t*>(d); Q.dst_stride_dw = dstride; Q.dst_off_dw = doff;
        };
        add(from.first, 1, 0, to.first, 1, 0);
        add(from.count, 1, 0, to.count, 1, 0);
        if (!wide) { add(from.key, 2, 0, to.key, 2, 0); add(from.key, 2, 1, to.key, 2, 1); }
        for (int a = 0; a < nacc; ++a) { add(from.acc[a], 2, 0, to.acc[a], 2, 0); add(from.acc[a], 2, 1, to.acc[a], 2, 1); }
        return pl;
    };
    uint32_t *seg, *tp, *cnt, *cur, *bsum;
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &seg));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &tp));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &cnt));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &cur));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP / 1024 + 64 + 8, &bsum));
    const uint32_t h0[2] = {0u, G};
    void* st = nullptr;
    AQG_TRY(aqg_host_stage(ctx, 16, &st));
    memcpy(st, h0, 8);
    AQG_HIP(ctx, hipMemcpyAsync(seg, st, 8, hipMemcpyHostToDevice, ctx->stream));
    const size_t scat_lds = (size_t)P2_PT * 4;
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, true, false>), scat_lds));
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, false, false>), scat_lds));
    uint32_t nseg = 1, bits_left = sp.bits;
    const Set* from = &set[2];
    for (uint32_t l = 0; l < sp.levels; ++l) {
        const uint32_t lb = (bits_left + (sp.levels - l) - 1) / (sp.levels - l);       // bits of this level (most significant first)
        bits_left -= lb;
        const uint32_t nb = 1u << lb, shift = bits_left, mask = nb - 1;
        const uint32_t* keys = from->first;
        const unsigned tiles = (unsigned)((uint64_t)G / P2_PT) + nseg + 1;
        hipLaunchKernelGGL(pn_tiles_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg, nseg, (uint32_t)P2_PT, tp);
        AQG_HIP(ctx, hipMemsetAsync(cnt, 0, ((size_t)nseg * nb + 1) * 4, ctx->stream));
        P2Level lv{seg, tp, cur, nseg, M, shift, mask, nb, nb};
        hipLaunchKernelGGL((pn_level_hist_kernel<P2_TB, P2_TR, false>), dim3(tiles), dim3(P2_TB), 0, ctx->stream, keys, lv, cnt);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, cnt, (uint64_t)nseg * nb + 1, bsum));
        AQG_HIP(ctx, hipMemcpyAsync(cur, cnt, (size_t)nseg * nb * 4, hipMemcpyDeviceToDevice, ctx->stream));
        const Set& to = set[l & 1];
        const Planes pl = planes(*from, to);
        hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, true, false>), dim3(tiles), dim3(P2_TB), scat_lds, ctx->stream, keys, pl, lv);
        hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, false, false>), dim3(nseg), dim3(P2_TB), scat_lds, ctx->stream, keys, pl, lv);
        AQG_TRY(aqg_check_launch(ctx, "ordered group table: level"));
        AQG_HIP(ctx, hipMemcpyAsync(seg, cnt, ((size_t)nseg * nb + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
        nseg *= nb;
        from = &to;
    }
    memset(out, 0, sizeof *out);
    out->first = from->first; out->count = from->count; out->key = from->key;
    for (int a = 0; a < nacc; ++a) out->acc[a] = from->acc[a];
    out->pstart = seg; out->nparts = nseg; out->M = M; out->cap = sp.cap; out->lds = sp.lds;
    return AQG_OK;
}

// ==== the BUILD through the partition plans: group id of every row ============================================================================
// aqg_groupby_build needs, beyond the group table, the dense id of every row (AQHashTable's reversemap, server/hasher.h:167-179).  Up to
// here a build above the LDS tables inserted every row into an HBM table and looked every row up again (1e9 rows, 1e7 groups: 37 + 47 ms
// of scattered HBM accesses).  Now the group table comes from the partition plan (no accumulators, counts only), and the rows -- still
// lying partitioned in the workspace, {key, row id} -- are walked ONCE more per partition: the partition's records (p1_agg notes which
// range of the record table it wrote) go into an LDS table {key -> dense id of the record}, every row probes it and writes
// reversemap[row id].  8 B/row read + a scattered 4-byte write per row.
namespace {
template <bool K64>
__global__ void __launch_bounds__(1024) p_assign_kernel(PartRows pr, GTable gt, const uint32_t* __restrict__ slot_gid, uint32_t* __restrict__ gid_part /* [ntotal]: the id of the row at every partitioned position */) {
    using K = key_t_<K64>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    K* ktab = reinterpret_cast<K*>(smem_raw);                        // [cap]
    uint32_t* gtab = reinterpret_cast<uint32_t*>(ktab + pr.cap);     // [cap]
    __shared__ uint32_t special_gid;
    const K EMPTYK = empty_key<K64>();
    const uint32_t cap = pr.cap, NB = pr.nparts;
    for (uint32_t part = blockIdx.x; part < NB; part += gridDim.x) {
        const uint32_t b = pr.pstart[(size_t)part * pr.pstride];
        const uint32_t e = part + 1 < NB ? pr.pstart[(size_t)(part + 1) * pr.pstride] : pr.ntotal;
        if (b == e) continue;
        const uint32_t base = pr.part_base[2 * (size_t)part], used = pr.part_base[2 * (size_t)part + 1];
        for (uint32_t s = threadIdx.x; s < cap; s += 1024) ktab[s] = EMPTYK;
        if (threadIdx.x == 0) special_gid = 0;
        __syncthreads();
        for (uint32_t r = threadIdx.x; r < used; r += 1024) {         // this partition's records -> {key -> dense id}
            const uint32_t rec = base + r;
            const K k = (K)*gt.key_p(rec);
            const uint32_t gid = slot_gid[rec];
            if (k == EMPTYK) { special_gid = gid; continue; }
            uint32_t slot = __umulhi(key_hash<K64>(k) * NB, cap);
            for (uint32_t step = 0; step < cap; ++step) {
                K c;
                if constexpr (K64) c = atomicCAS(reinterpret_cast<unsigned long long*>(&ktab[slot]), (unsigned long long)EMPTYK, (unsigned long long)k);
                else c = atomicCAS(&ktab[slot], EMPTYK, k);
                if (c == EMPTYK || c == k) { gtab[slot] = gid; break; }
                slot = slot + 1 == cap ? 0 : slot + 1;
            }
        }
        __syncthreads();
        {   // a lane takes AR consecutive rows of a step by 16-byte loads and stores (4-byte aligned: a partition starts anywhere), the next step's
            // keys in flight while this step's are looked up (four rows per lane by dword loads, one step at a time: 2.3 ms per 1e9 rows)
            constexpr int AR = K64 ? 4 : 8;
            constexpr uint32_t STEP = 1024 * AR;
            struct Batch { K key[AR]; };
            auto load_full = [&](uint32_t i0, Batch& t) { __builtin_memcpy(t.key, static_cast<const K*>(pr.keys) + i0 + threadIdx.x * AR, sizeof t.key); };
            const uint32_t nfull = (e - b) / STEP, nsteps = nfull + ((e - b) % STEP ? 1u : 0u);
            const uint32_t safe_last = nfull ? b + (nfull - 1) * STEP : (b + STEP <= pr.ntotal ? b : pr.ntotal - STEP);    // (the partitioned build runs from 2^20 rows)
            Batch cur;
            load_full(nfull ? b : safe_last, cur);
            uint32_t i0 = b;
            for (uint32_t st = 0; st < nsteps; ++st, i0 += STEP) {
                const bool edge = st >= nfull;
                const uint32_t o = i0 + threadIdx.x * AR;
                if (edge) {
#pragma unroll
                    for (int q = 0; q < AR; ++q) cur.key[q] = static_cast<const K*>(pr.keys)[o + q < e ? o + q : e - 1];
                }
                Batch nxt;
                load_full(st + 1 < nfull ? i0 + STEP : safe_last, nxt);
                __builtin_amdgcn_sched_barrier(0);
                uint32_t slot[AR], gid[AR]; K w[AR];
#pragma unroll
                for (int q = 0; q < AR; ++q) { slot[q] = __umulhi(key_hash<K64>(cur.key[q]) * NB, cap); w[q] = ktab[slot[q]]; }
#pragma unroll
                for (int q = 0; q < AR; ++q) {
                    if (cur.key[q] == EMPTYK) gid[q] = special_gid;
                    else {
                        uint32_t sl = slot[q];
                        K c = w[q];
                        for (uint32_t step = 0; c != cur.key[q] && step < cap; ++step) { sl = sl + 1 == cap ? 0 : sl + 1; c = ktab[sl]; }
                        gid[q] = gtab[sl];
                    }
                }
                if (!edge) __builtin_memcpy(gid_part + o, gid, sizeof gid);
                else {
#pragma unroll
                    for (int q = 0; q < AR; ++q) if (o + q < e) gid_part[o + q] = gid[q];
                }
                cur = nxt;
            }
        }
        __syncthreads();
    }
}
// out[idx[i]] = val[i] for a PERMUTATION idx of 0 .. n-1 (every row id once): a scattered 4-byte store per row runs at the rate of the
// memory side (~3e10/s: 43 ms per 1e9 rows as the build's last step), so the pairs {idx, val} are first partitioned on idx -- order-
// preserving bins, the tile scatter again, no histogram: a partition's size IS its index interval -- until an interval spans 64 K
// rows; the stores of a workgroup then land inside a 256 KB window that its L2 turns into whole lines.
// the last step: partition p holds exactly the pairs whose index lies in [pstart[p], pstart[p + 1]) -- as many pairs as indices.  One
// workgroup per partition places the values in LDS by index (a window of 32 K indices at a time: a partition of 64 K rows takes two sweeps
// over its pairs) and streams the window out: every store instruction writes whole lines.  (Plain stores through the index, every
// workgroup inside its own 256 KB window: 14 ms per 1e9 rows -- 2048 such windows do not fit the L2s.)
constexpr uint32_t ROUTE_W = 32768;
__global__ void __launch_bounds__(1024) route_final_kernel(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ val, const uint32_t* __restrict__ pstart, uint32_t nparts,
                                                           uint32_t n, uint32_t* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* win = reinterpret_cast<uint32_t*>(smem_raw);
    for (uint32_t p = blockIdx.x; p < nparts; p += gridDim.x) {
        const uint32_t b = pstart ? pstart[p] : 0u, e = pstart ? pstart[p + 1] : n;
        for (uint32_t w0 = b; w0 < e; w0 += ROUTE_W) {
            const uint32_t w1 = e - w0 < ROUTE_W ? e : w0 + ROUTE_W;
            // (consecutive pairs per lane by 16-byte loads with the next step in flight -- what took gid_agg from 2 to 5.6 TB/s -- changed nothing here:
            // 3.32 against 3.36 ms; a partition of up to 65536 rows is read once per 32768-row window and the window's random LDS stores are what it waits for)
            for (uint32_t i0 = b + threadIdx.x; i0 < e; i0 += 4 * 1024) {
                uint32_t r[4], v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const uint32_t i = i0 + q * 1024, ic = i < e ? i : e - 1; r[q] = idx[ic]; v[q] = val[ic]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) if (i0 + q * 1024 < e && r[q] >= w0 && r[q] < w1) win[r[q] - w0] = v[q];
            }
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < w1 - w0; j += 1024) out[w0 + j] = win[j];
            __syncthreads();
        }
    }
}
size_t aqg_route_ws_bytes(uint32_t n) { return ((size_t)n + 64) * 16 + ((size_t)1 << 20); }
// (declared below: gid_setup_kernel; offsets == nullptr: the identity)
__global__ void __launch_bounds__(256) gid_setup_kernel(const uint32_t* __restrict__ offsets, uint32_t G, uint32_t M, uint32_t PP, uint32_t* __restrict__ pstart, uint32_t* __restrict__ pfirst);
int aqg_route_by_row(aqg_ctx* ctx, const uint32_t* idx, const uint32_t* val, uint32_t n, uint32_t* out) {
    uint32_t bits = 0;
    while (bits < 21 && ((uint64_t)n >> bits) > 65536) ++bits;
    const uint32_t* isrc = idx;
    const uint32_t* vsrc = val;
    const uint32_t* final_pstart = nullptr;
    if (bits) {
        const uint32_t levels = (bits + 6) / 7, PP = 1u << bits;
        const uint32_t M = (uint32_t)((((uint64_t)1 << bits) << 32) / n);
        uint32_t *iA, *iB, *vA, *vB, *pstart, *pfirst, *seg, *tp, *cur;
        AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &iA)); AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &iB));
        AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &vA)); AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &vB));
        AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &pstart)); AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &pfirst));
        AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &seg)); AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &tp)); AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &cur));
        hipLaunchKernelGGL(gid_setup_kernel, dim3(aqg_grid(ctx, (uint64_t)PP + 1, 256, 1, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)nullptr, n, M, PP, pstart, pfirst);
        final_pstart = pstart;
        const size_t scat_lds = (size_t)P2_PT * 4;
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, true, false>), scat_lds));
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, false, false>), scat_lds));
        uint32_t nseg = 1, bits_left = bits;
        for (uint32_t l = 0; l < levels; ++l) {
            const uint32_t lb = (bits_left + (levels - l) - 1) / (levels - l);
            bits_left -= lb;
            const uint32_t nb = 1u << lb, shift = bits_left, mask = nb - 1;
            hipLaunchKernelGGL(pn_gather_strided_kernel, dim3(aqg_grid(ctx, (uint64_t)nseg + 1, 256, 1, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)pstart, nb << shift, nseg + 1, seg);
            hipLaunchKernelGGL(pn_gather_strided_kernel, dim3(aqg_grid(ctx, (uint64_t)nseg * nb, 256, 1, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)pstart, 1u << shift, nseg * nb, cur);
            hipLaunchKernelGGL(pn_tiles_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg, nseg, (uint32_t)P2_PT, tp);
            uint32_t* idst = (l & 1) ? iB : iA;
            uint32_t* vdst = (l & 1) ? vB : vA;
            Planes pl;
            memset(&pl, 0, sizeof pl);
            auto add = [&](const void* s_, void* d) { Plane& Q = pl.p[pl.n++]; Q.kind = PL_LOAD; Q.src = static_cast<const uint32_t*>(s_); Q.src_stride_dw = 1; Q.src_off_dw = 0; Q.dst = static_cast<uint32_t*>(d); Q.dst_stride_dw = 1; Q.dst_off_dw = 0; };
            add(isrc, idst); add(vsrc, vdst);
            P2Level lv{seg, tp, cur, nseg, M, shift, mask, nb, nb};
            const unsigned tiles = (unsigned)((uint64_t)n / P2_PT) + nseg + 1;
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, true, false>), dim3(tiles), dim3(P2_TB), scat_lds, ctx->stream, isrc, pl, lv);
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, false, false>), dim3(nseg), dim3(P2_TB), scat_lds, ctx->stream, isrc, pl, lv);
            AQG_TRY(aqg_check_launch(ctx, "route by row: level"));
            nseg *= nb;
            isrc = idst; vsrc = vdst;
        }
    }
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&route_final_kernel), (size_t)ROUTE_W * 4));
    hipLaunchKernelGGL(route_final_kernel, dim3(bits ? (1u << bits) : 1u), dim3(1024), (size_t)ROUTE_W * 4, ctx->stream, isrc, vsrc, final_pstart, bits ? (1u << bits) : 1u, n, out);
    return aqg_check_launch(ctx, "route_final_kernel");
}
} // namespace
size_t aqg_partition_assign_ws_bytes(uint32_t n) { return ((size_t)n + 64) * 4 + aqg_route_ws_bytes(n) + 65536; }
int aqg_partition_assign(aqg_ctx* ctx, const PartRows& pr, GTable gt, const uint32_t* slot_gid, uint32_t* reversemap) {
    if (!pr.valid) return aqg_fail(ctx, AQG_ERR_ARG, "partitioned build: no partitioned rows");
    uint32_t* gid_part;
    AQG_TRY(aqg_ws_get(ctx, (size_t)pr.ntotal + 64, &gid_part));
    const size_t lds = (size_t)pr.cap * (pr.ksz + 4) + 64;
    const unsigned grid = pr.nparts < 2u * (unsigned)ctx->num_cu ? pr.nparts : 2u * (unsigned)ctx->num_cu;
    if (pr.ksz == 4) {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p_assign_kernel<false>), lds));
        hipLaunchKernelGGL((p_assign_kernel<false>), dim3(grid), dim3(1024), lds, ctx->stream, pr, gt, slot_gid, gid_part);
    } else {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p_assign_kernel<true>), lds));
        hipLaunchKernelGGL((p_assign_kernel<true>), dim3(grid), dim3(1024), lds, ctx->stream, pr, gt, slot_gid, gid_part);
    }
    AQG_TRY(aqg_check_launch(ctx, "p_assign_kernel"));
    return aqg_route_by_row(ctx, pr.rows, gid_part, pr.ntotal, reversemap);      // the ids back into row order
}

// ==== grouped reductions keyed by DENSE group ids: aqg_grouped_reduce beyond the LDS tables =============================================
// `out[g] = op(x[rows of group g])` for the generated loop (engine/ast.py:722-789) groups by the build's id column.  Those ids are dense
// and the group sizes are known, which the hashed partition plans above cannot use: here the rows {id, value} are partitioned on the id
// itself -- ORDER-PRESERVING bins umulhi(id, M), so a partition owns a contiguous id range -- with the tile scatter of the two-level
// plan, in as many levels of <= 128 bins as it takes until a partition's id range fits an LDS array of accumulators.  No histogram
// pass at any level: a partition's rows are the rows of its groups, so every segment start and write cursor is an entry of the
// build's offsets (the exclusive scan of the group sizes).  The aggregation is then DIRECT-indexed -- acc[id - first id of the
// partition], no keys, no probing -- and every workgroup writes its id range of the result column front to back: no record table, no
// ranking, no emit.  16 B/row and level + 8 B/row for the aggregation (1e9 rows, 1e7 groups: two levels).
namespace {

__global__ void __launch_bounds__(256) gid_setup_kernel(const uint32_t* __restrict__ offsets, uint32_t G, uint32_t M, uint32_t PP,
                                                        uint32_t* __restrict__ pstart /* [PP + 1] */, uint32_t* __restrict__ pfirst /* [PP + 1] */) {
    for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p <= PP; p += gridDim.x * 256) {
        uint64_t g0 = p == PP ? G : (((uint64_t)p << 32) + M - 1) / M;        // smallest id whose bin is >= p
        if (g0 > G) g0 = G;
        pfirst[p] = (uint32_t)g0;
        pstart[p] = offsets ? offsets[g0] : (uint32_t)g0;
    }
}

__device__ inline aqg_i128 mul_128_p1(aqg_i128 a, aqg_i128 b) {   // low 128 bits of the product (two's complement: sign-agnostic)
    aqg_i128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
    return r;
}
struct GidAgg {
    const uint32_t* gid; const void* val; int vdt; int op;
    const uint32_t* pstart; const uint32_t* pfirst; const uint32_t* counts;
    void* out; uint32_t nparts, cap, ntotal; int opc;
    // the value travelled INSIDE the id word (a 4-byte integer column of a narrow sampled range above the id's bits): word = id | (v - pmin) << pshift
    uint32_t packed, idmask, pshift, pmin;
};
// value of row i as the operand of accumulator `which` (0: the value, 1: its square in the promoted type) -- wave-uniform dtype switch
__device__ inline uint64_t gid_operand(const GidAgg& a, size_t i, int kind, int square) {
    switch (a.vdt) {
    case AQG_INT8: return val_operand_t((int8_t)(uint8_t)static_cast<const uint32_t*>(a.val)[i], kind, square);      // (1- / 2-byte columns travel widened)
    case AQG_INT16: return val_operand_t((int16_t)(uint16_t)static_cast<const uint32_t*>(a.val)[i], kind, square);
    case AQG_UINT8: case AQG_BOOL: return val_operand_t((uint8_t)static_cast<const uint32_t*>(a.val)[i], kind, square);
    case AQG_UINT16: return val_operand_t((uint16_t)static_cast<const uint32_t*>(a.val)[i], kind, square);
    case AQG_INT32: return val_operand_t(static_cast<const int32_t*>(a.val)[i], kind, square);
    case AQG_UINT32: return val_operand_t(static_cast<const uint32_t*>(a.val)[i], kind, square);
    case AQG_FLOAT: return val_operand_t(static_cast<const float*>(a.val)[i], kind, square);
    case AQG_INT64: return val_operand_t(static_cast<const int64_t*>(a.val)[i], kind, square);
    case AQG_UINT64: return val_operand_t(static_cast<const uint64_t*>(a.val)[i], kind, square);
    default: return val_operand_t(static_cast<const double*>(a.val)[i], kind, square);
    }
}
template <class T> __device__ __noinline__ void gid_store_minmax(void* out, uint32_t g, uint64_t mapped, bool is_max) {
    T v;
    if constexpr (std::is_floating_point_v<T>) { v = (T)unmap_f(mapped); if (is_max) { T seed = dlimits<T>::min(); v = seed > v ? seed : v; } }   // (D8: max seeds with numeric_limits<T>::min())
    else if constexpr (std::is_unsigned_v<T>) v = (T)mapped;
    else v = (T)unmap_i(mapped);
    static_cast<T*>(out)[g] = v;
}
// V8: 8-byte values.  A lane takes GR consecutive rows of a step by 16-byte loads (4-byte aligned: a partition starts anywhere) and the next
// step's rows are in flight while this step's are accumulated -- with four rows per lane and step by dword loads, one step at a time, the
// kernel read at 1.95 TB/s (h2o v3 at 1e9 rows / 1e7 groups: 4.1 of the call's 10.6 ms).
template <bool V8>
__global__ void __launch_bounds__(1024, 8) gid_agg_kernel(GidAgg a) {      // (eight wavefronts per SIMD: two workgroups per CU)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t* acc0 = reinterpret_cast<uint64_t*>(smem_raw);
    uint64_t* acc1 = acc0 + a.cap;
    const bool two = a.op == AQG_RED_VAR || a.op == AQG_RED_STDDEV;
    const int vc = vclass(a.vdt);
    const int kind = a.op == AQG_RED_MIN ? ACC_MIN : a.op == AQG_RED_MAX ? ACC_MAX : vc == VC_F ? ACC_ADD_F : ACC_ADD_I;
    for (uint32_t p = blockIdx.x; p < a.nparts; p += gridDim.x) {
        const uint32_t g0 = a.pfirst[p], width = a.pfirst[p + 1] - g0;
        const uint32_t r0 = a.pstart[p], r1 = a.pstart[p + 1];
        for (uint32_t j = threadIdx.x; j < width; j += 1024) { acc0[j] = acc_init(kind); if (two) acc1[j] = 0; }
        __syncthreads();
        if (r0 < r1) {
            using VT = std::conditional_t<V8, uint64_t, uint32_t>;
            constexpr int GR = V8 ? 4 : 8;
            constexpr uint32_t STEP = 1024 * GR;
            struct Batch { uint32_t w[GR]; VT x[GR]; };
            const bool has_val = !a.packed;
            auto load_full = [&](uint32_t i0, Batch& t) {
                const uint32_t o = i0 + threadIdx.x * GR;
                __builtin_memcpy(t.w, a.gid + o, sizeof t.w);
                if (has_val) __builtin_memcpy(t.x, static_cast<const VT*>(a.val) + o, sizeof t.x);
            };
            auto load_edge = [&](uint32_t i0, Batch& t) {
                const uint32_t o = i0 + threadIdx.x * GR;
#pragma unroll
                for (int k = 0; k < GR; ++k) {
                    const uint32_t i = o + k < r1 ? o + k : r1 - 1;
                    t.w[k] = a.gid[i];
                    if (has_val) t.x[k] = static_cast<const VT*>(a.val)[i];
                }
            };
            const uint32_t nfull = (r1 - r0) / STEP, nsteps = nfull + ((r1 - r0) % STEP ? 1u : 0u);
            const uint32_t safe_last = nfull ? r0 + (nfull - 1) * STEP : (r0 + STEP <= a.ntotal ? r0 : a.ntotal - STEP);   // a whole step inside the arrays, for the prefetch that has nothing left to fetch
            Batch cur;
            load_full(nfull ? r0 : safe_last, cur);
            uint32_t i0 = r0;
            for (uint32_t st = 0; st < nsteps; ++st, i0 += STEP) {
                const bool edge = st >= nfull;
                if (edge) load_edge(i0, cur);
                Batch nxt;
                load_full(st + 1 < nfull ? i0 + STEP : safe_last, nxt);
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t o = i0 + threadIdx.x * GR;
                if (has_val && !two && a.opc != OPC_GENERIC) {                  // plain sums: straight-line rows (entry cap - 1 is nobody's: rows beyond the edge go there)
                    uint32_t g[GR];
#pragma unroll
                    for (int k = 0; k < GR; ++k) g[k] = edge && !(o + k < r1) ? a.cap - 1 : (cur.w[k] & a.idmask) - g0;
                    switch (a.opc) {
                    case OPC_ADDI_I32:
#pragma unroll
                        for (int k = 0; k < GR; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(&acc0[g[k]]), (unsigned long long)(long long)(int32_t)(uint32_t)cur.x[k]);
                        break;
                    case OPC_ADDI_U32:
#pragma unroll
                        for (int k = 0; k < GR; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(&acc0[g[k]]), (unsigned long long)(uint32_t)cur.x[k]);
                        break;
                    case OPC_ADDF_F32:
#pragma unroll
                        for (int k = 0; k < GR; ++k) atomicAdd(reinterpret_cast<double*>(&acc0[g[k]]), (double)__uint_as_float((uint32_t)cur.x[k]));
                        break;
                    default:
#pragma unroll
                        for (int k = 0; k < GR; ++k) atomicAdd(reinterpret_cast<double*>(&acc0[g[k]]), __builtin_bit_cast(double, (uint64_t)cur.x[k]));
                        break;
                    }
                } else
#pragma unroll
                for (int k = 0; k < GR; ++k) {
                    if (edge && !(o + k < r1)) continue;
                    const uint32_t g = (cur.w[k] & a.idmask) - g0;
                    uint64_t v, q = 0;
                    if (a.packed) {
                        const uint32_t raw = (cur.w[k] >> a.pshift) + a.pmin;
                        v = a.vdt == AQG_INT32 ? val_operand_t((int32_t)raw, kind, 0) : val_operand_t(raw, kind, 0);
                        if (two) q = a.vdt == AQG_INT32 ? val_operand_t((int32_t)raw, kind, 1) : val_operand_t(raw, kind, 1);
                    } else {
                        v = val_operand_bits(a.vdt, (uint64_t)cur.x[k], kind, 0, 0);
                        if (two) q = val_operand_bits(a.vdt, (uint64_t)cur.x[k], kind, 1, 0);
                    }
                    acc_apply(&acc0[g], kind, v);
                    if (two) acc_apply(&acc1[g], kind, q);
                }
                cur = nxt;
            }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < width; j += 1024) {
            const uint32_t gg = g0 + j;
            const uint64_t s = acc0[j];
            switch (a.op) {
            case AQG_RED_SUM:
                if (vc == VC_F) static_cast<double*>(a.out)[gg] = __builtin_bit_cast(double, s);
                else static_cast<aqg_i128*>(a.out)[gg] = vc == VC_U ? i128_from_u64(s) : i128_from_i64((int64_t)s);
                break;
            case AQG_RED_AVG: {
                const double sd = vc == VC_F ? __builtin_bit_cast(double, s) : vc == VC_U ? (double)s : (double)(int64_t)s;
                static_cast<double*>(a.out)[gg] = sd / (double)a.counts[gg];
            } break;
            case AQG_RED_VAR: case AQG_RED_STDDEV: {                            // (ssq - s * s / (n + 1)) / (n + 1): D9 kept
                const double np1 = (double)(uint32_t)(a.counts[gg] + 1);
                double d;
                if (vc == VC_F) { const double sd = __builtin_bit_cast(double, s), qd = __builtin_bit_cast(double, acc1[j]); d = (qd - sd * sd / np1) / np1; }
                else {
                    const aqg_i128 sm = vc == VC_U ? i128_from_u64(s) : i128_from_i64((int64_t)s);
                    const aqg_i128 qq = vc == VC_U ? i128_from_u64(acc1[j]) : i128_from_i64((int64_t)acc1[j]);
                    const aqg_i128 ss = mul_128_p1(sm, sm);
                    const double sq = vc == VC_U ? u128_to_double(ss.hi, ss.lo) : i128_to_double(ss);
                    const double qdd = vc == VC_U ? u128_to_double(qq.hi, qq.lo) : i128_to_double(qq);
                    d = (qdd - sq / np1) / np1;
                }
                static_cast<double*>(a.out)[gg] = a.op == AQG_RED_STDDEV ? sqrt(d) : d;
            } break;
            default: {
                const bool mx = a.op == AQG_RED_MAX;
                switch (a.vdt) {
                case AQG_INT8: gid_store_minmax<int8_t>(a.out, gg, s, mx); break;
                case AQG_INT16: gid_store_minmax<int16_t>(a.out, gg, s, mx); break;
                case AQG_INT32: gid_store_minmax<int32_t>(a.out, gg, s, mx); break;
                case AQG_INT64: gid_store_minmax<int64_t>(a.out, gg, s, mx); break;
                case AQG_UINT8: case AQG_BOOL: gid_store_minmax<uint8_t>(a.out, gg, s, mx); break;
                case AQG_UINT16: gid_store_minmax<uint16_t>(a.out, gg, s, mx); break;
                case AQG_UINT32: gid_store_minmax<uint32_t>(a.out, gg, s, mx); break;
                case AQG_UINT64: gid_store_minmax<uint64_t>(a.out, gg, s, mx); break;
                case AQG_FLOAT: gid_store_minmax<float>(a.out, gg, s, mx); break;
                default: gid_store_minmax<double>(a.out, gg, s, mx); break;
                }
            } break;
            }
        }
        __syncthreads();
    }
}

} // namespace

// out[g] = op(x[rows whose id is g]) for dense ids 0 .. G-1 with known group sizes (offsets = their exclusive scan, G + 1 entries).
// AQG_ERR_DTYPE: this (op, dtype) is not served here (8-byte integer sums need 128 bits per group): the caller takes the hashed plans.
static int gid_reduce_impl(aqg_ctx* ctx, const uint32_t* gid, const uint32_t* offsets, const uint32_t* counts, uint32_t n, uint32_t G, int op, int t, const void* x, void* out_dev, bool allow_pack);
int aqg_gid_reduce(aqg_ctx* ctx, const uint32_t* gid, const uint32_t* offsets, const uint32_t* counts, uint32_t n, uint32_t G, int op, int t, const void* x, void* out_dev) {
    const int rc = gid_reduce_impl(ctx, gid, offsets, counts, n, G, op, t, x, out_dev, true);
    return rc == -1001 ? gid_reduce_impl(ctx, gid, offsets, counts, n, G, op, t, x, out_dev, false) : rc;     // (a value outside the sampled range of its field: once more, as its own plane)
}
static int gid_reduce_impl(aqg_ctx* ctx, const uint32_t* gid, const uint32_t* offsets, const uint32_t* counts, uint32_t n, uint32_t G, int op, int t, const void* x, void* out_dev, bool allow_pack) {
    const bool two = op == AQG_RED_VAR || op == AQG_RED_STDDEV;
    if (!(op == AQG_RED_SUM || op == AQG_RED_AVG || op == AQG_RED_MIN || op == AQG_RED_MAX || two)) return AQG_ERR_DTYPE;
    const bool wide_int = t == AQG_INT64 || t == AQG_UINT64;
    if (wide_int && op != AQG_RED_MIN && op != AQG_RED_MAX) return AQG_ERR_DTYPE;
    const int esz = (int)aqg_dtype_size(t), vsz = esz == 8 ? 8 : 4;
    // ids of one partition: at most 78 KB of accumulators (two 1024-thread workgroups per CU), and at least 1024 partitions whatever the
    // group count -- the aggregation runs one workgroup per partition (1e5 groups in 8 partitions: 119 ms; in 1024: see DESIGN.md)
    const uint32_t cap_max = two ? 4992u : 9984u;
    uint32_t bits = 10;
    while (bits < 21 && ((uint64_t)G >> bits) + 2 > cap_max) ++bits;
    if (((uint64_t)G >> bits) + 2 > cap_max || G <= (8u << bits)) return AQG_ERR_DTYPE;
    const uint32_t cap = (uint32_t)((uint64_t)G >> bits) + 2;
    const uint32_t levels = (bits + 6) / 7, PP = 1u << bits;
    const uint32_t M = (uint32_t)((((uint64_t)1 << bits) << 32) / G);
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, ((size_t)n + 64) * (2 * (4 + (size_t)vsz) + (esz < 4 ? 4 : 0)) + (size_t)PP * 40 + 1048576));
    uint32_t *gA, *gB, *pstart, *pfirst, *seg, *tp, *cur;
    void *vA, *vB;
    AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &gA));
    AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &gB));
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * vsz, &vA));
    AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * vsz, &vB));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &pstart));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &pfirst));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &seg));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &tp));
    AQG_TRY(aqg_ws_get(ctx, (size_t)PP + 2, &cur));
    const void* vsrc = x;
    if (esz < 4) {
        uint32_t* wide;
        AQG_TRY(aqg_ws_get(ctx, (size_t)n + 64, &wide));
        hipLaunchKernelGGL(p1_widen_kernel, dim3(aqg_grid(ctx, n, 256, 4, 16)), dim3(256), 0, ctx->stream, x, esz, n, wide);
        vsrc = wide;
    }
    // a 4-byte integer value column of a narrow sampled range travels INSIDE the id word (ids below 2^24 leave eight bits: h2o v1, v2): one plane per
    // level instead of two.  Every row is verified while it is packed (p2_scatter's PL_PACK); a miss repeats the call with the value as its own plane.
    uint32_t pk_on = 0, pk_shift = 0, pk_min = 0, pk_mask = 0, *pk_flag = nullptr;
    static const bool pack_off = getenv("AQG_DISABLE_PACK") != nullptr;
    if (allow_pack && !pack_off && (t == AQG_INT32 || t == AQG_UINT32) && n >= (1u << 22) && ((uintptr_t)x & 15) == 0) {
        KeySpec probe;
        memset(&probe, 0, sizeof probe);
        probe.nkeys = 1; probe.dt[0] = t; probe.col[0] = x;
        long long mn[MAXKEYS], mx[MAXKEYS];
        bool ok = false;
        AQG_TRY(aqg_key_ranges(ctx, probe, 1u << 20, mn, mx, &ok, n));
        int gbits = 1;
        while (gbits < 32 && (1ull << gbits) < (unsigned long long)G) ++gbits;
        if (ok && mx[0] >= mn[0]) {
            const unsigned long long range = (unsigned long long)(mx[0] - mn[0]);
            int fb = 1;
            while (fb < 32 && (1ull << fb) <= range) ++fb;
            if (gbits + fb <= 32) {
                pk_on = 1; pk_shift = (uint32_t)gbits; pk_min = (uint32_t)mn[0]; pk_mask = (uint32_t)((1ull << fb) - 1);
                AQG_TRY(aqg_ws_get(ctx, 16, &pk_flag));
                AQG_HIP(ctx, hipMemsetAsync(pk_flag, 0, 4, ctx->stream));
            }
        }
    }
    const uint32_t kclear = pk_on ? pk_mask << pk_shift : 0u;
    hipLaunchKernelGGL(gid_setup_kernel, dim3(aqg_grid(ctx, (uint64_t)PP + 1, 256, 1, 4)), dim3(256), 0, ctx->stream, offsets, G, M, PP, pstart, pfirst);
    const size_t scat_lds = (size_t)P2_PT * 4;
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, true, false>), scat_lds));
    AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, false, false>), scat_lds));
    if (pk_on) {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, true, BIN_RAW, true>), scat_lds));
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&p2_scatter_kernel<P2_TB, P2_TR, false, false, BIN_RAW, true>), scat_lds));
    }
    uint32_t nseg = 1, bits_left = bits;
    const uint32_t* gsrc = gid;
    const void* vs = vsrc;
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t lb = (bits_left + (levels - l) - 1) / (levels - l);        // bits of this level, most significant first
        bits_left -= lb;
        const uint32_t nb = 1u << lb, shift = bits_left, mask = nb - 1;
        // this level's segments are the partitions of the levels before it, its cursors the starts of its own partitions: entries of pstart
        hipLaunchKernelGGL(pn_gather_strided_kernel, dim3(aqg_grid(ctx, (uint64_t)nseg + 1, 256, 1, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)pstart, nb << shift, nseg + 1, seg);
        hipLaunchKernelGGL(pn_gather_strided_kernel, dim3(aqg_grid(ctx, (uint64_t)nseg * nb, 256, 1, 4)), dim3(256), 0, ctx->stream, (const uint32_t*)pstart, 1u << shift, nseg * nb, cur);
        hipLaunchKernelGGL(pn_tiles_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg, nseg, (uint32_t)P2_PT, tp);
        uint32_t* gdst = (l & 1) ? gB : gA;
        void* vdst = (l & 1) ? vB : vA;
        Planes pl;
        memset(&pl, 0, sizeof pl);
        auto add = [&](const void* s_, int sstride, int soff, void* d, int dstride, int doff) {
            Plane& Q = pl.p[pl.n++];
            Q.kind = PL_LOAD; Q.src = static_cast<const uint32_t*>(s_); Q.src_stride_dw = sstride; Q.src_off_dw = soff;
            Q.dst = static_cast<uint32_t*>(d); Q.dst_stride_dw = dstride; Q.dst_off_dw = doff;
        };
        add(gsrc, 1, 0, gdst, 1, 0);
        if (pk_on) {
            if (l == 0) {
                pl.p[0].kind = PL_PACK;
                pl.pk.n = 1; pl.pk.kmax = 0xFFFFFFFFu; pl.pk.flag = pk_flag;
                pl.pk.src[0] = static_cast<const uint32_t*>(x); pl.pk.min[0] = pk_min; pl.pk.shift[0] = pk_shift; pl.pk.fmask[0] = pk_mask;
            }
        }
        else if (vsz == 4) add(vs, 1, 0, vdst, 1, 0); else { add(vs, 2, 0, vdst, 2, 0); add(vs, 2, 1, vdst, 2, 1); }
        P2Level lv{seg, tp, cur, nseg, M, shift, mask, nb, nb, l == 0 ? 0u : kclear};
        const unsigned tiles = (unsigned)((uint64_t)n / P2_PT) + nseg + 1;
        if (pk_on && l == 0) {
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, true, BIN_RAW, true>), dim3(tiles), dim3(P2_TB), scat_lds, ctx->stream, gsrc, pl, lv);
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, false, BIN_RAW, true>), dim3(nseg), dim3(P2_TB), scat_lds, ctx->stream, gsrc, pl, lv);
        } else {
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, true, false>), dim3(tiles), dim3(P2_TB), scat_lds, ctx->stream, gsrc, pl, lv);
            hipLaunchKernelGGL((p2_scatter_kernel<P2_TB, P2_TR, false, false, false>), dim3(nseg), dim3(P2_TB), scat_lds, ctx->stream, gsrc, pl, lv);
        }
        AQG_TRY(aqg_check_launch(ctx, "id-partitioned grouped reduce: level"));
        if (pk_on && l == 0) {                                  // a row that did not fit its field: the caller repeats the call unpacked
            uint32_t miss = 0;
            AQG_HIP(ctx, hipMemcpyAsync(&miss, pk_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
            AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (miss) return -1001;
        }
        nseg *= nb;
        gsrc = gdst; vs = vdst;
    }
    GidAgg a;
    a.gid = gsrc; a.val = vs; a.vdt = t; a.op = op; a.pstart = pstart; a.pfirst = pfirst; a.counts = counts; a.out = out_dev; a.nparts = PP; a.cap = cap;
    a.packed = pk_on; a.idmask = pk_on ? ~kclear : 0xFFFFFFFFu; a.pshift = pk_shift; a.pmin = pk_min;
    a.ntotal = n;
    a.opc = OPC_GENERIC;
    if (op == AQG_RED_SUM || op == AQG_RED_AVG) a.opc = t == AQG_INT32 ? OPC_ADDI_I32 : t == AQG_UINT32 ? OPC_ADDI_U32 : t == AQG_FLOAT ? OPC_ADDF_F32 : t == AQG_DOUBLE ? OPC_ADDF_F64 : OPC_GENERIC;
    if (n < 1024u * 8u) return AQG_ERR_DTYPE;                                 // (the kernel prefetches whole steps of 8192 rows; inputs this small never come here)
    const size_t lds = (size_t)cap * 8 * (two ? 2 : 1);
    auto launch = [&](auto kern) -> int {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(PP < 4096 ? PP : 4096), dim3(1024), lds, ctx->stream, a);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "gid_agg_kernel");
    };
    return vsz == 8 ? launch(&gid_agg_kernel<true>) : launch(&gid_agg_kernel<false>);
}

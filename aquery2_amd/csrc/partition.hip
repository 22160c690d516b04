// partition.hip -- high-cardinality group-by without global atomics: the plan of aqg_groupby_agg for more groups than LDS
// holds (replaces AQHashTable's robin-hood build, reference server/hasher.h:146-199 + server/unordered_dense.h:1117-1147, and
// the generated per-group loop engine/ast.py:722-789; h2o Q3 / Q5 / Q7).
//
// Scattered device-scope atomics top out near 3e10 per second on MI355X (they execute at the memory side; measured: h2o Q5,
// 1e9 rows, 1e7 groups: 130-140 ms through the HBM table), so groups that do not fit a workgroup's LDS table are handled by
// PARTITIONING the rows instead: {packed key, row id, value columns} are radix-partitioned on the top bits of a 32-bit key
// hash (MSD, two levels of <= 8 bits) until a partition holds about a thousand groups; each partition is then aggregated in
// LDS by one workgroup and its groups are appended to a compact record table.  The compact table feeds the same
// collect / rank / emit kernels as the hash path (first-occurrence order comes from the carried row ids).
//
// What bounds these kernels is memory-level parallelism, not the access pattern: every thread first issues ALL the loads of
// its rows of a tile (SR independent loads in flight, index clamped instead of branched), and only then ranks / stages them.
// (Round-1 measurements, 1e9 rows: the same scatter with one load -> use -> next load per round ran 57-66 ms per level;
// writing to linear instead of scattered positions did not change that, i.e. the scattered 128-byte runs were never the cost.)
#include "groupby_dev.hpp"

namespace {

constexpr int SB = 1024;          // threads per workgroup of the histogram / scatter kernels
constexpr int SR = 16;            // rows per thread and tile, all loaded before first use
constexpr int PT = SB * SR;       // rows per tile (16384: 128 KB of LDS for one staged plane + its destinations): ~64 rows = 256 B per bin
                                  // and tile at 256 bins (8192-row tiles: the 8-bit levels took 15 ms instead of 12 per 1e9 rows)
constexpr int MAXPL = 4 + 2 * MAXACC;

struct KeyIn {            // where a pass reads the packed key of row i from
    int from_cols;        // 1: pack from the user's key columns; 0: record array
    KeySpec ks;
    const void* rec;      // record keys
    int ksz;              // 4 or 8 bytes per record key
    uint32_t pbits;       // partition id = top pbits bits of the key hash
};
__device__ inline uint32_t part_id(uint64_t key, uint32_t pbits) { return lds_h1<false>(key) >> (32 - pbits); }

// A lane owns SR / 4 groups of FOUR consecutive rows of a tile: group c covers rows c*4*SB + tid*4 .. +3, so every 4-byte column
// is read with one 16-byte load per group and the lanes of a wavefront read 1 KB contiguously (rows of a partition start at
// arbitrary offsets: the loads are only element-aligned, which gfx950 global loads allow).  Groups that run past the tile's
// end read a clamped index row by row instead.
__device__ inline uint32_t tile_row(int r) { return (uint32_t)(r >> 2) * (SB * 4) + threadIdx.x * 4 + (r & 3); }
template <class T> __device__ inline void load_rows(const T* __restrict__ p, size_t tile_first, uint32_t nrows, T (&t)[SR]) {
#pragma unroll
    for (int c = 0; c < SR / 4; ++c) {
        const uint32_t o = tile_row(4 * c);
        if (o + 4 <= nrows) __builtin_memcpy(&t[4 * c], p + tile_first + o, 4 * sizeof(T));
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) t[4 * c + q] = p[tile_first + (o + q < nrows ? o + q : nrows - 1)];
        }
    }
}
template <class T> __device__ inline void or_key_rows(const void* col, size_t tile_first, uint32_t nrows, int sh, uint64_t (&key)[SR]) {
    T t[SR];
    load_rows(static_cast<const T*>(col), tile_first, nrows, t);
#pragma unroll
    for (int r = 0; r < SR; ++r) key[r] |= (uint64_t)t[r] << sh;
}
// packed keys of this lane's SR rows of the tile [rb, re)
__device__ inline void load_keys(const KeyIn& k, uint32_t rb, uint32_t re, uint64_t (&key)[SR]) {
    const uint32_t nrows = re - rb;
#pragma unroll
    for (int r = 0; r < SR; ++r) key[r] = 0;
    if (!k.from_cols) {
        if (k.ksz == 4) or_key_rows<uint32_t>(k.rec, rb, nrows, 0, key);
        else or_key_rows<uint64_t>(k.rec, rb, nrows, 0, key);
        return;
    }
    for (int c = 0; c < k.ks.nkeys; ++c) {
        const int sh = c ? k.ks.shift[c] : 0;
        switch (aqg_dtype_size_dev(k.ks.dt[c])) {
        case 1: or_key_rows<uint8_t>(k.ks.col[c], rb, nrows, sh, key); break;
        case 2: or_key_rows<uint16_t>(k.ks.col[c], rb, nrows, sh, key); break;
        case 4: or_key_rows<uint32_t>(k.ks.col[c], rb, nrows, sh, key); break;
        default: or_key_rows<uint64_t>(k.ks.col[c], rb, nrows, sh, key); break;
        }
    }
}

// Everything a scatter pass moves is a DWORD PLANE: one 32-bit word per row, read from a source (or made from the key / row
// index already in registers) and written to a destination at a dword stride.  8-byte columns are two planes.
enum : int { PL_LOAD = 0, PL_KEYLO = 1, PL_KEYHI = 2, PL_ROWIDX = 3 };
struct Plane {
    const unsigned char* src; int src_esz; int src_stride; int src_off;   // PL_LOAD: esz (1, 2, 4) bytes at src + row * stride + off
    uint32_t* dst; int dst_stride_dw; int dst_off_dw;
    int kind;
};
struct Planes { int n; Plane p[MAXPL]; };

// ---- MSD partitioning, up to two levels of <= 8 hash bits ---------------------------------------------------------------
// A level splits every SEGMENT of the current record set (level 1: the whole input = one segment; level 2: each level-1 bin)
// into 2^bits bins.  Tiles of PT rows never straddle segments: tile t belongs to segment b = upper_bound(tile_prefix, t) - 1.
// hist layout [segment][bin][tile in segment]: one global exclusive scan of it yields absolute destinations, because the
// counts of a segment add up to its length.  Ranks inside a tile come from returning LDS atomics (order inside a bin is
// irrelevant for aggregation; row ids travel with the records).
struct Segs {
    const uint32_t* seg_start;    // [nseg+1] row range of each segment
    const uint32_t* tile_prefix;  // [nseg+1] first tile of each segment
    uint32_t nseg;
};
__device__ inline bool tile_range(const Segs& sg, uint32_t t, uint32_t& seg, uint32_t& tin, uint32_t& ntseg, uint32_t& rb, uint32_t& re) {
    if (t >= sg.tile_prefix[sg.nseg]) return false;
    uint32_t lo = 0, hi = sg.nseg;                       // largest seg with tile_prefix[seg] <= t
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (sg.tile_prefix[mid] <= t) lo = mid; else hi = mid; }
    seg = lo;
    tin = t - sg.tile_prefix[seg];
    ntseg = sg.tile_prefix[seg + 1] - sg.tile_prefix[seg];
    rb = sg.seg_start[seg] + tin * PT;
    re = rb + PT < sg.seg_start[seg + 1] ? rb + PT : sg.seg_start[seg + 1];
    return true;
}

__global__ void __launch_bounds__(SB) part_hist_kernel(KeyIn kin, Segs sg, uint32_t shift, uint32_t bits, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    const uint32_t nb = 1u << bits;
    uint32_t seg, tin, ntseg, rb, re;
    if (!tile_range(sg, blockIdx.x, seg, tin, ntseg, rb, re)) return;
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    // which rows a lane counts is irrelevant here; for single 4-byte keys lane-strided scalar loads measured 0.3 ms per
    // 1e9 rows faster than the scatter's 16-byte groups (0.9 vs 1.2 ms), so that shape takes them
    const bool one_u32 = kin.from_cols ? (kin.ks.nkeys == 1 && aqg_dtype_size_dev(kin.ks.dt[0]) == 4) : kin.ksz == 4;
    if (one_u32) {
        const uint32_t* kp = static_cast<const uint32_t*>(kin.from_cols ? kin.ks.col[0] : kin.rec);
        uint32_t k32[SR];
#pragma unroll
        for (int r = 0; r < SR; ++r) { const uint32_t j = rb + r * SB + threadIdx.x; k32[r] = kp[j < re ? j : re - 1]; }
#pragma unroll
        for (int r = 0; r < SR; ++r)
            if (rb + r * SB + threadIdx.x < re) atomicAdd(&h[(part_id((uint64_t)k32[r], kin.pbits) >> shift) & (nb - 1)], 1u);
    } else {
        uint64_t key[SR];
        load_keys(kin, rb, re, key);
#pragma unroll
        for (int r = 0; r < SR; ++r)
            if (rb + tile_row(r) < re) atomicAdd(&h[(part_id(key[r], kin.pbits) >> shift) & (nb - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < nb) hist[(size_t)sg.tile_prefix[seg] * nb + (size_t)threadIdx.x * ntseg + tin] = h[threadIdx.x];
}

// Scatter with LDS-staged, coalesced writes: a tile's rows are ranked inside their bins (returning LDS atomics), each plane is
// laid out bin-major in LDS and streamed out so that consecutive lanes write consecutive addresses of a bin's run.
__global__ void __launch_bounds__(SB) part_scatter_kernel(KeyIn kin, Planes pl, Segs sg, uint32_t shift, uint32_t bits, const uint32_t* __restrict__ hist_scanned) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* stage = reinterpret_cast<uint32_t*>(smem_raw);      // [PT] one plane of the tile, bin-major
    uint32_t* delta = stage + PT;                                 // [PT] destination row minus staged position
    __shared__ uint32_t gbase[256], cnt[256], lbase[256], wsum[4];
    const uint32_t nb = 1u << bits;
    uint32_t seg, tin, ntseg, rb, re;
    if (!tile_range(sg, blockIdx.x, seg, tin, ntseg, rb, re)) return;
    const uint32_t nrows = re - rb;
    if (threadIdx.x < 256) {
        gbase[threadIdx.x] = threadIdx.x < nb ? hist_scanned[(size_t)sg.tile_prefix[seg] * nb + (size_t)threadIdx.x * ntseg + tin] : 0;
        cnt[threadIdx.x] = 0;
    }
    __syncthreads();
    uint64_t key[SR];
    load_keys(kin, rb, re, key);
    uint32_t pos[SR];                                             // (bin << 16) | rank, later the staged position
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const uint32_t d = (part_id(key[r], kin.pbits) >> shift) & (nb - 1);
        pos[r] = tile_row(r) < nrows ? (d << 16) | atomicAdd(&cnt[d], 1u) : 0;
    }
    __syncthreads();
    uint32_t c = 0, incl = 0;
    if (threadIdx.x < 256) {   // exclusive scan of the 256 bin counts by the first four wavefronts
        c = cnt[threadIdx.x];
        incl = wave_scan_incl(c, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t base = 0;
        for (int w = 0; w < wave_id(); ++w) base += wsum[w];
        lbase[threadIdx.x] = base + incl - c;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        if (tile_row(r) < nrows) {
            const uint32_t d = pos[r] >> 16, lb = lbase[d], p = lb + (pos[r] & 0xFFFF);
            pos[r] = p;
            delta[p] = gbase[d] - lb;
        }
    }
    for (int ci = 0; ci < pl.n; ++ci) {
        const Plane& P = pl.p[ci];
        uint32_t v[SR];
        if (P.kind == PL_KEYLO) {
#pragma unroll
            for (int r = 0; r < SR; ++r) v[r] = (uint32_t)key[r];
        } else if (P.kind == PL_KEYHI) {
#pragma unroll
            for (int r = 0; r < SR; ++r) v[r] = (uint32_t)(key[r] >> 32);
        } else if (P.kind == PL_ROWIDX) {
#pragma unroll
            for (int r = 0; r < SR; ++r) v[r] = rb + tile_row(r);
        } else if (P.src_stride == 8) {              // one half of an 8-byte column
            uint64_t t[SR];
            load_rows(reinterpret_cast<const uint64_t*>(P.src), rb, nrows, t);
#pragma unroll
            for (int r = 0; r < SR; ++r) v[r] = P.src_off ? (uint32_t)(t[r] >> 32) : (uint32_t)t[r];
        } else if (P.src_esz == 4) {
            load_rows(reinterpret_cast<const uint32_t*>(P.src), rb, nrows, v);
        } else if (P.src_esz == 2) {
            uint16_t t[SR];
            load_rows(reinterpret_cast<const uint16_t*>(P.src), rb, nrows, t);
#pragma unroll
            for (int r = 0; r < SR; ++r) v[r] = t[r];
        } else {
            uint8_t t[SR];
            load_rows(P.src, rb, nrows, t);
#pragma unroll
            for (int r = 0; r < SR; ++r) v[r] = t[r];
        }
        __syncthreads();                       // the previous plane has left `stage` (and, first time, `delta` is complete)
#pragma unroll
        for (int r = 0; r < SR; ++r) if (tile_row(r) < nrows) stage[pos[r]] = v[r];
        __syncthreads();
        uint32_t* dst = P.dst + P.dst_off_dw;
        const size_t dstride = (size_t)P.dst_stride_dw;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const uint32_t j = r * SB + threadIdx.x;
            if (j < nrows) dst[(size_t)(j + delta[j]) * dstride] = stage[j];
        }
    }
}

// after a level: start row of every bin of every segment = scanned count of its first tile (or the segment's end when empty);
// these become the next level's segments (or the final partitions).  One thread per (segment, bin).
__global__ void __launch_bounds__(256) bins_to_segments_kernel(Segs sg, uint32_t bits, const uint32_t* __restrict__ hist_scanned, uint32_t n,
                                                               uint32_t* __restrict__ out_start /* [nseg << bits | +1] */) {
    const uint32_t nb = 1u << bits, total = sg.nseg << bits;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= total; i += gridDim.x * blockDim.x) {
        if (i == total) { out_start[i] = n; continue; }
        uint32_t seg = i >> bits, d = i & (nb - 1);
        uint32_t ntseg = sg.tile_prefix[seg + 1] - sg.tile_prefix[seg];
        // empty segment: every bin starts (and ends) at the segment's start
        out_start[i] = ntseg ? hist_scanned[(size_t)sg.tile_prefix[seg] * nb + (size_t)d * ntseg] : sg.seg_start[seg];
    }
}
// tile_prefix[s] = number of tiles of the segments before s (single workgroup; nseg <= 65536)
__global__ void __launch_bounds__(1024) tile_prefix_kernel(const uint32_t* __restrict__ seg_start, uint32_t nseg, uint32_t* __restrict__ tile_prefix) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base <= nseg; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < nseg ? (seg_start[i + 1] - seg_start[i] + PT - 1) / PT : 0;
        uint32_t incl = wave_scan_incl(v, OpAdd{}, lane_id());
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

// ---- aggregate each partition in LDS -------------------------------------------------------------------------------------
constexpr int AR = 4;      // rows per thread and step, loaded together
struct AggIn { const void* col[MAXACC]; int esz[MAXACC]; };   // partitioned value arrays (4- or 8-byte elements); null: the row id

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

// one workgroup per partition (grid-stride): LDS open addressing {key64, first_row, count, acc...}; groups are appended to `out`
// AB threads per workgroup.  Measured at 1e9 rows / 1e7 groups (two accumulators): what this kernel's time follows is the
// table's load factor (3.9 ms at 0.20 ... 7.2 ms at 0.33) and having at least 16 wavefronts per CU (8: 12.5 ms); 1024-thread
// workgroups were no faster than 512 (the per-partition init / emit phases with their barriers grow with the workgroup).
template <int NACC, int AB>
__global__ void __launch_bounds__(AB) part_agg_kernel(const void* __restrict__ rkeys, int ksz, const uint32_t* __restrict__ rrows, AccSpec as, AggIn in,
                                                      const uint32_t* __restrict__ pstart, uint32_t nparts, uint32_t pbits, uint32_t lcap, int need_count,
                                                      GTable out, uint32_t out_cap) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t LT = lcap + 1;                 // slot lcap: the group whose packed key equals the empty mark
    uint64_t* lkey = reinterpret_cast<uint64_t*>(smem_raw);
    uint64_t* lacc = lkey + LT;
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(lacc + (size_t)NACC * LT);
    uint32_t* lcount = lfirst + LT;
    __shared__ uint32_t lused, lemit, gbase;
    const uint32_t llimit = lcap - (lcap >> 3);
    for (uint32_t part = blockIdx.x; part < nparts; part += gridDim.x) {
        const uint32_t b = pstart[part], e = pstart[part + 1];
        if (b == e) continue;
        for (uint32_t s = threadIdx.x; s < LT; s += AB) {
            lkey[s] = EMPTY64; lfirst[s] = NOROW; lcount[s] = 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * LT + s] = acc_init(as.kind[a]);
        }
        if (threadIdx.x == 0) { lused = 0; lemit = 0; }
        __syncthreads();
        for (uint32_t i0 = b; i0 < e; i0 += AB * AR) {
            size_t idx[AR];
            bool live[AR];
#pragma unroll
            for (int q = 0; q < AR; ++q) { uint32_t i = i0 + q * AB + threadIdx.x; live[q] = i < e; idx[q] = live[q] ? i : e - 1; }
            uint64_t key[AR], vb[NACC ? NACC : 1][AR];
            uint32_t row[AR];
            if (ksz == 4) {
#pragma unroll
                for (int q = 0; q < AR; ++q) key[q] = static_cast<const uint32_t*>(rkeys)[idx[q]];
            } else {
#pragma unroll
                for (int q = 0; q < AR; ++q) key[q] = static_cast<const uint64_t*>(rkeys)[idx[q]];
            }
#pragma unroll
            for (int q = 0; q < AR; ++q) row[q] = rrows[idx[q]];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                if (!in.col[a]) {
#pragma unroll
                    for (int q = 0; q < AR; ++q) vb[a][q] = row[q];
                } else if (in.esz[a] == 4) {
#pragma unroll
                    for (int q = 0; q < AR; ++q) vb[a][q] = static_cast<const uint32_t*>(in.col[a])[idx[q]];
                } else {
#pragma unroll
                    for (int q = 0; q < AR; ++q) vb[a][q] = static_cast<const uint64_t*>(in.col[a])[idx[q]];
                }
            }
            uint32_t slot[AR];
            uint64_t w[AR];
#pragma unroll
            for (int q = 0; q < AR; ++q) { slot[q] = __umulhi(lds_h1<false>(key[q]) << pbits, lcap); w[q] = lkey[slot[q]]; }   // AR probes in flight; slot = hash bits below the partition id
            // rows that missed on their first probe walk their probe sequences TOGETHER: one round trip to LDS per step for
            // all of a lane's pending rows (a wavefront stays in this loop for its longest sequence, not for the sum of them)
            uint32_t pend = 0;
#pragma unroll
            for (int q = 0; q < AR; ++q) {
                if (!live[q]) slot[q] = FAIL;
                else if (key[q] == EMPTY64) slot[q] = lcap;
                else if (w[q] != key[q]) pend |= 1u << q;
            }
            for (uint32_t step = 0; pend && step <= lcap; ++step) {
#pragma unroll
                for (int q = 0; q < AR; ++q) {
                    if (!(pend & (1u << q))) continue;
                    uint64_t cur = w[q];
                    if (cur == EMPTY64) {
                        if (lused >= llimit) { slot[q] = FAIL; pend &= ~(1u << q); out.flags[0] = 1; continue; }   // the host re-plans
                        cur = atomicCAS(reinterpret_cast<unsigned long long*>(&lkey[slot[q]]), EMPTY64, key[q]);
                        if (cur == EMPTY64) { atomicAdd(&lused, 1u); cur = key[q]; }
                    }
                    if (cur == key[q]) { pend &= ~(1u << q); continue; }
                    slot[q] = slot[q] + 1 == lcap ? 0 : slot[q] + 1;
                }
#pragma unroll
                for (int q = 0; q < AR; ++q) if (pend & (1u << q)) w[q] = lkey[slot[q]];
            }
#pragma unroll
            for (int q = 0; q < AR; ++q) if (pend & (1u << q)) { slot[q] = FAIL; out.flags[0] = 1; }
#pragma unroll
            for (int q = 0; q < AR; ++q) if (slot[q] != FAIL && row[q] < lfirst[slot[q]]) atomicMin(&lfirst[slot[q]], row[q]);
            if (need_count) {
#pragma unroll
                for (int q = 0; q < AR; ++q) if (slot[q] != FAIL) atomicAdd(&lcount[slot[q]], 1u);
            }
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
#pragma unroll
                for (int q = 0; q < AR; ++q)
                    if (slot[q] != FAIL) acc_apply(&lacc[(size_t)a * LT + slot[q]], as.kind[a], val_operand_bits(as.dt[a] == AQG_NONE ? AQG_UINT32 : as.dt[a], vb[a][q], as.kind[a], as.square[a], as.part[a]));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t used = lused + (lfirst[lcap] != NOROW ? 1u : 0u);
            gbase = atomicAdd(&out.flags[1], used);
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < LT; s += AB) {
            if (lfirst[s] == NOROW) continue;
            uint32_t g = gbase + atomicAdd(&lemit, 1u);
            if (g >= out_cap) { out.flags[0] = 1; continue; }
            *out.key_p(g) = lkey[s];
            *out.first_p(g) = lfirst[s];
            *out.count_p(g) = lcount[s];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) *out.acc_p(a, g) = lacc[(size_t)a * LT + s];
        }
        __syncthreads();
    }
}

} // namespace

// Partitioned aggregation of (ks, as) over n rows into the compact record table `out` (AoS records, `out_cap` slots,
// flags[1] = number of groups written, flags[0] = overflow).  Needs packed (<= 8 byte) keys.
// Workspace is taken from the context arena (caller has reset it and reserved `aqg_partition_ws_bytes`).
static size_t part_val_bytes(int dt) { return aqg_dtype_size(dt) <= 4 ? 4 : 8; }   // narrow values travel widened to one dword

size_t aqg_partition_ws_bytes(uint32_t n, int ksz, const AccSpec& as, uint32_t pbits) {
    size_t per_row = (size_t)ksz + 4;
    for (int a = 0; a < as.nacc; ++a) {
        bool dup = false;
        for (int b = 0; b < a; ++b) dup |= as.col[b] == as.col[a];
        if (!dup && as.dt[a] != AQG_NONE) per_row += part_val_bytes(as.dt[a]);
    }
    const size_t max_tiles = (size_t)n / PT + 258;
    return 2 * ((size_t)n + 64) * per_row + max_tiles * 256 * 4 + (max_tiles * 256 / 2048 + 64) * 4 + ((size_t)(1u << pbits) + 600) * 16 + 65536;
}

int aqg_partition_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as_in, uint32_t n, uint32_t pbits, uint32_t lcap, int need_count,
                            GTable out, uint32_t out_cap) {
    if (pbits > 16 || pbits < 1) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "partitioned group-by: 2..65536 partitions");
    const int ksz = ks.total_bytes <= 4 ? 4 : 8;
    const uint32_t nparts = 1u << pbits;
    const uint32_t bits1 = pbits > 8 ? (pbits + 1) / 2 : pbits;   // level 1: the high bits of the partition id
    const uint32_t bits2 = pbits - bits1;                         // level 2: the rest
    const size_t max_tiles = (size_t)n / PT + 258;

    AccSpec as = as_in;
    int ucols = 0;
    const void* ucol[MAXACC]; int udt[MAXACC]; int acc_ucol[MAXACC];
    for (int a = 0; a < as.nacc; ++a) {
        acc_ucol[a] = -1;
        if (as.dt[a] == AQG_NONE) continue;
        for (int u = 0; u < ucols; ++u) if (ucol[u] == as.col[a]) acc_ucol[a] = u;
        if (acc_ucol[a] < 0) { ucol[ucols] = as.col[a]; udt[ucols] = as.dt[a]; acc_ucol[a] = ucols++; }
    }
    void* bufs[2][2 + MAXACC];
    for (int set = 0; set < 2; ++set) {
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * ksz, &bufs[set][0]));
        AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * 4, &bufs[set][1]));
        for (int u = 0; u < ucols; ++u) AQG_TRY(aqg_ws_alloc(ctx, ((size_t)n + 64) * part_val_bytes(udt[u]), &bufs[set][2 + u]));
    }
    uint32_t *hist, *bsum, *seg0, *tp0, *seg1, *tp1, *pstart;
    AQG_TRY(aqg_ws_get(ctx, max_tiles * 256, &hist));
    AQG_TRY(aqg_ws_get(ctx, max_tiles * 256 / 2048 + 64, &bsum));
    AQG_TRY(aqg_ws_get(ctx, 4, &seg0));
    AQG_TRY(aqg_ws_get(ctx, 4, &tp0));
    AQG_TRY(aqg_ws_get(ctx, 260, &seg1));
    AQG_TRY(aqg_ws_get(ctx, 260, &tp1));
    AQG_TRY(aqg_ws_get(ctx, (size_t)nparts + 4, &pstart));

    // level-1 segment table: one segment [0, n)
    uint32_t h0[2] = {0, n};
    void* st = nullptr;
    AQG_TRY(aqg_host_stage(ctx, 16, &st));
    memcpy(st, h0, 8);
    AQG_HIP(ctx, hipMemcpyAsync(seg0, st, 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg0, 1u, tp0);

    KeyIn kin;
    kin.from_cols = 1; kin.ks = ks; kin.rec = nullptr; kin.ksz = ksz; kin.pbits = pbits;
    // planes of a pass: key (from registers), row id, then every distinct value column; src < 0: the user's columns
    auto planes = [&](int src, int dst) {
        Planes pl;
        memset(&pl, 0, sizeof pl);
        auto add = [&](int kind, const void* s, int esz, int stride, int off, void* d, int dstride, int doff) {
            Plane& P = pl.p[pl.n++];
            P.kind = kind; P.src = static_cast<const unsigned char*>(s); P.src_esz = esz; P.src_stride = stride; P.src_off = off;
            P.dst = static_cast<uint32_t*>(d); P.dst_stride_dw = dstride; P.dst_off_dw = doff;
        };
        if (ksz == 4) add(PL_KEYLO, nullptr, 0, 0, 0, bufs[dst][0], 1, 0);
        else { add(PL_KEYLO, nullptr, 0, 0, 0, bufs[dst][0], 2, 0); add(PL_KEYHI, nullptr, 0, 0, 0, bufs[dst][0], 2, 1); }
        if (src < 0) add(PL_ROWIDX, nullptr, 0, 0, 0, bufs[dst][1], 1, 0);
        else add(PL_LOAD, bufs[src][1], 4, 4, 0, bufs[dst][1], 1, 0);
        for (int u = 0; u < ucols; ++u) {
            const int usz = (int)aqg_dtype_size(udt[u]);                 // element size in the user's column
            const int psz = (int)part_val_bytes(udt[u]);                 // element size in the partition buffers
            const void* s = src < 0 ? ucol[u] : bufs[src][2 + u];
            const int ssz = src < 0 ? usz : psz;
            if (psz == 4) add(PL_LOAD, s, ssz < 4 ? ssz : 4, ssz, 0, bufs[dst][2 + u], 1, 0);
            else { add(PL_LOAD, s, 4, 8, 0, bufs[dst][2 + u], 2, 0); add(PL_LOAD, s, 4, 8, 4, bufs[dst][2 + u], 2, 1); }
        }
        return pl;
    };
    const unsigned grid_tiles = (unsigned)max_tiles;
    const size_t scatter_lds = (size_t)PT * 8;   // stage + delta
    AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&part_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds));
    // ---- level 1 -------------------------------------------------------------------------------------------------------
    Segs sg0{seg0, tp0, 1u};
    {
        const uint32_t shift = bits2;                        // high bits first
        const uint64_t hcount = ((uint64_t)n / PT + 2) * (1u << bits1);
        hipLaunchKernelGGL(part_hist_kernel, dim3(grid_tiles), dim3(SB), 0, ctx->stream, kin, sg0, shift, bits1, hist);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, hist, hcount, bsum));
        hipLaunchKernelGGL(part_scatter_kernel, dim3(grid_tiles), dim3(SB), scatter_lds, ctx->stream, kin, planes(-1, 0), sg0, shift, bits1, (const uint32_t*)hist);
        uint32_t* dst_start = bits2 ? seg1 : pstart;
        hipLaunchKernelGGL(bins_to_segments_kernel, dim3(1), dim3(256), 0, ctx->stream, sg0, bits1, (const uint32_t*)hist, n, dst_start);
        AQG_TRY(aqg_check_launch(ctx, "partition level 1"));
    }
    int cur = 0;
    // ---- level 2 -------------------------------------------------------------------------------------------------------
    if (bits2) {
        const uint32_t nseg = 1u << bits1;
        hipLaunchKernelGGL(tile_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const uint32_t*)seg1, nseg, tp1);
        Segs sg1{seg1, tp1, nseg};
        kin.from_cols = 0; kin.rec = bufs[0][0];
        const uint64_t hcount = (uint64_t)max_tiles * (1u << bits2);
        AQG_HIP(ctx, hipMemsetAsync(hist, 0, hcount * 4, ctx->stream));   // unused tail tiles must read as zero in the scan
        hipLaunchKernelGGL(part_hist_kernel, dim3(grid_tiles), dim3(SB), 0, ctx->stream, kin, sg1, 0u, bits2, hist);
        AQG_TRY(aqg_exclusive_scan_u32(ctx, hist, hcount, bsum));
        hipLaunchKernelGGL(part_scatter_kernel, dim3(grid_tiles), dim3(SB), scatter_lds, ctx->stream, kin, planes(0, 1), sg1, 0u, bits2, (const uint32_t*)hist);
        hipLaunchKernelGGL(bins_to_segments_kernel, dim3(aqg_grid(ctx, nparts, 256, 1, 4)), dim3(256), 0, ctx->stream, sg1, bits2, (const uint32_t*)hist, n, pstart);
        AQG_TRY(aqg_check_launch(ctx, "partition level 2"));
        cur = 1;
    }
    // ---- aggregate each partition in LDS -------------------------------------------------------------------------------
    AggIn in;
    memset(&in, 0, sizeof in);
    for (int a = 0; a < as.nacc; ++a) {
        if (acc_ucol[a] >= 0) { in.col[a] = bufs[cur][2 + acc_ucol[a]]; in.esz[a] = (int)part_val_bytes(udt[acc_ucol[a]]); }
        else { in.col[a] = nullptr; in.esz[a] = 4; }     // row-index operands: the carried row id
    }
    const size_t lds = ((size_t)lcap + 1) * (8 + 8 * (size_t)as.nacc + 4 + 4);
    unsigned per_cu = (unsigned)((160 * 1024) / (lds + 1024));
    if (per_cu < 1) per_cu = 1;
    const unsigned block = 512;
    if (per_cu * block > 2048) per_cu = 2048 / block;
    unsigned grid = nparts < (unsigned)ctx->num_cu * per_cu ? nparts : (unsigned)ctx->num_cu * per_cu;
    auto launch = [&](auto kern) -> int {
        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, ctx->stream, (const void*)bufs[cur][0], ksz, (const uint32_t*)bufs[cur][1], as, in, (const uint32_t*)pstart, nparts,
                           pbits, lcap, need_count, out, out_cap);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "part_agg_kernel");
    };
    switch (as.nacc) {
    case 0: return launch(&part_agg_kernel<0, 512>);
    case 1: return launch(&part_agg_kernel<1, 512>);
    case 2: return launch(&part_agg_kernel<2, 512>);
    case 3: return launch(&part_agg_kernel<3, 512>);
    case 4: return launch(&part_agg_kernel<4, 512>);
    case 5: return launch(&part_agg_kernel<5, 512>);
    case 6: return launch(&part_agg_kernel<6, 512>);
    case 7: return launch(&part_agg_kernel<7, 512>);
    default: return launch(&part_agg_kernel<8, 512>);
    }
}

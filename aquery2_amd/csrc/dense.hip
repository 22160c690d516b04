// dense.hip -- group-by over a DENSE key domain: the device form of the reference's PerfectHashTable plan
// (server/hasher.h:201-357 packs key columns by `TableStats` minima and bit widths, server/table.h:76-91 populate_stats).
//
// A pass over the key columns finds every column's [min, max]; if the product of the ranges D is small, a key tuple maps to
//     idx = sum_j (key_j - min_j) * mult_j          (mixed radix, mult_j = product of the ranges before j)
// and the group table is a direct-indexed array: no key storage, no probing, no insert race.  Every workgroup keeps
// {first_row, count, accumulators}[idx] in LDS (up to 150 KB: gfx950 gives a workgroup 160 KB) and merges it into the global
// array at the end; domains beyond one LDS table take up to DENSE_MAX_PASSES passes, pass p owning idx in
// [p * per_pass, (p + 1) * per_pass).  h2o Q2 (id1, id2: 100 x 100 values, 1e9 rows): see DESIGN.md for the measured time
// against the hashed LDS table (18 ms) and the HBM table (54 ms).
// HBM traffic: the range pass reads the key columns once more (Q2: +8 B/row on 12 B/row algorithmic).
#include "groupby_dev.hpp"
#include "dense.hpp"

namespace {

// ---- [min, max] of every key column, as int64 -------------------------------------------------------------------------------
// `total` != 0: the n rows are a SPREAD sample of a column of `total` rows -- 1024 blocks of n / 1024 consecutive rows, evenly spaced (a column
// sorted or clustered by its key shows its whole range to such a sample; the first n rows of it show one end)
template <class T>
__device__ inline void col_range(const void* col, uint32_t n, uint32_t total, long long& mn, long long& mx) {
    const T* p = static_cast<const T*>(col);
    const uint32_t nchunk = n >> 2;
    uint32_t c_lo, c_hi;
    wg_span(nchunk, c_lo, c_hi);
    const uint32_t cpb = nchunk >> 10;                       // chunks of four rows per block
    for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const size_t src = total ? (size_t)(((uint64_t)(c / cpb) * (total >> 2)) >> 10) + c % cpb : (size_t)c;
        pack<T, 4> v = *reinterpret_cast<const pack<T, 4>*>(p + src * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { long long x = (long long)v.v[j]; mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
    }
    if (!total && blockIdx.x == 0 && threadIdx.x < (n & 3)) { long long x = (long long)p[(nchunk << 2) + threadIdx.x]; mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
}
__global__ void __launch_bounds__(256) key_range_kernel(KeySpec ks, uint32_t n, uint32_t total, long long* __restrict__ out /* [2 * nkeys]: min_j, max_j */) {
    __shared__ long long smn[4], smx[4];
    for (int c = 0; c < ks.nkeys; ++c) {
        long long mn = INT64_MAX, mx = INT64_MIN;
        switch (ks.dt[c]) {
        case AQG_INT8: col_range<int8_t>(ks.col[c], n, total, mn, mx); break;
        case AQG_INT16: col_range<int16_t>(ks.col[c], n, total, mn, mx); break;
        case AQG_INT32: col_range<int32_t>(ks.col[c], n, total, mn, mx); break;
        case AQG_INT64: col_range<int64_t>(ks.col[c], n, total, mn, mx); break;
        case AQG_UINT8: case AQG_BOOL: col_range<uint8_t>(ks.col[c], n, total, mn, mx); break;
        case AQG_UINT16: col_range<uint16_t>(ks.col[c], n, total, mn, mx); break;
        default: col_range<uint32_t>(ks.col[c], n, total, mn, mx); break;      // AQG_UINT32 (uint64 keys never come here)
        }
        mn = wave_reduce((int64_t)mn, OpMin{});
        mx = wave_reduce((int64_t)mx, OpMax{});
        if (lane_id() == 0) { smn[wave_id()] = mn; smx[wave_id()] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; }
            atomicMin(&out[2 * c], mn);
            atomicMax(&out[2 * c + 1], mx);
        }
        __syncthreads();
    }
}
__global__ void range_init_kernel(long long* out, int nkeys) {
    if ((int)threadIdx.x < nkeys) { out[2 * threadIdx.x] = INT64_MAX; out[2 * threadIdx.x + 1] = INT64_MIN; }
}

// ---- dense index of four consecutive rows ------------------------------------------------------------------------------------
// a digit outside [0, range) marks the row's index invalid (all ones): ranges taken from a sample may miss values
template <class T> __device__ inline void add_digit4(const void* col, size_t base, long long kmin, uint32_t mult, uint32_t range, uint32_t (&idx)[4], uint32_t (&bad)[4]) {
    pack<T, 4> v = *reinterpret_cast<const pack<T, 4>*>(static_cast<const T*>(col) + base);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned long long d = (unsigned long long)((long long)v.v[j] - kmin);
        bad[j] |= d >= range;
        idx[j] += (uint32_t)d * mult;
    }
}
template <class T> __device__ inline uint32_t digit1(const void* col, size_t i, long long kmin, uint32_t mult, uint32_t range, uint32_t& bad) {
    const unsigned long long d = (unsigned long long)((long long)static_cast<const T*>(col)[i] - kmin);
    bad |= d >= range;
    return (uint32_t)d * mult;
}
__device__ inline void dense_idx4(const KeySpec& ks, const DenseSpec& ds, size_t base, uint32_t (&idx)[4]) {
    uint32_t bad[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) idx[j] = 0;
    for (int c = 0; c < ks.nkeys; ++c) {
        switch (ks.dt[c]) {
        case AQG_INT8: add_digit4<int8_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        case AQG_INT16: add_digit4<int16_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        case AQG_INT32: add_digit4<int32_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        case AQG_INT64: add_digit4<int64_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        case AQG_UINT8: case AQG_BOOL: add_digit4<uint8_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        case AQG_UINT16: add_digit4<uint16_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        default: add_digit4<uint32_t>(ks.col[c], base, ds.kmin[c], ds.mult[c], ds.range[c], idx, bad); break;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (bad[j]) idx[j] = 0xFFFFFFFFu;
}
__device__ inline uint32_t dense_idx1(const KeySpec& ks, const DenseSpec& ds, size_t i) {
    uint32_t idx = 0, bad = 0;
    for (int c = 0; c < ks.nkeys; ++c) {
        switch (ks.dt[c]) {
        case AQG_INT8: idx += digit1<int8_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        case AQG_INT16: idx += digit1<int16_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        case AQG_INT32: idx += digit1<int32_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        case AQG_INT64: idx += digit1<int64_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        case AQG_UINT8: case AQG_BOOL: idx += digit1<uint8_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        case AQG_UINT16: idx += digit1<uint16_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        default: idx += digit1<uint32_t>(ks.col[c], i, ds.kmin[c], ds.mult[c], ds.range[c], bad); break;
        }
    }
    return bad ? 0xFFFFFFFFu : idx;
}
// the packed key word the emit kernel expects in the table (wide tuples: any row of the group)
__device__ inline uint64_t dense_key_word(const KeySpec& ks, const DenseSpec& ds, uint32_t idx, uint32_t some_row) {
    if (ks.wide) return some_row;
    uint64_t key = 0;
    uint32_t rem = idx;
    for (int c = ks.nkeys - 1; c >= 0; --c) {
        const uint32_t digit = rem / ds.mult[c];
        rem -= digit * ds.mult[c];
        const int bytes = aqg_dtype_size_dev(ks.dt[c]);
        const uint64_t mask = bytes == 8 ? ~0ull : ((1ull << (8 * bytes)) - 1ull);
        key |= ((uint64_t)(ds.kmin[c] + (long long)digit) & mask) << ks.shift[c];
    }
    return key;
}

// NK > 0: exactly NK key columns, all int32 / uint32.  The generic form (NK = 0) walks ks / ds with run-time indices, which keeps those
// structs in kernel-argument memory: ~20 scalar loads per step INSIDE the loop, and every s_waitcnt lgkmcnt(0) behind one also waits
// for the LDS atomics in flight, so the next step's global loads start only after the table traffic of this one has drained (h2o Q2:
// 0.9 vector-memory instructions in flight per SQ where the hashed kernel keeps 4; 54 % of the HBM roofline).  With NK and NACC
// compile-time, the columns, minima, weights and accumulator kinds are copied into registers before the loop.
template <int NACC, int BLOCK, int NK = 0>
__global__ void __launch_bounds__(BLOCK) dense_agg_kernel(KeySpec ks, DenseSpec ds, AccSpec as, GTable gt, uint32_t n, int need_count) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t PP = ds.per_pass;
    const uint32_t* kcol[NK ? NK : 1]; uint32_t kmin[NK ? NK : 1], kmult[NK ? NK : 1], krange[NK ? NK : 1];
#pragma unroll
    for (int c = 0; c < NK; ++c) { kcol[c] = static_cast<const uint32_t*>(ks.col[c]); kmin[c] = (uint32_t)ds.kmin[c]; kmult[c] = ds.mult[c]; krange[c] = ds.range[c]; }
    int akind[NACC ? NACC : 1], adt[NACC ? NACC : 1], asq[NACC ? NACC : 1], apart[NACC ? NACC : 1]; const void* acol[NACC ? NACC : 1];
#pragma unroll
    for (int a = 0; a < NACC; ++a) { akind[a] = as.kind[a]; adt[a] = as.dt[a]; asq[a] = as.square[a]; apart[a] = as.part[a]; acol[a] = as.col[a]; }
    // dense index of four consecutive rows
    auto idx4 = [&](size_t base, uint32_t (&idx)[4]) {
        if constexpr (NK > 0) {
            uint32_t bad[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = 0;
#pragma unroll
            for (int c = 0; c < NK; ++c) {
                const pack<uint32_t, 4> v = *reinterpret_cast<const pack<uint32_t, 4>*>(kcol[c] + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t d = v.v[j] - kmin[c]; bad[j] |= d >= krange[c]; idx[j] += d * kmult[c]; }   // (mod 2^32: exact for either signedness)
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (bad[j]) idx[j] = 0xFFFFFFFFu;
        } else dense_idx4(ks, ds, base, idx);
    };
    uint64_t* lacc = reinterpret_cast<uint64_t*>(smem_raw);                  // [NACC][PP]
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(lacc + (size_t)NACC * PP);   // [PP]
    uint32_t* lcount = lfirst + PP;                                          // [PP] if need_count
    // A domain whose every index occurs (h2o Q2: all 100 x 100 pairs) is complete after the first ~1e5 rows of a workgroup's span; from
    // then on the first-row bookkeeping (an LDS read and a compare per row) can only matter for rows below the largest first row
    // recorded so far.  lseen[0] counts the indices with a first row, lseen[1] bounds those first rows from above (never lowered: a
    // stale bound is only conservative); rows beyond it skip the bookkeeping.  A domain with holes never gets there and runs as before.
    __shared__ uint32_t lseen[2];
    const uint32_t nchunk = n >> 2;
    for (uint32_t pass = 0; pass < ds.npass; ++pass) {
        const uint32_t lo = pass * PP;
        const uint32_t pass_groups = ds.D - lo < PP ? ds.D - lo : PP;
        if (pass) __syncthreads();
        if (threadIdx.x == 0) { lseen[0] = 0; lseen[1] = 0; }
        for (uint32_t s = threadIdx.x; s < PP; s += BLOCK) {
            lfirst[s] = NOROW;
            if (need_count) lcount[s] = 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * PP + s] = acc_init(as.kind[a]);
        }
        __syncthreads();
        // H groups of four consecutive rows per lane and step, all loads issued before the first LDS access
        constexpr int H = NACC <= 5 ? 2 : 1;                         // (more accumulators: the second group would spill)
        const uint32_t nstep = nchunk / H;                          // whole steps; the odd chunks behind them take the same path one by one
        uint32_t c_lo, c_hi;
        wg_span(nstep, c_lo, c_hi);
        auto rows = [&](const uint32_t (&idx)[4], const uint64_t (&vals)[NACC ? NACC : 1][4], size_t base) {
            uint32_t f[4], id[4];
            bool in[4];
            const uint32_t seen = *reinterpret_cast<volatile uint32_t*>(&lseen[0]), bound = *reinterpret_cast<volatile uint32_t*>(&lseen[1]);
            const bool settled = seen >= pass_groups && (uint32_t)base > bound;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (idx[j] == 0xFFFFFFFFu) gt.flags[3] = 1;           // a value outside the sampled ranges: the host redoes the call with exact ranges
                id[j] = idx[j] - lo; in[j] = idx[j] != 0xFFFFFFFFu && id[j] < PP;
            }
            if (!settled) {
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = lfirst[in[j] ? id[j] : 0];   // four LDS reads in flight
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (in[j] && (uint32_t)base + j < f[j]) {
                        const uint32_t old = atomicMin(&lfirst[id[j]], (uint32_t)base + j);
                        if (old == NOROW) { atomicMax(&lseen[1], (uint32_t)base + j); __threadfence_block(); atomicAdd(&lseen[0], 1u); }
                    }
                }
            }
            if (need_count) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (in[j]) atomicAdd(&lcount[id[j]], 1u);
            }
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                uint64_t* la = lacc + (size_t)a * PP;
                switch (akind[a]) {   // wave-uniform
                case ACC_ADD_I:
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (in[j]) atomicAdd(reinterpret_cast<unsigned long long*>(&la[id[j]]), (unsigned long long)vals[a][j]);
                    break;
                case ACC_ADD_F:
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (in[j]) atomicAdd(reinterpret_cast<double*>(&la[id[j]]), __builtin_bit_cast(double, vals[a][j]));
                    break;
                case ACC_MIN:
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (in[j]) atomicMin(reinterpret_cast<unsigned long long*>(&la[id[j]]), (unsigned long long)vals[a][j]);
                    break;
                default:
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (in[j]) atomicMax(reinterpret_cast<unsigned long long*>(&la[id[j]]), (unsigned long long)vals[a][j]);
                    break;
                }
            }
        };
        // the H groups of a lane lie BLOCK chunks apart: every load instruction of a wavefront reads 1 KB of consecutive bytes
        const uint32_t ch_lo = c_lo * H, ch_hi = c_hi * H;
        for (uint32_t c0 = ch_lo; c0 < ch_hi; c0 += BLOCK * H) {     // (uniform over the workgroup)
            uint32_t idx[H][4];
            uint64_t vals[H][NACC ? NACC : 1][4];
            bool ok[H];
#pragma unroll
            for (int h = 0; h < H; ++h) {
                const uint32_t chunk = c0 + h * BLOCK + threadIdx.x;
                ok[h] = chunk < ch_hi;
                const size_t base = (size_t)(ok[h] ? chunk : ch_lo) * 4;      // (a lane beyond the span re-reads its first chunk and drops it)
                idx4(base, idx[h]);
                _Pragma("unroll") for (int a = 0; a < NACC; ++a) val_operand4(adt[a], acol[a], base, akind[a], asq[a], apart[a], vals[h][a]);
            }
#pragma unroll
            for (int h = 0; h < H; ++h) if (ok[h]) rows(idx[h], vals[h], (size_t)(c0 + h * BLOCK + threadIdx.x) * 4);
        }
        if (blockIdx.x == 0) {                                       // chunks behind the last whole step (< H)
            const uint32_t c = nstep * H + threadIdx.x;
            if (c < nchunk) {
                const size_t base = (size_t)c * 4;
                uint32_t idx[4];
                uint64_t vals[NACC ? NACC : 1][4];
                idx4(base, idx);
                _Pragma("unroll") for (int a = 0; a < NACC; ++a) val_operand4(adt[a], acol[a], base, akind[a], asq[a], apart[a], vals[a]);
                rows(idx, vals, base);
            }
        }
        __syncthreads();
        // merge this workgroup's table into the global direct-indexed one (slot = idx)
        for (uint32_t s = threadIdx.x; s < PP; s += BLOCK) {
            const uint32_t first = lfirst[s];
            if (first == NOROW) continue;
            const uint32_t g = lo + s;
            *gt.key_p(g) = dense_key_word(ks, ds, g, first);
            atomicMin(gt.first_p(g), first);
            if (need_count) atomicAdd(gt.count_p(g), lcount[s]);
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) acc_apply(gt.acc_p(a, g), as.kind[a], lacc[(size_t)a * PP + s]);
        }
    }
    // tail rows (< 4): straight to the global table
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint32_t row = (nchunk << 2) + threadIdx.x;
        const uint32_t g = dense_idx1(ks, ds, row);
        if (g == 0xFFFFFFFFu) { gt.flags[3] = 1; return; }
        *gt.key_p(g) = dense_key_word(ks, ds, g, row);
        atomicMin(gt.first_p(g), row);
        if (need_count) atomicAdd(gt.count_p(g), 1u);
        _Pragma("unroll") for (int a = 0; a < NACC; ++a)
            acc_apply(gt.acc_p(a, g), as.kind[a], val_operand(as.dt[a], as.col[a], row, as.kind[a], as.square[a], as.part[a]));
    }
}

// second pass of aqg_groupby_build over a dense domain: reversemap[i] = dense id of row i's group, counts[g] += 1
// NK > 0: exactly NK key columns, all int32 / uint32, their specs in registers (see dense_agg_kernel: the generic form re-loads
// ks / ds from kernel-argument memory inside the loop)
// LDS_MAP: the {index -> dense id} map of the whole domain copied into LDS behind the counts (h2o Q2 keys: 1e4 + 1e4 words): a
// look-up per row out of L2 kept the pass at 5.0 ms per 1e9 rows (64 different lines per gather instruction); the workgroup then has
// 1024 threads, so that the table is paid for once per 16 wavefronts.
template <bool LDS_COUNTS, int NK = 0, bool LDS_MAP = false>
__global__ void __launch_bounds__(1024) dense_assign_kernel(KeySpec ks, DenseSpec ds, const uint32_t* __restrict__ slot_gid, uint32_t n, uint32_t G,
                                                            uint32_t* __restrict__ reversemap, uint32_t* __restrict__ counts) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* lc = reinterpret_cast<uint32_t*>(smem_raw);
    uint32_t* lm = lc + G;                                                    // [ds.D] if LDS_MAP
    if constexpr (LDS_MAP) for (uint32_t s2 = threadIdx.x; s2 < ds.D; s2 += blockDim.x) lm[s2] = slot_gid[s2];
    const uint32_t* kcol[NK ? NK : 1]; uint32_t kmin[NK ? NK : 1], kmult[NK ? NK : 1], krange[NK ? NK : 1];
#pragma unroll
    for (int c = 0; c < NK; ++c) { kcol[c] = static_cast<const uint32_t*>(ks.col[c]); kmin[c] = (uint32_t)ds.kmin[c]; kmult[c] = ds.mult[c]; krange[c] = ds.range[c]; }
    auto idx4 = [&](size_t base, uint32_t (&idx)[4]) {
        if constexpr (NK > 0) {
            uint32_t bad[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[j] = 0;
#pragma unroll
            for (int c = 0; c < NK; ++c) {
                const pack<uint32_t, 4> v = *reinterpret_cast<const pack<uint32_t, 4>*>(kcol[c] + base);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const uint32_t d = v.v[j] - kmin[c]; bad[j] |= d >= krange[c]; idx[j] += d * kmult[c]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (bad[j]) idx[j] = 0xFFFFFFFFu;
        } else dense_idx4(ks, ds, base, idx);
    };
    if constexpr (LDS_COUNTS) { for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) lc[g] = 0; __syncthreads(); }
    const uint32_t nchunk = n >> 2;
    uint32_t c_lo, c_hi;
    wg_span(nchunk, c_lo, c_hi);
    for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const size_t base = (size_t)c * 4;
        uint32_t idx[4];
        idx4(base, idx);
        pack<uint32_t, 4> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { if constexpr (LDS_MAP) o.v[j] = lm[idx[j]]; else o.v[j] = slot_gid[idx[j]]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { if constexpr (LDS_COUNTS) atomicAdd(&lc[o.v[j]], 1u); else atomicAdd(&counts[o.v[j]], 1u); }
        *reinterpret_cast<pack<uint32_t, 4>*>(reversemap + base) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint32_t row = (nchunk << 2) + threadIdx.x;
        const uint32_t g = slot_gid[dense_idx1(ks, ds, row)];
        reversemap[row] = g;
        atomicAdd(&counts[g], 1u);
    }
    if constexpr (LDS_COUNTS) {
        __syncthreads();
        for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) { const uint32_t c = lc[g]; if (c) atomicAdd(&counts[g], c); }
    }
}

} // namespace

int aqg_dense_assign(aqg_ctx* ctx, const KeySpec& ks, const DenseSpec& ds, const uint32_t* slot_gid, uint32_t n, uint32_t G, uint32_t* reversemap, uint32_t* counts) {
    const size_t lds_counts = (size_t)G * 4 + 16, lds_map = ((size_t)G + ds.D) * 4 + 16;
    if (lds_counts <= 144 * 1024) {
        const bool map = lds_map <= 144 * 1024;
        const size_t lds = map ? lds_map : lds_counts;
        const unsigned block = lds > 20 * 1024 ? 1024u : 256u;
        unsigned per_cu = (unsigned)((160 * 1024) / (lds + 1024));
        if (per_cu > 2048u / block) per_cu = 2048u / block;
        if (per_cu < 1) per_cu = 1;
        int nk = ks.nkeys <= 3 ? ks.nkeys : 0;
        for (int c = 0; c < ks.nkeys; ++c) if (ks.dt[c] != AQG_INT32 && ks.dt[c] != AQG_UINT32) nk = 0;
        auto go = [&](auto kern) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(kern, dim3(aqg_grid(ctx, n / 4 + 1, block, 2, per_cu)), dim3(block), lds, ctx->stream, ks, ds, slot_gid, n, G, reversemap, counts);
        };
        if (map) switch (nk) {
            case 1: go(&dense_assign_kernel<true, 1, true>); break;
            case 2: go(&dense_assign_kernel<true, 2, true>); break;
            case 3: go(&dense_assign_kernel<true, 3, true>); break;
            default: go(&dense_assign_kernel<true, 0, true>); break;
        } else switch (nk) {
            case 1: go(&dense_assign_kernel<true, 1, false>); break;
            case 2: go(&dense_assign_kernel<true, 2, false>); break;
            case 3: go(&dense_assign_kernel<true, 3, false>); break;
            default: go(&dense_assign_kernel<true, 0, false>); break;
        }
    } else {
        hipLaunchKernelGGL((dense_assign_kernel<false, 0, false>), dim3(aqg_grid(ctx, n / 4 + 1, 256, 2, 8)), dim3(256), 0, ctx->stream, ks, ds, slot_gid, n, G, reversemap, counts);
    }
    return aqg_check_launch(ctx, "dense_assign_kernel");
}

// Ranges of the key columns over n rows (synchronises).  Returns false when some column cannot take part (uint64 keys).
int aqg_key_ranges(aqg_ctx* ctx, const KeySpec& ks, uint32_t n, long long* mins, long long* maxs, bool* ok, uint32_t total) {
    if (total && (total < 4 * n || (n & 4095) || total < (1u << 22))) total = 0;      // (a spread sample needs room to spread: else the first n rows)
    if (total) for (int c = 0; c < ks.nkeys; ++c) if ((uintptr_t)ks.col[c] & 15) total = 0;
    *ok = false;
    for (int c = 0; c < ks.nkeys; ++c) if (ks.dt[c] == AQG_UINT64) return AQG_OK;
    long long* d = nullptr;
    AQG_TRY(aqg_ws_get(ctx, 2 * MAXKEYS, &d));
    hipLaunchKernelGGL(range_init_kernel, dim3(1), dim3(64), 0, ctx->stream, d, ks.nkeys);
    hipLaunchKernelGGL(key_range_kernel, dim3(aqg_grid(ctx, n / 4 + 1, 256, 4, 8)), dim3(256), 0, ctx->stream, ks, n, total, d);
    AQG_TRY(aqg_check_launch(ctx, "key_range_kernel"));
    long long h[2 * MAXKEYS];
    AQG_TRY(aqg_d2h(ctx, h, d, sizeof(long long) * 2 * ks.nkeys));
    for (int c = 0; c < ks.nkeys; ++c) { mins[c] = h[2 * c]; maxs[c] = h[2 * c + 1]; }
    *ok = true;
    return AQG_OK;
}

size_t aqg_dense_slot_bytes(const AccSpec& as, int need_count) { return 8 * (size_t)as.nacc + 4 + (need_count ? 4 : 0); }

// plan: fills ds and returns true when the domain fits DENSE_MAX_PASSES LDS tables
bool aqg_dense_plan(const KeySpec& ks, const long long* mins, const long long* maxs, const AccSpec& as, int need_count, DenseSpec* ds) {
    memset(ds, 0, sizeof *ds);
    const size_t sb = aqg_dense_slot_bytes(as, need_count);
    const uint64_t cap = (uint64_t)(DENSE_LDS_BYTES / sb) * DENSE_MAX_PASSES;
    uint64_t D = 1;
    for (int c = 0; c < ks.nkeys; ++c) {
        if (maxs[c] < mins[c]) return false;
        const unsigned long long range = (unsigned long long)maxs[c] - (unsigned long long)mins[c] + 1ull;
        if (range == 0 || range > cap) return false;
        ds->kmin[c] = mins[c];
        ds->range[c] = (uint32_t)range;
        ds->mult[c] = (uint32_t)D;
        D *= range;
        if (D > cap) return false;
    }
    ds->D = (uint32_t)D;
    const uint32_t max_pp = (uint32_t)(DENSE_LDS_BYTES / sb);
    ds->npass = (uint32_t)((D + max_pp - 1) / max_pp);
    ds->per_pass = (uint32_t)((D + ds->npass - 1) / ds->npass);
    return true;
}

int aqg_dense_aggregate(aqg_ctx* ctx, const KeySpec& ks, const DenseSpec& ds, const AccSpec& as, uint32_t n, int need_count, GTable gt) {
    const size_t lds = (size_t)ds.per_pass * aqg_dense_slot_bytes(as, need_count);
    const unsigned block = as.nacc <= 2 ? 1024 : 512;
    unsigned per_cu = (unsigned)((160 * 1024) / (lds + 1024));
    if (per_cu < 1) per_cu = 1;
    if (per_cu * block > 2048) per_cu = 2048 / block;
    const unsigned grid = (unsigned)ctx->num_cu * per_cu;
    auto launch = [&](auto kern) -> int {
        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, ctx->stream, ks, ds, as, gt, n, need_count);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "dense_agg_kernel");
    };
    // key columns all int32 / uint32, at most three of them, at most four accumulators: the register-resident instantiations
    int nk = ks.nkeys <= 3 && as.nacc <= 4 ? ks.nkeys : 0;
    for (int c = 0; c < ks.nkeys; ++c) if (ks.dt[c] != AQG_INT32 && ks.dt[c] != AQG_UINT32) nk = 0;
    static const bool nk_off = getenv("AQG_DENSE_GENERIC") != nullptr;
    if (nk_off) nk = 0;
    auto by_nk = [&](auto nacc_tag, auto block_tag) -> int {
        constexpr int N = decltype(nacc_tag)::value, B = decltype(block_tag)::value;
        switch (nk) {
        case 1: return launch(&dense_agg_kernel<N, B, 1>);
        case 2: return launch(&dense_agg_kernel<N, B, 2>);
        case 3: return launch(&dense_agg_kernel<N, B, 3>);
        default: return launch(&dense_agg_kernel<N, B, 0>);
        }
    };
    using B1024 = std::integral_constant<int, 1024>;
    using B512 = std::integral_constant<int, 512>;
    switch (as.nacc) {
    case 0: return by_nk(std::integral_constant<int, 0>{}, B1024{});          // (the first pass of aqg_groupby_build: keys only)
    case 1: return by_nk(std::integral_constant<int, 1>{}, B1024{});
    case 2: return by_nk(std::integral_constant<int, 2>{}, B1024{});
    case 3: return by_nk(std::integral_constant<int, 3>{}, B512{});
    case 4: return by_nk(std::integral_constant<int, 4>{}, B512{});
    case 5: return launch(&dense_agg_kernel<5, 512>);
    case 6: return launch(&dense_agg_kernel<6, 512>);
    case 7: return launch(&dense_agg_kernel<7, 512>);
    default: return launch(&dense_agg_kernel<8, 512>);
    }
}

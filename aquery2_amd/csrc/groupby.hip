// groupby.hip -- hash group-by and grouped aggregation.
//
// Replaces AQHashTable (reference server/hasher.h:146-199), set::hashtable_push
// (server/unordered_dense.h:1117-1147), ht_postproc (:181-198) and the generated per-group loop
// `out[g] = op(col[vecs[g]])` (engine/ast.py:722-789).
//
// Contract kept from the only executable path of the reference: group ids are dense and numbered by
// FIRST OCCURRENCE of the key tuple; row-id lists are DESCENDING inside a group.  The hash function
// is not observable in results (only dense ids are), so the device uses its own.
//
// Plans (run_agg picks by the group-count hint, key shape and row count)
//   agg32_kernel       (groupby_fast.hip) one or two 4-byte keys or one 8-byte key, up to four accumulators over 1- to 8-byte values,
//                      <= 3072 groups: LDS table, 8 probes in flight (h2o Q1/Q4)
//   agg_kernel<LDS>    any dtypes / MIN / MAX / VAR, packed tuples: {key, first_row, accumulators} open-addressing table in
//                      LDS (64 KB tables, several workgroups per CU; or one 150 KB table per CU and up to 4 passes over
//                      the rows), merged into the global table with device-scope atomics
//   dense.hip          small key DOMAIN (product of the column ranges): direct-indexed LDS tables (h2o Q2)
//   partition.hip      more groups than LDS holds: radix-partition the rows, aggregate each partition in LDS (h2o Q3/Q5/Q7)
//   agg_kernel<HBM>    rows straight to the global table (wide sparse tuples, the build path at high cardinality)
//   collect / rank / emit   occupied slots -> dense ids ordered by first row -> output columns.
//   assign_kernel     second pass for aqg_groupby_build: reversemap[i] = dense id, counts.
//   aqg_grouped_reduce: accumulators indexed by dense id (groups by the reversemap column).
//   postproc.hip      aqg_groupby_postproc: stable partition of row ids by group id.
// HBM roofline: agg = sum of key and value bytes per row (h2o Q1: 8 B/row); build = 12 B/row.
#include "groupby_dev.hpp"
#include "groupby_fast.hpp"
#include "dense.hpp"

// partition.hip
size_t aqg_partition_ws_bytes(uint32_t n, int ksz, const AccSpec& as, uint32_t pbits);
int aqg_partition_aggregate(aqg_ctx* ctx, const KeySpec& ks, const AccSpec& as, uint32_t n, uint32_t pbits, uint32_t lcap, int need_count, GTable out, uint32_t out_cap);
#include "partition1.hpp"

namespace {

// ---- the single-pass aggregation kernel -------------------------------------------------------
// K32: one 4-byte key column (h2o Q1/Q3/Q4/Q5).  LDS slot = {key32, first_row32} in one 8-byte
// word, so a hit costs one ds_read_b64 + one LDS atomic per accumulator.
// Tables that do not fit one 64 KB LDS table (up to ~25,000 groups: h2o Q2) use BLOCK = 1024, one workgroup per CU with a table
// of up to 150 KB (gfx950: 160 KB of LDS per workgroup), and `npass` passes over the rows: pass p aggregates only the keys
// whose pass hash equals p, so every pass's groups fit the table.  npass x (key + value bytes) of streaming reads beat the
// ~3e10/s scattered HBM atomics of the global table by an order of magnitude (Q2, 1e9 rows: 54 ms -> see DESIGN.md).
constexpr uint32_t SKIP = 0xFFFFFFFEu;          // row belongs to another pass
template <bool USE_LDS, bool K32, int NACC, int BLOCK = 256>
__global__ void __launch_bounds__(BLOCK) agg_kernel(KeySpec ks, AccSpec as, GTable gt, uint32_t n, uint32_t lcap, int need_count, uint32_t lrep, uint32_t npass) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS layout (USE_LDS): lkey u64[lcap+1] | lacc[a] u64[lcap+1] ... | lfirst u32[lcap+1] (wide keys) | lcount u32[lcap+1] | lused u32
    // `lrep` replicas of the table (lane l uses replica l % lrep) cut same-address / same-bank conflicts
    // of the LDS atomics when there are fewer groups than lanes
    const uint32_t LT = USE_LDS ? lrep * (lcap + 1) : 0;   // total LDS slots
    uint64_t* lkey = reinterpret_cast<uint64_t*>(smem_raw);
    uint64_t* lacc = lkey + LT;
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(lacc + (size_t)NACC * LT);
    uint32_t* lcount = lfirst + (K32 ? 0 : LT);
    uint32_t* lused = lcount + (need_count ? LT : 0);
    const uint32_t rbase = USE_LDS ? (threadIdx.x & (lrep - 1)) * (lcap + 1) : 0;
    const uint32_t llimit = lcap - (lcap >> 2);   // stop inserting at 75 % load; further new keys go to HBM
    // slot of a hash: multiply-shift, so the capacity need not be a power of two (it is sized to the LDS budget)
    // h1 is a Fibonacci hash: its top bits pick the pass, the remaining bits (h1 * npass drops exactly the pass bits) pick the
    // slot.  Consecutive integer keys -- dictionary ids, the usual group-by key -- land almost evenly spaced (three-distance
    // theorem), so they hardly ever collide; a second, random-looking mix here cost the build path 2.7x on h2o Q1 (100 dense keys).
    auto home = [&](uint32_t h1) -> uint32_t { return __umulhi(h1 * npass, lcap); };

  for (uint32_t pass = 0; pass < npass; ++pass) {
    if constexpr (USE_LDS) {
        if (pass) __syncthreads();
        for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {
            if constexpr (K32) lkey[s] = ((uint64_t)NOROW << 32) | EMPTY32; else { lkey[s] = EMPTY64; lfirst[s] = NOROW; }
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * LT + s] = acc_init(as.kind[a]);
            if (need_count) lcount[s] = 0;
        }
        if (threadIdx.x < lrep) lused[threadIdx.x] = 0;
        __syncthreads();
    }

    // returns the LDS slot of `key` (inserting it), or FAIL when the table is at its load limit
    auto lds_slot = [&](uint64_t key, uint32_t s) -> uint32_t {   // s: home slot
        if constexpr (K32) {
            uint32_t k = (uint32_t)key;
            if (k == EMPTY32) return rbase + lcap;
            uint32_t* kw = reinterpret_cast<uint32_t*>(lkey + rbase);
            uint32_t* used = lused + (threadIdx.x & (lrep - 1));
            for (uint32_t p = 0; p < lcap; ++p) {
                uint32_t cur = kw[2 * s];
                if (cur == k) return rbase + s;
                if (cur == EMPTY32) {
                    if (*used >= llimit) return FAIL;
                    uint32_t old = atomicCAS(&kw[2 * s], EMPTY32, k);
                    if (old == EMPTY32) { atomicAdd(used, 1u); return rbase + s; }
                    if (old == k) return rbase + s;
                }
                s = s + 1 == lcap ? 0 : s + 1;
            }
            return FAIL;
        } else {
            if (key == EMPTY64) return rbase + lcap;
            uint64_t* kw = lkey + rbase;
            uint32_t* used = lused + (threadIdx.x & (lrep - 1));
            for (uint32_t p = 0; p < lcap; ++p) {
                uint64_t cur = kw[s];
                if (cur == key) return rbase + s;
                if (cur == EMPTY64) {
                    if (*used >= llimit) return FAIL;
                    unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&kw[s]), EMPTY64, key);
                    if (old == EMPTY64) { atomicAdd(used, 1u); return rbase + s; }
                    if (old == key) return rbase + s;
                }
                s = s + 1 == lcap ? 0 : s + 1;
            }
            return FAIL;
        }
    };
    auto lds_touch_first = [&](uint32_t s, uint32_t row) {
        uint32_t* f = K32 ? reinterpret_cast<uint32_t*>(lkey) + 2 * s + 1 : lfirst + s;
        if (row < *f) atomicMin(f, row);
    };

    // one row whose slot is known
    auto to_global = [&](uint64_t key, uint32_t row, const uint64_t* vals) {
        uint32_t g = ks.wide ? gt_find_or_insert_wide(gt, ks, row) : gt_find_or_insert(gt, key);
        if (g == FAIL) return;
        gt_touch_first(gt, g, row);
        if (need_count) atomicAdd(gt.count_p(g), 1u);
        _Pragma("unroll") for (int a = 0; a < NACC; ++a) acc_apply(gt.acc_p(a, g), as.kind[a], vals[a]);
    };

    const uint32_t nchunk = n >> 2;   // 4 consecutive rows per lane per step
    const bool vec_ok = K32 && ks.nkeys == 1;
    uint32_t c_lo, c_hi;
    wg_span(nchunk, c_lo, c_hi);
    for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const size_t base = (size_t)c * 4;
        uint64_t key[4];
        if (vec_ok) {
            pack<uint32_t, 4> kv = *reinterpret_cast<const pack<uint32_t, 4>*>(static_cast<const uint32_t*>(ks.col[0]) + base);
#pragma unroll
            for (int j = 0; j < 4; ++j) key[j] = kv.v[j];
        } else if (!ks.wide) {
            pack_key4(ks, base, key);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) key[j] = base + j;
        }
        uint64_t vals[NACC ? NACC : 1][4];
        _Pragma("unroll") for (int a = 0; a < NACC; ++a) val_operand4(as.dt[a], as.col[a], base, as.kind[a], as.square[a], as.part[a], vals[a]);
        uint32_t slot[4];
        if constexpr (USE_LDS) {
            // speculative first probe of all four rows at once: one LDS round trip in the common (hit) case
            uint64_t w[4];
            uint32_t hs[4];                     // home slot (without the replica base)
#pragma unroll
            for (int j = 0; j < 4; ++j) {     // all four probes are issued unconditionally (a branch in front of an LDS read serialises them)
                const uint32_t h1 = lds_h1<K32>(key[j]);
                hs[j] = home(h1);
                w[j] = lkey[rbase + hs[j]];
                slot[j] = npass > 1 && __umulhi(h1, npass) != pass ? SKIP : rbase + hs[j];
            }
            if constexpr (K32) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t row = (uint32_t)base + j;
                    if (slot[j] == SKIP) continue;
                    if ((uint32_t)w[j] == (uint32_t)key[j] && (uint32_t)key[j] != EMPTY32) {
                        if (row < (uint32_t)(w[j] >> 32)) atomicMin(reinterpret_cast<uint32_t*>(lkey) + 2 * slot[j] + 1, row);
                    } else {
                        slot[j] = lds_slot(key[j], hs[j]);
                        if (slot[j] != FAIL) lds_touch_first(slot[j], row);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (slot[j] == SKIP) continue;
                    if (!(w[j] == key[j] && key[j] != EMPTY64)) slot[j] = lds_slot(key[j], hs[j]);
                    if (slot[j] != FAIL) lds_touch_first(slot[j], (uint32_t)base + j);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) slot[j] = FAIL;
        }
        if (need_count) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (slot[j] < SKIP) atomicAdd(&lcount[slot[j]], 1u);
        }
        _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
            uint64_t* la = lacc + (size_t)a * LT;
            switch (as.kind[a]) {   // wave-uniform: one branch per accumulator per four rows
            case ACC_ADD_I:
#pragma unroll
                for (int j = 0; j < 4; ++j) if (slot[j] < SKIP) atomicAdd(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)vals[a][j]);
                break;
            case ACC_ADD_F:
#pragma unroll
                for (int j = 0; j < 4; ++j) if (slot[j] < SKIP) atomicAdd(reinterpret_cast<double*>(&la[slot[j]]), __builtin_bit_cast(double, vals[a][j]));
                break;
            case ACC_MIN:
#pragma unroll
                for (int j = 0; j < 4; ++j) if (slot[j] < SKIP) atomicMin(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)vals[a][j]);
                break;
            default:
#pragma unroll
                for (int j = 0; j < 4; ++j) if (slot[j] < SKIP) atomicMax(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)vals[a][j]);
                break;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (slot[j] == FAIL) {   // LDS table full (or HBM mode): straight to the global table
                uint64_t v1[NACC ? NACC : 1];
                _Pragma("unroll") for (int a = 0; a < NACC; ++a) v1[a] = vals[a][j];
                to_global(key[j], (uint32_t)base + j, v1);
            }
        }
    }
    // tail rows (< 4) by the first lanes of block 0
    if (blockIdx.x == 0 && pass == 0) {
        uint32_t row = (nchunk << 2) + threadIdx.x;
        if (row < n) {
            uint64_t k = ks.wide ? (uint64_t)row : pack_key(ks, row);
            uint64_t v1[NACC ? NACC : 1];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) v1[a] = val_operand(as.dt[a], as.col[a], row, as.kind[a], as.square[a], as.part[a]);
            to_global(k, row, v1);
        }
    }

    if constexpr (USE_LDS) {
        __syncthreads();
        // merge this workgroup's table into the global one
        for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {
            uint64_t key; uint32_t first;
            if constexpr (K32) {
                uint64_t w = lkey[s];
                first = (uint32_t)(w >> 32);
                key = (uint32_t)w;
            } else { key = lkey[s]; first = lfirst[s]; }
            if (first == NOROW) continue;          // never touched (covers the sentinel slot too)
            uint32_t g = gt_find_or_insert(gt, key);
            if (g == FAIL) continue;
            atomicMin(gt.first_p(g), first);
            if (need_count) atomicAdd(gt.count_p(g), lcount[s]);
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) acc_apply(gt.acc_p(a, g), as.kind[a], lacc[(size_t)a * LT + s]);
        }
    }
  }   // passes
}

// (the fast LDS kernel for 4- / 8-byte keys and values, agg32_kernel, lives in groupby_fast.hip: its 36 instantiations compile
// beside this file instead of after it)

// ---- fused star join + group-by sum (BASELINE config 4: fact JOIN small(key, w) ON fk, sum(val * w) BY gkey) -----------------
// 12 B/row of HBM traffic (fk, gkey, val) instead of the 44 B/row of the composed lookup -> gather -> multiply -> group-by:
// the dimension side {key -> w} is an LDS open-addressing table built by every workgroup from the (small) dimension columns,
// the group table is the K32 LDS table of agg_kernel ({key, first_row} in one 8-byte word), and the exact 64-bit product is
// accumulated as two 64-bit sums of its 32-bit halves (no overflow for n < 2^32).  Fact rows without a partner are dropped
// (inner join); of duplicate dimension keys the lowest row wins (aqg_join_lookup's contract).
struct StarJoin {
    const uint32_t* dim_keys; const uint32_t* dim_vals; uint32_t nb; uint32_t dcap;   // dcap: power of two >= 2 * nb
    const uint32_t* fk; const uint32_t* vals; int val_signed; int dim_signed;
};

__global__ void __launch_bounds__(256) starjoin_kernel(const uint32_t* __restrict__ gkeys, StarJoin sj, GTable gt, uint32_t n, uint32_t lcap) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t LT = lcap + 1;
    uint64_t* lkey = reinterpret_cast<uint64_t*>(smem_raw);          // [LT] {first_row << 32 | key}
    uint64_t* lacc = lkey + LT;                                      // [2][LT] sums of the low / high halves of the products
    uint32_t* dkey = reinterpret_cast<uint32_t*>(lacc + 2 * (size_t)LT);   // [dcap]
    uint32_t* dval = dkey + sj.dcap;                                 // [dcap] row while building, then w
    __shared__ uint32_t lused, dsent;                                // dsent: row / w of the dimension key equal to EMPTY32
    const uint32_t lmask = lcap - 1, llimit = lcap - (lcap >> 2), dmask = sj.dcap - 1, lbits = 31 - __clz(lcap), dbits = 31 - __clz(sj.dcap);
    for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) { lkey[s] = ((uint64_t)NOROW << 32) | EMPTY32; lacc[s] = 0; lacc[LT + s] = 0; }
    for (uint32_t s = threadIdx.x; s < sj.dcap; s += blockDim.x) { dkey[s] = EMPTY32; dval[s] = NOROW; }
    if (threadIdx.x == 0) { lused = 0; dsent = NOROW; }
    __syncthreads();
    for (uint32_t r = threadIdx.x; r < sj.nb; r += blockDim.x) {     // dimension table: key -> lowest row
        const uint32_t k = sj.dim_keys[r];
        if (k == EMPTY32) { atomicMin(&dsent, r); continue; }
        uint32_t s = fib_slot(k, dbits);
        while (true) {
            uint32_t cur = dkey[s];
            if (cur == EMPTY32) { uint32_t old = atomicCAS(&dkey[s], EMPTY32, k); cur = old == EMPTY32 ? k : old; }
            if (cur == k) { atomicMin(&dval[s], r); break; }
            s = (s + 1) & dmask;
        }
    }
    __syncthreads();
    __shared__ uint32_t wmax;                                        // largest |w| of the dimension side
    if (threadIdx.x == 0) wmax = 0;
    __syncthreads();
    {
        uint32_t m = 0;
        for (uint32_t r = threadIdx.x; r < sj.nb; r += blockDim.x) {
            const uint32_t wb = sj.dim_vals[r];
            const uint32_t a = sj.dim_signed ? (uint32_t)((int32_t)wb < 0 ? 0u - wb : wb) : wb;
            m = a > m ? a : m;
        }
        m = wave_reduce(m, OpMax{});
        if (lane_id() == 0) atomicMax(&wmax, m);
    }
    for (uint32_t s = threadIdx.x; s < sj.dcap; s += blockDim.x) if (dval[s] != NOROW) dval[s] = sj.dim_vals[dval[s]];
    const bool has_sent = dsent != NOROW;
    const uint32_t sent_w = has_sent ? sj.dim_vals[dsent] : 0;
    __syncthreads();

    auto group_slot = [&](uint32_t k) -> uint32_t {                  // insert path of the group table
        if (k == EMPTY32) return lcap;
        uint32_t* kw = reinterpret_cast<uint32_t*>(lkey);
        uint32_t s = fib_slot(k, lbits);
        for (uint32_t p = 0; p <= lmask; ++p) {
            uint32_t cur = kw[2 * s];
            if (cur == k) return s;
            if (cur == EMPTY32) {
                if (lused >= llimit) return FAIL;
                uint32_t old = atomicCAS(&kw[2 * s], EMPTY32, k);
                if (old == EMPTY32) { atomicAdd(&lused, 1u); return s; }
                if (old == k) return s;
            }
            s = (s + 1) & lmask;
        }
        return FAIL;
    };
    const bool any_signed = sj.val_signed || sj.dim_signed;
    auto product = [&](uint32_t vbits, uint32_t wbits) -> uint64_t {   // exact 64-bit product (bits)
        const int64_t v = sj.val_signed ? (int64_t)(int32_t)vbits : (int64_t)vbits;
        const int64_t w = sj.dim_signed ? (int64_t)(int32_t)wbits : (int64_t)wbits;
        return (uint64_t)v * (uint64_t)w;
    };
    auto lo_half = [&](uint64_t p) -> unsigned long long { return p & 0xFFFFFFFFull; };
    auto hi_half = [&](uint64_t p) -> unsigned long long { return any_signed ? (unsigned long long)((int64_t)p >> 32) : p >> 32; };
    auto to_global = [&](uint32_t k, uint32_t row, uint64_t p) {     // LDS table at its load limit, or tail rows
        uint32_t g = gt_find_or_insert(gt, (uint64_t)k);
        if (g == FAIL) return;
        gt_touch_first(gt, g, row);
        atomicAdd(reinterpret_cast<unsigned long long*>(gt.acc_p(0, g)), lo_half(p));
        atomicAdd(reinterpret_cast<unsigned long long*>(gt.acc_p(1, g)), hi_half(p));
    };
    auto dim_lookup = [&](uint32_t k, uint32_t first_probe, uint32_t s, uint32_t& w) -> bool {   // first_probe = dkey[s]
        if (k == EMPTY32) { w = sent_w; return has_sent; }
        uint32_t cur = first_probe;
        for (uint32_t p = 0; p <= dmask; ++p) {
            if (cur == k) { w = dval[s]; return true; }
            if (cur == EMPTY32) return false;
            s = (s + 1) & dmask;
            cur = dkey[s];
        }
        return false;
    };

    constexpr int R = 8;                                             // rows per lane per step: two 16-byte loads per column
    const uint32_t nchunk = n / R;
    uint32_t c_lo, c_hi;
    wg_span(nchunk, c_lo, c_hi);
    // |product| < 2^32 * wmax; when this workgroup's rows cannot overflow 63 bits of that, ONE 64-bit LDS atomic per row carries
    // the whole product (split into its halves at the merge); otherwise the halves are summed separately
    const bool one_acc = (uint64_t)wmax * ((uint64_t)(c_hi - c_lo) * R + R) < (1ull << 31);
    for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const size_t base = (size_t)c * R;
        uint32_t f[R], g[R], v[R];
#pragma unroll
        for (int h = 0; h < R / 4; ++h) {
            const pack<uint32_t, 4> f4 = *reinterpret_cast<const pack<uint32_t, 4>*>(sj.fk + base + 4 * h);
            const pack<uint32_t, 4> g4 = *reinterpret_cast<const pack<uint32_t, 4>*>(gkeys + base + 4 * h);
            const pack<uint32_t, 4> v4 = *reinterpret_cast<const pack<uint32_t, 4>*>(sj.vals + base + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) { f[4 * h + j] = f4.v[j]; g[4 * h + j] = g4.v[j]; v[4 * h + j] = v4.v[j]; }
        }
        uint32_t ds[R], dk[R], gs[R];
        uint64_t gw[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {                                // 2 R LDS probes in flight
            ds[j] = fib_slot(f[j], dbits); dk[j] = dkey[ds[j]];
            gs[j] = fib_slot(g[j], lbits); gw[j] = lkey[gs[j]];
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            uint32_t w;
            if (!dim_lookup(f[j], dk[j], ds[j], w)) continue;        // no partner: the row is not in the join
            const uint64_t p = product(v[j], w);
            const uint32_t row = (uint32_t)base + j, k = g[j];
            uint32_t s = gs[j];
            if ((uint32_t)gw[j] == k && k != EMPTY32) {
                if (row < (uint32_t)(gw[j] >> 32)) atomicMin(reinterpret_cast<uint32_t*>(lkey) + 2 * s + 1, row);
            } else {
                s = group_slot(k);
                if (s == FAIL) { to_global(k, row, p); continue; }
                uint32_t* fr = reinterpret_cast<uint32_t*>(lkey) + 2 * s + 1;
                if (row < *fr) atomicMin(fr, row);
            }
            if (one_acc) atomicAdd(reinterpret_cast<unsigned long long*>(&lacc[s]), (unsigned long long)p);
            else {
                atomicAdd(reinterpret_cast<unsigned long long*>(&lacc[s]), lo_half(p));
                atomicAdd(reinterpret_cast<unsigned long long*>(&lacc[LT + s]), hi_half(p));
            }
        }
    }
    if (blockIdx.x == 0) {                                           // tail rows (< R)
        const uint32_t row = nchunk * R + threadIdx.x;
        if (row < n) {
            const uint32_t k = sj.fk[row], s0 = fib_slot(k, dbits);
            uint32_t w;
            if (dim_lookup(k, dkey[s0], s0, w)) to_global(gkeys[row], row, product(sj.vals[row], w));
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {        // merge into the global table
        const uint64_t wd = lkey[s];
        const uint32_t first = (uint32_t)(wd >> 32);
        if (first == NOROW) continue;
        uint32_t g = gt_find_or_insert(gt, (uint64_t)(uint32_t)wd);
        if (g == FAIL) continue;
        atomicMin(gt.first_p(g), first);
        if (one_acc) {                                              // lacc[s] is the exact (signed or unsigned) 64-bit sum of this workgroup
            atomicAdd(reinterpret_cast<unsigned long long*>(gt.acc_p(0, g)), lo_half(lacc[s]));
            atomicAdd(reinterpret_cast<unsigned long long*>(gt.acc_p(1, g)), hi_half(lacc[s]));
        } else {
            atomicAdd(reinterpret_cast<unsigned long long*>(gt.acc_p(0, g)), (unsigned long long)lacc[s]);
            atomicAdd(reinterpret_cast<unsigned long long*>(gt.acc_p(1, g)), (unsigned long long)lacc[LT + s]);
        }
    }
}

// first row of every group, after the fact: tiles are scanned in order by a small grid; once every group has a candidate,
// a workgroup stops as soon as its next tile starts beyond the largest candidate (no later row can lower any of them).
__global__ void __launch_bounds__(256) first_rows_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ keys_hi, int key8, uint32_t t0 /* first tile */, uint32_t n /* rows end */, GTable gt, const uint32_t* __restrict__ occ) {
    __shared__ uint32_t red[4];
    __shared__ uint32_t stop, all_seen;
    const uint32_t G = gt.flags[1];
    constexpr int FR = 4;                      // rows per lane and tile: the first round covers gridDim x 1024 rows (16 rows: 28 us on h2o Q1)
    constexpr uint32_t TILE = 256 * FR;
    for (uint32_t t = t0 + blockIdx.x; (uint64_t)t * TILE < n; t += gridDim.x) {
        const uint32_t tbase = t * TILE;
        // ONE lane samples the counter other workgroups keep incrementing: the branch below holds barriers, so every wavefront of
        // the workgroup must take the same side of it (a per-lane load could split them: divergent barrier, stale `red` / `stop`)
        if (threadIdx.x == 0) { stop = 0; all_seen = __hip_atomic_load(&gt.flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= G; }
        if (threadIdx.x < 4) red[threadIdx.x] = 0;
        __syncthreads();
        if (all_seen) {
            uint32_t m = 0;
            for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) {
                uint32_t f = __hip_atomic_load(gt.first_p(occ[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m = f > m ? f : m;
            }
            m = wave_reduce(m, OpMax{});
            if (lane_id() == 0) red[wave_id()] = m;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t bound = red[0] > red[1] ? red[0] : red[1];
                bound = bound > red[2] ? bound : red[2];
                bound = bound > red[3] ? bound : red[3];
                stop = tbase > bound;
            }
            __syncthreads();
        }
        if (stop) break;
        for (int r = 0; r < FR; ++r) {
            uint32_t row = tbase + r * 256 + threadIdx.x;
            if (row < n) {
                uint32_t s = gt_find(gt, key8 ? reinterpret_cast<const uint64_t*>(keys)[row] : keys_hi ? ((uint64_t)keys[row] | ((uint64_t)keys_hi[row] << 32)) : (uint64_t)keys[row]);
                if (s != FAIL && row < *gt.first_p(s)) {
                    uint32_t old = atomicMin(gt.first_p(s), row);
                    if (old >= OCCUPIED) atomicAdd(&gt.flags[2], 1u);
                }
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) occ_iota_kernel(uint32_t* __restrict__ occ, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) occ[i] = i;
}
__global__ void __launch_bounds__(256) gt_init_kernel(GTable gt, AccSpec as) {
    if (blockIdx.x == 0 && threadIdx.x < 64) gt.flags[threadIdx.x] = 0;          // the 64 flag words, too (one launch instead of a fill behind it)
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s <= gt.cap; s += gridDim.x * blockDim.x) {
        *gt.key_p(s) = EMPTY64;
        *gt.first_p(s) = NOROW;
        *gt.count_p(s) = 0;
        for (int a = 0; a < as.nacc; ++a) *gt.acc_p(a, s) = acc_init(as.kind[a]);
    }
}

// ---- dense ids in first-occurrence order --------------------------------------------------------
// occupied slots -> occ[] (any order).  One returning atomic per workgroup and step, not per wavefront: with 1e8 occupied slots the
// single counter word was the whole cost (47 ms; the word saturates near 9e7 atomics per second).
__global__ void __launch_bounds__(256) collect_kernel(GTable gt, uint32_t* __restrict__ occ) {
    __shared__ uint32_t wcount[4];
    __shared__ uint32_t base;
    const uint64_t total = (uint64_t)gt.cap + 1;
    for (uint64_t s0 = (uint64_t)blockIdx.x * blockDim.x; s0 < total; s0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t s = s0 + threadIdx.x;
        const bool used = s < total && (*gt.first_p((uint32_t)s)) != NOROW;
        const uint64_t bal = __ballot(used);
        if (lane_id() == 0) wcount[wave_id()] = (uint32_t)__popcll(bal);
        __syncthreads();
        if (threadIdx.x == 0) { const uint32_t c = wcount[0] + wcount[1] + wcount[2] + wcount[3]; base = c ? atomicAdd(&gt.flags[1], c) : 0; }
        __syncthreads();
        if (used) {
            uint32_t off = base + (uint32_t)__popcll(bal & ((1ull << lane_id()) - 1ull));
            for (int w = 0; w < wave_id(); ++w) off += wcount[w];
            occ[off] = (uint32_t)s;
        }
        __syncthreads();
    }
}
// G <= 4096: rank by counting inside one workgroup
__global__ void __launch_bounds__(1024) rank_small_kernel(GTable gt, const uint32_t* __restrict__ occ, uint32_t* __restrict__ gid_of_occ,
                                                          uint32_t* __restrict__ slot_gid) {
    __shared__ uint32_t f[4096];
    uint32_t G = gt.flags[1];
    if (G > 4096) return;
    for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) f[i] = (*gt.first_p(occ[i]));
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) {
        uint32_t mine = f[i], r = 0;
        for (uint32_t j = 0; j < G; ++j) r += f[j] < mine;
        gid_of_occ[i] = r;
        slot_gid[occ[i]] = r;
    }
}
// any G: mark first rows in a bitmap over the n rows, prefix-count it, look the rank up
// tile_mark (sparse ranking: few groups over many rows): the tiles of 1024 words that hold a bit at all -- the others are neither
// read nor given prefixes, and the bitmap itself is not cleared with a fill but bit by bit behind the ranking (bitmap_clear_kernel):
// 1e9 rows / 1e4 groups (h2o Q2) spent 0.13 ms filling and scanning 125 MB of zeros
__global__ void __launch_bounds__(256) bitmap_set_kernel(GTable gt, const uint32_t* __restrict__ occ, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ tile_mark) {
    uint32_t G = gt.flags[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += gridDim.x * blockDim.x) {
        uint32_t r = (*gt.first_p(occ[i]));
        atomicOr(&bitmap[r >> 5], 1u << (r & 31));
        if (tile_mark) tile_mark[r >> 15] = 1u;
    }
}
__global__ void __launch_bounds__(256) bitmap_clear_kernel(GTable gt, const uint32_t* __restrict__ occ, uint32_t* __restrict__ bitmap) {
    uint32_t G = gt.flags[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += gridDim.x * blockDim.x) bitmap[(*gt.first_p(occ[i])) >> 5] = 0u;
}
// tile = 1024 words (one per thread... 256 threads x 4 words): per-word exclusive prefix inside the tile + tile total
__global__ void __launch_bounds__(256) bitmap_tile_kernel(const uint32_t* __restrict__ bitmap, uint32_t nwords,
                                                          uint32_t* __restrict__ word_prefix, uint32_t* __restrict__ tile_total, const uint32_t* __restrict__ tile_mark) {
    __shared__ uint32_t wsum[4];
    uint32_t tile = blockIdx.x;
    if (tile_mark && !tile_mark[tile]) { if (threadIdx.x == 0) tile_total[tile] = 0; return; }      // (uniform over the workgroup)
    uint32_t w0 = tile * 1024 + threadIdx.x * 4;
    uint32_t c[4], tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { uint32_t w = w0 + j; c[j] = w < nwords ? __popc(bitmap[w]) : 0; tot += c[j]; }
    uint32_t incl = wave_scan_incl(tot, OpAdd{}, lane_id());
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (int w = 0; w < wave_id(); ++w) wbase += wsum[w];
    uint32_t excl = wbase + incl - tot;
#pragma unroll
    for (int j = 0; j < 4; ++j) { uint32_t w = w0 + j; if (w < nwords) word_prefix[w] = excl; excl += c[j]; }
    if (threadIdx.x == 255) tile_total[tile] = wbase + incl;
}
__global__ void __launch_bounds__(1024) tile_scan_kernel(uint32_t* __restrict__ tile_total, uint32_t ntiles) {
    // single workgroup exclusive scan, in place
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < ntiles ? tile_total[i] : 0;
        uint32_t incl = wave_scan_incl(v, OpAdd{}, lane_id());
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t wbase = carry;
        for (int w = 0; w < wave_id(); ++w) wbase += wsum[w];
        if (i < ntiles) tile_total[i] = wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = wbase + incl;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) rank_bitmap_kernel(GTable gt, const uint32_t* __restrict__ occ, const uint32_t* __restrict__ bitmap,
                                                          const uint32_t* __restrict__ word_prefix, const uint32_t* __restrict__ tile_prefix,
                                                          uint32_t* __restrict__ gid_of_occ, uint32_t* __restrict__ slot_gid) {
    uint32_t G = gt.flags[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += gridDim.x * blockDim.x) {
        uint32_t r = (*gt.first_p(occ[i])), w = r >> 5;
        uint32_t rank = tile_prefix[w >> 10] + word_prefix[w] + __popc(bitmap[w] & ((1u << (r & 31)) - 1u));
        gid_of_occ[i] = rank;
        slot_gid[occ[i]] = rank;
    }
}

// ---- 128-bit helpers for the emit epilogue ------------------------------------------------------
__device__ inline aqg_i128 mul_i64(int64_t a, int64_t b) {   // exact signed 64x64 -> 128
    aqg_i128 r;
    r.lo = (uint64_t)a * (uint64_t)b;
    r.hi = (uint64_t)__mul64hi(a, b);
    return r;
}
__device__ inline aqg_i128 mul_128(aqg_i128 a, aqg_i128 b) {   // low 128 bits of the product (two's complement: sign-agnostic)
    aqg_i128 r;
    r.lo = a.lo * b.lo;
    r.hi = __umul64hi(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
    return r;
}
__device__ inline aqg_i128 mul_u64(uint64_t a, uint64_t b) {
    aqg_i128 r;
    r.lo = a * b;
    r.hi = __umul64hi(a, b);
    return r;
}

// what each requested aggregate reads from the accumulators
struct AggOut { int op; int dt; int acc0; int acc1; int acc2; int acc3; void* out; };   // wide (8-byte integer) sums: acc0/acc2 = low, acc1/acc3 = high halves
struct EmitSpec { int nagg; AggOut agg[MAXAGG]; int nkeys; int key_dt[MAXKEYS]; int key_shift[MAXKEYS]; void* key_out[MAXKEYS]; int wide; const void* key_col[MAXKEYS];
                  uint32_t* first_out; uint32_t* count_out; };

// (not inlined: a size / dtype switch whose arms STORE, inlined into a loop with a 64-bit value live across it, is the shape hipcc
// 7.2 miscompiled in unpack_kernel -- profiles/r2_hipcc_switch_miscompile.md)
template <class T> __device__ __noinline__ void store_minmax(void* out, uint32_t g, uint64_t mapped, bool is_max) {
    T v;
    if constexpr (std::is_floating_point_v<T>) {
        double d = unmap_f(mapped);
        v = (T)d;
        if (is_max) { T seed = dlimits<T>::min(); v = seed > v ? seed : v; }   // max seeds with numeric_limits<T>::min() (D8)
    } else if constexpr (std::is_unsigned_v<T>) v = (T)mapped;
    else v = (T)unmap_i(mapped);
    static_cast<T*>(out)[g] = v;
}

// `order` (large G only): order[g] = the occ index of dense id g, so that the lanes walk the OUTPUT columns (and, for wide tuples,
// the key columns at the groups' first rows) in ascending order instead of scattering eight columns at random
__global__ void __launch_bounds__(256) emit_order_kernel(const uint32_t* __restrict__ gid_of_occ, uint32_t G, uint32_t* __restrict__ order) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G; i += gridDim.x * blockDim.x) order[gid_of_occ[i]] = i;
}
// one group: record `s` of `gt` -> row `g` of every output column.  `key`: the packed key, or (wide tuples) the group's first row
// `R` = where the record comes from: first(), count(), acc(a).  TableRecord: slot `s` of a group table; RowRecord (below): ONE ROW of
// the input taken as a whole group (every row its own group).
struct TableRecord {
    static constexpr bool inline_stores = false;
    const GTable& gt; uint32_t s;
    __device__ inline uint32_t first() const { return *gt.first_p(s); }
    __device__ inline uint32_t count() const { return gt.has_count ? *gt.count_p(s) : 0; }
    __device__ inline uint64_t acc(int a) const { return *gt.acc_p(a, s); }
};
template <bool KEYS, class R>
__device__ inline void emit_record_from(const R& rec, uint32_t g, const EmitSpec& es, uint64_t key) {
    if constexpr (KEYS) for (int k = 0; k < es.nkeys; ++k) {
        uint64_t bits = es.wide ? load_bits(es.key_dt[k], es.key_col[k], (size_t)(uint32_t)key) : key >> es.key_shift[k];
        store_sized(es.key_out[k], g, aqg_dtype_size_dev(es.key_dt[k]), bits);
    }
    // (table records store through calls -- see store_at; the streaming row map keeps its stores inline: 5.9 against 7.3 ms per 1e9 rows,
    // and tests/test_gpu_plans.py::test_every_row_its_own_group... checks every arm and value type of that kernel)
    auto put = [&](void* col, size_t i, auto v) { using T = decltype(v); if constexpr (R::inline_stores) static_cast<T*>(col)[i] = v; else store_at<T>(col, i, v); };
    es.first_out[g] = rec.first();
    uint32_t cnt = rec.count();
    if (es.count_out) es.count_out[g] = cnt;
    for (int j = 0; j < es.nagg; ++j) {
        const AggOut& a = es.agg[j];
        int vc = vclass(a.dt);
        uint64_t v0 = a.acc0 >= 0 ? rec.acc(a.acc0) : 0;
        const bool wide = a.dt == AQG_INT64 || a.dt == AQG_UINT64;
        // exact 128-bit sum (and sum of squares) of an integer column
        auto sum128 = [&](int lo_acc, int hi_acc) -> aqg_i128 {
            uint64_t lo = rec.acc(lo_acc);
            if (!wide) return vc == VC_U ? i128_from_u64(lo) : i128_from_i64((int64_t)lo);
            uint64_t hi = rec.acc(hi_acc);                       // sum of the high halves, to be shifted by 32
            aqg_i128 h = vc == VC_U ? i128_from_u64(hi) : i128_from_i64((int64_t)hi);
            aqg_i128 sh = {h.lo << 32, (h.hi << 32) | (h.lo >> 32)};
            return i128_add(sh, i128_from_u64(lo));
        };
        auto to_double = [&](aqg_i128 v) -> double { return vc == VC_U ? u128_to_double(v.hi, v.lo) : i128_to_double(v); };
        switch (a.op) {
        case AQG_RED_SUM: case AQG_RED_SUMSQ:                           // -> GetLongType
            if (vc == VC_F) put(a.out, g, __builtin_bit_cast(double, v0));
            else put(a.out, g, sum128(a.acc0, a.acc1));
            break;
        case AQG_RED_COUNT: put(a.out, g, (uint64_t)cnt); break;
        case AQG_RED_AVG: {                                             // sum / (double)size
            double sd = vc == VC_F ? __builtin_bit_cast(double, v0) : to_double(sum128(a.acc0, a.acc1));
            put(a.out, g, sd / (double)cnt);
        } break;
        case AQG_RED_VAR: case AQG_RED_STDDEV: {                        // (ssq - s*s/(double)(n+1)) / (double)(n+1)
            double np1 = (double)(uint32_t)(cnt + 1), d;
            if (vc == VC_F) {
                double sd = __builtin_bit_cast(double, v0), q = __builtin_bit_cast(double, rec.acc(a.acc2));
                d = (q - sd * sd / np1) / np1;
            } else {
                aqg_i128 sm = sum128(a.acc0, a.acc1), q = sum128(a.acc2, a.acc3);
                aqg_i128 ss = mul_128(sm, sm);                          // s * s in the 128-bit LongType (wraps like the reference)
                d = (to_double(q) - to_double(ss) / np1) / np1;
            }
            put(a.out, g, a.op == AQG_RED_STDDEV ? sqrt(d) : d);
        } break;
        case AQG_RED_MIN: case AQG_RED_MAX: {
            bool mx = a.op == AQG_RED_MAX;
            switch (a.dt) {
            case AQG_INT8: store_minmax<int8_t>(a.out, g, v0, mx); break;
            case AQG_INT16: store_minmax<int16_t>(a.out, g, v0, mx); break;
            case AQG_INT32: store_minmax<int32_t>(a.out, g, v0, mx); break;
            case AQG_INT64: store_minmax<int64_t>(a.out, g, v0, mx); break;
            case AQG_UINT8: store_minmax<uint8_t>(a.out, g, v0, mx); break;
            case AQG_UINT16: store_minmax<uint16_t>(a.out, g, v0, mx); break;
            case AQG_UINT32: store_minmax<uint32_t>(a.out, g, v0, mx); break;
            case AQG_UINT64: store_minmax<uint64_t>(a.out, g, v0, mx); break;
            case AQG_FLOAT: store_minmax<float>(a.out, g, v0, mx); break;
            default: store_minmax<double>(a.out, g, v0, mx); break;
            }
        } break;
        }
    }
}
template <bool KEYS = true>
__device__ inline void emit_record(const GTable& gt, uint32_t s, uint32_t g, const EmitSpec& es, uint64_t key) { emit_record_from<KEYS>(TableRecord{gt, s}, g, es, key); }

// Every row its own group (G == n: a grouping by a unique key, h2o Q10 at 1e9 rows): the groups in first-occurrence order ARE the rows in
// row order, so the result columns are a map of the input columns -- no ranking, no ordering of a billion records.  The record of
// group i is made from row i on the fly: the accumulator a table would hold after that one row (acc_init folded with the row's operand).
struct RowRecord {
    static constexpr bool inline_stores = true;
    const AccSpec& as; uint32_t i;
    __device__ inline uint32_t first() const { return i; }
    __device__ inline uint32_t count() const { return 1u; }
    __device__ inline uint64_t acc(int a) const {
        const uint64_t v = val_operand(as.dt[a], as.col[a], i, as.kind[a], as.square[a], as.part[a]);
        if (as.kind[a] == ACC_ADD_F) return __builtin_bit_cast(uint64_t, 0.0 + __builtin_bit_cast(double, v));     // (what the atomic add onto +0.0 leaves: -0.0 becomes +0.0)
        return v;                                                                                               // 0 + v; min(~0, v); max(0, v)
    }
};
// The BUILD over a dense 4-byte key domain of up to 2^21 values (8 MB: what the L2s hold of it, the Infinity Cache the rest): the id of every
// row comes from a look-up table key -> group id filled from the group table, read in ROW order -- instead of probing the partitioned rows
// and routing {row, id} pairs back by row (2.3 + 12 ms per 1e9 rows).  1e9 random 4-byte gathers cost 6.3 ms out of a 4 MB table and 17 ms
// out of a 40 MB one (request-rate bound), hence the limit.  The domain comes from a sample: a key outside it, or one the table does not
// know, sets the flag and the call repeats through the routed form.
__global__ void __launch_bounds__(256) lookup_fill_kernel(GTable gt, const uint32_t* __restrict__ slot_gid, uint32_t kmin, uint32_t D, uint32_t* __restrict__ table, uint32_t* __restrict__ flag) {
    const uint32_t G = gt.flags[1];
    for (uint32_t s = blockIdx.x * 256 + threadIdx.x; s < G; s += gridDim.x * 256) {
        const uint32_t x = (uint32_t)*gt.key_p(s) - kmin;
        if (x < D) table[x] = slot_gid[s]; else *flag = 1u;
    }
}
__global__ void __launch_bounds__(256) lookup_assign_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t kmin, uint32_t D, const uint32_t* __restrict__ table,
                                                            uint32_t* __restrict__ reversemap, uint32_t* __restrict__ flag) {
    const uint32_t nvec = n >> 2;
    uint32_t bad = 0;
    for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < nvec; c += gridDim.x * 256) {
        const uint4 k = reinterpret_cast<const uint4*>(keys)[c];
        const uint32_t x0 = k.x - kmin, x1 = k.y - kmin, x2 = k.z - kmin, x3 = k.w - kmin;
        bad |= (x0 >= D) | (x1 >= D) | (x2 >= D) | (x3 >= D);
        uint4 g;
        g.x = table[x0 < D ? x0 : 0]; g.y = table[x1 < D ? x1 : 0]; g.z = table[x2 < D ? x2 : 0]; g.w = table[x3 < D ? x3 : 0];
        bad |= (g.x == 0xFFFFFFFFu) | (g.y == 0xFFFFFFFFu) | (g.z == 0xFFFFFFFFu) | (g.w == 0xFFFFFFFFu);
        reinterpret_cast<uint4*>(reversemap)[c] = g;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint32_t i = (nvec << 2) + threadIdx.x, x = keys[i] - kmin;
        const uint32_t g = table[x < D ? x : 0];
        bad |= (x >= D) | (g == 0xFFFFFFFFu);
        reversemap[i] = g;
    }
    if (bad) *flag = 1u;
}
// a column copied at the rate the shifts stream at (one 16-byte vector per lane, exact grid: 6.0 TB/s of combined traffic; the runtime's
// device-to-device copy moves the same bytes at 4.4)
__global__ void __launch_bounds__(256) copy_vec_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) emit_rows_kernel(AccSpec as, EmitSpec es, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) emit_record_from<false>(RowRecord{as, i}, i, es, (uint64_t)i);     // (the key columns: plain copies)
}
__global__ void __launch_bounds__(256) emit_kernel(GTable gt, const uint32_t* __restrict__ occ, const uint32_t* __restrict__ gid_of_occ, EmitSpec es,
                                                   const uint32_t* __restrict__ order, uint32_t gmax /* 0: no bound; else give up beyond it (ranks were not computed) */,
                                                   int occ_identity /* occ[i] == i (the record tables of the partition plans) */) {
    uint32_t G = gt.flags[1];
    if (gmax && G > gmax) return;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < G; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = order ? order[i0] : i0;
        // with `order` the group id is the walk position itself (order[gid_of_occ[i]] = i), and a record table is its own occupancy
        // list: two of the three random lines a group cost at 1e7 groups are not fetched
        const uint32_t s = occ_identity ? i : occ[i], g = order ? i0 : gid_of_occ[i];
        emit_record(gt, s, g, es, s == gt.cap ? EMPTY64 : (*gt.key_p(s)));
    }
}

// Huge group tables (aqg_sorted_tail): one workgroup per partition of the record planes.  The partition holds the groups whose first
// rows lie in one interval of <= C rows, so its first group id is its start offset and a group's id is that plus the number of set
// bits below its first row in a bitmap of the interval.  The records are permuted into id order inside LDS and emitted from there:
// every output column is written front to back, the key columns (wide tuples) are read in ascending row order.
__global__ void __launch_bounds__(1024, 8) sorted_emit_kernel(SortedParts sp, uint32_t G, uint32_t n_rows, int nacc, int has_count, int wide, EmitSpec es, uint32_t* __restrict__ flags) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t C = sp.cap, W = C / 32 + 8;
    uint64_t* sacc = reinterpret_cast<uint64_t*>(smem_raw);                 // [nacc][C]
    uint64_t* skey = sacc + (size_t)nacc * C;                               // [C] (packed keys only)
    uint32_t* sfirst = reinterpret_cast<uint32_t*>(skey + (wide ? 0 : C));  // [C]
    uint32_t* scount = sfirst + C;                                          // [C]
    uint32_t* bm = scount + C;                                              // [W] bitmap of the row interval
    uint32_t* wp = bm + W;                                                  // [W] set bits before every word
    __shared__ uint32_t wsum[16];
    const uint32_t NT = blockDim.x;
    GTable lt;
    lt.kb = reinterpret_cast<unsigned char*>(skey); lt.fb = reinterpret_cast<unsigned char*>(sfirst); lt.cb = reinterpret_cast<unsigned char*>(scount);
    lt.ab = reinterpret_cast<unsigned char*>(sacc);
    lt.kst = 8; lt.fst = 4; lt.cst = 4; lt.ast = 8; lt.astep = (uint64_t)C * 8; lt.cap = 0xFFFFFFFFu; lt.flags = nullptr; lt.has_count = has_count;
    bool k32 = wide != 0;
    for (int k = 0; k < es.nkeys; ++k) k32 = k32 && aqg_dtype_size_dev(es.key_dt[k]) == 4;
    for (uint32_t part = blockIdx.x; part < sp.nparts; part += gridDim.x) {
        const uint32_t b = sp.pstart[part], e = sp.pstart[part + 1];
        if (b >= e) continue;
        const uint64_t lo64 = (((uint64_t)part << 32) + sp.M - 1) / sp.M, hi64 = ((((uint64_t)part + 1) << 32) + sp.M - 1) / sp.M;
        const uint32_t lo = (uint32_t)lo64, hi = hi64 < n_rows ? (uint32_t)hi64 : n_rows;
        const uint32_t c = e - b, nw = (hi - lo + 31) / 32;
        if (hi - lo > C || c > C || e > G) { if (threadIdx.x == 0) flags[7] = 1; continue; }       // (the plan rules it out; the host checks the word behind this kernel)
        for (uint32_t w = threadIdx.x; w < nw; w += NT) bm[w] = 0;
        __syncthreads();
        // (four rows of a lane per step, their loads issued together from clamped indices: a partition is a chain of barriers, and a
        // loop that loads, uses and loads again puts one memory latency per row between them)
        for (uint32_t j0 = threadIdx.x; j0 < c; j0 += 4 * NT) {
            uint32_t fr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t j = j0 + u * NT; fr[u] = sp.first[b + (j < c ? j : c - 1)]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (j0 + u * NT < c) { const uint32_t r = fr[u] - lo; atomicOr(&bm[r >> 5], 1u << (r & 31)); }
        }
        __syncthreads();
        {   // nw <= 512 <= NT: one word per thread
            const uint32_t t = threadIdx.x < nw ? __popc(bm[threadIdx.x]) : 0;
            const uint32_t incl = wave_scan_incl(t, OpAdd{}, lane_id());
            if (lane_id() == 63) wsum[wave_id()] = incl;
            __syncthreads();
            uint32_t base = incl - t;
            for (int w = 0; w < wave_id(); ++w) base += wsum[w];
            if (threadIdx.x < nw) wp[threadIdx.x] = base;
        }
        __syncthreads();
        for (uint32_t j0 = threadIdx.x; j0 < c; j0 += 2 * NT) {
            uint32_t fr[2], cn[2];
            uint64_t ky[2], ac[4][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t j = j0 + u * NT, jj = b + (j < c ? j : c - 1);
                fr[u] = sp.first[jj];
                cn[u] = has_count ? sp.count[jj] : 0;
                ky[u] = wide ? 0ull : sp.key[jj];
#pragma unroll
                for (int a = 0; a < 4; ++a) ac[a][u] = a < nacc ? sp.acc[a][jj] : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (j0 + u * NT >= c) continue;
                const uint32_t r = fr[u] - lo;
                const uint32_t rank = wp[r >> 5] + __popc(bm[r >> 5] & ((1u << (r & 31)) - 1u));
                sfirst[rank] = fr[u];
                scount[rank] = cn[u];
                if (!wide) skey[rank] = ky[u];
#pragma unroll
                for (int a = 0; a < 4; ++a) if (a < nacc) sacc[(size_t)a * C + rank] = ac[a][u];
            }
            for (int a = 4; a < nacc; ++a)                    // (more than four accumulators: the rest one by one)
                for (int u = 0; u < 2; ++u) { const uint32_t j = j0 + u * NT; if (j < c) { const uint32_t r = fr[u] - lo; sacc[(size_t)a * C + wp[r >> 5] + __popc(bm[r >> 5] & ((1u << (r & 31)) - 1u))] = sp.acc[a][b + j]; } }
        }
        __syncthreads();
        if (k32) {      // wide tuples of 4-byte columns: the key loads of a record issued together (emit_record's run one after the other)
            for (uint32_t i0 = threadIdx.x; i0 < c; i0 += 2 * NT) {          // two records per step: up to sixteen key loads in flight
                uint32_t kv[2][MAXKEYS];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t i = i0 + u * NT, row = sfirst[i < c ? i : c - 1];
#pragma unroll
                    for (int k = 0; k < MAXKEYS; ++k) kv[u][k] = k < es.nkeys ? static_cast<const uint32_t*>(es.key_col[k])[row] : 0;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t i = i0 + u * NT, g = b + i;
                    if (i >= c) continue;
#pragma unroll
                    for (int k = 0; k < MAXKEYS; ++k) if (k < es.nkeys) static_cast<uint32_t*>(es.key_out[k])[g] = kv[u][k];
                    emit_record<false>(lt, i, g, es, 0);
                }
            }
        } else {
            for (uint32_t i = threadIdx.x; i < c; i += NT) emit_record(lt, i, b + i, es, wide ? (uint64_t)sfirst[i] : skey[i]);
        }
        __syncthreads();
    }
}

// ---- second pass of aqg_groupby_build: reversemap + counts ---------------------------------------
// LDS_COUNTS: group counts in an LDS histogram.  LDS_MAP: additionally a private copy of the {key -> dense id} map in LDS
// (small group counts: every lookup becomes an LDS probe instead of an L2 round trip).
template <bool LDS_COUNTS, bool LDS_MAP>
__global__ void __launch_bounds__(256) assign_kernel(KeySpec ks, GTable gt, const uint32_t* __restrict__ slot_gid, const uint32_t* __restrict__ occ, uint32_t n,
                                                     uint32_t G, uint32_t mcap, uint32_t* __restrict__ reversemap, uint32_t* __restrict__ counts) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* lc = reinterpret_cast<uint32_t*>(smem_raw);                       // [G] counts
    uint64_t* mkey = reinterpret_cast<uint64_t*>(smem_raw + (((size_t)G * 4 + 15) & ~(size_t)15));   // [mcap] keys
    uint32_t* mgid = reinterpret_cast<uint32_t*>(mkey + mcap);                  // [mcap] dense ids
    __shared__ uint32_t sentinel_gid;
    if constexpr (LDS_COUNTS) for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) lc[g] = 0;
    if constexpr (LDS_MAP) {
        for (uint32_t s = threadIdx.x; s < mcap; s += blockDim.x) mkey[s] = EMPTY64;
        if (threadIdx.x == 0) sentinel_gid = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < G; i += blockDim.x) {
            const uint32_t s0 = occ[i];
            const uint32_t gid = slot_gid[s0];
            if (s0 == gt.cap) { sentinel_gid = gid; continue; }
            const uint64_t key = *gt.key_p(s0);
            uint32_t s = hash64(key) & (mcap - 1);
            while (true) {                                                       // keys are distinct: plain claim by CAS
                unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&mkey[s]), EMPTY64, key);
                if (old == EMPTY64) { mgid[s] = gid; break; }
                s = (s + 1) & (mcap - 1);
            }
        }
    }
    if constexpr (LDS_COUNTS || LDS_MAP) __syncthreads();
    const uint32_t nchunk = n >> 2;
    const bool vec_ok = ks.nkeys == 1 && ks.total_bytes == 4;
    auto one = [&](uint64_t key) -> uint32_t {
        uint32_t g = 0;
        if constexpr (LDS_MAP) {
            if (key == EMPTY64) g = sentinel_gid;
            else {
                uint32_t s = hash64(key) & (mcap - 1);
                while (mkey[s] != key) s = (s + 1) & (mcap - 1);               // every key of the column is in the map
                g = mgid[s];
            }
        } else {
            uint32_t s = ks.wide ? gt_find_wide(gt, ks, (uint32_t)key) : gt_find(gt, key);
            g = s == FAIL ? 0u : slot_gid[s];
        }
        if constexpr (LDS_COUNTS) atomicAdd(&lc[g], 1u); else atomicAdd(&counts[g], 1u);
        return g;
    };
    uint32_t c_lo, c_hi;
    wg_span(nchunk, c_lo, c_hi);
    for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const size_t base = (size_t)c * 4;
        uint64_t key[4];
        if (vec_ok) {
            pack<uint32_t, 4> kv = *reinterpret_cast<const pack<uint32_t, 4>*>(static_cast<const uint32_t*>(ks.col[0]) + base);
#pragma unroll
            for (int j = 0; j < 4; ++j) key[j] = kv.v[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) key[j] = ks.wide ? (uint64_t)(base + j) : pack_key(ks, base + j);
        }
        pack<uint32_t, 4> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = one(key[j]);
        *reinterpret_cast<pack<uint32_t, 4>*>(reversemap + base) = o;
    }
    if (blockIdx.x == 0) {
        uint32_t row = (nchunk << 2) + threadIdx.x;
        if (row < n) reversemap[row] = one(ks.wide ? (uint64_t)row : pack_key(ks, row));
    }
    if constexpr (LDS_COUNTS) {
        __syncthreads();
        for (uint32_t g = threadIdx.x; g < G; g += blockDim.x) { uint32_t c = lc[g]; if (c) atomicAdd(&counts[g], c); }
    }
}

// ---- the exchange step of row-sharded group-bys ----------------------------------------------------------------
// pack: {ngroups, 0; key, low 64 bits of the aggregate} as int64 pairs -- the payload of the one all_gather
__global__ void __launch_bounds__(256) pack_kernel(const void* __restrict__ keys, int key_dt, const void* __restrict__ res, int res_dt, uint32_t G,
                                                   long long* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= G; i += gridDim.x * blockDim.x) {
        if (i == 0) { out[0] = G; out[1] = 0; continue; }
        const uint32_t g = i - 1;
        long long k;
        switch (key_dt) {
        case AQG_INT8: k = static_cast<const int8_t*>(keys)[g]; break;
        case AQG_INT16: k = static_cast<const int16_t*>(keys)[g]; break;
        case AQG_INT32: k = static_cast<const int32_t*>(keys)[g]; break;
        case AQG_UINT8: case AQG_BOOL: k = static_cast<const uint8_t*>(keys)[g]; break;
        case AQG_UINT16: k = static_cast<const uint16_t*>(keys)[g]; break;
        case AQG_UINT32: k = static_cast<const uint32_t*>(keys)[g]; break;
        default: k = static_cast<const long long*>(keys)[g]; break;
        }
        long long v;
        if (res_dt == AQG_INT128 || res_dt == AQG_UINT128) v = (long long) static_cast<const aqg_i128*>(res)[g].lo;      // sums: low 64 bits
        else switch (res_dt) {                                                                                            // min / max keep the value dtype
        case AQG_INT8: v = static_cast<const int8_t*>(res)[g]; break;
        case AQG_INT16: v = static_cast<const int16_t*>(res)[g]; break;
        case AQG_INT32: v = static_cast<const int32_t*>(res)[g]; break;
        case AQG_UINT8: case AQG_BOOL: v = static_cast<const uint8_t*>(res)[g]; break;
        case AQG_UINT16: v = static_cast<const uint16_t*>(res)[g]; break;
        case AQG_UINT32: v = static_cast<const uint32_t*>(res)[g]; break;
        default: v = static_cast<const long long*>(res)[g]; break;                                                        // 8-byte values, counts (uint64)
        }
        out[2 * i] = k; out[2 * i + 1] = v;
    }
}
// unpack the gathered payloads of `world` shards into one key column and one value column, shards in rank order
__global__ void __launch_bounds__(256) unpack_kernel(const long long* __restrict__ gathered, uint32_t world, uint32_t gmax, int key_dt,
                                                     void* __restrict__ keys, long long* __restrict__ vals, uint32_t* __restrict__ total_out /* [0] rows, [1] bad header */) {
    __shared__ uint32_t off[65];
    if (threadIdx.x == 0) {
        uint32_t o = 0, bad = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const long long c = gathered[(size_t)r * (gmax + 1) * 2];
            if (c < 0 || c > (long long)gmax) bad = 1;
            off[r] = o;
            o += bad ? 0u : (uint32_t)c;
        }
        off[world] = o;
        if (blockIdx.x == 0) { total_out[0] = o; total_out[1] = bad; }
    }
    __syncthreads();
    if (off[world] > world * gmax) return;
    const int key_size = aqg_dtype_size_dev(key_dt);
    for (uint32_t r = blockIdx.x; r < world; r += gridDim.x) {
        const long long* src = gathered + (size_t)r * (gmax + 1) * 2;
        const uint32_t cnt = off[r + 1] - off[r];
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            const uint32_t d = off[r] + i;
            vals[d] = src[3 + 2 * i];
            store_sized(keys, d, key_size, (unsigned long long)src[2 + 2 * i]);
        }
    }
}


// The whole merge of a few shard tables in ONE workgroup (world x gmax <= 2048 rows: h2o Q1 on 8 GPUs is 8 x 100): concatenate in
// rank order, group in an LDS table, rank the groups by first occurrence, write keys / aggregates / first rows.  One launch and one
// 8-byte copy instead of the generic group-by's ten launches (the merge cost 0.13 ms of a 1.55 ms step).
// SUM adds sign-extended int64 partials into 128 bits with two atomics (the carry out of the low word is exact under any order).
constexpr uint32_t MERGE_ROWS = 2048, MERGE_CAP = 4096;
constexpr unsigned long long MERGE_EMPTY = 0x8000000000000001ull;
__global__ void __launch_bounds__(1024) merge_small_kernel(const long long* __restrict__ gathered, uint32_t world, uint32_t gmax, int key_dt, int op,
                                                           void* __restrict__ keys_out, void* __restrict__ res_out, uint32_t* __restrict__ first_out,
                                                           uint32_t* __restrict__ info /* [0] groups, [1] bad header, [2] rows */) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    unsigned long long* tkey = reinterpret_cast<unsigned long long*>(smem_raw);            // [MERGE_CAP + 1] (last: the key equal to the empty mark)
    unsigned long long* tlo = tkey + MERGE_CAP + 1;                                        // [MERGE_CAP + 1]
    unsigned long long* thi = tlo + MERGE_CAP + 1;                                         // [MERGE_CAP + 1]
    uint32_t* tfirst = reinterpret_cast<uint32_t*>(thi + MERGE_CAP + 1);                   // [MERGE_CAP + 1]
    uint32_t* occ = tfirst + MERGE_CAP + 1;                                                // [MERGE_ROWS] occupied slots
    __shared__ uint32_t off[65];
    __shared__ uint32_t s_bad, s_g;
    if (threadIdx.x == 0) {
        uint32_t o = 0, bad = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const long long c = gathered[(size_t)r * (gmax + 1) * 2];
            if (c < 0 || c > (long long)gmax) bad = 1;
            off[r] = o;
            o += bad ? 0u : (uint32_t)c;
        }
        off[world] = o;
        s_bad = bad; s_g = 0;
    }
    const unsigned long long init = op == AQG_RED_MIN ? 0x7FFFFFFFFFFFFFFFull : op == AQG_RED_MAX ? 0x8000000000000000ull : 0ull;
    for (uint32_t t = threadIdx.x; t <= MERGE_CAP; t += blockDim.x) { tkey[t] = MERGE_EMPTY; tlo[t] = init; thi[t] = 0; tfirst[t] = 0xFFFFFFFFu; }
    __syncthreads();
    const uint32_t total = off[world];
    for (uint32_t i = threadIdx.x; i < total && !s_bad; i += blockDim.x) {
        uint32_t r = 0;
        while (i >= off[r + 1]) ++r;                                   // world <= 64
        const long long* src = gathered + (size_t)r * (gmax + 1) * 2 + 2 + 2 * (size_t)(i - off[r]);
        const unsigned long long key = (unsigned long long)src[0];
        const long long val = src[1];
        uint32_t slot = MERGE_CAP;                                     // the key that equals the empty mark lives in the extra slot
        if (key != MERGE_EMPTY) {
            slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 52);    // top 12 bits: MERGE_CAP slots
            while (true) {
                const unsigned long long cur = tkey[slot];
                if (cur == key) break;
                if (cur == MERGE_EMPTY) {
                    const unsigned long long old = atomicCAS(&tkey[slot], MERGE_EMPTY, key);
                    if (old == MERGE_EMPTY || old == key) break;
                }
                slot = (slot + 1) & (MERGE_CAP - 1);                   // at most MERGE_ROWS keys in MERGE_CAP slots: always ends
            }
        }
        atomicMin(&tfirst[slot], i);
        if (op == AQG_RED_MIN) atomicMin(reinterpret_cast<long long*>(&tlo[slot]), val);
        else if (op == AQG_RED_MAX) atomicMax(reinterpret_cast<long long*>(&tlo[slot]), val);
        else {
            const unsigned long long old = atomicAdd(&tlo[slot], (unsigned long long)val);
            const unsigned long long carry = old + (unsigned long long)val < old ? 1ull : 0ull;
            const unsigned long long hi_add = (val < 0 ? ~0ull : 0ull) + carry;
            if (hi_add) atomicAdd(&thi[slot], hi_add);
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t <= MERGE_CAP; t += blockDim.x) if (tfirst[t] != 0xFFFFFFFFu) occ[atomicAdd(&s_g, 1u)] = t;
    __syncthreads();
    const uint32_t G = s_g;
    for (uint32_t e = threadIdx.x; e < G; e += blockDim.x) {
        const uint32_t slot = occ[e], mine = tfirst[slot];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < G; ++j) rank += tfirst[occ[j]] < mine;      // first occurrences are distinct rows
        const unsigned long long key = slot == MERGE_CAP ? MERGE_EMPTY : tkey[slot];
        first_out[rank] = mine;
        store_sized(keys_out, rank, aqg_dtype_size_dev(key_dt), key);
        if (op == AQG_RED_MIN || op == AQG_RED_MAX) static_cast<unsigned long long*>(res_out)[rank] = tlo[slot];
        else { static_cast<aqg_i128*>(res_out)[rank] = aqg_i128{tlo[slot], thi[slot]}; }
    }
    if (threadIdx.x == 0) { info[0] = s_bad ? 0u : G; info[1] = s_bad; info[2] = total; }
}

} // namespace

// =================================================================================================
// host side
// =================================================================================================
namespace {

uint32_t next_pow2(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return (uint32_t)(p > 0x80000000ull ? 0x80000000ull : p); }

int dev_realloc(aqg_ctx* ctx, void** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return AQG_OK;
    if (*p) { aqg_pool_give(ctx, *p, *cap); *p = nullptr; *cap = 0; }        // (stream-ordered reuse; a buffer too large for the pool is freed, which synchronises)
    size_t want = need < 256 ? 256 : need;
    if (void* q = aqg_pool_take(ctx, want, cap)) { *p = q; return AQG_OK; }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) { ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e); return AQG_ERR_NOMEM; }
    *cap = want;
    return AQG_OK;
}

int make_keyspec(aqg_ctx* ctx, int nkeys, const int* dts, const void* const* keys, uint32_t n, KeySpec* ks) {
    if (nkeys < 1 || nkeys > MAXKEYS) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: 1..8 key columns");
    int bits = 0;
    ks->nkeys = nkeys;
    ks->range_known = 0; ks->range_lo = ks->range_hi = 0;
    for (int j = 0; j < nkeys; ++j) {
        if (!(dt_is_num(dts[j]) || dts[j] == AQG_BOOL)) return aqg_fail(ctx, AQG_ERR_DTYPE, "group-by: key dtype");
        if (dt_is_fp(dts[j])) return aqg_fail(ctx, AQG_ERR_DTYPE, "group-by: internal: floating key column reached the packed-key layer");
        if (!keys[j] && n) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: null key column");
        ks->dt[j] = dts[j]; ks->col[j] = keys[j]; ks->shift[j] = bits;
        bits += 8 * (int)aqg_dtype_size(dts[j]);
    }
    ks->wide = bits > 64;
    if (ks->wide) for (int j = 0; j < nkeys; ++j) ks->shift[j] = 0;
    ks->total_bytes = bits / 8;
    return AQG_OK;
}

struct Plan {
    AccSpec as;
    int need_count;
    int nagg;
    AggOut agg[MAXAGG];
    const StarJoin* sj;        // aqg_join_groupby_sum: the row pass is starjoin_kernel
};

int add_acc(Plan* p, int kind, int dt, const void* col, int square, int part = 0) {
    for (int a = 0; a < p->as.nacc; ++a)
        if (p->as.kind[a] == kind && p->as.dt[a] == dt && p->as.col[a] == col && p->as.square[a] == square && p->as.part[a] == part) return a;
    if (p->as.nacc >= MAXACC) return -1;
    int a = p->as.nacc++;
    p->as.kind[a] = kind; p->as.dt[a] = dt; p->as.col[a] = col; p->as.square[a] = square; p->as.part[a] = part;
    return a;
}

int make_plan(aqg_ctx* ctx, int naggs, const int* ops, const int* dts, const void* const* vals, uint32_t n, Plan* p) {
    memset(p, 0, sizeof *p);
    if (naggs < 0 || naggs > MAXAGG) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: 0..8 aggregates");
    p->nagg = naggs;
    for (int j = 0; j < naggs; ++j) {
        int op = ops[j], dt = dts[j];
        if (!dt_is_num(dt)) return aqg_fail(ctx, AQG_ERR_DTYPE, "group-by: value dtype");
        if (!vals[j] && n && op != AQG_RED_COUNT) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: null value column");
        bool fp = dt_is_fp(dt);
        AggOut& a = p->agg[j];
        a.op = op; a.dt = dt; a.acc0 = a.acc1 = a.acc2 = a.acc3 = -1; a.out = nullptr;
        const bool wide = dt == AQG_INT64 || dt == AQG_UINT64;
        const int addk = fp ? ACC_ADD_F : ACC_ADD_I;
        bool ok = true;
        switch (op) {
        case AQG_RED_SUM: case AQG_RED_AVG: case AQG_RED_VAR: case AQG_RED_STDDEV:
            if (wide) { a.acc0 = add_acc(p, addk, dt, vals[j], 0, 1); a.acc1 = add_acc(p, addk, dt, vals[j], 0, 2); ok = a.acc0 >= 0 && a.acc1 >= 0; }
            else { a.acc0 = add_acc(p, addk, dt, vals[j], 0); ok = a.acc0 >= 0; }
            if (op == AQG_RED_VAR || op == AQG_RED_STDDEV) {
                if (wide) { a.acc2 = add_acc(p, addk, dt, vals[j], 1, 1); a.acc3 = add_acc(p, addk, dt, vals[j], 1, 2); ok = ok && a.acc2 >= 0 && a.acc3 >= 0; }
                else { a.acc2 = add_acc(p, addk, dt, vals[j], 1); ok = ok && a.acc2 >= 0; }
            }
            if (op != AQG_RED_SUM) p->need_count = 1;
            break;
        case AQG_RED_SUMSQ:                                               // (internal: the accumulators VAR calls acc2 / acc3, emitted like a SUM)
            if (wide) { a.acc0 = add_acc(p, addk, dt, vals[j], 1, 1); a.acc1 = add_acc(p, addk, dt, vals[j], 1, 2); ok = a.acc0 >= 0 && a.acc1 >= 0; }
            else { a.acc0 = add_acc(p, addk, dt, vals[j], 1); ok = a.acc0 >= 0; }
            break;
        case AQG_RED_MIN: a.acc0 = add_acc(p, ACC_MIN, dt, vals[j], 0); ok = a.acc0 >= 0; break;
        case AQG_RED_MAX: a.acc0 = add_acc(p, ACC_MAX, dt, vals[j], 0); ok = a.acc0 >= 0; break;
        case AQG_RED_COUNT: p->need_count = 1; break;
        default: return aqg_fail(ctx, AQG_ERR_DTYPE, "group-by: FIRST/LAST need row lists (use aqg_grouped_reduce)");
        }
        if (!ok) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: too many accumulators (8 per call)");
    }
    return AQG_OK;
}

// One attempt at a given global capacity.  Returns AQG_ERR_OVERFLOW when the table filled up.
struct DenseOut { bool used; DenseSpec spec; };
constexpr int AQG_ERR_RANGE_MISS = -1001;            // internal: sampled key ranges missed a value; run_with_retry repeats the attempt      // tells aqg_groupby_build that the table is the direct-indexed one

int run_agg(aqg_ctx* ctx, const KeySpec& ks_in, const Plan& plan_in, uint32_t n, uint32_t hint, bool for_build, aqg_groupby* h,
            GTable* gt_out, uint32_t** slot_gid_out, uint32_t** occ_out = nullptr, DenseOut* dense_out = nullptr) {
    // Packed keys (<= 8 bytes) with more groups expected than their partition plans reach (2^25) are handled as WIDE tuples: that plan
    // partitions on a hash of the tuple and compares tuples through representative rows, whatever the key width, up to one group per row
    // (2e8 unique 4-byte keys: 107 ms through the HBM table of scattered device atomics they fell to before)
    KeySpec ks = ks_in;
    static const bool widen_off = getenv("AQG_DISABLE_WIDEN_PACKED") != nullptr;      // A/B measurements only
    if (!widen_off && !ks.wide && !for_build && hint > (1u << 25) && n >= (1u << 20) && !plan_in.sj && plan_in.as.nacc <= 4) {
        ks.wide = 1;
        for (int j = 0; j < ks.nkeys; ++j) ks.shift[j] = 0;
    }
    // more than ~1.6e7 groups expected out of a partition plan: the records are ORDERED (aqg_sorted_tail) instead of ranked through a
    // bitmap over the rows and gathered (h2o Q10, 1e9 groups: that tail took 219 of 317 ms and fetched 900 GB)
    static const uint32_t sorted_min = getenv("AQG_SORTED_TAIL_MIN") ? (uint32_t)atoi(getenv("AQG_SORTED_TAIL_MIN")) : (1u << 24);
    // A BUILD above the LDS tables takes the partition plans too (partition1.hip: the group table with counts, then one more pass over the
    // partitioned rows for the id of every row) instead of inserting every row into an HBM table and looking every row up again
    static const bool build_part_off = getenv("AQG_DISABLE_BUILD_PARTITION") != nullptr;     // A/B measurements only
    const bool build_part = for_build && !build_part_off && !ks.wide && n >= (1u << 20) && hint > 3072 && hint <= (1u << 25) && hint < sorted_min;
    Plan plan = plan_in;
    if (build_part) plan.need_count = 1;          // the group sizes come out of the partition aggregation
    const AccSpec& as = plan.as;
    const bool k32 = ks.nkeys == 1 && ks.total_bytes == 4;
    uint32_t gcap = next_pow2((uint64_t)(hint < 512 ? 512 : hint) * 2);
    // LDS mode, small: one workgroup's table (75 % load) fits 64 KB, several workgroups per CU.
    // LDS mode, big:   one 1024-thread workgroup per CU with a table of up to 150 KB, and up to MAX_PASSES passes over the
    //                  rows, each aggregating the keys of one hash class (agg_kernel).  Beyond that rows go straight to HBM.
    constexpr uint32_t MAX_PASSES = 4;
    constexpr size_t LDS_SMALL = 76 * 1024, LDS_BIG = 150 * 1024;      // small: two workgroups per CU still fit
    bool use_lds = hint <= 3072 && !ks.wide;   // wide tuples compare against HBM-resident rows: HBM mode
    uint32_t lcap = use_lds ? next_pow2((uint64_t)(hint < 64 ? 64 : hint) * 4 / 3 + 1) : 0;
    if (use_lds && lcap < 256) lcap = 256;
    const size_t lds_slot_bytes = 8 + 8 * (size_t)as.nacc + (k32 ? 0 : 4) + (plan.need_count ? 4 : 0);
    if (use_lds && (size_t)(lcap + 1) * lds_slot_bytes > LDS_SMALL) { use_lds = false; lcap = 0; }
    if (plan.sj && !(use_lds && k32))
        return aqg_fail(ctx, AQG_ERR_ARG, "aqg_join_groupby_sum: needs one 4-byte group key and at most 3072 groups (compose aqg_join_lookup / aqg_gather / aqg_ewise / aqg_groupby_agg beyond that)");
    uint32_t npass = 1;
    bool big_lds = false;
    // dense key domain (dense.hip): direct-indexed tables when the product of the key columns' value ranges is small --
    // also for tuples wider than 64 bits.  Costs one more pass over the key columns, so it is only tried where the
    // alternatives are the multi-pass hashed table or the partition pipeline.
    bool dense = false;
    DenseSpec dspec;
    static const bool dense_off = getenv("AQG_DISABLE_DENSE") != nullptr;    // A/B measurements only
    if (!dense_off && !plan.sj && !use_lds && (!for_build || dense_out) && n >= (1u << 20) &&
        (uint64_t)hint <= (uint64_t)(DENSE_LDS_BYTES / aqg_dense_slot_bytes(as, plan.need_count)) * DENSE_MAX_PASSES) {
        long long mins[MAXKEYS], maxs[MAXKEYS];
        bool ok = false;
        AQG_TRY(aqg_ws_reset(ctx));
        AQG_TRY(aqg_ws_ensure(ctx, 4096));
        // large inputs: ranges from the first 2^20 rows (a full pass over the key columns costs a third of Q2); the kernels check
        // every row against them and flag a miss, which re-runs the call once with exact ranges (and remembers it in the handle)
        const bool sampled = n >= (1u << 22) && !h->dense_exact;
        bool cached = sampled && h->range_valid && h->range_nkeys == ks.nkeys && h->range_n == n;
        for (int c = 0; c < ks.nkeys && cached; ++c) cached = h->range_col[c] == ks.col[c] && h->range_dt[c] == ks.dt[c];
        if (cached) { for (int c = 0; c < ks.nkeys; ++c) { mins[c] = h->range_min[c]; maxs[c] = h->range_max[c]; } ok = true; }
        else {
            AQG_TRY(aqg_key_ranges(ctx, ks, sampled ? (1u << 20) : n, mins, maxs, &ok, sampled ? n : 0u));
            h->range_valid = sampled && ok;
            if (h->range_valid) {
                h->range_nkeys = ks.nkeys; h->range_n = n;
                for (int c = 0; c < ks.nkeys; ++c) { h->range_col[c] = ks.col[c]; h->range_dt[c] = ks.dt[c]; h->range_min[c] = mins[c]; h->range_max[c] = maxs[c]; }
            }
        }
        dense = ok && aqg_dense_plan(ks, mins, maxs, as, plan.need_count, &dspec);
        dspec.sampled = sampled;
        if (dense && dspec.D <= 1536) {          // (ranges from a sample are fine here: the hashed table takes any key)
            // a tiny domain under a large hint: the small hashed table after all (1024 lanes on a hundred hot direct-indexed
            // slots serialise on LDS atomics: 6.2 ms per 1e9 rows against 2.7 ms)
            const uint32_t small_cap = next_pow2((uint64_t)(dspec.D < 64 ? 64 : dspec.D) * 4 / 3 + 1);
            if ((size_t)(small_cap + 1) * lds_slot_bytes <= LDS_SMALL) { dense = false; use_lds = true; lcap = small_cap < 256 ? 256 : small_cap; hint = dspec.D; }
        }
        if (dense) gcap = dspec.D;
    }
    if (!dense && !use_lds && !ks.wide && n >= (1u << 20)) {
        const uint32_t max_slots = (uint32_t)(LDS_BIG / lds_slot_bytes) - 1;
        const uint32_t per_pass = max_slots - (max_slots >> 2);
        const uint64_t want = ((uint64_t)hint + per_pass - 1) / per_pass;
        if (want <= MAX_PASSES) { use_lds = big_lds = true; npass = (uint32_t)want; lcap = max_slots; }
    }
    const uint64_t lds_group_cap = use_lds ? (uint64_t)npass * (lcap - (lcap >> 2)) : 0;
    const bool small_rank = hint <= 4096;
    const uint32_t nwords = aqg_ceil_div(n, 32), ntiles = aqg_ceil_div(nwords, 1024);
    // groups beyond the LDS tables: partition the rows instead of hammering an HBM table with scattered atomics
    // (partition.hip: h2o Q5, 1e9 rows, 1e7 groups: 42 ms against 141 ms); the build path keeps the HBM table because
    // its second pass looks keys up in it.  AQG_DISABLE_PARTITION=1 forces the HBM table (A/B measurements only).
    static const bool part_off = getenv("AQG_DISABLE_PARTITION") != nullptr;
    bool use_part = !part_off && !dense && !use_lds && !ks.wide && (!for_build || build_part) && n >= (1u << 20) && hint <= (1u << 25);
    if (use_part && for_build) {                  // (the build's id pass knows the one- and two-level plans only)
        const uint32_t bp = aqg_partition_parts(ks.total_bytes <= 4 ? 4 : 8, as, plan.need_count, hint);
        if (!bp || bp > AQG_P2_MAXPARTS) use_part = false;
    }
    // the build over a dense key domain small enough for a key -> id look-up table (lookup_assign_kernel): no partitioned rows kept, no routing
    bool lookup_build = false;
    uint32_t lk_min = 0, lk_D = 0;
    static const bool lookup_off = getenv("AQG_DISABLE_LOOKUP_BUILD") != nullptr;     // A/B measurements only
    if (use_part && for_build && !lookup_off && !h->no_lookup_build && ks.nkeys == 1 && !ks.wide && (ks.dt[0] == AQG_INT32 || ks.dt[0] == AQG_UINT32) &&
        ((uintptr_t)ks.col[0] & 15) == 0 && n >= (1u << 22)) {
        long long mn[MAXKEYS], mx[MAXKEYS];
        bool ok = false;
        AQG_TRY(aqg_ws_reset(ctx));
        AQG_TRY(aqg_ws_ensure(ctx, 4096));
        AQG_TRY(aqg_key_ranges(ctx, ks, 1u << 20, mn, mx, &ok, n));         // (a sample spread over the column: the look-up pass checks every row)
        if (ok && mx[0] >= mn[0]) {
            const long long span = mx[0] - mn[0] + 1, room = span / 64 + 1024, lo = mn[0] - room, hi = mx[0] + room;
            if (hi - lo + 1 <= (1ll << 21)) { lookup_build = true; lk_min = (uint32_t)lo; lk_D = (uint32_t)(hi - lo + 1); }
        }
    }
    uint32_t part_lcap = 0, pbits = 0;
    if (use_part) {
        // LDS table of one partition: as many slots as fit the budget (the slot of a hash is a multiply-shift, so the
        // capacity need not be a power of two).  Partitions are sized for a LOW load factor: probe sequences are walked by
        // whole wavefronts, and measured at 1e9 rows / 1e7 groups the LDS aggregation takes 3.9 ms at load 0.20, 5.6 ms at
        // 0.22-0.25, 6.7-7.2 ms at 0.30-0.33 and 18 ms at 0.6 (two accumulators), while one more partition bit costs the two
        // scatter passes 1-2 ms.  AQG_PART_LF1000 / AQG_PART_LDSKB / AQG_PART_LCAP: measurement switches.
        static const int lf1000 = getenv("AQG_PART_LF1000") ? atoi(getenv("AQG_PART_LF1000")) : 210;
        static const int lds_kb = getenv("AQG_PART_LDSKB") ? atoi(getenv("AQG_PART_LDSKB")) : 60;
        const size_t sb = 16 + 8 * (size_t)as.nacc;
        part_lcap = (uint32_t)((size_t)lds_kb * 1024 / sb) - 1;
        { static const int lcap_env = getenv("AQG_PART_LCAP") ? atoi(getenv("AQG_PART_LCAP")) : 0; if (lcap_env > 0) part_lcap = (uint32_t)lcap_env; }
        pbits = 10;                                  // at least 1024 partitions: every CU gets several
        while (pbits < 16 && ((uint64_t)hint >> pbits) * 1000 > (uint64_t)part_lcap * lf1000) ++pbits;
        gcap = (uint32_t)((uint64_t)hint + hint / 4 + 4096 > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : (uint64_t)hint + hint / 4 + 4096);   // compact record table
    }
    // partition1.hip: ONE level up to ~1000 partitions (every plane moves once), two levels of <= 64 bins up to 4096 (the runs a
    // tile writes stay a kilobyte long); beyond that the round-1 pipeline of partition.hip
    static const bool p1_off_env = getenv("AQG_DISABLE_P1") != nullptr;      // A/B measurements only
    const bool p1_off = p1_off_env && !for_build;
    static const uint32_t p1_max = getenv("AQG_P1_MAX") ? (uint32_t)atoi(getenv("AQG_P1_MAX")) : 1024u;
    int part_layout = AQG_P1_LAYOUT_DENSE_IDS;
    const uint32_t parts = use_part && !p1_off ? aqg_partition_parts(ks.total_bytes <= 4 ? 4 : 8, as, plan.need_count, hint, &part_layout) : 0;
    const uint32_t p1_bins = parts && parts <= p1_max && parts <= AQG_P1_MAXBINS ? parts : 0;
    const uint32_t p2_parts = parts && !p1_bins && parts <= AQG_P2_MAXPARTS ? parts : 0;
    // tuples wider than 8 bytes with many groups (h2o Q10): hash-partitioned rows, every partition grouped inside LDS
    // every row its own group out of a partition plan: the result can be written from the input rows (emit_rows_kernel below) -- when nobody asks for the table itself
    static const bool rows_off = getenv("AQG_DISABLE_ROW_EMIT") != nullptr;                 // A/B measurements only
    const bool rows_possible = !rows_off && n >= (1u << 16) && !for_build && !plan.sj && !gt_out && !slot_gid_out && !occ_out;
    const bool use_wpart = !part_off && !p1_off && !dense && !use_lds && ks.wide && !for_build && !plan.sj && n >= (1u << 20) && hint > (1u << 20) && as.nacc <= 4 &&
                           !h->no_wide_part && aqg_partitionw_applies(ks, as, n, hint);
    if (use_wpart) gcap = (uint32_t)((uint64_t)hint + hint / 4 + 4096 > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : (uint64_t)hint + hint / 4 + 4096);

    const bool sorted_tail = (use_part || use_wpart) && !for_build && hint >= sorted_min && !h->no_sorted_tail && aqg_sorted_tail_plan(n, as.nacc, ks.wide != 0, nullptr);
    // ---- workspace ----------------------------------------------------------------------------
    size_t slots = (size_t)gcap + 1;
    uint32_t stride = 16;
    while (stride < 16 + 8 * (uint32_t)as.nacc) stride <<= 1;
    size_t need = slots * (size_t)stride + 4096 + 256 * 16;
    if (!sorted_tail) need += slots * (4 + 4 + 4);
    if (!small_rank && !sorted_tail) need += (size_t)nwords * 8 + (size_t)ntiles * 8 + 8192;
    const bool ordered_emit = !small_rank && hint >= (1u << 20) && !sorted_tail;
    if (ordered_emit) need += slots * 4 + 4096;
    size_t part_need = 0;
    if (use_wpart) part_need = aqg_partitionw_ws_bytes(ctx, ks, n, as, hint) + 65536;
    else if (p1_bins) part_need = aqg_partition1_ws_bytes(ctx, ks, n, as, p1_bins) + 65536;
    else if (p2_parts) part_need = aqg_partition2_ws_bytes(ctx, ks, n, as, p2_parts) + 65536;
    else if (use_part) part_need = aqg_partition_ws_bytes(n, ks.total_bytes <= 4 ? 4 : 8, as, pbits) + 65536;
    // (the partition buffers are dead once the record table is written: the ordering pass takes their place in the arena)
    const size_t sort_need = sorted_tail ? aqg_sorted_tail_ws_bytes(gcap, n, as.nacc, ks.wide != 0) : 0;
    need += part_need > sort_need ? part_need : sort_need;
    if (for_build && use_part) need += lookup_build ? ((size_t)lk_D + 64) * 4 + 4096 : aqg_partition_assign_ws_bytes(n);
    AQG_TRY(aqg_ws_reset(ctx));
    AQG_TRY(aqg_ws_ensure(ctx, need));
    GTable gt;
    memset(&gt, 0, sizeof gt);
    gt.cap = gcap;
    gt.has_count = plan.need_count;
    PartRows prows;
    memset(&prows, 0, sizeof prows);
    uint32_t *occ, *gid_of_occ, *slot_gid, *bitmap = nullptr, *word_prefix = nullptr, *tile_total = nullptr;
    {
        unsigned char* base = nullptr;
        AQG_TRY(aqg_ws_get(ctx, slots * stride, &base));
        const bool records = (use_part || use_wpart || hint > (1u << 17)) && !sorted_tail;   // (the ordering pass moves column planes)
        if (records) {
            gt.kb = base; gt.fb = base + 8; gt.cb = base + 12; gt.ab = base + 16;
            gt.kst = gt.fst = gt.cst = gt.ast = stride; gt.astep = 8;
        } else {
            gt.kb = base; gt.fb = base + slots * 8; gt.cb = gt.fb + slots * 4; gt.ab = gt.cb + slots * 4;
            gt.kst = 8; gt.fst = 4; gt.cst = 4; gt.ast = 8; gt.astep = (uint64_t)slots * 8;
        }
    }
    AQG_TRY(aqg_ws_get(ctx, 64, &gt.flags));
    if (sorted_tail && use_wpart) gt.kb = nullptr;       // wide tuples through the ordering tail: the key word would repeat the first-row plane
    occ = gid_of_occ = slot_gid = nullptr;
    if (!sorted_tail) {
        AQG_TRY(aqg_ws_get(ctx, slots, &occ));
        AQG_TRY(aqg_ws_get(ctx, slots, &gid_of_occ));
        AQG_TRY(aqg_ws_get(ctx, slots, &slot_gid));
    }
    // few groups over many rows: the context's all-zero bitmap, only the tiles that hold a bit are scanned (bitmap_set_kernel)
    static const bool sparse_off = getenv("AQG_DISABLE_SPARSE_RANK") != nullptr;
    const bool sparse_rank = !small_rank && !sorted_tail && !sparse_off && nwords >= (1u << 16) && (uint64_t)hint * 64 < nwords;
    uint32_t* tile_mark = nullptr;
    if (!small_rank && !sorted_tail) {
        if (sparse_rank) {
            if (ctx->rank_bm_words < nwords) {
                if (ctx->rank_bm) { AQG_HIP(ctx, hipStreamSynchronize(ctx->stream)); AQG_HIP(ctx, hipFree(ctx->rank_bm)); ctx->rank_bm = nullptr; ctx->rank_bm_words = 0; }
                if (hipMalloc(&ctx->rank_bm, (size_t)nwords * 4) != hipSuccess) { (void)hipGetLastError(); return aqg_fail(ctx, AQG_ERR_NOMEM, "group-by: no memory for the ranking bitmap"); }
                ctx->rank_bm_words = nwords;
                AQG_HIP(ctx, hipMemsetAsync(ctx->rank_bm, 0, (size_t)nwords * 4, ctx->stream));
            }
            bitmap = ctx->rank_bm;
            AQG_TRY(aqg_ws_get(ctx, ntiles + 1, &tile_mark));
        } else AQG_TRY(aqg_ws_get(ctx, nwords, &bitmap));
        AQG_TRY(aqg_ws_get(ctx, nwords, &word_prefix));
        AQG_TRY(aqg_ws_get(ctx, ntiles + 1, &tile_total));
    }
    if (!use_part && !use_wpart) hipLaunchKernelGGL(gt_init_kernel, dim3(aqg_grid(ctx, slots, 256, 1, 8)), dim3(256), 0, ctx->stream, gt, as);
    else AQG_HIP(ctx, hipMemsetAsync(gt.flags, 0, 64 * 4, ctx->stream));
    if (bitmap && !sparse_rank) AQG_HIP(ctx, hipMemsetAsync(bitmap, 0, (size_t)nwords * 4, ctx->stream));
    if (tile_mark) AQG_HIP(ctx, hipMemsetAsync(tile_mark, 0, ((size_t)ntiles + 1) * 4, ctx->stream));

    // fast path eligibility: LDS mode, 16-byte aligned columns, one or two 4-byte integer keys or one 8-byte key, up to four accumulators
    // of any kind over integer / floating value columns (also the first pass of aqg_groupby_build: no accumulators, only the distinct keys)
    auto key32 = [&](int j) { return (ks.dt[j] == AQG_INT32 || ks.dt[j] == AQG_UINT32) && ((uintptr_t)ks.col[j] & 15) == 0; };
    // two 4-byte key columns, or one 8-byte key column whose bits are the packed key
    static const bool fast64_off = getenv("AQG_DISABLE_FAST64") != nullptr;                  // A/B measurements only (read once per process)
    const bool fast_key8 = ks.nkeys == 1 && !ks.wide && (ks.dt[0] == AQG_INT64 || ks.dt[0] == AQG_UINT64) && ((uintptr_t)ks.col[0] & 15) == 0 && !fast64_off;
    const bool fast_k64 = (ks.nkeys == 2 && !ks.wide && ks.total_bytes == 8 && key32(0) && key32(1)) || fast_key8;
    bool fast = use_lds && !plan.sj && !big_lds && ((k32 && key32(0)) || fast_k64) && n >= 8 && (as.nacc >= 1 || plan.need_count || for_build) && as.nacc <= 4;
    FastVals fv;
    memset(&fv, 0, sizeof fv);
    // value columns: 4 bytes wide, or 4 and 8 bytes wide (an int64 sum takes two accumulators)
    auto wide_dt = [](int dt) { return dt == AQG_INT64 || dt == AQG_UINT64 || dt == AQG_DOUBLE; };
    auto narrow_dt = [](int dt) { return dt == AQG_INT32 || dt == AQG_UINT32 || dt == AQG_FLOAT; };
    auto tiny_dt = [](int dt) { return dt == AQG_INT8 || dt == AQG_UINT8 || dt == AQG_BOOL || dt == AQG_INT16 || dt == AQG_UINT16; };
    bool fast_v8 = false;                       // some value column is 1, 2 or 8 bytes wide: the VW = 8 instantiation (it takes 4-byte ones, too)
    for (int a = 0; a < as.nacc; ++a) fast_v8 = fast_v8 || wide_dt(as.dt[a]) || tiny_dt(as.dt[a]);
    fast_v8 = fast_v8 && !fast64_off;
    for (int a = 0; a < as.nacc && fast; ++a) {
        const int dt = as.dt[a];
        if ((uintptr_t)as.col[a] & 15) fast = false;
        if (fast_v8) { if (!wide_dt(dt) && !narrow_dt(dt) && !tiny_dt(dt)) fast = false; }
        else if (as.part[a] || !narrow_dt(dt)) fast = false;
        fv.col[a] = as.col[a];
        fv.vkind[a] = dt == AQG_INT32 ? 0 : dt == AQG_UINT32 ? 1 : dt == AQG_FLOAT ? 2 : dt == AQG_INT64 ? 3 : dt == AQG_UINT64 ? 4 : dt == AQG_DOUBLE ? 5 :
                      dt == AQG_INT8 ? 6 : (dt == AQG_UINT8 || dt == AQG_BOOL) ? 7 : dt == AQG_INT16 ? 8 : 9;
        fv.kind[a] = as.kind[a];
        fv.square[a] = as.square[a];
        fv.part[a] = as.part[a];
    }
    // ---- pass over the rows ---------------------------------------------------------------------
    if (n && plan.sj) {
        const size_t lds = (size_t)(lcap + 1) * 24 + (size_t)plan.sj->dcap * 8 + 64;
        unsigned bpc = lds <= 20 * 1024 ? 8 : lds <= 40 * 1024 ? 4 : lds <= 80 * 1024 ? 2 : 1;
        AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(&starjoin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(starjoin_kernel, dim3(aqg_grid(ctx, n / 8 + 1, 256, 2, bpc)), dim3(256), lds, ctx->stream, static_cast<const uint32_t*>(ks.col[0]), *plan.sj, gt, n, lcap);
        aqg_kernel_timer_end(ctx);
        AQG_TRY(aqg_check_launch(ctx, "starjoin_kernel"));
    } else if (n && fast) {
        const size_t lds = (size_t)(lcap + 1) * ((fast_k64 ? 8 : 4) + 8 * (size_t)as.nacc + (plan.need_count ? 4 : 0)) + 64;
        // the table is sized by the hint, not by the groups that show up: a large one leaves room for few workgroups per CU, and with 256
        // threads each the LDS round trips of two accumulators per row are no longer hidden (var(v1), 100 groups, hint 1024: two
        // workgroups = 8 wavefronts per CU ran at 43 % of the HBM roofline).  So the workgroup grows with the table: 32 wavefronts per CU.
        static const unsigned block_env = getenv("AQG_FAST_BLOCK") ? (unsigned)atoi(getenv("AQG_FAST_BLOCK")) : 0;
        const unsigned block = block_env ? block_env : lds <= 20 * 1024 ? 256 : lds <= 40 * 1024 ? 512 : 1024;
        unsigned bpc = lds <= 20 * 1024 ? 8 : lds <= 40 * 1024 ? 4 : lds <= 78 * 1024 ? 2 : 1;
        unsigned grid = aqg_grid(ctx, n / 8 + 1, block, 2, bpc);
        // (more workgroups than fit the chip cost more in table merges than they gain: 8192 -> +3 %, 32768 -> +30 % on Q1)
        const uint32_t* khi = fast_k64 && !fast_key8 ? static_cast<const uint32_t*>(ks.col[1]) : nullptr;
        h->plan_bits = AQG_PLAN_FAST_LDS;
        const int rc = aqg_fast_aggregate(ctx, static_cast<const uint32_t*>(ks.col[0]), khi, fast_k64, fast_v8, as.nacc, plan.need_count != 0, fv, gt, n, lcap, lds, grid, block);
        AQG_TRY(rc);
    } else if (n && dense) {
        h->plan_bits = AQG_PLAN_DENSE;
        AQG_TRY(aqg_dense_aggregate(ctx, ks, dspec, as, n, plan.need_count, gt));
    } else if (n && (use_wpart || use_part)) {
        const size_t mark = ctx->ws_off;
        memset(&prows, 0, sizeof prows);
        h->plan_bits = (use_wpart ? AQG_PLAN_PART_WIDE : p1_bins ? AQG_PLAN_PART_ONE : p2_parts ? AQG_PLAN_PART_TWO : AQG_PLAN_PART_ROUND1) | (sorted_tail ? AQG_PLAN_SORTED_TAIL : 0u);
        if (use_wpart) {
            int pack = h->no_pack ? 0 : 1;
            AQG_TRY(aqg_partitionw_aggregate(ctx, ks, as, n, plan.need_count, gt, gcap, h->wide_seed, hint, &pack, &h->wide_rows, rows_possible));
            if (pack) h->plan_bits |= AQG_PLAN_PACKED_KEYS;
        }
        else if (p1_bins) {
            int ranged = h->no_pack ? 0 : 1;
            AQG_TRY(aqg_partition1_aggregate(ctx, ks, as, n, p1_bins, plan.need_count, gt, gcap, for_build && !lookup_build ? &prows : nullptr, part_layout, &ranged));
            if (ranged & 2) h->plan_bits |= AQG_PLAN_RANGE_PARTITIONS;
            if (ranged & 1) h->plan_bits |= AQG_PLAN_PACKED_VALUES;
        }
        else if (p2_parts) {
            int pack = h->no_pack ? 0 : 1;
            AQG_TRY(aqg_partition2_aggregate(ctx, ks, as, n, p2_parts, plan.need_count, gt, gcap, for_build && !lookup_build ? &prows : nullptr, &pack, part_layout));
            if (pack & 1) h->plan_bits |= AQG_PLAN_PACKED_VALUES;
            if (pack & 2) h->plan_bits |= AQG_PLAN_RANGE_PARTITIONS;
        }
        else AQG_TRY(aqg_partition_aggregate(ctx, ks, as, n, pbits, part_lcap, plan.need_count, gt, gcap));
        if (sorted_tail) ctx->ws_off = mark;       // stream order: whatever is allocated there next is written after these kernels
        else hipLaunchKernelGGL(occ_iota_kernel, dim3(aqg_grid(ctx, slots, 256, 1, 8)), dim3(256), 0, ctx->stream, occ, (uint32_t)slots);
    } else if (n) {
        uint32_t lrep = 1;
        if (use_lds) {   // replicate small tables: conflicts fall, LDS stays under ~32 KB per workgroup
            size_t per = (size_t)(lcap + 1) * (8 + 8 * (size_t)as.nacc + (k32 ? 0 : 4) + (plan.need_count ? 4 : 0));
            uint32_t want = 1;   // measured on MI355X (h2o Q1, 100 groups): 1 replica 1.98 ms, 4 replicas 2.12 ms per 1e9 rows
            while (lrep < want && per * lrep * 2 <= 64 * 1024) lrep *= 2;
        }
        size_t lds = use_lds ? (size_t)lrep * (lcap + 1) * (8 + 8 * (size_t)as.nacc + (k32 ? 0 : 4) + (plan.need_count ? 4 : 0)) + 4 * 64 : 0;
        unsigned bpc = !use_lds ? 8 : lds <= 20 * 1024 ? 8 : lds <= 40 * 1024 ? 4 : lds <= 80 * 1024 ? 2 : 1;
        h->plan_bits = big_lds ? AQG_PLAN_BIG_LDS : use_lds ? AQG_PLAN_SMALL_LDS : AQG_PLAN_HBM_TABLE;
        const unsigned block = big_lds ? (as.nacc <= 2 ? 1024 : 512) : 256;
        unsigned grid = big_lds ? (unsigned)ctx->num_cu : aqg_grid(ctx, n / 4 + 1, 256, 2, bpc);
        auto launch = [&](auto kern) -> int {
            if (lds) AQG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, ctx->stream, ks, as, gt, n, lcap, plan.need_count, lrep, npass);
            return AQG_OK;
        };
        auto by_nacc = [&](auto lds_tag, auto k32_tag, auto block_tag) -> int {
            constexpr bool L = decltype(lds_tag)::value, K = decltype(k32_tag)::value;
            constexpr int B = decltype(block_tag)::value;
            // big tables: 1024 threads per workgroup up to 2 accumulators (<= 128 VGPRs without spills), 512 beyond
#define AQG_AGG_CASE(N) case N: if constexpr ((B == 1024 && N > 2) || (B == 512 && N <= 2)) return AQG_ERR_ARG; else return launch(&agg_kernel<L, K, N, B>);
            switch (as.nacc) {
            AQG_AGG_CASE(0) AQG_AGG_CASE(1) AQG_AGG_CASE(2) AQG_AGG_CASE(3) AQG_AGG_CASE(4) AQG_AGG_CASE(5) AQG_AGG_CASE(6) AQG_AGG_CASE(7)
            default: if constexpr (B == 1024) return AQG_ERR_ARG; else return launch(&agg_kernel<L, K, 8, B>);
            }
#undef AQG_AGG_CASE
        };
        using B256 = std::integral_constant<int, 256>;
        using B512 = std::integral_constant<int, 512>;
        using B1024 = std::integral_constant<int, 1024>;
        aqg_kernel_timer_begin(ctx);
        if (big_lds && block == 1024) { if (k32) AQG_TRY(by_nacc(std::true_type{}, std::true_type{}, B1024{})); else AQG_TRY(by_nacc(std::true_type{}, std::false_type{}, B1024{})); }
        else if (big_lds) { if (k32) AQG_TRY(by_nacc(std::true_type{}, std::true_type{}, B512{})); else AQG_TRY(by_nacc(std::true_type{}, std::false_type{}, B512{})); }
        else if (use_lds) { if (k32) AQG_TRY(by_nacc(std::true_type{}, std::true_type{}, B256{})); else AQG_TRY(by_nacc(std::true_type{}, std::false_type{}, B256{})); }
        else { if (k32) AQG_TRY(by_nacc(std::false_type{}, std::true_type{}, B256{})); else AQG_TRY(by_nacc(std::false_type{}, std::false_type{}, B256{})); }
        aqg_kernel_timer_end(ctx);
        AQG_TRY(aqg_check_launch(ctx, "agg_kernel"));
    }
    // ---- dense ids ---------------------------------------------------------------------------------
    unsigned cgrid = aqg_grid(ctx, slots, 256, 1, 8);
    if (!(n && (use_part || use_wpart))) hipLaunchKernelGGL(collect_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, gt, occ);
    uint32_t fl[8] = {0, 0, 0, 0, 0, 0, 0, 0};        // [0] overflow, [1] occupied slots, [3] a row outside the sampled key ranges, [4], [5] diagnostics, [6] a value outside its packed field
    uint32_t G = 0;
    auto judge_flags = [&]() -> int {
        if (dense && fl[3]) { h->dense_exact = true; h->range_valid = false; return AQG_ERR_RANGE_MISS; }
        if (use_wpart && fl[6]) { h->no_pack = true; return AQG_ERR_RANGE_MISS; }      // a key outside the sampled range of its packed field: once more, unpacked
        if (use_wpart && fl[0]) {
            // a partition larger than LDS holds (fl[5] rows).  A little over: chance (a million partitions sized at mean + 6 sigma) --
            // ONE more try with another seed of the partition hash; far over, or over again: a tuple that dominates the input, which no
            // seed spreads -- the HBM table, same hint
            static const bool debug_flags = getenv("AQG_DEBUG_FLAGS") != nullptr;
            if (debug_flags) fprintf(stderr, "aqg: wide partition plan gave up: flags %u %u %u %u, partition %u holds %u rows (n %u, hint %u, seed %u)\n", fl[0], fl[1], fl[2], fl[3], fl[4], fl[5], n, hint, h->wide_seed);
            const uint32_t rcap = h->wide_rows ? h->wide_rows : aqg_partitionw_rows(ks, as, n, hint);
            if (!fl[5]) return AQG_ERR_OVERFLOW;     // no partition was too large: the record table (out_cap) was -- more groups than hinted, the caller grows the hint
            if (h->wide_seed == 0 && fl[5] <= rcap + rcap / 2) h->wide_seed = 0x5BD1E995u; else h->no_wide_part = true;
            return AQG_ERR_RANGE_MISS;
        }
        if ((use_part || use_wpart) && fl[6]) { h->no_pack = true; return AQG_ERR_RANGE_MISS; }     // a value outside the sampled range of its packed field: once more, unpacked
        if (fl[0]) return AQG_ERR_OVERFLOW;
        G = fl[1];
        if (small_rank && G > 4096) return AQG_ERR_OVERFLOW;
        if (use_lds && G > lds_group_cap && G > hint) return AQG_ERR_OVERFLOW;   // correct but slow (overflow rows took the HBM path): re-plan
        return AQG_OK;
    };
    auto read_flags = [&]() -> int {
        AQG_HIP(ctx, hipMemcpyAsync(fl, gt.flags, 32, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return judge_flags();
    };
    // The fast kernel or the fused star join with a small table (h2o Q1 / Q4, config 4): every kernel of the tail reads the group count from the device flags and
    // the outputs are sized by the table, so nothing on the host stands between collect and emit (a round trip there cost 26 us
    // of a 1.45 ms step).  The flag words -- final once collect has run -- are copied to pinned memory right here, behind
    // collect and in FRONT of the tail, and the host waits for that copy only: the call returns with first rows / rank / emit
    // still queued (stream-ordered, like every device result of this library), so the host's way to the next call overlaps them.
    // An overflow is noticed with the tail already queued on it: those kernels are bounded by the table and by `gmax`, their
    // results are discarded and the call re-plans as before.
    const bool defer = n && (fast || plan.sj) && small_rank && !dense && !use_part && !use_wpart;
    const uint32_t gupper = (uint32_t)(slots + 1 < 4096 ? slots + 1 : 4096);
    uint32_t* pinned_flags = nullptr;
    if (defer) {
        AQG_TRY(aqg_host_stage(ctx, 16, reinterpret_cast<void**>(&pinned_flags)));
        AQG_HIP(ctx, hipMemcpyAsync(pinned_flags, gt.flags, 16, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipEventRecord(ctx->ev_flags, ctx->stream));
    }
    if (!defer) AQG_TRY(read_flags());
    if ((defer || G) && n && fast) {
        // two launches: 32 workgroups over the first 32768 rows (where every group of an h2o-like column already shows up), then
        // the whole chip over the rest, whose workgroups leave at once when nothing is missing.  One launch of 256 workgroups
        // starts with 65536 lanes pushing atomicMin at ~100 addresses: 28-31 us on h2o Q1.
        const uint32_t* k0 = static_cast<const uint32_t*>(ks.col[0]);
        const uint32_t* k1 = fast_k64 && !fast_key8 ? static_cast<const uint32_t*>(ks.col[1]) : (const uint32_t*)nullptr;
        const uint32_t head_tiles = 32, head_rows = head_tiles * 1024;
        hipLaunchKernelGGL(first_rows_kernel, dim3(head_tiles), dim3(256), 0, ctx->stream, k0, k1, fast_key8 ? 1 : 0, 0u, n < head_rows ? n : head_rows, gt, (const uint32_t*)occ);
        if (n > head_rows) {
            unsigned fgrid = aqg_grid(ctx, (n - head_rows) / 4 + 1, 256, 1, 1);
            hipLaunchKernelGGL(first_rows_kernel, dim3(fgrid), dim3(256), 0, ctx->stream, k0, k1, fast_key8 ? 1 : 0, head_tiles, n, gt, (const uint32_t*)occ);
        }
        AQG_TRY(aqg_check_launch(ctx, "first_rows_kernel"));
    }
    // every row its own group: the result is a map of the input (emit_rows_kernel) -- nothing to rank or order
    const bool row_emit = rows_possible && !defer && G == n && (use_part || use_wpart);
    SortedParts sparts;
    if (row_emit) {
        h->plan_bits |= AQG_PLAN_ROW_EMIT;
    } else if (sorted_tail && G) {
        AQG_TRY(aqg_sorted_tail(ctx, gt, G, n, as.nacc, ks.wide != 0, &sparts));
    } else if (defer || G) {
        if (small_rank) {
            hipLaunchKernelGGL(rank_small_kernel, dim3(1), dim3(1024), 0, ctx->stream, gt, occ, gid_of_occ, slot_gid);
        } else {
            unsigned g1 = aqg_grid(ctx, G, 256, 1, 8);
            hipLaunchKernelGGL(bitmap_set_kernel, dim3(g1), dim3(256), 0, ctx->stream, gt, occ, bitmap, tile_mark);
            hipLaunchKernelGGL(bitmap_tile_kernel, dim3(ntiles), dim3(256), 0, ctx->stream, bitmap, nwords, word_prefix, tile_total, (const uint32_t*)tile_mark);
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, tile_total, ntiles);
            hipLaunchKernelGGL(rank_bitmap_kernel, dim3(g1), dim3(256), 0, ctx->stream, gt, occ, bitmap, word_prefix, tile_total, gid_of_occ, slot_gid);
            if (sparse_rank) hipLaunchKernelGGL(bitmap_clear_kernel, dim3(g1), dim3(256), 0, ctx->stream, gt, occ, bitmap);      // the context's bitmap is all zero again
        }
    }
    // ---- outputs --------------------------------------------------------------------------------------
    h->nkeys = ks.nkeys;
    size_t gcapn = defer ? gupper : (G ? G : 1);
    EmitSpec es;
    memset(&es, 0, sizeof es);
    es.nkeys = ks.nkeys;
    es.wide = ks.wide;
    for (int k = 0; k < ks.nkeys; ++k) {
        h->key_dt[k] = ks.dt[k];
        AQG_TRY(dev_realloc(ctx, &h->keys_out[k], &h->cap_keys[k], gcapn * 8));
        es.key_dt[k] = ks.dt[k]; es.key_shift[k] = ks.shift[k]; es.key_out[k] = h->keys_out[k]; es.key_col[k] = ks.col[k];
    }
    AQG_TRY(dev_realloc(ctx, (void**)&h->first_rows, &h->cap_first, gcapn * 4));
    AQG_TRY(dev_realloc(ctx, (void**)&h->counts, &h->cap_counts, gcapn * 4));
    es.first_out = h->first_rows;
    h->has_counts = plan.need_count && (!for_build || (use_part && n));
    es.count_out = h->has_counts ? h->counts : nullptr;
    es.nagg = plan.nagg;
    h->nagg = plan.nagg;
    for (int j = 0; j < plan.nagg; ++j) {
        es.agg[j] = plan.agg[j];
        h->res_dt[j] = aqg_reduce_out_dtype(plan.agg[j].op, plan.agg[j].dt);
        AQG_TRY(dev_realloc(ctx, &h->results[j], &h->cap_results[j], gcapn * 16));
        es.agg[j].out = h->results[j];
    }
    if (row_emit) {
        for (int k = 0; k < ks.nkeys; ++k) {
            const size_t bytes = (size_t)n * aqg_dtype_size(ks.dt[k]), nvec = bytes / 16;
            if (((uintptr_t)ks.col[k] & 15) == 0 && nvec && nvec <= 0x7FFFFFFFull * 256) {
                hipLaunchKernelGGL(copy_vec_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, ctx->stream, static_cast<const uint4*>(ks.col[k]), static_cast<uint4*>(h->keys_out[k]), nvec);
                if (bytes & 15) AQG_HIP(ctx, hipMemcpyAsync(static_cast<char*>(h->keys_out[k]) + nvec * 16, static_cast<const char*>(ks.col[k]) + nvec * 16, bytes & 15, hipMemcpyDeviceToDevice, ctx->stream));
            } else AQG_HIP(ctx, hipMemcpyAsync(h->keys_out[k], ks.col[k], bytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
        hipLaunchKernelGGL(emit_rows_kernel, dim3(aqg_grid(ctx, n, 256, 1, 8)), dim3(256), 0, ctx->stream, as, es, n);
        AQG_TRY(aqg_check_launch(ctx, "emit_rows_kernel"));
    } else if (sorted_tail && G) {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(&sorted_emit_kernel), sparts.lds));
        const unsigned per_cu = sparts.lds <= 80 * 1024 ? 2 : 1;
        const unsigned sg = sparts.nparts < 4u * per_cu * ctx->num_cu ? sparts.nparts : 4u * per_cu * ctx->num_cu;
        static const unsigned se_env = getenv("AQG_SORTED_EMIT_BLOCK") ? (unsigned)atoi(getenv("AQG_SORTED_EMIT_BLOCK")) : 1024u;
        const unsigned se_block = se_env >= 512 && se_env <= 1024 && se_env % 64 == 0 ? se_env : 1024u;      // (one lane per bitmap word of a 16384-row interval: 512 at least)
        hipLaunchKernelGGL(sorted_emit_kernel, dim3(sg), dim3(se_block), sparts.lds, ctx->stream, sparts, G, n, as.nacc, (int)gt.has_count, (int)(ks.wide != 0), es, gt.flags);
        AQG_TRY(aqg_check_launch(ctx, "sorted_emit_kernel"));
        // a partition that does not keep to the plan (more records or a longer row interval than LDS was sized for) is skipped by the kernel
        // and reported in flag word 7: the output would miss its rows, so the call waits for the word (calls of this size run for tens of
        // milliseconds) and, should it ever be set, runs once more through the bitmap tail
        uint32_t bad = 0;
        AQG_HIP(ctx, hipMemcpyAsync(&bad, gt.flags + 7, 4, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (bad) { h->no_sorted_tail = true; return AQG_ERR_RANGE_MISS; }
    } else if (defer || G) {
        unsigned eg = aqg_grid(ctx, defer ? gupper : G, 256, 1, 8);
        uint32_t* order = nullptr;
        if (ordered_emit && G >= (1u << 20)) {
            AQG_TRY(aqg_ws_get(ctx, slots, &order));
            hipLaunchKernelGGL(emit_order_kernel, dim3(eg), dim3(256), 0, ctx->stream, (const uint32_t*)gid_of_occ, G, order);
        }
        hipLaunchKernelGGL(emit_kernel, dim3(eg), dim3(256), 0, ctx->stream, gt, (const uint32_t*)occ, (const uint32_t*)gid_of_occ, es, (const uint32_t*)order, defer ? 4096u : 0u, (int)(n && (use_part || use_wpart)));
        AQG_TRY(aqg_check_launch(ctx, "emit_kernel"));
    }
    if (defer) {
        AQG_HIP(ctx, hipEventSynchronize(ctx->ev_flags));
        memcpy(fl, pinned_flags, 16);
        AQG_TRY(judge_flags());
        ctx->tail_in_flight = true;
    }
    h->build_assigned = false;
    if (for_build && use_part && n && G && lookup_build) {       // the id of every row through a key -> id table, in row order
        size_t c = h->reversemap ? h->cap_rows * 4 : 0;
        AQG_TRY(dev_realloc(ctx, (void**)&h->reversemap, &c, ((size_t)n + 4) * 4));
        h->cap_rows = c / 4;
        uint32_t* table;
        AQG_TRY(aqg_ws_get(ctx, (size_t)lk_D + 64, &table));
        AQG_HIP(ctx, hipMemsetAsync(table, 0xFF, (size_t)lk_D * 4, ctx->stream));
        hipLaunchKernelGGL(lookup_fill_kernel, dim3(aqg_grid(ctx, G, 256, 1, 8)), dim3(256), 0, ctx->stream, gt, (const uint32_t*)slot_gid, lk_min, lk_D, table, gt.flags + 8);
        hipLaunchKernelGGL(lookup_assign_kernel, dim3(aqg_grid(ctx, n / 4 + 1, 256, 1, 8)), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(ks.col[0]), n, lk_min, lk_D,
                           (const uint32_t*)table, h->reversemap, gt.flags + 8);
        AQG_TRY(aqg_check_launch(ctx, "lookup_assign_kernel"));
        uint32_t miss = 0;
        AQG_HIP(ctx, hipMemcpyAsync(&miss, gt.flags + 8, 4, hipMemcpyDeviceToHost, ctx->stream));
        AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (miss) { h->no_lookup_build = true; return AQG_ERR_RANGE_MISS; }      // a key outside the sampled domain: once more, through the routed form
        h->build_assigned = true;
        h->plan_bits |= AQG_PLAN_BUILD_PARTITIONED | AQG_PLAN_BUILD_LOOKUP;
    } else if (for_build && use_part && n && G && prows.valid) {       // the id of every row from the rows still lying partitioned in the workspace
        size_t c = h->reversemap ? h->cap_rows * 4 : 0;
        AQG_TRY(dev_realloc(ctx, (void**)&h->reversemap, &c, ((size_t)n + 4) * 4));
        h->cap_rows = c / 4;
        AQG_TRY(aqg_partition_assign(ctx, prows, gt, slot_gid, h->reversemap));
        h->build_assigned = true;
        h->plan_bits |= AQG_PLAN_BUILD_PARTITIONED;
    }
    h->ngroups = G;
    if (gt_out) *gt_out = gt;
    if (dense_out) { dense_out->used = dense; if (dense) dense_out->spec = dspec; }
    if (slot_gid_out) *slot_gid_out = slot_gid;
    if (occ_out) *occ_out = occ;
    return AQG_OK;
}

int run_with_retry(aqg_ctx* ctx, const KeySpec& ks, const Plan& plan, uint32_t n, uint32_t hint, bool for_build, aqg_groupby* h,
                   GTable* gt_out, uint32_t** slot_gid_out, uint32_t** occ_out = nullptr, DenseOut* dense_out = nullptr);

// No hint and a large input: count the distinct tuples of the first 2^20 rows (a group-by without aggregates over a sample: well
// under a millisecond) and size the plan from that, instead of discovering the cardinality by running -- and overflowing --
// one plan after the other over all the rows (1e9 rows, 1e7 groups, hint 0: ~600 ms of escalations before).
// Uniformly spread keys: d = G (1 - exp(-s / G)) distinct tuples among s sampled rows; solved for G.  An estimate that is too
// small only costs the usual re-plan; one that is too large picks a plan for more groups than there are (still exact).
// The sample: 1024 blocks of 1024 consecutive rows spread evenly over the table, gathered into columns of their own.  Two counts come out of it:
// d = the distinct tuples of the whole sample (a group-by without aggregates), and D2 = the sum over the blocks of the distinct tuples INSIDE each
// block (sample_block_distinct_kernel).  Keys spread at random: d = G (1 - exp(-s / G)), solved for G, as before.  Keys CLUSTERED -- a table sorted
// by its key, or arriving key by key -- show themselves by blocks that share no tuples (d ~ D2) although rows repeat inside the blocks (D2 < s):
// every run of equal keys is then seen about once per n / s rows, G ~ d n / s.  (With the first 2^20 rows as the sample, 1e9 rows sorted by a key
// of 1e7 values were estimated at 13,000 groups; the escalation behind that ended in the HBM table: 7.9 s for a 21 ms call.)
// Both counts from ONE kernel over the rows where they lie (block b = sample block b, 1024 consecutive rows from row (b * total) >> 10): the
// tuples go into an open-addressing table in HBM (2^21 8-byte slots for 2^20 rows; wide tuples by their 32-bit hash: an estimate) and into one of
// 2048 slots in LDS; the first of every tuple is counted.  One launch, one host round trip: ~50 us.  (Before: the sample gathered into columns of its
// own, a quadratic per-block distinct count -- 69 us -- and a count-only group-by through the one-level partition plan: ~0.3 ms with its three host
// round trips, a fifth of h2o Q1's first call at 1e9 rows.)
constexpr uint32_t SAMPLE_ROWS = 1u << 20, SAMPLE_SLOTS = 1u << 21;
__device__ inline uint32_t sample_mix(uint64_t k) { k ^= k >> 33; k *= 0xFF51AFD7ED558CCDull; k ^= k >> 33; k *= 0xC4CEB9FE1A85EC53ull; return (uint32_t)(k >> 32); }
__global__ void __launch_bounds__(1024) sample_distinct_kernel(KeySpec ks, uint32_t total, unsigned long long* __restrict__ table /* [SAMPLE_SLOTS], all ones */,
                                                               uint32_t* __restrict__ out /* [0] distinct in the sample, [1] sum of the blocks' distinct counts, [2] the all-ones tuple seen */) {
    constexpr unsigned long long NONE = ~0ull;
    __shared__ unsigned long long lkey[2048];
    __shared__ uint32_t cnt[2];
    lkey[threadIdx.x] = NONE; lkey[threadIdx.x + 1024] = NONE;
    if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
    __syncthreads();
    const size_t row = (size_t)(((uint64_t)blockIdx.x * total) >> 10) + threadIdx.x;
    const unsigned long long k = ks.wide ? (unsigned long long)hash_wide(ks, row) : (unsigned long long)pack_key(ks, row);
    bool first_here = false, first_all = false;
    if (k == NONE) atomicOr(&out[2], 1u);                       // (the empty mark itself: counted once by the host)
    else {
        const uint32_t h = sample_mix(k);
        for (uint32_t s = h & 2047u, p = 0; p < 2048; ++p, s = (s + 1) & 2047u) {
            const unsigned long long old = atomicCAS(&lkey[s], NONE, k);
            if (old == NONE) { first_here = true; break; }
            if (old == k) break;
        }
        if (first_here) {                                       // (only a block's first row of a tuple goes to the shared table)
            for (uint32_t s = (h >> 11) & (SAMPLE_SLOTS - 1), p = 0; p < SAMPLE_SLOTS; ++p, s = (s + 1) & (SAMPLE_SLOTS - 1)) {
                const unsigned long long old = atomicCAS(&table[s], NONE, k);
                if (old == NONE) { first_all = true; break; }
                if (old == k) break;
            }
        }
    }
    const uint64_t mh = __ballot(first_here), ma = __ballot(first_all);
    if (lane_id() == 0) { atomicAdd(&cnt[0], (uint32_t)__popcll(ma)); atomicAdd(&cnt[1], (uint32_t)__popcll(mh)); }
    __syncthreads();
    if (threadIdx.x < 2 && cnt[threadIdx.x]) atomicAdd(&out[threadIdx.x], cnt[threadIdx.x]);
}
uint64_t estimate_groups(aqg_ctx* ctx, const KeySpec& ks, uint32_t n) {
    const uint32_t s = SAMPLE_ROWS;
    const size_t need = (size_t)SAMPLE_SLOTS * 8 + 64;
    void* buf = nullptr;
    size_t cap = 0;
    buf = aqg_pool_take(ctx, need, &cap);
    if (!buf) { if (hipMalloc(&buf, need) != hipSuccess) { (void)hipGetLastError(); return 0; } cap = need; }
    unsigned long long* table = static_cast<unsigned long long*>(buf);
    uint32_t* dout = reinterpret_cast<uint32_t*>(static_cast<char*>(buf) + (size_t)SAMPLE_SLOTS * 8);
    uint32_t got[4] = {0, 0, 0, 0};
    bool ok = hipMemsetAsync(table, 0xFF, (size_t)SAMPLE_SLOTS * 8, ctx->stream) == hipSuccess && hipMemsetAsync(dout, 0, 16, ctx->stream) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(sample_distinct_kernel, dim3(1024), dim3(1024), 0, ctx->stream, ks, n, table, dout);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(got, dout, 16, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    uint64_t est = 0;
    if (ok) {
        const double d = (double)got[0] + (got[2] ? 1.0 : 0.0), sd = (double)s, D2 = (double)got[1] + (got[2] ? 1.0 : 0.0);
        if (d <= 0.5 * sd) est = (uint64_t)(d * 1.25) + 64;                 // the sample has seen (nearly) every group
        else if (d >= 0.999 * sd) est = n;                                  // (nearly) all distinct
        else {
            double lo = d, hi = 1e12;                                       // d / G = 1 - exp(-s / G), monotone in G
            for (int it = 0; it < 60; ++it) { double g = 0.5 * (lo + hi); if (g * (1.0 - exp(-sd / g)) < d) lo = g; else hi = g; }
            est = (uint64_t)(hi * 1.25) + 64;
        }
        if (D2 > 0 && d >= 0.8 * D2 && D2 <= 0.9 * sd) {                    // clustered keys: blocks share (nearly) no tuples, rows repeat inside them
            const uint64_t clustered = (uint64_t)(d * ((double)n / sd) * 1.1) + 64;
            if (clustered > est) est = clustered;
        }
        if (est > n) est = n;
    }
    aqg_pool_give(ctx, buf, cap);
    return est;
}
int run_with_retry(aqg_ctx* ctx, const KeySpec& ks, const Plan& plan, uint32_t n, uint32_t hint, bool for_build, aqg_groupby* h,
                   GTable* gt_out, uint32_t** slot_gid_out, uint32_t** occ_out, DenseOut* dense_out) {
    uint64_t cur = hint ? hint : (h->hint_used ? h->hint_used : 1024);
    if (!hint && !h->hint_used && n >= (1u << 22) && !plan.sj) { const uint64_t e = estimate_groups(ctx, ks, n); if (e > cur) cur = e; }
    for (int attempt = 0; attempt < 12; ++attempt) {
        if (cur > n && n) cur = n;
        int rc = run_agg(ctx, ks, plan, n, (uint32_t)cur, for_build, h, gt_out, slot_gid_out, occ_out, dense_out);
        // (exact key ranges now; or the wide-tuple plan with another hash seed, then without it: at most three repeats)
        for (int again = 0; again < 3 && rc == AQG_ERR_RANGE_MISS; ++again) rc = run_agg(ctx, ks, plan, n, (uint32_t)cur, for_build, h, gt_out, slot_gid_out, occ_out, dense_out);
        if (rc != AQG_ERR_OVERFLOW) { if (rc == AQG_OK) h->hint_used = (uint32_t)cur; return rc; }
        if (n && cur >= n) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "group-by: table overflow at full capacity");
        // (x16 -- but not past 2^25 in one step: beyond it packed keys leave the partition plans)
        cur = cur < (1ull << 25) && cur * 16 > (1ull << 25) ? (1ull << 25) : cur * 16;
    }
    return aqg_fail(ctx, AQG_ERR_OVERFLOW, "group-by: table overflow");
}

} // namespace

// ---- key columns that are not plain integers ------------------------------------------------------------------------------------
// The reference groups by tuple `==` (server/hasher.h:66-144 hashes, std::equal_to on the tuple).  Probed against the reference
// itself (oracle/ref_harness.cpp, tests/golden): dates compare their 4 bytes; times their 7 bytes of fields (the 8th is padding);
// timestamps date + time; 128-bit integers all 16 bytes; `const char*` keys are POINTERS (8-byte integers); floating keys compare
// by value -- 0.0 and -0.0 are one group (libstdc++ hashes both to 0, == holds) and every NaN is a group of its own (same hash,
// == never holds).  Here such columns are grouped through normalised integer columns: masked / split copies, canonical zero, and
// for NaNs one more hidden key column holding row + 1.
namespace {
__global__ void __launch_bounds__(256) norm_time_kernel(const uint64_t* __restrict__ src, uint32_t n, uint64_t* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = src[i] & 0x00FFFFFFFFFFFFFFull;
}
__global__ void __launch_bounds__(256) norm_timestamp_kernel(const uint32_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ date, uint64_t* __restrict__ time) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        date[i] = src[3 * i];
        time[i] = ((uint64_t)src[3 * i + 1] | ((uint64_t)src[3 * i + 2] << 32)) & 0x00FFFFFFFFFFFFFFull;
    }
}
__global__ void __launch_bounds__(256) norm_i128_kernel(const uint64_t* __restrict__ src, uint32_t n, uint64_t* __restrict__ lo, uint64_t* __restrict__ hi) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { lo[i] = src[2 * i]; hi[i] = src[2 * i + 1]; }
}
// flags[0] |= a NaN exists, flags[1] |= a negative zero exists
template <class B> __global__ void __launch_bounds__(256) fp_scan_kernel(const B* __restrict__ bits, uint32_t n, uint32_t* __restrict__ flags) {
    constexpr B SIGN = (B)1 << (sizeof(B) * 8 - 1), EXP = sizeof(B) == 4 ? (B)0x7F800000u : (B)0x7FF0000000000000ull;
    bool nan = false, nz = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const B b = bits[i];
        nan |= (b & ~SIGN) > EXP;
        nz |= b == SIGN;
    }
    if (__any(nan) && lane_id() == 0) atomicOr(&flags[0], 1u);
    if (__any(nz) && lane_id() == 0) atomicOr(&flags[1], 1u);
}
template <class B> __global__ void __launch_bounds__(256) fp_norm_kernel(const B* __restrict__ bits, uint32_t n, B* __restrict__ out, uint32_t* __restrict__ nanid) {
    constexpr B SIGN = (B)1 << (sizeof(B) * 8 - 1), EXP = sizeof(B) == 4 ? (B)0x7F800000u : (B)0x7FF0000000000000ull;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const B b = bits[i];
        out[i] = b == SIGN ? (B)0 : b;
        if (nanid) nanid[i] = (b & ~SIGN) > EXP ? (uint32_t)i + 1u : 0u;
    }
}
// out[g] = element first_rows[g] of a column of `esz`-byte elements
__global__ void __launch_bounds__(256) key_fetch_kernel(const unsigned char* __restrict__ col, int esz, const uint32_t* __restrict__ first_rows, uint32_t G, unsigned char* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)G * esz; i += (size_t)gridDim.x * 256) {
        const uint32_t g = (uint32_t)(i / esz), b = (uint32_t)(i - (size_t)g * esz);
        out[i] = col[(size_t)first_rows[g] * esz + b];
    }
}

size_t key_elem_size(int dt) {
    switch (dt) {
    case AQG_DATE: return 4;
    case AQG_TIME: return 8;
    case AQG_TIMESTAMP: return 12;
    case AQG_INT128: case AQG_UINT128: return 16;
    default: return aqg_dtype_size(dt);
    }
}
bool key_is_plain(int dt) { return (dt_is_num(dt) && !dt_is_fp(dt)) || dt == AQG_BOOL; }

// the integer columns (ndt / ncol, *nn of them) that stand for the caller's key columns; bookkeeping for aqg_groupby_keys in `h`
int normalize_keys(aqg_ctx* ctx, aqg_groupby* h, int nkeys, const int* dts, const void* const* keys, uint32_t n, int* nn, int* ndt, const void** ncol) {
    if (nkeys < 1 || nkeys > MAXKEYS) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: 1..8 key columns");
    bool all_plain = true;
    for (int k = 0; k < nkeys; ++k) all_plain = all_plain && key_is_plain(dts[k]);
    if (all_plain) { h->nuser = 0; *nn = nkeys; for (int k = 0; k < nkeys; ++k) { ndt[k] = dts[k]; ncol[k] = keys[k]; } return AQG_OK; }
    h->nuser = nkeys;
    int m = 0, nb = 0;
    auto buf = [&](size_t bytes, void** out) -> int { AQG_TRY(dev_realloc(ctx, &h->norm_buf[nb], &h->cap_norm[nb], bytes ? bytes : 16)); *out = h->norm_buf[nb++]; return AQG_OK; };
    auto push = [&](int dt, const void* col) -> int { if (m >= MAXKEYS) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: the key columns normalise to more than 8 integer columns"); ndt[m] = dt; ncol[m] = col; ++m; return AQG_OK; };
    const unsigned grid = aqg_grid(ctx, n, 256, 4, 16);
    uint32_t* flags = nullptr;
    for (int k = 0; k < nkeys; ++k) {
        const int dt = dts[k];
        h->user_dt[k] = dt; h->user_col[k] = keys[k]; h->user_norm[k] = -1;
        if (!keys[k] && n) return aqg_fail(ctx, AQG_ERR_ARG, "group-by: null key column");
        if (key_is_plain(dt)) { h->user_norm[k] = m; AQG_TRY(push(dt, keys[k])); continue; }
        void *a = nullptr, *b = nullptr;
        switch (dt) {
        case AQG_DATE: AQG_TRY(push(AQG_UINT32, keys[k])); break;
        case AQG_TIME:
            AQG_TRY(buf((size_t)n * 8, &a));
            if (n) hipLaunchKernelGGL(norm_time_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint64_t*>(keys[k]), n, static_cast<uint64_t*>(a));
            AQG_TRY(push(AQG_UINT64, a));
            break;
        case AQG_TIMESTAMP:
            AQG_TRY(buf((size_t)n * 4, &a)); AQG_TRY(buf((size_t)n * 8, &b));
            if (n) hipLaunchKernelGGL(norm_timestamp_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(keys[k]), n, static_cast<uint32_t*>(a), static_cast<uint64_t*>(b));
            AQG_TRY(push(AQG_UINT32, a)); AQG_TRY(push(AQG_UINT64, b));
            break;
        case AQG_INT128: case AQG_UINT128:
            AQG_TRY(buf((size_t)n * 8, &a)); AQG_TRY(buf((size_t)n * 8, &b));
            if (n) hipLaunchKernelGGL(norm_i128_kernel, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint64_t*>(keys[k]), n, static_cast<uint64_t*>(a), static_cast<uint64_t*>(b));
            AQG_TRY(push(AQG_UINT64, a)); AQG_TRY(push(AQG_UINT64, b));
            break;
        case AQG_FLOAT: case AQG_DOUBLE: {
            const bool f32 = dt == AQG_FLOAT;
            if (!flags) { AQG_TRY(aqg_ws_reset(ctx)); AQG_TRY(aqg_ws_ensure(ctx, 4096)); AQG_TRY(aqg_ws_get(ctx, 16, &flags)); }
            uint32_t fl[2] = {0, 0};
            AQG_HIP(ctx, hipMemsetAsync(flags, 0, 8, ctx->stream));
            if (n) {
                if (f32) hipLaunchKernelGGL(fp_scan_kernel<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(keys[k]), n, flags);
                else hipLaunchKernelGGL(fp_scan_kernel<uint64_t>, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint64_t*>(keys[k]), n, flags);
            }
            AQG_HIP(ctx, hipMemcpyAsync(fl, flags, 8, hipMemcpyDeviceToHost, ctx->stream));
            AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!fl[0] && !fl[1]) { AQG_TRY(push(f32 ? AQG_UINT32 : AQG_UINT64, keys[k])); break; }      // the bit patterns are the values
            AQG_TRY(buf((size_t)n * (f32 ? 4 : 8), &a));
            if (fl[0]) AQG_TRY(buf((size_t)n * 4, &b));
            if (f32) hipLaunchKernelGGL(fp_norm_kernel<uint32_t>, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint32_t*>(keys[k]), n, static_cast<uint32_t*>(a), static_cast<uint32_t*>(b));
            else hipLaunchKernelGGL(fp_norm_kernel<uint64_t>, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint64_t*>(keys[k]), n, static_cast<uint64_t*>(a), static_cast<uint32_t*>(b));
            AQG_TRY(push(f32 ? AQG_UINT32 : AQG_UINT64, a));
            if (fl[0]) AQG_TRY(push(AQG_UINT32, b));                         // every NaN row its own group
        } break;
        default: return aqg_fail(ctx, AQG_ERR_DTYPE, "group-by: key dtype (strings are grouped through aqg_str_encode codes)");
        }
    }
    AQG_TRY(aqg_check_launch(ctx, "key normalisation"));
    *nn = m;
    return AQG_OK;
}
} // namespace

extern "C" {

void aqg_groupby_destroy(aqg_groupby* g) {
    if (!g) return;
    // result columns go back to the context's pool (the next handle takes them without a hipMalloc / hipFree pair); what does not
    // fit there is freed, and hipFree waits for the device
    aqg_ctx* ctx = g->ctx;
    for (int k = 0; k < MAXKEYS; ++k) aqg_pool_give(ctx, g->keys_out[k], g->cap_keys[k]);
    for (int j = 0; j < MAXAGG; ++j) aqg_pool_give(ctx, g->results[j], g->cap_results[j]);
    aqg_pool_give(ctx, g->first_rows, g->cap_first);
    aqg_pool_give(ctx, g->counts, g->cap_counts);
    aqg_pool_give(ctx, g->reversemap, g->cap_rows * 4);
    if (g->scratch) aqg_groupby_destroy(g->scratch);
    if (g->scratch2) aqg_groupby_destroy(g->scratch2);
    aqg_pool_give(ctx, g->flat_off, g->cap_flat_off);
    aqg_pool_give(ctx, g->flat_heads, g->cap_flat_heads);
    aqg_pool_give(ctx, g->flat_short, g->cap_flat_short);
    aqg_pool_give(ctx, g->flat_gid, g->cap_flat_gid);
    if (g->first_rows64) hipFree(g->first_rows64);
    for (int i = 0; i < 2 * MAXKEYS; ++i) if (g->norm_buf[i]) hipFree(g->norm_buf[i]);
    if (g->xkeys) hipFree(g->xkeys);
    if (g->xvals) hipFree(g->xvals);
    delete g;
}
uint32_t aqg_groupby_ngroups(const aqg_groupby* g) { return g ? g->ngroups : 0; }
uint32_t aqg_groupby_nrows(const aqg_groupby* g) { return g ? g->n : 0; }
const uint32_t* aqg_groupby_reversemap(const aqg_groupby* g) { return g && g->has_reversemap ? g->reversemap : nullptr; }
const uint32_t* aqg_groupby_counts(const aqg_groupby* g) { return g && g->has_counts ? g->counts : nullptr; }
const uint32_t* aqg_groupby_first_rows(const aqg_groupby* g) { return g && !g->sharded ? g->first_rows : nullptr; }
uint32_t aqg_groupby_plan(const aqg_groupby* g) { return g ? g->plan_bits : 0; }
const void* aqg_groupby_agg_result(const aqg_groupby* g, int j) { return g && j >= 0 && j < g->nagg ? g->results[j] : nullptr; }

int aqg_groupby_keys(aqg_groupby* g, int k, void* out_dev) {
    if (!g || k < 0 || k >= (g->nuser ? g->nuser : g->nkeys) || !out_dev) return AQG_ERR_ARG;
    aqg_ctx* ctx = g->ctx;
    if (!g->ngroups) return AQG_OK;
    if (g->nuser && g->user_norm[k] < 0) {        // not a plain integer column: the key of a group is the caller's element at its first row
        const int esz = (int)key_elem_size(g->user_dt[k]);
        hipLaunchKernelGGL(key_fetch_kernel, dim3(aqg_grid(ctx, (uint64_t)g->ngroups * esz, 256, 4, 8)), dim3(256), 0, ctx->stream,
                           static_cast<const unsigned char*>(g->user_col[k]), esz, (const uint32_t*)g->first_rows, g->ngroups, static_cast<unsigned char*>(out_dev));
        return aqg_check_launch(ctx, "key_fetch_kernel");
    }
    const int kk = g->nuser ? g->user_norm[k] : k;
    const size_t kesz = g->sharded && g->key_esz[kk] ? (size_t)g->key_esz[kk] : aqg_dtype_size(g->key_dt[kk]);
    AQG_HIP(ctx, hipMemcpyAsync(out_dev, g->keys_out[kk], (size_t)g->ngroups * kesz, hipMemcpyDeviceToDevice, ctx->stream));
    return AQG_OK;
}

int aqg_groupby_agg(aqg_ctx* ctx, int nkeys, const int* key_dtypes, const void* const* keys, int naggs, const int* ops,
                    const int* val_dtypes, const void* const* vals, uint32_t n, uint32_t max_groups_hint, aqg_groupby** out) {
    if (!ctx || !out || !key_dtypes || !keys) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_agg: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_groupby_agg");
    Plan plan;
    AQG_TRY(make_plan(ctx, naggs, ops, val_dtypes, vals, n, &plan));
    aqg_groupby* h = *out ? *out : new aqg_groupby();
    h->ctx = ctx; h->n = n; h->has_reversemap = false; h->sharded = false;
    ctx->tail_in_flight = false;
    KeySpec ks;
    int nn = 0, ndt[MAXKEYS];
    const void* ncol[MAXKEYS];
    int rc = normalize_keys(ctx, h, nkeys, key_dtypes, keys, n, &nn, ndt, ncol);
    if (rc == AQG_OK) rc = make_keyspec(ctx, nn, ndt, ncol, n, &ks);
    if (rc == AQG_OK) rc = run_with_retry(ctx, ks, plan, n, max_groups_hint, false, h, nullptr, nullptr);
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    if (!ctx->tail_in_flight) AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (small tables: the group count is known, the tail is stream-ordered)
    *out = h;
    return AQG_OK;
}

int aqg_join_groupby_sum(aqg_ctx* ctx, int key_dtype, const void* dim_keys, int dim_val_dtype, const void* dim_vals, uint32_t nb,
                         const void* fact_fk, int group_key_dtype, const void* group_keys, int val_dtype, const void* fact_vals, uint32_t n,
                         uint32_t max_groups_hint, aqg_groupby** out) {
    if (!ctx || !out || ((!dim_keys || !dim_vals) && nb) || ((!fact_fk || !group_keys || !fact_vals) && n))
        return aqg_fail(ctx, AQG_ERR_ARG, "aqg_join_groupby_sum: bad argument");
    auto i32 = [](int t) { return t == AQG_INT32 || t == AQG_UINT32; };
    if (!i32(key_dtype) || !i32(dim_val_dtype) || !i32(group_key_dtype) || !i32(val_dtype))
        return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_join_groupby_sum: 4-byte integer columns only");
    if (nb > 4096) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_join_groupby_sum: the dimension side must fit LDS (<= 4096 rows)");
    AQG_CHECK_ROWS(ctx, n, "aqg_join_groupby_sum");
    StarJoin sj;
    sj.dim_keys = static_cast<const uint32_t*>(dim_keys); sj.dim_vals = static_cast<const uint32_t*>(dim_vals); sj.nb = nb;
    sj.dcap = next_pow2((uint64_t)(nb < 8 ? 8 : nb) * 2);
    sj.fk = static_cast<const uint32_t*>(fact_fk); sj.vals = static_cast<const uint32_t*>(fact_vals);
    sj.val_signed = val_dtype == AQG_INT32; sj.dim_signed = dim_val_dtype == AQG_INT32;
    KeySpec ks;
    const void* kcols[1] = {group_keys};
    AQG_TRY(make_keyspec(ctx, 1, &group_key_dtype, kcols, n, &ks));
    // the sum of the exact products is an 8-byte-integer SUM: two accumulators (low / high halves), emitted as 128 bits;
    // unsigned x unsigned products are summed as unsigned
    Plan plan;
    const int op = AQG_RED_SUM, pdt = (sj.val_signed || sj.dim_signed) ? AQG_INT64 : AQG_UINT64;
    const void* pv[1] = {fact_vals};
    AQG_TRY(make_plan(ctx, 1, &op, &pdt, pv, n, &plan));
    plan.sj = &sj;
    aqg_groupby* h = *out ? *out : new aqg_groupby();
    h->ctx = ctx; h->n = n; h->has_reversemap = false;
    ctx->tail_in_flight = false;
    int rc = run_with_retry(ctx, ks, plan, n, max_groups_hint, false, h, nullptr, nullptr);
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    if (!ctx->tail_in_flight) AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));   // (small tables: the group count is known, the tail is stream-ordered)
    *out = h;
    return AQG_OK;
}

int aqg_groupby_pack(aqg_groupby* g, int agg_index, uint32_t gmax, int64_t* out_dev) {
    if (!g || !out_dev) return AQG_ERR_ARG;
    aqg_ctx* ctx = g->ctx;
    if (g->nkeys != 1) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_pack: one key column");
    if (agg_index < 0 || agg_index >= g->nagg) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_pack: aggregate index");
    if (g->ngroups > gmax) return aqg_fail(ctx, AQG_ERR_OVERFLOW, "aqg_groupby_pack: more groups than gmax");
    const int rdt = g->res_dt[agg_index];
    if (rdt == AQG_FLOAT || rdt == AQG_DOUBLE) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_groupby_pack: integer aggregates only (floating partials: gather the result columns)");
    hipLaunchKernelGGL(pack_kernel, dim3(aqg_grid(ctx, (uint64_t)g->ngroups + 1, 256, 1, 4)), dim3(256), 0, ctx->stream, (const void*)g->keys_out[0], g->key_dt[0],
                       (const void*)g->results[agg_index], rdt, g->ngroups, reinterpret_cast<long long*>(out_dev));
    return aqg_check_launch(ctx, "pack_kernel");
}

int aqg_groupby_merge_packed(aqg_ctx* ctx, const int64_t* gathered_dev, uint32_t world, uint32_t gmax, int key_dtype, int op, aqg_groupby** out) {
    if (!ctx || !gathered_dev || !out || world == 0 || world > 64) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_merge_packed: bad argument (1..64 shards)");
    if (!(op == AQG_RED_SUM || op == AQG_RED_MIN || op == AQG_RED_MAX || op == AQG_RED_COUNT)) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_merge_packed: SUM / COUNT / MIN / MAX");
    // the concatenation (offsets from the shard headers, on the device) and a plain group-by over it; the host learns the row
    // count from ONE small copy (it used to fetch every shard header: `world` copies and their latency in front of the merge)
    const uint64_t cap_rows = (uint64_t)world * gmax;
    if (cap_rows > AQG_MAX_ROWS) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_merge_packed: world x gmax too large");
    aqg_groupby* h = *out ? *out : new aqg_groupby();
    h->ctx = ctx;
    const bool int_key = key_dtype == AQG_INT8 || key_dtype == AQG_INT16 || key_dtype == AQG_INT32 || key_dtype == AQG_INT64 || key_dtype == AQG_UINT8 ||
                         key_dtype == AQG_UINT16 || key_dtype == AQG_UINT32 || key_dtype == AQG_UINT64 || key_dtype == AQG_BOOL;
    static const bool small_merge_off = getenv("AQG_DISABLE_SMALL_MERGE") != nullptr;        // A/B measurements only
    if (cap_rows <= MERGE_ROWS && int_key && !small_merge_off) {
        // ---- a few small shard tables: one workgroup does the whole merge ------------------------------------------------
        const int mop = op == AQG_RED_COUNT ? AQG_RED_SUM : op;
        int rc = AQG_OK;
        rc = dev_realloc(ctx, &h->keys_out[0], &h->cap_keys[0], (size_t)MERGE_ROWS * 8);
        if (rc == AQG_OK) rc = dev_realloc(ctx, (void**)&h->first_rows, &h->cap_first, (size_t)MERGE_ROWS * 4);
        if (rc == AQG_OK) rc = dev_realloc(ctx, (void**)&h->counts, &h->cap_counts, (size_t)MERGE_ROWS * 4);
        if (rc == AQG_OK) rc = dev_realloc(ctx, &h->results[0], &h->cap_results[0], (size_t)MERGE_ROWS * 16);
        if (rc == AQG_OK) rc = aqg_ws_reset(ctx);
        if (rc == AQG_OK) rc = aqg_ws_ensure(ctx, 4096);
        uint32_t* info = nullptr;
        if (rc == AQG_OK) rc = aqg_ws_get(ctx, 4, &info);
        if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
        const size_t lds = (size_t)(MERGE_CAP + 1) * 28 + (size_t)MERGE_ROWS * 4 + 64;
        rc = aqg_allow_lds(ctx, reinterpret_cast<const void*>(&merge_small_kernel), lds);
        if (rc == AQG_OK) {
            hipLaunchKernelGGL(merge_small_kernel, dim3(1), dim3(1024), lds, ctx->stream, reinterpret_cast<const long long*>(gathered_dev), world, gmax, key_dtype, mop,
                               h->keys_out[0], h->results[0], h->first_rows, info);
            rc = aqg_check_launch(ctx, "merge_small_kernel");
        }
        uint32_t ih[3] = {0, 0, 0};
        if (rc == AQG_OK) rc = aqg_d2h(ctx, ih, info, 12);
        if (rc == AQG_OK && ih[1]) rc = aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_merge_packed: corrupt shard header");
        if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
        h->n = ih[2]; h->ngroups = ih[0];
        h->nkeys = 1; h->key_dt[0] = key_dtype;
        h->has_counts = false; h->has_reversemap = false;
        h->nagg = 1; h->res_dt[0] = aqg_reduce_out_dtype(mop, AQG_INT64);
        *out = h;
        return AQG_OK;
    }
    int rc = dev_realloc(ctx, &h->xkeys, &h->cap_xkeys, (cap_rows + 2) * 8);
    if (rc == AQG_OK) rc = dev_realloc(ctx, &h->xvals, &h->cap_xvals, (cap_rows + 2) * 8);
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    uint32_t* total_dev = reinterpret_cast<uint32_t*>(static_cast<char*>(h->xvals) + (cap_rows + 1) * 8);     // the spare word behind the values
    hipLaunchKernelGGL(unpack_kernel, dim3(world), dim3(256), 0, ctx->stream, reinterpret_cast<const long long*>(gathered_dev), world, gmax, key_dtype, h->xkeys,
                       static_cast<long long*>(h->xvals), total_dev);
    uint32_t th[2] = {0, 0};
    rc = aqg_d2h(ctx, th, total_dev, 8);
    if (rc == AQG_OK && th[1]) rc = aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_merge_packed: corrupt shard header");
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    const uint64_t total = th[0];
    const void* kc[1] = {h->xkeys};
    const void* vc[1] = {h->xvals};
    const int mop = op == AQG_RED_COUNT ? AQG_RED_SUM : op, vdt = AQG_INT64;
    rc = aqg_groupby_agg(ctx, 1, &key_dtype, kc, 1, &mop, &vdt, vc, (uint32_t)total, gmax, &h);
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    *out = h;
    return AQG_OK;
}

int aqg_groupby_build(aqg_ctx* ctx, int nkeys, const int* key_dtypes, const void* const* keys, uint32_t n,
                      uint32_t max_groups_hint, aqg_groupby** out) {
    if (!ctx || !out || !key_dtypes || !keys) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_groupby_build: bad argument");
    AQG_CHECK_ROWS(ctx, n, "aqg_groupby_build");
    Plan plan;
    memset(&plan, 0, sizeof plan);
    aqg_groupby* h = *out ? *out : new aqg_groupby();
    h->ctx = ctx; h->n = n; h->sharded = false;
    h->flat_valid = h->flat_gid_valid = false; h->flat_short_w = 0;
    GTable gt; uint32_t* slot_gid = nullptr; uint32_t* occ_dev = nullptr;
    DenseOut dn;
    dn.used = false;
    KeySpec ks;
    int nn = 0, ndt[MAXKEYS];
    const void* ncol[MAXKEYS];
    int rc = normalize_keys(ctx, h, nkeys, key_dtypes, keys, n, &nn, ndt, ncol);
    if (rc == AQG_OK) rc = make_keyspec(ctx, nn, ndt, ncol, n, &ks);
    if (rc == AQG_OK) rc = run_with_retry(ctx, ks, plan, n, max_groups_hint, true, h, &gt, &slot_gid, &occ_dev, &dn);
    if (rc == AQG_OK) {
        size_t c = h->reversemap ? h->cap_rows * 4 : 0;
        rc = dev_realloc(ctx, (void**)&h->reversemap, &c, ((size_t)n + 4) * 4);
        if (rc == AQG_OK) h->cap_rows = c / 4;
    }
    if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    uint32_t G = h->ngroups;
    if (n && !h->build_assigned) {
        hipMemsetAsync(h->counts, 0, (size_t)(G ? G : 1) * 4, ctx->stream);
        unsigned grid = aqg_grid(ctx, n / 4 + 1, 256, 2, 8);
        if (dn.used) {                      // direct-indexed table: the dense id of a row is slot_gid[idx(row)]
            rc = aqg_dense_assign(ctx, ks, dn.spec, slot_gid, n, G, h->reversemap, h->counts);
            if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
        } else if (G <= 2048 && !ks.wide) {
            const uint32_t mcap = next_pow2((uint64_t)G * 2 + 2);
            size_t lds = (((size_t)G * 4 + 15) & ~(size_t)15) + (size_t)mcap * 12 + 16;
            hipLaunchKernelGGL((assign_kernel<true, true>), dim3(grid), dim3(256), lds, ctx->stream, ks, gt, slot_gid, occ_dev, n, G, mcap, h->reversemap, h->counts);
        } else if (G <= 36000) {            // group counts in an LDS histogram (up to 144 KB) instead of 1e9 global atomics
            size_t lds = (size_t)G * 4 + 16;
            hipFuncSetAttribute(reinterpret_cast<const void*>(&assign_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (lds > 20 * 1024) { const unsigned per_cu = (unsigned)((160 * 1024) / (lds + 1024)); grid = aqg_grid(ctx, n / 4 + 1, 256, 2, per_cu ? per_cu : 1); }
            hipLaunchKernelGGL((assign_kernel<true, false>), dim3(grid), dim3(256), lds, ctx->stream, ks, gt, slot_gid, occ_dev, n, G, 0u, h->reversemap, h->counts);
        } else {
            hipLaunchKernelGGL((assign_kernel<false, false>), dim3(grid), dim3(256), 0, ctx->stream, ks, gt, slot_gid, occ_dev, n, G, 0u, h->reversemap, h->counts);
        }
        rc = aqg_check_launch(ctx, "assign_kernel");
        if (rc != AQG_OK) { if (!*out) aqg_groupby_destroy(h); return rc; }
    }
    h->has_counts = true; h->has_reversemap = true;
    AQG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = h;
    return AQG_OK;
}


namespace {
// corr(x, y) of every group from its five sums (server/aggregations.h:401-406): all of them __int128 in the reference (InnerType there is
// the Coercion STRUCT, so GetLongType<InnerType> is __int128 whatever the inputs are), len * s wraps in 128 bits, FPType = double
__global__ void __launch_bounds__(256) corr_final_kernel(const aqg_i128* __restrict__ sx, const aqg_i128* __restrict__ sx2, const aqg_i128* __restrict__ sy,
                                                        const aqg_i128* __restrict__ sy2, const aqg_i128* __restrict__ sxy, const uint32_t* __restrict__ counts,
                                                        uint32_t G, double* __restrict__ out) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) {
        const aqg_i128 len = i128_from_u64(counts[g]);
        const double a = i128_to_double(mul_128(len, sxy[g])) - i128_to_double(mul_128(sx[g], sy[g]));
        const double b = i128_to_double(mul_128(len, sx2[g])) - i128_to_double(mul_128(sx[g], sx[g]));
        const double c = i128_to_double(mul_128(len, sy2[g])) - i128_to_double(mul_128(sy[g], sy[g]));
        out[g] = a / sqrt(b * c);
    }
}
__global__ void __launch_bounds__(256) take_rows_kernel(const uint64_t* __restrict__ acc_rows, uint32_t G, uint32_t* __restrict__ rows) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < G; g += gridDim.x * blockDim.x) rows[g] = (uint32_t)acc_rows[g];
}
} // namespace

// the core of aqg_grouped_reduce: groups by a column of dense group ids (the build's reversemap, or the group index of every position
// of the flat layout -- segscan.hip) through the ordinary group-by plans; ids appear in first-occurrence order, so group g is result g
int aqg_grouped_reduce_keyed(aqg_ctx* ctx, aqg_groupby* g, const uint32_t* gid_col, int op, int t, const void* x, void* out_dev) {
    const uint32_t G = g->ngroups, n = g->n;
    // beyond the LDS tables: the build's ids are dense and its group sizes known -- partitioned on the id, direct-indexed (partition1.hip)
    static const bool gid_off = getenv("AQG_DISABLE_GID_REDUCE") != nullptr;       // A/B measurements only
    static const uint32_t gid_min = getenv("AQG_GID_MIN") ? (uint32_t)atoi(getenv("AQG_GID_MIN")) : (1u << 16);       // (measured again in round 3, with the value inside the id word: 6.5 against 7.5 ms at 1e5 groups, equal for values that do not pack)
    if (!gid_off && gid_col == g->reversemap && g->has_counts && G > gid_min && n >= (1u << 22)) {       // (up to ~3e6 groups the one-level hashed plan is as fast: 8.0-8.4 ms against 9.0 per 1e9 rows; 1e7 groups: 17 against 9)
        const uint32_t* off = aqg_groupby_offsets(g);
        if (off) {
            const int rc = aqg_gid_reduce(ctx, gid_col, off, g->counts, n, G, op, t, x, out_dev);
            if (rc != AQG_ERR_DTYPE) { if (rc == AQG_OK) g->plan_bits = AQG_PLAN_GID_PARTITION; return rc; }
        }
    }
    const int kdt = AQG_UINT32;
    const void* kcol = gid_col;
    KeySpec ks;
    AQG_TRY(make_keyspec(ctx, 1, &kdt, &kcol, n, &ks));
    ks.range_known = 1; ks.range_lo = 0; ks.range_hi = (long long)G - 1;      // dense group ids
    if (!g->scratch) g->scratch = new aqg_groupby();
    aqg_groupby* h = g->scratch;
    h->ctx = ctx; h->n = n; h->has_reversemap = false;
    Plan plan;
    AQG_TRY(make_plan(ctx, 1, &op, &t, &x, n, &plan));
    AQG_TRY(run_with_retry(ctx, ks, plan, n, G, false, h, nullptr, nullptr));
    if (h->ngroups != G) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_reduce: group ids are not dense");
    g->plan_bits = h->plan_bits;
    AQG_HIP(ctx, hipMemcpyAsync(out_dev, h->results[0], (size_t)G * aqg_dtype_size(aqg_reduce_out_dtype(op, t)), hipMemcpyDeviceToDevice, ctx->stream));
    return AQG_OK;
}

// out[g] = op(col[vecs[g]]) for every group in one pass (generated loop engine/ast.py:722-789).
// The group id column (reversemap) is itself a dense first-occurrence key, so grouping by it
// reproduces the group order; the value column is read once.  vecs[g] is in DESCENDING row order
// (hasher.h:192-196), hence first(col[vecs[g]]) is the LAST row of the group and last(...) its first row.
int aqg_grouped_reduce(aqg_ctx* ctx, const aqg_groupby* gc, int op, int t, const void* x, void* out_dev) {
    aqg_groupby* g = const_cast<aqg_groupby*>(gc);
    if (!ctx || !g || (!x && g->n) || !out_dev) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_reduce: bad argument");
    if (!g->has_reversemap) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_reduce: handle has no reversemap (use aqg_groupby_build)");
    if (!dt_is_num(t)) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_grouped_reduce: value dtype");
    const uint32_t G = g->ngroups, n = g->n;
    if (G == 0) return AQG_OK;
    const int kdt = AQG_UINT32;
    const void* kcol = g->reversemap;
    if (op == AQG_RED_LAST) return aqg_gather(ctx, t, x, g->first_rows, G, out_dev);
    KeySpec ks;
    AQG_TRY(make_keyspec(ctx, 1, &kdt, &kcol, n, &ks));
    ks.range_known = 1; ks.range_lo = 0; ks.range_hi = (long long)G - 1;      // dense group ids
    if (!g->scratch) g->scratch = new aqg_groupby();
    aqg_groupby* h = g->scratch;
    h->ctx = ctx; h->n = n; h->has_reversemap = false;
    if (op == AQG_RED_FIRST) {
        Plan plan;
        memset(&plan, 0, sizeof plan);
        plan.nagg = 1;
        plan.agg[0].op = AQG_RED_MAX; plan.agg[0].dt = AQG_UINT64; plan.agg[0].acc1 = plan.agg[0].acc2 = plan.agg[0].acc3 = -1;
        plan.agg[0].acc0 = add_acc(&plan, ACC_MAX, AQG_NONE, nullptr, 0);
        AQG_TRY(run_with_retry(ctx, ks, plan, n, G, false, h, nullptr, nullptr));
        uint32_t* rows = nullptr;
        AQG_TRY(aqg_ws_reset(ctx));
        AQG_TRY(aqg_ws_get(ctx, G, &rows));
        hipLaunchKernelGGL(take_rows_kernel, dim3(aqg_grid(ctx, G, 256, 1, 8)), dim3(256), 0, ctx->stream, (const uint64_t*)h->results[0], G, rows);
        return aqg_gather(ctx, t, x, rows, G, out_dev);
    }
    return aqg_grouped_reduce_keyed(ctx, g, g->reversemap, op, t, x, out_dev);
}

// out[g] = corr(x[vecs[g]], y[vecs[g]]) for every group (h2o Q9 `pow(corr(v1, v2), 2) BY id2, id4`, benchmark/h2o/groupby.sql:20; the generated
// loop engine/ast.py:749-784 emits `corr(v1[val], v2[val])`): the product column x * y (evaluated in the C++ type of the operands like the
// reference's `x[i] * y[i]`, aggregations.h:397), then ONE grouped pass with five accumulators -- sum x, sum x*x, sum y, sum y*y, sum xy --
// and the reference's formula per group.  Integer columns of up to four bytes (every sum then fits a 64-bit accumulator exactly).
int aqg_grouped_corr(aqg_ctx* ctx, aqg_groupby* g, int tx, const void* x, int ty, const void* y, double* out_dev) {
    if (!ctx || !g || ((!x || !y) && g->n) || !out_dev) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_corr: bad argument");
    if (!g->has_reversemap || !g->has_counts) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_corr: handle has no reversemap (use aqg_groupby_build)");
    auto small_int = [](int dt) { return dt == AQG_INT8 || dt == AQG_INT16 || dt == AQG_INT32 || dt == AQG_UINT8 || dt == AQG_UINT16 || dt == AQG_UINT32; };
    if (!small_int(tx) || !small_int(ty)) return aqg_fail(ctx, AQG_ERR_DTYPE, "aqg_grouped_corr: integer columns of up to four bytes (others: aqg_corr per group)");
    const uint32_t G = g->ngroups, n = g->n;
    if (G == 0) return AQG_OK;
    const int pt = (tx == AQG_UINT32 || ty == AQG_UINT32) ? AQG_UINT32 : AQG_INT32;      // usual arithmetic conversions of the two operands
    size_t cap = 0;
    void* xy = aqg_pool_take(ctx, (size_t)n * 4 + 64, &cap);
    if (!xy) {
        cap = (size_t)n * 4 + 64;
        hipError_t e = hipMalloc(&xy, cap);
        if (e != hipSuccess) { (void)hipGetLastError(); return aqg_fail(ctx, AQG_ERR_NOMEM, "aqg_grouped_corr: product column"); }
    }
    int rc = aqg_ewise(ctx, AQG_OP_MUL, AQG_VEC_VEC, tx, x, ty, y, pt, xy, n);
    if (rc == AQG_OK) {
        const int kdt = AQG_UINT32;
        const void* kcol = g->reversemap;
        KeySpec ks;
        rc = make_keyspec(ctx, 1, &kdt, &kcol, n, &ks);
        ks.range_known = 1; ks.range_lo = 0; ks.range_hi = (long long)G - 1;      // dense group ids
        if (!g->scratch) g->scratch = new aqg_groupby();
        aqg_groupby* h = g->scratch;
        h->ctx = ctx; h->n = n; h->has_reversemap = false;
        // up to 3072 groups: two passes through the fast LDS plan (at most four accumulators each) -- {sum x, sum x*x, sum y, sum y*y} over
        // the two columns, then {sum xy} over the product column.  Beyond (h2o Q9: 1e4 groups): five single-accumulator passes -- four
        // accumulators per slot push a 1e4-slot table out of LDS (dense plan, three passes over the rows: 19.5 ms per 1e9 rows) while one
        // accumulator streams at 1.45 ms per pass
        void* sums = nullptr;                      // [5][G] 128-bit sums, copied out of the scratch handle pass by pass
        size_t sums_cap = 0;
        sums = aqg_pool_take(ctx, (size_t)G * 80 + 64, &sums_cap);
        if (!sums) { sums_cap = (size_t)G * 80 + 64; if (hipMalloc(&sums, sums_cap) != hipSuccess) { (void)hipGetLastError(); sums = nullptr; rc = aqg_fail(ctx, AQG_ERR_NOMEM, "aqg_grouped_corr: sums"); } }
        const int ops5[5] = {AQG_RED_SUM, AQG_RED_SUMSQ, AQG_RED_SUM, AQG_RED_SUMSQ, AQG_RED_SUM};
        const int dts5[5] = {tx, tx, ty, ty, pt};
        const void* vals5[5] = {x, x, y, y, xy};
        auto slot = [&](int j) { return static_cast<char*>(sums) + (size_t)j * G * 16; };
        auto pass = [&](int first, int count) -> int {
            Plan plan;
            AQG_TRY(make_plan(ctx, count, ops5 + first, dts5 + first, vals5 + first, n, &plan));
            AQG_TRY(run_with_retry(ctx, ks, plan, n, G, false, h, nullptr, nullptr));
            if (h->ngroups != G) return aqg_fail(ctx, AQG_ERR_ARG, "aqg_grouped_corr: group ids are not dense");
            for (int j = 0; j < count; ++j) AQG_HIP(ctx, hipMemcpyAsync(slot(first + j), h->results[j], (size_t)G * 16, hipMemcpyDeviceToDevice, ctx->stream));
            return AQG_OK;
        };
        if (rc == AQG_OK) {
            if (G <= 3072) { rc = pass(0, 4); if (rc == AQG_OK) rc = pass(4, 1); }
            else for (int j = 0; j < 5 && rc == AQG_OK; ++j) rc = pass(j, 1);
        }
        if (rc == AQG_OK) {
            hipLaunchKernelGGL(corr_final_kernel, dim3(aqg_grid(ctx, G, 256, 1, 8)), dim3(256), 0, ctx->stream, (const aqg_i128*)slot(0), (const aqg_i128*)slot(1),
                               (const aqg_i128*)slot(2), (const aqg_i128*)slot(3), (const aqg_i128*)slot(4), g->counts, G, out_dev);
            rc = aqg_check_launch(ctx, "corr_final_kernel");
        }
        if (sums) aqg_pool_give(ctx, sums, sums_cap);
    }
    aqg_pool_give(ctx, xy, cap);           // (stream-ordered reuse: every later user of the buffer runs on this stream)
    return rc;
}

} // extern "C"

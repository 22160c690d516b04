// partition1_agg.hip -- the per-partition aggregation kernels of the partition plans (partition1.hip) and their launchers: the hashed table
// with dense ids (p1_agg_kernel), with the accumulators inside the table (p1_agg_slot_kernel), and the direct-indexed form over range
// partitions (p1_agg_direct_kernel).  Reference: AQHashTable's build (server/hasher.h:146-199) and the generated per-group loop
// (engine/ast.py:722-789), for the groups of ONE partition at a time.
#include "partition1_int.hpp"

namespace {

// ---- aggregate each partition in LDS ---------------------------------------------------------------------------------------------
struct AggIn {
    const void* col[MAXACC]; int esz[MAXACC];    // partitioned value arrays (4- or 8-byte elements); null: the row id, or a packed field
    int packed[MAXACC]; uint32_t pshift[MAXACC], pmask[MAXACC], pmin[MAXACC];   // the operand is a field of the key word: ((key >> pshift) & pmask) + pmin
    uint32_t kclear;                             // the packed fields' bits of the key word (0: none)
};
constexpr uint32_t ID_PENDING = 0xFFFFu, ID_OVER = 0xFFFEu;


constexpr int AR = 4;      // consecutive rows per lane and step (one 16-byte load per 4-byte plane)
// LDS: acc u64[NACC][gmax] | keytab K[cap] | first u32[gmax] | count u32[gmax] (need_count) | idtab u16[cap]
// dense id 0 is reserved for the group whose packed key equals the empty mark.
// V8: some value plane has 8-byte elements (then every value travels through the loop as 64 bits)
template <int NACC, bool K64, bool V8>
__global__ void __launch_bounds__(SB) p1_agg_kernel(const void* __restrict__ rkeys, const uint32_t* __restrict__ rrows, AccSpec as, AggIn in, AggOps ops,
                                                    const uint32_t* __restrict__ pstart, uint32_t pstride, uint32_t NB, uint32_t ntotal, uint32_t cap, uint32_t gmax, int need_count,
                                                    GTable out, uint32_t out_cap, uint32_t* __restrict__ part_base /* null, or [2 NB]: {first record, records} of every partition */) {
    using K = key_t_<K64>;
    using VT = std::conditional_t<V8, uint64_t, uint32_t>;
    constexpr int NA = NACC ? NACC : 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint64_t* lacc = reinterpret_cast<uint64_t*>(smem_raw);                       // [NACC][gmax]
    K* ktab = reinterpret_cast<K*>(lacc + (size_t)NACC * gmax);                   // [cap]
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(ktab + cap);                   // [gmax]
    uint32_t* lcount = lfirst + gmax;                                             // [gmax] (only when need_count)
    uint16_t* idtab = reinterpret_cast<uint16_t*>(lcount + (need_count ? gmax : 0));   // [cap]
    __shared__ uint32_t lused, lemit, gbase;
    const K EMPTYK = empty_key<K64>();
    struct Batch { K key[AR]; uint32_t row[AR]; VT v[NA][AR]; };
    // A step = SB * AR consecutive rows of the partition, AR per lane.  load_full: a step that lies wholly inside the planes -- vector
    // loads, no branch, no clamp.  It is the ONLY form used for the prefetch of the next step: a loader with two code paths (or a
    // conditional call) makes hipcc wait for the loads right behind them -- the paths meet in the same registers -- and the
    // "prefetch" then overlaps nothing (seen in the ISA: vmcnt(0) ten instructions behind the loads; 5.5 ms per 1e9 rows of Q5).
    auto load_full = [&](uint32_t i0, Batch& t) {
        const uint32_t o = i0 + threadIdx.x * AR;
        __builtin_memcpy(t.key, static_cast<const K*>(rkeys) + o, sizeof t.key);
        __builtin_memcpy(t.row, rrows + o, sizeof t.row);
        _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
            if (!in.col[a]) continue;                            // row-index operand (the carried row id) or a field of the key word: taken at the use (a copy here would wait for the load)
            if (!V8 || in.esz[a] == 4) {
                uint32_t w[AR];
                __builtin_memcpy(w, static_cast<const uint32_t*>(in.col[a]) + o, sizeof w);
                _Pragma("unroll") for (int q = 0; q < AR; ++q) t.v[a][q] = w[q];
            } else {
                if constexpr (V8) __builtin_memcpy(t.v[a], static_cast<const uint64_t*>(in.col[a]) + o, sizeof(uint64_t) * AR);
            }
        }
    };
    // the last, partial step of a partition: row by row from clamped indices (every load is issued; the caller masks rows >= e)
    auto load_edge = [&](uint32_t i0, uint32_t e, Batch& t) {
        const uint32_t o = i0 + threadIdx.x * AR;
        _Pragma("unroll") for (int q = 0; q < AR; ++q) {
            const uint32_t i = o + q < e ? o + q : e - 1;
            t.key[q] = static_cast<const K*>(rkeys)[i]; t.row[q] = rrows[i];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                if (!in.col[a]) continue;
                if (!V8 || in.esz[a] == 4) t.v[a][q] = static_cast<const uint32_t*>(in.col[a])[i];
                else if constexpr (V8) t.v[a][q] = static_cast<const uint64_t*>(in.col[a])[i];
            }
        }
    };
    for (uint32_t part = blockIdx.x; part < NB; part += gridDim.x) {
        const uint32_t b = pstart[(size_t)part * pstride];
        const uint32_t e = part + 1 < NB ? pstart[(size_t)(part + 1) * pstride] : ntotal;
        if (b == e) continue;
        constexpr uint32_t STEP = SB * AR;
        const uint32_t nfull = (e - b) / STEP, nsteps = nfull + ((e - b) % STEP ? 1u : 0u);
        // where a prefetch may always read a whole step: the last full step of this partition, or (a partition shorter than a step)
        // any step inside the planes -- what it fetches then is never used
        const uint32_t safe_last = nfull ? b + (nfull - 1) * STEP : (b + STEP <= ntotal ? b : ntotal - STEP);
        Batch cur;
        load_full(nfull ? b : safe_last, cur);                 // in flight while the tables are cleared
        for (uint32_t s = threadIdx.x; s < cap; s += SB) { ktab[s] = EMPTYK; idtab[s] = (uint16_t)ID_PENDING; }
        for (uint32_t g = threadIdx.x; g < gmax; g += SB) {
            lfirst[g] = NOROW;
            if (need_count) lcount[g] = 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * gmax + g] = acc_init(as.kind[a]);
        }
        if (threadIdx.x == 0) { lused = 1; lemit = 0; }
        __syncthreads();
        uint32_t i0 = b;
        for (uint32_t st = 0; st < nsteps; ++st) {
            if (st >= nfull) load_edge(i0, e, cur);            // (the last, partial step: nothing was prefetched for it)
            Batch nxt;
            { const uint32_t inext = i0 + STEP; load_full(inext <= safe_last && st + 1 < nfull ? inext : safe_last, nxt); }   // in flight while this step is aggregated
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t o = i0 + threadIdx.x * AR;
            uint32_t slot[AR];
            K w[AR];
            K raw[AR];                                                    // the key word as it travelled (value fields included)
            if constexpr (!K64) {
#pragma unroll
                for (int q = 0; q < AR; ++q) { raw[q] = cur.key[q]; cur.key[q] &= ~in.kclear; }
            }
#pragma unroll
            for (int q = 0; q < AR; ++q) { slot[q] = __umulhi(key_hash<K64>(cur.key[q]) * NB, cap); w[q] = ktab[slot[q]]; }   // AR probes in flight
            uint32_t pend = 0, special = 0;
#pragma unroll
            for (int q = 0; q < AR; ++q) {
                if (!(o + q < e)) slot[q] = FAIL;
                else if (cur.key[q] == EMPTYK) special |= 1u << q;
                else if (w[q] != cur.key[q]) pend |= 1u << q;
            }
            // rows that missed on their first probe walk their probe sequences together: one LDS round trip per step
            for (uint32_t step = 0; pend && step <= cap; ++step) {
#pragma unroll
                for (int q = 0; q < AR; ++q) {
                    if (!(pend & (1u << q))) continue;
                    K c = w[q];
                    if (c == EMPTYK) {
                        if constexpr (K64) c = atomicCAS(reinterpret_cast<unsigned long long*>(&ktab[slot[q]]), (unsigned long long)EMPTYK, (unsigned long long)cur.key[q]);
                        else c = atomicCAS(&ktab[slot[q]], EMPTYK, cur.key[q]);
                        if (c == EMPTYK) {
                            const uint32_t id = atomicAdd(&lused, 1u);
                            idtab[slot[q]] = (uint16_t)(id < gmax ? id : ID_OVER);
                            c = cur.key[q];
                        }
                    }
                    if (c == cur.key[q]) { pend &= ~(1u << q); continue; }
                    slot[q] = slot[q] + 1 == cap ? 0 : slot[q] + 1;
                }
#pragma unroll
                for (int q = 0; q < AR; ++q) if (pend & (1u << q)) w[q] = ktab[slot[q]];
            }
            uint32_t id[AR];
            bool ok[AR];
#pragma unroll
            for (int q = 0; q < AR; ++q) {
                if (special & (1u << q)) { id[q] = 0; ok[q] = true; continue; }
                if (slot[q] == FAIL || (pend & (1u << q))) { id[q] = 0; ok[q] = false; if (slot[q] != FAIL) out.flags[0] = 1; continue; }
                const volatile uint16_t* ip = idtab + slot[q];
                uint32_t v = *ip;
                while (v == ID_PENDING) { __builtin_amdgcn_s_sleep(1); v = *ip; }   // the inserting lane (of another wavefront) is about to publish it
                ok[q] = v != ID_OVER;
                id[q] = ok[q] ? v : 0;
                if (!ok[q]) out.flags[0] = 1;                                      // more groups than the dense arrays hold: the host re-plans
            }
#pragma unroll
            for (int q = 0; q < AR; ++q) if (ok[q] && cur.row[q] < lfirst[id[q]]) atomicMin(&lfirst[id[q]], cur.row[q]);
            if (need_count) {
#pragma unroll
                for (int q = 0; q < AR; ++q) if (ok[q]) atomicAdd(&lcount[id[q]], 1u);
            }
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                uint64_t* acc = lacc + (size_t)a * gmax;
#define AQG_ROWS(expr) _Pragma("unroll") for (int q = 0; q < AR; ++q) if (ok[q]) { VT x = in.col[a] ? cur.v[a][q] : (VT)cur.row[q]; \
                if constexpr (!K64) { if (in.packed[a]) x = (VT)((((uint32_t)raw[q] >> in.pshift[a]) & in.pmask[a]) + in.pmin[a]); } (void)x; expr; } break
                switch (ops.opc[a]) {
                case OPC_ADDI_I32: AQG_ROWS(atomicAdd(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(long long)(int32_t)(uint32_t)x));
                case OPC_ADDI_U32: AQG_ROWS(atomicAdd(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(uint32_t)x));
                case OPC_ADDF_F32: AQG_ROWS(atomicAdd(reinterpret_cast<double*>(acc + id[q]), (double)__uint_as_float((uint32_t)x)));
                case OPC_ADDF_F64: AQG_ROWS(atomicAdd(reinterpret_cast<double*>(acc + id[q]), __builtin_bit_cast(double, (uint64_t)x)));
                case OPC_MIN_I32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_i((int32_t)(uint32_t)x)));
                case OPC_MAX_I32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_i((int32_t)(uint32_t)x)));
                case OPC_MIN_U32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(uint32_t)x));
                case OPC_MAX_U32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(uint32_t)x));
                case OPC_MIN_F32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_f((double)__uint_as_float((uint32_t)x))));
                case OPC_MAX_F32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_f((double)__uint_as_float((uint32_t)x))));
                default: AQG_ROWS(acc_apply(acc + id[q], as.kind[a], val_operand_bits(as.dt[a] == AQG_NONE ? AQG_UINT32 : as.dt[a], (uint64_t)x, as.kind[a], as.square[a], as.part[a])));
                }
#undef AQG_ROWS
            }
            cur = nxt;
            i0 += STEP;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t used = (lused < gmax ? lused : gmax) - 1 + (lfirst[0] != NOROW ? 1u : 0u);
            gbase = atomicAdd(&out.flags[1], used);
            if (part_base) { part_base[2 * (size_t)part] = gbase; part_base[2 * (size_t)part + 1] = used; }
        }
        __syncthreads();
        for (uint32_t s = threadIdx.x; s <= cap; s += SB) {
            uint32_t id; uint64_t k;
            if (s == cap) { if (lfirst[0] == NOROW) continue; id = 0; k = K64 ? EMPTY64 : (uint64_t)EMPTY32; }
            else { if (ktab[s] == EMPTYK) continue; id = idtab[s]; k = (uint64_t)ktab[s]; if (id >= gmax) continue; }
            const uint32_t g = gbase + atomicAdd(&lemit, 1u);
            if (g >= out_cap) { out.flags[0] = 1; continue; }
            *out.key_p(g) = k;
            *out.first_p(g) = lfirst[id];
            *out.count_p(g) = need_count ? lcount[id] : 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) *out.acc_p(a, g) = lacc[(size_t)a * gmax + id];
        }
        __syncthreads();
    }
}


// ---- the same over a DENSE key domain: no key table at all -------------------------------------------------------------------------
// When the key is one 4-byte integer column whose values fill their range (h2o id3 / id6: 1 .. 1e7), the two-level plan bins the rows by
// RANGE -- bin = umulhi(key - kmin, M), order-preserving, so a partition owns a contiguous piece of the domain -- and the aggregation is
// direct-indexed: acc[key - first key of the partition].  No probe, no compare-and-swap, no dense-id indirection: every LDS operation of a
// row is a fire-and-forget atomic, nothing in the row loop waits for LDS (the hashed kernel above: three dependent LDS round trips per
// row, ~1000 instructions per 256 rows; h2o Q5 at 1e9 rows: 5.3 ms for 12 GB).  Rows beyond the end of a partial step and keys outside
// the partition's piece (only possible when the sampled range missed a value: flagged, the call repeats hashed) go to one dummy entry.
// LDS: acc u64[NACC][W + 1] | first u32[W + 1] | count u32[W + 1] (need_count); entry W is the dummy.
struct DirectSpec { uint32_t M, kmin, D, W; uint32_t* miss; };
template <int NACC, bool V8>
__global__ void __launch_bounds__(SB) p1_agg_direct_kernel(const uint32_t* __restrict__ rkeys, const uint32_t* __restrict__ rrows, AccSpec as, AggIn in, AggOps ops,
                                                           const uint32_t* __restrict__ pstart, uint32_t pstride, uint32_t NB, uint32_t ntotal, DirectSpec ds, int need_count,
                                                           GTable out, uint32_t out_cap) {
    using VT = std::conditional_t<V8, uint64_t, uint32_t>;
    constexpr int NA = NACC ? NACC : 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t W1 = ds.W + 1;
    uint64_t* lacc = reinterpret_cast<uint64_t*>(smem_raw);                       // [NACC][W1]
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(lacc + (size_t)NACC * W1);     // [W1]
    uint32_t* lcount = lfirst + W1;                                               // [W1] (only when need_count)
    __shared__ uint32_t lused, lemit, gbase;
    struct Batch { uint32_t key[AR]; uint32_t row[AR]; VT v[NA][AR]; };
    auto load_full = [&](uint32_t i0, Batch& t) {                                 // (as in p1_agg_kernel: the only form the prefetch uses)
        const uint32_t o = i0 + threadIdx.x * AR;
        __builtin_memcpy(t.key, rkeys + o, sizeof t.key);
        __builtin_memcpy(t.row, rrows + o, sizeof t.row);
        _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
            if (!in.col[a]) continue;
            if (!V8 || in.esz[a] == 4) {
                uint32_t w[AR];
                __builtin_memcpy(w, static_cast<const uint32_t*>(in.col[a]) + o, sizeof w);
                _Pragma("unroll") for (int q = 0; q < AR; ++q) t.v[a][q] = w[q];
            } else {
                if constexpr (V8) __builtin_memcpy(t.v[a], static_cast<const uint64_t*>(in.col[a]) + o, sizeof(uint64_t) * AR);
            }
        }
    };
    auto load_edge = [&](uint32_t i0, uint32_t e, Batch& t) {
        const uint32_t o = i0 + threadIdx.x * AR;
        _Pragma("unroll") for (int q = 0; q < AR; ++q) {
            const uint32_t i = o + q < e ? o + q : e - 1;
            t.key[q] = rkeys[i]; t.row[q] = rrows[i];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                if (!in.col[a]) continue;
                if (!V8 || in.esz[a] == 4) t.v[a][q] = static_cast<const uint32_t*>(in.col[a])[i];
                else if constexpr (V8) t.v[a][q] = static_cast<const uint64_t*>(in.col[a])[i];
            }
        }
    };
    for (uint32_t part = blockIdx.x; part < NB; part += gridDim.x) {
        const uint32_t b = pstart[(size_t)part * pstride];
        const uint32_t e = part + 1 < NB ? pstart[(size_t)(part + 1) * pstride] : ntotal;
        if (b == e) continue;
        // the piece of the domain this partition owns: x in [lo, hi), lo = the smallest x with umulhi(x, M) >= part
        uint64_t lo64 = (((uint64_t)part << 32) + ds.M - 1) / ds.M, hi64 = ((((uint64_t)part + 1) << 32) + ds.M - 1) / ds.M;
        if (hi64 > ds.D) hi64 = ds.D;
        if (lo64 > hi64) lo64 = hi64;
        const uint32_t lo = (uint32_t)lo64;
        uint32_t width = (uint32_t)(hi64 - lo64);
        if (width > ds.W) { if (threadIdx.x == 0) out.flags[0] = 1; continue; }   // (the host sized W for every piece)
        constexpr uint32_t STEP = SB * AR;
        const uint32_t nfull = (e - b) / STEP, nsteps = nfull + ((e - b) % STEP ? 1u : 0u);
        const uint32_t safe_last = nfull ? b + (nfull - 1) * STEP : (b + STEP <= ntotal ? b : ntotal - STEP);
        Batch cur;
        load_full(nfull ? b : safe_last, cur);                 // in flight while the arrays are cleared
        for (uint32_t g = threadIdx.x; g < W1; g += SB) {
            if (g >= width && g != ds.W) continue;
            lfirst[g] = NOROW;
            if (need_count) lcount[g] = 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * W1 + g] = acc_init(as.kind[a]);
        }
        if (threadIdx.x == 0) { lused = 0; lemit = 0; }
        __syncthreads();
        const uint32_t base = ds.kmin + lo;                    // key of entry 0
        uint32_t miss = 0;
        uint32_t i0 = b;
        for (uint32_t st = 0; st < nsteps; ++st) {
            const bool edge = st >= nfull;
            if (edge) load_edge(i0, e, cur);
            Batch nxt;
            { const uint32_t inext = i0 + STEP; load_full(inext <= safe_last && st + 1 < nfull ? inext : safe_last, nxt); }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t o = i0 + threadIdx.x * AR;
            uint32_t id[AR];
#pragma unroll
            for (int q = 0; q < AR; ++q) {
                const uint32_t x = (cur.key[q] & ~in.kclear) - base;
                const bool live = !edge || o + q < e;
                miss |= live && x >= width ? 1u : 0u;
                id[q] = live && x < width ? x : ds.W;
            }
#pragma unroll
            for (int q = 0; q < AR; ++q) atomicMin(&lfirst[id[q]], cur.row[q]);
            if (need_count) {
#pragma unroll
                for (int q = 0; q < AR; ++q) atomicAdd(&lcount[id[q]], 1u);
            }
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                uint64_t* acc = lacc + (size_t)a * W1;
                VT x[AR];
                if (in.packed[a]) { _Pragma("unroll") for (int q = 0; q < AR; ++q) x[q] = (VT)(((cur.key[q] >> in.pshift[a]) & in.pmask[a]) + in.pmin[a]); }
                else if (in.col[a]) { _Pragma("unroll") for (int q = 0; q < AR; ++q) x[q] = cur.v[a][q]; }
                else { _Pragma("unroll") for (int q = 0; q < AR; ++q) x[q] = (VT)cur.row[q]; }
#define AQG_ROWS(expr) _Pragma("unroll") for (int q = 0; q < AR; ++q) { expr; } break
                switch (ops.opc[a]) {
                case OPC_ADDI_I32: AQG_ROWS(atomicAdd(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(long long)(int32_t)(uint32_t)x[q]));
                case OPC_ADDI_U32: AQG_ROWS(atomicAdd(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(uint32_t)x[q]));
                case OPC_ADDF_F32: AQG_ROWS(atomicAdd(reinterpret_cast<double*>(acc + id[q]), (double)__uint_as_float((uint32_t)x[q])));
                case OPC_ADDF_F64: AQG_ROWS(atomicAdd(reinterpret_cast<double*>(acc + id[q]), __builtin_bit_cast(double, (uint64_t)x[q])));
                case OPC_MIN_I32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_i((int32_t)(uint32_t)x[q])));
                case OPC_MAX_I32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_i((int32_t)(uint32_t)x[q])));
                case OPC_MIN_U32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(uint32_t)x[q]));
                case OPC_MAX_U32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)(uint32_t)x[q]));
                case OPC_MIN_F32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_f((double)__uint_as_float((uint32_t)x[q]))));
                case OPC_MAX_F32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + id[q]), (unsigned long long)map_f((double)__uint_as_float((uint32_t)x[q]))));
                default: AQG_ROWS(acc_apply(acc + id[q], as.kind[a], val_operand_bits(as.dt[a] == AQG_NONE ? AQG_UINT32 : as.dt[a], (uint64_t)x[q], as.kind[a], as.square[a], as.part[a])));
                }
#undef AQG_ROWS
            }
            cur = nxt;
            i0 += STEP;
        }
        if (miss) *ds.miss = 1u;
        __syncthreads();
        // the entries that saw a row become records, reserved with one global atomic per partition
        uint32_t mine = 0;
        for (uint32_t j = threadIdx.x; j < width; j += SB) mine += lfirst[j] != NOROW ? 1u : 0u;
        mine = wave_reduce(mine, OpAdd{});
        if (lane_id() == 0 && mine) atomicAdd(&lused, mine);
        __syncthreads();
        if (threadIdx.x == 0) gbase = atomicAdd(&out.flags[1], lused);
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < width; j += SB) {
            if (lfirst[j] == NOROW) continue;
            const uint32_t g = gbase + atomicAdd(&lemit, 1u);
            if (g >= out_cap) { out.flags[0] = 1; continue; }
            *out.key_p(g) = (uint64_t)(base + j);
            *out.first_p(g) = lfirst[j];
            *out.count_p(g) = need_count ? lcount[j] : 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) *out.acc_p(a, g) = lacc[(size_t)a * W1 + j];
        }
        __syncthreads();
    }
}

// ---- hashed, accumulators indexed by the SLOT ---------------------------------------------------------------------------------------
// p1_agg_kernel above keeps its accumulators dense (a 2-byte id per slot, accumulators per id) so that the table's slack does not
// multiply them; the price is the chain probe -> id -> first row -> atomics, three dependent LDS round trips per row, a spin on ids not
// yet published and a counter every insertion passes through.  Where the groups of a partition still fit with the accumulators INSIDE
// the table (key | first row | count | accumulators per slot at load 0.6: h2o Q5's three sums up to ~1e7 groups in 4096 partitions), a
// row needs ONE dependent round trip -- its probe -- and everything behind it is a fire-and-forget atomic on the slot found, as in the
// direct-indexed kernel.  Slot `cap` belongs to the key that equals the empty mark, slot `cap + 1` takes the masked rows.
template <int NACC, bool K64, bool V8>
__global__ void __launch_bounds__(SB) p1_agg_slot_kernel(const void* __restrict__ rkeys, const uint32_t* __restrict__ rrows, AccSpec as, AggIn in, AggOps ops,
                                                         const uint32_t* __restrict__ pstart, uint32_t pstride, uint32_t NB, uint32_t ntotal, uint32_t cap, int need_count,
                                                         GTable out, uint32_t out_cap, uint32_t* __restrict__ part_base) {
    using K = key_t_<K64>;
    using VT = std::conditional_t<V8, uint64_t, uint32_t>;
    constexpr int NA = NACC ? NACC : 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t C2 = cap + 2;
    uint64_t* lacc = reinterpret_cast<uint64_t*>(smem_raw);                       // [NACC][C2]
    K* ktab = reinterpret_cast<K*>(lacc + (size_t)NACC * C2);                     // [C2] (8-byte keys: behind the accumulators, aligned)
    uint32_t* lfirst = reinterpret_cast<uint32_t*>(ktab + C2);                    // [C2]
    uint32_t* lcount = lfirst + C2;                                               // [C2] (only when need_count)
    __shared__ uint32_t lused, lemit, gbase, lins;
    const K EMPTYK = empty_key<K64>();
    struct Batch { K key[AR]; uint32_t row[AR]; VT v[NA][AR]; };
    auto load_full = [&](uint32_t i0, Batch& t) {                                 // (as in p1_agg_kernel: the only form the prefetch uses)
        const uint32_t o = i0 + threadIdx.x * AR;
        __builtin_memcpy(t.key, static_cast<const K*>(rkeys) + o, sizeof t.key);
        __builtin_memcpy(t.row, rrows + o, sizeof t.row);
        _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
            if (!in.col[a]) continue;
            if (!V8 || in.esz[a] == 4) {
                uint32_t w[AR];
                __builtin_memcpy(w, static_cast<const uint32_t*>(in.col[a]) + o, sizeof w);
                _Pragma("unroll") for (int q = 0; q < AR; ++q) t.v[a][q] = w[q];
            } else {
                if constexpr (V8) __builtin_memcpy(t.v[a], static_cast<const uint64_t*>(in.col[a]) + o, sizeof(uint64_t) * AR);
            }
        }
    };
    auto load_edge = [&](uint32_t i0, uint32_t e, Batch& t) {
        const uint32_t o = i0 + threadIdx.x * AR;
        _Pragma("unroll") for (int q = 0; q < AR; ++q) {
            const uint32_t i = o + q < e ? o + q : e - 1;
            t.key[q] = static_cast<const K*>(rkeys)[i]; t.row[q] = rrows[i];
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                if (!in.col[a]) continue;
                if (!V8 || in.esz[a] == 4) t.v[a][q] = static_cast<const uint32_t*>(in.col[a])[i];
                else if constexpr (V8) t.v[a][q] = static_cast<const uint64_t*>(in.col[a])[i];
            }
        }
    };
    const uint32_t limit = cap - (cap >> 3);                   // more keys than this in one partition: the host re-plans (probe chains grow without bound towards a full table)
    for (uint32_t part = blockIdx.x; part < NB; part += gridDim.x) {
        const uint32_t b = pstart[(size_t)part * pstride];
        const uint32_t e = part + 1 < NB ? pstart[(size_t)(part + 1) * pstride] : ntotal;
        if (b == e) continue;
        constexpr uint32_t STEP = SB * AR;
        const uint32_t nfull = (e - b) / STEP, nsteps = nfull + ((e - b) % STEP ? 1u : 0u);
        const uint32_t safe_last = nfull ? b + (nfull - 1) * STEP : (b + STEP <= ntotal ? b : ntotal - STEP);
        Batch cur;
        load_full(nfull ? b : safe_last, cur);                 // in flight while the table is cleared
        for (uint32_t g = threadIdx.x; g < C2; g += SB) {
            ktab[g] = EMPTYK;
            lfirst[g] = NOROW;
            if (need_count) lcount[g] = 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) lacc[(size_t)a * C2 + g] = acc_init(as.kind[a]);
        }
        if (threadIdx.x == 0) { lused = 0; lemit = 0; lins = 0; }
        __syncthreads();
        uint32_t i0 = b;
        for (uint32_t st = 0; st < nsteps; ++st) {
            const bool edge = st >= nfull;
            if (edge) load_edge(i0, e, cur);
            Batch nxt;
            { const uint32_t inext = i0 + STEP; load_full(inext <= safe_last && st + 1 < nfull ? inext : safe_last, nxt); }
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t o = i0 + threadIdx.x * AR;
            uint32_t slot[AR];
            K key[AR], w[AR];
#pragma unroll
            for (int q = 0; q < AR; ++q) {
                key[q] = cur.key[q];
                if constexpr (!K64) key[q] &= ~in.kclear;
                slot[q] = __umulhi(key_hash<K64>(key[q]) * NB, cap);
                w[q] = ktab[slot[q]];                                      // AR probes in flight
            }
            uint32_t pend = 0;
#pragma unroll
            for (int q = 0; q < AR; ++q) {
                if (edge && !(o + q < e)) slot[q] = cap + 1;               // (masked)
                else if (key[q] == EMPTYK) slot[q] = cap;                  // the key that doubles as the empty mark
                else if (w[q] != key[q]) pend |= 1u << q;
            }
            // rows that missed on their first probe walk their probe sequences together: one LDS round trip per step
            for (uint32_t step = 0; pend && step <= cap; ++step) {
#pragma unroll
                for (int q = 0; q < AR; ++q) {
                    if (!(pend & (1u << q))) continue;
                    K c = w[q];
                    if (c == EMPTYK) {
                        if constexpr (K64) c = atomicCAS(reinterpret_cast<unsigned long long*>(&ktab[slot[q]]), (unsigned long long)EMPTYK, (unsigned long long)key[q]);
                        else c = atomicCAS(&ktab[slot[q]], EMPTYK, key[q]);
                        if (c == EMPTYK) { c = key[q]; if (atomicAdd(&lins, 1u) >= limit) out.flags[0] = 1; }
                    }
                    if (c == key[q]) { pend &= ~(1u << q); continue; }
                    slot[q] = slot[q] + 1 == cap ? 0 : slot[q] + 1;
                }
#pragma unroll
                for (int q = 0; q < AR; ++q) if (pend & (1u << q)) w[q] = ktab[slot[q]];
            }
            if (pend) {                                                    // a full table: the host re-plans
                out.flags[0] = 1;
#pragma unroll
                for (int q = 0; q < AR; ++q) if (pend & (1u << q)) slot[q] = cap + 1;
            }
#pragma unroll
            for (int q = 0; q < AR; ++q) atomicMin(&lfirst[slot[q]], cur.row[q]);
            if (need_count) {
#pragma unroll
                for (int q = 0; q < AR; ++q) atomicAdd(&lcount[slot[q]], 1u);
            }
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) {
                uint64_t* acc = lacc + (size_t)a * C2;
                VT x[AR];
                bool from_key = false;
                if constexpr (!K64) from_key = in.packed[a] != 0;
                if (from_key) { if constexpr (!K64) { _Pragma("unroll") for (int q = 0; q < AR; ++q) x[q] = (VT)((((uint32_t)cur.key[q] >> in.pshift[a]) & in.pmask[a]) + in.pmin[a]); } }
                else if (in.col[a]) { _Pragma("unroll") for (int q = 0; q < AR; ++q) x[q] = cur.v[a][q]; }
                else { _Pragma("unroll") for (int q = 0; q < AR; ++q) x[q] = (VT)cur.row[q]; }
#define AQG_ROWS(expr) _Pragma("unroll") for (int q = 0; q < AR; ++q) { expr; } break
                switch (ops.opc[a]) {
                case OPC_ADDI_I32: AQG_ROWS(atomicAdd(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)(long long)(int32_t)(uint32_t)x[q]));
                case OPC_ADDI_U32: AQG_ROWS(atomicAdd(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)(uint32_t)x[q]));
                case OPC_ADDF_F32: AQG_ROWS(atomicAdd(reinterpret_cast<double*>(acc + slot[q]), (double)__uint_as_float((uint32_t)x[q])));
                case OPC_ADDF_F64: AQG_ROWS(atomicAdd(reinterpret_cast<double*>(acc + slot[q]), __builtin_bit_cast(double, (uint64_t)x[q])));
                case OPC_MIN_I32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)map_i((int32_t)(uint32_t)x[q])));
                case OPC_MAX_I32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)map_i((int32_t)(uint32_t)x[q])));
                case OPC_MIN_U32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)(uint32_t)x[q]));
                case OPC_MAX_U32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)(uint32_t)x[q]));
                case OPC_MIN_F32: AQG_ROWS(atomicMin(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)map_f((double)__uint_as_float((uint32_t)x[q]))));
                case OPC_MAX_F32: AQG_ROWS(atomicMax(reinterpret_cast<unsigned long long*>(acc + slot[q]), (unsigned long long)map_f((double)__uint_as_float((uint32_t)x[q]))));
                default: AQG_ROWS(acc_apply(acc + slot[q], as.kind[a], val_operand_bits(as.dt[a] == AQG_NONE ? AQG_UINT32 : as.dt[a], (uint64_t)x[q], as.kind[a], as.square[a], as.part[a])));
                }
#undef AQG_ROWS
            }
            cur = nxt;
            i0 += STEP;
        }
        __syncthreads();
        // the slots that saw a row become records, reserved with one global atomic per partition
        uint32_t mine = 0;
        for (uint32_t j = threadIdx.x; j <= cap; j += SB) mine += lfirst[j] != NOROW ? 1u : 0u;
        mine = wave_reduce(mine, OpAdd{});
        if (lane_id() == 0 && mine) atomicAdd(&lused, mine);
        __syncthreads();
        if (threadIdx.x == 0) {
            gbase = atomicAdd(&out.flags[1], lused);
            if (part_base) { part_base[2 * (size_t)part] = gbase; part_base[2 * (size_t)part + 1] = lused; }
        }
        __syncthreads();
        for (uint32_t j = threadIdx.x; j <= cap; j += SB) {
            if (lfirst[j] == NOROW) continue;
            const uint32_t g = gbase + atomicAdd(&lemit, 1u);
            if (g >= out_cap) { out.flags[0] = 1; continue; }
            *out.key_p(g) = j == cap ? (K64 ? EMPTY64 : (uint64_t)EMPTY32) : (uint64_t)ktab[j];
            *out.first_p(g) = lfirst[j];
            *out.count_p(g) = need_count ? lcount[j] : 0;
            _Pragma("unroll") for (int a = 0; a < NACC; ++a) *out.acc_p(a, g) = lacc[(size_t)a * C2 + j];
        }
        __syncthreads();
    }
}


} // namespace

// groups one partition's LDS holds, and its key-table capacity, for (ksz, as, need_count)
static void p1_capacity(int ksz, const AccSpec& as, int need_count, uint32_t* gmax, uint32_t* cap) {
    static const int lf_env = getenv("AQG_P1_LF1000") ? atoi(getenv("AQG_P1_LF1000")) : 0;
    const uint32_t lf = lf_env > 0 ? (uint32_t)lf_env : LF1000;
    const double dense = 4.0 + (need_count ? 4.0 : 0.0) + 8.0 * as.nacc;
    const double slot = (double)(ksz + 2) * 1000.0 / lf;
    uint32_t g = (uint32_t)((double)(AGG_LDS - 64) / (dense + slot));
    if (g > 65000) g = 65000;                 // dense ids are 16 bits
    g &= ~3u;
    *gmax = g;
    *cap = ((uint32_t)((uint64_t)g * 1000 / lf) + 7) & ~7u;
}

// the slot-indexed layout (p1_agg_slot_kernel): slots one partition's LDS holds, and the groups it is planned for (load 0.6)
static void p1_slot_capacity(int ksz, const AccSpec& as, int need_count, uint32_t* cap, uint32_t* groups) {
    const size_t per = (size_t)ksz + 4 + (need_count ? 4 : 0) + 8 * (size_t)as.nacc;
    size_t c = (AGG_LDS - 64) / per;
    if (c > 32768) c = 32768;
    *cap = (uint32_t)(c > 16 ? c - 2 : 0) & ~7u;
    *groups = (uint32_t)((uint64_t)*cap * 600 / 1000);
}
// number of partitions for `hint` expected groups: mean + 5 sigma of a partition's group count must fit gmax (0: no plan).
// *layout (optional): AQG_P1_LAYOUT_SLOT when the accumulators can sit inside the key table within the two-level plan's partition limit
// (one dependent LDS round trip per row instead of three), else AQG_P1_LAYOUT_DENSE_IDS (fewer, fuller partitions)
uint32_t aqg_partition_parts(int ksz, const AccSpec& as, int need_count, uint32_t hint, int* layout) {
    if (layout) {
        *layout = AQG_P1_LAYOUT_DENSE_IDS;
        static const bool slot_off = getenv("AQG_DISABLE_SLOT_LAYOUT") != nullptr;     // A/B measurements only
        static const int forced = getenv("AQG_P1_BINS") ? atoi(getenv("AQG_P1_BINS")) : 0;
        uint32_t scap, sgroups;
        p1_slot_capacity(ksz, as, need_count, &scap, &sgroups);
        if (!slot_off && !forced && sgroups >= 256) {
            double mu = (double)sgroups;
            for (int it = 0; it < 8; ++it) mu = (double)sgroups - 5.0 * sqrt(mu);
            uint64_t bins = (uint64_t)((double)hint / mu) + 1;
            if (bins < 256) bins = 256;
            if (bins <= AQG_P2_MAXPARTS - 64) { *layout = AQG_P1_LAYOUT_SLOT; return (uint32_t)bins; }
        }
    }
    uint32_t gmax, cap;
    p1_capacity(ksz, as, need_count, &gmax, &cap);
    // mu + 5 sqrt(mu) <= gmax - 1
    double mu = (double)gmax - 1.0;
    for (int it = 0; it < 8; ++it) mu = (double)gmax - 1.0 - 5.0 * sqrt(mu);
    if (mu < 16) return 0;
    uint64_t bins = (uint64_t)((double)hint / mu) + 1;
    if (bins < 256) bins = 256;               // every CU gets a partition
    { static const int forced = getenv("AQG_P1_BINS") ? atoi(getenv("AQG_P1_BINS")) : 0; if (forced > 0) bins = (uint64_t)forced; }   // measurements only
    return bins <= (1u << 20) ? (uint32_t)bins : 0;
}

// aggregate the partitions [pstart[p * pstride], pstart[(p + 1) * pstride]) (the last one ends at n) of the partitioned planes
static void p1_agg_args(const AccSpec& as, const ValCols& vc, void* const* pvals, const PackPlan* pp, AggIn* inp, AggOps* opsp, bool* v8p) {
    AggIn& in = *inp;
    memset(&in, 0, sizeof in);
    if (pp) in.kclear = pp->kclear;
    for (int a = 0; a < as.nacc; ++a) {
        const int f = pp && as.dt[a] != AQG_NONE ? pack_field_of(*pp, as.col[a]) : -1;
        if (f >= 0) { in.col[a] = nullptr; in.esz[a] = 4; in.packed[a] = 1; in.pshift[a] = pp->shift[f]; in.pmask[a] = pp->fmask[f]; in.pmin[a] = pp->min[f]; }
        else if (vc.of_acc[a] >= 0) { in.col[a] = pvals[vc.of_acc[a]]; in.esz[a] = (int)part_val_bytes(vc.dt[vc.of_acc[a]]); }
        else { in.col[a] = nullptr; in.esz[a] = 4; }     // row-index operands: the carried row id
    }
    AggOps& ops = *opsp;
    memset(&ops, 0, sizeof ops);
    bool v8 = false;
    for (int a = 0; a < as.nacc; ++a) {
        const int dt = as.dt[a], kind = as.kind[a];
        int opc = OPC_GENERIC;
        if (!as.square[a] && !as.part[a]) {
            if (dt == AQG_INT32) opc = kind == ACC_ADD_I ? OPC_ADDI_I32 : kind == ACC_MIN ? OPC_MIN_I32 : kind == ACC_MAX ? OPC_MAX_I32 : OPC_GENERIC;
            else if (dt == AQG_UINT32) opc = kind == ACC_ADD_I ? OPC_ADDI_U32 : kind == ACC_MIN ? OPC_MIN_U32 : kind == ACC_MAX ? OPC_MAX_U32 : OPC_GENERIC;
            else if (dt == AQG_FLOAT) opc = kind == ACC_ADD_F ? OPC_ADDF_F32 : kind == ACC_MIN ? OPC_MIN_F32 : kind == ACC_MAX ? OPC_MAX_F32 : OPC_GENERIC;
            else if (dt == AQG_DOUBLE && kind == ACC_ADD_F) opc = OPC_ADDF_F64;
        }
        ops.opc[a] = opc;
        v8 = v8 || in.esz[a] == 8;
    }
    *v8p = v8;
}

uint32_t p1_direct_capacity(const AccSpec& as, int need_count) {
    const size_t per = 4 + (need_count ? 4 : 0) + 8 * (size_t)as.nacc;
    size_t w = (AGG_LDS - 64) / per;
    if (w > 32768) w = 32768;
    return (uint32_t)w - 1;                        // (one entry is the dummy)
}
int p1_launch_agg_direct(aqg_ctx* ctx, const AccSpec& as, const ValCols& vc, const void* pkeys, const void* prows, void* const* pvals,
                                const uint32_t* pstart, uint32_t pstride, uint32_t n, int need_count, GTable out, uint32_t out_cap, const PackPlan* pp, const RangePlan& rp) {
    AggIn in; AggOps ops; bool v8;
    p1_agg_args(as, vc, pvals, pp, &in, &ops, &v8);
    const size_t lds = (size_t)(rp.W + 1) * (4 + (need_count ? 4 : 0) + 8 * (size_t)as.nacc) + 16;
    const unsigned grid = rp.P < (unsigned)ctx->num_cu ? rp.P : (unsigned)ctx->num_cu;
    DirectSpec ds{rp.M, rp.kmin, rp.D, rp.W, out.flags + 6};
    auto launch = [&](auto kern) -> int {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SB), lds, ctx->stream, static_cast<const uint32_t*>(pkeys), static_cast<const uint32_t*>(prows), as, in, ops, pstart, pstride, rp.P, n, ds, need_count, out, out_cap);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "p1_agg_direct_kernel");
    };
    auto pick = [&](auto nacc) -> int {
        constexpr int N = decltype(nacc)::value;
        return v8 ? launch(&p1_agg_direct_kernel<N, true>) : launch(&p1_agg_direct_kernel<N, false>);
    };
#define AQG_P1_CASE(N) case N: return pick(std::integral_constant<int, N>{});
    switch (as.nacc) {
    AQG_P1_CASE(0) AQG_P1_CASE(1) AQG_P1_CASE(2) AQG_P1_CASE(3) AQG_P1_CASE(4) AQG_P1_CASE(5) AQG_P1_CASE(6) AQG_P1_CASE(7)
    default: return pick(std::integral_constant<int, 8>{});
    }
#undef AQG_P1_CASE
}

int p1_launch_agg(aqg_ctx* ctx, int ksz, const AccSpec& as, const ValCols& vc, const void* pkeys, const void* prows, void* const* pvals,
                  const uint32_t* pstart, uint32_t pstride, uint32_t nparts, uint32_t n, int need_count, GTable out, uint32_t out_cap, PartRows* pr,
                  const PackPlan* pp, int layout) {
    AggIn in; AggOps ops; bool v8;
    p1_agg_args(as, vc, pvals, pp, &in, &ops, &v8);
    if (layout == AQG_P1_LAYOUT_SLOT) {
        uint32_t scap, sgroups;
        p1_slot_capacity(ksz, as, need_count, &scap, &sgroups);
        const size_t lds = (size_t)(scap + 2) * ((size_t)ksz + 4 + (need_count ? 4 : 0) + 8 * (size_t)as.nacc) + 16;
        const unsigned grid = nparts < (unsigned)ctx->num_cu ? nparts : (unsigned)ctx->num_cu;
        uint32_t* part_base = nullptr;
        if (pr) {
            AQG_TRY(aqg_ws_get(ctx, 2 * (size_t)nparts + 2, &part_base));
            AQG_HIP(ctx, hipMemsetAsync(part_base, 0, (2 * (size_t)nparts + 2) * 4, ctx->stream));
            pr->keys = pkeys; pr->rows = static_cast<const uint32_t*>(prows); pr->pstart = pstart; pr->pstride = pstride; pr->nparts = nparts; pr->ntotal = n;
            pr->ksz = ksz; pr->part_base = part_base; pr->cap = scap; pr->valid = true;
        }
        auto launch = [&](auto kern) -> int {
            AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
            aqg_kernel_timer_begin(ctx);
            hipLaunchKernelGGL(kern, dim3(grid), dim3(SB), lds, ctx->stream, pkeys, static_cast<const uint32_t*>(prows), as, in, ops, pstart, pstride, nparts, n, scap, need_count, out, out_cap, part_base);
            aqg_kernel_timer_end(ctx);
            return aqg_check_launch(ctx, "p1_agg_slot_kernel");
        };
        auto pick = [&](auto nacc) -> int {
            constexpr int N = decltype(nacc)::value;
            if (ksz == 4) return v8 ? launch(&p1_agg_slot_kernel<N, false, true>) : launch(&p1_agg_slot_kernel<N, false, false>);
            return v8 ? launch(&p1_agg_slot_kernel<N, true, true>) : launch(&p1_agg_slot_kernel<N, true, false>);
        };
#define AQG_P1_CASE(N) case N: return pick(std::integral_constant<int, N>{});
        switch (as.nacc) {
        AQG_P1_CASE(0) AQG_P1_CASE(1) AQG_P1_CASE(2) AQG_P1_CASE(3) AQG_P1_CASE(4) AQG_P1_CASE(5) AQG_P1_CASE(6) AQG_P1_CASE(7)
        default: return pick(std::integral_constant<int, 8>{});
        }
#undef AQG_P1_CASE
    }
    uint32_t gmax, cap;
    p1_capacity(ksz, as, need_count, &gmax, &cap);
    const size_t lds = (size_t)gmax * (4 + (need_count ? 4 : 0) + 8 * (size_t)as.nacc) + (size_t)cap * (ksz + 2) + 16;
    const unsigned grid = nparts < (unsigned)ctx->num_cu ? nparts : (unsigned)ctx->num_cu;
    uint32_t* part_base = nullptr;
    if (pr) {                                      // the build: where the partitioned rows lie and which records every partition wrote
        AQG_TRY(aqg_ws_get(ctx, 2 * (size_t)nparts + 2, &part_base));
        AQG_HIP(ctx, hipMemsetAsync(part_base, 0, (2 * (size_t)nparts + 2) * 4, ctx->stream));
        pr->keys = pkeys; pr->rows = static_cast<const uint32_t*>(prows); pr->pstart = pstart; pr->pstride = pstride; pr->nparts = nparts; pr->ntotal = n;
        pr->ksz = ksz; pr->part_base = part_base; pr->cap = cap; pr->valid = true;
    }
    auto launch = [&](auto kern) -> int {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SB), lds, ctx->stream, pkeys, static_cast<const uint32_t*>(prows), as, in, ops, pstart, pstride, nparts, n, cap, gmax, need_count, out, out_cap, part_base);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "p1_agg_kernel");
    };
    auto pick = [&](auto nacc) -> int {
        constexpr int N = decltype(nacc)::value;
        if (ksz == 4) return v8 ? launch(&p1_agg_kernel<N, false, true>) : launch(&p1_agg_kernel<N, false, false>);
        return v8 ? launch(&p1_agg_kernel<N, true, true>) : launch(&p1_agg_kernel<N, true, false>);
    };
#define AQG_P1_CASE(N) case N: return pick(std::integral_constant<int, N>{});
    switch (as.nacc) {
    AQG_P1_CASE(0) AQG_P1_CASE(1) AQG_P1_CASE(2) AQG_P1_CASE(3) AQG_P1_CASE(4) AQG_P1_CASE(5) AQG_P1_CASE(6) AQG_P1_CASE(7)
    default: return pick(std::integral_constant<int, 8>{});
    }
#undef AQG_P1_CASE
}


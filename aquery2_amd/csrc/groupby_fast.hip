// groupby_fast.hip -- the fast LDS group-by kernel: one or two 4-byte keys or one 8-byte key, up to four accumulators over
// 4- and 8-byte value columns (h2o Q1, Q4 and most few-group shapes; the first pass of aqg_groupby_build).  Planner, table
// layout and everything downstream: groupby.hip.  Replaces the hash-table build + per-group loop of the reference for these
// shapes (server/hasher.h:146-199, engine/ast.py:722-789).
#include "groupby_fast.hpp"

namespace {

// ---- fast path: one or two 4-byte key columns, up to four accumulators over 4-byte value columns (h2o Q1, Q4, ...) ------------
// Same LDS open-addressing idea as agg_kernel, pared down to what these shapes need: the slot is the packed key alone (4 or 8
// bytes), eight rows per lane per step with all probes issued before the first compare, and NO first-row bookkeeping in the
// loop -- first rows are recovered afterwards by first_rows_kernel, which stops as soon as every group has one (a few tiles on
// h2o data).  vkind: 0 int32, 1 uint32, 2 float (VW = 4) or also 3 int64, 4 uint64, 5 double, 6 int8, 7 uint8 / bool, 8 int16, 9 uint16
// (VW = 8: some value column is not 4 bytes wide;
// an int64 sum is two accumulators over the same column, `part` 1 / 2 = its low / high half);
// kind: ACC_ADD_I / ACC_ADD_F / ACC_MIN / ACC_MAX; square: accumulate x*x.
// (The generic agg_kernel ran max(v1),min(v2) by id1 at 27 % of the HBM roofline and var(v1) at 18 %; this kernel does SUM at 75 %.)

template <int NV, bool COUNT, bool K64, int VW = 4>
__global__ void __launch_bounds__(1024) agg32_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ keys_hi, FastVals fv, GTable gt, uint32_t n, uint32_t lcap) {
    using KT = std::conditional_t<K64, uint64_t, uint32_t>;
    constexpr KT EMPTYK = K64 ? (KT)EMPTY64 : (KT)EMPTY32;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const uint32_t LT = lcap + 1;
    uint64_t* lacc = reinterpret_cast<uint64_t*>(smem_raw);                  // [NV][LT]
    KT* lkey = reinterpret_cast<KT*>(lacc + (size_t)NV * LT);                // [LT], slot lcap = the key equal to the empty mark
    uint32_t* lcount = reinterpret_cast<uint32_t*>(lkey + LT);               // [LT] if COUNT
    uint32_t* ltouch = lcount + (COUNT ? LT : 0);                            // [1]  sentinel slot used?
    __shared__ uint32_t lused;
    const uint32_t lmask = lcap - 1, llimit = lcap - (lcap >> 2), lbits = 31 - __clz(lcap);
    for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {
        lkey[s] = EMPTYK;
        _Pragma("unroll") for (int a = 0; a < NV; ++a) lacc[(size_t)a * LT + s] = acc_init(fv.kind[a]);
        if constexpr (COUNT) lcount[s] = 0;
    }
    if (threadIdx.x == 0) { lused = 0; *ltouch = 0; }
    __syncthreads();

    auto slot_of = [&](KT k) -> uint32_t {
        if constexpr (K64) return lds_h1<false>((uint64_t)k) >> (32 - lbits); else return fib_slot((uint32_t)k, lbits);
    };
    auto slow_slot = [&](KT k) -> uint32_t {            // insert path (first sight of a key in this workgroup)
        if (k == EMPTYK) { *ltouch = 1; return lcap; }
        uint32_t s = slot_of(k);
        for (uint32_t p = 0; p <= lmask; ++p) {
            KT cur = lkey[s];
            if (cur == k) return s;
            if (cur == EMPTYK) {
                if (lused >= llimit) return FAIL;
                KT old;
                if constexpr (K64) old = atomicCAS(reinterpret_cast<unsigned long long*>(&lkey[s]), (unsigned long long)EMPTYK, (unsigned long long)k);
                else old = atomicCAS(&lkey[s], EMPTYK, k);
                if (old == EMPTYK) { atomicAdd(&lused, 1u); return s; }
                if (old == k) return s;
            }
            s = (s + 1) & lmask;
        }
        return FAIL;
    };
    using VB = std::conditional_t<VW == 8, uint64_t, uint32_t>;       // raw bits of one value
    auto operand = [&](int a, VB bits) -> uint64_t {
        if constexpr (VW == 8) {
            switch (fv.vkind[a]) {
            case 0: return val_operand_t((int32_t)(uint32_t)bits, fv.kind[a], fv.square[a]);
            case 1: return val_operand_t((uint32_t)bits, fv.kind[a], fv.square[a]);
            case 2: return val_operand_t(__uint_as_float((uint32_t)bits), fv.kind[a], fv.square[a]);
            case 3: return val_operand_t((int64_t)bits, fv.kind[a], fv.square[a], fv.part[a]);
            case 4: return val_operand_t((uint64_t)bits, fv.kind[a], fv.square[a], fv.part[a]);
            case 5: return val_operand_t(__builtin_bit_cast(double, (uint64_t)bits), fv.kind[a], fv.square[a]);
            case 6: return val_operand_t((int8_t)(uint8_t)bits, fv.kind[a], fv.square[a]);
            case 7: return val_operand_t((uint8_t)bits, fv.kind[a], fv.square[a]);
            case 8: return val_operand_t((int16_t)(uint16_t)bits, fv.kind[a], fv.square[a]);
            default: return val_operand_t((uint16_t)bits, fv.kind[a], fv.square[a]);
            }
        } else {
            switch (fv.vkind[a]) {
            case 0: return val_operand_t((int32_t)bits, fv.kind[a], fv.square[a]);
            case 1: return val_operand_t((uint32_t)bits, fv.kind[a], fv.square[a]);
            default: return val_operand_t(__uint_as_float((uint32_t)bits), fv.kind[a], fv.square[a]);
            }
        }
    };
    auto to_table = [&](KT k, const VB* vbits) {   // rare: LDS table at its load limit, or tail rows
        uint32_t g = gt_find_or_insert(gt, K64 ? (uint64_t)k : (uint64_t)(uint32_t)k);
        if (g == FAIL) return;
        atomicMin(gt.first_p(g), OCCUPIED);
        if constexpr (COUNT) atomicAdd(gt.count_p(g), 1u);
        _Pragma("unroll") for (int a = 0; a < NV; ++a) acc_apply(gt.acc_p(a, g), fv.kind[a], operand(a, vbits[a]));
    };

    const uint32_t nchunk = n >> 3;            // 8 consecutive rows per lane per step
    uint32_t c_lo, c_hi;
    wg_span(nchunk, c_lo, c_hi);               // one contiguous span of rows per workgroup
    for (uint32_t c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        const size_t base = (size_t)c * 8;
        // K64 with keys_hi == nullptr: `keys` is ONE 8-byte column (its bits are the packed key); otherwise two 4-byte columns.
        // Either way four 16-byte loads into the same registers; only the way a key is put together differs.
        const bool key8 = K64 && keys_hi == nullptr;
        pack<uint32_t, 4> kq[K64 ? 4 : 2];
        if constexpr (K64) {
            _Pragma("unroll") for (int q = 0; q < 4; ++q) {
                const uint32_t* src = key8 ? keys + 2 * base + 4 * q : (q < 2 ? keys + base + 4 * q : keys_hi + base + 4 * (q - 2));
                kq[q] = *reinterpret_cast<const pack<uint32_t, 4>*>(src);
            }
        } else {
            kq[0] = *reinterpret_cast<const pack<uint32_t, 4>*>(keys + base);
            kq[1] = *reinterpret_cast<const pack<uint32_t, 4>*>(keys + base + 4);
        }
        pack<uint32_t, 4> v0[NV ? NV : 1], v1[NV ? NV : 1];
        pack<uint64_t, 2> w[VW == 8 && NV ? NV : 1][4];
        _Pragma("unroll") for (int a = 0; a < NV; ++a) {
            if constexpr (VW == 8) {
                if (a > 0 && fv.col[a] == fv.col[a - 1]) {   // both halves of an int64 sum (or sum and sum of squares) read one column
                    _Pragma("unroll") for (int q = 0; q < 4; ++q) w[a][q] = w[a - 1][q];
                } else if (fv.vkind[a] >= 3 && fv.vkind[a] <= 5) {
                    _Pragma("unroll") for (int q = 0; q < 4; ++q) w[a][q] = *reinterpret_cast<const pack<uint64_t, 2>*>(static_cast<const uint64_t*>(fv.col[a]) + base + 2 * q);
                } else if (fv.vkind[a] <= 2) {               // a 4-byte column: eight rows in the first two register pairs
                    _Pragma("unroll") for (int q = 0; q < 2; ++q) w[a][q] = *reinterpret_cast<const pack<uint64_t, 2>*>(static_cast<const uint32_t*>(fv.col[a]) + base + 4 * q);
                } else if (fv.vkind[a] <= 7) {               // a 1-byte column: eight rows in one 8-byte load
                    w[a][0].v[0] = *reinterpret_cast<const uint64_t*>(static_cast<const uint8_t*>(fv.col[a]) + base);
                } else {                                     // a 2-byte column: eight rows in one 16-byte load
                    w[a][0] = *reinterpret_cast<const pack<uint64_t, 2>*>(static_cast<const uint16_t*>(fv.col[a]) + base);
                }
            } else {
                v0[a] = *reinterpret_cast<const pack<uint32_t, 4>*>(static_cast<const uint32_t*>(fv.col[a]) + base);
                v1[a] = *reinterpret_cast<const pack<uint32_t, 4>*>(static_cast<const uint32_t*>(fv.col[a]) + base + 4);
            }
        }
        auto raw = [&](int a, int j) -> VB {
            if constexpr (VW == 8) {
                if (fv.vkind[a] >= 3 && fv.vkind[a] <= 5) return w[a][j >> 1].v[j & 1];
                if (fv.vkind[a] <= 2) return (w[a][j >> 2].v[(j >> 1) & 1] >> (32 * (j & 1))) & 0xFFFFFFFFull;
                if (fv.vkind[a] <= 7) return (w[a][0].v[0] >> (8 * j)) & 0xFFull;
                return (w[a][0].v[j >> 2] >> (16 * (j & 3))) & 0xFFFFull;
            } else return j < 4 ? v0[a].v[j] : v1[a].v[j - 4];
        };
        KT k[8], cur[8];
        uint32_t slot[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (K64) {
                const uint32_t lo = key8 ? kq[j >> 1].v[2 * (j & 1)] : kq[j >> 2].v[j & 3];
                const uint32_t hi = key8 ? kq[j >> 1].v[2 * (j & 1) + 1] : kq[2 + (j >> 2)].v[j & 3];
                k[j] = (uint64_t)lo | ((uint64_t)hi << 32);
            } else k[j] = kq[j >> 2].v[j & 3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { slot[j] = slot_of(k[j]); cur[j] = lkey[slot[j]]; }     // eight probes in flight
#pragma unroll
        for (int j = 0; j < 8; ++j) if (cur[j] != k[j] || k[j] == EMPTYK) slot[j] = slow_slot(k[j]);   // (walking the missed rows' probe sequences together measured slower here)
        if constexpr (COUNT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(&lcount[slot[j]], 1u);
        }
        _Pragma("unroll") for (int a = 0; a < NV; ++a) {
            uint64_t* la = lacc + (size_t)a * LT;
            if constexpr (VW == 8) {
                if (fv.kind[a] == ACC_ADD_F && !fv.square[a] && fv.vkind[a] == 5) {        // sum / avg of a double column
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<double*>(&la[slot[j]]), __builtin_bit_cast(double, (uint64_t)raw(a, j)));
                } else {
                    uint64_t o[8];
                    auto raw64 = [&](int j) -> uint64_t { return w[a][j >> 1].v[j & 1]; };
                    auto raw32 = [&](int j) -> uint32_t { return (uint32_t)(w[a][j >> 2].v[(j >> 1) & 1] >> (32 * (j & 1))); };
                    switch (fv.vkind[a]) {      // the dtype switch outside the eight rows
                    case 0:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t((int32_t)raw32(j), fv.kind[a], fv.square[a]);
                        break;
                    case 1:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t(raw32(j), fv.kind[a], fv.square[a]);
                        break;
                    case 2:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t(__uint_as_float(raw32(j)), fv.kind[a], fv.square[a]);
                        break;
                    case 3:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t((int64_t)raw64(j), fv.kind[a], fv.square[a], fv.part[a]);
                        break;
                    case 4:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t(raw64(j), fv.kind[a], fv.square[a], fv.part[a]);
                        break;
                    case 5:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t(__builtin_bit_cast(double, raw64(j)), fv.kind[a], fv.square[a]);
                        break;
                    case 6:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t((int8_t)(uint8_t)(w[a][0].v[0] >> (8 * j)), fv.kind[a], fv.square[a]);
                        break;
                    case 7:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t((uint8_t)(w[a][0].v[0] >> (8 * j)), fv.kind[a], fv.square[a]);
                        break;
                    case 8:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t((int16_t)(uint16_t)(w[a][0].v[j >> 2] >> (16 * (j & 3))), fv.kind[a], fv.square[a]);
                        break;
                    default:
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = val_operand_t((uint16_t)(w[a][0].v[j >> 2] >> (16 * (j & 3))), fv.kind[a], fv.square[a]);
                        break;
                    }
                    switch (fv.kind[a]) {
                    case ACC_ADD_I:
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)o[j]);
                        break;
                    case ACC_ADD_F:
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<double*>(&la[slot[j]]), __builtin_bit_cast(double, o[j]));
                        break;
                    case ACC_MIN:
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicMin(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)o[j]);
                        break;
                    default:
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicMax(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)o[j]);
                        break;
                    }
                }
                continue;
            }
            // wave-uniform branches, one per accumulator per eight rows; plain sums keep their own straight-line form
            if (fv.kind[a] == ACC_ADD_F && !fv.square[a]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<double*>(&la[slot[j]]), (double)__uint_as_float(j < 4 ? v0[a].v[j] : v1[a].v[j - 4]));
            } else if (fv.kind[a] == ACC_ADD_I && !fv.square[a] && fv.vkind[a] == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)(int64_t)(int32_t)(j < 4 ? v0[a].v[j] : v1[a].v[j - 4]));
            } else if (fv.kind[a] == ACC_ADD_I && !fv.square[a]) {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)(j < 4 ? v0[a].v[j] : v1[a].v[j - 4]));
            } else {
                uint64_t o[8];
                switch (fv.vkind[a]) {      // the dtype switch outside the eight rows
                case 0:
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = val_operand_t((int32_t)(j < 4 ? v0[a].v[j] : v1[a].v[j - 4]), fv.kind[a], fv.square[a]);
                    break;
                case 1:
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = val_operand_t((uint32_t)(j < 4 ? v0[a].v[j] : v1[a].v[j - 4]), fv.kind[a], fv.square[a]);
                    break;
                default:
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = val_operand_t(__uint_as_float(j < 4 ? v0[a].v[j] : v1[a].v[j - 4]), fv.kind[a], fv.square[a]);
                    break;
                }
                switch (fv.kind[a]) {
                case ACC_ADD_I:
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)o[j]);
                    break;
                case ACC_ADD_F:
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicAdd(reinterpret_cast<double*>(&la[slot[j]]), __builtin_bit_cast(double, o[j]));
                    break;
                case ACC_MIN:
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicMin(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)o[j]);
                    break;
                default:
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (slot[j] != FAIL) atomicMax(reinterpret_cast<unsigned long long*>(&la[slot[j]]), (unsigned long long)o[j]);
                    break;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (slot[j] == FAIL) {
                VB vb[NV ? NV : 1];
                _Pragma("unroll") for (int a = 0; a < NV; ++a) vb[a] = raw(a, j);
                to_table(k[j], vb);
            }
        }
    }
    if (blockIdx.x == 0) {                     // tail rows (< 8)
        uint32_t row = (nchunk << 3) + threadIdx.x;
        if (row < n) {
            VB vb[NV ? NV : 1];
            _Pragma("unroll") for (int a = 0; a < NV; ++a) {
                if (VW == 8 && fv.vkind[a] <= 2) vb[a] = static_cast<const uint32_t*>(fv.col[a])[row];
                else if (VW == 8 && fv.vkind[a] >= 8) vb[a] = static_cast<const uint16_t*>(fv.col[a])[row];
                else if (VW == 8 && fv.vkind[a] >= 6) vb[a] = static_cast<const uint8_t*>(fv.col[a])[row];
                else vb[a] = static_cast<const VB*>(fv.col[a])[row];
            }
            KT key;
            if constexpr (K64) key = keys_hi ? ((uint64_t)keys[row] | ((uint64_t)keys_hi[row] << 32)) : reinterpret_cast<const uint64_t*>(keys)[row]; else key = keys[row];
            to_table(key, vb);
        }
    }
    __syncthreads();
    for (uint32_t s = threadIdx.x; s < LT; s += blockDim.x) {
        const KT key = lkey[s];
        if (s < lcap ? key == EMPTYK : *ltouch == 0) continue;
        uint32_t g = gt_find_or_insert(gt, s < lcap ? (K64 ? (uint64_t)key : (uint64_t)(uint32_t)key) : (K64 ? EMPTY64 : (uint64_t)EMPTY32));
        if (g == FAIL) continue;
        atomicMin(gt.first_p(g), OCCUPIED);
        if constexpr (COUNT) atomicAdd(gt.count_p(g), lcount[s]);
        _Pragma("unroll") for (int a = 0; a < NV; ++a) acc_apply(gt.acc_p(a, g), fv.kind[a], lacc[(size_t)a * LT + s]);
    }
}

} // namespace

int aqg_fast_aggregate(aqg_ctx* ctx, const uint32_t* keys, const uint32_t* keys_hi, bool k64, bool v8, int nacc, bool need_count,
                       const FastVals& fv, GTable gt, uint32_t n, uint32_t lcap, size_t lds, unsigned grid, unsigned block) {
    auto launch = [&](auto kern) -> int {
        AQG_TRY(aqg_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
        aqg_kernel_timer_begin(ctx);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, ctx->stream, keys, keys_hi, fv, gt, n, lcap);
        aqg_kernel_timer_end(ctx);
        return aqg_check_launch(ctx, "agg32_kernel");
    };
    auto by_nv = [&](auto count_tag, auto k64_tag) -> int {
        constexpr bool C = decltype(count_tag)::value, K = decltype(k64_tag)::value;
        if (v8) switch (nacc) {
        case 1: return launch(&agg32_kernel<1, C, K, 8>);
        case 2: return launch(&agg32_kernel<2, C, K, 8>);
        case 3: return launch(&agg32_kernel<3, C, K, 8>);
        default: return launch(&agg32_kernel<4, C, K, 8>);
        }
        switch (nacc) {
        case 0: return launch(&agg32_kernel<0, C, K>);
        case 1: return launch(&agg32_kernel<1, C, K>);
        case 2: return launch(&agg32_kernel<2, C, K>);
        case 3: return launch(&agg32_kernel<3, C, K>);
        default: return launch(&agg32_kernel<4, C, K>);
        }
    };
    if (need_count) return k64 ? by_nv(std::true_type{}, std::true_type{}) : by_nv(std::true_type{}, std::false_type{});
    return k64 ? by_nv(std::false_type{}, std::true_type{}) : by_nv(std::false_type{}, std::false_type{});
}

// device.h -- host-side runtime that ties the header-level API (vector_type / ColRef / aggregations /
// hasher) to the C-ABI of the MI355X library (include/aqg.h).  Header-only; state lives in one
// process-wide Runtime (the reference's post-processor is single-caller: one engine thread,
// SURVEY 8b).
//
// Model: a column keeps the reference's 16-byte {container, size, capacity} triple.  Device residency
// is tracked out of band, keyed by the HOST address range of the buffer:
//   PINNED  a borrowed host column (capacity == 0: `ColRef<T>(len, server->getCol(i))`) uploaded on first
//           use and cached -- the reference's zero-copy view of the data source becomes a device mirror;
//           the host data is assumed immutable while the mirror exists (drop_pins() at session end).
//   RESULT  a column produced by a device kernel.  Its host buffer is allocated exactly as the reference
//           would (malloc / scratch arena) but filled lazily: the first host access (operator[], begin(),
//           out(), ...) downloads it AND DROPS THE DEVICE COPY -- vector_type hands out mutable host access
//           (operator[] returns _Ty&), so a mirror kept past that point could go stale behind a host write
//           (`auto x = a + b; x[0] = 5; auto y = x * c;`): a later device use uploads the host data again.
//           Chained expressions such as max(price - mins(price)) never touch the host and never leave HBM.
//           (A grouping's row-id buffer keeps its registration after a download: `col[vecs[g]]` is recognised through it.
//           Generated code only reads vecs[g]; its device copy and the per-grouping aggregate cache assume that, like PINNED.)
// There is no CPU fallback: without the library or a GPU every operation aborts with a message.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <array>
#include <map>
#include <utility>
#include <vector>

#include "../aqg.h"

namespace aq {
namespace dev {

[[noreturn]] inline void die(const char* what, int rc, aqg_ctx* ctx) {
    std::fprintf(stderr, "[aquery-mi355x] %s failed (status %d): %s\n", what, rc, ctx ? aqg_last_error(ctx) : "no context");
    std::abort();
}

// A column that exists (or can be made) for ALL groups of one grouping at once -- what a per-group temporary of the generated loop
// (`col[vecs[g]]`, `avgw(5, col[vecs[g]])`, `a[vecs[g]] - mins(a[vecs[g]])`, engine/ast.py:749-784) is a slice of.
//   layout ROW   n elements in row order: a source column, or an element-wise result of row-order columns
//                (gather commutes with element-wise operators: f(a[val], b[val]) = f(a, b)[val])
//   layout FLAT  n elements in the flat layout of the grouping: position offsets[g] + i <-> row vecs[g][i]; what the per-group scans
//                produce (aqg_grouped_scan_flat) and what the generated code's `buf + offsets[g]` output buffers are laid out as
struct VCol {
    int layout = 0, tag = 0;
    void* dptr = nullptr;                                    // computed columns: owned device column of n elements
    const void* src = nullptr; size_t src_bytes = 0; bool src_borrowed = false;   // source columns: host address, resolved at every use
    void* flat = nullptr;                                    // ROW columns: their flat form (aqg_grouped_flatten), made when first scanned
    std::vector<unsigned char> host; bool host_valid = false;   // host copy of the FLAT form, downloaded once, serves every group's slice
};

// One device group-by (the table behind HashTableFactory::get / AQHashTable).  Its row-id buffer is registered like any
// other RESULT; `vecs[g]` views into it let the runtime recognise `col[vecs[g]]` as "group g of this grouping".
struct GroupCtx {
    aqg_groupby* handle = nullptr;
    uint32_t n = 0, G = 0;
    uint32_t* offsets = nullptr;   // [G+1] host
    uint32_t* counts = nullptr;    // [G]   host
    uint32_t* row_ids = nullptr;   // [n]   host address (device copy registered)
    std::vector<VCol> vcols;
    std::map<std::array<uint64_t, 6>, int> memo;             // (kind, op, operand columns, window / scalar bits) -> column, made once for all groups
    struct ScalarUse { uint64_t bits[2]; int vcol; bool varies; };
    std::map<std::array<uint64_t, 4>, ScalarUse> smemo;      // element-wise (column OP scalar): poisoned once the scalar differs between groups
    // (column, op) -> G result slots of 16 bytes, filled for ALL groups by one kernel on first request
    std::map<std::pair<int, int>, std::vector<unsigned char>> cache;
    std::map<std::pair<int, int>, std::vector<double>> corr_cache;
};

struct Entry {
    void* dptr = nullptr;
    size_t bytes = 0;
    bool host_stale = false;   // RESULT not yet downloaded
    bool pinned = false;       // PINNED borrowed column
    GroupCtx* gctx = nullptr;  // set on a grouping's row-id buffer
    // DEFERRED per-group temporary: group `dg`'s slice of column `dv` of grouping `dgroup`; nothing per group has run.  Reductions,
    // scans and element-wise operators on it are answered for all groups at once; anything else materialises the slice first
    bool deferred = false;
    GroupCtx* dgroup = nullptr;
    uint32_t dg = 0;
    int dv = -1;
};

class Runtime {
public:
    static Runtime& get() {
        static Runtime r;
        return r;
    }
    aqg_ctx* ctx() {
        if (!ctx_) {
            int dev = 0;
            if (const char* e = std::getenv("AQ_GPU_DEVICE")) dev = std::atoi(e);
            int rc = aqg_ctx_create(dev, nullptr, &ctx_);
            if (rc != AQG_OK) die("aqg_ctx_create (no MI355X visible; this library has no CPU fallback)", rc, nullptr);
        }
        return ctx_;
    }
    size_t stale = 0;   // number of entries whose host copy is stale (fast path test in operator[])

    // containing entry of a host address, or end()
    std::map<uintptr_t, Entry>::iterator find(const void* p) {
        if (map_.empty()) return map_.end();
        auto it = map_.upper_bound((uintptr_t)p);
        if (it == map_.begin()) return map_.end();
        --it;
        if ((uintptr_t)p < it->first + it->second.bytes || ((uintptr_t)p == it->first && it->second.bytes == 0)) return it;
        return map_.end();
    }
    bool end(std::map<uintptr_t, Entry>::iterator it) { return it == map_.end(); }

    // device address of `bytes` bytes of host data at p.  borrowed != 0: cache the upload (PINNED).
    // `temp_out` receives a temporary device buffer the caller must release() when the data was not cacheable.
    const void* input(const void* p, size_t bytes, bool borrowed, void** temp_out) {
        *temp_out = nullptr;
        if (bytes == 0) return nullptr;
        auto it = find(p);
        if (it != map_.end() && it->second.deferred) materialize(it);
        if (it != map_.end() && it->second.gctx && !it->second.dptr) fill_rows(it->second);
        if (it != map_.end() && (uintptr_t)p + bytes <= it->first + it->second.bytes)
            return static_cast<char*>(it->second.dptr) + ((uintptr_t)p - it->first);
        void* d = nullptr;
        int rc = aqg_malloc(ctx(), bytes, &d);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        rc = aqg_h2d(ctx_, d, p, bytes);
        if (rc != AQG_OK) die("aqg_h2d", rc, ctx_);
        if (borrowed) {
            Entry e; e.dptr = d; e.bytes = bytes; e.pinned = true;
            map_[(uintptr_t)p] = e;
        } else *temp_out = d;
        return d;
    }
    void release(void* temp) { if (temp) aqg_free(ctx(), temp); }
    // register a host buffer whose device copy already exists (ownership of dptr moves to the registry)
    void adopt(void* p, size_t bytes, void* dptr, bool host_valid) {
        forget_range(p, bytes);
        Entry e; e.dptr = dptr; e.bytes = bytes; e.host_stale = !host_valid;
        map_[(uintptr_t)p] = e;
        if (!host_valid) ++stale;
    }

    // register a fresh RESULT buffer for the host range [p, p+bytes); returns its device address
    void* result(void* p, size_t bytes) {
        forget_range(p, bytes);
        void* d = nullptr;
        int rc = aqg_malloc(ctx(), bytes ? bytes : 16, &d);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        Entry e; e.dptr = d; e.bytes = bytes; e.host_stale = true;
        map_[(uintptr_t)p] = e;
        ++stale;
        return d;
    }
    // ---- grouped fast path ------------------------------------------------------------------------------------
    // The row-id lists of a grouping (aqg_groupby_postproc: a radix pass over the group-id column, 4.7 ms per 1e9 rows) are made
    // the first time somebody needs them on the device or on the host.  `col[vecs[g]]` followed by a reduction -- the shape the
    // code generator emits -- never does: it is answered by aqg_grouped_reduce from the group-id column of the build.
    void adopt_group(GroupCtx* g, void* drows) {
        adopt(g->row_ids, (size_t)g->n * 4, drows, /*host_valid=*/false);
        auto it = map_.find((uintptr_t)g->row_ids);
        if (it != map_.end()) it->second.gctx = g;
    }
    void fill_rows(Entry& e) {
        GroupCtx* c = e.gctx;
        void *doff = nullptr, *drows = nullptr;
        int rc = aqg_malloc(ctx(), ((size_t)c->G + 1) * 4, &doff);
        if (rc == AQG_OK) rc = aqg_malloc(ctx_, ((size_t)c->n + 1) * 4, &drows);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        rc = aqg_groupby_postproc(c->handle, static_cast<uint32_t*>(doff), static_cast<uint32_t*>(drows));
        if (rc != AQG_OK) die("aqg_groupby_postproc", rc, ctx_);
        aqg_free(ctx_, doff);
        e.dptr = drows;
    }
    // is [idx, idx+count) exactly the row list of one group of a registered grouping?
    bool group_of(const uint32_t* idx, uint32_t count, GroupCtx** gc, uint32_t* g) {
        auto it = find(idx);
        if (it == map_.end() || !it->second.gctx) return false;
        GroupCtx* c = it->second.gctx;
        uint32_t off = (uint32_t)(idx - c->row_ids);
        uint32_t lo = 0, hi = c->G;                       // largest g with offsets[g] <= off
        while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (c->offsets[mid] <= off) lo = mid; else hi = mid; }
        if (c->G == 0 || c->offsets[lo] != off || c->counts[lo] != count) return false;
        *gc = c; *g = lo;
        return true;
    }
    // ---- columns of a grouping (VCol) ----------------------------------------------------------------------------------------------
    static size_t esz(int tag) { return aqg_dtype_size(tag); }
    int vcol_new(GroupCtx* c, const std::array<uint64_t, 6>& key, const VCol& v) {
        c->vcols.push_back(v);
        const int id = (int)c->vcols.size() - 1;
        c->memo[key] = id;
        return id;
    }
    // the source column behind `col[vecs[g]]`
    int vcol_source(GroupCtx* c, const void* src, size_t src_bytes, bool borrowed, int tag) {
        const std::array<uint64_t, 6> key{0, (uint64_t)(uintptr_t)src, (uint64_t)tag, 0, 0, 0};
        auto it = c->memo.find(key);
        if (it != c->memo.end()) return it->second;
        VCol v; v.layout = 0; v.tag = tag; v.src = src; v.src_bytes = src_bytes; v.src_borrowed = borrowed;
        return vcol_new(c, key, v);
    }
    // device address of a ROW column (`*tmp` must be release()d by the caller)
    const void* vcol_row_ptr(GroupCtx* c, int v, void** tmp) {
        *tmp = nullptr;
        VCol& col = c->vcols[v];
        if (col.dptr) return col.dptr;
        return input(col.src, col.src_bytes, col.src_borrowed, tmp);
    }
    // device address of the FLAT form of a column (a ROW column is flattened once: value-carrying radix passes, no row lists)
    const void* vcol_flat_ptr(GroupCtx* c, int v) {
        if (c->vcols[v].layout == 1) return c->vcols[v].dptr;
        if (!c->vcols[v].flat) {
            void* tmp = nullptr;
            const void* d = vcol_row_ptr(c, v, &tmp);
            void* f = nullptr;
            int rc = aqg_malloc(ctx(), (size_t)c->n * esz(c->vcols[v].tag) + 16, &f);
            if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
            rc = aqg_grouped_flatten(ctx_, c->handle, c->vcols[v].tag, d, f);
            if (rc != AQG_OK) die("aqg_grouped_flatten", rc, ctx_);
            release(tmp);
            c->vcols[v].flat = f;
        }
        return c->vcols[v].flat;
    }
    // scan(op, w) of every group's slice: one segmented scan over the flat column
    int vcol_scan(GroupCtx* c, int v, int op, uint32_t w) {
        const std::array<uint64_t, 6> key{1, (uint64_t)op, (uint64_t)v, w, 0, 0};
        auto it = c->memo.find(key);
        if (it != c->memo.end()) return it->second;
        VCol r; r.layout = 1; r.tag = aqg_scan_out_dtype(op, c->vcols[v].tag);
        const void* xf = vcol_flat_ptr(c, v);
        int rc = aqg_malloc(ctx(), (size_t)c->n * esz(r.tag) + 16, &r.dptr);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        rc = aqg_grouped_scan_flat(ctx_, c->handle, op, c->vcols[v].tag, xf, w, r.dptr);
        if (rc != AQG_OK) die("aqg_grouped_scan_flat", rc, ctx_);
        return vcol_new(c, key, r);
    }
    // l OP r of two columns of the grouping: in row order when both are (one pass over the whole columns), else in the flat layout
    int vcol_ewise(GroupCtx* c, int op, int l, int r, int ot) {
        const std::array<uint64_t, 6> key{2, (uint64_t)op, (uint64_t)l, (uint64_t)r, (uint64_t)ot, 0};
        auto it = c->memo.find(key);
        if (it != c->memo.end()) return it->second;
        VCol o; o.tag = ot;
        o.layout = (c->vcols[l].layout == 0 && c->vcols[r].layout == 0) ? 0 : 1;
        void *tl = nullptr, *tr = nullptr;
        const void* dl = o.layout ? vcol_flat_ptr(c, l) : vcol_row_ptr(c, l, &tl);
        const void* dr = o.layout ? vcol_flat_ptr(c, r) : vcol_row_ptr(c, r, &tr);
        int rc = aqg_malloc(ctx(), (size_t)c->n * esz(ot) + 16, &o.dptr);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        rc = aqg_ewise(ctx_, op, AQG_VEC_VEC, c->vcols[l].tag, dl, c->vcols[r].tag, dr, ot, o.dptr, c->n);
        if (rc != AQG_OK) die("aqg_ewise", rc, ctx_);
        release(tl); release(tr);
        return vcol_new(c, key, o);
    }
    // column OP scalar (kind: AQG_VEC_SCALAR / AQG_SCALAR_VEC).  A scalar that differs between the groups (`x[val] - min(x[val])`) cannot be
    // answered for all groups at once: returns -1 from the second value on and the caller takes the per-group path
    int vcol_ewise_scalar(GroupCtx* c, int op, int kind, int v, int st, const void* scalar, size_t ssz, int ot) {
        uint64_t bits[2] = {0, 0};
        std::memcpy(bits, scalar, ssz < 16 ? ssz : 16);
        const std::array<uint64_t, 4> key{(uint64_t)op | ((uint64_t)kind << 32), (uint64_t)v, (uint64_t)st, (uint64_t)ot};
        auto it = c->smemo.find(key);
        if (it != c->smemo.end()) {
            if (it->second.varies) return -1;
            if (it->second.bits[0] == bits[0] && it->second.bits[1] == bits[1]) return it->second.vcol;
            it->second.varies = true;
            return -1;
        }
        VCol o; o.tag = ot; o.layout = c->vcols[v].layout;
        void* tv = nullptr;
        const void* dv = o.layout ? c->vcols[v].dptr : vcol_row_ptr(c, v, &tv);
        int rc = aqg_malloc(ctx(), (size_t)c->n * esz(ot) + 16, &o.dptr);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        if (kind == AQG_VEC_SCALAR) rc = aqg_ewise(ctx_, op, kind, c->vcols[v].tag, dv, st, scalar, ot, o.dptr, c->n);
        else rc = aqg_ewise(ctx_, op, kind, st, scalar, c->vcols[v].tag, dv, ot, o.dptr, c->n);
        if (rc != AQG_OK) die("aqg_ewise", rc, ctx_);
        release(tv);
        c->vcols.push_back(o);
        const int id = (int)c->vcols.size() - 1;
        c->smemo[key] = GroupCtx::ScalarUse{{bits[0], bits[1]}, id, false};
        return id;
    }
    void free_vcols(GroupCtx* c) {
        for (auto& v : c->vcols) { if (v.dptr) aqg_free(ctx(), v.dptr); if (v.flat) aqg_free(ctx(), v.flat); }
        c->vcols.clear(); c->memo.clear(); c->smemo.clear(); c->cache.clear(); c->corr_cache.clear();
    }

    // the deferred entry AT host address p (a per-group temporary), or nullptr
    Entry* deferred_at(const void* p) {
        auto it = map_.find((uintptr_t)p);
        return (it != map_.end() && it->second.deferred) ? &it->second : nullptr;
    }
    // host buffer [host_out, +bytes) := group g's slice of column v (nothing runs)
    void defer_slice(void* host_out, size_t bytes, GroupCtx* gc, uint32_t g, int v) {
        forget_range(host_out, bytes);
        Entry e; e.bytes = bytes; e.host_stale = true; e.deferred = true; e.dgroup = gc; e.dg = g; e.dv = v;
        map_[(uintptr_t)host_out] = e;
        ++stale;
    }
    void defer_gather(void* host_out, size_t bytes, GroupCtx* gc, uint32_t g, const void* src, size_t src_bytes, bool src_borrowed, int tag) {
        defer_slice(host_out, bytes, gc, g, vcol_source(gc, src, src_bytes, src_borrowed, tag));
    }
    // a device buffer holding just this group's slice (consumers that know nothing about groupings)
    void materialize(std::map<uintptr_t, Entry>::iterator it) {
        Entry& e = it->second;
        GroupCtx* c = e.dgroup;
        const int v = e.dv;
        void* d = nullptr;
        int rc = aqg_malloc(ctx(), e.bytes ? e.bytes : 16, &d);
        if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
        e.deferred = false;                                // (before input(): the source may be this very registry)
        e.dptr = d;
        const size_t es = esz(c->vcols[v].tag);
        if (c->vcols[v].layout == 1 || c->vcols[v].flat) {
            const char* f = static_cast<const char*>(vcol_flat_ptr(c, v));
            rc = aqg_d2d(ctx_, d, f + (size_t)c->offsets[e.dg] * es, (size_t)c->counts[e.dg] * es);
            if (rc != AQG_OK) die("aqg_d2d", rc, ctx_);
            return;
        }
        void* tmp = nullptr;
        const void* dsrc = vcol_row_ptr(c, v, &tmp);
        void* tmp2 = nullptr;
        const void* drows = input(c->row_ids + c->offsets[e.dg], (size_t)c->counts[e.dg] * 4, false, &tmp2);
        rc = aqg_gather(ctx_, c->vcols[v].tag, dsrc, static_cast<const uint32_t*>(drows), c->counts[e.dg], d);
        if (rc != AQG_OK) die("aqg_gather", rc, ctx_);
        release(tmp); release(tmp2);
    }
    // op(<per-group temporary>) at p: answered from the per-grouping cache (one kernel for all groups)
    bool deferred_reduce(const void* p, int op, void* out16) {
        Entry* ep = deferred_at(p);
        if (!ep) return false;
        Entry& e = *ep;
        GroupCtx* c = e.dgroup;
        auto key = std::make_pair(e.dv, op);
        auto hit = c->cache.find(key);
        if (hit == c->cache.end()) {
            const int tag = c->vcols[e.dv].tag;
            const int ot = aqg_reduce_out_dtype(op, tag);
            const size_t osz = aqg_dtype_size(ot);
            void* dout = nullptr;
            int rc = aqg_malloc(ctx(), (size_t)c->G * 16 + 16, &dout);
            if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
            if (c->vcols[e.dv].layout == 1) {
                rc = aqg_grouped_reduce_flat(ctx_, c->handle, op, tag, c->vcols[e.dv].dptr, dout);
                if (rc != AQG_OK) die("aqg_grouped_reduce_flat", rc, ctx_);
            } else {
                void* tmp = nullptr;
                const void* dsrc = vcol_row_ptr(c, e.dv, &tmp);
                rc = aqg_grouped_reduce(ctx_, c->handle, op, tag, dsrc, dout);
                if (rc != AQG_OK) die("aqg_grouped_reduce", rc, ctx_);
                release(tmp);
            }
            std::vector<unsigned char> packed((size_t)c->G * osz), slots((size_t)c->G * 16, 0);
            rc = aqg_d2h(ctx_, packed.data(), dout, packed.size());
            if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
            for (uint32_t g = 0; g < c->G; ++g) std::memcpy(&slots[(size_t)g * 16], &packed[(size_t)g * osz], osz);
            aqg_free(ctx_, dout);
            hit = c->cache.emplace(key, std::move(slots)).first;
        }
        std::memcpy(out16, &hit->second[(size_t)e.dg * 16], 16);
        return true;
    }
    // corr(<temporary>, <temporary>) of the same group, both row-order columns: aqg_grouped_corr once for all groups
    bool deferred_corr(const void* px, const void* py, double* out) {
        Entry *ex = deferred_at(px), *ey = deferred_at(py);
        if (!ex || !ey || ex->dgroup != ey->dgroup || ex->dg != ey->dg) return false;
        GroupCtx* c = ex->dgroup;
        if (c->vcols[ex->dv].layout != 0 || c->vcols[ey->dv].layout != 0) return false;
        auto key = std::make_pair(ex->dv, ey->dv);
        auto hit = c->corr_cache.find(key);
        if (hit == c->corr_cache.end()) {
            void *tx = nullptr, *ty = nullptr, *dout = nullptr;
            const void* dx = vcol_row_ptr(c, ex->dv, &tx);
            const void* dy = vcol_row_ptr(c, ey->dv, &ty);
            int rc = aqg_malloc(ctx(), (size_t)c->G * 8 + 16, &dout);
            if (rc != AQG_OK) die("aqg_malloc", rc, ctx_);
            rc = aqg_grouped_corr(ctx_, c->handle, c->vcols[ex->dv].tag, dx, c->vcols[ey->dv].tag, dy, static_cast<double*>(dout));
            release(tx); release(ty);
            if (rc == AQG_ERR_DTYPE) { aqg_free(ctx_, dout); return false; }     // (8-byte / floating columns: per group through aqg_corr)
            if (rc != AQG_OK) die("aqg_grouped_corr", rc, ctx_);
            std::vector<double> r(c->G);
            rc = aqg_d2h(ctx_, r.data(), dout, (size_t)c->G * 8);
            if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
            aqg_free(ctx_, dout);
            hit = c->corr_cache.emplace(key, std::move(r)).first;
        }
        *out = hit->second[ex->dg];
        return true;
    }
    // host copy of a per-group temporary: beyond a handful of groups the whole flat column comes down ONCE and every slice is a memcpy
    bool touch_deferred(std::map<uintptr_t, Entry>::iterator it) {
        Entry& e = it->second;
        GroupCtx* c = e.dgroup;
        if (c->G <= 8 && c->vcols[e.dv].layout == 0 && !c->vcols[e.dv].flat) return false;
        const size_t es = esz(c->vcols[e.dv].tag);
        if (!c->vcols[e.dv].host_valid) {
            const void* f = vcol_flat_ptr(c, e.dv);
            c->vcols[e.dv].host.resize((size_t)c->n * es);
            int rc = aqg_d2h(ctx(), c->vcols[e.dv].host.data(), f, (size_t)c->n * es);
            if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
            c->vcols[e.dv].host_valid = true;
        }
        std::memcpy((void*)it->first, c->vcols[e.dv].host.data() + (size_t)c->offsets[e.dg] * es, (size_t)c->counts[e.dg] * es);
        --stale;
        map_.erase(it);
        return true;
    }

    // make the host copy of the buffer containing p valid
    void touch(const void* p) {
        auto it = find(p);
        if (it != map_.end() && it->second.deferred) { if (touch_deferred(it)) return; materialize(it); }
        if (it == map_.end() || !it->second.host_stale) return;
        if (it->second.gctx && !it->second.dptr) fill_rows(it->second);
        int rc = aqg_d2h(ctx(), (void*)it->first, it->second.dptr, it->second.bytes);
        if (rc != AQG_OK) die("aqg_d2h", rc, ctx_);
        it->second.host_stale = false;
        --stale;
        // mutable host access follows: a RESULT's device copy cannot be trusted from here on
        if (!it->second.pinned && !it->second.gctx) { if (it->second.dptr) aqg_free(ctx_, it->second.dptr); map_.erase(it); }
    }
    // write-back: the host copies of several RESULT columns at once -- asynchronous egress of all of them (aqg_col_fetch: page-locked
    // destination, DMA on the copy stream, the copies overlap each other and the tail kernels still queued), ONE wait
    void fetch_all(const std::vector<const void*>& ps) {
        std::vector<std::map<uintptr_t, Entry>::iterator> pending;
        for (const void* p : ps) {
            if (!stale) break;
            auto it = find(p);
            if (it == map_.end() || !it->second.host_stale) continue;
            if (it->second.deferred || (it->second.gctx && !it->second.dptr)) { touch(p); continue; }
            int rc = aqg_col_fetch(ctx(), (void*)it->first, it->second.dptr, it->second.bytes);
            if (rc != AQG_OK) die("aqg_col_fetch", rc, ctx_);
            pending.push_back(it);
        }
        if (pending.empty()) return;
        int rc = aqg_col_fetch_wait(ctx());
        if (rc != AQG_OK) die("aqg_col_fetch_wait", rc, ctx_);
        for (auto it : pending) {
            it->second.host_stale = false;
            --stale;
            if (!it->second.pinned && !it->second.gctx) { if (it->second.dptr) aqg_free(ctx_, it->second.dptr); map_.erase(it); }
        }
    }
    // the host buffer at p is going away / being rewritten by the host
    void forget(const void* p) {
        auto it = map_.find((uintptr_t)p);
        if (it == map_.end()) return;
        if (it->second.host_stale) --stale;
        if (it->second.dptr) aqg_free(ctx(), it->second.dptr);
        map_.erase(it);
    }
    void forget_range(const void* p, size_t bytes) {
        if (map_.empty()) return;
        auto it = map_.lower_bound((uintptr_t)p);
        while (it != map_.end() && it->first < (uintptr_t)p + (bytes ? bytes : 1)) {
            if (it->second.host_stale) --stale;
            if (it->second.dptr) aqg_free(ctx(), it->second.dptr);
            it = map_.erase(it);
        }
    }
    void drop_pins() {
        for (auto it = map_.begin(); it != map_.end();) {
            if (it->second.pinned) { aqg_free(ctx(), it->second.dptr); it = map_.erase(it); } else ++it;
        }
    }
    void sync() { aqg_sync(ctx()); }

    // Groupings made through HashTableFactory::get belong to the SESSION of the module that made them (the reference leaks them:
    // "Memory leak here, cleanup after module is done", hasher.h:255): device handle, row ids, offsets / counts, the key vector and
    // the vecs array are released by Context::end_session() / the module's __AQ_End_Session__ hook / the unload of the module.
    struct SessionItem { GroupCtx* table; void* keys; void (*free_keys)(void*); void* vecs; };
    std::vector<SessionItem> session_items;
    void release_session() {
        for (auto& it : session_items) {
            if (it.table) {
                free_vcols(it.table);
                if (it.table->row_ids) { forget(it.table->row_ids); std::free(it.table->row_ids); }
                if (it.table->handle) aqg_groupby_destroy(it.table->handle);
                std::free(it.table->offsets); std::free(it.table->counts);
                delete it.table;
            }
            if (it.keys && it.free_keys) it.free_keys(it.keys);
            std::free(it.vecs);
        }
        session_items.clear();
    }

private:
    Runtime() = default;
    ~Runtime() {
        release_session();
        if (ctx_) {
            for (auto& kv : map_) if (kv.second.dptr) aqg_free(ctx_, kv.second.dptr);
            aqg_ctx_destroy(ctx_);
        }
    }
    aqg_ctx* ctx_ = nullptr;
    std::map<uintptr_t, Entry> map_;
};

inline void host_touch(const void* p) {
    Runtime& r = Runtime::get();
    if (r.stale) r.touch(p);
}

inline void check(int rc, const char* what) {
    if (rc != AQG_OK) die(what, rc, Runtime::get().ctx());
}

// dtype tag of a C++ element type (reference server/types.h:162-190 mapping)
template <class T> struct tag_of { static constexpr int value = AQG_ERROR; };
#define AQ_TAG(T, V) template <> struct tag_of<T> { static constexpr int value = V; };
AQ_TAG(int, AQG_INT32) AQ_TAG(float, AQG_FLOAT) AQ_TAG(double, AQG_DOUBLE) AQ_TAG(long, AQG_INT64) AQ_TAG(long long, AQG_INT64)
AQ_TAG(short, AQG_INT16) AQ_TAG(signed char, AQG_INT8) AQ_TAG(char, AQG_INT8) AQ_TAG(unsigned char, AQG_UINT8)
AQ_TAG(unsigned short, AQG_UINT16) AQ_TAG(unsigned int, AQG_UINT32) AQ_TAG(unsigned long, AQG_UINT64) AQ_TAG(unsigned long long, AQG_UINT64)
AQ_TAG(bool, AQG_BOOL)
#ifdef __SIZEOF_INT128__
AQ_TAG(__int128, AQG_INT128) AQ_TAG(unsigned __int128, AQG_UINT128)
#endif
#undef AQ_TAG
template <class T> constexpr bool on_device = tag_of<std::remove_cv_t<T>>::value != AQG_ERROR;

// RAII view of a column's device address for the duration of one call
struct In {
    const void* d = nullptr;
    void* temp = nullptr;
    In(const void* host, size_t bytes, bool borrowed) { d = Runtime::get().input(host, bytes, borrowed, &temp); }
    ~In() { Runtime::get().release(temp); }
    In(const In&) = delete;
    In& operator=(const In&) = delete;
};

} // namespace dev
} // namespace aq

/*
 * aq_oracle.h -- CPU restatement of the reference's column-batch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or called
 * by the product (aquery2_amd/, include/aquery/); only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, and only as the checker.
 *
 * Every function is a plain-C restatement of one loop of the reference's
 * "AQuery Library" headers and cites the file:line it follows (paths relative
 * to the reference tree).  The restatement is pinned two ways:
 *   1. tests/golden/ *.json -- vectors dumped from the REAL reference headers
 *      (oracle/ref_harness.cpp compiled against /root/reference into
 *      oracle/_ref/, script oracle/gen_golden.py), including every KAT listed
 *      in SURVEY.md 8c;
 *   2. when oracle/_ref/libaqref.so is present, tests/test_oracle_vs_ref.py
 *      compares oracle and reference on seeded random inputs, bit for bit.
 *
 * The checker ABI (same signatures are exported by oracle/ref_harness.cpp with
 * the prefix aqr_ instead of aqo_): host pointers, dtype tags of include/aqg.h.
 */
#ifndef AQ_ORACLE_H
#define AQ_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifndef AQCHK
#define AQCHK(name) aqo_##name
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* type rules: server/types.h:199-210,264-275 */
int AQCHK(long_type)(int dt);
int AQCHK(fp_type)(int dt);
int AQCHK(coercion)(int dt1, int dt2);

/* free operators server/table.h:820-937 (result dtype = *_out_dtype) and
 * aqop_* server/table.h:954-973 (result dtype `ot` chosen by the caller) */
int AQCHK(ewise_out_dtype)(int op, int lt, int rt);
int AQCHK(ewise)(int op, int kind, int lt, const void* l, int rt, const void* r, int ot, void* out, uint32_t n);
int AQCHK(unary)(int op, int t, const void* x, uint32_t n, uint32_t param, int ot, void* out);

/* reductions server/aggregations.h:10-32,71-86,332-348,413-416,487-497; out = 16 bytes */
int AQCHK(reduce_out_dtype)(int op, int t);
int AQCHK(reduce)(int op, int t, const void* x, uint32_t n, void* out16);
int AQCHK(corr)(int tx, const void* x, int ty, const void* y, uint32_t n, double* out);

/* scans / windows / shifts server/aggregations.h:89-330,350-381,439-485 */
int AQCHK(scan_out_dtype)(int op, int t);
int AQCHK(scan)(int op, int t, const void* x, uint32_t n, uint32_t w, void* out);

/* gather server/table.h:184-189; mask filter :190-198 (selected values only, see D11) */
int AQCHK(gather)(int t, const void* x, const uint32_t* idx, uint32_t m, void* out);
int AQCHK(compact)(int t, const void* x, const uint8_t* mask, uint32_t n, void* out, uint32_t* m);

/* hash of a key / key tuple: server/hasher.h:66-95, server/unordered_dense.h:212-214,279-310 */
uint64_t AQCHK(hash_scalar)(int t, const void* v);
uint64_t AQCHK(hash_tuple)(int nkeys, const int* dts, const void* const* vals);

/* group-by: AQHashTable::hashtable_push per row (server/hasher.h:176-179) then
 * ht_postproc (:181-198).
 *   reversemap[n]  group id per row (dense, first-occurrence order)
 *   counts[G]      ht_base BEFORE postproc (rows per group)  -- caller allocates n entries
 *   offsets[G]     ht_base AFTER postproc (start of group g in row_ids)
 *   row_ids[n]     mapbase: row ids, DESCENDING inside each group
 *   first_rows[G]  row whose key created the group (= keys of values())
 * returns number of groups through *ngroups.                                     */
int AQCHK(groupby)(int nkeys, const int* key_dts, const void* const* keys, uint32_t n,
                   uint32_t* reversemap, uint32_t* ngroups, uint32_t* counts,
                   uint32_t* offsets, uint32_t* row_ids, uint32_t* first_rows);

/* the generated per-group loop (engine/ast.py:722-789): out[g] = op(col[vecs[g]]),
 * i.e. gather (table.h:184-189) in row-id-descending order, then the reduction.
 * out has G elements of reduce_out_dtype(op, t).                                 */
/* key columns of any type the reference hashes (dates, times, timestamps, 128-bit integers, floating columns, astring_view):
 * dense first-occurrence ids under the reference's tuple ==; first_rows[g] = the row that introduced group g */
int AQCHK(groupby_typed)(int nkeys, const int* key_dts, const void* const* keys, uint32_t n,
                         uint32_t* reversemap, uint32_t* ngroups, uint32_t* first_rows);
int AQCHK(grouped_reduce)(int op, int t, const void* x, uint32_t G, const uint32_t* offsets,
                          const uint32_t* counts, const uint32_t* row_ids, void* out);

/* inner equi-join restated with aq_map semantics (no reference implementation: parity unpinned) */
int AQCHK(join_pairs)(int t, const void* build_keys, uint32_t nb, const void* probe_keys, uint32_t np,
                      uint32_t* probe_rows, uint32_t* build_rows, uint64_t cap, uint64_t* m);

/* synthetic columns: SAME generator as aquery2_amd/csrc/gen.hip */
int AQCHK(gen_column)(int col, uint64_t seed, uint64_t row_base, uint32_t n, uint64_t n_total, uint32_t K, void* out);

/* timed reference-shaped Q1/Q5 path for bench.py's cpu_baseline leg:
 * hash build -> ht_postproc -> per-group gather + sum.  Returns seconds.          */
double AQCHK(time_groupby_sum)(int nkeys, const int* key_dts, const void* const* keys,
                               int nvals, const int* val_dts, const void* const* vals, uint32_t n,
                               uint32_t* ngroups_out, double* split3 /* build, postproc, agg */);

#ifdef __cplusplus
}
#endif
#endif

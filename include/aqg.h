/*
 * aqg.h -- C-ABI of the MI355X (gfx950) AQuery execution library.
 *
 * This is the drop-in boundary for the column-batch hot path of the AQuery
 * "AQuery Library" headers (reference: server/vector_type.hpp, server/table.h,
 * server/aggregations.h, server/hasher.h).  Everything above this header is
 * host C++ (include/aquery/ *.h mirror the reference's header-level API);
 * everything below it is hand-written HIP for CDNA4.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types cross this boundary.
 *   - every column pointer is a DEVICE pointer (HBM) unless the parameter is
 *     documented "host".  Sizes are uint32_t like the reference
 *     (server/vector_type.hpp:66: `uint32_t size, capacity`).
 *   - dtype tags are the reference's own (server/aquery_types.h:1-5).
 *   - every entry point returns an int status (0 = AQG_OK).  The reference has
 *     no error channel on this path; a non-zero status here means
 *     "nothing was written, caller may fall back".
 *   - kernels are enqueued on the context's HIP stream.  Entry points that
 *     return a host value synchronise that stream before returning; all others
 *     are asynchronous (call aqg_sync).
 *   - __int128 results (reference: types::GetLongType, server/types.h:205-210)
 *     are written as 16-byte little-endian two's complement.
 */
#ifndef AQG_H
#define AQG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dtype tags: reference server/aquery_types.h:1-5 (same order) -------- */
typedef enum aqg_dtype {
    AQG_INT32 = 0, AQG_FLOAT = 1, AQG_STR = 2, AQG_DOUBLE = 3, AQG_LDOUBLE = 4,
    AQG_INT64 = 5, AQG_INT128 = 6, AQG_INT16 = 7, AQG_DATE = 8, AQG_TIME = 9,
    AQG_INT8 = 10, AQG_UINT32 = 11, AQG_UINT64 = 12, AQG_UINT128 = 13,
    AQG_UINT16 = 14, AQG_UINT8 = 15, AQG_BOOL = 16, AQG_VECTOR = 17,
    AQG_TIMESTAMP = 18, AQG_CHAR = 19, AQG_SV = 20, AQG_NONE = 21, AQG_ERROR = 22
} aqg_dtype;

/* ---- status codes --------------------------------------------------------- */
enum {
    AQG_OK = 0,
    AQG_ERR_HIP = 1,       /* a HIP runtime call failed (see aqg_last_error)  */
    AQG_ERR_DTYPE = 2,     /* dtype / op combination not implemented on device */
    AQG_ERR_ARG = 3,       /* bad argument                                     */
    AQG_ERR_NOMEM = 4,     /* device allocation failed                         */
    AQG_ERR_NODEVICE = 5,  /* no gfx950 device visible                         */
    AQG_ERR_OVERFLOW = 6   /* table / workspace capacity exceeded              */
};

/* Largest supported row count of one column.  Sizes are uint32_t like the reference's (server/vector_type.hpp:66); the tile
 * kernels compute row indices in 32 bits with up to one tile / one grid stride of slack, so counts within 2^20 of 2^32 are
 * rejected with AQG_ERR_ARG instead of wrapping. */
#define AQG_MAX_ROWS 4293918720u   /* 2^32 - 2^20 */

typedef struct aqg_ctx aqg_ctx;

/* ---- type rules (host, pure) ----------------------------------------------
 * Replace the reference's compile-time type functions so that every caller
 * (templates, tests, other languages) derives result dtypes the same way.   */
size_t aqg_dtype_size(int dt);           /* types::AType_sizes, server/types.h:192-193 */
int aqg_long_type(int dt);               /* types::GetLongType, server/types.h:205-210 */
int aqg_fp_type(int dt);                 /* types::GetFPType,   server/types.h:199-204 */
int aqg_coercion(int dt1, int dt2);      /* types::Coercion,    server/types.h:264-275 */

/* ---- context / stream / memory --------------------------------------------
 * One context per (process, GPU).  `stream` may be an existing hipStream_t
 * (e.g. torch's current stream) or NULL to let the library create its own.   */
int aqg_device_count(void);
int aqg_ctx_create(int device, void* hip_stream, aqg_ctx** out);
void aqg_ctx_destroy(aqg_ctx* ctx);
const char* aqg_last_error(aqg_ctx* ctx);
void* aqg_ctx_stream(aqg_ctx* ctx);
int aqg_sync(aqg_ctx* ctx);
/* pre-size the internal workspace arena (block partials, hash tables) so that
 * later calls never allocate: call once before a timed / graph-captured region */
int aqg_reserve_workspace(aqg_ctx* ctx, size_t bytes);

/* Replaces malloc/GC::reg/ScratchSpace for device temporaries
 * (server/vector_type.hpp:70-80,367-371; server/gc.h:7-41).                  */
int aqg_malloc(aqg_ctx* ctx, size_t bytes, void** dptr);
int aqg_free(aqg_ctx* ctx, void* dptr);
int aqg_h2d(aqg_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int aqg_d2h(aqg_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* device-to-device copy on the context's stream (vector_type::subvec_memcpy / the copy constructors, vector_type.hpp:83-135,228-240) */
int aqg_d2d(aqg_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
int aqg_memset(aqg_ctx* ctx, void* dst_dev, int byte, size_t bytes);

/* Device mirror of a borrowed host column (the ColRef<T>(len, server->getCol(i)) binding, engine/ast.py:367-370; the data
 * source hands out zero-copy pointers into its own result memory, server/monetdb_conn.cpp:203-224): on first sight of
 * (host_ptr, bytes) the column is uploaded, afterwards the cached device pointer is returned; aqg_col_unpin_all drops the
 * cache (the reference's per-dll session end).  The upload is ASYNCHRONOUS: the host range is page-locked chunk by chunk and
 * copied by DMA on a copy stream while the call returns; the context's stream waits for its completion, so every later call
 * of this library (and aqg_sync) is ordered behind it.  The host memory must stay valid and unchanged until then -- the
 * reference's borrowed columns live as long as the query.  PCIe line rate (56-57 GB/s measured on the MI355X box).          */
int aqg_col_pin(aqg_ctx* ctx, const void* host_ptr, size_t bytes, void** dptr);
/* how the chunks of the most recent upload of this context travelled: page-locked + DMA; staged through the library's own pinned buffers
 * (the chunk touches pages that are already page-locked -- by another column or context of this library, or by somebody else, which a
 * direct copy or a second registration must not touch: profiles/r3_hostregister_abort.md); plain pageable copy (short columns)       */
int aqg_col_pin_last(aqg_ctx* ctx, uint32_t* registered_chunks, uint32_t* staged_chunks, uint32_t* pageable_chunks);
/* Egress of a result column into caller-owned host memory (the write-back half of the seam: TableInfo::monetdb_append_table hands the
 * data source pointers to the result columns, server/table_ext_monetdb.hpp:34-87): asynchronous, ordered behind everything queued on the
 * context's stream, destination page-locked chunk by chunk + DMA on the copy stream; several columns overlap each other and the query's
 * tail kernels.  aqg_col_fetch_wait: every fetch of this context is complete and its pages are unlocked.                            */
int aqg_col_fetch(aqg_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int aqg_col_fetch_wait(aqg_ctx* ctx);
int aqg_col_unpin(aqg_ctx* ctx, const void* host_ptr);
int aqg_col_unpin_all(aqg_ctx* ctx);

/* ---- element-wise column arithmetic / compare ------------------------------
 * Replaces the free operators server/table.h:820-937 and aqop_* :954-973.
 * ret[i] = l[i] OP r[i] evaluated in the C++ usual-arithmetic-conversion type
 * of (lt, rt) exactly as the reference loop body does, then converted to `ot`.
 * `ot` is the caller's result dtype: Coercion for + and -, GetLongType for *,
 * GetFPType for / (so int32/int32 is an INTEGER quotient stored as float),
 * AQG_BOOL for comparisons.                                                   */
typedef enum aqg_binop {
    AQG_OP_ADD = 0, AQG_OP_SUB = 1, AQG_OP_MUL = 2, AQG_OP_DIV = 3, AQG_OP_MOD = 4,
    AQG_OP_AND = 5, AQG_OP_OR = 6, AQG_OP_XOR = 7,
    AQG_OP_GT = 8, AQG_OP_LT = 9, AQG_OP_GE = 10, AQG_OP_LE = 11, AQG_OP_EQ = 12, AQG_OP_NE = 13
} aqg_binop;
enum { AQG_VEC_VEC = 0, AQG_VEC_SCALAR = 1, AQG_SCALAR_VEC = 2 };
/* for *_SCALAR kinds the scalar side is a HOST pointer to one value of its dtype */
int aqg_ewise(aqg_ctx* ctx, int op, int kind, int lt, const void* l, int rt, const void* r,
              int ot, void* out, uint32_t n);
/* result dtype of the reference's FREE operator for (op, lt, rt): table.h:779-818 */
int aqg_ewise_out_dtype(int op, int lt, int rt);

/* sqrt (server/aggregations.h:34-46) / truncate (:57-69) */
typedef enum aqg_unop { AQG_UN_SQRT = 0, AQG_UN_TRUNCATE = 1 } aqg_unop;
int aqg_unary(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, uint32_t param, int ot, void* out);

/* ---- full-column reductions: server/aggregations.h:10-32,71-86,332-348,383-416,487-497
 * `out_host` receives 16 bytes: SUM -> GetLongType (int128/uint128/double),
 * MIN/MAX/FIRST/LAST -> T, COUNT -> uint64, AVG/VAR/STDDEV -> double.
 * Reference quirks kept on purpose: max seeds with numeric_limits<T>::min()
 * (:73, D8), var divides by len+1 (:347, D9).                                 */
typedef enum aqg_redop {
    AQG_RED_SUM = 0, AQG_RED_MIN = 1, AQG_RED_MAX = 2, AQG_RED_COUNT = 3, AQG_RED_AVG = 4,
    AQG_RED_VAR = 5, AQG_RED_STDDEV = 6, AQG_RED_FIRST = 7, AQG_RED_LAST = 8
} aqg_redop;
int aqg_reduce(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, void* out_host16);
int aqg_reduce_out_dtype(int op, int t);
/* asynchronous form: 16-byte result left in device memory */
int aqg_reduce_dev(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, void* out_dev16);
/* corr(x, y): server/aggregations.h:383-407 */
int aqg_corr(aqg_ctx* ctx, int tx, const void* x, int ty, const void* y, uint32_t n, double* out_host);

/* ---- prefix scans, sliding windows, shifts ---------------------------------
 * sums/avgs/mins/maxs  server/aggregations.h:89-125,203-236
 * sumw/avgw/minw/maxw/ratiow :127-191,238-281 ; varw/stddevw :283-330 (D9, unpinned)
 * deltas/prev/aggnext :439-485 ; ratios = ratiow(1) :193-201 ; vars/stddevs :350-381
 * window = elements [i-w+1, i]; growing prefix for i < w.                      */
typedef enum aqg_scanop {
    AQG_SCAN_SUMS = 0, AQG_SCAN_AVGS = 1, AQG_SCAN_MINS = 2, AQG_SCAN_MAXS = 3,
    AQG_SCAN_SUMW = 4, AQG_SCAN_AVGW = 5, AQG_SCAN_MINW = 6, AQG_SCAN_MAXW = 7,
    AQG_SCAN_RATIOW = 8, AQG_SCAN_DELTAS = 9, AQG_SCAN_PREV = 10, AQG_SCAN_NEXT = 11,
    AQG_SCAN_VARS = 12, AQG_SCAN_STDDEVS = 13, AQG_SCAN_VARW = 14, AQG_SCAN_STDDEVW = 15
} aqg_scanop;
int aqg_scan(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, uint32_t w, void* out);
int aqg_scan_out_dtype(int op, int t);
/* sums / avgs of one row-range shard of a column: rows [row_offset, row_offset + n) of the whole column.
 * `carry_host16` = the sum of every earlier row in the result's LongType (server/types.h:152-160): 16 bytes holding an
 * __int128 / unsigned __int128 for integer columns, a double in the first 8 bytes for floating ones; NULL = nothing before.
 * Element i is carry + x[0..i] (sums, aggregations.h:89-103) or that over (row_offset + i + 1) (avgs, :105-125), i.e. the
 * rows the whole-column scan would have produced.  Integer results are bit-identical to the unsharded scan.  */
int aqg_scan_resume(aqg_ctx* ctx, int op, int t, const void* x, uint32_t n, const void* carry_host16, uint64_t row_offset, void* out);

/* ---- gather / mask filter ---------------------------------------------------
 * ColRef::operator[](vector_type<uint32_t>&)  server/table.h:184-189
 * ColRef::operator[](const std::vector<bool>&) server/table.h:190-198 (as a true
 * stream compaction; the reference's N-junk prefix, defect D11, is not kept)   */
int aqg_gather(aqg_ctx* ctx, int t, const void* x, const uint32_t* idx, uint32_t m, void* out);
int aqg_compact(aqg_ctx* ctx, int t, const void* x, const uint8_t* mask, uint32_t n,
                void* out, uint32_t* m_host);
/* row ids of set mask entries, ascending (selection vector for multi-column filters) */
int aqg_mask_to_index(aqg_ctx* ctx, const uint8_t* mask, uint32_t n, uint32_t* idx_out, uint32_t* m_host);

/* ---- hash group-by -----------------------------------------------------------
 * Replaces AQHashTable (server/hasher.h:146-199) + set::hashtable_push
 * (server/unordered_dense.h:1117-1147) + HashTableFactory::get (:327-357).
 * Contract (the only executable one in the reference):
 *   group ids are dense, numbered by FIRST OCCURRENCE of the key tuple;
 *   ht_postproc row-id lists are DESCENDING row id within each group.
 * Up to 8 key columns; tuples of up to 8 bytes are packed into one word, wider ones compare through a representative row.       */
typedef struct aqg_groupby aqg_groupby;
int aqg_groupby_build(aqg_ctx* ctx, int nkeys, const int* key_dtypes, const void* const* keys,
                      uint32_t n, uint32_t max_groups_hint, aqg_groupby** out);
void aqg_groupby_destroy(aqg_groupby* g);
uint32_t aqg_groupby_ngroups(const aqg_groupby* g);
uint32_t aqg_groupby_nrows(const aqg_groupby* g);
/* device views owned by the handle (valid until destroy) */
const uint32_t* aqg_groupby_reversemap(const aqg_groupby* g); /* [n] group id of row      (hasher.h:149 reversemap) */
const uint32_t* aqg_groupby_counts(const aqg_groupby* g);     /* [G] rows per group       (ht_base before postproc) */
const uint32_t* aqg_groupby_first_rows(const aqg_groupby* g); /* [G] first row of group                            */
/* key column k of every group, in group order (AQHashTable::values()) */
int aqg_groupby_keys(aqg_groupby* g, int k, void* out_dev);
/* ht_postproc (hasher.h:181-198): offsets[G+1] (exclusive scan of counts; the
 * reference's ht_base after postproc is offsets[0..G)), row_ids[n] descending   */
int aqg_groupby_postproc(aqg_groupby* g, uint32_t* offsets_dev, uint32_t* row_ids_dev);

/* Key columns need not be plain integers (the reference hashes astring_view, date_t, time_t, timestamp_t and 128-bit integers too,
 * server/hasher.h:97-144, and groups by tuple ==): AQG_DATE (4-byte elements), AQG_TIME (8-byte elements, the 8th byte is padding
 * and ignored), AQG_TIMESTAMP (12-byte elements), AQG_INT128 / AQG_UINT128, and AQG_FLOAT / AQG_DOUBLE -- by value like the
 * reference's ==: 0.0 and -0.0 are one group, every NaN is a group of its own (probed against the reference headers,
 * tests/golden).  aqg_groupby_keys returns, for such a column, the caller's element at every group's first row (the column
 * must still be alive).  `const char*` keys are pointers in the reference (8-byte integers: pass AQG_UINT64); astring_view keys
 * compare string contents: aqg_str_encode turns the host strings into a uint32 code column (dense ids in first-occurrence order).  */
int aqg_str_encode(aqg_ctx* ctx, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_host);
/* the same over a table sharded by row range (aqg_comm_* below): codes of ONE dictionary over all shards in GLOBAL first-occurrence order,
 * the uint32 key column aqg_groupby_agg_sharded takes -- every rank encodes its rows, the ranks all-gather their dictionaries' strings
 * (two small collectives), merge them in rank order and remap their code columns on the device                                        */
struct aqg_comm;
int aqg_str_encode_sharded(struct aqg_comm* comm, const char* const* strs_host, uint32_t n, uint32_t* codes_dev, uint32_t* ndistinct_global_host);

/* one aggregate over all groups in ONE pass over the value column: the device
 * form of the generated per-group loop `out[g] = op(col[vecs[g]])`
 * (engine/ast.py:722-789, mem_opt.cpp:50-65).  Output dtype as aqg_reduce.     */
int aqg_grouped_reduce(aqg_ctx* ctx, const aqg_groupby* g, int op, int t, const void* x, void* out_dev);

/* ---- per-group scans / windows / two-column aggregates: what the reference's own queries put inside the generated group loop -------
 * `SELECT sym, avgs(5, price) ... ASSUMING ASC time GROUP BY sym` (benchmark/quries/Aquery/q7.a), `mins(2, sales) ... group by Mont`
 * (tests/moving_avg.a:13), `max(ratios(x)) ... group by ID` (tests/q4.a:23), `pow(corr(v1, v2), 2) BY id2, id4`
 * (benchmark/h2o/groupby.sql:20) are emitted as  for g: out[g] = f(col[vecs[g]], ...)  (engine/ast.py:722-789); the frozen sample
 * mem_opt.cpp:53-63 writes vector results into ONE flat buffer sliced by the group offsets: `col[i].init_from(vecs[i].size, buf + offsets[i])`.
 * The FLAT LAYOUT of a build is that buffer's: position offsets[g] + i <-> element i of vecs[g] (ht_postproc order: DESCENDING row id
 * inside a group, hasher.h:192-196).  Every call below handles ALL groups in a number of launches that does not depend on the group count.
 *   aqg_groupby_offsets      device offsets[G+1] (exclusive scan of the group sizes; owned by the handle)
 *   aqg_grouped_flatten      out_flat[offsets[g] + i] = x[vecs[g][i]]: a column (1-, 2-, 4-, 8-byte elements) brought into the flat layout by
 *                            value-carrying radix passes over the group ids (no row lists, no gather)
 *   aqg_grouped_scan_flat    out_flat = the scan `op` (aqg_scan's ops, window w) of every group's slice of xflat, restarted at every
 *                            group start; result dtype aqg_scan_out_dtype.  Integer results are exact, as for aqg_scan.
 *   aqg_grouped_scan         flatten + scan_flat in one call (x in row layout)
 *   aqg_grouped_reduce_flat  out[g] = op(xflat[offsets[g] .. offsets[g+1])): reductions OF scan results (aqg_reduce's ops and dtypes)
 *   aqg_grouped_corr         out[g] = corr(x[vecs[g]], y[vecs[g]]) (aggregations.h:383-407: five 128-bit sums, the products in the C++ type
 *                            of the operands), x / y in ROW layout, integer columns of up to four bytes; doubles in out_dev[G]           */
const uint32_t* aqg_groupby_offsets(aqg_groupby* g);
int aqg_grouped_flatten(aqg_ctx* ctx, aqg_groupby* g, int t, const void* x, void* out_flat);
int aqg_grouped_scan_flat(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* xflat, uint32_t w, void* out_flat);
int aqg_grouped_scan(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* x, uint32_t w, void* out_flat);
int aqg_grouped_reduce_flat(aqg_ctx* ctx, aqg_groupby* g, int op, int t, const void* xflat, void* out_dev);
int aqg_grouped_corr(aqg_ctx* ctx, aqg_groupby* g, int tx, const void* x, int ty, const void* y, double* out_dev);

/* fused single-pass group-by + aggregates (h2o Q1..Q5 shape): reads each key and value
 * column exactly once.  Group order = first occurrence, as aqg_groupby_build.  Result j
 * (aqg_groupby_agg_result) has ngroups elements of aqg_reduce_out_dtype(ops[j], val_dtypes[j])
 * and lives in the handle, like keys / first_rows (and counts when an op needs them); no
 * reversemap / postproc on such a handle.  ops: SUM MIN MAX COUNT AVG VAR STDDEV.
 * `*out` is in/out: pass NULL to create a handle, or an earlier handle to reuse its buffers
 * (steady-state calls then allocate nothing).  max_groups_hint sizes the hash tables
 * (0 = unknown: start small and grow); the call retries internally when the hint is too low.
 * Floating-point SUM / AVG / VAR accumulate with atomic adds (LDS and HBM), in whatever order the hardware schedules the rows:
 * results are NOT bit-reproducible from run to run (the reference adds the rows of a group in one fixed order); every result lies
 * within (n_g - 1) 2^-53 sum|x| of the exactly rounded sum of its group (n_g rows), which is what the parity tests assert.
 * Integer aggregates, counts, MIN / MAX, keys, first rows and the group order are exact and deterministic.
 * On return the handle's ngroups is final; the device columns behind the handle (keys, first rows,
 * counts, results) may still be being written by kernels queued on the context's stream: read
 * them through this library (aqg_groupby_keys, aqg_d2h, ...: all ordered behind those kernels),
 * from the same stream, or after aqg_sync.                                                       */
int aqg_groupby_agg(aqg_ctx* ctx, int nkeys, const int* key_dtypes, const void* const* keys,
                    int naggs, const int* ops, const int* val_dtypes, const void* const* vals,
                    uint32_t n, uint32_t max_groups_hint, aqg_groupby** out);
const void* aqg_groupby_agg_result(const aqg_groupby* g, int j);
/* diagnostic: which plan the last aqg_groupby_agg / aqg_groupby_build through this handle took; the tests pin every
 * plan with it, the product does not look at it                                                                                         */
enum { AQG_PLAN_FAST_LDS = 1, AQG_PLAN_SMALL_LDS = 2, AQG_PLAN_BIG_LDS = 4, AQG_PLAN_DENSE = 8, AQG_PLAN_PART_ONE = 16, AQG_PLAN_PART_TWO = 32,
       AQG_PLAN_PART_ROUND1 = 64, AQG_PLAN_PART_WIDE = 128, AQG_PLAN_SORTED_TAIL = 256, AQG_PLAN_HBM_TABLE = 512, AQG_PLAN_BUILD_PARTITIONED = 1024, AQG_PLAN_GID_PARTITION = 2048 /* aqg_grouped_reduce: rows partitioned on the dense group id */,
       AQG_PLAN_PACKED_VALUES = 4096 /* narrow value columns travelled inside the 4-byte key word of a two-level partition plan */,
       AQG_PLAN_RANGE_PARTITIONS = 8192 /* a dense 4-byte key domain: order-preserving range partitions, direct-indexed LDS accumulators */,
       AQG_PLAN_ROW_EMIT = 16384 /* every row turned out to be its own group: the result columns are written as a map of the input */,
       AQG_PLAN_PACKED_KEYS = 32768 /* the key columns of a wide tuple travelled packed into fewer dword planes (sampled ranges, every row verified) */,
       AQG_PLAN_BUILD_LOOKUP = 65536 /* aqg_groupby_build: row ids through a key -> group id look-up table (a dense key domain of up to 2^21 values) */ };
uint32_t aqg_groupby_plan(const aqg_groupby* g);

/* ---- hash join (new functionality; reference runs joins in MonetDB)
 * inner equi-join on one integer key: build on (build_keys, nb), probe with
 * (probe_keys, np).  Emits matching row-id pairs ordered by probe row, then by
 * build row ascending.  Two-call protocol: pass NULL outputs to get the count.  The count is exact in 64 bits (duplicate keys
 * pass 2^32 pairs at small inputs: 70,000 x 70,000 equal keys); aqg_join_pairs addresses its outputs with uint32 offsets and
 * returns AQG_ERR_OVERFLOW (with *m_host = the count, nothing written) beyond AQG_MAX_ROWS pairs.                          */
int aqg_join_count(aqg_ctx* ctx, int t, const void* build_keys, uint32_t nb,
                   const void* probe_keys, uint32_t np, uint64_t* m_host);
int aqg_join_pairs(aqg_ctx* ctx, int t, const void* build_keys, uint32_t nb,
                   const void* probe_keys, uint32_t np,
                   uint32_t* probe_rows_out, uint32_t* build_rows_out, uint64_t capacity, uint64_t* m_host);
/* unique-build-key lookup join fused with nothing: idx[i] = build row whose key
 * equals probe_keys[i], or 0xFFFFFFFF                                            */
int aqg_join_lookup(aqg_ctx* ctx, int t, const void* build_keys, uint32_t nb,
                    const void* probe_keys, uint32_t np, uint32_t* build_row_of_probe);

/* ---- the exchange step of row-sharded group-bys -------------------------------------------------
 * Tables shard by row range, one process per GPU; every shard groups its own rows and the shards' group tables are merged
 * by ONE all_gather (RCCL) of a fixed-size payload followed by a re-aggregation.  Because shards are contiguous row
 * ranges gathered in rank order, first occurrence in the concatenation is the global first occurrence.
 *   aqg_groupby_pack          writes the payload of this shard: (gmax + 1) int64 pairs -- {ngroups, 0} then per group
 *                             {key, low 64 bits of aggregate `agg_index`} (one key column; integer SUM / COUNT / MIN / MAX)
 *   aqg_groupby_merge_packed  takes the `world` gathered payloads (rank order, (gmax + 1) * 2 int64 each) and returns the
 *                             merged group-by: keys of `key_dtype`, one aggregate combined with `op` (COUNT -> SUM of counts),
 *                             result dtype as for an AQG_INT64 value column
 * (the generated code of the reference has no distributed form; this replaces nothing there)                       */
int aqg_groupby_pack(aqg_groupby* g, int agg_index, uint32_t gmax, int64_t* out_dev);
int aqg_groupby_merge_packed(aqg_ctx* ctx, const int64_t* gathered_dev, uint32_t world, uint32_t gmax, int key_dtype, int op,
                             aqg_groupby** out);

/* ---- the exchange inside the library: communicators and the sharded group-by ----------------------------------------
 * One communicator per (process, GPU).  aqg_comm_init_rccl: RCCL -- rank 0 makes an id with aqg_comm_unique_id (ncclGetUniqueId),
 * the host ships its AQG_COMM_ID_BYTES bytes to the other ranks by any side channel (file, socket, torch.distributed, MPI) and
 * every rank calls aqg_comm_init_rccl (ncclCommInitRank on the context's device); the all-gather is ncclAllGather on the
 * context's stream, over xGMI inside a node.  librccl is opened on first use (dlopen), single-GPU users never load it.
 * aqg_comm_init_custom: the caller supplies the all-gather (`bytes` bytes of every rank's device buffer `send_dev` into
 * `recv_dev`, rank order, ordered on `hip_stream` or complete on return; 0 = success) -- hosts with a transport of their own,
 * and the one-GPU rehearsals / tests of the sharded path.                                                                          */
typedef struct aqg_comm aqg_comm;
#define AQG_COMM_ID_BYTES 128
typedef int (*aqg_allgather_fn)(void* user, const void* send_dev, void* recv_dev, size_t bytes, void* hip_stream);
int aqg_comm_unique_id(void* id_out /* AQG_COMM_ID_BYTES, host */);
int aqg_comm_init_rccl(aqg_ctx* ctx, int rank, int world, const void* id, aqg_comm** out);
int aqg_comm_init_custom(aqg_ctx* ctx, int rank, int world, aqg_allgather_fn fn, void* user, aqg_comm** out);
void aqg_comm_destroy(aqg_comm* comm);
int aqg_comm_rank(const aqg_comm* comm);
int aqg_comm_world(const aqg_comm* comm);
/* aqg_groupby_agg over a table sharded by ROW RANGE: this rank holds rows [row_base, row_base + n) of every column.  Every rank
 * groups its own rows, ONE all-gather moves the shards' group tables (k key columns, the global first row and one partial per
 * aggregate: SUM -> sum, COUNT -> count, MIN / MAX -> itself, AVG -> sum and count, VAR / STDDEV -> sum, sum of squares and count;
 * a partial that needs 128 bits per shard -- sums of 8-byte integers, sums of squares -- as two 8-byte columns), every rank re-aggregates the concatenation
 * and gets the same merged result: keys / aggregates of all groups in GLOBAL first-occurrence order (shards are contiguous and
 * gathered in rank order, so first occurrence in the concatenation is the global one: server/hasher.h:176-198 semantics).
 * Integer sums are exact (128-bit results), AVG of integers is the exact sum over the count, as in the single-GPU call.
 * gmax: upper bound of a shard's group count (fixed-size payload, nothing but the one all-gather crosses the wire: h2o Q1 / Q4 /
 * config 4), or 0: the ranks first exchange their group counts (one 8-byte all-gather) and size the payload by the largest.
 * ops: SUM / COUNT / MIN / MAX / AVG / VAR / STDDEV over integer columns of 1 to 8 bytes and floating columns (VAR / STDDEV use the
 * reference's formula on the merged moments, bit-identical to the single-GPU call for integers; up to 8 distinct partials per call).
 * The result handle has no 32-bit first rows (aqg_groupby_first_rows returns NULL):
 * aqg_groupby_first_rows64 gives the global row id of every group's first row.  `*out` is in/out like aqg_groupby_agg.             */
int aqg_groupby_agg_sharded(aqg_comm* comm, int nkeys, const int* key_dtypes, const void* const* keys,
                            int naggs, const int* ops, const int* val_dtypes, const void* const* vals,
                            uint32_t n, uint64_t row_base, uint32_t max_groups_hint, uint32_t gmax, aqg_groupby** out);
const int64_t* aqg_groupby_first_rows64(const aqg_groupby* g);
/* The exchange alone, over a shard table the caller already has (any aqg_groupby_agg / aqg_join_groupby_sum handle whose first
 * `nparts` aggregates are decomposable partials): partial p combines with merge_ops[p] (SUM / MIN / MAX; counts and sums combine
 * with SUM).  128-bit integer sums travel as their low 64 bits (exact while a SHARD's sum fits 64 bits) and come back as the exact
 * 128-bit total; counts come back as 128-bit totals.  Result: keys, aqg_groupby_first_rows64 and aggregates 0 .. nparts-1 of the
 * merged table in global first-occurrence order, as aqg_groupby_agg_sharded.                                                      */
int aqg_groupby_exchange(aqg_comm* comm, aqg_groupby* local, int nparts, const int* merge_ops, uint64_t row_base, uint32_t gmax, aqg_groupby** out);

/* ---- reductions and scans of a column sharded by ROW RANGE ---------------------------------------------------------
 * This rank holds `n` rows (0 allowed) of a column whose shards, laid end to end in rank order, are the whole column.  Every call
 * makes ONE all-gather of a small record per rank -- raw moments, first / last row, the shard's last w rows -- and answers as the
 * single-GPU call over the whole column would (server/aggregations.h:19-32,71-86,332-407; :89-281,439-485):
 *   aqg_reduce_sharded   SUM / MIN / MAX / COUNT / AVG / VAR / STDDEV / FIRST / LAST; the same 16 bytes on every rank; integer results
 *                        bit-identical to aqg_reduce over the whole column, floating sums = the rank-order sum of the shards' sums
 *   aqg_corr_sharded     corr(x, y): the five 128-bit sums folded exactly (integer columns, as aqg_corr)
 *   aqg_scan_sharded     out = this rank's rows of aqg_scan over the whole column: sums / avgs resume from the earlier shards' exact
 *                        total and row count, mins / maxs from their min / max, windows and shifts take the last rows of the shards
 *                        before them (neighbour rows for deltas / prev / aggnext).  vars / stddevs are not offered.
 * A rank whose local part fails still joins the all-gather and EVERY rank returns its status.                                      */
int aqg_reduce_sharded(aqg_comm* comm, int op, int t, const void* x, uint32_t n, void* out_host16);
int aqg_corr_sharded(aqg_comm* comm, int tx, const void* x, int ty, const void* y, uint32_t n, double* out_host);
int aqg_scan_sharded(aqg_comm* comm, int op, int t, const void* x, uint32_t n, uint32_t w, void* out);

/* Fused star join + grouped sum (BASELINE config 4: `fact JOIN small(key, w) ON fact.fk = small.key`, then
 * `sum(fact.val * small.w) BY fact.gkey`): one pass over fk, gkey and val (12 B/row) instead of lookup -> gather ->
 * multiply -> group-by (44 B/row).  The reference emits this as SQL for MonetDB (engine/ast.py:874-1085) followed by
 * the generated group loop (engine/ast.py:722-789); `val * w` is the free operator* of table.h:866-876 (exact product in
 * GetLongType) and the sum is aggregations.h:62-70.  All five columns are 4-byte integers; the dimension side has at
 * most 4096 rows (it lives in LDS), unique keys (of duplicates the lowest row wins, as in aqg_join_lookup); fact rows
 * without a partner are dropped (inner join).  Result: a group-by handle whose keys / first rows are in first-occurrence
 * order among the JOINED rows and whose aqg_groupby_agg_result(h, 0) is the 128-bit sum per group.  At most 3072 groups. */
int aqg_join_groupby_sum(aqg_ctx* ctx, int key_dtype, const void* dim_keys, int dim_val_dtype, const void* dim_vals, uint32_t nb,
                         const void* fact_fk, int group_key_dtype, const void* group_keys, int val_dtype, const void* fact_vals,
                         uint32_t n, uint32_t max_groups_hint, aqg_groupby** out);

/* ---- synthetic h2o / time-series columns (bench + parity inputs) ---
 * Counter-based: row i of column `col` depends only on (seed, col, row_base+i),
 * so shards are reproducible independent of the GPU count.  The oracle carries
 * the same generator for the CPU side.                                          */
typedef enum aqg_gencol {
    AQG_GEN_ID1 = 0, AQG_GEN_ID2 = 1, AQG_GEN_ID3 = 2, AQG_GEN_ID4 = 3, AQG_GEN_ID5 = 4, AQG_GEN_ID6 = 5,
    AQG_GEN_V1 = 6, AQG_GEN_V2 = 7, AQG_GEN_V3 = 8, AQG_GEN_TIMESTAMP = 9, AQG_GEN_PRICE = 10
} aqg_gencol;
int aqg_gen_column(aqg_ctx* ctx, int col, uint64_t seed, uint64_t row_base, uint32_t n,
                   uint64_t n_total, uint32_t K, void* out_dev);

/* ---- HIP-event timing on the context's stream (bench.py roofline leg) ------ */
int aqg_timer_start(aqg_ctx* ctx);
int aqg_timer_stop_ms(aqg_ctx* ctx, float* ms_host);
/* duration of the DOMINANT kernel of the most recent call (group-by: the pass over the rows;
 * scans: the scan pass), bracketed by HIP events on the context's stream inside the library */
int aqg_last_kernel_ms(aqg_ctx* ctx, float* ms_host);

const char* aqg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* AQG_H */

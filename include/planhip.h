/*
 * planhip.h — C ABI of libplanhip.so, the MI355X (gfx950) execution backend for the hot path of
 * daviszhen/plan:  PhysicalScan -> Filter -> HashJoin build/probe -> HashAggregate
 * (pkg/compute executors over pkg/chunk vectors).
 *
 * This is the boundary a cgo shim binds (INTEGRATION.md shows the Go side). It replaces, inside
 * the reference's `buildOperatorExec` switch (pkg/compute/executor.go:305-350), the bodies of
 *   scanExecutor.Execute / runFilterExec        pkg/compute/executor_scan.go:144-241
 *   filterExecutor.Execute                      pkg/compute/executor_filter.go:27-114
 *   joinExecutor.Execute (build + probe)        pkg/compute/executor_join.go:54-264
 *   aggExecutor.Execute (sink + finalize)       pkg/compute/executor_aggr.go:106-265
 * while the OperatorExec interface itself (executor_operator.go:52-56) and pkg/chunk stay as they
 * are. Plain pointers and sizes only; no C++ or torch types cross it.
 *
 * Conventions
 *  - Every function returns 0 (PH_OK) or a negative PH_E* code; ph_last_error() gives the
 *    thread-local message. Nothing aborts: the shim turns a code into a Go `error`, which is how
 *    the reference reports operator failures (panic -> recover -> error, executor_bench.go:184-204).
 *  - A ph_ctx is bound to one HIP device and one stream. Calls on one ctx are stream-ordered and
 *    must come from one thread at a time (the reference drives a query from one goroutine).
 *    Separate ctxs are independent (concurrent queries of the psql server path).
 *  - Column encodings on the device = what the reference's loader reads from
 *    parquet (executor_scan.go:410-466): INTEGER int32, BIGINT int64, DATE int32 days since
 *    1970-01-01, DECIMAL int64 unscaled (+ scale), VARCHAR with <=256 distinct values as uint8
 *    dictionary codes, other VARCHAR as int32 offsets + bytes.
 *  - Validity is pkg/util/bitmap.go's: 1 bit per row, LSB first, NULL pointer = all valid.
 *  - Row ids / selection vectors on the device are int32 (a device batch is < 2^31 rows); the
 *    shim widens them to pkg/chunk's `[]int` (select_vector.go:7-9) when it hands them back.
 *  - "dev" pointers are HIP device pointers owned by the caller unless stated otherwise.
 */
#ifndef PLANHIP_H
#define PLANHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PH_OK 0
#define PH_EINVAL (-1)      /* bad argument */
#define PH_EHIP (-2)        /* a HIP runtime call failed */
#define PH_EUNSUPPORTED (-3)/* expression / type shape outside the device path: caller falls back */
#define PH_EOVERFLOW (-4)   /* decimal arithmetic would leave the exact int64/int128 domain */
#define PH_ECAPACITY (-5)   /* an output buffer supplied by the caller is too small */
#define PH_ECONSTRAINT (-6) /* a strict lookup met a probe row without exactly one build row */

typedef struct ph_ctx ph_ctx;

const char *ph_last_error(void);
const char *ph_version(void);

int ph_ctx_create(int device, ph_ctx **out);
/* Use an existing stream (e.g. the caller's torch stream) for all later calls; NULL = own stream */
int ph_ctx_set_stream(ph_ctx *ctx, void *hip_stream);
int ph_ctx_sync(ph_ctx *ctx);
/* Deferred errors. A call that only reads a FLAG back (ph_expr_eval's overflow flag) stalls the
 * stream for a host round trip (~25-45 us of idle GPU) although the flag is almost never set. With
 * `on` != 0 such flags stay on the device and the NEXT call on this ctx that reads anything back
 * (a count, a result download, ph_ctx_check_deferred) fetches them in the same synchronisation and
 * returns the deferred error (PH_EOVERFLOW / PH_ECONSTRAINT, ph_last_error names the origin)
 * instead of its own success: everything computed since the failing call must be discarded, which
 * is what a query does with PH_EOVERFLOW anyway (it falls back as a whole). Off by default: every
 * call then reports its own errors, like the reference's operators do per chunk.
 * `on` == 2 HOLDS deferred errors: no ordinary read-back reports them, only ph_ctx_check_deferred does.
 * That is the mode of a query that runs over several ranks: a rank must not leave the sequence of
 * collectives in the middle because of an error only it has seen (its peers would block in the next
 * collective) — it runs to the end, checks, and the ranks agree on the outcome (one small all-reduce)
 * before any of them reruns the query. The ph_comm_* calls never report a deferred error. */
int ph_ctx_set_deferred_errors(ph_ctx *ctx, int32_t on);
/* Asynchronous counts. With `on` != 0 the calls that hand a row count back to the host
 * (ph_filter_select's *n_out, ph_join_probe_inner*'s *n_out) return as soon as their kernels and an
 * 8-byte copy are queued, with *n_out = -1; ph_ctx_wait_counts fills every pending count in (waiting
 * for the last of those copies, not for the stream) and reports PH_ECAPACITY for a pair list that
 * overflowed. The count variables must stay alive until then. Independent work queued in between —
 * another join's build, a probe that needs the table but not the count — keeps the GPU busy while
 * the host wakes up and prepares the launches that depend on the count (~25 us per round trip).
 * Switching it off waits first. */
int ph_ctx_set_async_counts(ph_ctx *ctx, int32_t on);
int ph_ctx_wait_counts(ph_ctx *ctx);
/* synchronise and report a pending deferred error (PH_OK when none is pending) */
int ph_ctx_check_deferred(ph_ctx *ctx);
void ph_ctx_destroy(ph_ctx *ctx);

/* ------------------------------------------------------------------ columns */
typedef enum {
    PH_I32 = 1,
    PH_I64 = 2,
    PH_DATE = 3,  /* int32 days since epoch */
    PH_DEC64 = 4, /* int64 unscaled, `scale` fractional digits */
    PH_CODE8 = 5, /* uint8 dictionary code */
    PH_F32 = 6,
    PH_F64 = 7,
    PH_STR = 8,   /* int32 offsets[n+1] in `data`, bytes in `aux` */
    PH_COLREF = 9 /* ph_const only (resident plans): the right operand is another column, ph_const.i = its index */
} ph_type;

typedef struct {
    int32_t type;
    int32_t scale;
    const void *data;       /* host or device pointer, depending on the call */
    const uint8_t *validity;/* bitmap or NULL */
    const void *aux;        /* PH_STR: bytes */
    int64_t aux_bytes;
} ph_col;

/* ------------------------------------------------------------------ table residency
 * Replaces DataTable.Scan's per-chunk materialisation (pkg/storage/table.go:418-428 feeding
 * scanRows, executor_scan.go:158-241): the pruned columns of a table are loaded once into HBM
 * (pinned staging + hipMemcpyAsync) and stay resident across queries. */
typedef struct ph_table ph_table;

int ph_table_create(ph_ctx *ctx, int32_t ncols, const ph_col *host_cols, int64_t nrows,
                    ph_table **out);
int64_t ph_table_rows(const ph_table *t);
int32_t ph_table_ncols(const ph_table *t);
/* device view of column c (data/validity/aux are device pointers) */
int ph_table_col(const ph_table *t, int32_t c, ph_col *out);
/* per-column min/max gathered at load (int64 domain; used for overflow proofs) */
int ph_table_col_range(const ph_table *t, int32_t c, int64_t *min, int64_t *max);
/* ---- Arrow C data interface (the stable ABI of https://arrow.apache.org/docs/format/CDataInterface.html; the two
 * structs are declared here exactly as the specification prints them, under its own include guard) */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char *format;
    const char *name;
    const char *metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema **children;
    struct ArrowSchema *dictionary;
    void (*release)(struct ArrowSchema *);
    void *private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void **buffers;
    struct ArrowArray **children;
    struct ArrowArray *dictionary;
    void (*release)(struct ArrowArray *);
    void *private_data;
};
#endif
/* The column load path through the C-ABI: a record batch — a struct array ("+s") whose children
 * are the columns, as parquet readers, Arrow Flight and Go's arrow/cdata hand it over — becomes a resident table in the
 * device encodings, whole columns at a time. Replaces COPY FROM parquet + the scan's materialisation
 * (pkg/compute/executor_scan.go:272-309 readers, :410-466 parquetColToValue + Vector.SetValue: one VALUE at a time into
 * 24-byte Decimal / 12-byte Date / malloc'd String cells). Arrow's buffers already hold the narrow encodings:
 *   int32 "i" -> PH_I32, int64 "l" -> PH_I64, date32 "tdD" -> PH_DATE (days since 1970-01-01, the same number),
 *   decimal128 "d:p,s" -> PH_DEC64 (the low word; PH_EOVERFLOW when a value needs more than 64 bits),
 *   utf8 "u" / large_utf8 "U" -> PH_CODE8 + dictionary in byte order when the column has <= 256 distinct strings
 *   (also dictionary-encoded arrays of such strings), else PH_STR offsets + bytes;
 * validity bitmaps have pkg/util/bitmap.go's convention (1 bit per row, LSB first, 1 = valid) and pass through (re-packed
 * when the array's offset is not a multiple of 8). The buffers are read during the call only; the caller keeps
 * ownership and releases the batch as usual. `cols` (optional, ncols entries) selects and orders the children, else all. */
int ph_table_create_arrow(ph_ctx *ctx, const struct ArrowSchema *schema, const struct ArrowArray *batch, const int32_t *cols,
                          int32_t ncols, ph_table **out);
/* dictionary of a PH_CODE8 column (code -> string), for callers that did not build it themselves */
int32_t ph_table_dict_size(const ph_table *t, int32_t c);
const char *ph_table_dict_entry(const ph_table *t, int32_t c, int32_t code);

/* Order statistics gathered at load like min / max (one pass per integer column without NULLs), and what the
 * catalog declares. They are what a planner's choice of table form rests on (ph_plan below; ph_join_build_ex's
 * hints): PH_STAT_ASCENDING = values non-decreasing in storage order (a clustering column: lineitem by
 * l_orderkey), PH_STAT_STRICT = strictly ascending (a primary key stored in key order),
 * PH_STAT_DECLARED_UNIQUE = ph_table_declare_unique named this single column. */
#define PH_STAT_ASCENDING 1
#define PH_STAT_STRICT 2
#define PH_STAT_DECLARED_UNIQUE 4
int ph_table_col_stats(const ph_table *t, int32_t c, int32_t *flags);
/* > 0: the column is ascending in runs of this ONE length over consecutive values — row i holds min + i / run_len (PARTSUPP by ps_partkey: four
 * suppliers per part) — so the rows of a key are found by arithmetic (ph_join_run_lookup). Measured at load like the order; 0 = not that shape. */
int32_t ph_table_col_run_len(const ph_table *t, int32_t c);
/* the catalog's PRIMARY KEY / UNIQUE constraint over 1..4 columns of the table (cases/tpch/query/ddl.sql: every
 * TPC-H table declares one). A join whose build key covers a declared-unique set is N:1. Trusted, like the
 * reference trusts its catalog; a lookup that meets two build rows for one key reports it (PH_ECONSTRAINT). */
int ph_table_declare_unique(ph_table *t, int32_t ncols, const int32_t *cols);
/* A co-located copy of 2..8 fixed-width columns without NULLs: row r of the copy holds the columns' values of row r side
 * by side (widest first, the row padded to a power of two up to 64 bytes), beside the column arrays, which stay. It serves
 * ph_gather_multi (and with it the late materialisation of ph_plan): the reference's row-format TupleDataCollection
 * (join_collection.go:268-529) for exactly the columns that are fetched together by row id. A host that knows such a set
 * (the planner: the probe-side columns that survive a selective join) calls this at load; without it the library builds
 * the copy itself the SECOND time the same set of three or more columns of a table is gathered at no more than an eighth of
 * its rows (PH_COLOCATE=0 switches both off). Costs stride x rows bytes of HBM and one pass. */
int ph_table_colocate(ph_table *t, int32_t ncols, const int32_t *cols);
/* 1 when a co-located copy covers the given columns */
int32_t ph_table_colocated(const ph_table *t, int32_t ncols, const int32_t *cols);
/* What the library may spend ON ITS OWN on co-located copies of this table (the automatic build above): `bytes` of HBM in all, default
 * 4 GiB; 0 = never (the host's veto; copies asked for by name with ph_table_colocate are the host's decision and not limited). A build
 * that would exceed the budget is skipped and the gather reads the column arrays. ph_table_colocate_bytes: what the copies hold now. */
int ph_table_set_colocate_budget(ph_table *t, int64_t bytes);
int64_t ph_table_colocate_bytes(const ph_table *t);
/* Sharing (SURVEY.md §8(b) "threading"; the psql server path, cmd/main/main.go:71-122: every connection plans and runs its own query
 * over the same storage). A resident table may be read by plans and operator calls on ANY ctx of its device, from several threads at
 * once: columns, statistics and declared keys are immutable once loading is done (ph_table_create* / ph_table_declare_unique /
 * ph_table_colocate belong to the load phase, one thread); the only state a query may add — an automatic co-located copy — is guarded
 * by a mutex inside the table, built on the calling ctx's stream and ordered against every other consumer's stream by an event. The
 * ctx rule stands: ONE thread at a time per ctx; concurrent queries take a ctx each. ph_table_free: when no plan reads the table. */
void ph_table_free(ph_table *t);

/* plain device buffers for callers without their own allocator */
int ph_dev_alloc(ph_ctx *ctx, int64_t bytes, void **dev);
int ph_dev_free(ph_ctx *ctx, void *dev);
/* n buffers in one call (the end of a query plan frees dozens of intermediates: one FFI crossing) */
int ph_dev_free_many(ph_ctx *ctx, void *const *devs, int64_t n);
int ph_dev_upload(ph_ctx *ctx, void *dev, const void *host, int64_t bytes);
int ph_dev_download(ph_ctx *ctx, void *host, const void *dev, int64_t bytes);
int ph_dev_memset(ph_ctx *ctx, void *dev, int value, int64_t bytes);

/* ------------------------------------------------------------------ filter
 * ExprExec.executeSelect / selectOperation / selectFlatLoop
 * (pkg/compute/expr_exec.go:342-486, function_operator_boolean.go:393-521, 780-868):
 * one comparison `col OP const`, optionally narrowing an input selection (the AND chain of
 * execSelectAnd). Output: ascending row ids, wavefront-ballot compacted.
 * The (type, op) pairs the reference does not implement select nothing, as there. */
typedef enum { PH_EQ = 1, PH_NE, PH_LT, PH_LE, PH_GT, PH_GE, PH_LIKE, PH_NOTLIKE } ph_cmp;

/* A dictionary-code column (PH_CODE8) takes the CODE of a string literal as a PH_I32 constant ('=' / '!='), or — type
 * PH_CODE8, op PH_EQ — a RUN of codes i .. scale: what `LIKE 'prefix%'` or a sorted IN list is over a dictionary in
 * byte order. */
typedef struct {
    int32_t type;  /* PH_I32, PH_DATE, PH_F32 (decimal column vs float literal), PH_DEC64, PH_STR */
    int32_t scale;
    int64_t i;
    double f;
    const char *s; /* host string: '=' operand or LIKE pattern */
} ph_const;

/* col: device column of n rows. sel_in (dev, may be NULL = identity over n rows), n_in rows.
 * sel_out: dev buffer of >= n_in int32. *n_out (host) receives the count (synchronises). */
int ph_filter_select(ph_ctx *ctx, const ph_col *col, int64_t n, int32_t op, const ph_const *k,
                     const int32_t *sel_in, int64_t n_in, int32_t *sel_out, int64_t *n_out);

/* Two conjuncts `col OP1 k1 AND col OP2 k2` over ONE column in one pass (a date range: execSelectAnd, expr_exec.go:430-486, runs the second over
 * the first's selection). Both must lower to value ranges of one kind (integer / date / decimal orderings, '=', a dictionary code);
 * PH_EUNSUPPORTED otherwise — run them one after the other with ph_filter_select. Same outputs as ph_filter_select. */
int ph_filter_select_and(ph_ctx *ctx, const ph_col *col, int64_t n, int32_t op1, const ph_const *k1, int32_t op2, const ph_const *k2,
                         const int32_t *sel_in, int64_t n_in, int32_t *sel_out, int64_t *n_out);

/* `col IN (v1 .. vk)` in one pass (InExpr = an OR of equalities: execSelectOr, expr_exec.go:488-530, runs k passes and a union). INTEGER columns and
 * (k <= 16) and dictionary-code columns (values = codes, any number: what a LIKE over a dictionary selects); PH_EUNSUPPORTED otherwise (the
 * caller unites ph_filter_select results with ph_sel_union). */
int ph_filter_select_in(ph_ctx *ctx, const ph_col *col, int64_t n, const int64_t *values, int32_t nvalues, const int32_t *sel_in, int64_t n_in,
                        int32_t *sel_out, int64_t *n_out);

/* column OP column over the same rows (selectBinary with two FLAT vectors, function_operator_boolean.go:506-521),
 * e.g. Q4 / Q12's l_commitdate < l_receiptdate. The (type, op) pairs are selectOperation's: INTEGER all six, DATE
 * the four orderings, DECIMAL (one scale) '>' only; the pairs the reference does not implement select nothing. */
int ph_filter_select_cols(ph_ctx *ctx, const ph_col *a, const ph_col *b, int64_t n, int32_t op, const int32_t *sel_in,
                          int64_t n_in, int32_t *sel_out, int64_t *n_out);

/* out_dev[i] = i for i < n; marks_dev[sel_dev[i]] = 1 for i < n (bytes; the caller clears them first): the two small device helpers a shim needs to
 * turn a filtered pair list into marks of its probe rows (a join with a residual condition, as ph_plan does it) */
int ph_dev_iota(ph_ctx *ctx, int32_t *out_dev, int64_t n);
int ph_sel_mark(ph_ctx *ctx, const int32_t *sel_dev, int64_t n, uint8_t *marks_dev);

/* OR of predicates = union of their selections: execSelectOr (expr_exec.go:488-530), which is
 * also how `a IN (x, y, ...)` runs (in(a,x) OR in(a,y) ..., `in` selecting like `=`,
 * function_operator_boolean.go:419-429). sels_dev[i] (device, counts[i] ascending row ids < n_rows)
 * are the children's selections, each produced by ph_filter_select over the same input; the
 * output is their union in ascending order (the reference emits it child by child; the row set
 * is the same). out_sel_dev: >= min(n_rows, sum counts) int32. */
int ph_sel_union(ph_ctx *ctx, const int32_t *const *sels_dev, const int64_t *counts, int32_t k,
                 int64_t n_rows, int32_t *out_sel_dev, int64_t *n_out);

/* falseSel of a predicate: the rows of the parent selection (NULL = identity over n_rows) that
 * the child selection does not contain, ascending — what execSelectExpr hands back next to the
 * true rows and executeCase continues with (expr_exec.go:144-246: curSel = curFalseSel). */
int ph_sel_difference(ph_ctx *ctx, const int32_t *parent_dev, int64_t n_parent, const int32_t *child_dev,
                      int64_t n_child, int64_t n_rows, int32_t *out_sel_dev, int64_t *n_out);
/* FillSwitch / TemplatedFillLoop (expr_exec.go:559-606): out[sel[i]] = values[i] for a positional
 * value column (PH_I32 or PH_DEC64 — the two result types FillSwitch handles; n rows), i.e. a CASE branch's THEN/ELSE results written
 * back at the rows the branch selected. out_validity_dev (optional bitmap, zeroed by the caller
 * before the first branch) gets the bit of every row whose value is non-NULL. CASE on the device
 * = ph_filter_select (WHEN) -> ph_expr_eval over the true rows -> ph_scatter, then the same for
 * the next WHEN / the ELSE over ph_sel_difference's rows. */
int ph_scatter(ph_ctx *ctx, const ph_col *values, const int32_t *sel_dev, int64_t n, void *out_data_dev,
               uint8_t *out_validity_dev);

/* ------------------------------------------------------------------ hash
 * Chunk.Hash / HashTypeSwitch / CombineHashTypeSwitch (pkg/chunk/chunk.go:160-166,
 * hash.go:26-41, 182-413; util.HashBytes pkg/util/hash.go:13-65) — bit-identical values.
 * PH_CODE8 columns need `dict_hashes` (dev uint64 per code = HashBytes of the string). */
int ph_hash(ph_ctx *ctx, const ph_col *cols, const uint64_t *const *dict_hashes, int32_t ncols,
            int64_t n, uint64_t *out_dev);
/* host helper: HashBytes of one string (for building dict_hashes) */
uint64_t ph_hash_bytes(const void *p, uint64_t len);

/* ------------------------------------------------------------------ decimal expressions
 * ExprExec.executeExprs over DECIMAL/INTEGER operands (expr_exec.go:85-340;
 * function_operator_binary.go:134-207). Values are exact unscaled int64 at a scale fixed by the
 * binder's typing rules (Mul: sum of scales, Add/Sub: max; function_scalar.go:37-84, 429-475).
 * Program = RPN over columns and literals. */
typedef enum { PH_X_COL = 1, PH_X_CONST, PH_X_ADD, PH_X_SUB, PH_X_MUL,
               PH_X_DIV, PH_X_LT, PH_X_LE, PH_X_GT, PH_X_GE /* ph_float_eval only */ } ph_xop;

typedef struct {
    int32_t op;
    int32_t col;   /* PH_X_COL: index into cols (PH_DEC64 or PH_I32/PH_I64 = scale 0) */
    int64_t ival;  /* PH_X_CONST: unscaled value */
    int32_t scale; /* PH_X_CONST: its scale */
} ph_rpn;

/* FLOAT / DOUBLE arithmetic and comparisons (the FLOAT and DOUBLE overloads of + - * / and of the comparison operators: function_scalar.go:476-512,
 * 960-1025, 1335-1470): the program per row in float32 — every operation rounded to it — or, wide != 0, in float64. PH_X_COL casts the column as the
 * binder does (INTEGER -> float, DECIMAL -> float64 -> float32; a HUGEINT travels as a scale-0 decimal), PH_X_CONST is a FLOAT literal (its float32
 * bits in ival; widened for DOUBLE arithmetic). Comparisons give 1 / 0 and follow selectOperation: FLOAT has > >= <=, DOUBLE has < — the others are
 * never true. out_type PH_I32: the truth value of a program that ends in a comparison (a NULL operand: 0); PH_F32: the value (float32 only). */
int ph_float_eval(ph_ctx *ctx, const ph_col *cols, int32_t ncols, const ph_rpn *prog, int32_t nprog, int32_t wide, const int32_t *sel, int64_t n,
                  int32_t out_type, void *out_dev, uint8_t *out_validity_dev);
/* result scale of a program (host side, no device work); PH_EUNSUPPORTED if malformed */
int ph_expr_scale(const ph_col *cols, const ph_rpn *prog, int32_t nprog, int32_t *scale);
/* out_dev[i] = value of row sel[i] (or i), unscaled at ph_expr_scale's scale. Every add/sub/mul
 * is overflow-checked on the device; if any row leaves the int64 domain the call returns
 * PH_EOVERFLOW (the reference's govalues arithmetic would start rounding there, so the caller
 * must fall back; with ph_ctx_set_deferred_errors the error is reported by the next call that
 * reads back instead). NULL inputs give a NULL result: out_validity_dev (bitmap, may be NULL when no
 * input column has validity) receives the result validity. */
int ph_expr_eval(ph_ctx *ctx, const ph_col *cols, int32_t ncols, const ph_rpn *prog,
                 int32_t nprog, const int32_t *sel, int64_t n, int64_t *out_dev,
                 uint8_t *out_validity_dev);

/* Build check without a device of the generated expression kernels (ph_expr_eval compiles the RPN
 * of batches >= 2^18 rows into straight-line code with hiprtc, cached per expression shape;
 * PH_EXPR_JIT=0 keeps the interpreter): canned shape `which` (0..2) compiles for gfx950. */
int ph_expr_jit_selfcheck(int32_t which);

/* extract(year|month|day from date) — ExtractFunc (pkg/compute/function_scalar.go:1509-1563) over a
 * PH_DATE column: out_dev[i] (int32) for row sel[i] (or i). */
typedef enum { PH_PART_YEAR = 1, PH_PART_MONTH = 2, PH_PART_DAY = 3 } ph_datepart;
int ph_date_extract(ph_ctx *ctx, int32_t part, const ph_col *col, const int32_t *sel, int64_t n,
                    int32_t *out_dev);

/* substring(s FROM offset FOR length) — substringFunc / substringStartEnd
 * (pkg/compute/function_operator_binary.go:553-625; SubstringFunc function_scalar.go:1530-1563): byte
 * positions, 1-based, a negative offset counts from the end, a negative length reads leftwards,
 * offset 0 shortens the length by one; NULL rows give the empty string (and stay NULL through the
 * column's validity). `length` = INT64_MAX is the two-argument form. Result: a PH_STR column over
 * rows sel[0..n) (or 0..n): out_offsets_dev (n+1 int32) and out_bytes_dev; *out_bytes = bytes
 * written (PH_ECAPACITY when out_bytes_capacity is too small; the source's aux_bytes always is
 * enough for lengths >= 0 without repeated rows). */
int ph_substring(ph_ctx *ctx, const ph_col *col, int64_t offset, int64_t length, const int32_t *sel, int64_t n,
                 int32_t *out_offsets_dev, uint8_t *out_bytes_dev, int64_t out_bytes_capacity, int64_t *out_bytes);

/* ------------------------------------------------------------------ string keys
 * VARCHAR group-by / join keys that are not <= 256-value dictionaries (PH_STR columns). The reference hashes them with
 * util.HashBytes (pkg/chunk/hash.go:182-207, pkg/util/hash.go:13-65) and compares candidates byte by byte (Match,
 * pkg/compute/util_match.go:25-301) inside its group table and its join table. On the device both steps are ONE
 * primitive, string interning: every row's string is looked up in an open-addressing table keyed by the same hash and
 * verified by the same byte compare; the first row to claim a slot becomes the string's representative, and every row
 * gets the representative's ROW ID as its int32 code. Equal strings have equal codes, different strings different
 * codes (colliding hashes included: the compare decides), so ph_agg_* and ph_join_* run unchanged on the codes
 * (PH_I32 columns carrying the string column's validity), and a code leads back to the bytes (ph_table_strings).
 * ph_strdict_build: codes of rows sel[0..n) / 0..n of `col` (which must outlive the dictionary); NULL rows get -1.
 * ph_strdict_lookup: codes of another column's rows in that dictionary; absent strings and NULL rows get -2. */
typedef struct ph_strdict ph_strdict;
int ph_strdict_build(ph_ctx *ctx, const ph_col *col, const int32_t *sel, int64_t n, int32_t *codes_out_dev, ph_strdict **out);
int ph_strdict_lookup(ph_strdict *d, const ph_col *col, const int32_t *sel, int64_t n, int32_t *codes_out_dev);
void ph_strdict_free(ph_strdict *d);
/* strings of rows rows_host[0..n) of PH_STR column c of a resident table (group keys that came back as codes):
 * out_offsets n+1 int32 into out_bytes */
int ph_table_strings(ph_ctx *ctx, const ph_table *t, int32_t c, const int64_t *rows_host, int64_t n, int32_t *out_offsets, char *out_bytes,
                     int64_t out_capacity);

/* ------------------------------------------------------------------ hash aggregate
 * GroupedAggrHashTable.AddChunk/FindOrCreateGroups + UpdateStates + FinalizeStates
 * (pkg/compute/aggregate_hash.go:136-391, aggregate_exec.go:456-475,
 * function_aggr.go:420-1365). Device form: open-addressing table in HBM keyed by the packed
 * group key, LDS-staged per workgroup, 128-bit integer sums, counts, min/max; first-seen row id
 * per group so that groups come back in the reference's insertion order. */
typedef enum { PH_A_SUM = 1, PH_A_AVG, PH_A_COUNT, PH_A_MIN, PH_A_MAX, PH_A_COUNT_STAR,
               PH_A_COUNT_DISTINCT /* resident plans only (ph_plan_agg.kind): count(distinct x) — the distinct (group keys, x) rows of a side
                                      table feed the aggregate, as SinkDistinctGrouping / DistinctGrouping do (aggregate_exec.go:76-105,
                                      201-304); at the operator level the same is ph_agg_sink_masked + ph_agg_keys_dev */
} ph_aggkind;

typedef struct {
    int32_t kind;
    int32_t arg; /* index into the args given to sink; ignored for COUNT_STAR */
} ph_aggspec;

typedef struct ph_agg ph_agg;

/* key_types: PH_I32/PH_I64/PH_DATE/PH_DEC64/PH_CODE8 per key column (<= 4 keys).
 * expected_groups sizes the table initially; it is doubled when `capacity - groups <= incoming
 * rows`, the reference's own Resize rule (aggregate_hash.go:214-217). */
int ph_agg_create(ph_ctx *ctx, int32_t nkeys, const int32_t *key_types, int32_t naggs,
                  const ph_aggspec *aggs, int64_t expected_groups, ph_agg **out);
/* keys/args: device columns addressed by row id; rows = sel[0..n) or 0..n. args are PH_I32 /
 * PH_I64 / PH_DEC64 columns (validity honoured: NULL inputs are skipped, NULL keys group together).
 * `positional` != 0: args are addressed by position i instead of row id sel[i] (expression
 * results from ph_expr_eval). The first-seen row recorded for a group is row_base + the row id
 * (sel[i], or i without a selection); give row_base the rows consumed by earlier sinks. */
int ph_agg_sink(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs,
                const int32_t *sel, int64_t n, int32_t positional, int64_t row_base);
/* Same, but only the aggregates whose bit is set in agg_mask are updated (groups are still found
 * or created for every row) — AddChunk's `filter []int` (aggregate_hash.go:155-199), which the
 * reference uses to feed non-DISTINCT aggregates from the raw rows and each DISTINCT aggregate
 * from its own (group keys + argument) table (aggregate_exec.go:74-99, 201-304). */
int ph_agg_sink_masked(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs,
                       const int32_t *sel, int64_t n, int32_t positional, int64_t row_base,
                       uint32_t agg_mask);
/* Key column `key_index` of all current groups, in group-id order, as a dense DEVICE column of
 * the key's own element width (4 B for PH_I32/PH_DATE, 1 B for PH_CODE8, else 8 B) plus an
 * optional validity bitmap (1 bit per group, LSB first, at least (ngroups+7)/8 bytes; NULL keys
 * clear their bit). This is RadixPartitionedHashTable.GetData for the DISTINCT tables
 * (aggregate_exec.go:266-279): the distinct (group keys, argument) rows, re-sunk into the main
 * table with ph_agg_sink_masked. Returns the group count through *ngroups. */
int ph_agg_keys_dev(ph_agg *a, int32_t key_index, void *out_data_dev, uint8_t *out_validity_dev,
                    int64_t capacity, int64_t *ngroups);
/* Aggregate `agg_index` of all current groups, in group-id order, as a dense DEVICE column of int64 (SUM: the sum, PH_EOVERFLOW
 * when one exceeds int64; COUNT / COUNT_STAR: the count; MIN / MAX: the value) plus a validity bitmap (a group no input
 * reached is NULL, as SumOp / CountOp / MinMaxOp.Finalize make it, function_aggr.go:813-823, 950-962). Together with
 * ph_agg_keys_dev this turns an aggregate into a device-resident relation: an aggregate BELOW other operators (a
 * subquery's GROUP BY .. HAVING under a join) never travels through the host. AVG: PH_EUNSUPPORTED. */
int ph_agg_values_dev(ph_agg *a, int32_t agg_index, int64_t *out_dev, uint8_t *out_validity_dev, int64_t capacity, int64_t *ngroups);
/* Streaming aggregate (the planner's StreamAggregate): the FIRST sink into an empty table whose n rows
 * (positions 0..n of keys and args: no selection) arrive ordered by the group-key tuple — e.g. the
 * output of a join whose probe side is clustered by the key. Every group is then one run of adjacent
 * rows: run heads are marked, counted and scanned, and each head's thread reduces its run and writes
 * the group's record, in first-seen order by construction; no hash table, so no later ph_agg_sink into
 * the same table (PH_EUNSUPPORTED). Results are those of ph_agg_sink (exact 128-bit sums, NULL inputs
 * skipped). The order is verified on the device: a key tuple lexicographically below its predecessor's
 * is a DEFERRED PH_ECONSTRAINT of the ctx, reported by the next call that reads back, and the caller
 * aggregates again with ph_agg_sink. Runs may be of any length: a 1024-row tile is reduced as a segmented scan and the pieces of a run that
 * crosses tiles are joined one step per TILE (a table clustered by a low-cardinality key included). NULL-able keys: PH_EUNSUPPORTED. */
int ph_agg_sink_sorted(ph_agg *a, const ph_col *keys, const ph_col *args, int32_t nargs, int64_t n, int64_t row_base);
int ph_agg_group_count(ph_agg *a, int64_t *ngroups);
/* Build check without a device of the plan-specialised sink: ph_agg_sink calls of >= 2^20 rows run
 * the sink kernel compiled (hiprtc, cached per shape) with the key types / aggregate kinds /
 * argument types / NULL-ability of THIS call as constants instead of interpreting them per row
 * (PH_AGG_JIT=0 switches it off). Compiles canned shape `which` (0, 1) for gfx950. */
int ph_agg_jit_selfcheck(int32_t which);
/* Host outputs, groups in first-seen order:
 *   first_row[g]; keys[g*nkeys+c] (int64-widened), key_null[g*nkeys+c];
 *   sum_lo/sum_hi[g*naggs+a] = 128-bit sum (SUM/AVG) or min/max value in sum_lo;
 *   count[g*naggs+a] = non-NULL inputs seen (COUNT_STAR: rows). */
int ph_agg_finalize(ph_agg *a, int64_t max_groups, int64_t *first_row, int64_t *keys,
                    uint8_t *key_null, uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count);
/* ph_agg_finalize that also reports the group count, so a caller with room for max_groups groups
 * needs no ph_agg_group_count first: ONE host round trip for the whole result when header + records
 * fit 64 KiB (the records are packed on the device, which reads the count there). *ngroups is set
 * even when it exceeds max_groups (PH_ECAPACITY: call again with room). */
int ph_agg_fetch(ph_agg *a, int64_t max_groups, int64_t *ngroups, int64_t *first_row, int64_t *keys,
                 uint8_t *key_null, uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count);
/* ph_agg_fetch restricted to the groups whose aggregates satisfy `value OP k` for every conjunct (the HAVING of aggExecutor's output phase,
 * executor_aggr.go:143-263, evaluated on the device: the values as columns, ph_filter_select over the group ids, the survivors packed):
 * agg_index[c] names the aggregate (SUM / MIN / MAX / COUNT: a value that fits int64; AVG is PH_EUNSUPPORTED), value_scale[c] the scale its
 * values carry (ph_agg_result.scale), k[c] a PH_I32 / PH_DEC64 / PH_F32 constant — a DECIMAL value against a FLOAT literal compares in
 * float32, as everywhere. NULL aggregates (no input reached them) fail every comparison. */
int ph_agg_fetch_where(ph_agg *a, int32_t nconj, const int32_t *agg_index, const int32_t *op, const ph_const *k, const int32_t *value_scale,
                       int64_t max_groups, int64_t *ngroups, int64_t *first_row, int64_t *keys, uint8_t *key_null, uint64_t *sum_lo,
                       int64_t *sum_hi, uint64_t *count);
/* Top-N pre-selection for an `ORDER BY <aggregate> [DESC] ... LIMIT k` tail (the reference's
 * orderExecutor + limit, executor_order.go:56-138, executor_limit.go:105-238, sort over all group
 * rows on the host). A one-workgroup radix select finds the k-th best value
 * of aggregate `agg_index` on the device and only the groups at least that good come back
 * (>= k of them when ties exist, fewer when there are fewer groups), in first-seen order, in the
 * same layout as ph_agg_finalize. The caller applies the full ORDER BY (tie-breaks) and LIMIT to
 * those few rows. Sums must fit int64 (PH_EOVERFLOW otherwise: use ph_agg_finalize). SUM / MIN / MAX rank by their value, COUNT / COUNT(*)
 * by their count; AVG and COUNT(DISTINCT) are refused (PH_EUNSUPPORTED). */
int ph_agg_topk(ph_agg *a, int32_t agg_index, int32_t descending, int64_t k, int64_t max_groups,
                int64_t *n_out, int64_t *first_row, int64_t *keys, uint8_t *key_null,
                uint64_t *sum_lo, int64_t *sum_hi, uint64_t *count);
void ph_agg_free(ph_agg *a);

/* ------------------------------------------------------------------ hash join
 * JoinHashTable.Build/Finalize/Probe + Scan.NextInnerJoin (pkg/compute/join_table.go:85-336,
 * join_scan.go:30-300, util_match.go:25-301). Device form: bucket-head table + next[] chains
 * (the reference's layout: head insertion with the previous head kept per row), built with
 * atomic exchange; rows with a NULL key are dropped on both sides. */
typedef struct ph_join ph_join;

int ph_join_build(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n,
                  ph_join **out);
/* ph_join_build with the value range of the (single, integer) key column, as column statistics
 * give it (ph_table_col_range; any superset of the build keys' range). When the build side fills
 * the range densely (at most 8 slots per build row: a primary-key column, possibly filtered) the
 * table is a DIRECT table addressed by key - key_lo: no hashing, no key compares, one 4-byte read
 * per probe, and probes in key order read it front to back. The reference has no counterpart (it
 * always hashes, join_table.go:197-288); every probe call returns what it returns for
 * ph_join_build's tables. Sparse ranges, several key columns and 1-byte keys fall back to
 * ph_join_build. A build key outside [key_lo, key_hi] is an error, reported by ph_join_count
 * (-1 + ph_last_error) — the build itself makes no host round trip. */
int ph_join_build_range(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n,
                        int64_t key_lo, int64_t key_hi, ph_join **out);
/* ph_join_build with what the planner knows about the join:
 *   PH_JOIN_KEY_RANGE  key_lo / key_hi hold the key column's value range (as ph_join_build_range)
 *   PH_JOIN_FK_PROBES  the probe side is a foreign key into this side's key: nearly every probe row
 *                      matches, so a Bloom bitmap would reject nothing. Build sides of >= 32 K rows with
 *                      one key column or two 4-byte ones then take the NODE table (16-byte {key, row,
 *                      next} records: a chain step is one read instead of next + one per key column)
 *                      at every size, not only above 4 M rows.
 * Every probe call works on every table form; the flags only pick the faster one. */
#define PH_JOIN_KEY_RANGE 1
#define PH_JOIN_FK_PROBES 2
/*   PH_JOIN_KEYS_SORTED_UNIQUE  column statistics say the build keys are strictly ascending (a primary key in
 *                      storage order). With PH_JOIN_KEY_RANGE and a dense range the direct table is then ONE
 *                      kernel: the sorted fill, which still verifies the claim, but reports a violation as a
 *                      DEFERRED PH_ECONSTRAINT of the ctx (see ph_ctx_set_deferred_errors: the next call that
 *                      reads back fails, whether or not that option is on) instead of launching the general
 *                      passes behind itself. The caller then builds again without the flag. Ignored where the
 *                      sorted fill does not apply (selections, NULL-able keys, small or sparse tables). */
#define PH_JOIN_KEYS_SORTED_UNIQUE 4
/*   PH_JOIN_EXISTS_ONLY  the table will only be asked WHETHER a key has a build row (ph_join_probe_mark / _mark_where: SEMI, ANTI and mark
 *                      joins). With a key range of at most 64 M values and a big build side it is then a flag table — one byte per key value,
 *                      one pass of plain byte stores, duplicates and row ids never materialised (ph_join_kind "bitmap"); every other probe is
 *                      PH_EUNSUPPORTED on it. Ignored where it does not apply. */
#define PH_JOIN_EXISTS_ONLY 8
int ph_join_build_ex(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const int32_t *sel, int64_t n, int32_t flags,
                     int64_t key_lo, int64_t key_hi, ph_join **out);
/* Filter -> HashJoin build in one pass (filterExecutor under joinExecutor's build child): the rows
 * sel[0..n) / 0..n that pass the comparison are built, without a selection vector and WITHOUT a
 * host read of how many pass — possible because a direct table is sized by the key range, not by
 * the row count. Only where ph_join_build_range picks a direct table (the test uses n, an upper
 * bound of the rows built) and the comparison lowers to an integer range over a column without
 * NULLs; PH_EUNSUPPORTED otherwise, and the caller runs ph_filter_select + ph_join_build_range.
 * Build row ids reported by probes are positions in the UNFILTERED input (sel[i] or i). */
int ph_join_build_where(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const ph_col *where_col, int32_t where_op,
                        const ph_const *where_k, const int32_t *sel, int64_t n, int64_t key_lo, int64_t key_hi,
                        ph_join **out);
/* ... with the planner hints of ph_join_build_ex (PH_JOIN_KEY_RANGE is implied). With
 * PH_JOIN_KEYS_SORTED_UNIQUE the Filter rides along in the verified one-pass sorted fill, which also
 * writes the table's occupancy bitmap (ranges up to 128 M slots): the candidate pass of a later inner probe
 * tests the bitmap and reads the slot array only for keys that are present — Q3's orders build after the
 * customer semi-join flags (the planner's Filter over executor_join.go:54-264's build child). */
int ph_join_build_where_ex(ph_ctx *ctx, const ph_col *keys, int32_t nkeys, const ph_col *where_col, int32_t where_op,
                           const ph_const *where_k, const int32_t *sel, int64_t n, int32_t flags, int64_t key_lo,
                           int64_t key_hi, ph_join **out);
/* the table form a build chose: "direct", "radix", "nodes", "chained+bloom" or "chained".
 * ph_join_build WITHOUT a range, over a million build rows or more, reads the key range off the column and takes the
 * direct table when the keys are dense in it; build sides above 4 M rows whose keys are not (and at most 29 M rows)
 * take the "radix" form: both sides partitioned by key hash, open-addressing tables built in LDS and kept as images
 * that the probes of one partition find on chip (joinExecutor's chained table, join_table.go:85-288, for build sides
 * that no cache holds). It answers inner probes itself; lookups and marks build the node table on first use. */
const char *ph_join_kind(const ph_join *j);
/* 1 when ph_join_probe_inner* emit their pairs in probe-row order (every form except "radix", whose pairs come out
 * partition by partition — the same SET of pairs; a caller that relies on the order asks here) */
int ph_join_pairs_ordered(const ph_join *j);
int64_t ph_join_count(const ph_join *j);
/* Inner probe: writes (probe row id, build row id) pairs to dev buffers of `cap` entries.
 * *n_out (host) = number of matches (may exceed cap -> PH_ECAPACITY, nothing lost but the tail). */
int ph_join_probe_inner(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n,
                        int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap,
                        int64_t *n_out);
/* Filter -> HashJoin probe in one pass: the same pairs as ph_filter_select(where) followed by
 * ph_join_probe_inner over its selection (filterExecutor under joinExecutor's probe child,
 * executor_filter.go:27-114 + executor_join.go:54-264), without materialising the selection: the
 * kernel that streams the probe keys tests the comparison first. Only comparisons that lower to
 * an integer range (INTEGER / DATE / DECIMAL-vs-integer / dictionary-code columns, not `!=`) and
 * tables that carry a Bloom bitmap (build side <= 4 M keys) or are direct tables; PH_EUNSUPPORTED
 * otherwise, and the caller runs the two calls. `sel` narrows the probe rows first, as in ph_join_probe_inner. */
int ph_join_probe_inner_where(ph_join *j, const ph_col *keys, const ph_col *where_col, int32_t where_op,
                              const ph_const *where_k, const int32_t *sel, int64_t n,
                              int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap,
                              int64_t *n_out);
/* ph_join_probe_inner[_where] with a RESIDUAL predicate on the build row: only pairs whose build row r
 * has build_flags_dev[r] != 0 (a byte per row of the build-side table, e.g. from
 * ph_join_probe_mark_where). This is the form a join takes whose build child is Filter / SemiJoin(T): T
 * itself is built — whole, so a primary key in storage order builds in one pass and the filter's
 * row count never reaches the host — and the filter becomes a flag per row of T that the probe tests
 * on the row it found (Q3: lineitem JOIN (orders SEMI JOIN customer WHERE o_orderdate < ..)).
 * where_col may be NULL (no probe-side filter). Direct tables built without a selection only;
 * PH_EUNSUPPORTED otherwise. */
int ph_join_probe_inner_residual(ph_join *j, const ph_col *keys, const ph_col *where_col, int32_t where_op,
                                 const ph_const *where_k, const uint8_t *build_flags_dev, const int32_t *sel, int64_t n,
                                 int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out);
/* Filter -> semi-join mark in one pass over rows 0..n: found_dev[i] = the comparison holds for row i
 * AND its key is in the table (0 / 1). Direct tables, integer-range comparisons, 16-byte aligned
 * columns without NULLs; PH_EUNSUPPORTED otherwise (ph_filter_select + ph_join_probe_mark). */
int ph_join_probe_mark_where(ph_join *j, const ph_col *keys, const ph_col *where_col, int32_t where_op,
                             const ph_const *where_k, int64_t n, uint8_t *found_dev);
/* Semi/anti/mark: found_dev[i] = 1 when probe row sel[i] (or i) has a match */
int ph_join_probe_mark(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n,
                       uint8_t *found_dev);
/* Lookup probe for N:1 joins (unique build keys: a primary key — the shape of every join of Q9 and of
 * the foreign-key joins generally): out_build_dev[i] = the build row matching probe row sel[i] (or
 * i), or -1. One kernel and no compaction, so a chain of such joins keeps ONE row-id array for the
 * probe side (late materialisation: columns are gathered once, after the last join, instead of
 * after every join as Scan.gatherResult does per chunk, join_scan.go:250-278). When a build key is
 * not unique the LAST row of its chain that matches is reported; stats_dev (optional, 2 int32 the
 * caller zeroes) receives the number of probe rows without a match and with more than one, so a
 * caller can verify the uniqueness it assumed with one read at the end of a pipeline. */
int ph_join_lookup(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, int32_t *out_build_dev,
                   int32_t *stats_dev);
/* ph_join_lookup for foreign keys the plan trusts (every probe row has exactly one build row): no
 * statistics buffer and no host read. A probe row without a match (out = -1) or with several is a
 * DEFERRED error: the next call on this ctx that reads anything back from the device (a count, a
 * result download, ph_ctx_check_deferred) fails with PH_ECONSTRAINT instead, and the caller reruns
 * the stage with ph_join_lookup / ph_join_probe_inner. ph_gather, ph_gather_multi and ph_date_extract
 * read row 0 for a negative row id, so the positional pipeline behind a strict lookup stays in
 * bounds until the error is seen. */
int ph_join_lookup_strict(ph_join *j, const ph_col *keys, const int32_t *sel, int64_t n, int32_t *out_build_dev);
void ph_join_free(ph_join *j);

/* Merge-style N:1 lookup, no table: build_key is a unique key column in ascending order (a primary key stored in
 * key order), the probe rows sel[0..n) / 0..n arrive ordered by the key too (a clustered table behind
 * order-preserving operators). out[i] = build row whose key equals probe row i's, or -1. A block of probe rows
 * finds the slice of build keys it spans with two searches, streams it through LDS once and searches there:
 * the build column is read once and nothing is written — where the reference builds one hash table whatever
 * its children's order (executor_join.go:54-264). Both orders are checked on the device as the rows stream by
 * (blocks of very sparse probes search the column directly and check the probe order only); a violation is the
 * ctx's deferred PH_ECONSTRAINT and the caller joins again with ph_join_build + ph_join_lookup. strict != 0:
 * a probe row without a match is a deferred PH_ECONSTRAINT as in ph_join_lookup_strict. */
int ph_merge_lookup(ph_ctx *ctx, const ph_col *build_key, int64_t n_build, const ph_col *probe_key, const int32_t *sel,
                    int64_t n, int32_t strict, int32_t *out_build_dev);

/* Inner pairs against a CLUSTERED build key column, no table: build_key is a key column in ascending order WITH duplicates (lineitem by
 * l_orderkey; PH_STAT_ASCENDING), the probe rows sel[0..n) / 0..n come in any order. Every probe row finds the run of its key with a binary
 * search over the column; the pairs are (probe row id, each row of the run), in probe order. For a probe side far smaller than the build
 * side this replaces every pass a table build makes over ALL build rows (JoinHashTable.Build / Finalize, join_table.go:85-288) by
 * ~log2(n_build) reads per probe row. Same outputs and capacity protocol as ph_join_probe_inner. The column's order is the caller's claim
 * (ph_table_col_stats reports what the library measured at load). */
int ph_join_sorted_pairs(ph_ctx *ctx, const ph_col *build_key, int64_t n_build, const ph_col *probe_key, const int32_t *sel, int64_t n,
                         int32_t *out_probe_dev, int32_t *out_build_dev, int64_t cap, int64_t *n_out);

/* N:1 lookup on a two-column unique key into a table stored in runs of one length by the FIRST key column (ph_table_col_run_len > 0; key1_min =
 * that column's minimum, ph_table_col_range), no table: the candidate rows of probe row i are (probe_keys[0][i] - key1_min) * run_len .. + run_len,
 * the one whose build_key2 equals probe_keys[1][i] is the match (out_build_dev[i], -1 = none). Replaces JoinHashTable.Build + Probe
 * (join_table.go:85-288, join_scan.go:67-165) for a composite foreign key into such a table (Q9's lineitem -> partsupp). strict != 0: a probe row
 * without a match, or with several, is a deferred PH_ECONSTRAINT as in ph_join_lookup_strict. The run structure is the caller's claim. */
int ph_join_run_lookup(ph_ctx *ctx, const ph_col *build_key2, int64_t n_build, int64_t key1_min, int32_t run_len, const ph_col *probe_keys,
                       const int32_t *sel, int64_t n, int32_t strict, int32_t *out_build_dev);

/* Children per parent: what Agg(parent key; count(child column)) <- LEFT JOIN(parent, child ON parent key = child key) computes when the parent
 * key is unique (Q13: orders per customer) without the pair list (NextLeftJoin, join_scan.go) and the second fold (AddChunk, aggregate_hash.go):
 * the child rows child_sel[0..n_child) / 0..n_child are counted by key into an array over [key_min, key_min + key_range), parent row i receives
 * out_counts_dev[i] = the count of its key and bit i of out_valid_dev = count > 0 (a group without a non-NULL input finalises as NULL: CountOp,
 * aggr_ops.go). out_valid_dev: (n_parent + 63) / 64 * 8 bytes. Child rows with a NULL key or a key outside the range match nothing. */
int ph_count_by_key(ph_ctx *ctx, const ph_col *child_key, const int32_t *child_sel, int64_t n_child, int64_t key_min, int64_t key_range,
                    const ph_col *parent_key, const int32_t *parent_sel, int64_t n_parent, int64_t *out_counts_dev, uint8_t *out_valid_dev);

/* Cross product (CrossProduct / CrossProductExec, pkg/compute/join_cross.go:34-230) as row-id pairs:
 * for every right row, all left rows in order — the order the reference emits (one output chunk
 * per (left chunk, right row)) — so that both sides materialise with ph_gather like a join's
 * output. n_left * n_right must stay below 2^31. */
int ph_cross_pairs(ph_ctx *ctx, int64_t n_left, int64_t n_right, int32_t *out_left_dev, int32_t *out_right_dev);

/* gather: out[i] = col[idx[i]] for fixed-width columns (join payload materialisation,
 * TupleDataTemplatedGather join_collection.go:501-529) */
int ph_gather(ph_ctx *ctx, const ph_col *col, const int32_t *idx_dev, int64_t n, void *out_dev);

/* Validity of a row-id vector: bit i of bitmap_dev (pkg/util/bitmap.go's layout, (n + 7) / 8 bytes) = ids_dev[i] >= 0. The build side of a
 * LEFT OUTER join reports row id -1 for a probe row without a match (NextLeftJoin, join_scan.go:67-88: the build side's vectors are constant
 * NULL for those rows); values gathered through such ids (ph_gather* read row 0 for a negative id) carry this bitmap as their validity. */
int ph_rowid_validity(ph_ctx *ctx, const int32_t *ids_dev, int64_t n, uint8_t *bitmap_dev);
/* dictionary codes (PH_CODE8) of rows sel[0..n) / 0..n as INTEGERs: out_dev[i] = code (a code column as a 4-byte group-key part) */
int ph_widen_codes(ph_ctx *ctx, const ph_col *col, const int32_t *sel, int64_t n, int32_t *out_dev);

/* the same for up to 8 columns through one row-id array, in one pass: the late materialisation of a
 * join chain's probe side (all column reads of a row are in flight together, the index is read
 * once). out_dev[c]: n elements of column c's width. */
/* When the views are columns of ONE resident table and the table holds a co-located copy of them (below), the gather
 * reads it instead: one 64-byte sector per row id where the column arrays cost one per column (Q9's five lineitem
 * columns at 5 % of the rows: 1.27 GB of traffic for 0.1 GB of values). */
int ph_gather_multi(ph_ctx *ctx, int32_t ncols, const ph_col *cols, const int32_t *idx_dev, int64_t n,
                    void *const *out_dev);

/* ------------------------------------------------------------------ fused pipelines
 * The measured mode: Agg <- Scan(filter) collapsed into one pass over the resident table, which
 * is what `gpuScanAggExecutor` calls when the sub-plan matches. */
typedef struct {
    int32_t col;  /* table column */
    int32_t op;   /* ph_cmp */
    ph_const k;
} ph_pred; /* conjuncts, AND-ed in order */

typedef struct {
    int32_t kind;      /* ph_aggkind */
    int32_t nprog;     /* 0 for COUNT_STAR */
    ph_rpn prog[12];   /* argument expression over table columns */
} ph_aggexpr;

typedef struct {
    int64_t ngroups;
    /* arrays owned by the result, groups in first-seen order */
    int64_t *first_row;
    int64_t *keys;      /* ngroups * nkeys */
    uint64_t *sum_lo;   /* ngroups * naggs */
    int64_t *sum_hi;
    uint64_t *count;
    int32_t *scale;     /* naggs: scale of each aggregate's argument */
    int32_t nkeys, naggs;
    uint8_t *key_null;  /* ngroups * nkeys, 1 = the key is NULL (the NULL group; keys[] holds 0 there); NULL pointer = no NULL-able key */
} ph_agg_result;

/* group_cols: table columns to group by (0 columns = one global group, the reference's
 * constant-key ungrouped aggregate, executor_aggr.go:37-48).
 * rows [row_begin,row_end) of the table are scanned. */
int ph_scan_filter_agg(ph_ctx *ctx, const ph_table *t, int64_t row_begin, int64_t row_end,
                       const ph_pred *preds, int32_t npreds, const int32_t *group_cols,
                       int32_t ngroup_cols, const ph_aggexpr *aggs, int32_t naggs,
                       ph_agg_result **out);
/* the same launch sequence without the final download (bench inner loop); result stays on device
 * until ph_scan_filter_agg_fetch */
typedef struct ph_scan_plan ph_scan_plan;
int ph_scan_plan_create(ph_ctx *ctx, const ph_table *t, const ph_pred *preds, int32_t npreds,
                        const int32_t *group_cols, int32_t ngroup_cols, const ph_aggexpr *aggs,
                        int32_t naggs, ph_scan_plan **out);
int ph_scan_plan_run(ph_scan_plan *p, int64_t row_begin, int64_t row_end);
int ph_scan_plan_fetch(ph_scan_plan *p, ph_agg_result **out);
/* Multi-GPU merge of fused plans (row-range sharded tables, no data-path collective): the raw
 * partial result of the last run is 2*nacc 64-bit words on the device (ph_scan_plan_partials_dev);
 * the ranks all-gather those few hundred bytes and any rank turns the concatenation of all ranks'
 * words (host memory, rank-major) into the merged result: 128-bit sums and counts add, the
 * first-seen row of a group is the one of the lowest rank that saw it. */
int ph_scan_plan_partials_dev(ph_scan_plan *p, void **dev, int32_t *nwords);
int ph_scan_plan_fetch_merged(ph_scan_plan *p, const uint64_t *words, int32_t nranks,
                              ph_agg_result **out);
/* Build check of the plan-specialised path without a device: generates the kernel source of canned
 * shape `which` (0 Q1, 1 Q6, 2 three keys + MIN/MAX + `!=`, 3 predicate-only COUNT) and compiles it
 * with hiprtc for gfx950; src_out (optional, cap bytes) receives the source. */
int ph_scan_jit_selfcheck(int32_t which, char *src_out, int64_t cap);
/* name of the kernel family the plan dispatches to: "lowcard_chain" / "filter_sumprod" (the two
 * precompiled fused kernels), "jit" (a kernel generated from the plan's shape and compiled with
 * hiprtc at plan creation: any range / != conjunction over NULL-free columns, <= 4 dictionary-code
 * group columns whose slots fit LDS, SUM/AVG/COUNT of products of affine column factors, MIN/MAX of
 * a column), "generic" (the operator chain: filter -> expression -> aggregate sink) */
const char *ph_scan_plan_kind(const ph_scan_plan *p);
void ph_scan_plan_free(ph_scan_plan *p);
void ph_agg_result_free(ph_agg_result *r);

/* ------------------------------------------------------------------ resident plans
 * The join analogue of ph_scan_plan: a whole operator SUBTREE over resident tables,
 *     Agg <- [Project] <- [Filter] <- HashJoin* <- Scan(filter)
 * exactly as buildOperatorExec (pkg/compute/executor.go:305-350) receives it from the planner, handed to the
 * library as a flat array of node descriptors and run there without a chunk crossing the boundary:
 * joinExecutor.Execute's build + probe (executor_join.go:54-264), filterExecutor (executor_filter.go:27-114),
 * projectExecutor (executor_project.go:39-78) and aggExecutor's sink + finalize (executor_aggr.go:106-265).
 * The shim's gpuResidentPlanExecutor translates a PhysicalOperator subtree whose leaves are
 * resident tables into this descriptor, else the tree falls through to the per-operator executors.
 *
 * The library is the physical planner below the operator interface. Per join it picks, from the tables'
 * STATISTICS (ph_table_col_range, ph_table_col_stats, ph_table_declare_unique) — never from hints of the caller:
 *   - the table form: direct table over a dense integer key range, the one-pass sorted fill for a key in
 *     storage order, the gated fill when a filter or a semi-join's marks sit under the build child, node table
 *     for composite foreign keys, chained + Bloom otherwise (ph_join_build_ex / ph_join_build_where_ex);
 *   - the probe form: N:1 lookup when the build key is unique (no pair list; the intermediate stays aligned
 *     with the probe side), a merge lookup without any table when both sides are ordered by the key, marks
 *     instead of pairs for a join that only tests existence (SEMI, or INNER with a unique build key none of
 *     whose columns is used above) and feeds another build, the fused Filter -> probe otherwise;
 *   - sideways information passing: a big build side whose key the probe side has already joined against a
 *     small table is first reduced to the rows that can match (a mark probe against that table);
 *   - late materialisation: intermediates are row-id vectors per base table; columns are gathered once, when an
 *     expression, a key or the aggregate needs them;
 *   - the aggregate form: the fused scan kernels for Agg <- Scan, the streaming aggregate when the rows reach
 *     the aggregate ordered by the first group key, the LDS hash aggregate otherwise, a top-k preselection
 *     when the caller announces ORDER BY <aggregate> LIMIT k (ph_plan_set_topk).
 * Claims derived from statistics (sorted, unique, every foreign key has its row) are verified on the device
 * while the data streams by; a broken claim is the deferred PH_ECONSTRAINT, and ph_plan_fetch then runs the
 * plan again in its conservative forms (general builds, counted lookups, hash aggregate) — results never
 * depend on a statistic being right. ph_plan_explain names the forms the last run chose. */
typedef enum { PH_PN_SCAN = 1, PH_PN_FILTER, PH_PN_JOIN, PH_PN_PROJECT, PH_PN_AGG } ph_plan_kind;
typedef enum { PH_JT_INNER = 1, PH_JT_SEMI, PH_JT_ANTI,
               PH_JT_LEFT   /* LEFT OUTER (NextLeftJoin, join_scan.go:67-88): the inner matches, then every probe row without one with the build
                               side's columns NULL. Those columns carry a validity bitmap from then on: aggregates skip the NULLs (count(x) of a
                               customer without orders is 0 and finalises to NULL, function_aggr.go:950-962), operators that cannot take a
                               NULL-able column answer PH_EUNSUPPORTED. Rows come out matches first, then the unmatched probe rows. */
} ph_plan_join_type;   /* LOT_JoinType* of join_scan.go:47-165 */
typedef enum {
    PH_PE_COL = 1,    /* column reference (executeColumnRef: zero copy) */
    PH_PE_DECIMAL,    /* decimal / integer arithmetic, RPN over the child's output columns (executeFunc) */
    PH_PE_YEAR,       /* extract(year from <DATE column>)  (ExtractFunc, function_scalar.go:1509-1563) */
    PH_PE_CASE,       /* CASE WHEN <when> THEN <prog> ELSE <else_prog> END (executeCase, expr_exec.go:144-246):
                         the WHEN is a select, THEN is evaluated on its true rows and ELSE on the others
                         (FillSwitch, :559-606). Both branches are DECIMAL programs of ONE result scale (an integer
                         constant in a branch — `ELSE 0` — is cast to it), or — result_int != 0 — both INTEGER
                         constants. */
    PH_PE_SUBSTR,     /* substring(<VARCHAR table column col> FROM sub_offset FOR sub_length) (substringFunc,
                         function_operator_binary.go:553-625; ph_substring). The value is VARCHAR computed inside the plan:
                         it may be compared with VARCHAR constants by = / <> in a PH_PN_FILTER above (an IN list is their OR),
                         be a group key or pass through joins; as a group key it is reported as PH_STR by ph_plan_key_info,
                         whose table is then a one-column relation the PLAN owns (valid until the plan runs again or is
                         freed): the key values are its rows, ph_table_strings reads them. */
    PH_PE_FLOAT       /* FLOAT (float_wide = 0) or DOUBLE (float_wide != 0) arithmetic over the child's columns, as ph_float_eval: `prog` with
                         PH_X_DIV and the comparison steps allowed. result_int != 0: the program ends in a comparison and the value is its truth
                         (an INTEGER 1 / 0 column: a PH_PN_FILTER above keeps `column = 1` — Q17's `l_quantity < 0.2 * avg`, Q20's
                         `ps_availqty > 0.5 * sum`); else FLOAT values (float32 only). */
} ph_plan_expr_kind;

/* A boolean expression over a node's input columns as a flat tree (node 0 = the root): what ExprExec.executeSelect
 * walks (execSelectExpr / And / Or / Compare, expr_exec.go:342-530). AND narrows the selection child by child, OR
 * evaluates every child on the parent's rows and unites the true rows (`a IN (x, y)` is in(a,x) OR in(a,y)).
 * A comparison's right operand is a constant or — k.type = PH_COLREF — the column k.i. */
typedef enum { PH_B_CMP = 1, PH_B_AND, PH_B_OR } ph_bool_kind;
typedef struct {
    int32_t kind;          /* ph_bool_kind */
    int32_t col, op;       /* PH_B_CMP: column OP k  (ph_cmp; PH_LIKE / PH_NOTLIKE also over dictionary-code columns) */
    ph_const k;
    int32_t first_child;   /* PH_B_AND / PH_B_OR: children are nodes first_child .. first_child + nchildren - 1 */
    int32_t nchildren;
} ph_bool;

typedef struct {
    int32_t kind;     /* ph_plan_expr_kind */
    int32_t col;      /* PH_PE_COL / PH_PE_YEAR: child output column */
    int32_t nprog;
    ph_rpn prog[12];  /* PH_PE_DECIMAL: ph_rpn.col indexes the child's output columns. PH_PE_CASE: the THEN branch */
    /* PH_PE_CASE */
    int32_t nwhen;
    const ph_bool *when;   /* the WHEN condition over the child's output columns */
    int32_t nelse;
    ph_rpn else_prog[12];
    int32_t result_int;    /* != 0: THEN / ELSE are single PH_X_CONST programs of scale 0 and the result is INTEGER */
    /* PH_PE_SUBSTR */
    int64_t sub_offset, sub_length;   /* as ph_substring's; sub_length = INT64_MAX is the two-argument form */
    /* PH_PE_FLOAT */
    int32_t float_wide;               /* != 0: DOUBLE arithmetic */
} ph_plan_expr;

typedef struct {
    int32_t kind;     /* ph_aggkind */
    ph_plan_expr arg; /* over the child's output columns; ignored for PH_A_COUNT_STAR */
} ph_plan_agg;

typedef struct {
    int32_t kind;                /* ph_plan_kind */
    int32_t child[2];            /* node indexes (children precede their parent), -1 = none.
                                    PH_PN_JOIN: child[0] probes, child[1] is built (executor_join.go:237-264) */
    /* PH_PN_SCAN: the pruned columns of a resident table + the conjuncts pushed into the scan
       (scanExecutor.runFilterExec, executor_scan.go:225). ph_pred.col is a TABLE column. */
    const ph_table *table;
    int32_t ncols;
    const int32_t *cols;         /* table columns the scan emits, in output order */
    /* PH_PN_SCAN / PH_PN_FILTER: conjuncts, AND-ed in order (PH_PN_FILTER: ph_pred.col = child output column) */
    int32_t npreds;
    const ph_pred *preds;
    /* ... and one more conjunct of any shape (OR / IN lists, column-vs-column comparisons), AND-ed behind them.
       Simple `column OP constant` conjuncts belong in preds: those are the ones a build or a probe can absorb. */
    int32_t nbools;
    const ph_bool *bools;
    /* PH_PN_JOIN: equi-join on nkeys column pairs; the output picks from [probe child's columns | build
       child's columns] (SEMI / ANTI: probe columns only). A join's `bools` is its RESIDUAL condition — the non-equi
       conjuncts of the ON clause (Q21's `l2.l_suppkey <> l1.l_suppkey` inside EXISTS: the reference's HashJoin keeps them
       as join conditions beside the keys), a tree over [probe columns | build columns]: an INNER join keeps the pairs that
       satisfy it, a SEMI (ANTI) join the probe rows with at least one (without any) such pair. Not with PH_JT_LEFT. */
    int32_t join_type;           /* ph_plan_join_type */
    int32_t nkeys;
    const int32_t *probe_keys;   /* probe child's output columns */
    const int32_t *build_keys;   /* build child's output columns */
    int32_t nout;
    const int32_t *out;          /* indexes into the concatenation [probe columns | build columns] */
    /* PH_PN_PROJECT */
    int32_t nexprs;
    const ph_plan_expr *exprs;
    /* PH_PN_AGG (the root): group-by expressions and aggregates over the child's output columns; no group
       expression = one global group (executor_aggr.go:37-48) */
    int32_t ngroups;
    const ph_plan_expr *groups;
    int32_t naggs;
    const ph_plan_agg *aggs;
} ph_plan_node;

typedef struct ph_plan ph_plan;
typedef struct ph_comm ph_comm;   /* the multi-GPU communicator (declared with its entry points further down) */
/* nodes[nnodes-1] is the root: a PH_PN_AGG (the groups come back through ph_plan_fetch) or a join / filter / project (its rows through
 * ph_plan_fetch_rows). The descriptor (and the strings of its predicates) is
 * copied; the tables must outlive the plan. PH_EUNSUPPORTED for a shape outside the device path (the caller
 * keeps its per-operator executors). */
int ph_plan_create(ph_ctx *ctx, const ph_plan_node *nodes, int32_t nnodes, ph_plan **out);
/* ORDER BY <aggregate agg_index> [DESC] ... LIMIT k sits above the aggregate: only the groups at least as good
 * as the k-th come back (>= k with ties), as ph_agg_topk; the caller applies the full ORDER BY and the LIMIT. PH_EUNSUPPORTED beside
 * ph_plan_set_having: HAVING runs in the aggregate's output phase, before Order and Limit (executor_aggr.go:143-263) — the k best groups
 * could fail it while later ones pass — so a plan carries one or the other, and with a HAVING the caller sorts the survivors. */
int ph_plan_set_topk(ph_plan *p, int32_t agg_index, int32_t descending, int64_t k);
/* The same announcement for a JOIN-rooted plan (ph_plan_fetch_rows) under `ORDER BY <column col of the rows> [DESC] ... LIMIT k` (orderExecutor +
 * limit above the join, executor_order.go:56-138, executor_limit.go:105-238): only the rows whose key is at least as good as the k-th best come
 * back (>= k with ties; all rows when the key can be NULL, is a VARCHAR / BIGINT, or a DECIMAL in ascending order or of a scale above 2 — the
 * orderings the reference's selection and sort encoder do not share). The caller applies the full ORDER BY and the LIMIT to those rows. */
int ph_plan_set_rows_topk(ph_plan *p, int32_t col, int32_t descending, int64_t k);
/* HAVING conjuncts `result column OP constant` over the root's AGGREGATE columns (ph_pred.col counts the result's columns: group keys first,
 * then the aggregates), applied on the device when the groups are fetched (ph_agg_fetch_where): only the surviving groups come back.
 * PH_EUNSUPPORTED — and the caller filters the fetched rows itself — for a conjunct over a key column or an AVG, for Agg <- Scan plans
 * (a fused scan: few groups) and beside ph_plan_set_topk. */
int ph_plan_set_having(ph_plan *p, int32_t nconj, const ph_pred *conj);
/* 1 when the last ph_plan_fetch applied the conjuncts of ph_plan_set_having. The device compares the aggregates as int64 values: when
 * a sum of the run exceeds that range every group comes back with its exact 128-bit sums instead (0 here; likewise a top-k
 * preselection is then skipped) and the caller applies its HAVING to the fetched rows, as without ph_plan_set_having. */
int32_t ph_plan_having_applied(const ph_plan *p);
/* enqueue one execution of the whole subtree (host round trips only where a row count sizes the next step) */
int ph_plan_run(ph_plan *p);
/* the group rows of the last run, in first-seen order (ph_agg_result_free releases them). If the run's
 * optimistic forms met a broken statistic (deferred PH_ECONSTRAINT), the plan is run again conservatively
 * first; later runs of this plan then start conservatively. */
int ph_plan_fetch(ph_plan *p, ph_agg_result **out);
/* A plan whose root is NOT an aggregate — a join, filter or project (Q15's final Join(supplier, revenue = max)) — returns the root
 * relation's rows: every fixed-width column as nrows values widened to 64 bits (type[c] / scale[c] say what they are; NULL-able
 * columns are PH_EUNSUPPORTED), every VARCHAR column — a table column behind row ids, or a value computed in the plan — as
 * offsets[c] (nrows + 1) and bytes[c], gathered on the device. Row order is the root relation's (probe order for joins the
 * library ran in probe order; apply the query's ORDER BY above). */
typedef struct {
    int64_t nrows;
    int32_t ncols;
    int32_t *type, *scale;
    int64_t **values;    /* [ncols]: NULL for VARCHAR columns */
    int32_t **offsets;   /* [ncols]: NULL for fixed-width columns */
    char **bytes;
} ph_rows_result;
int ph_plan_fetch_rows(ph_plan *p, ph_rows_result **out);
void ph_rows_result_free(ph_rows_result *r);
/* group key k of the result: device type and scale, and — when it is a table column carried through unchanged
 * (dictionary codes need their dictionary) — the table and column it comes from (else *table = NULL) */
int ph_plan_key_info(const ph_plan *p, int32_t k, int32_t *type, int32_t *scale, const ph_table **table, int32_t *col);
/* type (PH_I32 / PH_I64 / PH_DEC64 ...) of aggregate a's argument; its scale is ph_agg_result.scale[a] */
int ph_plan_agg_arg_type(const ph_plan *p, int32_t a, int32_t *type);
/* one line per operator of the last run: the forms chosen and the row counts seen */
const char *ph_plan_explain(const ph_plan *p);
/* Multi-rank execution (no reference counterpart: the reference runs one goroutine). Every rank creates the SAME plan over its shard
 * of the sharded tables (row ranges: the default) and its copy of the replicated ones (ph_table_set_replicated: NATION, REGION, any small table),
 * announces the communicator, and runs and fetches like a single rank; every rank receives the complete result. The library inserts the exchanges:
 *   join: build side replicated -> local; both sides co-located by key RANGE (the ranks' column statistics: a database split by order ranges)
 *         -> local; build side small (ph_plan_set_broadcast_rows, default 4 Mi rows in all) -> all-gathered (VARCHAR columns included) into a
 *         replicated temporary table; otherwise both sides hash-partitioned by the first key (ph_partition_dev) and exchanged all-to-all;
 *   aggregate below other operators: whole groups per rank — by disjoint key ranges, else its input hash-partitioned by a group key;
 *   root aggregate: local, the ranks' partial states (128-bit sums, counts, min / max) merged at ph_plan_fetch; with a top-k, a HAVING or a
 *         DISTINCT aggregate its input is hash-partitioned first and the ranks' whole groups are concatenated;
 *   join-rooted plans: the root relation's rows all-gathered.
 * Every decision is taken from all-reduced values, so all ranks walk the same sequence of collectives; a broken statistic on any rank is agreed on at
 * the end of the run and all ranks rerun conservatively, together. Works over RCCL (ph_comm_init) and the in-process transport alike. */
int ph_plan_set_comm(ph_plan *p, ph_comm *comm);
int ph_plan_set_broadcast_rows(ph_plan *p, int64_t rows);
/* this rank holds ALL rows of the table (every rank loaded the same rows) — the default is a shard (a row range) */
int ph_table_set_replicated(ph_table *t, int32_t on);
void ph_plan_free(ph_plan *p);

/* ------------------------------------------------------------------ multi-GPU partitioning
 * No reference counterpart (the reference is single-threaded): hash-partition
 * rows by key so that join build/probe sides and group-by keys co-locate per GPU.
 * dest = mix64(key) % nparts. Writes per-partition counts and a permutation of row ids grouped
 * by destination, input order kept inside every partition (reproducible); the caller gathers
 * columns with ph_gather and exchanges them (ph_comm_* below). */
int ph_partition(ph_ctx *ctx, const ph_col *key, const int32_t *sel, int64_t n, int32_t nparts,
                 int64_t *counts_host, int32_t *perm_dev);
/* Same, with the nparts counts left on the device (int64 each) and no host round trip: what the
 * exchange below consumes (ph_comm_exchange_counts reads them there). */
int ph_partition_dev(ph_ctx *ctx, const ph_col *key, const int32_t *sel, int64_t n, int32_t nparts,
                     int64_t *counts_dev, int32_t *perm_dev);

/* ------------------------------------------------------------------ multi-GPU exchange (RCCL over xGMI)
 * One process (or goroutine-pinned OS thread) per GPU; SURVEY.md §8(e). No reference counterpart.
 * The plug point that drives it is the same one that builds the executors (buildOperatorExec,
 * pkg/compute/executor.go:305-350): a partitioned gpuJoinExecutor partitions its build and probe
 * batches with ph_partition_dev, gathers the needed columns (ph_gather) and exchanges them.
 * Everything is stream-ordered on the ctx stream; a stage makes ONE host round trip (the count
 * matrix), none between the partition, the gathers and the all-to-all. */
#define PH_COMM_ID_BYTES 128
typedef struct ph_comm ph_comm;
typedef enum { PH_RED_SUM = 1, PH_RED_MAX = 2, PH_RED_MIN = 3 } ph_redop;
/* rank 0 creates the id (ncclGetUniqueId) and ships its 128 bytes to the other ranks by any host
 * channel (the Go side: the coordinator's RPC; the tests: a file / torch.distributed store) */
int ph_comm_unique_id(void *id_out);
int ph_comm_init(ph_ctx *ctx, int32_t nranks, int32_t rank, const void *id, ph_comm **out);
/* The in-process transport: the ranks are THREADS of one process, each with a ctx of its own — on different devices (a host that drives every
 * GPU of a node from one process) or on one (the single-GPU test box: RCCL refuses two ranks on one device). ph_local_group_create makes the
 * rendezvous object the threads share; every thread calls ph_comm_init_local with its rank. Every ph_comm_* call below then works as over
 * RCCL (same arguments, same results); the copies are device-to-device on the calling ctx's stream between host barriers. */
typedef struct ph_local_group ph_local_group;
int ph_local_group_create(int32_t nranks, ph_local_group **out);
void ph_local_group_free(ph_local_group *g);
int ph_comm_init_local(ph_ctx *ctx, ph_local_group *g, int32_t rank, ph_comm **out);
int32_t ph_comm_nranks(const ph_comm *c);
int32_t ph_comm_rank(const ph_comm *c);
void ph_comm_destroy(ph_comm *c);
/* every rank contributes `bytes` from send_dev; recv_dev gets nranks*bytes, rank-major. async != 0:
 * the collective runs on the communicator's own stream behind everything already queued on the ctx
 * stream, so later kernels on the ctx stream overlap it; ph_comm_wait makes the ctx stream wait for
 * it (stream-side, no host block). Used for the per-step merge of fused-plan partials. */
int ph_comm_allgather(ph_comm *c, const void *send_dev, void *recv_dev, int64_t bytes, int32_t async);
int ph_comm_wait(ph_comm *c);
/* same, except that the newest `keep` (0..3) asynchronous collectives may still be running: with
 * two alternating result buffers, `keep` = 1 before a step lets the previous step's collective
 * overlap this step's kernels while the buffer about to be overwritten is safe */
int ph_comm_wait_keep(ph_comm *c, int32_t keep);
/* n <= 64 host values reduced over all ranks, result on every rank (timings, row totals, flags) */
int ph_comm_allreduce_i64(ph_comm *c, int64_t *host_vals, int32_t n, int32_t op);
int ph_comm_barrier(ph_comm *c);
/* send_counts_dev: this rank's nranks per-destination row counts (ph_partition_dev's output).
 * matrix_host[s*nranks + d] = rows rank s sends to rank d, on every rank: one all-gather and the
 * stage's one device->host copy. */
int ph_comm_exchange_counts(ph_comm *c, const int64_t *send_counts_dev, int64_t *matrix_host);
/* host-only (no device, no RCCL): offsets of every peer's rows in this rank's send buffers (rows
 * ordered by destination, as ph_partition orders them) and receive buffers (ordered by source rank).
 * send_off / recv_off: nranks+1 entries; recv_off[nranks] = rows this rank ends up with. */
int ph_exchange_layout(const int64_t *matrix, int32_t nranks, int32_t rank, int64_t *send_off, int64_t *recv_off);
/* all-to-all of ncols column buffers (elem_bytes[k] bytes per row) as one group of
 * ncclSend/ncclRecv pairs: every peer pair and every column concurrently. recv_dev[k] needs
 * recv_off[nranks]*elem_bytes[k] bytes. */
int ph_comm_exchange_columns(ph_comm *c, int32_t ncols, const void *const *send_dev, void *const *recv_dev,
                             const int32_t *elem_bytes, const int64_t *matrix_host);
/* variable-length all-gather of one column (broadcast of a small build side): counts_host[r] =
 * rows of rank r, recv_dev = all rows in rank order (PH_ECAPACITY when recv_capacity is too small;
 * counts_host is valid then) */
int ph_comm_allgather_rows(ph_comm *c, const void *send_dev, int64_t count, int32_t elem_bytes, void *recv_dev,
                           int64_t recv_capacity, int64_t *counts_host);
/* The capacity test above is collective: the ranks exchange (count, capacity) pairs and either ALL return
 * PH_ECAPACITY (the total exceeds the smallest capacity) or all exchange — no rank leaves alone. This form needs
 * no capacity at all: the library allocates exactly the total (*recv_dev_out, from ph_dev_alloc's pool; free it
 * with ph_dev_free). */
int ph_comm_allgather_rows_alloc(ph_comm *c, const void *send_dev, int64_t count, int32_t elem_bytes, void **recv_dev_out,
                                 int64_t *counts_host);

/* ------------------------------------------------------------------ measurement
 * streaming-read ceiling: a read-only reduce over `bytes` of device memory with the fused scan
 * kernels' access pattern (grid workgroups of 256 threads, 16-byte non-temporal loads); one word
 * per workgroup is written to out_words_dev. The caller times it (HIP events on the ctx stream). */
int ph_dev_read_reduce(ph_ctx *ctx, const void *dev, int64_t bytes, uint64_t *out_words_dev, int32_t grid);

/* ------------------------------------------------------------------ ORDER BY
 * LocalSort over fixed-size keys (sort_local.go:64-250; key layout sort_layout.go:29-88; encoders
 * sort_encoder.go:33-114; RadixScatter sort_radix.go:242-380): rows sel[0..n) (or 0..n) ordered
 * by the ORDER BY columns `keys` (first = most significant), descending[c] != 0 for DESC.
 * As in the reference NULLs always sort first (sort_layout.go:46), DECIMAL keys compare by their
 * value rounded half-even to two decimals (decimalEncoder: dec.Int64(2)), DATE by (year, month,
 * day), INTEGER as int32; PH_CODE8 keys compare by code and therefore need a dictionary in
 * ascending byte order (the loader's dictionaries are). BIGINT / DOUBLE keys: PH_EUNSUPPORTED (the
 * reference's RadixScatter has no case for them either). Rows with equal keys keep their input
 * order (the reference leaves their order undefined). out_rows_dev: n int32 row ids, sorted. */
int ph_sort_rows(ph_ctx *ctx, const ph_col *keys, const int32_t *descending, int32_t nkeys,
                 const int32_t *sel, int64_t n, int32_t *out_rows_dev);

#ifdef __cplusplus
}
#endif
#endif

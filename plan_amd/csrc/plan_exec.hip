// Resident plans (ph_plan_*): an operator subtree Agg <- [Project] <- [Filter] <- HashJoin* <- Scan(filter) over
// resident tables, lowered here to the operator-granular entry points of this library. This is the physical
// planner below the reference's operator interface: what buildOperatorExec (pkg/compute/executor.go:305-350)
// instantiates as a tree of joinExecutor / filterExecutor / projectExecutor / aggExecutor objects pulling
// 2048-row chunks through each other (executor_join.go:54-264, executor_filter.go:27-114,
// executor_project.go:39-78, executor_aggr.go:106-265) runs as ONE descriptor in the library, and every choice the
// hand-assembled Q3 / Q9 pipelines of round 2 made by hand (plan_amd/pipelines.py) is made here from the tables'
// statistics. Host code only: the kernels are the ones behind ph_join_*, ph_filter_select, ph_gather*, ph_expr_eval,
// ph_agg_*, ph_merge_lookup and ph_scan_plan_*.
//
// An intermediate result ("relation") is a set of LANES — one row-id vector per base table taking part, all of
// the relation's length and aligned by position — plus output columns that are either references into a lane's
// table (late materialisation: gathered when something needs the values) or positional device columns.
#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace {

struct BoolTree {                   // a deep copy of a ph_bool array (node 0 = root); string constants owned
    std::vector<ph_bool> nodes;
    std::vector<std::string> strs;
    void fix() { for (size_t i = 0; i < nodes.size(); i++) nodes[i].k.s = strs[i].empty() ? nullptr : strs[i].c_str(); }
    bool empty() const { return nodes.empty(); }
};

BoolTree copy_bools(const ph_bool *b, int32_t n) {
    BoolTree t;
    for (int32_t i = 0; i < n; i++) {
        t.nodes.push_back(b[i]);
        t.strs.push_back(b[i].k.s ? std::string(b[i].k.s) : std::string());
        t.nodes.back().k.s = nullptr;
    }
    return t;
}

struct PCol {                       // an output column of a relation
    int32_t type = 0, scale = 0;
    int lane = -1, tcol = -1;       // lane >= 0: column tcol of lanes[lane].t, addressed through lanes[lane].rows
    const void *data = nullptr;     // lane < 0: positional values, one per row of the relation
    const uint8_t *validity = nullptr;
    const ph_table *src = nullptr;  // provenance: the table column these values are unchanged copies of
    int src_col = -1;
    bool ordered = false;           // values are non-decreasing in the relation's row order
    int domain = -1;                // index into ph_plan::domains: the values are a subset of that join table's keys
    ph_strdict *sdict = nullptr;    // a VARCHAR value COMPUTED in the plan (substring): the values are int32 codes of this dictionary,
                                    // a code = the row of `src` (a one-column relation the plan owns) that holds the string
    bool grange = false;            // multi-rank plans: the values lie in [gmin, gmax] on EVERY rank (the ranks' column statistics reduced before an
    int64_t gmin = 0, gmax = 0;     // exchange; `src` statistics are a rank's own and say nothing about rows that arrived from elsewhere)
};

struct Lane {
    const ph_table *t = nullptr;
    const int32_t *rows = nullptr;  // nullptr = identity (row i of the relation is row i of the table)
    bool asc = true;                // the row ids are ascending (table order preserved)
    bool dup_free = true;           // no table row occurs twice
    bool nullable = false;          // row id -1 = no row (the build side of a LEFT OUTER join): the lane's columns are NULL there
};

struct Rel {
    int64_t n = 0;                  // rows; an upper bound (the table's rows) while `pending` / `flags` are not applied
    std::vector<Lane> lanes;
    std::vector<PCol> cols;
    // single identity lane only: what still filters the table's rows
    std::vector<ph_pred> pending;   // conjuncts over TABLE columns, not applied yet (fused into the next build / probe when possible)
    std::vector<BoolTree> complex;  // conjuncts of any shape over TABLE columns (OR lists, column-vs-column), applied behind them
    const uint8_t *flags = nullptr; // a byte per table row, 0 = filtered out (the marks of a semi-join)
    bool covers = true;             // every table row is (still) there, up to a reduction by the probing side's own key domain
    bool replicated = false;        // multi-rank plans: every rank holds the same rows (else: the ranks' rows together are the relation)
    bool lazy() const { return !pending.empty() || !complex.empty() || flags != nullptr; }
    bool single_identity() const { return lanes.size() == 1 && lanes[0].rows == nullptr; }
};

struct Expr {                       // ph_plan_expr + the WHEN tree it points to, owned
    ph_plan_expr e;
    BoolTree when;
};

struct AggDesc { int32_t kind; Expr arg; };

struct Node {
    int32_t kind = 0, child[2] = {-1, -1};
    const ph_table *table = nullptr;
    std::vector<int32_t> cols;
    std::vector<ph_pred> preds;
    std::vector<std::string> pred_strs;
    BoolTree bools;
    int32_t join_type = 0;
    std::vector<int32_t> pkeys, bkeys, out;
    std::vector<Expr> exprs, groups;
    std::vector<AggDesc> aggs;
};

struct Domain { ph_join *j; int64_t nkeys; };

struct KeyInfo { int32_t type = 0, scale = 0; const ph_table *table = nullptr; int32_t col = -1; };
// where original group key k lives in the aggregate's key words: part 0 = a whole word, 1 = the high half, 2 = the low half (biased by 2^31)
struct KeyPack { int word; int part; };

}  // namespace

struct ph_plan {
    ph_ctx *ctx = nullptr;
    std::vector<Node> nodes;
    int32_t topk_agg = -1, topk_desc = 0;
    int64_t topk_k = 0;
    int32_t rows_topk_col = -1, rows_topk_desc = 0;   // a join-rooted plan under ORDER BY <column> LIMIT k (ph_plan_set_rows_topk)
    int64_t rows_topk_k = 0;
    bool conservative = false;          // statistics are not trusted (after a broken claim)
    bool no_stream_agg = false;         // only the streaming aggregate's order claim broke (rows ordered by the first key, not by the key TUPLE): the
                                        // hash aggregate from then on — the joins keep their optimistic forms
    // state of the last run
    bool ran = false;
    std::vector<void *> temps;
    std::vector<ph_join *> joins;
    std::vector<ph_strdict *> strdicts;
    std::vector<ph_agg *> inner_aggs;   // aggregates below other operators
    bool rows_root = false;             // the root is no aggregate: the plan returns the root relation's rows (ph_plan_fetch_rows)
    std::shared_ptr<Rel> rows_rel;      // ... of the last run (its buffers are run temporaries: fetched before they are released)
    std::vector<ph_pred> having;        // conjuncts over the root's aggregate columns, applied where the groups are (ph_plan_set_having)
    bool having_applied = false;        // ... and whether the last fetch did apply them (a sum beyond int64 hands every group back)
    // A join whose build side is an aggregate grouped by the join key: the probe side's keys, lowered first, filter the aggregate's INPUT (groups no
    // probe row asks for are never built). Set by lower_join around the build child's lowering, consumed by that aggregate node.
    struct PushedKeys { std::shared_ptr<Rel> probe; int32_t pkey = -1, gcol = -1; int join = -1; };
    std::map<int, PushedKeys> pushed;
    std::vector<int> parents;           // how many nodes reference node i as a child
    std::map<std::pair<int, bool>, std::shared_ptr<Rel>> memo;   // relations of nodes with several parents, per run
    std::vector<ph_table *> computed;   // one-column relations of computed VARCHAR values: like the aggregate they outlive the fetch (the host
    std::vector<void *> computed_bufs;  // reads a group key's strings from them) and go with the next run
    std::vector<Domain> domains;
    ph_agg *agg = nullptr;
    ph_scan_plan *scan = nullptr;       // Agg <- Scan: the fused scan plan
    std::vector<KeyInfo> keys;
    std::vector<KeyPack> key_packs;     // how the result's key words unpack into the original group keys
    std::vector<int32_t> agg_scale, agg_arg_type;
    std::string explain;
    int64_t expected_groups = 1024;
    // multi-rank execution (ph_plan_set_comm)
    ph_comm *comm = nullptr;
    int64_t bcast_rows = 4ll << 20;     // build sides up to this many rows IN ALL are replicated (all-gather) instead of hash-partitioned
    bool root_disjoint = false;         // the root aggregate's groups of this rank are nobody else's (no merge of partial states at fetch)
    bool no_sideways = false;           // (transient) a join lowered with its roles swapped for the table-less form: the build table is NOT to be reduced first
    bool topk_off = false;              // this run leaves the top-k preselection out (across ranks its groups could not be made whole): all groups come back
    bool root_replicated = false;       // every rank computed the whole result
};

namespace {

using ph::set_error;

#define PL_CHECK(expr) do { int rc__ = (expr); if (rc__ != PH_OK) return rc__; } while (0)

int width_of(int32_t t) { return t == PH_STR ? 0 : ph::type_width(t); }

void note(ph_plan *p, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    p->explain += buf;
    p->explain += '\n';
}

int palloc(ph_plan *p, int64_t bytes, void **out) {
    PL_CHECK(p->ctx->pool_alloc(bytes > 0 ? bytes : 8, out));
    p->temps.push_back(*out);
    return PH_OK;
}

void release_run(ph_plan *p, bool keep_agg) {
    for (ph_join *j : p->joins) ph_join_free(j);
    p->joins.clear();
    for (ph_strdict *d : p->strdicts) ph_strdict_free(d);
    p->strdicts.clear();
    for (ph_agg *a : p->inner_aggs) ph_agg_free(a);
    p->inner_aggs.clear();
    p->domains.clear();
    p->memo.clear();
    p->rows_rel.reset();
    for (void *q : p->temps) p->ctx->pool_release(q);
    p->temps.clear();
    if (!keep_agg) {
        if (p->agg) { ph_agg_free(p->agg); p->agg = nullptr; }
        for (ph_table *vt : p->computed) delete vt;
        p->computed.clear();
        for (void *q : p->computed_bufs) p->ctx->pool_release(q);
        p->computed_bufs.clear();
    }
}

ph_col table_view(const ph_table *t, int c) {
    ph_col v{};
    const auto &d = t->cols[(size_t)c];
    v.type = d.type; v.scale = d.scale; v.data = d.data; v.validity = d.validity; v.aux = d.aux; v.aux_bytes = d.aux_bytes;
    return v;
}

// device view of a column and the selection that addresses it (nullptr = positions 0..n)
ph_col col_view(const Rel &r, const PCol &c, const int32_t **sel) {
    if (c.lane >= 0) {
        *sel = r.lanes[(size_t)c.lane].rows;
        return table_view(r.lanes[(size_t)c.lane].t, c.tcol);
    }
    ph_col v{};
    v.type = c.type; v.scale = c.scale; v.data = c.data; v.validity = c.validity;
    *sel = nullptr;
    return v;
}

// an INTEGER literal against a DECIMAL column (or a HUGEINT aggregate carried as a scale-0 decimal) is cast to the column's
// type by the binder (DecimalSizeCheck + tryCastInt32ToDecimal, ltype.go:601-624, function_cast.go:337-347)
void fix_num_const(const ph_col &v, ph_const *k) {
    if (v.type == PH_DEC64 && k->type == PH_I32) { k->type = PH_DEC64; k->scale = 0; }
}

// a string literal against a dictionary-code column: the code of the literal (999 = not in the dictionary)
void fix_dict_const(const ph_table *t, int tcol, ph_const *k) {
    if (!t || tcol < 0) return;
    const auto &d = t->cols[(size_t)tcol];
    if (d.type == PH_CODE8 && k->type == PH_STR) {
        int code = 999;
        for (size_t i = 0; i < d.dict.size(); i++) if (k->s && d.dict[i] == k->s) code = (int)i;
        k->type = PH_I32;
        k->i = code;
        k->s = nullptr;
    }
}

int positional(ph_plan *p, Rel *r, const std::vector<int> &want);
int const_code(ph_plan *p, ph_strdict *d, const char *s, int32_t *code);
int const_codes(ph_plan *p, ph_strdict *d, const std::vector<const char *> &strs, std::vector<int32_t> *codes);

// the reference's LIKE (likeOp: % = any run, _ = any one byte), for patterns applied to a DICTIONARY on the host
bool host_like(const char *s, size_t sl, const char *pat, size_t pl) {
    size_t si = 0, pi = 0, star = (size_t)-1, mark = 0;
    while (si < sl) {
        if (pi < pl && (pat[pi] == '_' || pat[pi] == s[si])) { si++; pi++; }
        else if (pi < pl && pat[pi] == '%') { star = pi++; mark = si; }
        else if (star != (size_t)-1) { pi = star + 1; si = ++mark; }
        else return false;
    }
    while (pi < pl && pat[pi] == '%') pi++;
    return pi == pl;
}

// ---- a boolean tree over a relation -> the selection of its true rows (ExprExec.executeSelect).
// table mode (a lazy single-lane relation): ph_bool.col is a TABLE column, selections hold row ids of the table;
// positional mode: ph_bool.col is an output column of the relation, selections hold positions.
// sel_in == nullptr: all N rows. The result is ascending (AND narrows, OR unites in row order).
int eval_bool(ph_plan *p, Rel *r, bool table_mode, const BoolTree &bt, int idx, const int32_t *sel_in, int64_t n_in,
              const int32_t **sel_out, int64_t *n_out) {
    ph_ctx *ctx = p->ctx;
    const int64_t N = table_mode ? r->lanes[0].t->nrows : r->n;
    if (idx < 0 || idx >= (int)bt.nodes.size()) { set_error("ph_plan: boolean node %d out of range", idx); return PH_EINVAL; }
    const ph_bool &b = bt.nodes[(size_t)idx];
    if (n_in == 0) { *sel_out = sel_in; *n_out = 0; return PH_OK; }
    auto view_of = [&](int col, ph_col *v, const ph_table **dt, int *dc) -> int {
        if (table_mode) {
            if (col < 0 || col >= (int)r->lanes[0].t->cols.size()) { set_error("ph_plan: predicate column %d out of range", col); return PH_EINVAL; }
            *v = table_view(r->lanes[0].t, col); *dt = r->lanes[0].t; *dc = col;
            return PH_OK;
        }
        if (col < 0 || col >= (int)r->cols.size()) { set_error("ph_plan: predicate column %d out of range", col); return PH_EINVAL; }
        PL_CHECK(positional(p, r, {col}));
        const int32_t *s = nullptr;
        *v = col_view(*r, r->cols[(size_t)col], &s);
        *dt = r->cols[(size_t)col].src; *dc = r->cols[(size_t)col].src_col;
        return PH_OK;
    };
    switch (b.kind) {
    case PH_B_CMP: {
        ph_col v{};
        const ph_table *dt = nullptr;
        int dc = -1;
        PL_CHECK(view_of(b.col, &v, &dt, &dc));
        ph_const k = b.k;
        void *out = nullptr;
        PL_CHECK(palloc(p, n_in * 4, &out));
        int64_t m = 0;
        if (k.type == PH_COLREF) {
            ph_col v2{};
            const ph_table *dt2 = nullptr;
            int dc2 = -1;
            PL_CHECK(view_of((int)k.i, &v2, &dt2, &dc2));
            PL_CHECK(ph_filter_select_cols(ctx, &v, &v2, N, b.op, sel_in, n_in, (int32_t *)out, &m));
        } else if (v.type == PH_CODE8 && (b.op == PH_LIKE || b.op == PH_NOTLIKE) && k.type == PH_STR) {
            // LIKE over a dictionary column: the pattern is matched against the dictionary on the host; the codes that
            // match form runs (a prefix pattern over a dictionary in byte order: one run), each a range predicate
            if (!dt || dc < 0 || !k.s) { set_error("ph_plan: LIKE over a dictionary column needs the table's dictionary"); return PH_EUNSUPPORTED; }
            const auto &dict = dt->cols[(size_t)dc].dict;
            std::vector<std::pair<int, int>> runs;
            for (int c = 0; c < (int)dict.size(); c++) {
                const bool hit = host_like(dict[(size_t)c].data(), dict[(size_t)c].size(), k.s, strlen(k.s)) == (b.op == PH_LIKE);
                if (!hit) continue;
                if (!runs.empty() && runs.back().second == c - 1) runs.back().second = c; else runs.push_back({c, c});
            }
            std::vector<const int32_t *> sels;
            std::vector<int64_t> counts;
            if (runs.size() > 1 && !getenv("PH_PLAN_NO_IN_LIST")) {   // scattered codes ('%TIN' over p_type: 30 of 150): ONE pass over a bitmap of the codes
                std::vector<int64_t> codes;
                for (auto &rn : runs) for (int c = rn.first; c <= rn.second; c++) codes.push_back(c);
                PL_CHECK(ph_filter_select_in(ctx, &v, N, codes.data(), (int32_t)codes.size(), sel_in, n_in, (int32_t *)out, &m));
                runs.clear();
                sels.push_back((const int32_t *)out); counts.push_back(m);
            }
            for (auto &rn : runs) {
                ph_const kr{};
                kr.type = PH_CODE8; kr.i = rn.first; kr.scale = rn.second;
                void *o2 = nullptr;
                PL_CHECK(palloc(p, n_in * 4, &o2));
                int64_t m2 = 0;
                PL_CHECK(ph_filter_select(ctx, &v, N, PH_EQ, &kr, sel_in, n_in, (int32_t *)o2, &m2));
                sels.push_back((const int32_t *)o2); counts.push_back(m2);
            }
            if (sels.size() == 1) { out = const_cast<int32_t *>(sels[0]); m = counts[0]; }
            else if (!sels.empty()) PL_CHECK(ph_sel_union(ctx, sels.data(), counts.data(), (int32_t)sels.size(), N, (int32_t *)out, &m));
        } else if (!table_mode && r->cols[(size_t)b.col].sdict && k.type == PH_STR) {
            // a computed VARCHAR column (codes) against a VARCHAR constant: `=` / `<>` on the constant's code (equalStrOp / notEqualStrOp)
            if (b.op != PH_EQ && b.op != PH_NE) { set_error("ph_plan: only = / <> over a computed VARCHAR column"); return PH_EUNSUPPORTED; }
            int32_t code = -2;
            PL_CHECK(const_code(p, r->cols[(size_t)b.col].sdict, k.s, &code));
            ph_const kc{};
            kc.type = PH_I32; kc.i = code;
            PL_CHECK(ph_filter_select(ctx, &v, N, b.op, &kc, sel_in, n_in, (int32_t *)out, &m));
        } else {
            fix_dict_const(dt, dc, &k);
            fix_num_const(v, &k);
            PL_CHECK(ph_filter_select(ctx, &v, N, b.op, &k, sel_in, n_in, (int32_t *)out, &m));
        }
        *sel_out = (const int32_t *)out;
        *n_out = m;
        return PH_OK;
    }
    case PH_B_AND: {
        const int32_t *cur = sel_in;
        int64_t cnt = n_in;
        for (int c = 0; c < b.nchildren && cnt > 0; c++) {
            const int32_t *o = nullptr;
            int64_t m = 0;
            PL_CHECK(eval_bool(p, r, table_mode, bt, b.first_child + c, cur, cnt, &o, &m));
            cur = o; cnt = m;
        }
        if (!cur) { void *o = nullptr; PL_CHECK(palloc(p, 8, &o)); cur = (const int32_t *)o; }   // (an AND without children keeps everything: not produced by a binder)
        *sel_out = cur;
        *n_out = cnt;
        return PH_OK;
    }
    case PH_B_OR: {
        // an IN list — every child `col = constant` over ONE integer / dictionary-code / computed-VARCHAR column: one pass (ph_filter_select_in), the
        // string constants' codes looked up together
        if (b.nchildren >= 2 && b.nchildren <= 16 && !getenv("PH_PLAN_NO_IN_LIST")) {
            bool in_list = true;
            const ph_bool &c0 = bt.nodes[(size_t)b.first_child];
            for (int c = 0; c < b.nchildren && in_list; c++) {
                const ph_bool &ch = bt.nodes[(size_t)(b.first_child + c)];
                in_list = ch.kind == PH_B_CMP && ch.op == PH_EQ && ch.col == c0.col && ch.k.type != PH_COLREF;
            }
            if (in_list) {
                ph_col v{};
                const ph_table *dt = nullptr;
                int dc = -1;
                PL_CHECK(view_of(c0.col, &v, &dt, &dc));
                const bool computed = !table_mode && r->cols[(size_t)c0.col].sdict != nullptr;
                std::vector<int64_t> vals;
                bool ok = v.type == PH_I32 || v.type == PH_CODE8;
                if (ok && computed) {
                    std::vector<const char *> strs;
                    for (int c = 0; c < b.nchildren && ok; c++) { const ph_const &k = bt.nodes[(size_t)(b.first_child + c)].k; ok = k.type == PH_STR && k.s; strs.push_back(k.s); }
                    std::vector<int32_t> codes;
                    if (ok) { PL_CHECK(const_codes(p, r->cols[(size_t)c0.col].sdict, strs, &codes)); for (int32_t cd : codes) if (cd >= 0) vals.push_back(cd); }
                } else if (ok) {
                    for (int c = 0; c < b.nchildren && ok; c++) {
                        ph_const k = bt.nodes[(size_t)(b.first_child + c)].k;
                        fix_dict_const(dt, dc, &k);
                        fix_num_const(v, &k);
                        ok = k.type == PH_I32;   // an INTEGER constant, or the code of a dictionary string (a string not in the dictionary: a code no row holds)
                        vals.push_back(k.i);
                    }
                }
                if (ok) {
                    void *out = nullptr;
                    PL_CHECK(palloc(p, n_in * 4, &out));
                    int64_t m = 0;
                    const int rc = vals.empty() ? PH_OK : ph_filter_select_in(ctx, &v, N, vals.data(), (int32_t)vals.size(), sel_in, n_in, (int32_t *)out, &m);
                    if (rc == PH_OK) { *sel_out = (const int32_t *)out; *n_out = m; return PH_OK; }
                    if (rc != PH_EUNSUPPORTED) return rc;
                }
            }
        }
        std::vector<const int32_t *> sels;
        std::vector<int64_t> counts;
        for (int c = 0; c < b.nchildren; c++) {
            const int32_t *o = nullptr;
            int64_t m = 0;
            PL_CHECK(eval_bool(p, r, table_mode, bt, b.first_child + c, sel_in, n_in, &o, &m));
            if (m > 0) { sels.push_back(o); counts.push_back(m); }
        }
        void *out = nullptr;
        PL_CHECK(palloc(p, n_in * 4, &out));
        int64_t m = 0;
        if (sels.size() == 1) { out = const_cast<int32_t *>(sels[0]); m = counts[0]; }
        else if (!sels.empty()) PL_CHECK(ph_sel_union(ctx, sels.data(), counts.data(), (int32_t)sels.size(), N, (int32_t *)out, &m));
        *sel_out = (const int32_t *)out;
        *n_out = m;
        return PH_OK;
    }
    default:
        set_error("ph_plan: boolean node %d has unknown kind %d", idx, b.kind);
        return PH_EINVAL;
    }
}

// ---- apply what still filters a lazy single-lane relation: conjunct by conjunct (execSelectAnd), then the marks
int apply_pending(ph_plan *p, Rel *r) {
    if (!r->lazy()) return PH_OK;
    ph_ctx *ctx = p->ctx;
    const ph_table *t = r->lanes[0].t;
    const int64_t N = t->nrows;
    const int32_t *sel = nullptr;
    int64_t cnt = N;
    std::vector<bool> done(r->pending.size(), false);
    for (size_t i = 0; i < r->pending.size() && cnt > 0; i++) {
        if (done[i]) continue;
        ph_pred pr = r->pending[i];
        fix_dict_const(t, pr.col, &pr.k);
        ph_col v = table_view(t, pr.col);
        fix_num_const(v, &pr.k);
        void *out = nullptr;
        PL_CHECK(palloc(p, cnt * 4, &out));
        int64_t m = 0;
        // a second conjunct over the same column (a date range) rides in the same pass when both are value ranges
        int rc = PH_EUNSUPPORTED;
        for (size_t j = i + 1; j < r->pending.size() && rc == PH_EUNSUPPORTED; j++) {
            if (done[j] || r->pending[j].col != pr.col) continue;
            ph_pred p2 = r->pending[j];
            fix_dict_const(t, p2.col, &p2.k);
            fix_num_const(v, &p2.k);
            rc = ph_filter_select_and(ctx, &v, N, pr.op, &pr.k, p2.op, &p2.k, sel, cnt, (int32_t *)out, &m);
            if (rc == PH_OK) done[j] = true;
            else if (rc != PH_EUNSUPPORTED) return rc;
        }
        if (rc == PH_EUNSUPPORTED) PL_CHECK(ph_filter_select(ctx, &v, N, pr.op, &pr.k, sel, cnt, (int32_t *)out, &m));
        sel = (const int32_t *)out;
        cnt = m;
    }
    for (size_t i = 0; i < r->complex.size() && cnt > 0; i++) {
        const int32_t *o = nullptr;
        int64_t m = 0;
        if (sel == nullptr) {   // eval_bool's "all rows" is sel_in == nullptr with n_in == N
            PL_CHECK(eval_bool(p, r, true, r->complex[i], 0, nullptr, N, &o, &m));
        } else PL_CHECK(eval_bool(p, r, true, r->complex[i], 0, sel, cnt, &o, &m));
        sel = o;
        cnt = m;
    }
    if (r->flags && cnt > 0) {
        ph_col f{};
        f.type = PH_CODE8; f.data = r->flags;
        ph_const one{};
        one.type = PH_I32; one.i = 1;   // marks are 0 / 1
        void *out = nullptr;
        PL_CHECK(palloc(p, cnt * 4, &out));
        int64_t m = 0;
        PL_CHECK(ph_filter_select(ctx, &f, N, PH_EQ, &one, sel, cnt, (int32_t *)out, &m));
        sel = (const int32_t *)out;
        cnt = m;
    }
    if (!sel) { void *out = nullptr; PL_CHECK(palloc(p, 8, &out)); sel = (const int32_t *)out; }
    note(p, "  select %lld of %lld rows (%zu conjuncts%s)", (long long)cnt, (long long)N, r->pending.size() + r->complex.size(), r->flags ? " + marks" : "");
    r->lanes[0].rows = sel;
    r->n = cnt;
    r->pending.clear();
    r->complex.clear();
    r->flags = nullptr;
    return PH_OK;
}

// ---- make the listed columns positional (late materialisation: one gather pass per lane, up to 8 columns each)
int positional(ph_plan *p, Rel *r, const std::vector<int> &want) {
    PL_CHECK(apply_pending(p, r));
    for (size_t L = 0; L < r->lanes.size(); L++) {
        std::vector<int> cs;
        for (int c : want)
            if (r->cols[(size_t)c].lane == (int)L && std::find(cs.begin(), cs.end(), c) == cs.end()) cs.push_back(c);
        // several output columns may reference the same table column: gather it once
        for (size_t base = 0; base < cs.size();) {
            std::vector<int> batch;       // distinct table columns of this pass
            std::vector<int> members;     // output columns served by it
            size_t k = base;
            for (; k < cs.size(); k++) {
                const int tc = r->cols[(size_t)cs[k]].tcol;
                if (std::find(batch.begin(), batch.end(), tc) == batch.end()) {
                    if (batch.size() == 8) break;
                    batch.push_back(tc);
                }
                members.push_back(cs[k]);
            }
            base = k;
            const Lane &ln = r->lanes[L];
            std::vector<const void *> outp(batch.size());
            if (!ln.rows) {   // identity over the whole table: the table column IS the positional column
                for (size_t b = 0; b < batch.size(); b++) outp[b] = ln.t->cols[(size_t)batch[b]].data;
            } else {
                std::vector<ph_col> views;
                std::vector<void *> outs(batch.size());
                for (size_t b = 0; b < batch.size(); b++) {
                    ph_col v = table_view(ln.t, batch[b]);
                    const int w = width_of(v.type);
                    if (w == 0 || v.validity) { set_error("ph_plan: column %d of a joined table cannot be materialised (VARCHAR bytes / NULL-able)", batch[b]); return PH_EUNSUPPORTED; }
                    views.push_back(v);
                    PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * w, &outs[b]));
                    outp[b] = outs[b];
                }
                if (r->n > 0) {
                    if (views.size() == 1) PL_CHECK(ph_gather(p->ctx, &views[0], ln.rows, r->n, outs[0]));
                    else PL_CHECK(ph_gather_multi(p->ctx, (int32_t)views.size(), views.data(), ln.rows, r->n, outs.data()));
                }
            }
            const uint8_t *lane_validity = nullptr;
            if (ln.rows && ln.nullable) {   // (the gathers read row 0 for a negative id; the bitmap says which values are real)
                void *vb = nullptr;
                PL_CHECK(palloc(p, (std::max<int64_t>(r->n, 1) + 7) / 8 + 64, &vb));
                if (r->n > 0) PL_CHECK(ph_rowid_validity(p->ctx, ln.rows, r->n, (uint8_t *)vb));
                lane_validity = (const uint8_t *)vb;
            }
            for (int c : members) {
                PCol &pc = r->cols[(size_t)c];
                const size_t b = (size_t)(std::find(batch.begin(), batch.end(), pc.tcol) - batch.begin());
                pc.data = outp[b];
                pc.validity = ln.rows ? lane_validity : ln.t->cols[(size_t)pc.tcol].validity;
                pc.lane = -1;
                pc.tcol = -1;
            }
        }
    }
    return PH_OK;
}

// lanes no column references any more are dropped (their row ids would only be dragged through later compactions)
void drop_unused_lanes(Rel *r) {
    std::vector<int> remap(r->lanes.size(), -1);
    std::vector<Lane> keep;
    for (size_t L = 0; L < r->lanes.size(); L++) {
        bool used = false;
        for (auto &c : r->cols) used |= c.lane == (int)L;
        if (used) { remap[L] = (int)keep.size(); keep.push_back(r->lanes[L]); }
    }
    if (keep.empty() && !r->lanes.empty()) return;   // keep at least the shape of a lazy scan
    for (auto &c : r->cols) if (c.lane >= 0) c.lane = remap[(size_t)c.lane];
    r->lanes = keep;
}

// ---- keep rows idx[0..m) (positions) of a relation: every lane's row ids and every positional column
int compact(ph_plan *p, Rel *r, const int32_t *idx, int64_t m) {
    for (auto &ln : r->lanes) {
        if (!ln.rows) { ln.rows = idx; continue; }   // identity: position = row id
        ph_col v{};
        v.type = PH_I32; v.data = ln.rows;
        void *out = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(m, 1) * 4, &out));
        if (m > 0) PL_CHECK(ph_gather(p->ctx, &v, idx, m, out));
        ln.rows = (const int32_t *)out;
    }
    std::vector<ph_col> views;
    std::vector<void *> outs;
    std::vector<PCol *> cols;
    auto flush = [&]() -> int {
        if (views.empty()) return PH_OK;
        if (m > 0) {
            if (views.size() == 1) PL_CHECK(ph_gather(p->ctx, &views[0], idx, m, outs[0]));
            else PL_CHECK(ph_gather_multi(p->ctx, (int32_t)views.size(), views.data(), idx, m, outs.data()));
        }
        for (size_t i = 0; i < cols.size(); i++) cols[i]->data = outs[i];
        views.clear(); outs.clear(); cols.clear();
        return PH_OK;
    };
    for (auto &c : r->cols) {
        if (c.lane >= 0) continue;
        if (c.validity) { set_error("ph_plan: NULL-able intermediate column in a compaction"); return PH_EUNSUPPORTED; }
        ph_col v{};
        v.type = c.type; v.scale = c.scale; v.data = c.data;
        void *out = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(m, 1) * width_of(c.type), &out));
        views.push_back(v); outs.push_back(out); cols.push_back(&c);
        if (views.size() == 8) PL_CHECK(flush());
    }
    PL_CHECK(flush());
    r->n = m;
    return PH_OK;
}

// ---- expression over a relation's columns -> a positional decimal column
int eval_rpn(ph_plan *p, Rel *r, const ph_rpn *prog, int nprog, PCol *out) {
    PL_CHECK(apply_pending(p, r));
    std::vector<int> operands;
    for (int i = 0; i < nprog; i++)
        if (prog[i].op == PH_X_COL && std::find(operands.begin(), operands.end(), prog[i].col) == operands.end()) operands.push_back(prog[i].col);
    for (int c : operands) if (c < 0 || c >= (int)r->cols.size()) { set_error("ph_plan: expression column %d out of range", c); return PH_EINVAL; }
    // all operands in one lane: evaluate straight over the table's columns through the lane's row ids (a fused gather)
    int lane = operands.empty() ? -1 : r->cols[(size_t)operands[0]].lane;
    for (int c : operands) if (r->cols[(size_t)c].lane != lane) lane = -2;
    if (lane >= 0 && r->lanes[(size_t)lane].nullable) lane = -2;   // row ids of -1 (a LEFT join's build side): gathered with their validity first
    if (lane < 0 && !operands.empty()) PL_CHECK(positional(p, r, operands));
    std::vector<ph_col> views;
    const int32_t *sel = nullptr;
    bool any_validity = false;
    for (int c : operands) {
        const int32_t *s = nullptr;
        views.push_back(col_view(*r, r->cols[(size_t)c], &s));
        sel = s;
        any_validity |= views.back().validity != nullptr;
    }
    std::vector<ph_rpn> pr(prog, prog + nprog);
    for (auto &o : pr) if (o.op == PH_X_COL) o.col = (int32_t)(std::find(operands.begin(), operands.end(), o.col) - operands.begin());
    int32_t scale = 0;
    PL_CHECK(ph_expr_scale(views.data(), pr.data(), nprog, &scale));
    void *o = nullptr, *val = nullptr;
    PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 8, &o));
    if (any_validity) PL_CHECK(palloc(p, (r->n + 7) / 8 + 64, &val));
    if (r->n > 0) PL_CHECK(ph_expr_eval(p->ctx, views.data(), (int32_t)views.size(), pr.data(), nprog, sel, r->n, (int64_t *)o, (uint8_t *)val));
    *out = PCol{};
    out->type = PH_DEC64; out->scale = scale; out->data = o; out->validity = (const uint8_t *)val;
    return PH_OK;
}

// ---- FLOAT / DOUBLE program over a relation's columns -> a positional column: the int32 truth value of a comparison, or float32 values
int eval_float(ph_plan *p, Rel *r, const ph_plan_expr &e, PCol *out) {
    PL_CHECK(apply_pending(p, r));
    std::vector<int> operands;
    for (int i = 0; i < e.nprog; i++)
        if (e.prog[i].op == PH_X_COL && std::find(operands.begin(), operands.end(), e.prog[i].col) == operands.end()) operands.push_back(e.prog[i].col);
    if (operands.empty()) { set_error("ph_plan: a FLOAT expression without a column"); return PH_EUNSUPPORTED; }
    for (int c : operands) if (c < 0 || c >= (int)r->cols.size()) { set_error("ph_plan: expression column %d out of range", c); return PH_EINVAL; }
    int lane = r->cols[(size_t)operands[0]].lane;
    for (int c : operands) if (r->cols[(size_t)c].lane != lane) lane = -2;
    if (lane >= 0 && r->lanes[(size_t)lane].nullable) lane = -2;
    if (lane < 0) PL_CHECK(positional(p, r, operands));
    std::vector<ph_col> views;
    const int32_t *sel = nullptr;
    bool any_validity = false;
    for (int c : operands) {
        const int32_t *s = nullptr;
        views.push_back(col_view(*r, r->cols[(size_t)c], &s));
        sel = s;
        any_validity |= views.back().validity != nullptr;
    }
    std::vector<ph_rpn> pr(e.prog, e.prog + e.nprog);
    for (auto &o : pr) if (o.op == PH_X_COL) o.col = (int32_t)(std::find(operands.begin(), operands.end(), o.col) - operands.begin());
    const int32_t out_type = e.result_int ? PH_I32 : PH_F32;
    void *o = nullptr, *val = nullptr;
    PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &o));
    if (any_validity && !e.result_int) PL_CHECK(palloc(p, (r->n + 7) / 8 + 64, &val));
    if (r->n > 0) PL_CHECK(ph_float_eval(p->ctx, views.data(), (int32_t)views.size(), pr.data(), e.nprog, e.float_wide, sel, r->n, out_type, o, (uint8_t *)val));
    *out = PCol{};
    out->type = out_type; out->data = o; out->validity = (const uint8_t *)val;
    return PH_OK;
}

// expression program over the rows sel[0..m) (positions) of a relation -> m positional decimal values
int eval_rpn_at(ph_plan *p, Rel *r, const ph_rpn *prog, int nprog, const int32_t *sel, int64_t m, void **out, int32_t *scale) {
    std::vector<int> operands;
    for (int i = 0; i < nprog; i++)
        if (prog[i].op == PH_X_COL && std::find(operands.begin(), operands.end(), prog[i].col) == operands.end()) operands.push_back(prog[i].col);
    for (int c : operands) if (c < 0 || c >= (int)r->cols.size()) { set_error("ph_plan: expression column %d out of range", c); return PH_EINVAL; }
    if (!operands.empty()) PL_CHECK(positional(p, r, operands));
    std::vector<ph_col> views;
    for (int c : operands) {
        const int32_t *s = nullptr;
        views.push_back(col_view(*r, r->cols[(size_t)c], &s));
        if (views.back().validity) { set_error("ph_plan: NULL-able operand in a CASE branch"); return PH_EUNSUPPORTED; }
    }
    std::vector<ph_rpn> pr(prog, prog + nprog);
    for (auto &o : pr) if (o.op == PH_X_COL) o.col = (int32_t)(std::find(operands.begin(), operands.end(), o.col) - operands.begin());
    if (views.empty()) {   // a program of constants only (`THEN 1`) names no column: the evaluator still wants one to address
        void *z = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &z));
        PL_CHECK(ph_dev_memset(p->ctx, z, 0, std::max<int64_t>(r->n, 1) * 4));
        ph_col dummy{};
        dummy.type = PH_I32; dummy.data = z;
        views.push_back(dummy);
    }
    PL_CHECK(ph_expr_scale(views.data(), pr.data(), nprog, scale));
    PL_CHECK(palloc(p, std::max<int64_t>(m, 1) * 8, out));
    if (m > 0) PL_CHECK(ph_expr_eval(p->ctx, views.data(), (int32_t)views.size(), pr.data(), nprog, sel, m, (int64_t *)*out, nullptr));
    return PH_OK;
}

bool is_int_const(const ph_rpn *prog, int n) { return n == 1 && prog[0].op == PH_X_CONST && prog[0].scale == 0; }

int eval_expr(ph_plan *p, Rel *r, const Expr &ex, PCol *out) {
    const ph_plan_expr &e = ex.e;
    switch (e.kind) {
    case PH_PE_CASE: {
        // executeCase (expr_exec.go:144-246): the WHEN is a select over the rows, THEN is evaluated on its true rows and
        // filled in at those rows (FillSwitch), ELSE on the remaining rows
        PL_CHECK(apply_pending(p, r));
        const int64_t n = r->n;
        const int32_t *st = nullptr;
        int64_t nt = 0;
        if (n > 0) PL_CHECK(eval_bool(p, r, false, ex.when, 0, nullptr, n, &st, &nt));
        int32_t ts = 0, es = 0;
        void *tv = nullptr, *outv = nullptr;
        PL_CHECK(eval_rpn_at(p, r, e.prog, e.nprog, st, nt, &tv, &ts));
        PL_CHECK(palloc(p, std::max<int64_t>(n, 1) * 8, &outv));
        if (is_int_const(e.else_prog, e.nelse)) {
            // `ELSE <integer literal>`: cast to the THEN branch's type — the unscaled value at the result scale
            long long v = e.else_prog[0].ival;
            for (int i = 0; i < ts; i++) if (__builtin_mul_overflow(v, 10ll, &v)) { set_error("ph_plan: CASE ELSE constant overflows"); return PH_EOVERFLOW; }
            if (v == 0) PL_CHECK(ph_dev_memset(p->ctx, outv, 0, std::max<int64_t>(n, 1) * 8));
            else {
                ph_rpn k{PH_X_CONST, -1, v, ts};
                void *cv = nullptr;
                int32_t cs = 0;
                PL_CHECK(eval_rpn_at(p, r, &k, 1, nullptr, n, &cv, &cs));
                outv = cv;
            }
        } else {
            void *sf = nullptr, *ev = nullptr;
            int64_t nf = 0;
            PL_CHECK(palloc(p, std::max<int64_t>(n, 1) * 4, &sf));
            if (n > 0) PL_CHECK(ph_sel_difference(p->ctx, nullptr, n, st, nt, n, (int32_t *)sf, &nf));
            PL_CHECK(eval_rpn_at(p, r, e.else_prog, e.nelse, (const int32_t *)sf, nf, &ev, &es));
            if (es != ts) { set_error("ph_plan: CASE branches of different scales (%d / %d)", ts, es); return PH_EUNSUPPORTED; }
            ph_col vals{};
            vals.type = PH_DEC64; vals.scale = es; vals.data = ev;
            if (nf > 0) PL_CHECK(ph_scatter(p->ctx, &vals, (const int32_t *)sf, nf, outv, nullptr));
        }
        ph_col tvals{};
        tvals.type = PH_DEC64; tvals.scale = ts; tvals.data = tv;
        if (nt > 0) PL_CHECK(ph_scatter(p->ctx, &tvals, st, nt, outv, nullptr));
        *out = PCol{};
        out->type = PH_DEC64; out->scale = ts; out->data = outv;
        if (e.result_int) out->scale = 0;   // INTEGER result carried as a scale-0 decimal column (the sums are the same numbers)
        return PH_OK;
    }
    case PH_PE_COL:
        if (e.col < 0 || e.col >= (int)r->cols.size()) { set_error("ph_plan: column %d out of range", e.col); return PH_EINVAL; }
        *out = r->cols[(size_t)e.col];
        return PH_OK;
    case PH_PE_DECIMAL:
        return eval_rpn(p, r, e.prog, e.nprog, out);
    case PH_PE_YEAR: {
        if (e.col < 0 || e.col >= (int)r->cols.size() || r->cols[(size_t)e.col].type != PH_DATE) { set_error("ph_plan: extract(year) needs a DATE column"); return PH_EINVAL; }
        PL_CHECK(apply_pending(p, r));
        if (r->cols[(size_t)e.col].lane >= 0 && r->lanes[(size_t)r->cols[(size_t)e.col].lane].nullable) { set_error("ph_plan: extract over a NULL-able column"); return PH_EUNSUPPORTED; }
        const int32_t *sel = nullptr;
        ph_col v = col_view(*r, r->cols[(size_t)e.col], &sel);
        if (v.validity) { set_error("ph_plan: extract over a NULL-able column"); return PH_EUNSUPPORTED; }
        void *o = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &o));
        if (r->n > 0) PL_CHECK(ph_date_extract(p->ctx, PH_PART_YEAR, &v, sel, r->n, (int32_t *)o));
        *out = PCol{};
        out->type = PH_I32; out->data = o;
        out->ordered = r->cols[(size_t)e.col].ordered;   // the year of a non-decreasing date is non-decreasing
        return PH_OK;
    }
    case PH_PE_FLOAT:
        return eval_float(p, r, e, out);
    case PH_PE_SUBSTR: {
        // substring(<VARCHAR table column> FROM offset FOR length): the result is interned at once — a positional int32 column of
        // string codes (equal strings, equal codes: what a filter's `=`, a group key and a join key need) whose dictionary rows are the
        // substrings themselves, kept as a one-column relation the plan owns so that a code leads back to its bytes (ph_table_strings)
        if (e.col < 0 || e.col >= (int)r->cols.size()) { set_error("ph_plan: column %d out of range", e.col); return PH_EINVAL; }
        PL_CHECK(apply_pending(p, r));
        const PCol &pc = r->cols[(size_t)e.col];
        if (pc.type != PH_STR || pc.lane < 0) { set_error("ph_plan: substring needs a VARCHAR table column"); return PH_EUNSUPPORTED; }
        const Lane &ln = r->lanes[(size_t)pc.lane];
        if (ln.nullable) { set_error("ph_plan: substring over a NULL-able column"); return PH_EUNSUPPORTED; }
        ph_col v = table_view(ln.t, pc.tcol);
        const int64_t n = r->n;
        // capacity: a relation whose row ids do not repeat cannot ask for more bytes than the column holds; row ids that repeat (a small
        // VARCHAR table fanned out below a join) can — up to n * sub_length for a bounded length — so that case starts from the length
        // bound when it is small and otherwise retries once with the exact size ph_substring reports (as fetch_rows_once does)
        const bool repeats = ln.rows && !ln.dup_free;
        int64_t cap = v.aux_bytes + 64;
        if (e.sub_length >= 0 && e.sub_length <= 256) cap = repeats ? n * e.sub_length + 64 : std::min(cap, n * e.sub_length + 64);
        if (cap >= (1ll << 31)) cap = v.aux_bytes + 64;   // (the exact size decides)
        void *off = nullptr, *bytes = nullptr, *codes = nullptr;
        PL_CHECK(p->ctx->pool_alloc((n + 1) * 4, &off));
        p->computed_bufs.push_back(off);
        PL_CHECK(palloc(p, std::max<int64_t>(n, 1) * 4, &codes));
        int64_t nbytes = 0;
        for (int attempt = 0; attempt < 2; attempt++) {
            PL_CHECK(p->ctx->pool_alloc(cap, &bytes));
            p->computed_bufs.push_back(bytes);
            int rc = ph_substring(p->ctx, &v, e.sub_offset, e.sub_length, ln.rows, n, (int32_t *)off, (uint8_t *)bytes, cap, &nbytes);
            if (rc == PH_ECAPACITY && attempt == 0 && nbytes > cap) { cap = nbytes + 64; continue; }
            PL_CHECK(rc);
            break;
        }
        ph_table *vt = new ph_table();
        vt->ctx = p->ctx;
        vt->nrows = n;
        vt->cols.resize(1);
        vt->cols[0].type = PH_STR; vt->cols[0].data = off; vt->cols[0].aux = bytes; vt->cols[0].aux_bytes = nbytes;
        vt->cols[0].validity = nullptr;
        p->computed.push_back(vt);
        if (v.validity) { set_error("ph_plan: substring over a NULL-able column"); return PH_EUNSUPPORTED; }
        ph_col sv = table_view(vt, 0);
        ph_strdict *d = nullptr;
        PL_CHECK(ph_strdict_build(p->ctx, &sv, nullptr, n, (int32_t *)codes, &d));
        p->strdicts.push_back(d);
        *out = PCol{};
        out->type = PH_I32; out->data = codes;
        out->src = vt; out->src_col = 0;
        out->sdict = d;
        return PH_OK;
    }
    default:
        set_error("ph_plan: unknown expression kind %d", e.kind);
        return PH_EINVAL;
    }
}

// a VARCHAR constant compared with a computed VARCHAR column: its code in the column's dictionary (-2 = no row holds that string:
// `=` selects nothing, `<>` everything — codes are >= 0)
int const_code(ph_plan *p, ph_strdict *d, const char *s, int32_t *code) {
    const int64_t len = s ? (int64_t)strlen(s) : 0;
    void *off = nullptr, *bytes = nullptr, *out = nullptr;
    PL_CHECK(palloc(p, 8, &off));
    PL_CHECK(palloc(p, len + 64, &bytes));
    PL_CHECK(palloc(p, 8, &out));
    const int32_t offs[2] = {0, (int32_t)len};
    PL_CHECK(ph_dev_upload(p->ctx, off, offs, 8));
    if (len > 0) PL_CHECK(ph_dev_upload(p->ctx, bytes, s, len));
    ph_col c{};
    c.type = PH_STR; c.data = off; c.aux = bytes; c.aux_bytes = len;
    PL_CHECK(ph_strdict_lookup(d, &c, nullptr, 1, (int32_t *)out));
    return p->ctx->download(code, out, 4);
}

// the codes of several constants in one lookup (an IN list over a computed VARCHAR column): -2 = not in the dictionary
int const_codes(ph_plan *p, ph_strdict *d, const std::vector<const char *> &strs, std::vector<int32_t> *codes) {
    const int64_t k = (int64_t)strs.size();
    std::vector<int32_t> offs((size_t)k + 1, 0);
    std::string bytes;
    for (int64_t i = 0; i < k; i++) { bytes += strs[(size_t)i] ? strs[(size_t)i] : ""; offs[(size_t)i + 1] = (int32_t)bytes.size(); }
    void *off = nullptr, *by = nullptr, *out = nullptr;
    PL_CHECK(palloc(p, (k + 1) * 4 + 8, &off));
    PL_CHECK(palloc(p, (int64_t)bytes.size() + 64, &by));
    PL_CHECK(palloc(p, k * 4 + 8, &out));
    PL_CHECK(ph_dev_upload(p->ctx, off, offs.data(), (k + 1) * 4));
    if (!bytes.empty()) PL_CHECK(ph_dev_upload(p->ctx, by, bytes.data(), (int64_t)bytes.size()));
    ph_col c{};
    c.type = PH_STR; c.data = off; c.aux = by; c.aux_bytes = (int64_t)bytes.size();
    PL_CHECK(ph_strdict_lookup(d, &c, nullptr, k, (int32_t *)out));
    codes->assign((size_t)k, -2);
    return p->ctx->download(codes->data(), out, k * 4);
}

// ---- VARCHAR keys: a PH_STR column of the relation becomes a positional int32 column of string codes (ph_strdict_*):
// equal strings, equal codes. dict == nullptr: intern the column's own rows (a group key, a join's build key; *dict_out
// receives the dictionary); else look the rows up in that dictionary (a join's probe key).
int string_codes(ph_plan *p, Rel *r, int c, ph_strdict *dict, ph_strdict **dict_out, PCol *out) {
    PL_CHECK(apply_pending(p, r));
    const PCol &pc = r->cols[(size_t)c];
    if (pc.lane < 0) { set_error("ph_plan: a VARCHAR key must be a table column (offsets + bytes cannot be gathered)"); return PH_EUNSUPPORTED; }
    const Lane &ln = r->lanes[(size_t)pc.lane];
    if (ln.nullable) { set_error("ph_plan: a VARCHAR key from the NULL-able side of a LEFT join"); return PH_EUNSUPPORTED; }
    ph_col v = table_view(ln.t, pc.tcol);
    void *codes = nullptr;
    PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &codes));
    if (dict) PL_CHECK(ph_strdict_lookup(dict, &v, ln.rows, r->n, (int32_t *)codes));
    else {
        ph_strdict *d = nullptr;
        PL_CHECK(ph_strdict_build(p->ctx, &v, ln.rows, r->n, (int32_t *)codes, &d));
        p->strdicts.push_back(d);
        if (dict_out) *dict_out = d;
    }
    *out = PCol{};
    out->type = PH_I32; out->data = codes;
    out->src = ln.t; out->src_col = pc.tcol;   // a code is a row of this column: where the host finds the string
    return PH_OK;
}

// ---- is the (multi-column) key of this relation unique? only base-table keys the statistics or the catalog vouch for
// 0 = no; 2 = the key IS a unique column set (a foreign key into it finds exactly one row); 1 = it CONTAINS one (still at
// most one match, but the extra columns act as a filter: misses are expected)
int key_unique(const Rel &r, const std::vector<int32_t> &keys) {
    int lane = -1;
    std::vector<int32_t> tcols;
    const ph_table *t = nullptr;
    for (int32_t k : keys) {
        const PCol &c = r.cols[(size_t)k];
        if (c.lane < 0) {
            // a positional copy of a table column keeps the table's uniqueness only in a relation of that one table
            if (!c.src || r.lanes.size() != 1 || r.lanes[0].t != c.src || !r.lanes[0].dup_free) return 0;
            if (lane >= 0 && lane != 0) return 0;
            lane = 0; t = c.src; tcols.push_back(c.src_col);
            continue;
        }
        if (lane >= 0 && c.lane != lane) continue;   // a key column of another table: an extra filter, at best
        if (lane < 0) { lane = c.lane; t = r.lanes[(size_t)lane].t; }
        tcols.push_back(c.tcol);
    }
    if (lane < 0 || !r.lanes[(size_t)lane].dup_free) return 0;
    std::sort(tcols.begin(), tcols.end());
    tcols.erase(std::unique(tcols.begin(), tcols.end()), tcols.end());
    int best = 0;
    for (int32_t c : tcols) if (t->cols[(size_t)c].strict) best = std::max(best, tcols.size() == 1 && keys.size() == 1 ? 2 : 1);
    for (auto &u : t->unique_keys)
        if (std::includes(tcols.begin(), tcols.end(), u.begin(), u.end())) best = std::max(best, u.size() == tcols.size() && keys.size() == tcols.size() ? 2 : 1);
    return best;
}

int lower(ph_plan *p, int idx, bool as_build, Rel *out);
int whole_groups(ph_plan *p, int idx, Rel *R);
int replicate_rel(ph_plan *p, Rel *r, const char *what);
int sink_into_agg(ph_plan *p, int idx, Rel &R, bool allow_pack, ph_agg **aggp, std::vector<KeyInfo> *kinfo, std::vector<int32_t> *ascale,
                  std::vector<int32_t> *atype, std::vector<KeyPack> *packs, std::vector<bool> *nullable = nullptr);

struct KeySide {            // the key columns of one join side as the kernels want them
    std::vector<ph_col> views;
    const int32_t *sel = nullptr;
    int64_t n = 0;
    bool rowids = false;    // the kernels report base row ids of lane 0 (table columns through the lane's row ids)
};

// key columns of a relation: table columns addressed through the one lane's row ids when the relation is a
// (filtered) base table, positional columns otherwise
int key_side(ph_plan *p, Rel *r, const std::vector<int32_t> &keys, KeySide *ks) {
    bool all_lane0 = r->lanes.size() == 1 && !r->lanes[0].nullable;
    for (int32_t k : keys) all_lane0 = all_lane0 && r->cols[(size_t)k].lane == 0;
    bool any_positional = false;
    for (auto &c : r->cols) any_positional |= c.lane < 0;
    if (all_lane0 && !any_positional) {
        for (int32_t k : keys) ks->views.push_back(table_view(r->lanes[0].t, r->cols[(size_t)k].tcol));
        ks->sel = r->lanes[0].rows;
        ks->n = r->lanes[0].rows ? r->n : r->lanes[0].t->nrows;
        ks->rowids = true;
        return PH_OK;
    }
    std::vector<int> want(keys.begin(), keys.end());
    PL_CHECK(positional(p, r, want));
    for (int32_t k : keys) { const int32_t *s = nullptr; ks->views.push_back(col_view(*r, r->cols[(size_t)k], &s)); }
    ks->sel = nullptr;
    ks->n = r->n;
    ks->rowids = false;
    return PH_OK;
}

// pairs (probe position / row id, build row id) of an inner probe; retried once with the exact size when a
// non-unique build side produced more pairs than probe rows
// cap_hint: the pairs expected when the build key has duplicates (its rows: a foreign key's side) — a pair list that does not fit costs the whole probe again
int pair_probe(ph_plan *p, ph_join *j, const KeySide &pk, const ph_pred *where, const ph_table *where_t, const uint8_t *residual,
               int64_t *m_out, int32_t **prow, int32_t **brow, const char **form, int64_t cap_hint = 0) {
    int64_t cap = std::max<int64_t>(std::max<int64_t>(pk.n, 1), std::min<int64_t>(cap_hint, 64ll << 20)), m = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        void *op = nullptr, *ob = nullptr;
        PL_CHECK(palloc(p, cap * 4, &op));
        PL_CHECK(palloc(p, cap * 4, &ob));
        int rc = PH_EUNSUPPORTED;
        if (where || residual) {
            ph_pred w{};
            ph_col wv{};
            if (where) { w = *where; fix_dict_const(where_t, w.col, &w.k); wv = table_view(where_t, w.col); }
            if (residual) { rc = ph_join_probe_inner_residual(j, pk.views.data(), where ? &wv : nullptr, w.op, where ? &w.k : nullptr, residual, pk.sel, pk.n, (int32_t *)op, (int32_t *)ob, cap, &m); *form = "fused filter+probe with residual marks"; }
            else { rc = ph_join_probe_inner_where(j, pk.views.data(), &wv, w.op, &w.k, pk.sel, pk.n, (int32_t *)op, (int32_t *)ob, cap, &m); *form = "fused filter+probe"; }
            if (rc == PH_EUNSUPPORTED) return rc;   // the caller applies the filter first
        } else {
            rc = ph_join_probe_inner(j, pk.views.data(), pk.sel, pk.n, (int32_t *)op, (int32_t *)ob, cap, &m);
            *form = "probe";
        }
        if (rc == PH_ECAPACITY && attempt == 0 && m > cap) { cap = m; continue; }
        PL_CHECK(rc);
        *m_out = m; *prow = (int32_t *)op; *brow = (int32_t *)ob;
        return PH_OK;
    }
    return PH_ECAPACITY;
}


// ================================================================== multi-rank execution (ph_plan_set_comm)
// No reference counterpart (the reference runs one goroutine). Every rank runs the SAME descriptor over its shard of the
// sharded tables (row ranges) and its copy of the replicated ones (ph_table_set_replicated), and the library inserts the exchanges:
//   join     build side replicated                      -> local
//            both sides co-located by key RANGE          -> local (the ranks' column statistics: build ranges pairwise disjoint, each rank's
//                                                           probe range inside its own build range — a database split by order ranges)
//            build side small (<= bcast_rows in all)     -> all-gathered into a replicated temporary table, then local
//            otherwise                                   -> both sides hash-partitioned by the (first) key and exchanged all-to-all, then local
//   aggregate below other operators: groups disjoint by the RANGE of a group key -> local; else its input is hash-partitioned by the first key
//   root aggregate: local; ph_plan_fetch gathers every rank's groups and merges the partial states on the host (128-bit sums, counts,
//            min / max) — or, with a top-k / HAVING announced (which need whole groups), the input is hash-partitioned first and the ranks'
//            results are concatenated.
// Every decision is taken from values all ranks hold alike (all-reduced counts and statistics), so all ranks walk the same sequence of
// collectives; broken statistics are agreed on at the end of the run (deferred errors held, one all-reduce) and all ranks rerun together.
bool multi(const ph_plan *p) { return p->comm != nullptr && ph_comm_nranks(p->comm) > 1; }

int global_sum(ph_plan *p, int64_t v, int64_t *out) {
    int64_t x = v;
    PL_CHECK(ph_comm_allreduce_i64(p->comm, &x, 1, PH_RED_SUM));
    *out = x;
    return PH_OK;
}

// the rank's own statistics of a column's source: false = unknown
bool local_range(const PCol &c, int64_t *lo, int64_t *hi, bool *empty) {
    *empty = false;
    if (!c.src || c.src_col < 0) return false;
    const auto &sc = c.src->cols[(size_t)c.src_col];
    if (c.src->nrows == 0) { *empty = true; return true; }
    if (!sc.has_range) return false;
    *lo = sc.min; *hi = sc.max;
    return true;
}

// [gmin, gmax] over all ranks of a column's source statistics
int global_range(ph_plan *p, const PCol &c, bool *ok, int64_t *gmin, int64_t *gmax) {
    int64_t lo = 0, hi = 0;
    bool empty = false;
    const bool known = local_range(c, &lo, &hi, &empty);
    int64_t v[3] = {known && !empty ? -lo : INT64_MIN + 1, known && !empty ? hi : INT64_MIN + 1, known ? 0 : 1};   // max(-lo) = -min(lo)
    PL_CHECK(ph_comm_allreduce_i64(p->comm, v, 3, PH_RED_MAX));
    *ok = v[2] == 0 && v[0] != INT64_MIN + 1;
    *gmin = -v[0]; *gmax = v[1];
    return PH_OK;
}

// per-rank (lo, hi, state) of up to two columns: state 0 = range known, 1 = no rows, 2 = unknown
int gather_ranges(ph_plan *p, const PCol *a, const PCol *b, std::vector<int64_t> *out) {
    const int n = ph_comm_nranks(p->comm), me = ph_comm_rank(p->comm), per = b ? 6 : 3;
    if (per * n > 64) { out->clear(); return PH_OK; }   // (beyond ten ranks: no range reasoning)
    std::vector<int64_t> v((size_t)per * n, 0);
    const PCol *cs[2] = {a, b};
    for (int k = 0; k < (b ? 2 : 1); k++) {
        int64_t lo = 0, hi = 0;
        bool empty = false;
        const bool known = local_range(*cs[k], &lo, &hi, &empty);
        v[(size_t)(me * per + k * 3 + 0)] = known && !empty ? lo : 0;
        v[(size_t)(me * per + k * 3 + 1)] = known && !empty ? hi : 0;
        v[(size_t)(me * per + k * 3 + 2)] = !known ? 2 : empty ? 1 : 0;
    }
    PL_CHECK(ph_comm_allreduce_i64(p->comm, v.data(), per * n, PH_RED_SUM));
    *out = v;
    return PH_OK;
}

// are the ranks' value ranges of this column pairwise disjoint? (then equal values never sit on two ranks)
int ranks_disjoint(ph_plan *p, const PCol &c, bool *yes) {
    *yes = false;
    std::vector<int64_t> v;
    PL_CHECK(gather_ranges(p, &c, nullptr, &v));
    if (v.empty()) return PH_OK;
    const int n = ph_comm_nranks(p->comm);
    for (int r = 0; r < n; r++) if (v[(size_t)r * 3 + 2] == 2) return PH_OK;
    for (int r = 0; r < n; r++)
        for (int q = r + 1; q < n; q++) {
            if (v[(size_t)r * 3 + 2] || v[(size_t)q * 3 + 2]) continue;
            if (!(v[(size_t)r * 3 + 1] < v[(size_t)q * 3] || v[(size_t)q * 3 + 1] < v[(size_t)r * 3])) return PH_OK;
        }
    *yes = true;
    return PH_OK;
}

// probe and build side co-located by key range: the build ranges are pairwise disjoint and every rank's probe range lies inside its own build range
int colocated(ph_plan *p, const PCol &pc, const PCol &bc, bool *yes) {
    *yes = false;
    std::vector<int64_t> v;
    PL_CHECK(gather_ranges(p, &pc, &bc, &v));
    if (v.empty()) return PH_OK;
    const int n = ph_comm_nranks(p->comm);
    for (int r = 0; r < n; r++) {
        const int64_t *x = &v[(size_t)r * 6];
        if (x[2] == 2 || x[5] == 2) return PH_OK;
        if (x[2] == 0 && (x[5] != 0 || x[0] < x[3] || x[1] > x[4])) return PH_OK;   // probe rows without a covering build range on this rank
    }
    for (int r = 0; r < n; r++)
        for (int q = r + 1; q < n; q++) {
            const int64_t *x = &v[(size_t)r * 6], *y = &v[(size_t)q * 6];
            if (x[5] || y[5]) continue;
            if (!(x[4] < y[3] || y[4] < x[3])) return PH_OK;
        }
    *yes = true;
    return PH_OK;
}

// what travels for a relation's columns: the row ids of lanes over REPLICATED tables (they mean the same on every rank) and the values of
// every other fixed-width column
struct Travel {
    std::vector<int> lane_of;            // per relation lane: index of its row-id travel column, or -1
    std::vector<int> col_of;             // per output column: index of its value travel column, or -1 (it rides on its lane)
    std::vector<const void *> data;      // travel columns (device, relation length)
    std::vector<int32_t> width;
};

int prepare_travel(ph_plan *p, Rel *r, Travel *tv, bool strings_ok, std::vector<int> *string_cols) {
    PL_CHECK(apply_pending(p, r));
    tv->lane_of.assign(r->lanes.size(), -1);
    tv->col_of.assign(r->cols.size(), -1);
    std::vector<int> need;
    for (size_t c = 0; c < r->cols.size(); c++) {
        const PCol &pc = r->cols[c];
        if (pc.sdict) { set_error("ph_plan: a computed VARCHAR column cannot cross ranks"); return PH_EUNSUPPORTED; }
        if (pc.lane >= 0 && r->lanes[(size_t)pc.lane].t->replicated) continue;     // rides on its lane's row ids
        if (pc.type == PH_STR) {
            if (!strings_ok) { set_error("ph_plan: VARCHAR column %zu of a sharded table cannot be hash-partitioned across ranks", c); return PH_EUNSUPPORTED; }
            string_cols->push_back((int)c);
            continue;
        }
        need.push_back((int)c);
    }
    PL_CHECK(positional(p, r, need));
    for (size_t L = 0; L < r->lanes.size(); L++) {
        bool used = false;
        for (auto &pc : r->cols) used |= pc.lane == (int)L;
        if (!used) continue;
        const Lane &ln = r->lanes[L];
        if (ln.nullable) { set_error("ph_plan: the NULL-able side of a LEFT join cannot cross ranks"); return PH_EUNSUPPORTED; }
        if (!ln.t->replicated) continue;   // (a sharded table's lane only carries VARCHAR columns now: replicate_rel packs their strings itself)
        const void *ids = ln.rows;
        if (!ids) {   // identity: materialise 0..n-1 (a sequence through the expression evaluator would do; a gather of nothing is simpler: ph_sel_difference)
            void *seq = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &seq));
            int64_t m = 0;
            if (r->n > 0) PL_CHECK(ph_sel_difference(p->ctx, nullptr, r->n, nullptr, 0, r->n, (int32_t *)seq, &m));
            ids = seq;
        }
        tv->lane_of[L] = (int)tv->data.size();
        tv->data.push_back(ids);
        tv->width.push_back(4);
    }
    for (int c : need) {
        const PCol &pc = r->cols[(size_t)c];
        if (pc.validity) { set_error("ph_plan: a NULL-able column cannot cross ranks"); return PH_EUNSUPPORTED; }
        tv->col_of[(size_t)c] = (int)tv->data.size();
        tv->data.push_back(pc.data);
        tv->width.push_back(width_of(pc.type));
    }
    return PH_OK;
}

// the relation after its travel columns arrived as `recv` (n rows): lanes over the replicated tables, positional values otherwise
void rebuild_after_travel(const Rel &src, const Travel &tv, const std::vector<void *> &recv, int64_t n, Rel *out) {
    *out = Rel{};
    out->n = n;
    out->covers = false;
    std::vector<int> lane_map(src.lanes.size(), -1);
    for (size_t L = 0; L < src.lanes.size(); L++) {
        if (tv.lane_of[L] < 0) continue;
        Lane ln;
        ln.t = src.lanes[L].t; ln.rows = (const int32_t *)recv[(size_t)tv.lane_of[L]]; ln.asc = false; ln.dup_free = false;
        lane_map[L] = (int)out->lanes.size();
        out->lanes.push_back(ln);
    }
    for (size_t c = 0; c < src.cols.size(); c++) {
        PCol pc = src.cols[c];
        pc.ordered = false; pc.domain = -1;
        if (tv.col_of[c] >= 0) { pc.lane = -1; pc.tcol = -1; pc.data = recv[(size_t)tv.col_of[c]]; pc.validity = nullptr; pc.src = nullptr; pc.src_col = -1; }
        else pc.lane = lane_map[(size_t)pc.lane];
        out->cols.push_back(pc);
    }
}

// ---- hash-partition a relation by column `keycol` and exchange it all-to-all: afterwards every row sits on rank mix64(key) mod N
int repartition_rel(ph_plan *p, Rel *r, int keycol, const char *what) {
    ph_ctx *ctx = p->ctx;
    const int n = ph_comm_nranks(p->comm);
    // the key column's range over all ranks, before its provenance goes
    bool gok = false;
    int64_t gmin = 0, gmax = 0;
    PL_CHECK(global_range(p, r->cols[(size_t)keycol], &gok, &gmin, &gmax));
    Travel tv;
    std::vector<int> strs;
    PL_CHECK(prepare_travel(p, r, &tv, false, &strs));
    // the key's values: positional already (prepare_travel), or behind a replicated table's lane
    const int32_t *ks = nullptr;
    ph_col kv = col_view(*r, r->cols[(size_t)keycol], &ks);
    if (ks || width_of(kv.type) < 4) {   // (a key behind a replicated lane's row ids, or a 1-byte key: gathered / widened first)
        if (kv.type == PH_CODE8) {
            void *w32 = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &w32));
            if (r->n > 0) PL_CHECK(ph_widen_codes(ctx, &kv, ks, r->n, (int32_t *)w32));
            kv = ph_col{}; kv.type = PH_I32; kv.data = w32;
        } else {
            void *g = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * width_of(kv.type), &g));
            if (r->n > 0) PL_CHECK(ph_gather(ctx, &kv, ks, r->n, g));
            kv.data = g;
        }
    }
    if (kv.validity) { set_error("ph_plan: a NULL-able partition key"); return PH_EUNSUPPORTED; }
    void *counts = nullptr, *perm = nullptr;
    PL_CHECK(palloc(p, (int64_t)n * 8, &counts));
    PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * 4, &perm));
    PL_CHECK(ph_partition_dev(ctx, &kv, nullptr, r->n, n, (int64_t *)counts, (int32_t *)perm));
    std::vector<int64_t> matrix((size_t)n * n), so((size_t)n + 1), ro((size_t)n + 1);
    PL_CHECK(ph_comm_exchange_counts(p->comm, (const int64_t *)counts, matrix.data()));
    PL_CHECK(ph_exchange_layout(matrix.data(), n, ph_comm_rank(p->comm), so.data(), ro.data()));
    const int64_t nrecv = ro[(size_t)n];
    std::vector<const void *> send(tv.data.size());
    std::vector<void *> recv(tv.data.size());
    for (size_t k = 0; k < tv.data.size(); k++) {
        ph_col v{};
        v.type = tv.width[k] == 8 ? PH_I64 : tv.width[k] == 4 ? PH_I32 : PH_CODE8; v.data = tv.data[k];
        void *g = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(r->n, 1) * tv.width[k], &g));
        if (r->n > 0) PL_CHECK(ph_gather(ctx, &v, (const int32_t *)perm, r->n, g));
        send[k] = g;
        PL_CHECK(palloc(p, std::max<int64_t>(nrecv, 1) * tv.width[k], &recv[k]));
    }
    PL_CHECK(ph_comm_exchange_columns(p->comm, (int32_t)tv.data.size(), send.data(), recv.data(), tv.width.data(), matrix.data()));
    const int64_t sent = r->n;
    Rel out;
    rebuild_after_travel(*r, tv, recv, nrecv, &out);
    if (gok && out.cols[(size_t)keycol].lane < 0) { out.cols[(size_t)keycol].grange = true; out.cols[(size_t)keycol].gmin = gmin; out.cols[(size_t)keycol].gmax = gmax; }
    out.replicated = false;
    *r = out;
    note(p, "  exchange (%s): %lld rows hash-partitioned by column %d over %d ranks, %lld received, %zu columns travel", what, (long long)sent, keycol, n,
         (long long)nrecv, tv.data.size());
    return PH_OK;
}

// ---- make a relation the same on every rank: all-gather its columns into a temporary table the plan owns (VARCHAR columns included: lengths
// and bytes travel, the offsets are rebuilt); the result is a plain single-table relation again, with the table forms its statistics allow
int replicate_rel(ph_plan *p, Rel *r, const char *what) {
    ph_ctx *ctx = p->ctx;
    const int n = ph_comm_nranks(p->comm);
    std::vector<bool> gok(r->cols.size(), false);
    std::vector<int64_t> gmin(r->cols.size(), 0), gmax(r->cols.size(), 0);
    for (size_t c = 0; c < r->cols.size(); c++) {
        const PCol &pc = r->cols[c];
        if (pc.type == PH_I32 || pc.type == PH_I64) { bool ok = false; PL_CHECK(global_range(p, pc, &ok, &gmin[c], &gmax[c])); gok[c] = ok; }
    }
    // declared-unique column sets of the one source table survive (a primary key is unique over all shards)
    const ph_table *src_t = nullptr;
    bool one_src = true;
    for (auto &pc : r->cols) { if (!pc.src) { one_src = false; break; } if (!src_t) src_t = pc.src; else if (src_t != pc.src) { one_src = false; break; } }
    one_src = one_src && r->lanes.size() == 1 && r->lanes[0].dup_free;
    std::vector<int> src_cols;
    for (auto &pc : r->cols) src_cols.push_back(pc.src_col);
    Travel tv;
    std::vector<int> strs;
    PL_CHECK(prepare_travel(p, r, &tv, true, &strs));
    std::vector<int64_t> counts((size_t)n);
    ph_table *vt = new ph_table();
    p->computed.push_back(vt);
    vt->ctx = ctx;
    vt->replicated = true;
    vt->cols.resize(r->cols.size());
    int64_t total = -1;
    std::vector<void *> recv(tv.data.size(), nullptr);
    for (size_t k = 0; k < tv.data.size(); k++) {
        void *out = nullptr;
        PL_CHECK(ph_comm_allgather_rows_alloc(p->comm, tv.data[k], r->n, tv.width[k], &out, counts.data()));
        p->computed_bufs.push_back(out);
        recv[k] = out;
    }
    {
        int64_t t2 = 0;
        PL_CHECK(global_sum(p, r->n, &t2));
        total = t2;
    }
    // VARCHAR columns of sharded tables: the selected rows' strings packed on the device (ph_substring over the row ids), bytes and offsets gathered,
    // the offsets re-based on the host (a broadcast side is small by definition)
    std::vector<void *> str_off(r->cols.size(), nullptr), str_bytes(r->cols.size(), nullptr);
    std::vector<int64_t> str_nbytes(r->cols.size(), 0);
    for (int c : strs) {
        const PCol &pc = r->cols[(size_t)c];
        if (pc.lane < 0) { set_error("ph_plan: VARCHAR column %d is not a table column", c); return PH_EUNSUPPORTED; }
        const Lane &ln = r->lanes[(size_t)pc.lane];
        ph_col v = table_view(ln.t, pc.tcol);
        if (v.validity) { set_error("ph_plan: a NULL-able VARCHAR column cannot cross ranks"); return PH_EUNSUPPORTED; }
        void *off = nullptr, *bytes = nullptr;
        int64_t cap = v.aux_bytes + 64, nb = 0;
        PL_CHECK(palloc(p, (r->n + 1) * 4 + 64, &off));
        for (int attempt = 0; attempt < 2; attempt++) {
            PL_CHECK(palloc(p, cap, &bytes));
            int rc = r->n > 0 ? ph_substring(ctx, &v, 1, INT64_MAX, ln.rows, r->n, (int32_t *)off, (uint8_t *)bytes, cap, &nb) : PH_OK;
            if (rc == PH_ECAPACITY && attempt == 0 && nb > cap) { cap = nb + 64; continue; }
            PL_CHECK(rc);
            break;
        }
        if (r->n == 0) PL_CHECK(ph_dev_memset(ctx, off, 0, 8));
        void *all_off = nullptr, *all_bytes = nullptr;
        std::vector<int64_t> oc((size_t)n), bc((size_t)n);
        PL_CHECK(ph_comm_allgather_rows_alloc(p->comm, off, r->n + 1, 4, &all_off, oc.data()));       // every rank's n_r + 1 local offsets
        PL_CHECK(ph_comm_allgather_rows_alloc(p->comm, bytes, nb, 1, &all_bytes, bc.data()));
        p->computed_bufs.push_back(all_bytes);
        int64_t tot_off = 0, tot_bytes = 0;
        for (int q = 0; q < n; q++) { tot_off += oc[(size_t)q]; tot_bytes += bc[(size_t)q]; }
        if (tot_bytes >= (1ll << 31)) { ph_dev_free(ctx, all_off); set_error("ph_plan: %lld bytes of strings in a broadcast side", (long long)tot_bytes); return PH_EUNSUPPORTED; }
        std::vector<int32_t> lo((size_t)tot_off), fixed((size_t)total + 1);
        int rc = tot_off > 0 ? ctx->download(lo.data(), all_off, tot_off * 4) : PH_OK;
        ph_dev_free(ctx, all_off);
        PL_CHECK(rc);
        int64_t at = 0, pos = 0, base = 0;
        for (int q = 0; q < n; q++) {
            const int64_t rows = oc[(size_t)q] - 1;
            for (int64_t i = 0; i < rows; i++) fixed[(size_t)at++] = (int32_t)(base + lo[(size_t)(pos + i)]);
            pos += oc[(size_t)q];
            base += bc[(size_t)q];
        }
        fixed[(size_t)at] = (int32_t)base;
        void *fo = nullptr;
        PL_CHECK(ctx->pool_alloc((total + 1) * 4 + 64, &fo));
        p->computed_bufs.push_back(fo);
        PL_CHECK(ph_dev_upload(ctx, fo, fixed.data(), (total + 1) * 4));
        str_off[(size_t)c] = fo; str_bytes[(size_t)c] = all_bytes; str_nbytes[(size_t)c] = tot_bytes;
    }
    vt->nrows = total;
    // columns of the temporary table: gathered values, gathered strings, or — for columns that rode on a replicated table's lane — nothing (they stay lane columns)
    Rel out;
    out.n = total;
    out.covers = false;
    out.replicated = true;
    Lane tl;
    tl.t = vt;
    out.lanes.push_back(tl);
    std::vector<int> lane_map(r->lanes.size(), -1);
    for (size_t L = 0; L < r->lanes.size(); L++) {
        if (tv.lane_of[L] < 0 || !r->lanes[L].t->replicated) continue;
        Lane ln;
        ln.t = r->lanes[L].t; ln.rows = (const int32_t *)recv[(size_t)tv.lane_of[L]]; ln.asc = false; ln.dup_free = false;
        lane_map[L] = (int)out.lanes.size();
        out.lanes.push_back(ln);
    }
    for (size_t c = 0; c < r->cols.size(); c++) {
        PCol pc = r->cols[c];
        pc.ordered = false; pc.domain = -1;
        auto &tc = vt->cols[c];
        tc.type = pc.type; tc.scale = pc.scale;
        if (tv.col_of[c] >= 0) {
            tc.data = recv[(size_t)tv.col_of[c]];
            if (gok[c]) { tc.has_range = true; tc.min = gmin[c]; tc.max = gmax[c]; }
            if (pc.src && pc.src_col >= 0 && pc.type == PH_CODE8) tc.dict = pc.src->cols[(size_t)pc.src_col].dict;
            pc.lane = 0; pc.tcol = (int)c; pc.data = nullptr; pc.validity = nullptr; pc.src = vt; pc.src_col = (int)c;
        } else if (str_off[c]) {
            tc.data = str_off[c]; tc.aux = str_bytes[c]; tc.aux_bytes = str_nbytes[c];
            pc.lane = 0; pc.tcol = (int)c; pc.data = nullptr; pc.validity = nullptr; pc.src = vt; pc.src_col = (int)c;
        } else {
            tc.type = 0;   // (unused slot: the column lives behind a replicated table's lane)
            pc.lane = lane_map[(size_t)pc.lane];
        }
        out.cols.push_back(pc);
    }
    if (one_src && src_t)
        for (auto &u : src_t->unique_keys) {
            std::vector<int32_t> mapped;
            for (int32_t sc : u) {
                int at = -1;
                for (size_t c = 0; c < src_cols.size(); c++) if (src_cols[c] == sc && tv.col_of[c] >= 0) { at = (int)c; break; }
                if (at < 0) { mapped.clear(); break; }
                mapped.push_back(at);
            }
            if (!mapped.empty()) { std::sort(mapped.begin(), mapped.end()); vt->unique_keys.push_back(mapped); }
        }
    note(p, "  broadcast (%s): %lld rows of this rank, %lld on every rank afterwards (a replicated temporary table, %zu gathered columns%s)", what,
         (long long)r->n, (long long)total, tv.data.size(), strs.empty() ? "" : " + VARCHAR");
    *r = out;
    return PH_OK;
}

// ---- where a join's sides must meet: decided from all-reduced facts, identically on every rank
int distribute_join(ph_plan *p, int idx, const Node &nd, Rel *P, Rel *B) {
    if (B->replicated) return PH_OK;                                              // local; the result is as distributed as P
    if (P->replicated && nd.join_type == PH_JT_INNER) { note(p, "join#%d: replicated probe side x sharded build side: local (every pair is found on the build row's rank)", idx); return PH_OK; }
    if (!P->replicated) {
        bool co = false;
        PL_CHECK(colocated(p, P->cols[(size_t)nd.pkeys[0]], B->cols[(size_t)nd.bkeys[0]], &co));
        if (co) { note(p, "join#%d: co-located by key range (the ranks' statistics): no exchange", idx); return PH_OK; }
    }
    PL_CHECK(apply_pending(p, B));
    int64_t total = 0;
    PL_CHECK(global_sum(p, B->n, &total));
    if (total <= p->bcast_rows || P->replicated) {
        char what[64];
        snprintf(what, sizeof what, "build side of join#%d", idx);
        return replicate_rel(p, B, what);
    }
    char wb[64], wp[64];
    snprintf(wb, sizeof wb, "build side of join#%d", idx);
    snprintf(wp, sizeof wp, "probe side of join#%d", idx);
    PL_CHECK(repartition_rel(p, B, nd.bkeys[0], wb));
    PL_CHECK(repartition_rel(p, P, nd.pkeys[0], wp));
    return PH_OK;
}

int join_rels(ph_plan *p, int idx, const Node &nd, Rel P, Rel B, bool as_build, Rel *out);
int left_join_rels(ph_plan *p, int idx, const Node &nd, Rel P, Rel B, Rel *out);

// ---- HashJoin
int lower_join(ph_plan *p, int idx, bool as_build, Rel *out) {
    Rel P, B;
    PL_CHECK(lower(p, p->nodes[(size_t)idx].child[0], false, &P));
    // ---- the probe side's keys go DOWN into a build-side aggregate grouped by the join key (Q17: avg(l_quantity) by l_partkey over sixty million
    // rows, asked for by the 2 000 parts of one brand; Q20, Q2 alike): a group no probe row matches never reaches the result of an INNER / SEMI
    // join, so its input rows need not be aggregated — the aggregate's input is semi-joined with the probe keys first (lower_node, PH_PN_AGG).
    // Through Filters over the groups (a HAVING), not through anything that has another parent.
    const Node &nd = p->nodes[(size_t)idx];
    int target = -1;
    if ((nd.join_type == PH_JT_INNER || nd.join_type == PH_JT_SEMI) && !multi(p) && !getenv("PH_PLAN_NO_KEY_PUSHDOWN")) {
        int c = nd.child[1];
        while (c >= 0 && c < (int)p->nodes.size() && p->nodes[(size_t)c].kind == PH_PN_FILTER && p->parents[(size_t)c] < 2) c = p->nodes[(size_t)c].child[0];
        if (c >= 0 && c < (int)p->nodes.size() && p->nodes[(size_t)c].kind == PH_PN_AGG && p->parents[(size_t)c] < 2) {
            const Node &ag = p->nodes[(size_t)c];
            for (size_t k = 0; k < nd.bkeys.size() && target < 0; k++) {
                const int32_t b = nd.bkeys[k];
                if (b < 0 || b >= (int32_t)ag.groups.size() || ag.groups[(size_t)b].e.kind != PH_PE_COL) continue;
                if (nd.pkeys[k] < 0 || (size_t)nd.pkeys[k] >= P.cols.size() || P.cols[(size_t)nd.pkeys[k]].type == PH_STR) continue;
                ph_plan::PushedKeys pk;
                pk.probe = std::make_shared<Rel>(P);
                pk.pkey = nd.pkeys[k];
                pk.gcol = ag.groups[(size_t)b].e.col;
                pk.join = idx;
                p->pushed[c] = pk;
                target = c;
            }
        }
    }
    const int rc = lower(p, nd.child[1], true, &B);
    if (target >= 0) p->pushed.erase(target);
    PL_CHECK(rc);
    return join_rels(p, idx, nd, P, B, as_build, out);
}

int join_rels_local(ph_plan *p, int idx, const Node &nd, Rel P, Rel B, bool as_build, Rel *out);

// one join: across ranks first the question where its sides meet (distribute_join), then the single-rank lowering
int join_rels(ph_plan *p, int idx, const Node &nd, Rel P, Rel B, bool as_build, Rel *out) {
    if (!multi(p)) return join_rels_local(p, idx, nd, P, B, as_build, out);
    for (size_t k = 0; k < nd.pkeys.size(); k++)
        if (nd.pkeys[k] < 0 || (size_t)nd.pkeys[k] >= P.cols.size() || nd.bkeys[k] < 0 || (size_t)nd.bkeys[k] >= B.cols.size()) { set_error("ph_plan: join key out of range"); return PH_EINVAL; }
    PL_CHECK(distribute_join(p, idx, nd, &P, &B));
    const bool result_replicated = P.replicated && B.replicated;
    if (!B.replicated) B.covers = false;   // a rank's share of the build side: "every probe row finds its row" is not this rank's to claim
    for (auto &c : P.cols) c.domain = -1;  // (sideways information passing reasons about ONE rank's rows: off across ranks)
    PL_CHECK(join_rels_local(p, idx, nd, P, B, as_build, out));
    out->replicated = result_replicated;
    return PH_OK;
}

int join_rels_local(ph_plan *p, int idx, const Node &nd, Rel P, Rel B, bool as_build, Rel *out) {
    ph_ctx *ctx = p->ctx;
    const size_t nP = P.cols.size(), nB = B.cols.size();
    const size_t nk = nd.pkeys.size();
    // VARCHAR key pairs: the build side's strings are interned, the probe side's looked up in that dictionary; the join
    // then runs on the int32 codes. The code columns REPLACE the string columns in the working copies of the key lists
    // (the relations' own columns, which the output picks from, stay as they are).
    Node ndx = nd;
    bool has_str = false;
    for (size_t k = 0; k < nk; k++) {
        if (nd.pkeys[k] < 0 || (size_t)nd.pkeys[k] >= nP || nd.bkeys[k] < 0 || (size_t)nd.bkeys[k] >= nB) { set_error("ph_plan: join key out of range"); return PH_EINVAL; }
        if (P.cols[(size_t)nd.pkeys[k]].type == PH_STR && B.cols[(size_t)nd.bkeys[k]].type == PH_STR) has_str = true;
    }
    if (has_str) {
        for (size_t k = 0; k < nk; k++) {
            if (P.cols[(size_t)nd.pkeys[k]].type != PH_STR) continue;
            ph_strdict *d = nullptr;
            PCol bc, pc;
            PL_CHECK(string_codes(p, &B, nd.bkeys[k], nullptr, &d, &bc));
            PL_CHECK(string_codes(p, &P, nd.pkeys[k], d, nullptr, &pc));
            bc.src = nullptr; pc.src = nullptr;   // codes of two different columns: no shared provenance, no statistics
            B.cols.push_back(bc); ndx.bkeys[k] = (int32_t)B.cols.size() - 1;
            P.cols.push_back(pc); ndx.pkeys[k] = (int32_t)P.cols.size() - 1;
        }
        // the output indexes address [P's original columns | B's original columns]: re-base the build half behind P's new width
        for (auto &o : ndx.out) if ((size_t)o >= nP) o = (int32_t)((size_t)o - nP + P.cols.size());
        return join_rels_local(p, idx, ndx, P, B, as_build, out);
    }
    for (size_t k = 0; k < nk; k++) {
        const int a = P.cols[(size_t)nd.pkeys[k]].type, b = B.cols[(size_t)nd.bkeys[k]].type;
        if (width_of(a) == 0 || width_of(a) != width_of(b)) { set_error("ph_plan: join key %zu types differ or are VARCHAR (%d / %d)", k, a, b); return PH_EUNSUPPORTED; }
    }
    // ---- a residual condition (the join's non-equi conjuncts over [probe | build] columns: `l2.l_suppkey <> l1.l_suppkey` inside Q21's EXISTS):
    // the equi-join's PAIRS carry the columns it reads and the probe row's position; the condition filters the pairs; an INNER join keeps those,
    // a SEMI / ANTI join marks the probe rows that kept a pair (one scattered byte per surviving pair) and selects the marked / unmarked rows.
    // ... and the same path WITHOUT a condition for a SEMI / ANTI join whose build side is a big table clustered by the key and whose probe side is a
    // sliver of it (Q4: 570 k orders of one quarter against the 38 M late lines): the pairs of the table-less join (a binary search per probe row,
    // the build table's own filters applied to the pairs) mark their probe rows — instead of an existence table over every qualifying build row
    bool exists_by_pairs = false;
    if (nd.bools.empty() && (nd.join_type == PH_JT_SEMI || nd.join_type == PH_JT_ANTI) && nk == 1 && B.single_identity() && !B.flags && !p->conservative &&
        !getenv("PH_PLAN_NO_SORTED_PAIRS") && !getenv("PH_PLAN_NO_EXISTS_PAIRS")) {
        const PCol &bc = B.cols[(size_t)nd.bkeys[0]];
        const ph_table *bt = B.lanes[0].t;
        bool all_lane0 = bc.lane == 0 && bc.tcol >= 0;
        for (auto &c : B.cols) all_lane0 = all_lane0 && c.lane == 0;
        if (all_lane0 && bt->cols[(size_t)bc.tcol].ascending && !bt->cols[(size_t)bc.tcol].strict && bt->nrows >= (1 << 22)) {
            PL_CHECK(apply_pending(p, &P));
            exists_by_pairs = P.n * 32 <= bt->nrows;
        }
    }
    if (!nd.bools.empty() || exists_by_pairs) {
        if (nd.join_type == PH_JT_LEFT) { set_error("ph_plan: a LEFT join with a residual condition"); return PH_EUNSUPPORTED; }
        PL_CHECK(apply_pending(p, &P));
        const bool exists = nd.join_type == PH_JT_SEMI || nd.join_type == PH_JT_ANTI;
        Rel P2 = P;
        if (exists) {   // the probe row's position rides along as one more column
            void *pos = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &pos));
            if (P.n > 0) PL_CHECK(ph_dev_iota(ctx, (int32_t *)pos, P.n));
            PCol c; c.type = PH_I32; c.data = pos;
            P2.cols.push_back(c);
        }
        const size_t nP2 = P2.cols.size();
        Node in = nd;
        in.join_type = PH_JT_INNER;
        in.bools = BoolTree{};
        in.out.clear();
        std::vector<int> where((size_t)(nP + nB), -1);   // original [P | B] column -> its position among the pairs' columns
        auto carried = [&](int oc) -> int {
            if (oc < 0 || (size_t)oc >= nP + nB) return -1;
            if (where[(size_t)oc] < 0) { in.out.push_back((size_t)oc < nP ? oc : (int32_t)((size_t)oc - nP + nP2)); where[(size_t)oc] = (int)in.out.size() - 1; }
            return where[(size_t)oc];
        };
        if (!exists) for (int32_t o : nd.out) { if (carried(o) < 0) { set_error("ph_plan: join output column out of range"); return PH_EINVAL; } }
        BoolTree tree = nd.bools;
        for (auto &b : tree.nodes) {
            if (b.kind != PH_B_CMP) continue;
            b.col = carried(b.col);
            if (b.k.type == PH_COLREF) b.k.i = carried((int)b.k.i);
            if (b.col < 0 || (b.k.type == PH_COLREF && b.k.i < 0)) { set_error("ph_plan: residual condition column out of range"); return PH_EINVAL; }
        }
        tree.fix();
        int poscol = -1;
        if (exists) { in.out.push_back((int32_t)nP); poscol = (int)in.out.size() - 1; }   // (nP = the position column's index in P2)
        Rel J;
        const bool was = p->no_sideways;
        if (exists_by_pairs) p->no_sideways = true;   // (the clustered build table is searched, not reduced)
        const int jrc = join_rels_local(p, idx, in, P2, B, false, &J);
        p->no_sideways = was;
        PL_CHECK(jrc);
        PL_CHECK(apply_pending(p, &J));
        const int64_t pairs = J.n;
        const int32_t *keep = nullptr;
        int64_t nkeep = J.n;   // (no condition: every pair counts)
        if (J.n > 0 && !tree.empty()) PL_CHECK(eval_bool(p, &J, false, tree, 0, nullptr, J.n, &keep, &nkeep));
        if (!exists) {
            if (nkeep == 0) J.n = 0; else PL_CHECK(compact(p, &J, keep, nkeep));
            // the pairs' columns in the order of nd.out
            std::vector<PCol> cols;
            for (int32_t o : nd.out) cols.push_back(J.cols[(size_t)where[(size_t)o]]);
            J.cols = cols;
            J.covers = false;
            drop_unused_lanes(&J);
            note(p, "join#%d: residual condition over the pairs: %lld of %lld kept", idx, (long long)nkeep, (long long)pairs);
            *out = J;
            return PH_OK;
        }
        // marks: a byte per probe row, set by every surviving pair
        void *marks = nullptr, *sel = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) + 64, &marks));
        PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &sel));
        PL_CHECK(ph_dev_memset(ctx, marks, 0, std::max<int64_t>(P.n, 1)));
        if (nkeep > 0) {
            PL_CHECK(positional(p, &J, {poscol}));
            const int32_t *s0 = nullptr;
            ph_col pv = col_view(J, J.cols[(size_t)poscol], &s0);
            const void *kept = pv.data;
            if (keep) {
                void *g = nullptr;
                PL_CHECK(palloc(p, nkeep * 4, &g));
                PL_CHECK(ph_gather(ctx, &pv, keep, nkeep, g));
                kept = g;
            }
            PL_CHECK(ph_sel_mark(ctx, (const int32_t *)kept, nkeep, (uint8_t *)marks));
        }
        int64_t m = 0;
        if (P.n > 0) {
            ph_col fc{};
            fc.type = PH_CODE8; fc.data = marks;
            ph_const want{};
            want.type = PH_I32; want.i = nd.join_type == PH_JT_ANTI ? 0 : 1;
            PL_CHECK(ph_filter_select(ctx, &fc, P.n, PH_EQ, &want, nullptr, P.n, (int32_t *)sel, &m));
        }
        *out = P;
        PL_CHECK(compact(p, out, (const int32_t *)sel, m));
        std::vector<PCol> cols;
        for (int32_t o : nd.out) {
            if (o < 0 || (size_t)o >= nP) { set_error("ph_plan: a SEMI / ANTI join emits probe columns only"); return PH_EINVAL; }
            cols.push_back(out->cols[(size_t)o]);
        }
        out->cols = cols;
        out->covers = false;
        drop_unused_lanes(out);
        if (tree.empty()) note(p, "join#%d: %s through the pairs of the table-less join: %lld pairs mark %lld of %lld probe rows", idx, nd.join_type == PH_JT_ANTI ? "ANTI" : "SEMI",
                               (long long)pairs, (long long)(nd.join_type == PH_JT_ANTI ? P.n - m : m), (long long)P.n);
        else note(p, "join#%d: %s with a residual condition: %lld of %lld pairs kept it, %lld of %lld probe rows %s", idx, nd.join_type == PH_JT_ANTI ? "ANTI" : "SEMI",
                  (long long)nkeep, (long long)pairs, (long long)m, (long long)P.n, nd.join_type == PH_JT_ANTI ? "have none" : "have one");
        return PH_OK;
    }
    bool need_build_cols = false;
    for (int32_t o : nd.out) {
        if (o < 0 || (size_t)o >= nP + nB) { set_error("ph_plan: join output column out of range"); return PH_EINVAL; }
        need_build_cols |= (size_t)o >= nP;
    }
    if (nd.join_type != PH_JT_INNER && nd.join_type != PH_JT_LEFT && need_build_cols) { set_error("ph_plan: a SEMI / ANTI join emits probe columns only"); return PH_EINVAL; }
    if (nd.join_type == PH_JT_LEFT) return left_join_rels(p, idx, nd, P, B, out);
    // ---- a SELECTIVE N:1 join first, the probe scan's column-vs-column / OR conjuncts after it. Such a conjunct costs two passes over the table and a
    // selection vector of what it keeps (Q21: l_receiptdate > l_commitdate keeps 38 of 60 M rows, 400 us); a join that keeps a twentieth of the rows
    // needs one pass over the key column, and the conjunct then reads the survivors only. The estimate: the build side's rows against the width of the
    // probe key's value range (the table's load-time statistics). The conjuncts' columns ride through the join as extra output columns of this call.
    if (nd.join_type == PH_JT_INNER && nk == 1 && P.single_identity() && !P.complex.empty() && !P.flags && !getenv("PH_PLAN_NO_LATE_FILTER")) {
        const ph_table *pt = P.lanes[0].t;
        const PCol &kc = P.cols[(size_t)nd.pkeys[0]];
        bool pays = pt->nrows >= (1 << 22) && kc.lane == 0 && kc.tcol >= 0 && pt->cols[(size_t)kc.tcol].has_range && key_unique(B, nd.bkeys) > 0;
        // (a filtered SMALL build table knows its rows only as an upper bound: its selection now — a pass over a table a hundredth of the probe side)
        if (pays && B.lazy() && B.single_identity() && B.lanes[0].t->nrows * 64 <= pt->nrows) PL_CHECK(apply_pending(p, &B));
        if (pays) {
            const long double range = (long double)pt->cols[(size_t)kc.tcol].max - (long double)pt->cols[(size_t)kc.tcol].min + 1.0L;
            pays = (long double)B.n * 8.0L <= range;
        }
        if (pays) {
            Rel P2 = P;
            P2.complex.clear();
            Node nd2 = nd;
            for (auto &o : nd2.out) if ((size_t)o >= nP) o = -1 - (int32_t)((size_t)o - nP);   // build columns: marked, re-based below
            std::vector<BoolTree> trees = P.complex;
            auto carried = [&](int tc) -> int {   // table column tc of the probe table as an output column of the join; its position there
                int pc = -1;
                for (size_t i = 0; i < P2.cols.size() && pc < 0; i++) if (P2.cols[i].lane == 0 && P2.cols[i].tcol == tc) pc = (int)i;
                if (pc < 0) {
                    PCol c; c.type = pt->cols[(size_t)tc].type; c.scale = pt->cols[(size_t)tc].scale; c.lane = 0; c.tcol = tc; c.src = pt; c.src_col = tc;
                    P2.cols.push_back(c);
                    pc = (int)P2.cols.size() - 1;
                }
                for (size_t i = 0; i < nd2.out.size(); i++) if (nd2.out[i] == pc) return (int)i;
                nd2.out.push_back(pc);
                return (int)nd2.out.size() - 1;
            };
            bool ok = true;
            for (auto &t : trees) {
                for (auto &b : t.nodes) {
                    if (b.kind != PH_B_CMP) continue;
                    if (b.col < 0 || b.col >= (int)pt->cols.size() || (b.k.type == PH_COLREF && (b.k.i < 0 || b.k.i >= (int64_t)pt->cols.size()))) { ok = false; break; }
                    b.col = carried(b.col);
                    if (b.k.type == PH_COLREF) b.k.i = carried((int)b.k.i);
                }
                t.fix();
            }
            if (ok) {
                for (auto &o : nd2.out) if (o < 0) o = (int32_t)((size_t)(-1 - o) + P2.cols.size());
                Rel J;
                PL_CHECK(join_rels_local(p, idx, nd2, P2, B, as_build, &J));
                PL_CHECK(apply_pending(p, &J));   // (an existence-only form leaves marks: rows now)
                const int64_t joined = J.n;
                const int32_t *selp = nullptr;
                int64_t cnt = J.n;
                for (size_t i = 0; i < trees.size() && cnt > 0; i++) {
                    const int32_t *o = nullptr;
                    int64_t m2 = 0;
                    PL_CHECK(eval_bool(p, &J, false, trees[i], 0, selp, selp ? cnt : J.n, &o, &m2));
                    selp = o; cnt = m2;
                }
                if (cnt == 0) { J.n = 0; }
                else if (selp) PL_CHECK(compact(p, &J, selp, cnt));
                J.cols.resize(nd.out.size());
                J.covers = false;
                drop_unused_lanes(&J);
                note(p, "join#%d: the probe scan's %zu column-vs-column / OR conjuncts evaluated BEHIND the join: %lld of its %lld rows kept", idx, trees.size(), (long long)cnt,
                     (long long)joined);
                *out = J;
                return PH_OK;
            }
        }
    }
    const bool optimistic = !p->conservative;
    const int uniq = key_unique(B, nd.bkeys);
    const bool unique = uniq > 0;
    const bool exists_only = nd.join_type == PH_JT_SEMI || nd.join_type == PH_JT_ANTI || (nd.join_type == PH_JT_INNER && !need_build_cols && unique);
    const bool marks_only = nd.join_type == PH_JT_ANTI || (exists_only && !unique);   // probe form (2) below: ph_join_probe_mark and nothing else
    std::string how;

    // ---- run lookup: the build table is stored in runs of one length by the first key column and (first, second) is its unique key — the row is
    // found by arithmetic and a look at the run's second keys (ph_join_run_lookup): no table, no reduction of the build side, one short read per
    // probe row. The run structure is the library's own load-time measurement (ph_table_col_run_len).
    if (nd.join_type == PH_JT_INNER && unique && !exists_only && uniq == 2 && optimistic && nk == 2 && B.covers && B.single_identity() && !B.lazy() &&
        !getenv("PH_PLAN_NO_RUN_LOOKUP")) {
        const ph_table *bt = B.lanes[0].t;
        int first = -1;
        for (int k = 0; k < 2; k++) {
            const PCol &bc = B.cols[(size_t)nd.bkeys[(size_t)k]];
            if (bc.lane == 0 && bc.tcol >= 0 && bt->cols[(size_t)bc.tcol].run_len > 1) first = k;
        }
        const PCol &b2 = B.cols[(size_t)nd.bkeys[(size_t)(first < 0 ? 0 : 1 - first)]];
        if (first >= 0 && b2.lane == 0 && b2.tcol >= 0 && !bt->cols[(size_t)b2.tcol].validity) {
            const PCol &b1 = B.cols[(size_t)nd.bkeys[(size_t)first]];
            PL_CHECK(apply_pending(p, &P));
            KeySide pk;
            PL_CHECK(key_side(p, &P, {nd.pkeys[(size_t)first], nd.pkeys[(size_t)(1 - first)]}, &pk));
            ph_col bv2 = table_view(bt, b2.tcol);
            void *o = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &o));
            int rc = ph_join_run_lookup(ctx, &bv2, bt->nrows, bt->cols[(size_t)b1.tcol].min, bt->cols[(size_t)b1.tcol].run_len, pk.views.data(), pk.sel, P.n, 1, (int32_t *)o);
            if (rc == PH_OK) {
                *out = P;
                Lane bl; bl.t = bt; bl.rows = (const int32_t *)o; bl.asc = false; bl.dup_free = false;
                out->lanes.push_back(bl);
                std::vector<PCol> all = P.cols;
                for (auto c : B.cols) { c.lane = (int)P.lanes.size(); c.ordered = false; c.domain = -1; all.push_back(c); }
                out->cols.clear();
                for (int32_t oi : nd.out) out->cols.push_back(all[(size_t)oi]);
                out->covers = false;
                drop_unused_lanes(out);
                note(p, "join#%d: run lookup (the build table is stored in runs of %d by its first key: row = arithmetic + the run's second keys, no table), %lld probe rows", idx,
                     (int)bt->cols[(size_t)b1.tcol].run_len, (long long)P.n);
                return PH_OK;
            }
            if (rc != PH_EUNSUPPORTED) return rc;
        }
    }

    // ---- sideways information passing: a big, unfiltered build table whose key the probe side has already joined
    // against a small table is reduced to the rows that can match (marks from a probe of that small table)
    if (B.single_identity() && !B.lazy() && B.n >= (1 << 18) && !p->no_sideways) {
        for (size_t k = 0; k < nk; k++) {
            const int d = P.cols[(size_t)nd.pkeys[k]].domain;
            const PCol &bc = B.cols[(size_t)nd.bkeys[k]];
            if (d < 0 || bc.lane != 0 || p->domains[(size_t)d].nkeys * 8 >= B.n) continue;
            ph_col bv = table_view(B.lanes[0].t, bc.tcol);
            void *f = nullptr;
            PL_CHECK(palloc(p, B.n + 64, &f));
            int rc = ph_join_probe_mark(p->domains[(size_t)d].j, &bv, nullptr, B.n, (uint8_t *)f);
            if (rc == PH_EUNSUPPORTED) break;
            PL_CHECK(rc);
            B.flags = (const uint8_t *)f;   // B.covers stays: only rows the probe side cannot reference are gone
            how += " build side reduced by the probe key's domain;";
            break;
        }
    }

    // ---- merge lookup: both sides ordered by the key, no table at all
    const bool n_to_1 = nd.join_type == PH_JT_INNER && unique && !exists_only;
    if (n_to_1 && uniq == 2 && optimistic && nk == 1 && B.covers && B.single_identity() && !B.lazy() && !getenv("PH_PLAN_NO_MERGE")) {
        const PCol &bc = B.cols[(size_t)nd.bkeys[0]];
        const PCol &pc = P.cols[(size_t)nd.pkeys[0]];
        if (bc.lane == 0 && B.lanes[0].t->cols[(size_t)bc.tcol].strict && pc.ordered) {
            PL_CHECK(apply_pending(p, &P));
            const int32_t *psel = nullptr;
            ph_col pv = col_view(P, pc, &psel), bv = table_view(B.lanes[0].t, bc.tcol);
            void *o = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &o));
            int rc = ph_merge_lookup(ctx, &bv, B.n, &pv, psel, P.n, 1, (int32_t *)o);
            if (rc == PH_OK) {
                *out = P;
                Lane bl; bl.t = B.lanes[0].t; bl.rows = (const int32_t *)o; bl.asc = true; bl.dup_free = false;
                out->lanes.push_back(bl);
                std::vector<PCol> all = P.cols;
                for (auto c : B.cols) { c.lane = (int)P.lanes.size(); c.ordered = false; c.domain = -1; all.push_back(c); }
                out->cols.clear();
                for (int32_t oi : nd.out) out->cols.push_back(all[(size_t)oi]);
                out->covers = false;
                note(p, "join#%d: merge lookup (both sides ordered by the key, no table), %lld probe rows", idx, (long long)P.n);
                return PH_OK;
            }
            if (rc != PH_EUNSUPPORTED) return rc;
        }
    }

    // ---- roles swapped: the PROBE side is a big base table clustered by the key and the BUILD side a handful of rows (Q18: sixty million lines against
    // the 624 orders that pass the HAVING) — an INNER join's pairs are the same either way, and with the roles swapped they are the table-less
    // form below (a few hundred binary searches) instead of a pass over the whole key column (163 us at 2.9 TB/s)
    if (nd.join_type == PH_JT_INNER && nk == 1 && P.single_identity() && !P.flags && !getenv("PH_PLAN_NO_SORTED_PAIRS") && !getenv("PH_PLAN_NO_SWAP")) {
        const PCol &pc = P.cols[(size_t)nd.pkeys[0]];
        const ph_table *pt = P.lanes[0].t;
        bool all_lane0 = pc.lane == 0;
        for (auto &c : P.cols) all_lane0 = all_lane0 && c.lane == 0;
        if (all_lane0 && pc.tcol >= 0 && pt->cols[(size_t)pc.tcol].ascending && !pt->cols[(size_t)pc.tcol].strict && pt->nrows >= (1 << 22)) {
            if (B.lazy() && B.single_identity() && B.lanes[0].t->nrows * 64 <= pt->nrows) PL_CHECK(apply_pending(p, &B));
            if (!B.lazy() && B.n * 32 <= pt->nrows) {
                Node sw = nd;
                sw.pkeys = nd.bkeys;
                sw.bkeys = nd.pkeys;
                for (auto &o : sw.out) o = (size_t)o < nP ? (int32_t)(nB + (size_t)o) : (int32_t)((size_t)o - nP);
                note(p, "join#%d: roles swapped (the probe side is a table clustered by the key, the build side %lld rows)", idx, (long long)B.n);
                p->no_sideways = true;
                const int src = join_rels_local(p, idx, sw, B, P, as_build, out);
                p->no_sideways = false;
                return src;
            }
        }
    }

    // ---- no table: the build side is a big base table CLUSTERED by the key and the probe side is a sliver of it — every probe row binary-searches
    // the key column for its run (ph_join_sorted_pairs); what still filters the build table is applied to the PAIRS' build rows afterwards (a few
    // million rows instead of the table's sixty). The order is the library's own load-time measurement (ph_table_col_stats), not a claim of the caller.
    if (nd.join_type == PH_JT_INNER && nk == 1 && B.single_identity() && !B.flags && !getenv("PH_PLAN_NO_SORTED_PAIRS")) {
        const PCol &bc = B.cols[(size_t)nd.bkeys[0]];
        const ph_table *bt = B.lanes[0].t;
        bool all_lane0 = bc.lane == 0;
        for (auto &c : B.cols) all_lane0 = all_lane0 && c.lane == 0;
        if (all_lane0 && bt->cols[(size_t)bc.tcol].ascending && !bt->cols[(size_t)bc.tcol].strict && bt->nrows >= (1 << 22)) {
            PL_CHECK(apply_pending(p, &P));
            if (P.n * 32 <= bt->nrows) {
                KeySide pk;
                PL_CHECK(key_side(p, &P, nd.pkeys, &pk));
                ph_col bv = table_view(bt, bc.tcol);
                int64_t cap = std::max<int64_t>(pk.n * 8, 1024), m = 0;   // (a retry repeats the searches: room for eight rows per probe row up front)
                void *op = nullptr, *ob = nullptr;
                int rc = PH_OK;
                for (int attempt = 0; attempt < 2; attempt++) {
                    PL_CHECK(palloc(p, cap * 4, &op));
                    PL_CHECK(palloc(p, cap * 4, &ob));
                    rc = ph_join_sorted_pairs(ctx, &bv, bt->nrows, &pk.views[0], pk.sel, pk.n, (int32_t *)op, (int32_t *)ob, cap, &m);
                    if (rc == PH_ECAPACITY && attempt == 0 && m > cap) { cap = m; continue; }
                    break;
                }
                if (rc == PH_OK) {
                    const int64_t found = m;
                    // the build table's own filters over the matched build rows: a relation of the pairs' build rows, filtered, gives the surviving pairs
                    const int32_t *keep = nullptr;
                    int64_t nkeep = m;
                    if (B.lazy() && m > 0) {
                        Rel Bm;
                        Bm.n = m;
                        Lane bl; bl.t = bt; bl.rows = (const int32_t *)ob; bl.asc = false; bl.dup_free = false;
                        Bm.lanes.push_back(bl);
                        const int32_t *selp = nullptr;   // positions among the pairs
                        int64_t cnt = m;
                        for (size_t i = 0; i < B.pending.size() && cnt > 0; i++) {
                            ph_pred pr = B.pending[i];
                            fix_dict_const(bt, pr.col, &pr.k);
                            PCol tmp; tmp.type = bt->cols[(size_t)pr.col].type; tmp.scale = bt->cols[(size_t)pr.col].scale; tmp.lane = 0; tmp.tcol = pr.col; tmp.src = bt; tmp.src_col = pr.col;
                            Bm.cols.push_back(tmp);
                            const int ci = (int)Bm.cols.size() - 1;
                            PL_CHECK(positional(p, &Bm, {ci}));
                            const int32_t *s0 = nullptr;
                            ph_col v = col_view(Bm, Bm.cols[(size_t)ci], &s0);
                            fix_num_const(v, &pr.k);
                            void *o = nullptr;
                            PL_CHECK(palloc(p, cnt * 4, &o));
                            int64_t m2 = 0;
                            PL_CHECK(ph_filter_select(ctx, &v, m, pr.op, &pr.k, selp, selp ? cnt : m, (int32_t *)o, &m2));
                            selp = (const int32_t *)o; cnt = m2;
                        }
                        for (size_t i = 0; i < B.complex.size() && cnt > 0; i++) {
                            // the tree's columns are TABLE columns: as output columns of the pairs' relation
                            BoolTree bt2 = B.complex[i];
                            for (auto &b : bt2.nodes) {
                                if (b.kind != PH_B_CMP) continue;
                                auto as_col = [&](int tc) {
                                    PCol tmp; tmp.type = bt->cols[(size_t)tc].type; tmp.scale = bt->cols[(size_t)tc].scale; tmp.lane = 0; tmp.tcol = tc; tmp.src = bt; tmp.src_col = tc;
                                    Bm.cols.push_back(tmp);
                                    return (int)Bm.cols.size() - 1;
                                };
                                b.col = as_col(b.col);
                                if (b.k.type == PH_COLREF) b.k.i = as_col((int)b.k.i);
                            }
                            bt2.fix();
                            const int32_t *o = nullptr;
                            int64_t m2 = 0;
                            PL_CHECK(eval_bool(p, &Bm, false, bt2, 0, selp, selp ? cnt : m, &o, &m2));
                            selp = o; cnt = m2;
                        }
                        keep = selp; nkeep = cnt;
                    }
                    int32_t *prow = (int32_t *)op, *brow = (int32_t *)ob;
                    if (keep) {
                        ph_col pv{}; pv.type = PH_I32; pv.data = op;
                        ph_col bvv{}; bvv.type = PH_I32; bvv.data = ob;
                        void *p2 = nullptr, *b2 = nullptr;
                        PL_CHECK(palloc(p, std::max<int64_t>(nkeep, 1) * 4, &p2));
                        PL_CHECK(palloc(p, std::max<int64_t>(nkeep, 1) * 4, &b2));
                        if (nkeep > 0) { PL_CHECK(ph_gather(ctx, &pv, keep, nkeep, p2)); PL_CHECK(ph_gather(ctx, &bvv, keep, nkeep, b2)); }
                        prow = (int32_t *)p2; brow = (int32_t *)b2; m = nkeep;
                    }
                    // the result: as the general pair probe assembles it (probe rows are ROW IDS of lane 0 when the keys were table columns, positions otherwise)
                    *out = P;
                    if (pk.rowids) {
                        out->lanes[0].rows = prow;
                        out->lanes[0].dup_free = false;
                        out->n = m;
                    } else {
                        // prow holds positions i of the probe relation (ph_join_sorted_pairs reports sel[i] or i: there was no selection)
                        PL_CHECK(compact(p, out, prow, m));
                        for (auto &ln : out->lanes) ln.dup_free = false;
                    }
                    std::vector<PCol> all = out->cols;
                    Lane bl2; bl2.t = bt; bl2.rows = brow; bl2.asc = false; bl2.dup_free = false;
                    out->lanes.push_back(bl2);
                    for (auto c : B.cols) { c.lane = (int)out->lanes.size() - 1; c.ordered = false; c.domain = -1; all.push_back(c); }
                    out->cols.clear();
                    for (int32_t oi : nd.out) out->cols.push_back(all[(size_t)oi]);
                    out->covers = false;
                    out->pending.clear(); out->complex.clear(); out->flags = nullptr;
                    drop_unused_lanes(out);
                    note(p, "join#%d: no table — the build side is clustered by the key: %lld probe rows binary-search the column, %lld pairs%s, %lld kept", idx, (long long)pk.n,
                         (long long)found, B.lazy() ? " (the build side's filters applied to the pairs)" : "", (long long)m);
                    return PH_OK;
                }
                if (rc != PH_EUNSUPPORTED) return rc;
            }
        }
    }

    // ---- build
    ph_join *j = nullptr;
    const uint8_t *residual = nullptr;   // marks of the build table's rows tested by the probe instead of the build
    bool brow_is_rowid = false;          // build row ids reported by probes are row ids of B's lane 0 (else positions in B)
    {
        bool all_lane0 = B.lanes.size() == 1;
        for (int32_t k : nd.bkeys) all_lane0 = all_lane0 && B.cols[(size_t)k].lane == 0;
        bool any_positional = false;
        for (auto &c : B.cols) any_positional |= c.lane < 0;
        int32_t flags = 0;
        int64_t lo = 0, hi = 0;
        if (all_lane0 && !any_positional) {
            const ph_table *t = B.lanes[0].t;
            std::vector<ph_col> kv;
            for (int32_t k : nd.bkeys) kv.push_back(table_view(t, B.cols[(size_t)k].tcol));
            const auto &kc = t->cols[(size_t)B.cols[(size_t)nd.bkeys[0]].tcol];
            if (nk == 1 && kc.has_range && (kc.type == PH_I32 || kc.type == PH_I64)) { flags |= PH_JOIN_KEY_RANGE; lo = kc.min; hi = kc.max; }
            const bool su = optimistic && nk == 1 && kc.strict && B.single_identity();
            // every probe row is expected to find its row (a foreign key into a table nothing but the probe side's own
            // key domain has reduced): no Bloom bitmap, node table for composite keys
            if (B.covers && uniq == 2 && !(flags & PH_JOIN_KEY_RANGE)) flags |= PH_JOIN_FK_PROBES;
            int rc = PH_EUNSUPPORTED;
            // (a small filtered table probed by a table sixteen times its size is selected first: what a selective filter keeps — Q8's 13 k parts of one
            // type — builds the table whose occupied slot groups fit an LDS bitmap, and the 60 M-row probe reads that instead of the L2 bitmap of
            // the gated fill: 238 -> ~120 us)
            const bool small_under_big = B.single_identity() && !B.flags && t->nrows <= (4 << 20) && t->nrows * 16 <= P.n && P.n >= (32ll << 20) && !getenv("PH_PLAN_GATED_SMALL");   // (the gain is ~2 us per million probe rows, the selection's count a round trip)
            if (B.single_identity() && (flags & PH_JOIN_KEY_RANGE) && B.complex.empty() && B.pending.size() + (B.flags ? 1 : 0) == 1 && !small_under_big) {
                // Filter (or a semi-join's marks) under the build child rides along in the build
                ph_pred w{};
                ph_col wv{};
                if (B.flags) { wv.type = PH_CODE8; wv.data = B.flags; w.op = PH_EQ; w.k.type = PH_I32; w.k.i = 1; }   // marks are 0 / 1
                else { w = B.pending[0]; fix_dict_const(t, w.col, &w.k); wv = table_view(t, w.col); }
                rc = ph_join_build_where_ex(ctx, kv.data(), 1, &wv, w.op, &w.k, nullptr, t->nrows, su ? PH_JOIN_KEYS_SORTED_UNIQUE : 0, lo, hi, &j);
                if (rc == PH_OK) { how += su ? " build: gated sorted fill (filter rides along);" : " build: filter fused into the direct build;"; brow_is_rowid = true; }
                else if (rc != PH_EUNSUPPORTED) return rc;
            }
            if (rc == PH_EUNSUPPORTED) {
                PL_CHECK(apply_pending(p, &B));
                const bool ident = B.lanes[0].rows == nullptr;
                // SEMI against a key with duplicates / ANTI: the probe below only marks — a bitmap of the key values is the whole table
                const int32_t f2 = flags | (su && ident ? PH_JOIN_KEYS_SORTED_UNIQUE : 0) | (marks_only ? PH_JOIN_EXISTS_ONLY : 0);
                PL_CHECK(ph_join_build_ex(ctx, kv.data(), (int32_t)nk, B.lanes[0].rows, ident ? t->nrows : B.n, f2, lo, hi, &j));
                how += (f2 & PH_JOIN_KEYS_SORTED_UNIQUE) ? " build: sorted fill;" : ident ? " build: whole table;" : " build: selected rows;";
                brow_is_rowid = true;
            }
        } else {
            std::vector<int> want(nd.bkeys.begin(), nd.bkeys.end());
            PL_CHECK(positional(p, &B, want));
            std::vector<ph_col> kv;
            for (int32_t k : nd.bkeys) { const int32_t *s = nullptr; kv.push_back(col_view(B, B.cols[(size_t)k], &s)); }
            const PCol &k0 = B.cols[(size_t)nd.bkeys[0]];
            if (nk == 1 && k0.src && k0.src->cols[(size_t)k0.src_col].has_range && (k0.type == PH_I32 || k0.type == PH_I64)) {
                flags |= PH_JOIN_KEY_RANGE; lo = k0.src->cols[(size_t)k0.src_col].min; hi = k0.src->cols[(size_t)k0.src_col].max;
            } else if (nk == 1 && k0.grange && (k0.type == PH_I32 || k0.type == PH_I64)) {   // rows that arrived over an exchange: the ranks' reduced range
                flags |= PH_JOIN_KEY_RANGE; lo = k0.gmin; hi = k0.gmax;
            }
            PL_CHECK(ph_join_build_ex(ctx, kv.data(), (int32_t)nk, nullptr, B.n, flags | (marks_only ? PH_JOIN_EXISTS_ONLY : 0), lo, hi, &j));
            how += " build: intermediate rows;";
        }
        p->joins.push_back(j);
        how += std::string(" table=") + ph_join_kind(j) + ";";
    }
    (void)residual;

    // what the probe side's key columns are known to be afterwards: a subset of this table's keys
    int dom = -1;
    if (nd.join_type != PH_JT_ANTI && !B.covers) {
        p->domains.push_back(Domain{j, B.lazy() || B.single_identity() ? B.n : B.n});
        dom = (int)p->domains.size() - 1;
    }

    // ---- probe
    auto build_cols = [&](const int32_t *brow, std::vector<PCol> *cols, Rel *res) -> int {
        // the build side's columns in the result, addressed through the build row ids of the matches
        if (brow_is_rowid) {
            Lane bl; bl.t = B.lanes[0].t; bl.rows = brow; bl.asc = false; bl.dup_free = false;
            res->lanes.push_back(bl);
            for (auto c : B.cols) { c.lane = (int)res->lanes.size() - 1; c.ordered = false; c.domain = -1; cols->push_back(c); }
            return PH_OK;
        }
        // positions in B: B's lanes and positional columns gathered by them
        Rel Bc = B;
        PL_CHECK(compact(p, &Bc, brow, res->n));
        const int base = (int)res->lanes.size();
        for (auto &ln : Bc.lanes) { Lane l2 = ln; l2.asc = false; l2.dup_free = false; res->lanes.push_back(l2); }
        for (auto c : Bc.cols) { if (c.lane >= 0) c.lane += base; c.ordered = false; c.domain = -1; cols->push_back(c); }
        return PH_OK;
    };
    auto finish = [&](Rel *res, std::vector<PCol> &all) {
        res->cols.clear();
        for (int32_t oi : nd.out) res->cols.push_back(all[(size_t)oi]);
        res->covers = false;
        drop_unused_lanes(res);
    };
    auto tag_domain = [&](std::vector<PCol> &cols) {
        if (dom < 0) return;
        for (size_t k = 0; k < nk; k++) if (nk == 1) cols[(size_t)nd.pkeys[k]].domain = dom;
    };

    // (1) existence only and the result feeds another build: marks, no pair list, no count
    if (exists_only && nd.join_type != PH_JT_ANTI && as_build && P.single_identity() && P.pending.size() <= 1 && P.complex.empty() && !P.flags) {
        bool lane_keys = true;
        for (int32_t k : nd.pkeys) lane_keys = lane_keys && P.cols[(size_t)k].lane == 0;
        if (lane_keys) {
            const ph_table *t = P.lanes[0].t;
            std::vector<ph_col> kv;
            for (int32_t k : nd.pkeys) kv.push_back(table_view(t, P.cols[(size_t)k].tcol));
            void *f = nullptr;
            PL_CHECK(palloc(p, t->nrows + 64, &f));
            int rc = PH_EUNSUPPORTED;
            if (P.pending.size() == 1) {
                ph_pred w = P.pending[0];
                fix_dict_const(t, w.col, &w.k);
                ph_col wv = table_view(t, w.col);
                rc = ph_join_probe_mark_where(j, kv.data(), &wv, w.op, &w.k, t->nrows, (uint8_t *)f);
            } else rc = ph_join_probe_mark(j, kv.data(), nullptr, t->nrows, (uint8_t *)f);
            if (rc == PH_OK) {
                *out = P;
                out->pending.clear();
                out->flags = (const uint8_t *)f;
                std::vector<PCol> all = P.cols;
                tag_domain(all);
                out->cols.clear();
                for (int32_t oi : nd.out) out->cols.push_back(all[(size_t)oi]);
                out->covers = false;
                note(p, "join#%d:%s probe: filter + semi-join marks in one pass over %lld rows (no pairs, no count)", idx, how.c_str(), (long long)t->nrows);
                return PH_OK;
            }
            if (rc != PH_EUNSUPPORTED) return rc;
        }
    }

    // (2) ANTI: marks, then the rows without one; SEMI against a build key with duplicates: marks, then the rows with one
    // (a pair list would repeat the probe row once per duplicate)
    if (nd.join_type == PH_JT_ANTI || (exists_only && !unique)) {
        const bool anti = nd.join_type == PH_JT_ANTI;
        PL_CHECK(apply_pending(p, &P));
        KeySide pk;
        PL_CHECK(key_side(p, &P, nd.pkeys, &pk));
        void *f = nullptr, *sel = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) + 64, &f));
        PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &sel));
        int64_t m = 0;
        if (P.n > 0) {
            PL_CHECK(ph_join_probe_mark(j, pk.views.data(), pk.sel, P.n, (uint8_t *)f));
            ph_col fc{};
            fc.type = PH_CODE8; fc.data = f;
            ph_const want{};
            want.type = PH_I32; want.i = anti ? 0 : 1;
            PL_CHECK(ph_filter_select(ctx, &fc, P.n, PH_EQ, &want, nullptr, P.n, (int32_t *)sel, &m));
        }
        *out = P;
        PL_CHECK(compact(p, out, (const int32_t *)sel, m));
        std::vector<PCol> all = out->cols;
        if (!anti) tag_domain(all);
        finish(out, all);
        note(p, "join#%d:%s probe: %s (marks + selection), %lld of %lld rows", idx, how.c_str(), anti ? "anti" : "semi", (long long)m, (long long)P.n);
        return PH_OK;
    }

    // (3) N:1 where every probe row is expected to find its row: a lookup, the intermediate keeps its rows
    if (n_to_1 && B.covers && uniq == 2) {
        PL_CHECK(apply_pending(p, &P));
        KeySide pk;
        PL_CHECK(key_side(p, &P, nd.pkeys, &pk));
        void *o = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &o));
        bool compacted = false;
        if (optimistic) {
            if (P.n > 0) PL_CHECK(ph_join_lookup_strict(j, pk.views.data(), pk.sel, P.n, (int32_t *)o));
            how += " probe: strict N:1 lookup";
        } else {
            void *st = nullptr;
            PL_CHECK(palloc(p, 8, &st));
            PL_CHECK(ph_dev_memset(ctx, st, 0, 8));
            if (P.n > 0) PL_CHECK(ph_join_lookup(j, pk.views.data(), pk.sel, P.n, (int32_t *)o, (int32_t *)st));
            int32_t stats[2] = {0, 0};
            PL_CHECK(ph_dev_download(ctx, stats, st, 8));
            if (stats[1]) { set_error("ph_plan: join#%d build key declared unique has duplicates (%d probe rows met several build rows)", idx, stats[1]); return PH_ECONSTRAINT; }
            how += " probe: counted N:1 lookup";
            if (stats[0]) {   // inner-join semantics: probe rows without a build row leave the result
                ph_col rc{};
                rc.type = PH_I32; rc.data = o;
                ph_const zero{};
                zero.type = PH_I32; zero.i = 0;
                void *keep = nullptr;
                PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 4, &keep));
                int64_t m = 0;
                PL_CHECK(ph_filter_select(ctx, &rc, P.n, PH_GE, &zero, nullptr, P.n, (int32_t *)keep, &m));
                // the lookup result joins the relation as a positional column so that the compaction carries it along
                Rel Pc = P;
                PCol tmp; tmp.type = PH_I32; tmp.data = o;
                Pc.cols.push_back(tmp);
                if (pk.rowids && Pc.lanes[0].rows == nullptr) { /* identity lane: positions are row ids */ }
                PL_CHECK(compact(p, &Pc, (const int32_t *)keep, m));
                o = const_cast<void *>(Pc.cols.back().data);
                Pc.cols.pop_back();
                P = Pc;
                compacted = true;
                how += " (+ compaction of the misses)";
            }
        }
        (void)compacted;
        *out = P;
        std::vector<PCol> all = P.cols;
        tag_domain(all);
        out->cols = all;   // build_cols appends lanes to *out
        std::vector<PCol> bcols;
        PL_CHECK(build_cols((const int32_t *)o, &bcols, out));
        for (auto &c : bcols) all.push_back(c);
        finish(out, all);
        note(p, "join#%d:%s, %lld rows", idx, how.c_str(), (long long)out->n);
        return PH_OK;
    }

    // (4) pairs: the general inner probe (also existence-only joins whose result is probed or aggregated next, and N:1
    // joins against a filtered build side, where misses are the rule)
    {
        int64_t m = 0;
        int32_t *prow = nullptr, *brow = nullptr;
        const char *form = "";
        int rc = PH_EUNSUPPORTED;
        KeySide pk;
        bool lane_keys = P.lanes.size() == 1;
        for (int32_t k : nd.pkeys) lane_keys = lane_keys && P.cols[(size_t)k].lane == 0;
        for (auto &c : P.cols) lane_keys = lane_keys && c.lane == 0;
        if (lane_keys && P.single_identity() && P.pending.size() == 1 && P.complex.empty() && !P.flags) {   // Filter -> probe in one pass
            const ph_table *t = P.lanes[0].t;
            for (int32_t k : nd.pkeys) pk.views.push_back(table_view(t, P.cols[(size_t)k].tcol));
            pk.sel = nullptr; pk.n = t->nrows; pk.rowids = true;
            rc = pair_probe(p, j, pk, &P.pending[0], t, nullptr, &m, &prow, &brow, &form);
            if (rc == PH_OK) P.pending.clear();
            else if (rc != PH_EUNSUPPORTED) return rc;
        }
        if (rc == PH_EUNSUPPORTED) {
            PL_CHECK(apply_pending(p, &P));
            pk = KeySide{};
            PL_CHECK(key_side(p, &P, nd.pkeys, &pk));
            PL_CHECK(pair_probe(p, j, pk, nullptr, nullptr, nullptr, &m, &prow, &brow, &form, unique ? 0 : B.n));
        }
        *out = P;
        if (pk.rowids) {   // prow = row ids of the one lane, ascending: the lane IS the pair list's probe side
            out->lanes[0].rows = prow;
            out->lanes[0].dup_free = out->lanes[0].dup_free && unique;
            out->n = m;
        } else {
            PL_CHECK(compact(p, out, prow, m));
            for (auto &ln : out->lanes) ln.dup_free = ln.dup_free && unique;
        }
        std::vector<PCol> all = out->cols;
        // duplicates of a probe row stay adjacent, so a column's order survives the pair list — unless the table form emits
        // its pairs partition by partition (the radix form of big build sides without a dense key range)
        if (!ph_join_pairs_ordered(j)) {
            for (auto &c : all) c.ordered = false;
            for (auto &ln : out->lanes) ln.asc = false;
        }
        tag_domain(all);
        out->cols = all;
        if (nd.join_type == PH_JT_INNER) {
            std::vector<PCol> bcols;
            PL_CHECK(build_cols(brow, &bcols, out));
            for (auto &c : bcols) all.push_back(c);
            // The probe key of a matched row EQUALS the build key it matched: when the build side is a small table and the
            // probe side a big one, later readers of the probe key are served from the build key through the pair's build
            // row (a cache-resident table) instead of one more 128-byte line per surviving row of the big table.
            if (nk == 1 && brow_is_rowid) {
                PCol &pkc = all[(size_t)nd.pkeys[0]];
                const PCol &bkc = all[nP + (size_t)nd.bkeys[0]];
                if (pkc.lane >= 0 && bkc.lane >= 0 && pkc.type == bkc.type && out->lanes[(size_t)pkc.lane].t->nrows >= 8 * out->lanes[(size_t)bkc.lane].t->nrows) {
                    const bool ordered = pkc.ordered;
                    const int domain = pkc.domain;
                    pkc = bkc;
                    pkc.ordered = ordered;
                    pkc.domain = domain;
                }
            }
        }
        finish(out, all);
        // late materialisation, ONCE: a sparse row-id vector into a big table — every column the operators above
        // need from it is fetched in one pass (all reads of a row in flight together, the ids read once)
        for (size_t L = 0; L < out->lanes.size(); L++) {
            std::vector<int> cs;
            for (size_t c = 0; c < out->cols.size(); c++) {
                if (out->cols[c].lane != (int)L) continue;
                const auto &tc = out->lanes[L].t->cols[(size_t)out->cols[c].tcol];
                if (width_of(tc.type) == 0 || tc.validity) continue;   // VARCHAR (offsets + bytes) and NULL-able columns stay behind the row ids
                cs.push_back((int)c);
            }
            if (cs.size() >= 2 && out->lanes[L].rows && out->lanes[L].t->nrows >= 8 * std::max<int64_t>(m, 1) && out->lanes[L].t->nrows >= (1 << 20))
                PL_CHECK(positional(p, out, cs));
        }
        drop_unused_lanes(out);
        note(p, "join#%d:%s probe: %s, %lld pairs from %lld probe rows", idx, how.c_str(), form, (long long)m, (long long)pk.n);
        return PH_OK;
    }
}

// ---- LEFT OUTER join (NextLeftJoin, join_scan.go:67-88): the inner matches, then the probe rows without one, their build side NULL.
// Pairs come from the general inner probe; the unmatched probe rows from a mark probe of the same table (found == 0). The result's
// probe lanes are [pair probe rows | unmatched rows], its build lane [pair build rows | -1 ...] and marked NULL-able: positional()
// gives the columns read through it a validity bitmap (ph_rowid_validity). The reference emits the two kinds chunk by chunk; here all
// matches come first — the same rows (an ORDER BY above decides the order, as it must for a hash join's output anyway).
int left_join_rels(ph_plan *p, int idx, const Node &nd, Rel P, Rel B, Rel *out) {
    ph_ctx *ctx = p->ctx;
    const size_t nP = P.cols.size();
    PL_CHECK(apply_pending(p, &P));
    PL_CHECK(apply_pending(p, &B));
    // build: the build side's rows (a filtered base table: row ids of the table; else positions in B)
    KeySide bk;
    PL_CHECK(key_side(p, &B, nd.bkeys, &bk));
    if (!bk.rowids) { set_error("ph_plan: the build side of a LEFT join must be a (filtered) base table"); return PH_EUNSUPPORTED; }
    int32_t flags = 0;
    int64_t lo = 0, hi = 0;
    {
        const auto &kc = B.lanes[0].t->cols[(size_t)B.cols[(size_t)nd.bkeys[0]].tcol];
        if (nd.bkeys.size() == 1 && kc.has_range && (kc.type == PH_I32 || kc.type == PH_I64)) { flags |= PH_JOIN_KEY_RANGE; lo = kc.min; hi = kc.max; }
    }
    ph_join *j = nullptr;
    PL_CHECK(ph_join_build_ex(ctx, bk.views.data(), (int32_t)nd.bkeys.size(), bk.sel, bk.sel ? B.n : B.lanes[0].t->nrows, flags, lo, hi, &j));
    p->joins.push_back(j);
    KeySide pk;
    PL_CHECK(key_side(p, &P, nd.pkeys, &pk));
    // pairs
    int64_t m = 0;
    int32_t *prow = nullptr, *brow = nullptr;
    const char *form = "";
    PL_CHECK(pair_probe(p, j, pk, nullptr, nullptr, nullptr, &m, &prow, &brow, &form, B.n));
    // the probe rows without a match
    void *f = nullptr, *un = nullptr;
    PL_CHECK(palloc(p, std::max<int64_t>(pk.n, 1) + 64, &f));
    PL_CHECK(palloc(p, std::max<int64_t>(pk.n, 1) * 4, &un));
    int64_t u = 0;
    if (pk.n > 0) {
        PL_CHECK(ph_join_probe_mark(j, pk.views.data(), pk.sel, pk.n, (uint8_t *)f));
        ph_col fc{};
        fc.type = PH_CODE8; fc.data = f;
        ph_const zero{};
        zero.type = PH_I32; zero.i = 0;
        PL_CHECK(ph_filter_select(ctx, &fc, pk.n, PH_EQ, &zero, nullptr, pk.n, (int32_t *)un, &u));   // positions among the probe rows
    }
    const int64_t total = m + u;
    // the unmatched rows in the pair list's terms: with row-id keys the probe rows are row ids of lane 0 (sel[position], or the position itself)
    const int32_t *un_rows = (const int32_t *)un;
    if (pk.rowids && pk.sel && u > 0) {
        ph_col sv{};
        sv.type = PH_I32; sv.data = pk.sel;
        void *g = nullptr;
        PL_CHECK(palloc(p, u * 4, &g));
        PL_CHECK(ph_gather(ctx, &sv, (const int32_t *)un, u, g));
        un_rows = (const int32_t *)g;
    }
    void *pall = nullptr, *ball = nullptr;
    PL_CHECK(palloc(p, std::max<int64_t>(total, 1) * 4, &pall));
    PL_CHECK(palloc(p, std::max<int64_t>(total, 1) * 4, &ball));
    if (m > 0) {
        PH_HIP(hipMemcpyAsync(pall, prow, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemcpyAsync(ball, brow, (size_t)m * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (u > 0) {
        PH_HIP(hipMemcpyAsync((int32_t *)pall + m, un_rows, (size_t)u * 4, hipMemcpyDeviceToDevice, ctx->stream));
        PH_HIP(hipMemsetAsync((int32_t *)ball + m, 0xFF, (size_t)u * 4, ctx->stream));   // row id -1
    }
    *out = P;
    if (pk.rowids) {
        out->lanes[0].rows = (const int32_t *)pall;
        out->lanes[0].asc = false;
        out->lanes[0].dup_free = false;
        out->n = total;
    } else {
        PL_CHECK(compact(p, out, (const int32_t *)pall, total));
        for (auto &ln : out->lanes) { ln.asc = false; ln.dup_free = false; }
    }
    std::vector<PCol> all = out->cols;
    for (auto &c : all) { c.ordered = false; c.domain = -1; }
    Lane bl;
    bl.t = B.lanes[0].t; bl.rows = (const int32_t *)ball; bl.asc = false; bl.dup_free = false; bl.nullable = true;
    out->lanes.push_back(bl);
    for (auto c : B.cols) {
        if (c.lane != 0) { set_error("ph_plan: the build side of a LEFT join must be a (filtered) base table"); return PH_EUNSUPPORTED; }
        c.lane = (int)out->lanes.size() - 1; c.ordered = false; c.domain = -1;
        all.push_back(c);
    }
    (void)nP;
    out->cols.clear();
    for (int32_t oi : nd.out) out->cols.push_back(all[(size_t)oi]);
    out->covers = false;
    out->pending.clear(); out->complex.clear(); out->flags = nullptr;
    drop_unused_lanes(out);
    note(p, "join#%d: LEFT OUTER: table=%s, %lld pairs + %lld probe rows without a match (their build side NULL)", idx, ph_join_kind(j), (long long)m, (long long)u);
    return PH_OK;
}

// ---- multi-rank: bring every group's rows onto one rank before an aggregate that needs WHOLE groups (an aggregate below other operators; a root
// with a top-k, a HAVING or a DISTINCT aggregate). Nothing moves when a group key's per-rank value ranges are disjoint (the ranks' statistics:
// a database split by order ranges grouped by the order key); an ungrouped aggregate gets its (small) input replicated; otherwise the rows are
// hash-partitioned by the first group key that is a plain column.
int whole_groups(ph_plan *p, int idx, Rel *R) {
    const Node &nd = p->nodes[(size_t)idx];
    if (nd.groups.empty()) {
        char what[64];
        snprintf(what, sizeof what, "input of the ungrouped aggregate #%d", idx);
        return replicate_rel(p, R, what);
    }
    int part_key = -1;
    for (auto &g : nd.groups) {
        if (g.e.kind != PH_PE_COL || g.e.col < 0 || g.e.col >= (int)R->cols.size()) continue;
        const PCol &c = R->cols[(size_t)g.e.col];
        bool dj = false;
        PL_CHECK(ranks_disjoint(p, c, &dj));
        if (dj) { note(p, "agg#%d: the ranks' ranges of group key column %d are disjoint: every group is whole on its rank", idx, g.e.col); return PH_OK; }
        if (part_key < 0 && (c.type == PH_I32 || c.type == PH_I64 || c.type == PH_DATE || c.type == PH_DEC64 || c.type == PH_CODE8) && !c.validity) part_key = g.e.col;
    }
    if (part_key < 0) { set_error("ph_plan: aggregate #%d across ranks needs a group key that is a plain fixed-width column", idx); return PH_EUNSUPPORTED; }
    char what[64];
    snprintf(what, sizeof what, "input of aggregate #%d", idx);
    return repartition_rel(p, R, part_key, what);
}

int lower_node(ph_plan *p, int idx, bool as_build, Rel *out);

// A node that several parents reference (a subtree used twice: Q21's l1 side feeds the pair join and the SEMI / ANTI join that closes
// the step) is lowered once per run and role: the relation is a value — row-id vectors and positional columns in run temporaries that
// nobody writes again — so every further parent starts from a copy of it.
int lower(ph_plan *p, int idx, bool as_build, Rel *out) {
    if (idx < 0 || idx >= (int)p->nodes.size()) { set_error("ph_plan: child index %d out of range", idx); return PH_EINVAL; }
    if (p->parents[(size_t)idx] < 2 || p->nodes[(size_t)idx].kind == PH_PN_SCAN) return lower_node(p, idx, as_build, out);
    const auto key = std::make_pair(idx, as_build);
    auto it = p->memo.find(key);
    if (it != p->memo.end()) { *out = *it->second; note(p, "node#%d: lowered before in this run, reused", idx); return PH_OK; }
    PL_CHECK(lower_node(p, idx, as_build, out));
    PL_CHECK(apply_pending(p, out));   // (filters still pending would be applied once per parent)
    p->memo[key] = std::make_shared<Rel>(*out);
    return PH_OK;
}

int lower_node(ph_plan *p, int idx, bool as_build, Rel *out) {
    const Node &nd = p->nodes[(size_t)idx];
    switch (nd.kind) {
    case PH_PN_SCAN: {
        const ph_table *t = nd.table;
        *out = Rel{};
        out->n = t->nrows;
        Lane ln; ln.t = t;
        out->lanes.push_back(ln);
        for (int32_t c : nd.cols) {
            if (c < 0 || c >= (int32_t)t->cols.size()) { set_error("ph_plan: scan column %d out of range", c); return PH_EINVAL; }
            PCol pc;
            pc.type = t->cols[(size_t)c].type; pc.scale = t->cols[(size_t)c].scale; pc.lane = 0; pc.tcol = c; pc.src = t; pc.src_col = c;
            pc.ordered = !p->conservative && t->cols[(size_t)c].ascending;
            out->cols.push_back(pc);
        }
        for (size_t i = 0; i < nd.preds.size(); i++) {
            ph_pred pr = nd.preds[i];
            if (pr.col < 0 || pr.col >= (int32_t)t->cols.size()) { set_error("ph_plan: scan predicate column %d out of range", pr.col); return PH_EINVAL; }
            pr.k.s = nd.pred_strs[i].empty() ? nullptr : nd.pred_strs[i].c_str();
            out->pending.push_back(pr);
        }
        if (!nd.bools.empty()) { out->complex.push_back(nd.bools); out->complex.back().fix(); }
        out->covers = nd.preds.empty() && nd.bools.empty();
        out->replicated = multi(p) && t->replicated;
        note(p, "scan#%d: %lld rows, %zu pushed conjuncts", idx, (long long)t->nrows, nd.preds.size() + (nd.bools.empty() ? 0 : 1));
        return PH_OK;
    }
    case PH_PN_FILTER: {
        PL_CHECK(lower(p, nd.child[0], as_build, out));
        bool pushed = true;
        // conjuncts over the columns of a lazy single-table relation join its pending list (and may still fuse)
        for (size_t i = 0; i < nd.preds.size(); i++) {
            const ph_pred &pr = nd.preds[i];
            if (pr.col < 0 || pr.col >= (int32_t)out->cols.size()) { set_error("ph_plan: filter column %d out of range", pr.col); return PH_EINVAL; }
            pushed = pushed && out->single_identity() && !out->flags && out->cols[(size_t)pr.col].lane == 0;
        }
        for (auto &b : nd.bools.nodes)
            if (b.kind == PH_B_CMP) pushed = pushed && out->single_identity() && !out->flags && b.col >= 0 && b.col < (int)out->cols.size() && out->cols[(size_t)b.col].lane == 0 &&
                                             (b.k.type != PH_COLREF || (b.k.i >= 0 && b.k.i < (int64_t)out->cols.size() && out->cols[(size_t)b.k.i].lane == 0));
        if (pushed) {
            for (size_t i = 0; i < nd.preds.size(); i++) {
                ph_pred pr = nd.preds[i];
                pr.col = out->cols[(size_t)pr.col].tcol;
                pr.k.s = nd.pred_strs[i].empty() ? nullptr : nd.pred_strs[i].c_str();
                out->pending.push_back(pr);
            }
            if (!nd.bools.empty()) {   // the tree's columns become table columns
                BoolTree bt = nd.bools;
                for (auto &b : bt.nodes) if (b.kind == PH_B_CMP) { b.col = out->cols[(size_t)b.col].tcol; if (b.k.type == PH_COLREF) b.k.i = out->cols[(size_t)b.k.i].tcol; }
                out->complex.push_back(bt);
                out->complex.back().fix();
            }
            out->covers = false;
            return PH_OK;
        }
        PL_CHECK(apply_pending(p, out));
        const int32_t *sel = nullptr;
        int64_t cnt = out->n;
        for (size_t i = 0; i < nd.preds.size() && cnt > 0; i++) {
            ph_pred pr = nd.preds[i];
            pr.k.s = nd.pred_strs[i].empty() ? nullptr : nd.pred_strs[i].c_str();
            PL_CHECK(positional(p, out, {pr.col}));
            const PCol &pc = out->cols[(size_t)pr.col];
            fix_dict_const(pc.src, pc.src_col, &pr.k);
            const int32_t *s = nullptr;
            ph_col v = col_view(*out, pc, &s);
            fix_num_const(v, &pr.k);
            void *o = nullptr;
            PL_CHECK(palloc(p, cnt * 4, &o));
            int64_t m = 0;
            PL_CHECK(ph_filter_select(p->ctx, &v, out->n, pr.op, &pr.k, sel, cnt, (int32_t *)o, &m));
            sel = (const int32_t *)o;
            cnt = m;
        }
        if (!nd.bools.empty() && cnt > 0) {
            BoolTree bt = nd.bools;
            bt.fix();
            const int32_t *o = nullptr;
            int64_t m = 0;
            PL_CHECK(eval_bool(p, out, false, bt, 0, sel, sel ? cnt : out->n, &o, &m));
            sel = o;
            cnt = m;
        }
        if (sel) PL_CHECK(compact(p, out, sel, cnt));
        out->covers = false;
        note(p, "filter#%d: %lld rows kept", idx, (long long)out->n);
        return PH_OK;
    }
    case PH_PN_JOIN:
        return lower_join(p, idx, as_build, out);
    case PH_PN_PROJECT: {
        Rel child;
        PL_CHECK(lower(p, nd.child[0], as_build, &child));
        std::vector<PCol> cols;
        for (auto &e : nd.exprs) {
            PCol c;
            PL_CHECK(eval_expr(p, &child, e, &c));
            cols.push_back(c);
        }
        *out = child;
        out->cols = cols;
        drop_unused_lanes(out);
        note(p, "project#%d: %zu expressions", idx, nd.exprs.size());
        return PH_OK;
    }
    case PH_PN_AGG: {
        // an aggregate BELOW other operators (a subquery's GROUP BY [.. HAVING = the Filter above it] under a join): its groups
        // become a device-resident relation — key columns and aggregate values as positional columns — and never visit the host
        // ---- children per parent: Agg(parent key; count(child column)...) <- LEFT JOIN(parent, child) on that key, the parent key unique (Q13's
        // orders per customer): the child keys are counted into an array over their value range and every parent row reads its count
        // (ph_count_by_key) — no pair list, no second fold. NULL where there is no child (CountOp over the NULL-extended row).
        Rel R;
        bool have_R = false;
        if (!multi(p) && nd.groups.size() == 1 && nd.groups[0].e.kind == PH_PE_COL && !nd.aggs.empty() && nd.child[0] >= 0 && p->parents[(size_t)nd.child[0]] < 2 &&
            p->nodes[(size_t)nd.child[0]].kind == PH_PN_JOIN && p->nodes[(size_t)nd.child[0]].join_type == PH_JT_LEFT && p->nodes[(size_t)nd.child[0]].pkeys.size() == 1 &&
            !getenv("PH_PLAN_NO_COUNT_PUSHDOWN")) {
            const Node &jn = p->nodes[(size_t)nd.child[0]];
            bool shape = true;
            for (auto &a : nd.aggs) shape = shape && a.kind == PH_A_COUNT && a.arg.e.kind == PH_PE_COL && a.arg.e.col >= 0 && (size_t)a.arg.e.col < jn.out.size();
            shape = shape && nd.groups[0].e.col >= 0 && (size_t)nd.groups[0].e.col < jn.out.size() && jn.out[(size_t)nd.groups[0].e.col] == jn.pkeys[0];
            if (shape) {
                Rel P, B;
                PL_CHECK(lower(p, jn.child[0], false, &P));
                PL_CHECK(lower(p, jn.child[1], true, &B));
                const size_t nP = P.cols.size();
                bool ok = jn.pkeys[0] >= 0 && (size_t)jn.pkeys[0] < nP && jn.bkeys[0] >= 0 && (size_t)jn.bkeys[0] < B.cols.size() && key_unique(P, {jn.pkeys[0]}) > 0 &&
                          B.lanes.size() == 1 && !B.lanes[0].nullable;
                if (ok) {
                    const PCol &bk = B.cols[(size_t)jn.bkeys[0]];
                    ok = bk.lane == 0 && bk.tcol >= 0 && B.lanes[0].t->cols[(size_t)bk.tcol].has_range && width_of(bk.type) != 0 &&
                         width_of(bk.type) == width_of(P.cols[(size_t)jn.pkeys[0]].type);
                    for (auto &a : nd.aggs) {   // count(child column): a column of the build side that cannot be NULL itself
                        const int32_t oc = jn.out[(size_t)a.arg.e.col];
                        ok = ok && oc >= (int32_t)nP && (size_t)oc - nP < B.cols.size();
                        if (ok) { const PCol &c = B.cols[(size_t)oc - nP]; ok = c.lane == 0 && c.tcol >= 0 && !B.lanes[0].t->cols[(size_t)c.tcol].validity; }
                    }
                    if (ok) {
                        const auto &kc = B.lanes[0].t->cols[(size_t)bk.tcol];
                        const __int128 range = (__int128)kc.max - (__int128)kc.min + 1;
                        ok = range >= 1 && range <= (1ll << 28);
                        if (ok) {
                            PL_CHECK(apply_pending(p, &P));
                            PL_CHECK(apply_pending(p, &B));
                            KeySide pk, ck;
                            PL_CHECK(key_side(p, &P, {jn.pkeys[0]}, &pk));
                            PL_CHECK(key_side(p, &B, {jn.bkeys[0]}, &ck));
                            void *cnt = nullptr, *val = nullptr;
                            PL_CHECK(palloc(p, std::max<int64_t>(P.n, 1) * 8, &cnt));
                            PL_CHECK(palloc(p, (P.n + 63) / 64 * 8 + 64, &val));
                            PL_CHECK(ph_count_by_key(p->ctx, &ck.views[0], ck.sel, ck.n, kc.min, (int64_t)range, &pk.views[0], pk.sel, pk.n, (int64_t *)cnt, (uint8_t *)val));
                            *out = P;
                            out->cols.clear();
                            out->cols.push_back(P.cols[(size_t)jn.pkeys[0]]);
                            for (size_t a = 0; a < nd.aggs.size(); a++) {
                                PCol c;
                                c.type = PH_DEC64; c.scale = 0; c.data = cnt; c.validity = (const uint8_t *)val;
                                out->cols.push_back(c);
                            }
                            out->covers = false;
                            out->pending.clear(); out->complex.clear(); out->flags = nullptr;
                            drop_unused_lanes(out);
                            note(p, "agg#%d over join#%d: children per parent — %lld child keys counted over a range of %lld, %lld parent rows read their count (no pairs, no second fold)", idx,
                                 nd.child[0], (long long)ck.n, (long long)range, (long long)P.n);
                            return PH_OK;
                        }
                    }
                }
                // not the shape after all: the join over the children lowered here (they ran once)
                PL_CHECK(join_rels(p, nd.child[0], jn, P, B, false, &R));
                have_R = true;
            }
        }
        if (!have_R) PL_CHECK(lower(p, nd.child[0], false, &R));
        {   // the keys of the join above (lower_join): only input rows some probe row can meet are aggregated
            auto it = p->pushed.find(idx);
            if (it != p->pushed.end() && it->second.gcol >= 0 && (size_t)it->second.gcol < R.cols.size()) {
                Rel K = *it->second.probe;
                if (R.n >= (1 << 20) && K.n * 4 <= R.n && width_of(R.cols[(size_t)it->second.gcol].type) == width_of(K.cols[(size_t)it->second.pkey].type) &&
                    width_of(R.cols[(size_t)it->second.gcol].type) != 0) {
                    Node sj;
                    sj.kind = PH_PN_JOIN;
                    sj.join_type = PH_JT_SEMI;
                    sj.pkeys = {it->second.gcol};
                    sj.bkeys = {it->second.pkey};
                    for (size_t c = 0; c < R.cols.size(); c++) sj.out.push_back((int32_t)c);
                    const int64_t before = R.n;
                    Rel R2;
                    PL_CHECK(join_rels_local(p, idx, sj, R, K, false, &R2));
                    PL_CHECK(apply_pending(p, &R2));
                    note(p, "agg#%d: input reduced to the rows the keys of join#%d ask for: %lld of at most %lld", idx, it->second.join, (long long)R2.n, (long long)before);
                    R = R2;
                }
            }
        }
        if (multi(p) && !R.replicated) PL_CHECK(whole_groups(p, idx, &R));   // every group's rows on one rank before they are aggregated
        const bool agg_replicated = R.replicated;
        ph_agg *agg = nullptr;
        std::vector<KeyInfo> kinfo;
        std::vector<int32_t> ascale, atype;
        std::vector<KeyPack> packs;
        std::vector<bool> nullable;
        int rc = sink_into_agg(p, idx, R, false, &agg, &kinfo, &ascale, &atype, &packs, &nullable);
        if (agg) p->inner_aggs.push_back(agg);
        PL_CHECK(rc);
        int64_t ng = 0;
        PL_CHECK(ph_agg_group_count(agg, &ng));
        *out = Rel{};
        out->n = ng;
        out->covers = false;
        out->replicated = agg_replicated;
        for (size_t k = 0; k < nd.groups.size(); k++) {
            PCol c;
            c.type = kinfo[k].type; c.scale = kinfo[k].scale; c.src = kinfo[k].table; c.src_col = kinfo[k].col;
            if (c.type == PH_STR) { set_error("ph_plan: VARCHAR keys of an aggregate below other operators"); return PH_EUNSUPPORTED; }
            void *d = nullptr, *val = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(ng, 1) * 8, &d));
            if (nullable[k]) PL_CHECK(palloc(p, (ng + 7) / 8 + 64, &val));   // the NULL key's group keeps its NULL (operators that cannot take NULLs refuse it)
            int64_t n2 = 0;
            PL_CHECK(ph_agg_keys_dev(agg, (int32_t)k, d, (uint8_t *)val, ng, &n2));
            c.data = d;
            c.validity = (const uint8_t *)val;
            out->cols.push_back(c);
        }
        for (size_t a = 0; a < nd.aggs.size(); a++) {
            PCol c;
            // SUM / COUNT results are HUGEINT (integer input) or DECIMAL: both travel as int64 at the argument's scale — a scale-0
            // decimal stands in for the HUGEINT, whose one comparison ('>') DECIMAL has too (function_operator_boolean.go:431-442)
            const bool minmax = nd.aggs[a].kind == PH_A_MIN || nd.aggs[a].kind == PH_A_MAX;
            c.type = minmax && (atype[a] == PH_I32 || atype[a] == PH_DATE) ? PH_I64 : PH_DEC64;
            c.scale = nd.aggs[a].kind == PH_A_COUNT || nd.aggs[a].kind == PH_A_COUNT_STAR ? 0 : ascale[a];
            void *d = nullptr, *val = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(ng, 1) * 8, &d));
            // a group whose inputs were all NULL is NULL, not 0 / INT64_MAX: the value column carries the bitmap wherever that can happen
            if (nullable[nd.groups.size() + a]) PL_CHECK(palloc(p, (ng + 7) / 8 + 64, &val));
            int64_t n2 = 0;
            PL_CHECK(ph_agg_values_dev(agg, (int32_t)a, (int64_t *)d, (uint8_t *)val, ng, &n2));
            c.data = d;
            c.validity = (const uint8_t *)val;
            out->cols.push_back(c);
        }
        note(p, "agg#%d: %lld groups stay on the device as a relation (%zu keys, %zu aggregates)", idx, (long long)ng, nd.groups.size(), nd.aggs.size());
        return PH_OK;
    }
    default:
        set_error("ph_plan: node %d has kind %d, which cannot be a child", idx, nd.kind);
        return PH_EINVAL;
    }
}

// ---- one relation into one aggregate table (aggExecutor's build phase: executeExprs for the group keys and the aggregate
// arguments, then Sink; executor_aggr.go:110-142). Used by the root and by aggregates below other operators.
// nullable (optional): per group key, then per aggregate — can the RESULT column hold a NULL? A key can when its column carries a validity
// bitmap; an aggregate only when its argument does (SUM / MIN / MAX / COUNT of a group whose inputs were all NULL finalise to NULL,
// function_aggr.go:813-823, 950-962; a group exists because a row reached it, so without NULL inputs every state is set).
int sink_into_agg(ph_plan *p, int idx, Rel &R, bool allow_pack, ph_agg **aggp, std::vector<KeyInfo> *kinfo, std::vector<int32_t> *ascale,
                  std::vector<int32_t> *atype, std::vector<KeyPack> *packs, std::vector<bool> *nullable) {
    const Node &nd = p->nodes[(size_t)idx];
    ph_ctx *ctx = p->ctx;
    PL_CHECK(apply_pending(p, &R));
    // group keys and aggregate arguments as columns of the relation
    std::vector<PCol> kc(nd.groups.size()), ac(nd.aggs.size());
    std::vector<bool> str_key(nd.groups.size(), false);
    for (size_t g = 0; g < nd.groups.size(); g++) {
        PL_CHECK(eval_expr(p, &R, nd.groups[g], &kc[g]));
        if (kc[g].type == PH_STR) {   // a VARCHAR key: group by the string's code (its representative row in the column)
            if (nd.groups[g].e.kind != PH_PE_COL) { set_error("ph_plan: VARCHAR group key must be a column"); return PH_EUNSUPPORTED; }
            // ... unless the table's unique key is a group key too (Q10 groups by c_custkey AND c_name, c_phone, c_address, c_comment): rows of one
            // group are then ONE table row, and that row's id is as good a code as the first row holding the same string — no interning pass
            // over the strings (four of them: 600 us of Q10 at SF10)
            const PCol &sc = R.cols[(size_t)nd.groups[g].e.col];
            bool by_row = false;
            if (sc.lane >= 0 && sc.tcol >= 0 && !R.lanes[(size_t)sc.lane].nullable && R.lanes[(size_t)sc.lane].rows != nullptr && !getenv("PH_PLAN_NO_ROW_CODES")) {
                const ph_table *t = R.lanes[(size_t)sc.lane].t;
                for (size_t g2 = 0; g2 < nd.groups.size() && !by_row; g2++) {
                    if (g2 == g || nd.groups[g2].e.kind != PH_PE_COL || nd.groups[g2].e.col < 0 || (size_t)nd.groups[g2].e.col >= R.cols.size()) continue;
                    const PCol &c2 = R.cols[(size_t)nd.groups[g2].e.col];
                    if (c2.lane != sc.lane || c2.tcol < 0 || t->cols[(size_t)c2.tcol].validity) continue;
                    by_row = t->cols[(size_t)c2.tcol].strict;
                    for (auto &u : t->unique_keys) by_row = by_row || (u.size() == 1 && u[0] == c2.tcol);
                }
                if (by_row) {
                    kc[g] = PCol{};
                    kc[g].type = PH_I32; kc[g].data = R.lanes[(size_t)sc.lane].rows;
                    kc[g].src = t; kc[g].src_col = sc.tcol;
                }
            }
            if (!by_row) PL_CHECK(string_codes(p, &R, nd.groups[g].e.col, nullptr, nullptr, &kc[g]));
            str_key[g] = true;
        } else if (kc[g].sdict) str_key[g] = true;   // a computed VARCHAR (already codes): reported as PH_STR, its strings are rows of kc[g].src
    }
    for (size_t a = 0; a < nd.aggs.size(); a++) {
        if (nd.aggs[a].kind == PH_A_COUNT_STAR) continue;
        PL_CHECK(eval_expr(p, &R, nd.aggs[a].arg, &ac[a]));
    }
    // everything positional: the sink reads position i of every key and argument
    Rel S = R;
    S.cols.clear();
    for (auto &c : kc) S.cols.push_back(c);
    for (size_t a = 0; a < nd.aggs.size(); a++) if (nd.aggs[a].kind != PH_A_COUNT_STAR) S.cols.push_back(ac[a]);
    std::vector<int> all;
    for (size_t c = 0; c < S.cols.size(); c++) if (S.cols[c].lane >= 0) all.push_back((int)c);
    PL_CHECK(positional(p, &S, all));
    std::vector<ph_col> keys, args(nd.aggs.size());
    std::vector<int32_t> key_types;
    size_t ci = nd.groups.size();
    packs->clear();
    {
        // The aggregate table holds up to four 8-byte key words. More group keys than that (Q18 groups by five columns) are
        // PACKED: two 4-byte keys without NULLs share one word, high * 2^32 + (low + 2^31) — exact in int64, and unpacked
        // again when the result is fetched.
        std::vector<int> word_of(nd.groups.size(), -1);
        std::vector<std::pair<int, int>> words;   // (key, partner or -1)
        size_t nwords = nd.groups.size();
        // (a dictionary code counts as narrow: widened to an INTEGER first, ph_widen_codes — Q10 groups by seven columns)
        auto narrow = [&](size_t g) { const PCol &c = S.cols[g]; return (c.type == PH_I32 || c.type == PH_DATE || c.type == PH_CODE8) && !c.validity; };
        std::vector<bool> used(nd.groups.size(), false);
        if (allow_pack)
            for (size_t g = 0; g < nd.groups.size() && nwords > 4; g++) {
                if (used[g] || !narrow(g)) continue;
                for (size_t h = g + 1; h < nd.groups.size(); h++)
                    if (!used[h] && narrow(h)) { used[g] = used[h] = true; words.push_back({(int)g, (int)h}); nwords--; break; }
            }
        if (nwords > 4) { set_error("ph_plan: %zu group keys do not fit the aggregate table's four key words", nd.groups.size()); return PH_EUNSUPPORTED; }
        // word order: the packed pairs and the single keys, in the order of their first key
        std::vector<std::pair<int, int>> order;
        for (size_t g = 0; g < nd.groups.size(); g++) {
            if (!used[g]) order.push_back({(int)g, -1});
            else for (auto &w : words) if (w.first == (int)g) order.push_back(w);
        }
        packs->assign(nd.groups.size(), KeyPack{0, 0});
        for (size_t w = 0; w < order.size(); w++) {
            const int g = order[w].first, h = order[w].second;
            if (h < 0) {
                const PCol &c = S.cols[(size_t)g];
                if (width_of(c.type) == 0) { set_error("ph_plan: VARCHAR group key that is not a dictionary-code column"); return PH_EUNSUPPORTED; }
                const int32_t *sx = nullptr;
                keys.push_back(col_view(S, c, &sx));
                key_types.push_back(c.type);
                (*packs)[(size_t)g] = KeyPack{(int)w, 0};
                continue;
            }
            ph_col two[2];
            for (int t = 0; t < 2; t++) {
                const int32_t *sx = nullptr;
                two[t] = col_view(S, S.cols[(size_t)(t ? h : g)], &sx);
                if (two[t].type == PH_CODE8) {   // codes as INTEGERs
                    void *w32 = nullptr;
                    PL_CHECK(palloc(p, std::max<int64_t>(S.n, 1) * 4, &w32));
                    if (S.n > 0) PL_CHECK(ph_widen_codes(ctx, &two[t], nullptr, S.n, (int32_t *)w32));
                    two[t].data = w32;
                }
                two[t].type = PH_I32;   // DATE days as the integers they are
            }
            const ph_rpn prog[7] = {{PH_X_COL, 0, 0, 0}, {PH_X_CONST, -1, 1ll << 32, 0}, {PH_X_MUL, -1, 0, 0}, {PH_X_COL, 1, 0, 0}, {PH_X_ADD, -1, 0, 0},
                                    {PH_X_CONST, -1, 1ll << 31, 0}, {PH_X_ADD, -1, 0, 0}};
            void *wv = nullptr;
            PL_CHECK(palloc(p, std::max<int64_t>(S.n, 1) * 8, &wv));
            if (S.n > 0) PL_CHECK(ph_expr_eval(ctx, two, 2, prog, 7, nullptr, S.n, (int64_t *)wv, nullptr));
            ph_col wc{};
            wc.type = PH_I64; wc.data = wv;
            keys.push_back(wc);
            key_types.push_back(PH_I64);
            (*packs)[(size_t)g] = KeyPack{(int)w, 1};
            (*packs)[(size_t)h] = KeyPack{(int)w, 2};
        }
        for (size_t g = 0; g < nd.groups.size(); g++) {
            const PCol &c = S.cols[g];
            // a VARCHAR key reports PH_STR: its int64 key values are ROW IDS of (table, col) whose strings are the keys (-1 = NULL)
            kinfo->push_back(KeyInfo{str_key[g] ? (int32_t)PH_STR : c.type, c.scale, c.src, c.src_col});
        }
    }
    std::vector<ph_aggspec> specs;
    uint32_t distinct_mask = 0;   // count(distinct x): COUNT in the main table, fed from a distinct side table (below)
    for (size_t a = 0; a < nd.aggs.size(); a++) {
        if (nd.aggs[a].kind == PH_A_COUNT_DISTINCT) distinct_mask |= 1u << a;
        specs.push_back(ph_aggspec{nd.aggs[a].kind == PH_A_COUNT_DISTINCT ? (int32_t)PH_A_COUNT : nd.aggs[a].kind, (int32_t)a});
        if (nd.aggs[a].kind == PH_A_COUNT_STAR) { ascale->push_back(0); atype->push_back(PH_I32); continue; }
        const PCol &c = S.cols[ci++];
        const int32_t *s = nullptr;
        args[a] = col_view(S, c, &s);
        ascale->push_back(c.scale);
        atype->push_back(nd.aggs[a].arg.e.kind == PH_PE_CASE && nd.aggs[a].arg.e.result_int ? PH_I32 : c.type);
    }
    if (keys.empty()) {   // one global group: a constant key (executor_aggr.go:37-48)
        void *zero = nullptr;
        PL_CHECK(palloc(p, std::max<int64_t>(S.n, 1) * 4, &zero));
        PL_CHECK(ph_dev_memset(ctx, zero, 0, std::max<int64_t>(S.n, 1) * 4));
        ph_col c{};
        c.type = PH_I32; c.data = zero;
        keys.push_back(c);
        key_types.push_back(PH_I32);
    }
    if (nullable) {
        nullable->clear();
        for (size_t g = 0; g < nd.groups.size(); g++) nullable->push_back(S.cols[g].validity != nullptr);
        for (size_t a = 0; a < nd.aggs.size(); a++) nullable->push_back(nd.aggs[a].kind != PH_A_COUNT_STAR && args[a].validity != nullptr);
    }
    for (size_t a = 0; a < nd.aggs.size(); a++) if (nd.aggs[a].kind == PH_A_COUNT_STAR) args[a] = keys[0];
    // expected groups: a key that is (a copy of) a wide integer table column is taken to be high-cardinality
    int64_t expected = 1024;
    if (!nd.groups.empty() && (S.cols[0].type == PH_I64 || S.cols[0].type == PH_I32) && S.cols[0].src) {
        const auto &sc = S.cols[0].src->cols[(size_t)S.cols[0].src_col];
        if (sc.has_range && sc.max - sc.min > 65536) expected = std::max<int64_t>(S.n / 2, 1024);
        // ... but a lone key has no more groups than its column has values (Q17's subquery: 60 M rows by l_partkey, 2 M values — the hint
        // decides between the bulk build's forms)
        if (sc.has_range && nd.groups.size() == 1) expected = std::min<int64_t>(expected, std::max<int64_t>(sc.max - sc.min + 1, 1024));
    }
    if ((*aggp)) { ph_agg_free((*aggp)); (*aggp) = nullptr; }
    PL_CHECK(ph_agg_create(ctx, (int32_t)key_types.size(), key_types.data(), (int32_t)specs.size(), specs.data(), expected, &(*aggp)));
    bool streamed = false;
    if (!distinct_mask && !p->conservative && !p->no_stream_agg && !nd.groups.empty() && S.cols[0].ordered && (*packs)[0].word == 0 && (*packs)[0].part == 0 && S.n > 0 && !getenv("PH_PLAN_NO_STREAM_AGG")) {
        int rc = ph_agg_sink_sorted((*aggp), keys.data(), args.data(), (int32_t)args.size(), S.n, 0);
        if (rc == PH_OK) streamed = true;
        else if (rc != PH_EUNSUPPORTED) return rc;
    }
    if (distinct_mask) {
        // DISTINCT aggregates (SinkDistinctGrouping / DistinctGrouping, aggregate_exec.go:76-105, 201-304): the raw rows create the groups
        // and feed the other aggregates (AddChunk's filter); every distinct aggregate has a side table keyed by (group keys, its argument)
        // whose rows — the distinct combinations — are then sunk into the main table with only that aggregate enabled.
        const uint32_t all_mask = nd.aggs.size() >= 32 ? 0xFFFFFFFFu : ((1u << nd.aggs.size()) - 1u);
        if (S.n > 0) PL_CHECK(ph_agg_sink_masked((*aggp), keys.data(), args.data(), (int32_t)args.size(), nullptr, S.n, 1, 0, all_mask & ~distinct_mask));
        const bool const_key = nd.groups.empty();
        for (size_t a = 0; a < nd.aggs.size() && S.n > 0; a++) {
            if (!((distinct_mask >> a) & 1)) continue;
            if (keys.size() + 1 > 4) { set_error("ph_plan: count(distinct) needs a key word beside the %zu group key words", keys.size()); return PH_EUNSUPPORTED; }
            std::vector<ph_col> dkeys = keys;
            std::vector<int32_t> dtypes = key_types;
            ph_col av = args[a];
            if (width_of(av.type) == 0) { set_error("ph_plan: count(distinct) over a VARCHAR argument"); return PH_EUNSUPPORTED; }
            dkeys.push_back(av);
            dtypes.push_back(av.type);
            const ph_aggspec star{PH_A_COUNT_STAR, 0};
            ph_agg *d = nullptr;
            PL_CHECK(ph_agg_create(ctx, (int32_t)dtypes.size(), dtypes.data(), 1, &star, std::max<int64_t>(S.n / 2, 1024), &d));
            p->inner_aggs.push_back(d);
            PL_CHECK(ph_agg_sink(d, dkeys.data(), dkeys.data(), 1, nullptr, S.n, 1, 0));
            int64_t nd_rows = 0;
            PL_CHECK(ph_agg_group_count(d, &nd_rows));
            // the side table's rows as columns again (ph_agg_keys_dev: RadixPartitionedHashTable.GetData for the distinct tables)
            std::vector<ph_col> rk(dkeys.size());
            for (size_t k = 0; k < dkeys.size(); k++) {
                void *kd = nullptr, *kv = nullptr;
                PL_CHECK(palloc(p, std::max<int64_t>(nd_rows, 1) * 8, &kd));
                if (dkeys[k].validity) PL_CHECK(palloc(p, (std::max<int64_t>(nd_rows, 1) + 7) / 8 + 64, &kv));
                int64_t n2 = 0;
                PL_CHECK(ph_agg_keys_dev(d, (int32_t)k, kd, (uint8_t *)kv, nd_rows, &n2));
                rk[k] = ph_col{};
                rk[k].type = dtypes[k]; rk[k].scale = dkeys[k].scale; rk[k].data = kd; rk[k].validity = (const uint8_t *)kv;
            }
            std::vector<ph_col> rargs(args.size(), rk.back());   // only aggregate a reads its argument: the distinct values
            (void)const_key;
            if (nd_rows > 0) PL_CHECK(ph_agg_sink_masked((*aggp), rk.data(), rargs.data(), (int32_t)rargs.size(), nullptr, nd_rows, 1, 0, 1u << a));
            note(p, "agg#%d: count(distinct) #%zu through a side table of %lld distinct (keys, argument) rows", idx, a, (long long)nd_rows);
        }
        streamed = false;
    } else if (!streamed && S.n > 0) PL_CHECK(ph_agg_sink((*aggp), keys.data(), args.data(), (int32_t)args.size(), nullptr, S.n, 1, 0));
    note(p, "agg#%d: %s over %lld rows, %zu keys, %zu aggregates", idx, streamed ? "streaming aggregate (rows ordered by the first key)" : "hash aggregate",
         (long long)S.n, nd.groups.size(), nd.aggs.size());
    return PH_OK;
}


// ---- the root: HashAggregate
int lower_agg(ph_plan *p) {
    const int idx = (int)p->nodes.size() - 1;
    const Node &nd = p->nodes[(size_t)idx];
    ph_ctx *ctx = p->ctx;
    p->keys.clear(); p->agg_scale.clear(); p->agg_arg_type.clear();
    Rel R;
    PL_CHECK(lower(p, nd.child[0], false, &R));
    p->root_replicated = p->root_disjoint = false;
    p->topk_off = false;
    if (multi(p)) {
        p->root_replicated = R.replicated;
        bool distinct = false;
        for (auto &a : nd.aggs) distinct |= a.kind == PH_A_COUNT_DISTINCT;
        // a top-k preselection, a HAVING and a DISTINCT aggregate are properties of WHOLE groups; plain sums, counts, minima and maxima merge at fetch
        if (!R.replicated && (p->topk_agg >= 0 || !p->having.empty() || distinct)) {
            if (nd.groups.empty()) { set_error("ph_plan: an ungrouped root aggregate with HAVING / DISTINCT across ranks"); return PH_EUNSUPPORTED; }
            const int wrc = whole_groups(p, idx, &R);
            if (wrc == PH_EUNSUPPORTED && p->having.empty() && !distinct) {
                // only the top-k asked for whole groups (a VARCHAR group key: no partition key): it is a preselection, not a semantic — the ranks'
                // partial states merge at fetch as without it and every group comes back (the same decision on every rank: it follows from the plan)
                p->topk_off = true;
                note(p, "agg#%d: groups cannot be made whole across ranks (no fixed-width group key): top-k preselection left out", idx);
            } else {
                PL_CHECK(wrc);
                p->root_disjoint = true;
            }
        }
    }

    // Agg <- Scan(filter): the fused scan kernels (ph_scan_plan) — the Q1 / Q6 shapes and everything scan_jit generates
    const Node &ch = p->nodes[(size_t)nd.child[0]];
    if (ch.kind == PH_PN_SCAN && R.single_identity() && !R.flags) {
        bool ok = true;
        std::vector<int32_t> gcols;
        ok = ok && R.complex.empty();
        // (the descriptor's column indexes are checked HERE: this lowering reads R.cols before eval_expr, which checks them, ever runs)
        auto in_range = [&](int32_t c) { return c >= 0 && (size_t)c < R.cols.size(); };
        for (auto &g : nd.groups) if (g.e.kind == PH_PE_COL && !in_range(g.e.col)) { set_error("ph_plan: group column %d out of range", g.e.col); return PH_EINVAL; }
        for (auto &a : nd.aggs) {
            if (a.kind == PH_A_COUNT_STAR) continue;
            if (a.arg.e.kind == PH_PE_COL && !in_range(a.arg.e.col)) { set_error("ph_plan: aggregate column %d out of range", a.arg.e.col); return PH_EINVAL; }
            if (a.arg.e.kind == PH_PE_DECIMAL)
                for (int i = 0; i < a.arg.e.nprog; i++)
                    if (a.arg.e.prog[i].op == PH_X_COL && !in_range(a.arg.e.prog[i].col)) { set_error("ph_plan: expression column %d out of range", a.arg.e.prog[i].col); return PH_EINVAL; }
        }
        for (auto &g : nd.groups) { ok = ok && g.e.kind == PH_PE_COL; if (ok) gcols.push_back(R.cols[(size_t)g.e.col].tcol); }
        std::vector<ph_aggexpr> ax(nd.aggs.size());
        for (size_t a = 0; a < nd.aggs.size() && ok; a++) {
            ax[a].kind = nd.aggs[a].kind;
            if (nd.aggs[a].kind == PH_A_COUNT_DISTINCT) { ok = false; break; }
            const ph_plan_expr &e = nd.aggs[a].arg.e;
            if (nd.aggs[a].kind == PH_A_COUNT_STAR) { ax[a].nprog = 0; continue; }
            if (e.kind == PH_PE_COL) { ax[a].nprog = 1; ax[a].prog[0] = ph_rpn{PH_X_COL, R.cols[(size_t)e.col].tcol, 0, 0}; }
            else if (e.kind == PH_PE_DECIMAL) {
                ax[a].nprog = e.nprog;
                for (int i = 0; i < e.nprog; i++) { ax[a].prog[i] = e.prog[i]; if (e.prog[i].op == PH_X_COL) ax[a].prog[i].col = R.cols[(size_t)e.prog[i].col].tcol; }
            } else ok = false;
        }
        if (ok) {
            std::vector<ph_pred> preds = R.pending;
            if (p->scan) { ph_scan_plan_free(p->scan); p->scan = nullptr; }
            int rc = ph_scan_plan_create(ctx, ch.table, preds.data(), (int32_t)preds.size(), gcols.data(), (int32_t)gcols.size(), ax.data(), (int32_t)ax.size(), &p->scan);
            if (rc == PH_OK) {
                PL_CHECK(ph_scan_plan_run(p->scan, 0, ch.table->nrows));
                for (auto &g : nd.groups) { const PCol &c = R.cols[(size_t)g.e.col]; p->keys.push_back(KeyInfo{c.type, c.scale, c.src, c.src_col}); }
                for (size_t a = 0; a < nd.aggs.size(); a++) {
                    int32_t t = PH_I32;
                    const ph_plan_expr &e = nd.aggs[a].arg.e;
                    if (nd.aggs[a].kind != PH_A_COUNT_STAR) t = e.kind == PH_PE_COL ? R.cols[(size_t)e.col].type : PH_DEC64;
                    p->agg_arg_type.push_back(t);
                }
                note(p, "agg#%d: fused scan plan (%s)", idx, ph_scan_plan_kind(p->scan));
                return PH_OK;
            }
            if (rc != PH_EUNSUPPORTED) return rc;
        }
    }

    return sink_into_agg(p, idx, R, true, &p->agg, &p->keys, &p->agg_scale, &p->agg_arg_type, &p->key_packs);
}

// a deferred PH_ECONSTRAINT surfaced: which claim broke? The streaming aggregate's message names it: only that form is retired (the plan's
// joins keep theirs); anything else (a sorted fill, a strict lookup) makes the plan conservative as before.
void retire_broken_claim(ph_plan *p) {
    const char *msg = ph_last_error();
    if (msg && strstr(msg, "ph_agg_sink_sorted") && !p->no_stream_agg) {
        p->no_stream_agg = true;
        p->explain += "  -> the rows were not ordered by the whole group-key tuple: hash aggregate from here on (the joins keep their forms)\n";
    } else {
        p->explain += "  -> a statistic did not hold (" + std::string(msg ? msg : "") + "): conservative forms from here on\n";
        p->conservative = true;
    }
}

// ---- ORDER BY <column> [DESC] LIMIT k above a join-rooted plan: only rows whose key is at least as good as the k-th best can be among the first k
// (ties kept; the host orders those few rows by the full key list). The k-th key: one device sort of the key column; then a threshold
// selection and a compaction — before any VARCHAR column is gathered and before 4 673 rows become chunks on the host (Q2 at SF10: 100 of them).
int rows_topk(ph_plan *p, Rel *R) {
    ph_ctx *ctx = p->ctx;
    const int c = p->rows_topk_col;
    const int64_t k = p->rows_topk_k, n = R->n;
    if (c < 0 || c >= (int)R->cols.size() || n <= std::max<int64_t>(4 * k, 256)) return PH_OK;
    const PCol &pc0 = R->cols[(size_t)c];
    const bool desc = p->rows_topk_desc != 0;
    // the orderings the selection has for the type (selectOperation): INTEGER and DATE all four, DECIMAL only '>' — and ORDER BY compares a
    // DECIMAL at two digits (sort_encoder.go:65-70), which is its value only up to scale 2
    const bool okt = pc0.type == PH_I32 || pc0.type == PH_DATE || (pc0.type == PH_DEC64 && desc && pc0.scale <= 2);
    if (!okt || pc0.sdict) return PH_OK;
    PL_CHECK(positional(p, R, {c}));
    const PCol &pc = R->cols[(size_t)c];
    if (pc.validity) return PH_OK;   // NULLs sort first whatever the direction: not a threshold's business
    const int32_t *s0 = nullptr;
    ph_col v = col_view(*R, pc, &s0);
    void *order = nullptr, *kth = nullptr, *keep = nullptr;
    PL_CHECK(palloc(p, n * 4, &order));
    PL_CHECK(palloc(p, 16, &kth));
    PL_CHECK(palloc(p, n * 4, &keep));
    const int32_t d = desc ? 1 : 0;
    int rc = ph_sort_rows(ctx, &v, &d, 1, nullptr, n, (int32_t *)order);
    if (rc == PH_EUNSUPPORTED) return PH_OK;
    PL_CHECK(rc);
    PL_CHECK(ph_gather(ctx, &v, (const int32_t *)order + (k - 1), 1, kth));
    int64_t kv = 0;
    if (width_of(pc.type) == 4) { int32_t x = 0; PL_CHECK(ctx->download(&x, kth, 4)); kv = x; } else PL_CHECK(ctx->download(&kv, kth, 8));
    ph_const kc{};
    kc.type = pc.type; kc.scale = pc.scale;
    int32_t op = desc ? PH_GE : PH_LE;
    kc.i = kv;
    if (pc.type == PH_DEC64) { op = PH_GT; kc.i = kv - 1; if (kv == INT64_MIN) return PH_OK; }   // unscaled integers: v > kth - 1  <=>  v >= kth
    int64_t m = 0;
    PL_CHECK(ph_filter_select(ctx, &v, n, op, &kc, nullptr, n, (int32_t *)keep, &m));
    if (m < k) { set_error("ph_plan: top-k preselection of the rows kept %lld of %lld rows for k = %lld", (long long)m, (long long)n, (long long)k); return PH_EHIP; }
    PL_CHECK(compact(p, R, (const int32_t *)keep, m));
    note(p, "rows: ORDER BY column %d %s LIMIT %lld — %lld of %lld rows are at least as good as the k-th (the host orders those)", c, desc ? "DESC" : "ASC", (long long)k,
         (long long)m, (long long)n);
    return PH_OK;
}

int run_once(ph_plan *p) {
    release_run(p, false);
    if (p->scan) { ph_scan_plan_free(p->scan); p->scan = nullptr; }
    p->explain.clear();
    note(p, "plan run (%s forms)", p->conservative ? "conservative" : "optimistic");
    PL_CHECK(ph_ctx_set_deferred_errors(p->ctx, multi(p) ? 2 : 1));   // across ranks deferred errors are HELD: the ranks agree on them at the end of the run
    int rc = PH_OK;
    if (p->rows_root) {
        Rel R;
        rc = lower(p, (int)p->nodes.size() - 1, false, &R);
        if (rc == PH_OK) rc = apply_pending(p, &R);
        if (rc == PH_OK && multi(p) && !R.replicated) rc = replicate_rel(p, &R, "the root relation's rows");   // every rank returns all rows
        if (rc == PH_OK && p->rows_topk_col >= 0) rc = rows_topk(p, &R);
        if (rc == PH_OK) p->rows_rel = std::make_shared<Rel>(R);
    } else rc = lower_agg(p);
    if (rc != PH_OK) { release_run(p, false); (void)ph_ctx_set_deferred_errors(p->ctx, 0); return rc; }
    p->ran = true;
    return PH_OK;
}

// the root relation's rows to the host: fixed-width columns positional and widened to 64 bits, VARCHAR columns (table columns behind
// row ids, or values computed in the plan behind their codes) gathered on the device into offsets + bytes
int fetch_rows_once(ph_plan *p, ph_rows_result **out) {
    Rel &R = *p->rows_rel;
    ph_ctx *ctx = p->ctx;
    const int64_t n = R.n;
    const int nc = (int)R.cols.size();
    std::vector<int> fixed;
    for (int c = 0; c < nc; c++) if (R.cols[(size_t)c].type != PH_STR) fixed.push_back(c);
    PL_CHECK(positional(p, &R, fixed));
    ph_rows_result *r = (ph_rows_result *)calloc(1, sizeof *r);
    r->nrows = n; r->ncols = nc;
    r->type = (int32_t *)calloc((size_t)std::max(nc, 1), 4);
    r->scale = (int32_t *)calloc((size_t)std::max(nc, 1), 4);
    r->values = (int64_t **)calloc((size_t)std::max(nc, 1), sizeof(int64_t *));
    r->offsets = (int32_t **)calloc((size_t)std::max(nc, 1), sizeof(int32_t *));
    r->bytes = (char **)calloc((size_t)std::max(nc, 1), sizeof(char *));
    auto fail = [&](int rc) { ph_rows_result_free(r); return rc; };
    for (int c = 0; c < nc; c++) {
        const PCol &pc = R.cols[(size_t)c];
        r->type[c] = pc.sdict ? (int32_t)PH_STR : pc.type;
        r->scale[c] = pc.scale;
        if (pc.type == PH_STR || pc.sdict) {
            // strings: ph_substring(1, whole) over the column with the row ids (or the codes of a computed value) as its selection
            ph_col v{};
            const int32_t *sel = nullptr;
            if (pc.sdict) { v = table_view(pc.src, pc.src_col); sel = (const int32_t *)pc.data; }
            else {
                if (pc.lane < 0) { set_error("ph_plan_fetch_rows: VARCHAR column %d is not a table column", c); return fail(PH_EUNSUPPORTED); }
                if (R.lanes[(size_t)pc.lane].nullable) { set_error("ph_plan_fetch_rows: VARCHAR column %d of the NULL-able side of a LEFT join", c); return fail(PH_EUNSUPPORTED); }
                v = table_view(R.lanes[(size_t)pc.lane].t, pc.tcol);
                sel = R.lanes[(size_t)pc.lane].rows;
            }
            r->offsets[c] = (int32_t *)calloc((size_t)n + 1, 4);
            if (n == 0) { r->bytes[c] = (char *)calloc(1, 1); continue; }
            int64_t cap = v.aux_bytes + 64, nbytes = 0;
            void *off = nullptr, *bytes = nullptr;
            int rc = PH_OK;
            for (int attempt = 0; attempt < 2; attempt++) {
                if ((rc = palloc(p, (n + 1) * 4, &off)) != PH_OK || (rc = palloc(p, cap + 64, &bytes)) != PH_OK) return fail(rc);
                rc = ph_substring(ctx, &v, 1, INT64_MAX, sel, n, (int32_t *)off, (uint8_t *)bytes, cap, &nbytes);
                if (rc == PH_ECAPACITY && attempt == 0 && nbytes > cap) { cap = nbytes; continue; }   // (repeated rows below a join)
                break;
            }
            if (rc != PH_OK) return fail(rc);
            r->bytes[c] = (char *)calloc((size_t)std::max<int64_t>(nbytes, 1), 1);
            if ((rc = ctx->download(r->offsets[c], off, (n + 1) * 4)) != PH_OK) return fail(rc);
            if (nbytes > 0 && (rc = ctx->download(r->bytes[c], bytes, nbytes)) != PH_OK) return fail(rc);
            continue;
        }
        if (pc.validity) { set_error("ph_plan_fetch_rows: NULL-able column %d", c); return fail(PH_EUNSUPPORTED); }
        r->values[c] = (int64_t *)calloc((size_t)std::max<int64_t>(n, 1), 8);
        if (n == 0) continue;
        const int w = width_of(pc.type);
        std::vector<unsigned char> raw((size_t)n * (size_t)w);
        int rc = ctx->download(raw.data(), pc.data, n * w);
        if (rc != PH_OK) return fail(rc);
        for (int64_t i = 0; i < n; i++) {
            if (w == 8) r->values[c][i] = reinterpret_cast<const int64_t *>(raw.data())[i];
            else if (w == 4) r->values[c][i] = reinterpret_cast<const int32_t *>(raw.data())[i];
            else r->values[c][i] = raw[(size_t)i];
        }
    }
    *out = r;
    return PH_OK;
}

ph_agg_result *new_result(int64_t ng, int nkeys, int naggs) {
    const size_t g = (size_t)std::max<int64_t>(ng, 1), nk = (size_t)std::max(nkeys, 1), na = (size_t)std::max(naggs, 1);
    ph_agg_result *r = (ph_agg_result *)calloc(1, sizeof *r);
    r->ngroups = ng; r->nkeys = nkeys; r->naggs = naggs;
    r->first_row = (int64_t *)calloc(g, 8);
    r->keys = (int64_t *)calloc(g * nk, 8);
    r->sum_lo = (uint64_t *)calloc(g * na, 8);
    r->sum_hi = (int64_t *)calloc(g * na, 8);
    r->count = (uint64_t *)calloc(g * na, 8);
    r->scale = (int32_t *)calloc(na, 4);
    return r;
}

int fetch_once(ph_plan *p, ph_agg_result **out) {
    const Node &nd = p->nodes.back();
    if (p->scan) return ph_scan_plan_fetch(p->scan, out);
    const int nkeys = (int)nd.groups.size(), naggs = (int)nd.aggs.size();
    const int nk = std::max(nkeys, 1);
    int nw = 1;   // key words of the aggregate table (packed keys share one)
    for (auto &kp : p->key_packs) nw = std::max(nw, kp.word + 1);
    int64_t room = p->topk_agg >= 0 ? 4096 : 1024, ng = 0;
    bool skip_device_forms = false;
    for (int attempt = 0; attempt < 3; attempt++) {
        std::vector<int64_t> first((size_t)room), keys((size_t)room * nw), hi((size_t)room * std::max(naggs, 1));
        std::vector<uint8_t> knull((size_t)room * nw);
        std::vector<uint64_t> lo((size_t)room * std::max(naggs, 1)), cnt((size_t)room * std::max(naggs, 1));
        int rc;
        p->having_applied = false;
        if (p->topk_agg >= 0 && !skip_device_forms && !p->topk_off) rc = ph_agg_topk(p->agg, p->topk_agg, p->topk_desc, p->topk_k, room, &ng, first.data(), keys.data(), knull.data(), lo.data(), hi.data(), cnt.data());
        else if (!p->having.empty() && !skip_device_forms) {
            std::vector<int32_t> ai, op, sc;
            std::vector<ph_const> ks;
            for (auto &h : p->having) {
                const int a = h.col - nkeys;
                ai.push_back(a); op.push_back(h.op); ks.push_back(h.k);
                const int kind = nd.aggs[(size_t)a].kind;
                sc.push_back(kind == PH_A_COUNT || kind == PH_A_COUNT_STAR ? 0 : p->agg_scale[(size_t)a]);
            }
            rc = ph_agg_fetch_where(p->agg, (int32_t)ai.size(), ai.data(), op.data(), ks.data(), sc.data(), room, &ng, first.data(), keys.data(), knull.data(),
                                    lo.data(), hi.data(), cnt.data());
            p->having_applied = rc == PH_OK;
        } else rc = ph_agg_fetch(p->agg, room, &ng, first.data(), keys.data(), knull.data(), lo.data(), hi.data(), cnt.data());
        // the preselection and the device HAVING compare the sums as int64 values: a sum beyond that range is no error of the query —
        // every group comes back with its 128-bit sums and the host filters / sorts (ph_plan_having_applied tells it so)
        if (rc == PH_EOVERFLOW && !skip_device_forms) { skip_device_forms = true; note(p, "  fetch: a sum exceeds int64 — HAVING / top-k left to the host"); attempt--; continue; }
        if (rc == PH_ECAPACITY && ng > room) { room = ng; continue; }
        PL_CHECK(rc);
        ph_agg_result *r = new_result(ng, nkeys, naggs);
        bool any_null = false;
        for (size_t i = 0; i < (size_t)ng * (size_t)nw; i++) any_null |= knull[i] != 0;
        if (any_null) r->key_null = (uint8_t *)calloc((size_t)std::max<int64_t>(ng, 1) * (size_t)nk, 1);
        for (int64_t g = 0; g < ng; g++) {
            r->first_row[g] = first[(size_t)g];
            for (int k = 0; k < nkeys; k++) {
                const KeyPack kp = p->key_packs[(size_t)k];
                const int64_t w = keys[(size_t)(g * nw + kp.word)];
                r->keys[g * nk + k] = kp.part == 0 ? w : kp.part == 1 ? (w >> 32) : (int64_t)(uint32_t)w - (1ll << 31);
                if (any_null && kp.part == 0 && knull[(size_t)(g * nw + kp.word)]) { r->key_null[g * nk + k] = 1; r->keys[g * nk + k] = 0; }
            }
            for (int a = 0; a < naggs; a++) {
                r->sum_lo[g * naggs + a] = lo[(size_t)(g * naggs + a)];
                r->sum_hi[g * naggs + a] = hi[(size_t)(g * naggs + a)];
                r->count[g * naggs + a] = cnt[(size_t)(g * naggs + a)];
            }
        }
        for (int a = 0; a < naggs; a++) r->scale[a] = p->agg_scale[(size_t)a];
        *out = r;
        return PH_OK;
    }
    set_error("ph_plan_fetch: group count kept growing");
    return PH_ECAPACITY;
}

}  // namespace

extern "C" int ph_plan_create(ph_ctx *ctx, const ph_plan_node *nodes, int32_t nnodes, ph_plan **out) {
    PH_REQUIRE(ctx && nodes && out && nnodes >= 2 && nnodes <= 64, "ph_plan_create: bad arguments (2..64 nodes)");
    const int32_t rk = nodes[nnodes - 1].kind;
    PH_REQUIRE(rk == PH_PN_AGG || rk == PH_PN_JOIN || rk == PH_PN_FILTER || rk == PH_PN_PROJECT,
               "ph_plan_create: the last node is the root: a PH_PN_AGG (ph_plan_fetch), or a join / filter / project whose rows ph_plan_fetch_rows returns");
    ph_plan *p = new ph_plan();
    p->ctx = ctx;
    p->rows_root = rk != PH_PN_AGG;
    auto fail = [&](int rc) { delete p; return rc; };
    for (int32_t i = 0; i < nnodes; i++) {
        const ph_plan_node &s = nodes[i];
        Node n;
        n.kind = s.kind;
        n.child[0] = s.child[0]; n.child[1] = s.child[1];
        for (int c = 0; c < 2; c++)
            if (n.child[c] >= i) { set_error("ph_plan_create: node %d: children must precede their parent", i); return fail(PH_EINVAL); }
        switch (s.kind) {
        case PH_PN_SCAN:
            if (!s.table || s.ncols < 0 || (s.ncols && !s.cols) || s.npreds < 0 || (s.npreds && !s.preds)) { set_error("ph_plan_create: node %d: bad scan", i); return fail(PH_EINVAL); }
            n.table = s.table;
            n.cols.assign(s.cols, s.cols + s.ncols);
            break;
        case PH_PN_FILTER:
            if (s.child[0] < 0 || s.npreds < 0 || (s.npreds && !s.preds)) { set_error("ph_plan_create: node %d: bad filter", i); return fail(PH_EINVAL); }
            break;
        case PH_PN_JOIN:
            if (s.child[0] < 0 || s.child[1] < 0 || s.nkeys < 1 || s.nkeys > 4 || !s.probe_keys || !s.build_keys || s.nout < 0 || (s.nout && !s.out) ||
                s.join_type < PH_JT_INNER || s.join_type > PH_JT_LEFT) { set_error("ph_plan_create: node %d: bad join", i); return fail(PH_EINVAL); }
            n.join_type = s.join_type;
            n.pkeys.assign(s.probe_keys, s.probe_keys + s.nkeys);
            n.bkeys.assign(s.build_keys, s.build_keys + s.nkeys);
            n.out.assign(s.out, s.out + s.nout);
            break;
        case PH_PN_PROJECT:
            if (s.child[0] < 0 || s.nexprs < 1 || !s.exprs) { set_error("ph_plan_create: node %d: bad project", i); return fail(PH_EINVAL); }
            for (int32_t k = 0; k < s.nexprs; k++) n.exprs.push_back(Expr{s.exprs[k], copy_bools(s.exprs[k].when, s.exprs[k].kind == PH_PE_CASE ? s.exprs[k].nwhen : 0)});
            break;
        case PH_PN_AGG:
            if (s.child[0] < 0 || s.naggs < 1 || !s.aggs || s.ngroups < 0 || s.ngroups > 8 || (s.ngroups && !s.groups) || s.naggs > 16) {
                set_error("ph_plan_create: node %d: bad aggregate (<= 8 group expressions, 1..16 aggregates)", i);
                return fail(PH_EINVAL);
            }
            for (int32_t k = 0; k < s.ngroups; k++) n.groups.push_back(Expr{s.groups[k], copy_bools(s.groups[k].when, s.groups[k].kind == PH_PE_CASE ? s.groups[k].nwhen : 0)});
            for (int32_t k = 0; k < s.naggs; k++)
                n.aggs.push_back(AggDesc{s.aggs[k].kind, Expr{s.aggs[k].arg, copy_bools(s.aggs[k].arg.when, s.aggs[k].kind != PH_A_COUNT_STAR && s.aggs[k].arg.kind == PH_PE_CASE ? s.aggs[k].arg.nwhen : 0)}});
            break;
        default:
            set_error("ph_plan_create: node %d: unknown kind %d", i, s.kind);
            return fail(PH_EINVAL);
        }
        if (s.kind == PH_PN_SCAN || s.kind == PH_PN_FILTER) {
            n.pred_strs.resize((size_t)s.npreds);
            for (int32_t k = 0; k < s.npreds; k++) {
                n.preds.push_back(s.preds[k]);
                if (s.preds[k].k.s) n.pred_strs[(size_t)k] = s.preds[k].k.s;
                n.preds.back().k.s = nullptr;
            }
        }
        if (s.kind == PH_PN_SCAN || s.kind == PH_PN_FILTER || s.kind == PH_PN_JOIN) {   // (a join's tree: its residual condition over [probe | build] columns)
            if (s.nbools < 0 || (s.nbools && !s.bools)) { set_error("ph_plan_create: node %d: bad boolean tree", i); return fail(PH_EINVAL); }
            n.bools = copy_bools(s.bools, s.nbools);
            // children FOLLOW their parent in the flat array: a node that named itself or an earlier node would make eval_bool recurse forever
            for (size_t bi = 0; bi < n.bools.nodes.size(); bi++) {
                const ph_bool &b = n.bools.nodes[bi];
                if ((b.kind == PH_B_AND || b.kind == PH_B_OR) && (b.first_child <= (int32_t)bi || b.nchildren < 1 || b.first_child + b.nchildren > s.nbools)) {
                    set_error("ph_plan_create: node %d: boolean tree children out of range (they must follow their parent)", i);
                    return fail(PH_EINVAL);
                }
            }
        }
        auto check_expr = [&](Expr &x) {
            x.when.fix();
            x.e.when = nullptr;
            if (x.e.nprog < 0 || x.e.nprog > 12 || (x.e.kind == PH_PE_CASE && (x.e.nelse < 1 || x.e.nelse > 12 || x.when.empty()))) return false;
            for (size_t bi = 0; bi < x.when.nodes.size(); bi++) {
                const ph_bool &b = x.when.nodes[bi];
                if ((b.kind == PH_B_AND || b.kind == PH_B_OR) && (b.first_child <= (int32_t)bi || b.nchildren < 1 || b.first_child + b.nchildren > (int)x.when.nodes.size())) return false;
            }
            return true;
        };
        bool okx = true;
        for (auto &e : n.exprs) okx = okx && check_expr(e);
        for (auto &e : n.groups) okx = okx && check_expr(e);
        for (auto &a : n.aggs) if (a.kind != PH_A_COUNT_STAR) okx = okx && check_expr(a.arg);
        if (!okx) { set_error("ph_plan_create: node %d: malformed expression (program length, CASE without WHEN / ELSE, boolean tree children)", i); return fail(PH_EINVAL); }
        p->nodes.push_back(std::move(n));
    }
    p->parents.assign(p->nodes.size(), 0);
    for (auto &n : p->nodes)
        for (int c = 0; c < 2; c++) if (n.child[c] >= 0) p->parents[(size_t)n.child[c]]++;
    *out = p;
    return PH_OK;
}

extern "C" int ph_plan_set_topk(ph_plan *p, int32_t agg_index, int32_t descending, int64_t k) {
    PH_REQUIRE(p && !p->rows_root && k > 0 && agg_index >= 0 && agg_index < (int32_t)p->nodes.back().aggs.size(), "ph_plan_set_topk: bad arguments");
    // HAVING runs inside the aggregate's output phase, BEFORE Order and Limit (executor_aggr.go:143-263): a preselection of the best k
    // groups taken first would drop groups the HAVING keeps out of reach. The two are exclusive in both directions.
    if (!p->having.empty()) { set_error("ph_plan_set_topk: the plan has a HAVING (the host sorts the surviving groups)"); return PH_EUNSUPPORTED; }
    p->topk_agg = agg_index;
    p->topk_desc = descending ? 1 : 0;
    p->topk_k = k;
    return PH_OK;
}

extern "C" int ph_plan_set_rows_topk(ph_plan *p, int32_t col, int32_t descending, int64_t k) {
    PH_REQUIRE(p && p->rows_root && k > 0 && col >= 0, "ph_plan_set_rows_topk: a join-rooted plan, a column of its rows, k > 0");
    p->rows_topk_col = col;
    p->rows_topk_desc = descending ? 1 : 0;
    p->rows_topk_k = k;
    return PH_OK;
}

extern "C" int ph_plan_set_having(ph_plan *p, int32_t nconj, const ph_pred *conj) {
    PH_REQUIRE(p && !p->rows_root && nconj >= 0 && (nconj == 0 || conj), "ph_plan_set_having: bad arguments");
    const Node &root = p->nodes.back();
    const int nkeys = (int)root.groups.size(), naggs = (int)root.aggs.size();
    if (p->topk_agg >= 0) { set_error("ph_plan_set_having: the plan has a top-k preselection"); return PH_EUNSUPPORTED; }
    if (root.child[0] >= 0 && p->nodes[(size_t)root.child[0]].kind == PH_PN_SCAN) { set_error("ph_plan_set_having: Agg <- Scan runs as a fused scan (few groups: the host filters them)"); return PH_EUNSUPPORTED; }
    for (int32_t c = 0; c < nconj; c++) {
        const int a = conj[c].col - nkeys;
        if (a < 0 || a >= naggs) { set_error("ph_plan_set_having: conjunct %d is not over an aggregate column", c); return PH_EUNSUPPORTED; }
        if (root.aggs[(size_t)a].kind == PH_A_AVG) { set_error("ph_plan_set_having: AVG is a quotient the host owns"); return PH_EUNSUPPORTED; }
        const int kt = conj[c].k.type;
        if (kt != PH_I32 && kt != PH_DEC64 && kt != PH_F32) { set_error("ph_plan_set_having: constant type %d", kt); return PH_EUNSUPPORTED; }
    }
    p->having.assign(conj, conj + nconj);
    return PH_OK;
}

extern "C" int32_t ph_plan_having_applied(const ph_plan *p) { return p && p->having_applied ? 1 : 0; }

// multi-rank run: deferred errors were held; now the ranks agree — 0 all fine, 1 a statistic did not hold somewhere (every rank reruns in the
// conservative forms, together), 2 a rank failed otherwise (every rank reports failure)
static int run_agreed(ph_plan *p) {
    for (int attempt = 0; attempt < 2; attempt++) {
        int rc = run_once(p);
        int64_t state = 0;
        if (rc == PH_OK) {
            const int d = ph_ctx_check_deferred(p->ctx);
            state = d == PH_OK ? 0 : d == PH_ECONSTRAINT ? 1 : 2;
        } else state = rc == PH_ECONSTRAINT ? 1 : 2;
        const std::string local_err = state ? ph_last_error() : "";
        int64_t agreed = state;
        // (a rank whose lowering failed before the others' next collective leaves them waiting there: only data-independent failures, which every
        // rank meets at the same node, and the deferred ones, which wait until here, are expected)
        if (ph_comm_allreduce_i64(p->comm, &agreed, 1, PH_RED_MAX) != PH_OK) return PH_EHIP;
        if (agreed == 0) return PH_OK;
        if (agreed == 2 || p->conservative) {
            release_run(p, false);
            (void)ph_ctx_set_deferred_errors(p->ctx, 0);
            set_error("ph_plan_run (rank %d of %d): %s", ph_comm_rank(p->comm), ph_comm_nranks(p->comm), local_err.empty() ? "another rank failed" : local_err.c_str());
            return state == 1 ? PH_ECONSTRAINT : rc != PH_OK ? rc : PH_EHIP;
        }
        p->explain += "  -> a statistic did not hold on some rank: every rank runs the plan again in its conservative forms\n";
        p->conservative = true;
    }
    return PH_ECONSTRAINT;
}

extern "C" int ph_plan_set_comm(ph_plan *p, ph_comm *comm) {
    PH_REQUIRE(p != nullptr, "ph_plan_set_comm: plan is NULL");
    p->comm = comm;
    return PH_OK;
}

extern "C" int ph_plan_set_broadcast_rows(ph_plan *p, int64_t rows) {
    PH_REQUIRE(p != nullptr && rows >= 0, "ph_plan_set_broadcast_rows: bad arguments");
    p->bcast_rows = rows;
    return PH_OK;
}

extern "C" int ph_plan_run(ph_plan *p) {
    PH_REQUIRE(p != nullptr, "ph_plan_run: plan is NULL");
    PH_HIP(hipSetDevice(p->ctx->device));
    if (multi(p)) return run_agreed(p);
    int rc = run_once(p);
    for (int again = 0; again < 2 && rc == PH_ECONSTRAINT && !p->conservative; again++) {   // a broken claim surfaced at a count read-back in the middle of the run
        retire_broken_claim(p);
        const std::string first = p->explain;
        rc = run_once(p);
        p->explain = first + p->explain;
    }
    return rc;
}

extern "C" void ph_rows_result_free(ph_rows_result *r) {
    if (!r) return;
    for (int c = 0; c < r->ncols; c++) {
        if (r->values) free(r->values[c]);
        if (r->offsets) free(r->offsets[c]);
        if (r->bytes) free(r->bytes[c]);
    }
    free(r->type); free(r->scale); free(r->values); free(r->offsets); free(r->bytes);
    free(r);
}

extern "C" int ph_plan_fetch_rows(ph_plan *p, ph_rows_result **out) {
    PH_REQUIRE(p && out, "ph_plan_fetch_rows: bad arguments");
    PH_REQUIRE(p->rows_root, "ph_plan_fetch_rows: the plan's root is an aggregate (ph_plan_fetch)");
    PH_REQUIRE(p->ran && p->rows_rel, "ph_plan_fetch_rows: ph_plan_run first");
    int rc = fetch_rows_once(p, out);
    for (int again = 0; again < 2 && rc == PH_ECONSTRAINT && !p->conservative; again++) {
        retire_broken_claim(p);
        const std::string first = p->explain;
        rc = run_once(p);
        p->explain = first + p->explain;
        if (rc == PH_OK) rc = fetch_rows_once(p, out);
    }
    release_run(p, true);
    (void)ph_ctx_set_deferred_errors(p->ctx, 0);
    p->ran = false;
    return rc;
}

// ---- multi-rank fetch: every rank's groups to every rank (one all-gather of the serialised results), then — unless the ranks' groups are disjoint —
// the partial states of equal keys merged: 128-bit sums and counts add, minima / maxima compare (a state no input reached does not take part).
namespace {
void put64(std::vector<unsigned char> *b, uint64_t v) { for (int i = 0; i < 8; i++) b->push_back((unsigned char)(v >> (8 * i))); }
uint64_t get64(const unsigned char *&q) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)q[i] << (8 * i); q += 8; return v; }

int merge_over_ranks(ph_plan *p, ph_agg_result *mine, ph_agg_result **out) {
    const Node &nd = p->nodes.back();
    const int nk = std::max(mine->nkeys, 1), na = mine->naggs, nkeys = mine->nkeys;
    ph_ctx *ctx = p->ctx;
    // A VARCHAR key is a row id of a column: the same on every rank only for a replicated (or broadcast) table. Keys that name rows of a rank's OWN
    // table (a sharded table's column, a substring computed in the plan) travel as their STRINGS and are merged by them; the merged distinct
    // strings then form a small table the plan owns, and the key values name ITS rows (ph_plan_key_info reports it).
    std::vector<bool> by_string((size_t)nk, false);
    std::vector<std::vector<std::string>> my_strings((size_t)nk);
    for (int k = 0; k < nkeys; k++) {
        const KeyInfo &ki = p->keys[(size_t)k];
        if (ki.type != PH_STR || (ki.table && ki.table->replicated)) continue;
        if (!ki.table) { set_error("ph_plan_fetch: VARCHAR group key %d has no table", k); return PH_EUNSUPPORTED; }
        by_string[(size_t)k] = true;
        std::vector<int64_t> rows;
        for (int64_t g = 0; g < mine->ngroups; g++) if (!(mine->key_null && mine->key_null[g * nk + k])) rows.push_back(mine->keys[g * nk + k]);
        std::vector<int32_t> off(rows.size() + 1, 0);
        int64_t cap = 1 << 20;
        std::vector<char> bytes;
        int rc = PH_OK;
        for (int attempt = 0; attempt < 8; attempt++) {
            bytes.assign((size_t)cap, 0);
            rc = rows.empty() ? PH_OK : ph_table_strings(ctx, ki.table, ki.col, rows.data(), (int64_t)rows.size(), off.data(), bytes.data(), cap);
            if (rc != PH_ECAPACITY) break;
            cap *= 8;
        }
        PL_CHECK(rc);
        size_t at = 0;
        for (int64_t g = 0; g < mine->ngroups; g++) {
            if (mine->key_null && mine->key_null[g * nk + k]) { my_strings[(size_t)k].push_back(std::string()); continue; }
            my_strings[(size_t)k].emplace_back(bytes.data() + off[at], (size_t)(off[at + 1] - off[at]));
            at++;
        }
    }
    std::vector<unsigned char> blob;
    put64(&blob, (uint64_t)mine->ngroups);
    for (int64_t g = 0; g < mine->ngroups; g++) {
        for (int k = 0; k < nk; k++) {
            if (by_string[(size_t)k]) {
                const std::string &sv = my_strings[(size_t)k][(size_t)g];
                put64(&blob, (uint64_t)sv.size());
                blob.insert(blob.end(), sv.begin(), sv.end());
            } else put64(&blob, (uint64_t)mine->keys[g * nk + k]);
            blob.push_back(mine->key_null ? mine->key_null[g * nk + k] : 0);
        }
        for (int a = 0; a < na; a++) { put64(&blob, mine->sum_lo[g * na + a]); put64(&blob, (uint64_t)mine->sum_hi[g * na + a]); put64(&blob, mine->count[g * na + a]); }
    }
    std::vector<std::map<std::string, int64_t>> interned((size_t)nk);
    std::vector<std::vector<std::string>> merged_strings((size_t)nk);
    const int n = ph_comm_nranks(p->comm);
    void *send = nullptr, *recv = nullptr;
    PL_CHECK(ctx->pool_alloc((int64_t)blob.size() + 64, &send));
    int rc = ph_dev_upload(ctx, send, blob.data(), (int64_t)blob.size());
    std::vector<int64_t> counts((size_t)n);
    if (rc == PH_OK) rc = ph_comm_allgather_rows_alloc(p->comm, send, (int64_t)blob.size(), 1, &recv, counts.data());
    int64_t total = 0;
    for (int r = 0; r < n; r++) total += counts[(size_t)r];
    std::vector<unsigned char> all((size_t)std::max<int64_t>(total, 1));
    if (rc == PH_OK && total > 0) rc = ctx->download_plain(all.data(), recv, total);
    ctx->pool_release(send);
    if (recv) ph_dev_free(ctx, recv);
    PL_CHECK(rc);
    struct G { std::vector<int64_t> key; std::vector<uint8_t> kn; std::vector<uint64_t> lo, cnt; std::vector<int64_t> hi; };
    std::vector<G> groups;
    std::map<std::pair<std::vector<int64_t>, std::vector<uint8_t>>, size_t> index;
    const unsigned char *q = all.data();
    for (int r = 0; r < n; r++) {
        const unsigned char *end = q + counts[(size_t)r];
        const int64_t ng = (int64_t)get64(q);
        for (int64_t g = 0; g < ng; g++) {
            G x;
            x.key.resize((size_t)nk); x.kn.resize((size_t)nk); x.lo.resize((size_t)na); x.hi.resize((size_t)na); x.cnt.resize((size_t)na);
            for (int k = 0; k < nk; k++) {
                if (by_string[(size_t)k]) {
                    const uint64_t len = get64(q);
                    std::string sv((const char *)q, (size_t)len);
                    q += len;
                    auto it = interned[(size_t)k].find(sv);
                    if (it == interned[(size_t)k].end()) { it = interned[(size_t)k].emplace(sv, (int64_t)merged_strings[(size_t)k].size()).first; merged_strings[(size_t)k].push_back(sv); }
                    x.key[(size_t)k] = it->second;
                } else x.key[(size_t)k] = (int64_t)get64(q);
                x.kn[(size_t)k] = *q++;
            }
            for (int a = 0; a < na; a++) { x.lo[(size_t)a] = get64(q); x.hi[(size_t)a] = (int64_t)get64(q); x.cnt[(size_t)a] = get64(q); }
            if (p->root_disjoint) { groups.push_back(std::move(x)); continue; }
            auto key = std::make_pair(x.key, x.kn);
            auto it = index.find(key);
            if (it == index.end()) { index[key] = groups.size(); groups.push_back(std::move(x)); continue; }
            G &t = groups[it->second];
            for (int a = 0; a < na; a++) {
                const int kind = nd.aggs[(size_t)a].kind;
                if (kind == PH_A_MIN || kind == PH_A_MAX) {
                    if (x.cnt[(size_t)a] == 0) continue;
                    const int64_t v = (int64_t)x.lo[(size_t)a], w = (int64_t)t.lo[(size_t)a];
                    if (t.cnt[(size_t)a] == 0 || (kind == PH_A_MIN ? v < w : v > w)) { t.lo[(size_t)a] = x.lo[(size_t)a]; t.hi[(size_t)a] = x.hi[(size_t)a]; }
                    t.cnt[(size_t)a] += x.cnt[(size_t)a];
                } else {
                    const unsigned __int128 s = (((unsigned __int128)(uint64_t)t.hi[(size_t)a]) << 64 | t.lo[(size_t)a]) + (((unsigned __int128)(uint64_t)x.hi[(size_t)a]) << 64 | x.lo[(size_t)a]);
                    t.lo[(size_t)a] = (uint64_t)s; t.hi[(size_t)a] = (int64_t)(uint64_t)(s >> 64);
                    t.cnt[(size_t)a] += x.cnt[(size_t)a];
                }
            }
        }
        q = end;
    }
    // the merged strings of every by-string key as a one-column table of the plan: the key values are its rows
    for (int k = 0; k < nkeys; k++) {
        if (!by_string[(size_t)k]) continue;
        const auto &ms = merged_strings[(size_t)k];
        std::vector<int32_t> off(ms.size() + 1, 0);
        std::string bytes;
        for (size_t i = 0; i < ms.size(); i++) { bytes += ms[i]; off[i + 1] = (int32_t)bytes.size(); }
        void *od = nullptr, *bd = nullptr;
        PL_CHECK(ctx->pool_alloc((int64_t)off.size() * 4 + 64, &od));
        p->computed_bufs.push_back(od);
        PL_CHECK(ctx->pool_alloc((int64_t)bytes.size() + 64, &bd));
        p->computed_bufs.push_back(bd);
        PL_CHECK(ph_dev_upload(ctx, od, off.data(), (int64_t)off.size() * 4));
        if (!bytes.empty()) PL_CHECK(ph_dev_upload(ctx, bd, bytes.data(), (int64_t)bytes.size()));
        ph_table *vt = new ph_table();
        p->computed.push_back(vt);
        vt->ctx = ctx; vt->nrows = (int64_t)ms.size(); vt->replicated = true;
        vt->cols.resize(1);
        vt->cols[0].type = PH_STR; vt->cols[0].data = od; vt->cols[0].aux = bd; vt->cols[0].aux_bytes = (int64_t)bytes.size();
        p->keys[(size_t)k].table = vt;
        p->keys[(size_t)k].col = 0;
    }
    ph_agg_result *r = new_result((int64_t)groups.size(), nkeys, na);
    bool any_null = false;
    for (auto &g : groups) for (auto v : g.kn) any_null |= v != 0;
    if (any_null) r->key_null = (uint8_t *)calloc(std::max<size_t>(groups.size(), 1) * (size_t)nk, 1);
    for (size_t g = 0; g < groups.size(); g++) {
        r->first_row[g] = (int64_t)g;
        for (int k = 0; k < nkeys; k++) { r->keys[g * (size_t)nk + (size_t)k] = groups[g].key[(size_t)k]; if (any_null) r->key_null[g * (size_t)nk + (size_t)k] = groups[g].kn[(size_t)k]; }
        for (int a = 0; a < na; a++) { r->sum_lo[g * (size_t)na + (size_t)a] = groups[g].lo[(size_t)a]; r->sum_hi[g * (size_t)na + (size_t)a] = groups[g].hi[(size_t)a]; r->count[g * (size_t)na + (size_t)a] = groups[g].cnt[(size_t)a]; }
    }
    for (int a = 0; a < na; a++) r->scale[a] = mine->scale[a];
    *out = r;
    return PH_OK;
}
}  // namespace

extern "C" int ph_plan_fetch(ph_plan *p, ph_agg_result **out) {
    PH_REQUIRE(p && out, "ph_plan_fetch: bad arguments");
    PH_REQUIRE(!p->rows_root, "ph_plan_fetch: the plan's root is no aggregate (ph_plan_fetch_rows)");
    PH_REQUIRE(p->ran, "ph_plan_fetch: ph_plan_run first");
    if (multi(p)) {   // (broken statistics were settled, by all ranks together, at the end of the run)
        release_run(p, true);
        ph_agg_result *mine = nullptr;
        int rc = fetch_once(p, &mine);
        if (rc == PH_OK && !p->root_replicated) {
            ph_agg_result *merged = nullptr;
            rc = merge_over_ranks(p, mine, &merged);
            ph_agg_result_free(mine);
            mine = merged;
        }
        (void)ph_ctx_set_deferred_errors(p->ctx, 0);
        p->ran = false;
        if (rc == PH_OK) *out = mine;
        return rc;
    }
    // the intermediates go back to the pool BEFORE the host blocks in the download (bookkeeping while the GPU is busy)
    release_run(p, true);
    int rc = fetch_once(p, out);
    for (int again = 0; again < 2 && rc == PH_ECONSTRAINT && !p->conservative; again++) {
        retire_broken_claim(p);
        const std::string first = p->explain;
        rc = run_once(p);
        p->explain = first + p->explain;
        if (rc == PH_OK) { release_run(p, true); rc = fetch_once(p, out); }
    }
    (void)ph_ctx_set_deferred_errors(p->ctx, 0);
    p->ran = false;
    return rc;
}

extern "C" int ph_plan_key_info(const ph_plan *p, int32_t k, int32_t *type, int32_t *scale, const ph_table **table, int32_t *col) {
    PH_REQUIRE(p && k >= 0 && k < (int32_t)p->keys.size(), "ph_plan_key_info: key %d of %zu (after ph_plan_run)", k, p ? p->keys.size() : (size_t)0);
    const KeyInfo &ki = p->keys[(size_t)k];
    if (type) *type = ki.type;
    if (scale) *scale = ki.scale;
    if (table) *table = ki.table;
    if (col) *col = ki.col;
    return PH_OK;
}

extern "C" int ph_plan_agg_arg_type(const ph_plan *p, int32_t a, int32_t *type) {
    PH_REQUIRE(p && type && a >= 0 && a < (int32_t)p->agg_arg_type.size(), "ph_plan_agg_arg_type: aggregate %d (after ph_plan_run)", a);
    *type = p->agg_arg_type[(size_t)a];
    return PH_OK;
}

extern "C" const char *ph_plan_explain(const ph_plan *p) { return p ? p->explain.c_str() : ""; }

extern "C" void ph_plan_free(ph_plan *p) {
    if (!p) return;
    release_run(p, false);
    if (p->scan) ph_scan_plan_free(p->scan);
    delete p;
}

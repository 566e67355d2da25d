// Fused Agg <- Scan(filter) plans: descriptor normalisation, shape matching, overflow proofs,
// launch sequencing and result assembly for ph_scan_plan_* / ph_scan_filter_agg.
//
// What is normalised here are the reference's plan-time typing rules, because they decide which
// integer kernel an SQL operator becomes:
//  * a DECIMAL column compared with a FLOAT literal is compared as float32 after
//    decimal -> float64 -> float32 (bindAConst builder_binder.go:264-273, MaxLType ltype.go:626-643,
//    tryCastDecimalToFloat32 function_cast.go:349-354). That cast is monotone in the unscaled
//    integer, so the predicate is lowered on the host to an integer threshold found by bisection
//    with the exact cast; the device compares integers.
//  * `<` on FLOAT, `=`/`<`/`<=`/`>=` on DECIMAL, `=` on BIGINT ... have no implementation in
//    selectOperation (function_operator_boolean.go:393-504) and select nothing.
//  * decimal Mul adds scales, Add/Sub takes the max (function_scalar.go:37-84, 429-475); integer
//    literals enter at scale 0 (tryCastInt32ToDecimal function_cast.go:337-347).
#include <algorithm>
#include <cmath>

#include "common.h"
#include "scan_jit.h"
#include "scan_kernels.h"

namespace {

using ph::set_error;

struct Affine {  // A + B * col   (col < 0: constant A)
    int64_t A = 0, B = 0;
    int32_t col = -1;
    bool operator==(const Affine &o) const { return A == o.A && B == o.B && col == o.col; }
};

struct Prod {  // product of factors at `scale` fractional digits
    std::vector<Affine> f;
    int32_t scale = 0;
    bool operator==(const Prod &o) const { return f == o.f && scale == o.scale; }
};

bool mul_ok(int64_t a, int64_t b, int64_t *out) { return !__builtin_mul_overflow(a, b, out); }

bool pow10_i64(int k, int64_t *out) {
    int64_t r = 1;
    for (int i = 0; i < k; i++)
        if (!mul_ok(r, 10, &r)) return false;
    *out = r;
    return true;
}

// RPN -> product of affine factors; false = shape outside the fused kernels
bool normalize(const ph_table *t, const ph_rpn *prog, int32_t n, Prod *out) {
    std::vector<Prod> st;
    for (int32_t i = 0; i < n; i++) {
        const ph_rpn &o = prog[i];
        switch (o.op) {
        case PH_X_COL: {
            if (o.col < 0 || o.col >= (int32_t)t->cols.size()) return false;
            const auto &c = t->cols[(size_t)o.col];
            if (c.validity) return false;  // NULL-able inputs take the generic path
            Prod p;
            Affine a;
            a.B = 1;
            a.col = o.col;
            if (c.type == PH_DEC64) p.scale = c.scale;
            else if (c.type == PH_I32 || c.type == PH_I64) p.scale = 0;
            else return false;
            p.f.push_back(a);
            st.push_back(p);
            break;
        }
        case PH_X_CONST: {
            Prod p;
            Affine a;
            a.A = o.ival;
            p.scale = o.scale;
            p.f.push_back(a);
            st.push_back(p);
            break;
        }
        case PH_X_ADD: case PH_X_SUB: {
            if (st.size() < 2) return false;
            Prod y = st.back(); st.pop_back();
            Prod x = st.back(); st.pop_back();
            if (x.f.size() != 1 || y.f.size() != 1) return false;
            Affine a = x.f[0], b = y.f[0];
            if (a.col >= 0 && b.col >= 0) return false;
            int32_t s = std::max(x.scale, y.scale);
            int64_t mx, my;
            if (!pow10_i64(s - x.scale, &mx) || !pow10_i64(s - y.scale, &my)) return false;
            if (!mul_ok(a.A, mx, &a.A) || !mul_ok(a.B, mx, &a.B) || !mul_ok(b.A, my, &b.A) ||
                !mul_ok(b.B, my, &b.B)) return false;
            if (o.op == PH_X_SUB) { b.A = -b.A; b.B = -b.B; }
            Affine r;
            if (__builtin_add_overflow(a.A, b.A, &r.A)) return false;
            r.B = a.col >= 0 ? a.B : b.B;
            r.col = a.col >= 0 ? a.col : b.col;
            Prod p;
            p.scale = s;
            p.f.push_back(r);
            st.push_back(p);
            break;
        }
        case PH_X_MUL: {
            if (st.size() < 2) return false;
            Prod y = st.back(); st.pop_back();
            Prod x = st.back(); st.pop_back();
            x.f.insert(x.f.end(), y.f.begin(), y.f.end());
            x.scale += y.scale;
            st.push_back(x);
            break;
        }
        default: return false;
        }
    }
    if (st.size() != 1) return false;
    // fold constant factors into the first column factor
    Prod p = st[0];
    int64_t k = 1;
    std::vector<Affine> cols;
    for (auto &a : p.f) {
        if (a.col < 0) { if (!mul_ok(k, a.A, &k)) return false; }
        else cols.push_back(a);
    }
    if (cols.empty()) return false;
    if (!mul_ok(cols[0].A, k, &cols[0].A) || !mul_ok(cols[0].B, k, &cols[0].B)) return false;
    p.f = cols;
    *out = p;
    return true;
}

// inclusive integer range a predicate restricts a column to; never = selects nothing
struct Range {
    int32_t col = -1;
    int64_t lo = INT64_MIN, hi = INT64_MAX;
    bool never = false;
};

float dec_to_f32(int64_t d, int scale) {  // tryCastDecimalToFloat32 for |d| < 2^53
    double p = 1;
    for (int i = 0; i < scale; i++) p *= 10;
    return (float)((double)d / p);
}

// smallest d in [lo,hi] with pred(d) true, for a monotone (false..true) predicate; hi+1 if none
template <typename F> int64_t first_true(int64_t lo, int64_t hi, F pred) {
    if (!pred(hi)) return hi == INT64_MAX ? hi : hi + 1;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (pred(mid)) hi = mid; else lo = mid + 1;
    }
    return lo;
}

int lower_pred(const ph_table *t, const ph_pred &p, Range *r) {
    if (p.col < 0 || p.col >= (int32_t)t->cols.size()) { set_error("predicate column %d out of range", p.col); return PH_EINVAL; }
    const auto &c = t->cols[(size_t)p.col];
    if (c.validity) return PH_EUNSUPPORTED;
    r->col = p.col;
    int32_t op = p.op;
    auto range_from = [&](int64_t k) {
        switch (op) {
        case PH_EQ: r->lo = r->hi = k; break;
        case PH_LT: if (k == INT64_MIN) r->never = true; else r->hi = k - 1; break;
        case PH_LE: r->hi = k; break;
        case PH_GT: if (k == INT64_MAX) r->never = true; else r->lo = k + 1; break;
        case PH_GE: r->lo = k; break;
        default: break;
        }
    };
    switch (c.type) {
    case PH_I32:
        if (p.k.type != PH_I32) return PH_EUNSUPPORTED;
        if (op == PH_NE) return PH_EUNSUPPORTED;  // not a range: generic path
        if (op < PH_EQ || op > PH_GE) { r->never = true; return PH_OK; }
        range_from((int32_t)p.k.i);
        return PH_OK;
    case PH_DATE:
        if (p.k.type != PH_DATE) return PH_EUNSUPPORTED;
        if (op == PH_EQ || op == PH_NE || op < PH_EQ || op > PH_GE) { r->never = true; return PH_OK; }  // no DATE '='
        range_from((int32_t)p.k.i);
        return PH_OK;
    case PH_I64:
        r->never = true;  // no BIGINT comparison is implemented in selectOperation
        return PH_OK;
    case PH_CODE8: {
        if (p.k.type != PH_STR || !p.k.s) return PH_EUNSUPPORTED;
        if (op != PH_EQ) return PH_EUNSUPPORTED;
        int code = -1;
        for (size_t i = 0; i < c.dict.size(); i++)
            if (c.dict[i] == p.k.s) code = (int)i;
        if (code < 0) r->never = true; else r->lo = r->hi = code;
        return PH_OK;
    }
    case PH_DEC64: {
        if (p.k.type == PH_F32) {
            // float32 comparison: >, >=, <= exist; <, =, != do not (FLOAT rows of selectOperation)
            if (op != PH_GT && op != PH_GE && op != PH_LE) { r->never = true; return PH_OK; }
            if (!c.has_range) return PH_EUNSUPPORTED;
            const int64_t LIM = 1ll << 53;
            if (c.min <= -LIM || c.max >= LIM) return PH_EUNSUPPORTED;
            float kf = (float)p.k.f;
            int sc = c.scale;
            int64_t lo = c.min, hi = c.max;
            if (op == PH_GE) {
                int64_t d = first_true(lo, hi, [&](int64_t x) { return dec_to_f32(x, sc) >= kf; });
                if (d > hi) r->never = true; else r->lo = d;
            } else if (op == PH_GT) {
                int64_t d = first_true(lo, hi, [&](int64_t x) { return dec_to_f32(x, sc) > kf; });
                if (d > hi) r->never = true; else r->lo = d;
            } else {  // LE: everything below the first value that is > kf
                int64_t d = first_true(lo, hi, [&](int64_t x) { return dec_to_f32(x, sc) > kf; });
                if (d == lo && dec_to_f32(lo, sc) > kf) r->never = true; else r->hi = d > hi ? hi : d - 1;
            }
            return PH_OK;
        }
        if (p.k.type == PH_DEC64) {
            if (op != PH_GT) { r->never = true; return PH_OK; }  // only DECIMAL '>' exists
            // align the literal to the column scale (exact when the literal has fewer digits)
            if (p.k.scale > c.scale) return PH_EUNSUPPORTED;
            int64_t m, k;
            if (!pow10_i64(c.scale - p.k.scale, &m) || !mul_ok(p.k.i, m, &k)) return PH_EUNSUPPORTED;
            range_from(k);
            return PH_OK;
        }
        return PH_EUNSUPPORTED;
    }
    default:
        return PH_EUNSUPPORTED;
    }
}

// |A + B*x| over x in [mn,mx], as a long double bound
long double affine_bound(const Affine &a, int64_t mn, int64_t mx) {
    long double v1 = (long double)a.A + (long double)a.B * (long double)mn;
    long double v2 = (long double)a.A + (long double)a.B * (long double)mx;
    return std::max(fabsl(v1), fabsl(v2));
}

}  // namespace

enum PlanKind { PK_FILTER_SUMPROD = 1, PK_LOWCARD_CHAIN = 2, PK_GENERIC = 3, PK_JIT = 4 };

struct ph_scan_plan {
    ph_ctx *ctx = nullptr;
    const ph_table *t = nullptr;
    int kind = 0;
    bool never = false;  // some conjunct selects nothing
    ph::FilterSumProdParams fs{};
    ph::LowcardChainParams lc{};
    int max_grid = 0;
    int nacc = 0;  // accumulators per launch (nslots * LC_NACC, or 2)
    // requested aggregates -> accumulator
    struct AggMap { int32_t kind; int acc; int32_t scale; };
    std::vector<AggMap> aggs;
    int32_t nkeys = 0;
    int32_t group_cols[4] = {-1, -1, -1, -1};
    // layout of the raw accumulator words of the fused kinds: per group slot `stride` words, the
    // count at cnt_idx, the first-seen row id at first_idx (-1: none), ops[j] = how word j merges
    // (0 sum, 1 min, 2 max); slot = dense index over gcard
    int nslots = 1, stride = 0, cnt_idx = 0, first_idx = -1;
    std::vector<int> ops, gcard;
    // PK_JIT: the plan-specialised kernel
    ph::JitShape jshape;
    ph::JitKernel jkernel;
    ph::JitParams jparams{};
    // device buffers
    long long *partials = nullptr;
    unsigned long long *out_lo = nullptr;
    long long *out_hi = nullptr;
    long double row_bound = 0;  // largest |per-row accumulator value|
    int64_t last_rows = 0;
    int last_grid = 0;
    unsigned long long armed_seq = 0;   // the last run's own publish (ScanTail), 0 = none: fetch downloads
    bool unfetched = false;             // the last run's result was never fetched: a caller that runs back to back (a throughput loop, the N-rank
                                        // path reading the partials on the device) gets no publish armed — the mailbox store is a system fence per run
    // PK_GENERIC: the descriptor itself, run through the operator-granular kernels
    std::vector<ph_pred> g_preds;
    std::vector<std::string> g_pred_strs;
    std::vector<int32_t> g_groups;
    std::vector<ph_aggexpr> g_aggs;
    ph_agg *g_agg = nullptr;
};

// The plan's two device buffers come from the ctx's stream-ordered pool: a plan that is created, run, fetched and freed per query — the
// executor behind the operator interface is built and closed per query, like the reference's — then costs no hipMalloc / hipFree. Both
// synchronise the device: rocprofv3 --hip-trace of Q1 at SF10 behind OperatorExec showed two hipFree calls of ~160 us each per query,
// the whole of the 0.28 ms it stood above the fused kernel (profiles/r04_q1_opif_before_hip_api_stats.csv).
static int plan_alloc(ph_scan_plan *p) {
    PH_CHECK(p->ctx->pool_alloc((int64_t)std::max(p->max_grid, 8192) * p->nacc * (int64_t)sizeof(long long), (void **)&p->partials));
    // one allocation, lo[nacc] then hi[nacc]: the raw partial result other ranks all-gather
    PH_CHECK(p->ctx->pool_alloc((int64_t)p->nacc * 2 * (int64_t)sizeof(unsigned long long), (void **)&p->out_lo));
    p->out_hi = (long long *)(p->out_lo + p->nacc);
    return PH_OK;
}

extern "C" void ph_scan_plan_free(ph_scan_plan *p) {
    if (!p) return;
    // stream-ordered reuse: whatever takes these blocks next is queued behind the plan's last kernel on the ctx's one stream
    if (p->partials) p->ctx->pool_release(p->partials);
    if (p->out_lo) p->ctx->pool_release(p->out_lo);
    if (p->g_agg) ph_agg_free(p->g_agg);
    delete p;
}

extern "C" const char *ph_scan_plan_kind(const ph_scan_plan *p) {
    if (!p) return "";
    return p->kind == PK_FILTER_SUMPROD ? "filter_sumprod" : p->kind == PK_LOWCARD_CHAIN ? "lowcard_chain" : p->kind == PK_JIT ? "jit" : "generic";
}

static int try_fused(ph_ctx *ctx, const ph_table *t, const ph_pred *preds, int32_t npreds,
                     const int32_t *group_cols, int32_t ngroup_cols, const ph_aggexpr *aggs, int32_t naggs,
                     ph_scan_plan **out) {
    // ---- predicates -> one range per column
    std::vector<Range> ranges;
    bool never = false;
    for (int32_t i = 0; i < npreds; i++) {
        Range r;
        int rc = lower_pred(t, preds[i], &r);
        if (rc == PH_EUNSUPPORTED) { set_error("predicate %d is outside the fused shapes", i); return rc; }
        PH_CHECK(rc);
        if (r.never) never = true;
        bool merged = false;
        for (auto &q : ranges)
            if (q.col == r.col) { q.lo = std::max(q.lo, r.lo); q.hi = std::min(q.hi, r.hi); merged = true; }
        if (!merged) ranges.push_back(r);
    }
    for (auto &q : ranges) if (q.lo > q.hi) never = true;

    // ---- aggregates -> normalised products
    struct Req { int32_t kind; Prod p; bool star; };
    std::vector<Req> reqs;
    for (int32_t a = 0; a < naggs; a++) {
        Req r;
        r.kind = aggs[a].kind;
        r.star = aggs[a].kind == PH_A_COUNT_STAR;
        if (!r.star) {
            if (aggs[a].kind != PH_A_SUM && aggs[a].kind != PH_A_AVG && aggs[a].kind != PH_A_COUNT) {
                set_error("aggregate %d: MIN/MAX take the generic path", a);
                return PH_EUNSUPPORTED;
            }
            if (aggs[a].nprog <= 0 || aggs[a].nprog > 12 || !normalize(t, aggs[a].prog, aggs[a].nprog, &r.p)) {
                set_error("aggregate %d: argument expression is outside the fused shapes", a);
                return PH_EUNSUPPORTED;
            }
        }
        reqs.push_back(r);
    }

    ph_scan_plan *p = new ph_scan_plan();
    p->ctx = ctx;
    p->t = t;
    p->never = never;
    auto fail = [&](int rc) { ph_scan_plan_free(p); return rc; };
    auto col = [&](int32_t c) -> const ph_table::column & { return t->cols[(size_t)c]; };
    auto is_i32 = [&](int32_t c) { return col(c).type == PH_I32 || col(c).type == PH_DATE; };
    auto is_i64 = [&](int32_t c) { return col(c).type == PH_I64 || col(c).type == PH_DEC64; };
    auto pure = [](const Affine &a) { return a.A == 0 && a.B == 1 && a.col >= 0; };

    if (ngroup_cols == 0) {
        // ---------------- filter_sumprod: SUM(a*b) [+ COUNT(*)], ranges on <=2 int32 + <=1 int64 col
        p->kind = PK_FILTER_SUMPROD;
        int32_t a_col = -1, b_col = -1;
        for (size_t i = 0; i < reqs.size(); i++) {
            ph_scan_plan::AggMap m{reqs[i].kind, 1, 0};
            if (!reqs[i].star) {
                const Prod &pr = reqs[i].p;
                if (pr.f.size() != 2 || !pure(pr.f[0]) || !pure(pr.f[1]) || !is_i64(pr.f[0].col) || !is_i64(pr.f[1].col)) {
                    set_error("aggregate %zu is not SUM(col*col) over 64-bit columns", i);
                    return fail(PH_EUNSUPPORTED);
                }
                if (a_col >= 0 && !(a_col == pr.f[0].col && b_col == pr.f[1].col)) return fail(PH_EUNSUPPORTED);
                a_col = pr.f[0].col;
                b_col = pr.f[1].col;
                m.acc = 0;
                m.scale = pr.scale;
            }
            p->aggs.push_back(m);
        }
        if (a_col < 0) { set_error("filter_sumprod needs one SUM(col*col)"); return fail(PH_EUNSUPPORTED); }
        std::vector<Range> r32, r64;
        for (auto &r : ranges) (is_i32(r.col) ? r32 : r64).push_back(r);
        for (auto &r : ranges) if (!is_i32(r.col) && !is_i64(r.col)) return fail(PH_EUNSUPPORTED);
        if (r32.empty() || r32.size() > 2 || r64.size() > 1) { set_error("filter_sumprod: predicate columns do not fit (need 1-2 int32, 0-1 int64)"); return fail(PH_EUNSUPPORTED); }
        if (!r64.empty()) {
            // the 64-bit predicate column must be one of the factors; make it `b`
            if (r64[0].col == a_col) std::swap(a_col, b_col);
            if (r64[0].col != b_col) return fail(PH_EUNSUPPORTED);
            p->fs.b_lo = r64[0].lo;
            p->fs.b_hi = r64[0].hi;
        } else {
            p->fs.b_lo = INT64_MIN;
            p->fs.b_hi = INT64_MAX;
        }
        auto clamp32 = [](int64_t v) { return (int32_t)std::max<int64_t>(INT32_MIN, std::min<int64_t>(INT32_MAX, v)); };
        p->fs.p0 = (const int32_t *)col(r32[0].col).data;
        p->fs.p0_lo = clamp32(r32[0].lo);
        p->fs.p0_hi = clamp32(r32[0].hi);
        const Range &second = r32.size() > 1 ? r32[1] : r32[0];
        p->fs.p2 = (const int32_t *)col(second.col).data;
        p->fs.p2_lo = clamp32(second.lo);
        p->fs.p2_hi = clamp32(second.hi);
        p->fs.a = (const int64_t *)col(a_col).data;
        p->fs.b = (const int64_t *)col(b_col).data;
        if (!col(a_col).has_range || !col(b_col).has_range) return fail(PH_EUNSUPPORTED);
        Affine one; one.B = 1;
        p->row_bound = affine_bound(one, col(a_col).min, col(a_col).max) * affine_bound(one, col(b_col).min, col(b_col).max);
        p->nacc = 2;
        p->nslots = 1; p->stride = 2; p->cnt_idx = 1; p->first_idx = -1; p->ops = {0, 0};
        // one 256-thread workgroup per CU (1 wave per SIMD) streams fastest: measured on MI355X,
        // SF10: 6.48 TB/s at grid 256 vs 5.95 at 2048 and 4.6-5.4 at grids that are not a
        // multiple of the CU count (tail imbalance); scripts/ab_scan2.sh
        p->max_grid = ctx->cu_count;
    } else {
        // ---------------- lowcard_chain: two dictionary-code group columns, one int32 range predicate
        p->kind = PK_LOWCARD_CHAIN;
        if (ngroup_cols != 2 || col(group_cols[0]).type != PH_CODE8 || col(group_cols[1]).type != PH_CODE8 ||
            col(group_cols[0]).validity || col(group_cols[1]).validity) {
            set_error("lowcard_chain needs two non-NULL dictionary-code group columns");
            return fail(PH_EUNSUPPORTED);
        }
        int n0 = (int)col(group_cols[0]).dict.size(), n1 = (int)col(group_cols[1]).dict.size();
        if (n0 == 0) n0 = (int)col(group_cols[0]).max + 1;
        if (n1 == 0) n1 = (int)col(group_cols[1]).max + 1;
        // every code must index inside the LDS slots sized from n0*n1: the recorded maxima prove it
        for (int gi = 0; gi < 2; gi++) {
            const auto &gc = col(group_cols[gi]);
            if (!gc.has_range || gc.min < 0 || gc.max >= (gi == 0 ? n0 : n1)) {
                set_error("lowcard_chain: group column %d holds codes outside its dictionary", group_cols[gi]);
                return fail(PH_EUNSUPPORTED);
            }
        }
        if (n0 * n1 > ph::LC_MAX_SLOTS || n0 * n1 <= 0) { set_error("lowcard_chain: %d group slots exceed %d", n0 * n1, ph::LC_MAX_SLOTS); return fail(PH_EUNSUPPORTED); }
        p->nkeys = 2;
        p->group_cols[0] = group_cols[0];
        p->group_cols[1] = group_cols[1];
        if (ranges.size() != 1 || !is_i32(ranges[0].col)) { set_error("lowcard_chain needs exactly one int32/date range predicate"); return fail(PH_EUNSUPPORTED); }
        // roles: q (int32 sum), e, d, t with f1 = A1+B1*d, f2 = A2+B2*t
        int32_t q = -1, e = -1, d = -1, tt = -1;
        Affine f1, f2;
        bool have_f1 = false, have_f2 = false;
        // longest chain first
        std::vector<size_t> order(reqs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return reqs[x].p.f.size() > reqs[y].p.f.size(); });
        p->aggs.resize(reqs.size());
        int32_t singles64[2] = {-1, -1};
        for (size_t oi : order) {
            const Req &r = reqs[oi];
            ph_scan_plan::AggMap m{r.kind, 5, 0};
            if (!r.star) {
                const Prod &pr = r.p;
                m.scale = pr.scale;
                if (pr.f.size() == 3) {
                    if (!pure(pr.f[0]) || !is_i64(pr.f[0].col) || !is_i64(pr.f[1].col) || !is_i64(pr.f[2].col)) return fail(PH_EUNSUPPORTED);
                    if (have_f2 && !(e == pr.f[0].col && f1 == pr.f[1] && f2 == pr.f[2])) return fail(PH_EUNSUPPORTED);
                    e = pr.f[0].col; f1 = pr.f[1]; f2 = pr.f[2]; have_f1 = have_f2 = true;
                    d = f1.col; tt = f2.col;
                    m.acc = 3;
                } else if (pr.f.size() == 2) {
                    if (!pure(pr.f[0]) || !is_i64(pr.f[0].col) || !is_i64(pr.f[1].col)) return fail(PH_EUNSUPPORTED);
                    if (have_f1 && !(e == pr.f[0].col && f1 == pr.f[1])) return fail(PH_EUNSUPPORTED);
                    e = pr.f[0].col; f1 = pr.f[1]; have_f1 = true; d = f1.col;
                    m.acc = 2;
                } else if (pr.f.size() == 1 && pure(pr.f[0])) {
                    int32_t c = pr.f[0].col;
                    if (is_i32(c)) { if (q >= 0 && q != c) return fail(PH_EUNSUPPORTED); q = c; m.acc = 0; }
                    else if (is_i64(c)) {
                        if (c == e || (e < 0 && singles64[0] < 0)) { if (e < 0) { e = c; } m.acc = 1; singles64[0] = c; }
                        else if (c == d || d < 0) { d = c; m.acc = 4; singles64[1] = c; }
                        else return fail(PH_EUNSUPPORTED);
                    } else return fail(PH_EUNSUPPORTED);
                } else return fail(PH_EUNSUPPORTED);
            }
            p->aggs[oi] = m;
        }
        if (e < 0) { set_error("lowcard_chain needs at least one 64-bit summed column"); return fail(PH_EUNSUPPORTED); }
        // The precompiled kernel always loads its four roles; a plan that names fewer columns would
        // re-read a stand-in (measured: 3.5 TB/s of useful bytes instead of 6.3). Such plans get a
        // generated kernel that reads only what they name — unless code generation is switched off.
        const char *je = getenv("PH_SCAN_JIT");
        const bool nojit = je && atoi(je) == 0;
        if ((d < 0 || tt < 0 || q < 0) && !nojit) { set_error("lowcard_chain: the plan names fewer columns than the kernel reads"); return fail(PH_EUNSUPPORTED); }
        if (d < 0) d = e;
        if (tt < 0) tt = e;
        if (q < 0) q = ranges[0].col;
        if (!have_f1) { f1 = Affine(); f1.A = 1; f1.col = d; }
        if (!have_f2) { f2 = Affine(); f2.A = 1; f2.col = tt; }
        // an accumulator 4 request must be over the SAME column as f1's (both called d)
        for (auto &m : p->aggs) (void)m;
        p->lc.p = (const int32_t *)col(ranges[0].col).data;
        auto clamp32 = [](int64_t v) { return (int32_t)std::max<int64_t>(INT32_MIN, std::min<int64_t>(INT32_MAX, v)); };
        p->lc.p_lo = clamp32(ranges[0].lo);
        p->lc.p_hi = clamp32(ranges[0].hi);
        p->lc.q = (const int32_t *)col(q).data;
        p->lc.e = (const int64_t *)col(e).data;
        p->lc.d = (const int64_t *)col(d).data;
        p->lc.t = (const int64_t *)col(tt).data;
        p->lc.k0 = (const uint8_t *)col(group_cols[0]).data;
        p->lc.k1 = (const uint8_t *)col(group_cols[1]).data;
        p->lc.nk1 = n1;
        p->lc.nslots = n0 * n1;
        p->lc.A1 = f1.A; p->lc.B1 = f1.B; p->lc.A2 = f2.A; p->lc.B2 = f2.B;
        for (int32_t c : {q, e, d, tt}) if (!col(c).has_range) return fail(PH_EUNSUPPORTED);
        Affine one; one.B = 1;
        long double be = affine_bound(one, col(e).min, col(e).max);
        long double b1 = affine_bound(f1, col(d).min, col(d).max);
        long double b2 = affine_bound(f2, col(tt).min, col(tt).max);
        p->row_bound = std::max({be * b1 * b2, be * b1, be, affine_bound(one, col(d).min, col(d).max),
                                 affine_bound(one, col(q).min, col(q).max)});
        p->nacc = p->lc.nslots * (ph::LC_NACC + 1);  // + first_row
        p->nslots = p->lc.nslots; p->stride = ph::LC_NACC + 1; p->cnt_idx = 5; p->first_idx = ph::LC_NACC;
        p->ops = {0, 0, 0, 0, 0, 0, 1};
        p->gcard = {n0, n1};
        // one workgroup per CU: 6.35 TB/s at grid 256 vs 6.0 at 512 (2 per CU is what the
        // per-thread-private LDS accumulators would still allow); scripts/ab_scan2.sh
        p->max_grid = ctx->cu_count;
    }
    int rc = plan_alloc(p);
    if (rc != PH_OK) return fail(rc);
    *out = p;
    return PH_OK;
}



// ---------------------------------------------------------------- plan-specialised (hiprtc) plans
// Any conjunction of range / `!=` predicates over NULL-free fixed-width columns, grouped by up to
// four dictionary-code columns (dense slots that fit the CU's LDS as per-thread-private
// accumulator columns) and aggregated with SUM/AVG/COUNT over products of affine column factors or
// MIN/MAX of a column, becomes one generated kernel (scan_jit.h): only the columns the plan names
// are read, each byte once.
static int try_jit(ph_ctx *ctx, const ph_table *t, const ph_pred *preds, int32_t npreds,
                   const int32_t *group_cols, int32_t ngroup_cols, const ph_aggexpr *aggs, int32_t naggs,
                   ph_scan_plan **out) {
    auto unsupported = [&](const char *why) { set_error("plan-specialised kernel: %s", why); return PH_EUNSUPPORTED; };
    if (ngroup_cols > 4) return unsupported("more than 4 group columns");
    ph::JitShape S;
    std::vector<long long> consts;
    std::vector<int32_t> tcol;   // loaded column slot -> table column
    auto slot_of = [&](int32_t c) -> int {
        for (size_t i = 0; i < tcol.size(); i++) if (tcol[i] == c) return (int)i;
        const auto &d = t->cols[(size_t)c];
        int w = d.type == PH_CODE8 ? 1 : (d.type == PH_I32 || d.type == PH_DATE) ? 4 : (d.type == PH_I64 || d.type == PH_DEC64) ? 8 : 0;
        if (w == 0 || d.validity) return -1;
        tcol.push_back(c);
        S.col_width.push_back(w);
        return (int)tcol.size() - 1;
    };
    // ---- predicates: one merged range per column, `!=` separately
    bool never = false;
    std::vector<Range> ranges;
    struct Ne { int32_t col; long long k; };
    std::vector<Ne> nes;
    for (int32_t i = 0; i < npreds; i++) {
        const ph_pred &pr = preds[i];
        if (pr.col < 0 || pr.col >= (int32_t)t->cols.size()) { set_error("predicate column %d out of range", pr.col); return PH_EINVAL; }
        const auto &c = t->cols[(size_t)pr.col];
        if (pr.op == PH_NE && c.type == PH_I32 && pr.k.type == PH_I32 && !c.validity) { nes.push_back({pr.col, (int32_t)pr.k.i}); continue; }
        Range r;
        int rc = lower_pred(t, pr, &r);
        if (rc == PH_EUNSUPPORTED) return unsupported("a predicate does not lower to an integer range");
        PH_CHECK(rc);
        if (r.never) never = true;
        bool merged = false;
        for (auto &q : ranges) if (q.col == r.col) { q.lo = std::max(q.lo, r.lo); q.hi = std::min(q.hi, r.hi); merged = true; }
        if (!merged) ranges.push_back(r);
    }
    for (auto &q : ranges) if (q.lo > q.hi) never = true;
    for (auto &q : ranges) {
        int sl = slot_of(q.col);
        if (sl < 0) return unsupported("predicate column type");
        S.preds.push_back({sl, false});
        consts.push_back(q.lo);
        consts.push_back(q.hi);
    }
    for (auto &q : nes) {
        int sl = slot_of(q.col);
        if (sl < 0) return unsupported("predicate column type");
        S.preds.push_back({sl, true});
        consts.push_back(q.k);
    }
    // ---- group columns: dense slot over dictionary codes
    ph_scan_plan *p = new ph_scan_plan();
    p->ctx = ctx; p->t = t; p->kind = PK_JIT; p->never = never;
    auto fail = [&](int rc) { ph_scan_plan_free(p); return rc; };
    int64_t ns = 1;
    for (int32_t g = 0; g < ngroup_cols; g++) {
        int32_t gc = group_cols[g];
        if (gc < 0 || gc >= (int32_t)t->cols.size()) { set_error("group column %d out of range", gc); return fail(PH_EINVAL); }
        const auto &c = t->cols[(size_t)gc];
        if (c.type != PH_CODE8 || c.validity || !c.has_range || c.min < 0) return fail(unsupported("group columns must be NULL-free dictionary codes"));
        int card = (int)std::max<int64_t>((int64_t)c.dict.size(), c.max + 1);
        ns *= card;
        if (ns > 4096) return fail(unsupported("too many group slots"));
        S.group_col.push_back(slot_of(gc));
        S.group_card.push_back(card);
        p->group_cols[g] = gc;
        p->gcard.push_back(card);
    }
    p->nkeys = ngroup_cols;
    S.nslots = (int)ns;
    // ---- aggregates -> deduplicated accumulators
    struct AccReq { int op; Prod pr; };
    std::vector<AccReq> accs;
    long double row_bound = 0;
    for (int32_t a = 0; a < naggs; a++) {
        ph_scan_plan::AggMap m{aggs[a].kind, -1, 0};
        if (aggs[a].kind != PH_A_COUNT_STAR) {
            Prod pr;
            if (aggs[a].nprog <= 0 || aggs[a].nprog > 12 || !normalize(t, aggs[a].prog, aggs[a].nprog, &pr)) return fail(unsupported("an aggregate argument is not a product of affine column factors"));
            m.scale = pr.scale;
            int op = aggs[a].kind == PH_A_MIN ? 1 : aggs[a].kind == PH_A_MAX ? 2 : 0;
            if (aggs[a].kind != PH_A_COUNT) {
                if (op != 0 && !(pr.f.size() == 1 && pr.f[0].A == 0 && pr.f[0].B == 1)) return fail(unsupported("MIN/MAX of an expression"));
                if (pr.f.size() > 4) return fail(unsupported("more than 4 factors"));
                int found = -1;
                for (size_t i = 0; i < accs.size(); i++) if (accs[i].op == op && accs[i].pr == pr) found = (int)i;
                if (found < 0) { accs.push_back({op, pr}); found = (int)accs.size() - 1; }
                m.acc = found;
            }
        }
        p->aggs.push_back(m);
    }
    if (accs.size() > 16) return fail(unsupported("more than 16 distinct accumulators"));
    for (auto &aq : accs) {
        ph::JitShape::Acc A;
        A.op = aq.op;
        long double b = 1;
        for (auto &f : aq.pr.f) {
            const auto &c = t->cols[(size_t)f.col];
            if (!c.has_range) return fail(unsupported("a summed column has no range statistics"));
            int sl = slot_of(f.col);
            if (sl < 0) return fail(unsupported("aggregate column type"));
            bool pure = f.A == 0 && f.B == 1;
            A.factors.push_back({sl, pure});
            if (!pure) { consts.push_back(f.A); consts.push_back(f.B); }
            b *= affine_bound(f, c.min, c.max);
        }
        if (aq.op == 0) row_bound = std::max(row_bound, b);
        S.accs.push_back(A);
    }
    if (tcol.empty()) return fail(unsupported("the plan reads no column"));   // count(*) without predicates
    if ((int)tcol.size() > ph::JIT_MAX_COLS || (int)consts.size() > ph::JIT_MAX_CONSTS) return fail(unsupported("too many columns / constants"));
    const int na = (int)S.accs.size();
    S.lds_cols = ph::JitShape::fit_lds_cols(S.nslots, na);
    if (S.lds_cols == 0) return fail(unsupported("group slots x accumulators exceed the CU's LDS"));
    // Bytes in flight: a workgroup keeps one 1024-row tile prefetched, i.e. 1024 x (row bytes). The
    // 34 B/row of Q1 (34 KiB per CU) covers the HBM latency-bandwidth product with one workgroup
    // per CU; narrower plans need proportionally more resident workgroups (measured: 9 B/row at one
    // workgroup per CU reads 4.0 TB/s, latency bound), as far as their LDS accumulators allow.
    int row_bytes = 0;
    for (int w : S.col_width) row_bytes += w;
    int per_cu = std::max(1, std::min(8, (34 + row_bytes - 1) / row_bytes));
    const int64_t lds_bytes = (int64_t)S.nslots * (na * 8 + 8) * S.lds_cols;
    while (per_cu > 1 && per_cu * lds_bytes > 150 * 1024) per_cu--;
    p->row_bound = row_bound;
    p->jshape = S;
    for (size_t i = 0; i < tcol.size(); i++) p->jparams.col[i] = t->cols[(size_t)tcol[i]].data;
    for (size_t i = 0; i < consts.size(); i++) p->jparams.k[i] = consts[i];
    p->nslots = S.nslots; p->stride = na + 2; p->cnt_idx = na; p->first_idx = na + 1;
    for (auto &A : S.accs) p->ops.push_back(A.op);
    p->ops.push_back(0);
    p->ops.push_back(1);
    p->nacc = p->nslots * p->stride;
    p->max_grid = ctx->cu_count * per_cu;
    int rc = ph::jit_get(ctx, S, &p->jkernel);
    if (rc != PH_OK) return fail(rc == PH_EHIP ? PH_EUNSUPPORTED : rc);   // no hiprtc / compile trouble: the generic chain still runs
    rc = plan_alloc(p);
    if (rc != PH_OK) return fail(rc);
    *out = p;
    return PH_OK;
}

// ---------------------------------------------------------------- generic plans
// Any Agg <- Scan(filter) descriptor outside the fused shapes still runs on the device, as the
// chain the operator-granular executors would run: ph_filter_select per conjunct (narrowing the
// selection) -> ph_expr_eval per aggregate argument -> ph_agg_sink -> ph_agg_finalize.

extern "C" int ph_scan_plan_create(ph_ctx *ctx, const ph_table *t, const ph_pred *preds,
                                   int32_t npreds, const int32_t *group_cols, int32_t ngroup_cols,
                                   const ph_aggexpr *aggs, int32_t naggs, ph_scan_plan **out) {
    PH_REQUIRE(ctx && t && out && naggs > 0 && aggs && npreds >= 0 && ngroup_cols >= 0,
               "ph_scan_plan_create: bad arguments");
    // PH_SCAN_JIT=1: prefer the plan-specialised kernel even for the two precompiled shapes (A/B);
    // PH_SCAN_JIT=0: never generate code
    const char *je = getenv("PH_SCAN_JIT");
    const int jit_mode = je ? atoi(je) : -1;
    int rc = PH_EUNSUPPORTED;
    if (jit_mode == 1) rc = try_jit(ctx, t, preds, npreds, group_cols, ngroup_cols, aggs, naggs, out);
    if (rc == PH_EUNSUPPORTED) rc = try_fused(ctx, t, preds, npreds, group_cols, ngroup_cols, aggs, naggs, out);
    if (rc == PH_EUNSUPPORTED && jit_mode != 0 && jit_mode != 1) rc = try_jit(ctx, t, preds, npreds, group_cols, ngroup_cols, aggs, naggs, out);
    if (rc != PH_EUNSUPPORTED) return rc;
    PH_REQUIRE(ngroup_cols <= 4 && naggs <= 16, "ph_scan_plan_create: at most 4 group columns and 16 aggregates");
    ph_scan_plan *p = new ph_scan_plan();
    p->ctx = ctx;
    p->t = t;
    p->kind = PK_GENERIC;
    p->g_pred_strs.resize((size_t)npreds);
    for (int32_t i = 0; i < npreds; i++) {
        p->g_preds.push_back(preds[i]);
        if (preds[i].k.s) p->g_pred_strs[(size_t)i] = preds[i].k.s;
        if (preds[i].col < 0 || preds[i].col >= (int32_t)t->cols.size()) { delete p; set_error("predicate column %d out of range", preds[i].col); return PH_EINVAL; }
    }
    for (int32_t i = 0; i < ngroup_cols; i++) {
        if (group_cols[i] < 0 || group_cols[i] >= (int32_t)t->cols.size()) { delete p; set_error("group column %d out of range", group_cols[i]); return PH_EINVAL; }
        p->g_groups.push_back(group_cols[i]);
    }
    p->nkeys = ngroup_cols;
    std::vector<ph_col> protos(t->cols.size());
    for (size_t c = 0; c < t->cols.size(); c++) { protos[c].type = t->cols[c].type; protos[c].scale = t->cols[c].scale; }
    for (int32_t a = 0; a < naggs; a++) {
        p->g_aggs.push_back(aggs[a]);
        ph_scan_plan::AggMap m{aggs[a].kind, a, 0};
        if (aggs[a].kind != PH_A_COUNT_STAR) {
            int rc2 = ph_expr_scale(protos.data(), aggs[a].prog, aggs[a].nprog, &m.scale);
            if (rc2 != PH_OK) { delete p; return rc2; }
        }
        p->aggs.push_back(m);
    }
    *out = p;
    return PH_OK;
}

static int generic_run(ph_scan_plan *p, int64_t row_begin, int64_t row_end) {
    ph_ctx *ctx = p->ctx;
    const ph_table *t = p->t;
    int64_t n = row_end - row_begin;
    // device views of the table columns, shifted to row_begin
    std::vector<ph_col> cols(t->cols.size());
    for (size_t c = 0; c < t->cols.size(); c++) {
        const auto &d = t->cols[c];
        ph_col v{};
        v.type = d.type; v.scale = d.scale; v.aux = d.aux; v.aux_bytes = d.aux_bytes;
        int w = d.type == PH_STR ? 4 : ph::type_width(d.type);
        v.data = (const char *)d.data + row_begin * w;
        if (d.validity) {
            PH_REQUIRE(row_begin % 8 == 0, "generic plan over NULL-able columns needs row_begin %% 8 == 0");
            v.validity = d.validity + row_begin / 8;
        }
        cols[c] = v;
    }
    if (p->g_agg) { ph_agg_free(p->g_agg); p->g_agg = nullptr; }
    std::vector<int32_t> key_types;
    for (int32_t g : p->g_groups) key_types.push_back(t->cols[(size_t)g].type);
    if (key_types.empty()) key_types.push_back(PH_I32);
    std::vector<ph_aggspec> specs;
    for (size_t a = 0; a < p->g_aggs.size(); a++) specs.push_back(ph_aggspec{p->g_aggs[a].kind, (int32_t)a});
    PH_CHECK(ph_agg_create(ctx, (int32_t)key_types.size(), key_types.data(), (int32_t)specs.size(), specs.data(), 1024, &p->g_agg));
    if (n <= 0) return PH_OK;
    std::vector<void *> temps;
    auto cleanup = [&]() { for (void *q : temps) ctx->pool_release(q); };
    auto fail = [&](int rc) { cleanup(); return rc; };
    // ---- conjuncts narrow the selection one after the other (execSelectAnd)
    int32_t *sel = nullptr;
    int64_t cnt = n;
    for (size_t i = 0; i < p->g_preds.size() && cnt > 0; i++) {
        ph_pred pr = p->g_preds[i];
        pr.k.s = p->g_pred_strs[i].empty() ? nullptr : p->g_pred_strs[i].c_str();
        const auto &d = t->cols[(size_t)pr.col];
        if (d.type == PH_CODE8 && pr.k.type == PH_STR) {  // literal -> dictionary code
            int code = 999;
            for (size_t k = 0; k < d.dict.size(); k++) if (pr.k.s && d.dict[k] == pr.k.s) code = (int)k;
            pr.k.type = PH_I32;
            pr.k.i = code;
        }
        int32_t *out = nullptr;
        if (ctx->pool_alloc(cnt * 4, (void **)&out) != PH_OK) return fail(PH_EHIP);
        temps.push_back(out);
        int64_t m = 0;
        int rc = ph_filter_select(ctx, &cols[(size_t)pr.col], n, pr.op, &pr.k, sel, cnt, out, &m);
        if (rc != PH_OK) return fail(rc);
        sel = out;
        cnt = m;
    }
    if (cnt == 0) { cleanup(); return PH_OK; }
    // ---- keys
    std::vector<ph_col> keys;
    for (int32_t g : p->g_groups) keys.push_back(cols[(size_t)g]);
    if (keys.empty()) {
        void *zero = nullptr;
        int64_t span = sel ? n : cnt;  // addressed by row id
        if (ctx->pool_alloc(span * 4, &zero) != PH_OK) return fail(PH_EHIP);
        temps.push_back(zero);
        if (hipMemsetAsync(zero, 0, (size_t)span * 4, ctx->stream) != hipSuccess) return fail(PH_EHIP);
        ph_col c{}; c.type = PH_I32; c.data = zero;
        keys.push_back(c);
    }
    // ---- aggregate arguments, evaluated positionally over the selected rows
    bool any_validity = false;
    for (auto &c : cols) any_validity |= c.validity != nullptr;
    std::vector<ph_col> args(p->g_aggs.size());
    for (size_t a = 0; a < p->g_aggs.size(); a++) {
        const ph_aggexpr &ax = p->g_aggs[a];
        if (ax.kind == PH_A_COUNT_STAR) continue;
        void *out = nullptr, *val = nullptr;
        if (ctx->pool_alloc(cnt * 8, &out) != PH_OK) return fail(PH_EHIP);
        temps.push_back(out);
        if (any_validity) { if (ctx->pool_alloc((cnt + 7) / 8 + 64, &val) != PH_OK) return fail(PH_EHIP); temps.push_back(val); }
        int rc = ph_expr_eval(ctx, cols.data(), (int32_t)cols.size(), ax.prog, ax.nprog, sel, cnt, (int64_t *)out, (uint8_t *)val);
        if (rc != PH_OK) return fail(rc);
        ph_col c{}; c.type = PH_DEC64; c.scale = p->aggs[a].scale; c.data = out; c.validity = (const uint8_t *)val;
        args[a] = c;
    }
    int rc = ph_agg_sink(p->g_agg, keys.data(), args.data(), (int32_t)args.size(), sel, cnt, 1, 0);
    if (rc == PH_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = PH_EHIP;
    cleanup();
    return rc;
}

static int generic_fetch(ph_scan_plan *p, ph_agg_result **out) {
    int64_t ng = 0;
    PH_CHECK(ph_agg_group_count(p->g_agg, &ng));
    int naggs = (int)p->aggs.size();
    size_t g = (size_t)std::max<int64_t>(ng, 1), nk = (size_t)std::max<int>(p->nkeys, 1);
    ph_agg_result *r = (ph_agg_result *)calloc(1, sizeof *r);
    r->ngroups = ng;
    r->nkeys = p->nkeys;
    r->naggs = naggs;
    r->first_row = (int64_t *)calloc(g, 8);
    r->keys = (int64_t *)calloc(g * nk, 8);
    r->sum_lo = (uint64_t *)calloc(g * naggs, 8);
    r->sum_hi = (int64_t *)calloc(g * naggs, 8);
    r->count = (uint64_t *)calloc(g * naggs, 8);
    r->scale = (int32_t *)calloc((size_t)naggs, 4);
    for (int a = 0; a < naggs; a++) r->scale[a] = p->aggs[(size_t)a].scale;
    std::vector<uint8_t> knull(g * nk);
    int rc = ph_agg_finalize(p->g_agg, (int64_t)g, r->first_row, r->keys, knull.data(), r->sum_lo, r->sum_hi, r->count);
    if (rc != PH_OK) { ph_agg_result_free(r); return rc; }
    *out = r;
    return PH_OK;
}

extern "C" int ph_scan_plan_run(ph_scan_plan *p, int64_t row_begin, int64_t row_end) {
    PH_REQUIRE(p != nullptr, "ph_scan_plan_run: plan is NULL");
    PH_REQUIRE(row_begin >= 0 && row_end >= row_begin && row_end <= p->t->nrows && row_begin % 4 == 0,
               "ph_scan_plan_run: rows [%lld,%lld) invalid (begin must be a multiple of 4, table has %lld rows)",
               (long long)row_begin, (long long)row_end, (long long)p->t->nrows);
    if (p->kind == PK_GENERIC) return generic_run(p, row_begin, row_end);
    int64_t rows = p->never ? 0 : row_end - row_begin;
    int64_t tiles = (rows + 1023) / 1024;
    int max_grid = p->max_grid;
    // tuning knob, clamped to the workgroups `partials` was allocated for (plan_alloc)
    if (const char *e = getenv("PH_SCAN_GRID")) { int g = atoi(e); if (g > 0) max_grid = std::min(g, std::max(p->max_grid, 8192)); }
    int grid = (int)std::min<int64_t>(max_grid, std::max<int64_t>(tiles, 1));
    // overflow proof: a workgroup's int64 partial sums at most rows_per_block values of
    // magnitude <= row_bound
    long double rows_per_block = (long double)((tiles + grid - 1) / grid) * 1024.0L;
    if (p->row_bound * rows_per_block >= 4.0e18L) {
        set_error("decimal overflow proof failed: per-row bound %.3Lg x %.0Lf rows per workgroup", p->row_bound, rows_per_block);
        return PH_EOVERFLOW;
    }
    p->last_rows = rows;
    p->last_grid = grid;
    p->armed_seq = 0;
    // One host-visible result per run without a publish launch (PH_SCAN_TAIL=0: merge kernel, then publish_kernel at the fetch). The one-group
    // sum kernel (Q6 shape: two words per workgroup) merges AND publishes in its last workgroup — one launch per run; for the others the merge
    // kernel's last wave publishes — two launches (a single workgroup folding 42 words x 256 workgroups was slower than the 42-wave merge launch).
    ph::ScanTail tail = {};
    static const bool fused_tail = !(getenv("PH_SCAN_TAIL") && getenv("PH_SCAN_TAIL")[0] == '0');
    const bool back_to_back = p->unfetched;
    p->unfetched = true;
    if (fused_tail && p->nacc <= ph::SCAN_TAIL_MAX_ACC && !(back_to_back && p->kind != PK_FILTER_SUMPROD)) {
        ph_ctx *ctx = p->ctx;
        if (!ctx->scan_done_dev) {   // the ticket: zero between launches
            PH_HIP(hipMalloc((void **)&ctx->scan_done_dev, 64));
            PH_HIP(hipMemsetAsync(ctx->scan_done_dev, 0, 64, ctx->stream));
        }
        tail.done = ctx->scan_done_dev;
        tail.out_lo = p->out_lo;
        tail.out_hi = p->out_hi;
        tail.nacc = p->nacc;
        if (!back_to_back) PH_CHECK(ctx->arm_publish((int64_t)p->nacc * 16, &tail.mbox, &tail.flag, &tail.seq));
        p->armed_seq = tail.seq;
    }
    if (p->kind == PK_FILTER_SUMPROD) {
        ph::FilterSumProdParams P = p->fs;
        P.row_begin = row_begin;
        P.row_end = row_begin + rows;
        P.partials = p->partials;
        P.tail = tail;
        P.tail.min_stride = 0;
        PH_CHECK(ph::launch_filter_sumprod(p->ctx, P, grid));
        if (!tail.done) PH_CHECK(ph::launch_merge_partials(p->ctx, p->partials, grid, 2, 0, p->out_lo, p->out_hi));
    } else if (p->kind == PK_JIT) {
        ph::JitParams P = p->jparams;
        P.row_begin = row_begin;
        P.row_end = row_begin + rows;
        P.partials = p->partials;
        PH_CHECK(ph::jit_launch(p->ctx, p->jkernel, P, grid));
        unsigned long long opmask = 0;
        for (int j = 0; j < p->stride; j++) opmask |= (unsigned long long)p->ops[(size_t)j] << (2 * j);
        PH_CHECK(ph::launch_merge_partials_ops(p->ctx, p->partials, grid, p->nacc, p->stride, opmask, p->out_lo, p->out_hi, &tail));
    } else {
        ph::LowcardChainParams P = p->lc;
        P.row_begin = row_begin;
        P.row_end = row_begin + rows;
        P.partials = p->partials;
        P.tail = ph::ScanTail{};
        PH_CHECK(ph::launch_lowcard_chain(p->ctx, P, grid));
        PH_CHECK(ph::launch_merge_partials(p->ctx, p->partials, grid, p->nacc, ph::LC_NACC + 1, p->out_lo, p->out_hi, &tail));
    }
    return PH_OK;
}

extern "C" void ph_agg_result_free(ph_agg_result *r) {
    if (!r) return;
    free(r->first_row); free(r->keys); free(r->sum_lo); free(r->sum_hi); free(r->count); free(r->scale); free(r->key_null);
    free(r);
}

// result rows from raw accumulator words (lo[nacc], hi[nacc]); first-row words are global ids
static int assemble(ph_scan_plan *p, const std::vector<unsigned long long> &lo, const std::vector<long long> &hi,
                    ph_agg_result **out) {
    int naggs = (int)p->aggs.size();
    struct G { int64_t first; int slot; };
    std::vector<G> groups;
    const int stride = p->stride;
    for (int s = 0; s < p->nslots; s++)
        if (lo[(size_t)s * stride + p->cnt_idx] > 0)
            groups.push_back({p->first_idx >= 0 ? (int64_t)lo[(size_t)s * stride + p->first_idx] : 0, s});
    std::sort(groups.begin(), groups.end(), [](const G &a, const G &b) { return a.first < b.first; });
    ph_agg_result *r = (ph_agg_result *)calloc(1, sizeof *r);
    size_t ng = groups.size(), nk = (size_t)p->nkeys;
    r->ngroups = (int64_t)ng;
    r->nkeys = p->nkeys;
    r->naggs = naggs;
    r->first_row = (int64_t *)calloc(ng ? ng : 1, 8);
    r->keys = (int64_t *)calloc((ng ? ng : 1) * (nk ? nk : 1), 8);
    r->sum_lo = (uint64_t *)calloc((ng ? ng : 1) * naggs, 8);
    r->sum_hi = (int64_t *)calloc((ng ? ng : 1) * naggs, 8);
    r->count = (uint64_t *)calloc((ng ? ng : 1) * naggs, 8);
    r->scale = (int32_t *)calloc((size_t)naggs, 4);
    for (int a = 0; a < naggs; a++) r->scale[a] = p->aggs[(size_t)a].scale;
    for (size_t g = 0; g < ng; g++) {
        int s = groups[g].slot;
        r->first_row[g] = groups[g].first;
        int rem = s;   // slot = dense index over the group columns' dictionaries, last column fastest
        for (int c = (int)nk - 1; c >= 0; c--) {
            r->keys[g * nk + (size_t)c] = rem % p->gcard[(size_t)c];
            rem /= p->gcard[(size_t)c];
        }
        uint64_t cnt = lo[(size_t)s * stride + p->cnt_idx];
        for (int a = 0; a < naggs; a++) {
            const auto &m = p->aggs[(size_t)a];
            r->count[g * naggs + a] = cnt;  // no NULL inputs on this path
            if (m.kind == PH_A_COUNT_STAR || m.kind == PH_A_COUNT) continue;
            size_t idx = (size_t)s * stride + (size_t)m.acc;
            r->sum_lo[g * naggs + a] = lo[idx];
            r->sum_hi[g * naggs + a] = hi[idx];
        }
    }
    *out = r;
    return PH_OK;
}

extern "C" int ph_scan_plan_fetch(ph_scan_plan *p, ph_agg_result **out) {
    PH_REQUIRE(p && out, "ph_scan_plan_fetch: bad arguments");
    if (p->kind == PK_GENERIC) {
        PH_REQUIRE(p->g_agg != nullptr, "ph_scan_plan_fetch: run the plan first");
        return generic_fetch(p, out);
    }
    // lo[nacc] and hi[nacc] are ONE allocation (plan_alloc): one host round trip for both (two cost ~45 us more per query)
    std::vector<unsigned long long> words((size_t)p->nacc * 2);
    const int armed = p->ctx->collect_armed(words.data(), (int64_t)words.size() * 8, p->armed_seq);   // 1: the mailbox has been used since
    p->armed_seq = 0;
    p->unfetched = false;
    if (armed < 0) return armed;
    if (armed == 1) PH_CHECK(p->ctx->download(words.data(), p->out_lo, (int64_t)words.size() * 8));
    std::vector<unsigned long long> lo(words.begin(), words.begin() + p->nacc);
    std::vector<long long> hi((size_t)p->nacc);
    memcpy(hi.data(), words.data() + p->nacc, (size_t)p->nacc * 8);
    return assemble(p, lo, hi, out);
}

extern "C" int ph_scan_plan_partials_dev(ph_scan_plan *p, void **dev, int32_t *nwords) {
    PH_REQUIRE(p && dev && nwords, "ph_scan_plan_partials_dev: bad arguments");
    if (p->kind == PK_GENERIC) { set_error("generic plans keep their state in a hash table, not in a fixed partial array"); return PH_EUNSUPPORTED; }
    *dev = p->out_lo;
    *nwords = 2 * p->nacc;
    return PH_OK;
}

extern "C" int ph_scan_plan_fetch_merged(ph_scan_plan *p, const uint64_t *words, int32_t nranks, ph_agg_result **out) {
    PH_REQUIRE(p && words && out && nranks >= 1, "ph_scan_plan_fetch_merged: bad arguments");
    if (p->kind == PK_GENERIC) { set_error("ph_scan_plan_fetch_merged: fused plans only"); return PH_EUNSUPPORTED; }
    size_t n = (size_t)p->nacc;
    std::vector<unsigned long long> lo(n, 0);
    std::vector<long long> hi(n, 0);
    for (size_t j = 0; j < n; j++) {
        const int wi = (int)(j % (size_t)p->stride);
        const int op = p->ops[(size_t)wi];
        if (wi == p->first_idx) {
            lo[j] = ~0ull;
            for (int32_t r = 0; r < nranks; r++) {
                const uint64_t *w = words + (size_t)r * 2 * n;
                // first-seen row across ranks: rank r's rows come after rank r-1's
                if (w[j] != 0xffffffffull && w[j] != (uint64_t)INT64_MAX) {
                    unsigned long long gfirst = ((unsigned long long)r << 40) + w[j];
                    if (gfirst < lo[j]) lo[j] = gfirst;
                }
            }
        } else if (op == 0) {
            for (int32_t r = 0; r < nranks; r++) {
                const uint64_t *w = words + (size_t)r * 2 * n;
                unsigned long long nl = lo[j] + w[j];
                hi[j] += (long long)w[n + j] + (nl < lo[j] ? 1 : 0);
                lo[j] = nl;
            }
        } else {   // MIN / MAX words are plain int64 values (identity when a rank saw no row)
            long long best = op == 1 ? INT64_MAX : INT64_MIN;
            for (int32_t r = 0; r < nranks; r++) {
                long long v = (long long)words[(size_t)r * 2 * n + j];
                best = op == 1 ? std::min(best, v) : std::max(best, v);
            }
            lo[j] = (unsigned long long)best;
            hi[j] = best < 0 ? -1 : 0;
        }
    }
    return assemble(p, lo, hi, out);
}

extern "C" int ph_scan_filter_agg(ph_ctx *ctx, const ph_table *t, int64_t row_begin, int64_t row_end,
                                  const ph_pred *preds, int32_t npreds, const int32_t *group_cols,
                                  int32_t ngroup_cols, const ph_aggexpr *aggs, int32_t naggs,
                                  ph_agg_result **out) {
    ph_scan_plan *p = nullptr;
    PH_CHECK(ph_scan_plan_create(ctx, t, preds, npreds, group_cols, ngroup_cols, aggs, naggs, &p));
    int rc = ph_scan_plan_run(p, row_begin, row_end);
    if (rc == PH_OK) rc = ph_scan_plan_fetch(p, out);
    ph_scan_plan_free(p);
    return rc;
}

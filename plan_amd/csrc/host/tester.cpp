// host_tester — drives the GPU executors through the reference's pull protocol (the loop of
// execOps, pkg/compute/executor.go:151-188) on generated TPC-H data and prints the result in the
// reference's text format (headline "#\t..." of execQuery, executor_bench.go:229-238; rows via
// Chunk.SaveToFile). The role `tester tpch1g --query_id N` plays for the reference.
//   host_tester roundtrip | formats
//   host_tester semi|anti|left|order|having|cross <sf_num> <sf_den>
//   host_tester substr <sf_num> <sf_den> <offset> <length>
//   host_tester q1|q6|q3|q9 <sf_num> <sf_den> [stub|resident]
//   host_tester q3|q9 <sf_num> <sf_den> resident [repeat]     the whole subtree as ONE gpuResidentPlanExecutor (ph_plan)
//   host_tester tpch <query_id> <sf_num> <sf_den> [repeat]    any query tpch_plans.cpp has a resident plan for
//   host_tester concurrent <sf_num> <sf_den> [iterations]     the tables on ONE context, Q3 and Q9 from two threads on two others at once
//   host_tester ranks <n> <query_id> <sf_num> <sf_den>         the database split over n ranks (threads of this process, the in-process transport);
//                                                              every rank runs the query over its shard through ph_plan_set_comm; rank 0 prints
// The resident-plan forms print "Query N took <dur> success" per repeat on stderr, like Run (executor_bench.go:126-137).
// q1 / q3 / q9 run the whole plan tail on the library: gpuOrderExecutor (ORDER BY) and limitExecutor
// (LIMIT), so their output is the reference's result file byte for byte with no sorting here.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>

#include "operator_exec.h"
#include "tpch_plans.h"
#include "tpchgen.h"

using namespace plan;

static void die(const std::string &m) { fprintf(stderr, "host_tester: %s\n", m.c_str()); exit(1); }

struct Lineitem {
    int64_t n = 0;
    std::vector<int64_t> okey, ext, disc, tax;
    std::vector<int32_t> qty, ship;
    std::vector<uint8_t> rf, ls;
};

static Lineitem gen_lineitem(int64_t num, int64_t den) {
    Lineitem L;
    int64_t no = tpchgen_orders_count(num, den);
    L.n = tpchgen_lineitem_count(num, den, 0, no);
    size_t n = (size_t)L.n;
    L.okey.resize(n); L.ext.resize(n); L.disc.resize(n); L.tax.resize(n); L.qty.resize(n); L.ship.resize(n); L.rf.resize(n); L.ls.resize(n);
    tpchgen_lineitem_cols c{};
    c.l_orderkey = L.okey.data(); c.l_quantity = L.qty.data(); c.l_extendedprice = L.ext.data(); c.l_discount = L.disc.data();
    c.l_tax = L.tax.data(); c.l_returnflag = L.rf.data(); c.l_linestatus = L.ls.data(); c.l_shipdate = L.ship.data();
    tpchgen_lineitem(num, den, 0, no, &c);
    return L;
}

// lineitem as the scan would hand it over: reference in-memory types, 2048 rows per chunk
static std::vector<LType> lineitem_types() {
    return {IntegerType(), DecimalType(15, 2), DecimalType(15, 2), DecimalType(15, 2), VarcharType(), VarcharType(),
            DateType(), BigintType()};
}

static bool fill_lineitem(const Lineitem &L, int64_t *pos, Chunk *out) {
    if (*pos >= L.n) return false;
    int card = (int)std::min<int64_t>(DefaultVectorSize, L.n - *pos);
    out->Init(lineitem_types(), DefaultVectorSize);
    for (int i = 0; i < card; i++) {
        size_t r = (size_t)(*pos + i);
        out->Data[0]->Slice<int32_t>()[i] = L.qty[r];
        out->Data[1]->Slice<Decimal>()[i] = DecimalFromUnscaled(L.ext[r], 2);
        out->Data[2]->Slice<Decimal>()[i] = DecimalFromUnscaled(L.disc[r], 2);
        out->Data[3]->Slice<Decimal>()[i] = DecimalFromUnscaled(L.tax[r], 2);
        out->Data[4]->SetString(i, TPCHGEN_RETURNFLAG_DICT[L.rf[r]], 1);
        out->Data[5]->SetString(i, TPCHGEN_LINESTATUS_DICT[L.ls[r]], 1);
        out->Data[6]->Slice<Date>()[i] = DateFromDays(L.ship[r]);
        out->Data[7]->Slice<int64_t>()[i] = L.okey[r];
    }
    out->SetCard(card);
    *pos += card;
    return true;
}

static std::unique_ptr<OperatorExec> lineitem_source(const Lineitem &L, bool stub) {
    if (stub) {  // through Chunk.Serialize / Deserialize, like Test_q18_with_stub
        std::string blob;
        int64_t pos = 0;
        Chunk c;
        while (fill_lineitem(L, &pos, &c)) c.Serialize(&blob);
        return std::unique_ptr<OperatorExec>(new stubExecutor(lineitem_types(), std::move(blob)));
    }
    auto pos = std::make_shared<int64_t>(0);
    return std::unique_ptr<OperatorExec>(new sourceExecutor(lineitem_types(), [&L, pos](Chunk *out) { return fill_lineitem(L, pos.get(), out); }));
}

// a table as the scan would hand it over: typed column arrays -> 2048-row chunks of the reference's
// in-memory types (Decimal / Date / String structs)
struct SrcCol {
    LType type;
    const void *data = nullptr;                 // int32 / int64 / unscaled int64 / days / uint8 codes
    const char *const *dict = nullptr;          // VARCHAR from dictionary codes
    std::function<std::string(int64_t)> str;    // VARCHAR from a generator
};

static std::unique_ptr<OperatorExec> table_source(std::vector<SrcCol> cols, int64_t n) {
    std::vector<LType> types;
    for (auto &c : cols) types.push_back(c.type);
    auto pos = std::make_shared<int64_t>(0);
    return std::unique_ptr<OperatorExec>(new sourceExecutor(types, [cols, types, n, pos](Chunk *out) {
        if (*pos >= n) return false;
        int card = (int)std::min<int64_t>(DefaultVectorSize, n - *pos);
        out->Init(types, DefaultVectorSize);
        for (size_t c = 0; c < cols.size(); c++) {
            Vector &v = *out->Data[c];
            for (int i = 0; i < card; i++) {
                size_t r = (size_t)(*pos + i);
                switch (types[c].GetInternalType()) {
                case PT_INT32: v.Slice<int32_t>()[i] = ((const int32_t *)cols[c].data)[r]; break;
                case PT_INT64: v.Slice<int64_t>()[i] = ((const int64_t *)cols[c].data)[r]; break;
                case PT_DECIMAL: v.Slice<Decimal>()[i] = DecimalFromUnscaled(((const int64_t *)cols[c].data)[r], types[c].Scale); break;
                case PT_DATE: v.Slice<Date>()[i] = DateFromDays(((const int32_t *)cols[c].data)[r]); break;
                case PT_VARCHAR: {
                    std::string sv = cols[c].str ? cols[c].str((int64_t)r) : std::string(cols[c].dict[((const uint8_t *)cols[c].data)[r]]);
                    v.SetString(i, sv.data(), (int64_t)sv.size());
                    break;
                }
                default: break;
                }
            }
        }
        out->SetCard(card);
        *pos += card;
        return true;
    }));
}

static ph_rpn X_COL(int c) { return ph_rpn{PH_X_COL, c, 0, 0}; }
static ph_rpn X_CONST(int64_t v, int s) { return ph_rpn{PH_X_CONST, -1, v, s}; }
static ph_rpn X_OP(int op) { return ph_rpn{op, -1, 0, 0}; }

// pull loop of execOps; returns the rows as text lines
static std::vector<std::string> run(OperatorExec *root) {
    std::string e = root->Init();
    if (!e.empty()) die("Init: " + e);
    std::vector<std::string> lines;
    for (;;) {
        Chunk out;
        std::string err;
        OperatorResult r = root->Execute(nullptr, &out, &err);
        if (r == InvalidOpResult) die("Execute: " + err);
        if (r == Done) break;
        std::string text;
        out.AppendText(&text);
        size_t p = 0;
        while (p < text.size()) {
            size_t q = text.find('\n', p);
            lines.push_back(text.substr(p, q - p));
            p = q + 1;
        }
    }
    e = root->Close();
    if (!e.empty()) die("Close: " + e);
    return lines;
}

static void print(int ncols, const std::vector<std::string> &lines) {
    std::string head = "#";
    for (int i = 1; i < ncols; i++) head += "\t";
    printf("%s\n", head.c_str());
    for (auto &l : lines) printf("%s\n", l.c_str());
}

static int roundtrip() {
    Chunk c;
    c.Init({IntegerType(), BigintType(), DecimalType(15, 2), VarcharType(), DateType(), DoubleType(), HugeintType()}, DefaultVectorSize);
    for (int i = 0; i < 5; i++) {
        c.Data[0]->Slice<int32_t>()[i] = i - 2;
        c.Data[1]->Slice<int64_t>()[i] = (int64_t)i * 10000000000ll;
        c.Data[2]->Slice<Decimal>()[i] = DecimalFromUnscaled(1050 * i - 7, 2);
        std::string s = i == 3 ? "" : std::string((size_t)i + 1, (char)('a' + i));
        c.Data[3]->SetString(i, s.data(), (int64_t)s.size());
        c.Data[4]->Slice<Date>()[i] = DateFromDays(8035 + 400 * i);
        c.Data[5]->Slice<double>()[i] = 25.5 + i / 3.0;
        c.Data[6]->Slice<Hugeint>()[i] = Hugeint{(uint64_t)i * 7, i == 4 ? -1 : 0};
    }
    c.Data[0]->Mask.SetInvalid(1, DefaultVectorSize);
    c.Data[3]->Mask.SetInvalid(2, DefaultVectorSize);
    c.SetCard(5);
    std::string blob, err, a, b;
    c.Serialize(&blob);
    Chunk d;
    size_t pos = 0;
    if (!d.Deserialize(blob, &pos, &err)) die(err);
    c.AppendText(&a);
    d.AppendText(&b);
    if (a != b || pos != blob.size()) die("round trip mismatch");
    printf("%s", b.c_str());
    return 0;
}

// every physical format through ToUnifiedFormat / SliceIndice / Serialize (vector_format.go:64-97)
static int formats() {
    Chunk c;
    c.Init({IntegerType(), VarcharType(), DecimalType(15, 2), BigintType(), IntegerType()}, DefaultVectorSize);
    c.Data[0]->Sequence(5, 3, 6);                       // 5 8 11 14 17 20
    c.Data[1]->_PhyFormat = PF_CONST;
    c.Data[1]->SetString(0, "k", 1);
    c.Data[2]->SetConstNull();
    c.Data[3]->Sequence(-10000000000ll, 10000000000ll, 6);
    for (int i = 0; i < 6; i++) c.Data[4]->Slice<int32_t>()[i] = 100 + i;
    c.Data[4]->Mask.SetInvalid(3, DefaultVectorSize);
    c.SetCard(6);
    std::string out;
    c.AppendText(&out);
    auto pick = [](std::initializer_list<int64_t> l) {
        auto s = std::make_shared<SelectVector>();
        s->identity = false;
        s->SelVec.assign(l);
        return s;
    };
    std::vector<int> all = {0, 1, 2, 3, 4};
    Chunk d, e, f;
    std::vector<LType> types = {IntegerType(), VarcharType(), DecimalType(15, 2), BigintType(), IntegerType()};
    d.Init(types, DefaultVectorSize);
    d.SliceIndice(c, pick({4, 1, 3, 0}), 4, 0, all);   // DICT over SEQUENCE / CONST / FLAT
    out += "--\n";
    d.AppendText(&out);
    e.Init(types, DefaultVectorSize);
    e.SliceIndice(d, pick({2, 0}), 2, 0, all);         // selection of a selection
    out += "--\n";
    e.AppendText(&out);
    std::string blob, err;
    e.Serialize(&blob);
    size_t pos = 0;
    if (!f.Deserialize(blob, &pos, &err)) die(err);
    out += "--\n";
    f.AppendText(&out);
    printf("%s", out.c_str());
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "roundtrip")) return roundtrip();
    if (argc >= 2 && !strcmp(argv[1], "formats")) return formats();
    if (argc < 4) die("usage: host_tester roundtrip | q1|q6|q3 <sf_num> <sf_den> [stub]");
    std::string q = argv[1];
    if (q == "ranks") {
        // The N-rank split behind the operator interface: n threads = n ranks, each with its own ctx (device 0: one GPU here) and its SHARD of the
        // database, joined by the in-process transport. Every rank builds the same executor tree, announces the communicator (SetComm ->
        // ph_plan_set_comm) and pulls it; every rank must produce the same, complete result. Prints rank 0's text.
        if (argc < 6) die("usage: host_tester ranks <n> <query_id> <sf_num> <sf_den>");
        const int n = atoi(argv[2]), id = atoi(argv[3]);
        const int64_t num = atoll(argv[4]), den = atoll(argv[5]);
        if (n < 1 || n > 8 || num <= 0 || den <= 0) die("ranks: 1..8 ranks, positive scale factor");
        ph_local_group *grp = nullptr;
        if (ph_local_group_create(n, &grp) != PH_OK) die(std::string("ph_local_group_create: ") + ph_last_error());
        std::vector<std::vector<std::string>> result((size_t)n);
        std::vector<std::string> fail((size_t)n), explain((size_t)n);
        std::vector<int> ncols((size_t)n, 0);
        auto work = [&](int r) {
            ph_ctx *ctx = nullptr;
            ph_comm *comm = nullptr;
            if (ph_ctx_create(0, &ctx) != PH_OK || ph_comm_init_local(ctx, grp, r, &comm) != PH_OK) { fail[(size_t)r] = std::string("ctx / comm: ") + ph_last_error(); return; }
            {
                TpchDatabase db;
                std::string e = db.Load(ctx, num, den, r, n);
                TpchQuery tq;
                if (e.empty()) e = BuildTpchQuery(db, id, &tq);
                // (a rank that failed BEFORE the first collective would leave the others waiting in it: every rank reaches the query or none does)
                int64_t bad = e.empty() ? 0 : 1;
                ph_comm_allreduce_i64(comm, &bad, 1, PH_RED_MAX);
                if (bad) fail[(size_t)r] = e.empty() ? "another rank failed to load" : e;
                else {
                    auto t0 = std::chrono::steady_clock::now();
                    e = RunTpchQuery(ctx, tq, &result[(size_t)r], &explain[(size_t)r], comm);
                    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (!e.empty()) fail[(size_t)r] = e;
                    else if (r == 0) fprintf(stderr, "Query %d took %.3fms success (%d ranks)\n", id, ms, n);
                    ncols[(size_t)r] = tq.ncols;
                }
            }
            ph_comm_destroy(comm);
            ph_ctx_destroy(ctx);
        };
        std::vector<std::thread> ts;
        for (int r = 0; r < n; r++) ts.emplace_back(work, r);
        for (auto &t : ts) t.join();
        ph_local_group_free(grp);
        int rc = 0;
        for (int r = 0; r < n; r++) if (!fail[(size_t)r].empty()) { fprintf(stderr, "host_tester ranks: rank %d: %s\n", r, fail[(size_t)r].c_str()); rc = 1; }
        for (int r = 1; r < n && rc == 0; r++) if (result[(size_t)r] != result[0]) { fprintf(stderr, "host_tester ranks: rank %d's result differs from rank 0's\n", r); rc = 1; }
        if (rc == 0) { fprintf(stderr, "%s", explain[0].c_str()); print(ncols[0], result[0]); }
        return rc;
    }
    if (q == "concurrent") {
        // Concurrent queries over SHARED resident tables (SURVEY.md §8(b) threading; the psql server path, cmd/main/main.go:71-122): the
        // database is loaded on context A; two threads own a context each and run whole queries — executors built, pulled, closed — over
        // A's tables at the same time: Q3 on B, Q9 on C (whose late materialisation makes the library build a co-located copy of lineitem
        // columns while Q3 reads the same table). Prints Q3's text, "--", Q9's text; every iteration of a thread must give the same lines.
        int64_t num = atoll(argv[2]), den = atoll(argv[3]);
        if (num <= 0 || den <= 0) { fprintf(stderr, "scale factor: <num> <den> must both be positive integers\n"); return 2; }
        const int iters = argc > 4 ? std::max(atoi(argv[4]), 1) : 4;
        ph_ctx *a = nullptr, *bc[2] = {nullptr, nullptr};
        if (ph_ctx_create(0, &a) != PH_OK || ph_ctx_create(0, &bc[0]) != PH_OK || ph_ctx_create(0, &bc[1]) != PH_OK) die(std::string("ph_ctx_create: ") + ph_last_error());
        int rc = 0;
        {
            TpchDatabase db;
            std::string e = db.Load(a, num, den);
            if (!e.empty()) die(e);
            const int ids[2] = {3, 9};
            TpchQuery tq[2];
            for (int k = 0; k < 2; k++) { e = BuildTpchQuery(db, ids[k], &tq[k]); if (!e.empty()) die(e); }
            std::vector<std::string> result[2];
            std::string fail[2];
            auto work = [&](int k) {
                for (int it = 0; it < iters && fail[k].empty(); it++) {
                    std::vector<std::string> lines;
                    std::string explain;
                    auto t0 = std::chrono::steady_clock::now();
                    std::string err = RunTpchQuery(bc[k], tq[k], &lines, &explain);
                    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (!err.empty()) { fail[k] = err; break; }
                    fprintf(stderr, "Query %d took %.3fms success\n", ids[k], ms);
                    if (it == 0) result[k] = lines;
                    else if (lines != result[k]) fail[k] = "iteration " + std::to_string(it) + " of Q" + std::to_string(ids[k]) + " differs from the first";
                }
            };
            std::thread t0(work, 0), t1(work, 1);
            t0.join(); t1.join();
            for (int k = 0; k < 2; k++) if (!fail[k].empty()) { fprintf(stderr, "host_tester concurrent: Q%d: %s\n", ids[k], fail[k].c_str()); rc = 1; }
            int32_t cols5[5] = {L_SUPPKEY, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_ORDERKEY};
            fprintf(stderr, "lineitem co-located copies: %lld bytes (the five Q9 columns covered: %d)\n", (long long)ph_table_colocate_bytes(db.lineitem.table),
                    (int)ph_table_colocated(db.lineitem.table, 5, cols5));
            if (rc == 0) { print(tq[0].ncols, result[0]); printf("--\n"); print(tq[1].ncols, result[1]); }
        }
        ph_ctx_destroy(bc[0]); ph_ctx_destroy(bc[1]); ph_ctx_destroy(a);
        return rc;
    }
    {   // resident plans: the whole operator subtree behind one OperatorExec
        int id = 0, a = 2;
        if (q == "tpch" && argc >= 5) { id = atoi(argv[2]); a = 3; }
        else if ((q == "q3" || q == "q9") && argc > 4 && !strcmp(argv[4], "resident")) id = atoi(q.c_str() + 1);
        if (id > 0) {
            int64_t num = atoll(argv[a]), den = atoll(argv[a + 1]);
            if (num <= 0 || den <= 0) { fprintf(stderr, "scale factor: <num> <den> must both be positive integers\n"); return 2; }
            int repeat = argc > a + (q == "tpch" ? 2 : 3) ? atoi(argv[a + (q == "tpch" ? 2 : 3)]) : 1;
            ph_ctx *ctx = nullptr;
            if (ph_ctx_create(0, &ctx) != PH_OK) die(std::string("ph_ctx_create: ") + ph_last_error());
            {
                TpchDatabase db;
                std::string e = db.Load(ctx, num, den);
                if (!e.empty()) die(e);
                fprintf(stderr, "loaded SF%g: generate %.2f s, load %.2f s (%.2f GB, %.1f GB/s)\n", (double)num / (double)den, db.generate_s, db.load_s,
                        (double)db.loaded_bytes / 1e9, (double)db.loaded_bytes / 1e9 / std::max(db.load_s, 1e-9));
                TpchQuery tq;
                e = BuildTpchQuery(db, id, &tq);
                if (!e.empty()) die(e);
                std::vector<std::string> lines;
                std::string explain;
                for (int r = 0; r < std::max(repeat, 1); r++) {
                    auto t0 = std::chrono::steady_clock::now();
                    e = RunTpchQuery(ctx, tq, &lines, &explain);
                    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    if (!e.empty()) die(e);
                    fprintf(stderr, "Query %d took %.3fms success\n", id, ms);
                }
                fprintf(stderr, "%s", explain.c_str());
                print(tq.ncols, lines);
            }
            ph_ctx_destroy(ctx);
            return 0;
        }
    }
    int64_t num = atoll(argv[2]), den = atoll(argv[3]);
    if (num <= 0 || den <= 0) { fprintf(stderr, "scale factor: <num> <den> must both be positive integers\n"); return 2; }
    bool stub = argc > 4 && !strcmp(argv[4], "stub");
    ph_ctx *ctx = nullptr;
    if (ph_ctx_create(0, &ctx) != PH_OK) die(std::string("ph_ctx_create: ") + ph_last_error());
    Lineitem L;
    if (q == "q1" || q == "q6" || q == "q3") L = gen_lineitem(num, den);
    auto scan = lineitem_source(L, stub);
    auto lit_date = [](int32_t d) { Literal k; k.kind = Literal::DateDays; k.i = d; return k; };

    bool resident = argc > 4 && !strcmp(argv[4], "resident");
    if (resident && (q == "q1" || q == "q6")) {
        // the measured mode behind the same interface: columns loaded once with ph_table_create,
        // Agg <- Scan(filter) collapsed into a ph_scan_plan (fused kernel)
        std::string rfd, lsd;
        for (auto s : TPCHGEN_RETURNFLAG_DICT) { rfd += s; rfd.push_back('\0'); }
        for (auto s : TPCHGEN_LINESTATUS_DICT) { lsd += s; lsd.push_back('\0'); }
        ph_col hc[7] = {};
        hc[0].type = PH_I32; hc[0].data = L.qty.data();
        hc[1].type = PH_DEC64; hc[1].scale = 2; hc[1].data = L.ext.data();
        hc[2].type = PH_DEC64; hc[2].scale = 2; hc[2].data = L.disc.data();
        hc[3].type = PH_DEC64; hc[3].scale = 2; hc[3].data = L.tax.data();
        hc[4].type = PH_CODE8; hc[4].data = L.rf.data(); hc[4].aux = rfd.data(); hc[4].aux_bytes = (int64_t)rfd.size();
        hc[5].type = PH_CODE8; hc[5].data = L.ls.data(); hc[5].aux = lsd.data(); hc[5].aux_bytes = (int64_t)lsd.size();
        hc[6].type = PH_DATE; hc[6].data = L.ship.data();
        ph_table *tab = nullptr;
        if (ph_table_create(ctx, 7, hc, L.n, &tab) != PH_OK) die(std::string("ph_table_create: ") + ph_last_error());
        std::vector<ResidentColumn> rc = {{IntegerType(), {}}, {DecimalType(15, 2), {}}, {DecimalType(15, 2), {}}, {DecimalType(15, 2), {}},
                                          {VarcharType(), {"A", "N", "R"}}, {VarcharType(), {"F", "O"}}, {DateType(), {}}};
        if (q == "q1") {
            Compare c{6, PH_LE, lit_date(tpchgen_days_from_civil(1998, 12, 1) - 112)};
            std::vector<ph_rpn> dp = {X_COL(1), X_CONST(1, 0), X_COL(2), X_OP(PH_X_SUB), X_OP(PH_X_MUL)};
            std::vector<ph_rpn> ch = dp;
            ch.push_back(X_CONST(1, 0)); ch.push_back(X_COL(3)); ch.push_back(X_OP(PH_X_ADD)); ch.push_back(X_OP(PH_X_MUL));
            std::vector<AggExpr> aggs = {{PH_A_SUM, {X_COL(0)}}, {PH_A_SUM, {X_COL(1)}}, {PH_A_SUM, dp}, {PH_A_SUM, ch},
                                         {PH_A_AVG, {X_COL(0)}}, {PH_A_AVG, {X_COL(1)}}, {PH_A_AVG, {X_COL(2)}}, {PH_A_COUNT_STAR, {}}};
            gpuScanAggExecutor agg(ctx, tab, rc, {c}, {4, 5}, aggs);
            if (!agg.Init().empty()) die("agg init");
            gpuOrderExecutor ord(ctx, {{0, false}, {1, false}}, &agg);   // ORDER BY l_returnflag, l_linestatus
            print(10, run(&ord));
            agg.Close();
        } else {
            Literal lo, hi, qty;
            lo.kind = hi.kind = Literal::Float;
            lo.f = (double)(0.03f - 0.01f);
            hi.f = (double)(0.03f + 0.01f);
            qty.kind = Literal::Int; qty.i = 24;
            std::vector<Compare> conj = {{6, PH_GE, lit_date(tpchgen_days_from_civil(1994, 1, 1))},
                                         {6, PH_LT, lit_date(tpchgen_days_from_civil(1995, 1, 1))},
                                         {2, PH_GE, lo}, {2, PH_LE, hi}, {0, PH_LT, qty}};
            gpuScanAggExecutor agg(ctx, tab, rc, conj, {}, {{PH_A_SUM, {X_COL(1), X_COL(2), X_OP(PH_X_MUL)}}});
            print(1, run(&agg));
        }
        ph_table_free(tab);
    } else if (q == "q1") {
        // Order <- Project <- Agg <- Scan(filter l_shipdate <= date '1998-12-01' - 112 days)
        Compare c{6, PH_LE, lit_date(tpchgen_days_from_civil(1998, 12, 1) - 112)};
        gpuFilterExecutor filt(ctx, {c}, scan.get());
        std::vector<ph_rpn> dp = {X_COL(1), X_CONST(1, 0), X_COL(2), X_OP(PH_X_SUB), X_OP(PH_X_MUL)};
        std::vector<ph_rpn> ch = dp;
        ch.push_back(X_CONST(1, 0)); ch.push_back(X_COL(3)); ch.push_back(X_OP(PH_X_ADD)); ch.push_back(X_OP(PH_X_MUL));
        std::vector<AggExpr> aggs = {{PH_A_SUM, {X_COL(0)}}, {PH_A_SUM, {X_COL(1)}}, {PH_A_SUM, dp}, {PH_A_SUM, ch},
                                     {PH_A_AVG, {X_COL(0)}}, {PH_A_AVG, {X_COL(1)}}, {PH_A_AVG, {X_COL(2)}}, {PH_A_COUNT_STAR, {}}};
        gpuAggExecutor agg(ctx, {4, 5}, aggs, &filt);
        if (!filt.Init().empty() || !agg.Init().empty()) die("init");
        gpuOrderExecutor ord(ctx, {{0, false}, {1, false}}, &agg);   // ORDER BY l_returnflag, l_linestatus
        print(10, run(&ord));
        agg.Close();
        filt.Close();
    } else if (q == "q6") {
        Literal lo, hi, qty;
        lo.kind = hi.kind = Literal::Float;
        lo.f = (double)(0.03f - 0.01f);   // folded in float32 (rule_constant_folding.go:34-70)
        hi.f = (double)(0.03f + 0.01f);
        qty.kind = Literal::Int; qty.i = 24;
        std::vector<Compare> conj = {{6, PH_GE, lit_date(tpchgen_days_from_civil(1994, 1, 1))},
                                     {6, PH_LT, lit_date(tpchgen_days_from_civil(1995, 1, 1))},
                                     {2, PH_GE, lo}, {2, PH_LE, hi}, {0, PH_LT, qty}};
        gpuFilterExecutor filt(ctx, conj, scan.get());
        gpuAggExecutor agg(ctx, {}, {{PH_A_SUM, {X_COL(1), X_COL(2), X_OP(PH_X_MUL)}}}, &filt);
        if (!filt.Init().empty()) die("filter init");
        print(1, run(&agg));
        filt.Close();
    } else if (q == "q3") {
        int32_t date = tpchgen_days_from_civil(1995, 3, 29);
        // customer
        int64_t nc = tpchgen_customer_count(num, den), no = tpchgen_orders_count(num, den);
        std::vector<int32_t> ckey((size_t)nc), ocust((size_t)no), odate((size_t)no), oprio((size_t)no);
        std::vector<uint8_t> cseg((size_t)nc);
        std::vector<int64_t> okey((size_t)no);
        tpchgen_customer_cols cc{}; cc.c_custkey = ckey.data(); cc.c_mktsegment = cseg.data();
        tpchgen_customer(num, den, 0, nc, &cc);
        tpchgen_orders_cols oc{}; oc.o_orderkey = okey.data(); oc.o_custkey = ocust.data(); oc.o_orderdate = odate.data(); oc.o_shippriority = oprio.data();
        tpchgen_orders(num, den, 0, no, &oc);
        auto cpos = std::make_shared<int64_t>(0), opos = std::make_shared<int64_t>(0);
        std::vector<LType> ctypes = {IntegerType(), VarcharType()}, otypes = {BigintType(), IntegerType(), DateType(), IntegerType()};
        sourceExecutor csrc(ctypes, [&](Chunk *out) {
            if (*cpos >= nc) return false;
            int card = (int)std::min<int64_t>(DefaultVectorSize, nc - *cpos);
            out->Init(ctypes, DefaultVectorSize);
            for (int i = 0; i < card; i++) {
                size_t r = (size_t)(*cpos + i);
                out->Data[0]->Slice<int32_t>()[i] = ckey[r];
                const char *s = TPCHGEN_MKTSEGMENT_DICT[cseg[r]];
                out->Data[1]->SetString(i, s, (int64_t)strlen(s));
            }
            out->SetCard(card);
            *cpos += card;
            return true;
        });
        sourceExecutor osrc(otypes, [&](Chunk *out) {
            if (*opos >= no) return false;
            int card = (int)std::min<int64_t>(DefaultVectorSize, no - *opos);
            out->Init(otypes, DefaultVectorSize);
            for (int i = 0; i < card; i++) {
                size_t r = (size_t)(*opos + i);
                out->Data[0]->Slice<int64_t>()[i] = okey[r];
                out->Data[1]->Slice<int32_t>()[i] = ocust[r];
                out->Data[2]->Slice<Date>()[i] = DateFromDays(odate[r]);
                out->Data[3]->Slice<int32_t>()[i] = oprio[r];
            }
            out->SetCard(card);
            *opos += card;
            return true;
        });
        Literal seg; seg.kind = Literal::Str; seg.s = "HOUSEHOLD";
        gpuFilterExecutor cf(ctx, {{1, PH_EQ, seg}}, &csrc);
        gpuFilterExecutor of(ctx, {{2, PH_LT, lit_date(date)}}, &osrc);
        gpuFilterExecutor lf(ctx, {{6, PH_GT, lit_date(date)}}, scan.get());
        // join1: orders (probe) x customer (build) on o_custkey = c_custkey
        gpuJoinExecutor j1(ctx, &of, &cf, {1}, {0}, {});
        // join2: lineitem (probe) x join1 (build) on l_orderkey = o_orderkey, payload o_orderdate, o_shippriority
        gpuJoinExecutor j2(ctx, &lf, &j1, {7}, {0}, {2, 3});
        // group by l_orderkey (7), o_orderdate (8), o_shippriority (9); sum(ext*(1-disc))
        gpuAggExecutor agg(ctx, {7, 8, 9}, {{PH_A_SUM, {X_COL(1), X_CONST(1, 0), X_COL(2), X_OP(PH_X_SUB), X_OP(PH_X_MUL)}}}, &j2);
        // the aggregate's output expressions put the row in select-list order: l_orderkey, revenue,
        // o_orderdate, o_shippriority (executor_aggr.go:229-247); then ORDER BY revenue DESC,
        // o_orderdate LIMIT 10 (orderExecutor + limitExecutor), all through the library
        agg.SetOutputs({ProjExpr::Col(0), ProjExpr::Col(3), ProjExpr::Col(1), ProjExpr::Col(2)});
        for (OperatorExec *e : std::vector<OperatorExec *>{&cf, &of, &lf, &j1, &j2})
            if (!e->Init().empty()) die("init");
        bool all = argc > 4 && !strcmp(argv[4], "groups");
        if (all) print(4, run(&agg));   // every group, unordered (what the aggregate itself emits)
        else {
            if (!agg.Init().empty()) die("agg init");
            gpuOrderExecutor ord(ctx, {{1, true}, {2, false}}, &agg);
            limitExecutor lim(10, 0, &ord);
            print(4, run(&lim));
        }
        for (OperatorExec *e : std::vector<OperatorExec *>{&cf, &of, &lf, &j1, &j2, &agg}) e->Close();
    } else if (q == "semi" || q == "anti") {
        // customer SEMI/ANTI JOIN orders ON c_custkey = o_custkey: customers with / without orders
        int64_t nc = tpchgen_customer_count(num, den), no = tpchgen_orders_count(num, den);
        std::vector<int32_t> ckey((size_t)nc), ocust((size_t)no);
        tpchgen_customer_cols cc{}; cc.c_custkey = ckey.data();
        tpchgen_customer(num, den, 0, nc, &cc);
        tpchgen_orders_cols oc{}; oc.o_custkey = ocust.data();
        tpchgen_orders(num, den, 0, no, &oc);
        auto cpos = std::make_shared<int64_t>(0), opos = std::make_shared<int64_t>(0);
        auto int_source = [](const std::vector<int32_t> &v, std::shared_ptr<int64_t> pos) {
            return [&v, pos](Chunk *out) {
                int64_t n = (int64_t)v.size();
                if (*pos >= n) return false;
                int card = (int)std::min<int64_t>(DefaultVectorSize, n - *pos);
                out->Init({IntegerType()}, DefaultVectorSize);
                memcpy(out->Data[0]->Data.data(), v.data() + *pos, (size_t)card * 4);
                out->SetCard(card);
                *pos += card;
                return true;
            };
        };
        sourceExecutor csrc({IntegerType()}, int_source(ckey, cpos)), osrc({IntegerType()}, int_source(ocust, opos));
        gpuJoinExecutor j(ctx, &csrc, &osrc, {0}, {0}, {}, 512, q == "semi" ? JoinSemi : JoinAnti);
        print(1, run(&j));
    } else if (q == "left") {
        // customer LEFT JOIN orders ON c_custkey = o_custkey, payload o_orderkey: one row per order
        // plus one (custkey, NULL) row per customer that never ordered
        int64_t nc = tpchgen_customer_count(num, den), no = tpchgen_orders_count(num, den);
        std::vector<int32_t> ckey((size_t)nc), ocust((size_t)no);
        std::vector<int64_t> okey((size_t)no);
        tpchgen_customer_cols cc{}; cc.c_custkey = ckey.data();
        tpchgen_customer(num, den, 0, nc, &cc);
        tpchgen_orders_cols oc{}; oc.o_custkey = ocust.data(); oc.o_orderkey = okey.data();
        tpchgen_orders(num, den, 0, no, &oc);
        int64_t cpos = 0, opos = 0;
        sourceExecutor csrc({IntegerType()}, [&](Chunk *out) {
            if (cpos >= nc) return false;
            int card = (int)std::min<int64_t>(DefaultVectorSize, nc - cpos);
            out->Init({IntegerType()}, DefaultVectorSize);
            memcpy(out->Data[0]->Data.data(), ckey.data() + cpos, (size_t)card * 4);
            out->SetCard(card);
            cpos += card;
            return true;
        });
        sourceExecutor osrc({IntegerType(), BigintType()}, [&](Chunk *out) {
            if (opos >= no) return false;
            int card = (int)std::min<int64_t>(DefaultVectorSize, no - opos);
            out->Init({IntegerType(), BigintType()}, DefaultVectorSize);
            memcpy(out->Data[0]->Data.data(), ocust.data() + opos, (size_t)card * 4);
            memcpy(out->Data[1]->Data.data(), okey.data() + opos, (size_t)card * 8);
            out->SetCard(card);
            opos += card;
            return true;
        });
        gpuJoinExecutor j(ctx, &csrc, &osrc, {0}, {0}, {1}, 512, JoinLeft);
        print(2, run(&j));
    } else if (q == "order") {
        // SELECT c_mktsegment, c_custkey FROM customer ORDER BY c_mktsegment DESC, c_custkey % 97, c_custkey DESC
        // (a VARCHAR key, an INTEGER key with many ties, a tie-breaker): gpuOrderExecutor over 2048-row chunks
        int64_t nc = tpchgen_customer_count(num, den);
        std::vector<int32_t> ckey((size_t)nc);
        std::vector<uint8_t> seg((size_t)nc);
        tpchgen_customer_cols cc{}; cc.c_custkey = ckey.data(); cc.c_mktsegment = seg.data();
        tpchgen_customer(num, den, 0, nc, &cc);
        int64_t pos = 0;
        std::vector<LType> types = {VarcharType(), IntegerType(), IntegerType()};
        sourceExecutor src(types, [&](Chunk *out) {
            if (pos >= nc) return false;
            int card = (int)std::min<int64_t>(DefaultVectorSize, nc - pos);
            out->Init(types, DefaultVectorSize);
            for (int i = 0; i < card; i++) {
                const char *sname = TPCHGEN_MKTSEGMENT_DICT[seg[(size_t)(pos + i)]];
                out->Data[0]->SetString(i, sname, (int64_t)strlen(sname));
                out->Data[1]->Slice<int32_t>()[i] = ckey[(size_t)(pos + i)] % 97;
                out->Data[2]->Slice<int32_t>()[i] = ckey[(size_t)(pos + i)];
            }
            out->SetCard(card);
            pos += card;
            return true;
        });
        gpuOrderExecutor ord(ctx, {{0, true}, {1, false}, {2, true}}, &src);
        print(3, run(&ord));
    } else if (q == "q9") {
        // cases/tpch/query/q9.sql through the operator interface: LIKE filter on part, the join chain
        // lineitem |x| part |x| partsupp (composite key) |x| supplier |x| orders |x| nation, a Project with
        // extract(year ...) and the profit expression, the aggregate, ORDER BY nation, o_year DESC
        int64_t no = tpchgen_orders_count(num, den), npart = tpchgen_part_count(num, den), ns = tpchgen_supplier_count(num, den);
        int64_t nl = tpchgen_lineitem_count(num, den, 0, no);
        std::vector<int64_t> lokey((size_t)nl), lext((size_t)nl), ldisc((size_t)nl), okey((size_t)no), pscost((size_t)npart * 4);
        std::vector<int32_t> lpart((size_t)nl), lsupp((size_t)nl), lqty((size_t)nl), odate((size_t)no), pkey((size_t)npart),
            pspart((size_t)npart * 4), pssupp((size_t)npart * 4), skey((size_t)ns), snat((size_t)ns), nkey(25);
        std::vector<uint8_t> pcolors((size_t)npart * 5), ncode(25);
        tpchgen_lineitem_cols lc{}; lc.l_orderkey = lokey.data(); lc.l_partkey = lpart.data(); lc.l_suppkey = lsupp.data();
        lc.l_quantity = lqty.data(); lc.l_extendedprice = lext.data(); lc.l_discount = ldisc.data();
        tpchgen_lineitem(num, den, 0, no, &lc);
        tpchgen_orders_cols oc{}; oc.o_orderkey = okey.data(); oc.o_orderdate = odate.data();
        tpchgen_orders(num, den, 0, no, &oc);
        tpchgen_part_cols pc{}; pc.p_partkey = pkey.data(); pc.p_name_colors = pcolors.data();
        tpchgen_part(num, den, 0, npart, &pc);
        tpchgen_partsupp_cols psc{}; psc.ps_partkey = pspart.data(); psc.ps_suppkey = pssupp.data(); psc.ps_supplycost = pscost.data();
        tpchgen_partsupp(num, den, 0, npart, &psc);
        tpchgen_supplier_cols sc{}; sc.s_suppkey = skey.data(); sc.s_nationkey = snat.data();
        tpchgen_supplier(num, den, 0, ns, &sc);
        for (int i = 0; i < 25; i++) { nkey[(size_t)i] = i; ncode[(size_t)i] = (uint8_t)i; }
        auto pname = [&pcolors](int64_t r) {
            std::string s;
            for (int k = 0; k < 5; k++) { if (k) s += ' '; s += TPCHGEN_COLORS[pcolors[(size_t)r * 5 + (size_t)k]]; }
            return s;
        };
        LType dec = DecimalType(15, 2);
        auto lsrc = table_source({{BigintType(), lokey.data()}, {IntegerType(), lpart.data()}, {IntegerType(), lsupp.data()},
                                  {IntegerType(), lqty.data()}, {dec, lext.data()}, {dec, ldisc.data()}}, nl);
        auto psrc = table_source({{IntegerType(), pkey.data()}, {VarcharType(), nullptr, nullptr, pname}}, npart);
        auto pssrc = table_source({{IntegerType(), pspart.data()}, {IntegerType(), pssupp.data()}, {dec, pscost.data()}}, npart * 4);
        auto ssrc = table_source({{IntegerType(), skey.data()}, {IntegerType(), snat.data()}}, ns);
        auto osrc = table_source({{BigintType(), okey.data()}, {DateType(), odate.data()}}, no);
        auto nsrc = table_source({{IntegerType(), nkey.data()}, {VarcharType(), ncode.data(), TPCHGEN_NATION_NAMES}}, 25);
        Literal pat; pat.kind = Literal::Str; pat.s = "%pink%";
        gpuFilterExecutor pf(ctx, {{1, PH_LIKE, pat}}, psrc.get());
        // lineitem: 0 l_orderkey 1 l_partkey 2 l_suppkey 3 l_quantity 4 l_extendedprice 5 l_discount
        gpuJoinExecutor j1(ctx, lsrc.get(), &pf, {1}, {0}, {});                 // p_partkey = l_partkey
        gpuJoinExecutor j2(ctx, &j1, pssrc.get(), {1, 2}, {0, 1}, {2});         // + 6 ps_supplycost
        gpuJoinExecutor j3(ctx, &j2, ssrc.get(), {2}, {0}, {1});                // + 7 s_nationkey
        gpuJoinExecutor j4(ctx, &j3, osrc.get(), {0}, {0}, {1});                // + 8 o_orderdate
        gpuJoinExecutor j5(ctx, &j4, nsrc.get(), {7}, {0}, {1});                // + 9 n_name
        // nation, o_year, amount = l_extendedprice * (1 - l_discount) - ps_supplycost * l_quantity
        gpuProjectExecutor proj(ctx, {ProjExpr::Col(9), ProjExpr::Year(8),
                                      ProjExpr::Dec({X_COL(4), X_CONST(1, 0), X_COL(5), X_OP(PH_X_SUB), X_OP(PH_X_MUL),
                                                     X_COL(6), X_COL(3), X_OP(PH_X_MUL), X_OP(PH_X_SUB)})}, &j5);
        gpuAggExecutor agg(ctx, {0, 1}, {{PH_A_SUM, {X_COL(2)}}}, &proj);
        std::vector<OperatorExec *> ops = {&pf, &j1, &j2, &j3, &j4, &j5, &proj, &agg};
        for (OperatorExec *e : ops) { std::string er = e->Init(); if (!er.empty()) die("init: " + er); }
        gpuOrderExecutor ord(ctx, {{0, false}, {1, true}}, &agg);               // ORDER BY nation, o_year DESC
        print(3, run(&ord));
        for (OperatorExec *e : ops) e->Close();
    } else if (q == "having") {
        // SELECT l_suppkey, sum(l_extendedprice) * 2, count(*) FROM lineitem GROUP BY l_suppkey
        // HAVING sum(l_extendedprice) > 20000000.00 AND count(*) > 600  — DECIMAL '>' and HUGEINT '>' are
        // the comparisons HAVING can use (function_operator_boolean.go:431-442); the output list is
        // evaluated over the surviving group rows
        int64_t no = tpchgen_orders_count(num, den);
        int64_t nl = tpchgen_lineitem_count(num, den, 0, no);
        std::vector<int32_t> lsupp((size_t)nl);
        std::vector<int64_t> lext((size_t)nl);
        tpchgen_lineitem_cols lc{}; lc.l_suppkey = lsupp.data(); lc.l_extendedprice = lext.data();
        tpchgen_lineitem(num, den, 0, no, &lc);
        auto src = table_source({{IntegerType(), lsupp.data()}, {DecimalType(15, 2), lext.data()}}, nl);
        gpuAggExecutor agg(ctx, {0}, {{PH_A_SUM, {X_COL(1)}}, {PH_A_COUNT_STAR, {}}}, src.get());
        Literal lim; lim.kind = Literal::Dec; lim.i = 2000000000; lim.scale = 2;
        Literal cnt; cnt.kind = Literal::Int; cnt.i = 600;
        agg.SetHaving({{1, PH_GT, lim}, {2, PH_GT, cnt}});
        agg.SetOutputs({ProjExpr::Col(0), ProjExpr::Dec({X_COL(1), X_CONST(2, 0), X_OP(PH_X_MUL)}), ProjExpr::Col(2)});
        print(3, run(&agg));
    } else if (q == "cross") {
        // (first 5000 customers) x (3 constant rows): CrossProduct emits, per left chunk, one chunk per right row
        int64_t nc = std::min<int64_t>(tpchgen_customer_count(num, den), 5000);
        std::vector<int32_t> ckey((size_t)nc), rv = {7, 8, 9};
        std::vector<uint8_t> cseg((size_t)nc), rcode = {2, 0, 4};
        tpchgen_customer_cols cc{}; cc.c_custkey = ckey.data(); cc.c_mktsegment = cseg.data();
        tpchgen_customer(num, den, 0, nc, &cc);
        auto left = table_source({{IntegerType(), ckey.data()}, {VarcharType(), cseg.data(), TPCHGEN_MKTSEGMENT_DICT}}, nc);
        auto right = table_source({{IntegerType(), rv.data()}, {VarcharType(), rcode.data(), TPCHGEN_MKTSEGMENT_DICT}}, 3);
        crossProductExecutor cross(left.get(), right.get());
        print(4, run(&cross));
    } else if (q == "substr") {
        if (argc < 6) die("usage: host_tester substr <sf_num> <sf_den> <offset> <length>");
        int64_t nc = tpchgen_customer_count(num, den);
        std::vector<int32_t> ckey((size_t)nc);
        std::vector<uint8_t> cseg((size_t)nc);
        tpchgen_customer_cols cc{}; cc.c_custkey = ckey.data(); cc.c_mktsegment = cseg.data();
        tpchgen_customer(num, den, 0, nc, &cc);
        auto src = table_source({{IntegerType(), ckey.data()}, {VarcharType(), cseg.data(), TPCHGEN_MKTSEGMENT_DICT}}, nc);
        gpuProjectExecutor proj(ctx, {ProjExpr::Col(0), ProjExpr::Substr(1, atoll(argv[4]), atoll(argv[5]))}, src.get());
        print(2, run(&proj));
    } else die("unknown query " + q);
    ph_ctx_destroy(ctx);
    return 0;
}

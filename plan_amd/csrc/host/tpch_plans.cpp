// See tpch_plans.h.
#include "tpch_plans.h"

#include <chrono>
#include <cstring>

#include "tpchgen.h"

namespace plan {

static std::string herr(const char *what) { return std::string(what) + ": " + ph_last_error(); }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace {

struct HostCol {
    LType type;
    int32_t ph = 0, scale = 0;
    const void *data = nullptr;
    std::vector<std::string> dict;           // PH_CODE8
    const void *aux = nullptr;               // PH_STR bytes
    int64_t aux_bytes = 0;
};

std::vector<std::string> dictOf(const char *const *d, int n) { return std::vector<std::string>(d, d + n); }

std::string loadTable(ph_ctx *ctx, const std::vector<HostCol> &cols, int64_t n, const std::vector<int> &primaryKey, ResidentTable *out, int64_t *bytes) {
    std::vector<ph_col> hc(cols.size());
    std::vector<std::string> blobs(cols.size());
    for (size_t c = 0; c < cols.size(); c++) {
        hc[c] = ph_col{};
        hc[c].type = cols[c].ph; hc[c].scale = cols[c].scale; hc[c].data = cols[c].data;
        if (cols[c].ph == PH_CODE8) {
            for (auto &s : cols[c].dict) { blobs[c] += s; blobs[c].push_back('\0'); }
            hc[c].aux = blobs[c].data(); hc[c].aux_bytes = (int64_t)blobs[c].size();
        } else if (cols[c].ph == PH_STR) { hc[c].aux = cols[c].aux; hc[c].aux_bytes = cols[c].aux_bytes; *bytes += cols[c].aux_bytes; }
        *bytes += n * (cols[c].ph == PH_CODE8 ? 1 : (cols[c].ph == PH_I64 || cols[c].ph == PH_DEC64) ? 8 : 4);
        out->cols.push_back(ResidentColumn{cols[c].type, cols[c].dict});
    }
    ph_table *t = nullptr;
    if (ph_table_create(ctx, (int32_t)hc.size(), hc.data(), n, &t) != PH_OK) return herr("ph_table_create");
    out->table = t;
    if (!primaryKey.empty()) {
        std::vector<int32_t> pk(primaryKey.begin(), primaryKey.end());
        if (ph_table_declare_unique(t, (int32_t)pk.size(), pk.data()) != PH_OK) return herr("ph_table_declare_unique");
    }
    return "";
}

// the generator's fixed-stride characters + lengths as offsets + bytes (a PH_STR column)
void packStrings(const std::vector<char> &fixed, int stride, const std::vector<uint8_t> &len, std::vector<int32_t> *off, std::string *bytes) {
    const size_t n = len.size();
    off->assign(n + 1, 0);
    size_t total = 0;
    for (size_t r = 0; r < n; r++) total += len[r];
    bytes->clear();
    bytes->reserve(total);
    for (size_t r = 0; r < n; r++) {
        bytes->append(fixed.data() + r * (size_t)stride, len[r]);
        (*off)[r + 1] = (int32_t)bytes->size();
    }
}
HostCol STR(const std::vector<int32_t> &off, const std::string &bytes) { return HostCol{VarcharType(), PH_STR, 0, off.data(), {}, bytes.data(), (int64_t)bytes.size()}; }

HostCol I32(const void *d) { return HostCol{IntegerType(), PH_I32, 0, d, {}, nullptr, 0}; }
HostCol I64(const void *d) { return HostCol{BigintType(), PH_I64, 0, d, {}, nullptr, 0}; }
HostCol DEC(const void *d) { return HostCol{DecimalType(15, 2), PH_DEC64, 2, d, {}, nullptr, 0}; }
HostCol DATE(const void *d) { return HostCol{DateType(), PH_DATE, 0, d, {}, nullptr, 0}; }
HostCol CODE(const void *d, std::vector<std::string> dict) { return HostCol{VarcharType(), PH_CODE8, 0, d, std::move(dict), nullptr, 0}; }

}  // namespace

TpchDatabase::~TpchDatabase() {
    for (ResidentTable *t : {&lineitem, &orders, &customer, &part, &partsupp, &supplier, &nation, &region})
        if (t->table) ph_table_free(const_cast<ph_table *>(t->table));
}

std::string TpchDatabase::Load(ph_ctx *c, int64_t sf_num, int64_t sf_den, int rank, int nranks) {
    ctx = c; num = sf_num; den = sf_den;
    if (nranks < 1 || rank < 0 || rank >= nranks) return "TpchDatabase::Load: bad rank";
    // this rank's row ranges [first, first + n) of every table (the generator starts at any row)
    auto cut = [&](int64_t n, int64_t *first) { *first = n * rank / nranks; return n * (rank + 1) / nranks - *first; };
    int64_t o0 = 0, c0 = 0, p0 = 0, s0 = 0;
    const int64_t no = cut(tpchgen_orders_count(num, den), &o0), nc = cut(tpchgen_customer_count(num, den), &c0), np = cut(tpchgen_part_count(num, den), &p0),
                  ns = cut(tpchgen_supplier_count(num, den), &s0), nl = tpchgen_lineitem_count(num, den, o0, no);
    std::string e;
    double t0 = now_s();
    {   // lineitem
        std::vector<int64_t> okey((size_t)nl), ext((size_t)nl), disc((size_t)nl), tax((size_t)nl);
        std::vector<int32_t> pk((size_t)nl), sk((size_t)nl), qty((size_t)nl), ship((size_t)nl), commit((size_t)nl), receipt((size_t)nl), lno((size_t)nl);
        std::vector<uint8_t> rf((size_t)nl), ls((size_t)nl), mode((size_t)nl), instr((size_t)nl);
        tpchgen_lineitem_cols lc{};
        lc.l_orderkey = okey.data(); lc.l_partkey = pk.data(); lc.l_suppkey = sk.data(); lc.l_quantity = qty.data(); lc.l_extendedprice = ext.data();
        lc.l_discount = disc.data(); lc.l_tax = tax.data(); lc.l_returnflag = rf.data(); lc.l_linestatus = ls.data(); lc.l_shipdate = ship.data();
        lc.l_commitdate = commit.data(); lc.l_receiptdate = receipt.data(); lc.l_shipmode = mode.data(); lc.l_shipinstruct = instr.data();
        lc.l_linenumber = lno.data();
        tpchgen_lineitem(num, den, o0, no, &lc);
        generate_s += now_s() - t0; t0 = now_s();
        e = loadTable(ctx, {I64(okey.data()), I32(pk.data()), I32(sk.data()), I32(qty.data()), DEC(ext.data()), DEC(disc.data()), DEC(tax.data()),
                            CODE(rf.data(), dictOf(TPCHGEN_RETURNFLAG_DICT, 3)), CODE(ls.data(), dictOf(TPCHGEN_LINESTATUS_DICT, 2)), DATE(ship.data()),
                            DATE(commit.data()), DATE(receipt.data()), CODE(mode.data(), dictOf(TPCHGEN_SHIPMODE_DICT, 7)),
                            CODE(instr.data(), dictOf(TPCHGEN_SHIPINSTRUCT_DICT, 4)), I32(lno.data())}, nl, {}, &lineitem, &loaded_bytes);
        if (!e.empty()) return e;
        load_s += now_s() - t0; t0 = now_s();
    }
    {   // orders
        std::vector<int64_t> okey((size_t)no), total((size_t)no);
        std::vector<int32_t> cust((size_t)no), date((size_t)no), sprio((size_t)no);
        std::vector<uint8_t> oprio((size_t)no), ostat((size_t)no), clen((size_t)no);
        std::vector<char> cmnt((size_t)no * TPCHGEN_O_COMMENT_STRIDE);
        tpchgen_orders_cols oc{};
        oc.o_orderkey = okey.data(); oc.o_custkey = cust.data(); oc.o_orderdate = date.data(); oc.o_shippriority = sprio.data(); oc.o_orderpriority = oprio.data(); oc.o_totalprice = total.data();
        oc.o_orderstatus = ostat.data(); oc.o_comment = cmnt.data(); oc.o_comment_len = clen.data();
        tpchgen_orders(num, den, o0, no, &oc);
        std::vector<int32_t> coff;
        std::string cbytes;
        packStrings(cmnt, TPCHGEN_O_COMMENT_STRIDE, clen, &coff, &cbytes);   // o_comment: 19..78 characters of the generator's text pool (Q13's NOT LIKE)
        std::vector<char>().swap(cmnt);
        for (auto &b : ostat) b = b == 'F' ? 0 : b == 'O' ? 1 : 2;   // the generator writes the raw byte: codes into {"F", "O", "P"}
        generate_s += now_s() - t0; t0 = now_s();
        e = loadTable(ctx, {I64(okey.data()), I32(cust.data()), DATE(date.data()), I32(sprio.data()), CODE(oprio.data(), dictOf(TPCHGEN_ORDERPRIORITY_DICT, 5)),
                            DEC(total.data()), CODE(ostat.data(), {"F", "O", "P"}), STR(coff, cbytes)}, no, {O_ORDERKEY}, &orders, &loaded_bytes);
        if (!e.empty()) return e;
        load_s += now_s() - t0; t0 = now_s();
    }
    {   // customer
        std::vector<int32_t> key((size_t)nc), nat((size_t)nc);
        std::vector<uint8_t> seg((size_t)nc);
        std::vector<char> phone((size_t)nc * TPCHGEN_S_PHONE_LEN);
        std::vector<int64_t> bal((size_t)nc);
        std::vector<int32_t> poff((size_t)nc + 1);
        std::vector<char> addr((size_t)nc * TPCHGEN_S_ADDRESS_STRIDE), cmnt((size_t)nc * TPCHGEN_C_COMMENT_STRIDE);
        std::vector<uint8_t> alen((size_t)nc), clen((size_t)nc);
        tpchgen_customer_cols cc{};
        cc.c_custkey = key.data(); cc.c_nationkey = nat.data(); cc.c_mktsegment = seg.data(); cc.c_phone = phone.data(); cc.c_acctbal = bal.data();
        cc.c_address = addr.data(); cc.c_address_len = alen.data(); cc.c_comment = cmnt.data(); cc.c_comment_len = clen.data();
        tpchgen_customer(num, den, c0, nc, &cc);
        std::vector<int32_t> aoff, coff;
        std::string abytes, cbytes;
        packStrings(addr, TPCHGEN_S_ADDRESS_STRIDE, alen, &aoff, &abytes);     // c_address / c_comment: Q10's select list
        packStrings(cmnt, TPCHGEN_C_COMMENT_STRIDE, clen, &coff, &cbytes);
        for (int64_t r = 0; r <= nc; r++) poff[(size_t)r] = (int32_t)(r * TPCHGEN_S_PHONE_LEN);
        HostCol phoneCol{VarcharType(), PH_STR, 0, poff.data(), {}, phone.data(), (int64_t)phone.size()};
        // c_name = 'Customer#' + the key as nine digits (TPC-H 4.2.3): 1.5 M distinct strings at SF10, so offsets + bytes, no dictionary
        std::vector<int32_t> off((size_t)nc + 1);
        std::string bytes((size_t)nc * 18, '0');
        for (int64_t r = 0; r < nc; r++) {
            off[(size_t)r] = (int32_t)(r * 18);
            char *b = &bytes[(size_t)r * 18];
            memcpy(b, "Customer#", 9);
            int32_t k = key[(size_t)r];
            for (int d = 17; d >= 9; d--) { b[d] = (char)('0' + k % 10); k /= 10; }
        }
        off[(size_t)nc] = (int32_t)(nc * 18);
        HostCol name{VarcharType(), PH_STR, 0, off.data(), {}, bytes.data(), (int64_t)bytes.size()};
        e = loadTable(ctx, {I32(key.data()), I32(nat.data()), CODE(seg.data(), dictOf(TPCHGEN_MKTSEGMENT_DICT, 5)), name, phoneCol, DEC(bal.data()), STR(aoff, abytes),
                            STR(coff, cbytes)}, nc, {C_CUSTKEY}, &customer, &loaded_bytes);
        if (!e.empty()) return e;
    }
    {   // part: p_name as offsets + bytes (LIKE operand), the other VARCHAR columns as dictionary codes
        std::vector<int32_t> key((size_t)np), size((size_t)np), off((size_t)np + 1);
        std::vector<uint8_t> colors((size_t)np * 5), brand((size_t)np), type((size_t)np), cntr((size_t)np), mfgr((size_t)np);
        tpchgen_part_cols pc{};
        pc.p_partkey = key.data(); pc.p_name_colors = colors.data(); pc.p_brand = brand.data(); pc.p_type = type.data(); pc.p_size = size.data(); pc.p_container = cntr.data();
        pc.p_mfgr = mfgr.data();
        tpchgen_part(num, den, p0, np, &pc);
        std::string bytes;
        for (int64_t r = 0; r < np; r++) {
            off[(size_t)r] = (int32_t)bytes.size();
            for (int k = 0; k < 5; k++) { if (k) bytes.push_back(' '); bytes += TPCHGEN_COLORS[colors[(size_t)r * 5 + (size_t)k]]; }
        }
        off[(size_t)np] = (int32_t)bytes.size();
        HostCol name{VarcharType(), PH_STR, 0, off.data(), {}, bytes.data(), (int64_t)bytes.size()};
        e = loadTable(ctx, {I32(key.data()), name, CODE(brand.data(), dictOf(tpchgen_part_brand_dict(), 25)), CODE(type.data(), dictOf(tpchgen_part_type_dict(), 150)),
                            I32(size.data()), CODE(cntr.data(), dictOf(tpchgen_part_container_dict(), 40)),
                            CODE(mfgr.data(), {"Manufacturer#1", "Manufacturer#2", "Manufacturer#3", "Manufacturer#4", "Manufacturer#5"})}, np, {P_PARTKEY}, &part, &loaded_bytes);
        if (!e.empty()) return e;
    }
    {   // partsupp
        std::vector<int32_t> pk((size_t)np * 4), sk((size_t)np * 4), qty((size_t)np * 4);
        std::vector<int64_t> cost((size_t)np * 4);
        tpchgen_partsupp_cols pc{};
        pc.ps_partkey = pk.data(); pc.ps_suppkey = sk.data(); pc.ps_supplycost = cost.data(); pc.ps_availqty = qty.data();
        tpchgen_partsupp(num, den, p0, np, &pc);
        e = loadTable(ctx, {I32(pk.data()), I32(sk.data()), DEC(cost.data()), I32(qty.data())}, np * 4, {PS_PARTKEY, PS_SUPPKEY}, &partsupp, &loaded_bytes);
        if (!e.empty()) return e;
    }
    {   // supplier
        std::vector<int32_t> key((size_t)ns), nat((size_t)ns);
        std::vector<char> addr((size_t)ns * TPCHGEN_S_ADDRESS_STRIDE), phone((size_t)ns * TPCHGEN_S_PHONE_LEN);
        std::vector<uint8_t> alen((size_t)ns), clen((size_t)ns);
        std::vector<char> cmnt((size_t)ns * TPCHGEN_S_COMMENT_STRIDE);
        std::vector<int64_t> bal((size_t)ns);
        tpchgen_supplier_cols sc{};
        sc.s_suppkey = key.data(); sc.s_nationkey = nat.data(); sc.s_address = addr.data(); sc.s_address_len = alen.data(); sc.s_phone = phone.data();
        sc.s_acctbal = bal.data(); sc.s_comment = cmnt.data(); sc.s_comment_len = clen.data();
        tpchgen_supplier(num, den, s0, ns, &sc);
        std::vector<int32_t> coff;
        std::string cbytes;
        packStrings(cmnt, TPCHGEN_S_COMMENT_STRIDE, clen, &coff, &cbytes);   // s_comment with the "Customer ... Complaints" injection (Q16, Q2)
        std::vector<int32_t> aoff((size_t)ns + 1, 0), poff((size_t)ns + 1);
        std::string abytes;
        for (int64_t r = 0; r < ns; r++) {
            abytes.append(addr.data() + r * TPCHGEN_S_ADDRESS_STRIDE, alen[(size_t)r]);
            aoff[(size_t)r + 1] = (int32_t)abytes.size();
            poff[(size_t)r] = (int32_t)(r * TPCHGEN_S_PHONE_LEN);
        }
        poff[(size_t)ns] = (int32_t)(ns * TPCHGEN_S_PHONE_LEN);
        HostCol saddr{VarcharType(), PH_STR, 0, aoff.data(), {}, abytes.data(), (int64_t)abytes.size()};
        HostCol sphone{VarcharType(), PH_STR, 0, poff.data(), {}, phone.data(), (int64_t)phone.size()};
        std::vector<int32_t> noff((size_t)ns + 1);
        std::string nbytes((size_t)ns * 18, '0');
        for (int64_t r = 0; r < ns; r++) {   // s_name = 'Supplier#' + the key as nine digits (TPC-H 4.2.3)
            noff[(size_t)r] = (int32_t)(r * 18);
            char *b = &nbytes[(size_t)r * 18];
            memcpy(b, "Supplier#", 9);
            int32_t k = key[(size_t)r];
            for (int d = 17; d >= 9; d--) { b[d] = (char)('0' + k % 10); k /= 10; }
        }
        noff[(size_t)ns] = (int32_t)(ns * 18);
        HostCol sname{VarcharType(), PH_STR, 0, noff.data(), {}, nbytes.data(), (int64_t)nbytes.size()};
        e = loadTable(ctx, {I32(key.data()), I32(nat.data()), sname, saddr, sphone, DEC(bal.data()), STR(coff, cbytes)}, ns, {S_SUPPKEY}, &supplier, &loaded_bytes);
        if (!e.empty()) return e;
    }
    {   // nation, region: the specification's fixed tables
        std::vector<int32_t> nk(25), nr(25), rk(5);
        std::vector<uint8_t> nn(25), rn(5);
        for (int i = 0; i < 25; i++) { nk[(size_t)i] = i; nn[(size_t)i] = (uint8_t)i; nr[(size_t)i] = TPCHGEN_NATION_REGION[i]; }
        for (int i = 0; i < 5; i++) { rk[(size_t)i] = i; rn[(size_t)i] = (uint8_t)i; }
        e = loadTable(ctx, {I32(nk.data()), CODE(nn.data(), dictOf(TPCHGEN_NATION_NAMES, 25)), I32(nr.data())}, 25, {N_NATIONKEY}, &nation, &loaded_bytes);
        if (!e.empty()) return e;
        e = loadTable(ctx, {I32(rk.data()), CODE(rn.data(), dictOf(TPCHGEN_REGION_NAMES, 5))}, 5, {R_REGIONKEY}, &region, &loaded_bytes);
        if (!e.empty()) return e;
        if (nranks > 1) {   // the specification's fixed tables are whole on every rank
            ph_table_set_replicated(const_cast<ph_table *>(nation.table), 1);
            ph_table_set_replicated(const_cast<ph_table *>(region.table), 1);
        }
    }
    load_s += now_s() - t0;
    return "";
}

// ---- literals and expression shorthands
static Literal LDate(int y, int m, int d) { Literal k; k.kind = Literal::DateDays; k.i = tpchgen_days_from_civil(y, m, d); return k; }
static Literal LDays(int32_t days) { Literal k; k.kind = Literal::DateDays; k.i = days; return k; }
static Literal LStr(const char *s) { Literal k; k.kind = Literal::Str; k.s = s; return k; }
static Literal LInt(int64_t v) { Literal k; k.kind = Literal::Int; k.i = v; return k; }
static Literal LFloat(float f) { Literal k; k.kind = Literal::Float; k.f = (double)f; return k; }
static Literal LDec(int64_t unscaled, int scale) { Literal k; k.kind = Literal::Dec; k.i = unscaled; k.scale = scale; return k; }
static ph_rpn XC(int c) { return ph_rpn{PH_X_COL, c, 0, 0}; }
static ph_rpn XK(int64_t v, int s = 0) { return ph_rpn{PH_X_CONST, -1, v, s}; }
static ph_rpn XO(int op) { return ph_rpn{op, -1, 0, 0}; }
// e * (1 - d)
static std::vector<ph_rpn> DiscPrice(int e, int d) { return {XC(e), XK(1), XC(d), XO(PH_X_SUB), XO(PH_X_MUL)}; }

std::string BuildTpchQuery(const TpchDatabase &db, int id, TpchQuery *q) {
    *q = TpchQuery{};
    q->id = id;
    ResidentPlan &p = q->plan;
    switch (id) {
    case 1: {   // Order <- Agg <- Scan(lineitem, l_shipdate <= date '1998-12-01' - interval '112 day')
        int s = p.Scan(&db.lineitem, {L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_RETURNFLAG, L_LINESTATUS},
                       {{L_SHIPDATE, PH_LE, LDays(tpchgen_days_from_civil(1998, 12, 1) - 112)}});
        std::vector<ph_rpn> dp = DiscPrice(1, 2), ch = dp;
        ch.push_back(XK(1)); ch.push_back(XC(3)); ch.push_back(XO(PH_X_ADD)); ch.push_back(XO(PH_X_MUL));
        p.Agg(s, {ProjExpr::Col(4), ProjExpr::Col(5)},
              {{PH_A_SUM, {XC(0)}}, {PH_A_SUM, {XC(1)}}, {PH_A_SUM, dp}, {PH_A_SUM, ch}, {PH_A_AVG, {XC(0)}}, {PH_A_AVG, {XC(1)}}, {PH_A_AVG, {XC(2)}},
               {PH_A_COUNT_STAR, {}}});
        q->order = {{0, false}, {1, false}};
        q->ncols = 10;
        break;
    }
    case 6: {   // Agg <- Scan(lineitem, shipdate range, discount between 0.03 -/+ 0.01 (float32 literals), quantity < 24)
        int s = p.Scan(&db.lineitem, {L_EXTENDEDPRICE, L_DISCOUNT},
                       {{L_SHIPDATE, PH_GE, LDate(1994, 1, 1)}, {L_SHIPDATE, PH_LT, LDate(1995, 1, 1)}, {L_DISCOUNT, PH_GE, LFloat(0.03f - 0.01f)},
                        {L_DISCOUNT, PH_LE, LFloat(0.03f + 0.01f)}, {L_QUANTITY, PH_LT, LInt(24)}});
        p.Agg(s, {}, {{PH_A_SUM, {XC(0), XC(1), XO(PH_X_MUL)}}});
        q->ncols = 1;
        break;
    }
    case 3: {
        // Limit <- Order <- Agg(l_orderkey, o_orderdate, o_shippriority; sum(l_extendedprice * (1 - l_discount)))
        //   <- Join(l_orderkey = o_orderkey) probe Scan(lineitem, l_shipdate > d)
        //        build <- Join(o_custkey = c_custkey) probe Scan(orders, o_orderdate < d), build Scan(customer, c_mktsegment = 'HOUSEHOLD')
        Literal d = LDate(1995, 3, 29);
        int cust = p.Scan(&db.customer, {C_CUSTKEY}, {{C_MKTSEGMENT, PH_EQ, LStr("HOUSEHOLD")}});
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_SHIPPRIORITY}, {{O_ORDERDATE, PH_LT, d}});
        int j1 = p.Join(ord, cust, {1}, {0}, {0, 2, 3});
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_EXTENDEDPRICE, L_DISCOUNT}, {{L_SHIPDATE, PH_GT, d}});
        int j2 = p.Join(line, j1, {0}, {0}, {0, 1, 2, 4, 5});
        p.Agg(j2, {ProjExpr::Col(0), ProjExpr::Col(3), ProjExpr::Col(4)}, {{PH_A_SUM, DiscPrice(1, 2)}});
        // select list: l_orderkey, revenue, o_orderdate, o_shippriority; ORDER BY revenue DESC, o_orderdate LIMIT 10
        q->outputs = {ProjExpr::Col(0), ProjExpr::Col(3), ProjExpr::Col(1), ProjExpr::Col(2)};
        q->order = {{1, true}, {2, false}};
        q->limit = 10;
        q->topkAgg = 0; q->topkDesc = true;
        q->ncols = 4;
        break;
    }
    case 9: {
        // Order <- Agg(nation, o_year; sum(amount)) <- Project(n_name, extract(year from o_orderdate),
        //   l_extendedprice * (1 - l_discount) - ps_supplycost * l_quantity)
        //   <- Join(s_nationkey = n_nationkey) <- Join(l_orderkey = o_orderkey) <- Join(l_suppkey = s_suppkey)
        //   <- Join((l_partkey, l_suppkey) = (ps_partkey, ps_suppkey)) <- Join(l_partkey = p_partkey) <- Scan(lineitem); part filtered by LIKE
        int part = p.Scan(&db.part, {P_PARTKEY}, {{P_NAME, PH_LIKE, LStr("%pink%")}});
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_PARTKEY, L_SUPPKEY, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT});
        int j1 = p.Join(line, part, {1}, {0}, {0, 1, 2, 3, 4, 5});
        int ps = p.Scan(&db.partsupp, {PS_PARTKEY, PS_SUPPKEY, PS_SUPPLYCOST});
        int j2 = p.Join(j1, ps, {1, 2}, {0, 1}, {0, 2, 3, 4, 5, 8});     // l_orderkey, l_suppkey, qty, ext, disc, ps_supplycost
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY});
        int j3 = p.Join(j2, supp, {1}, {0}, {0, 2, 3, 4, 5, 7});           // l_orderkey, qty, ext, disc, cost, s_nationkey
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_ORDERDATE});
        int j4 = p.Join(j3, ord, {0}, {0}, {1, 2, 3, 4, 5, 7});            // qty, ext, disc, cost, s_nationkey, o_orderdate
        int nat = p.Scan(&db.nation, {N_NATIONKEY, N_NAME});
        int j5 = p.Join(j4, nat, {4}, {0}, {0, 1, 2, 3, 5, 7});            // qty, ext, disc, cost, o_orderdate, n_name
        std::vector<ph_rpn> amount = DiscPrice(1, 2);
        amount.push_back(XC(3)); amount.push_back(XC(0)); amount.push_back(XO(PH_X_MUL)); amount.push_back(XO(PH_X_SUB));
        int proj = p.Project(j5, {ProjExpr::Col(5), ProjExpr::Year(4), ProjExpr::Dec(amount)});
        p.Agg(proj, {ProjExpr::Col(0), ProjExpr::Col(1)}, {{PH_A_SUM, {XC(2)}}});
        q->order = {{0, false}, {1, true}};
        q->ncols = 3;
        break;
    }
    case 4: {
        // Order <- Agg(o_orderpriority; count(*)) <- SemiJoin(o_orderkey = l_orderkey) probe Scan(orders, date range)
        //   build Scan(lineitem, l_commitdate < l_receiptdate): the decorrelated EXISTS
        int line = p.Scan(&db.lineitem, {L_ORDERKEY}, {}, BoolExpr::CC(L_COMMITDATE, PH_LT, L_RECEIPTDATE));
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_ORDERPRIORITY}, {{O_ORDERDATE, PH_GE, LDate(1997, 7, 1)}, {O_ORDERDATE, PH_LT, LDate(1997, 10, 1)}});
        int j = p.Join(ord, line, {0}, {0}, {1}, JoinSemi);
        p.Agg(j, {ProjExpr::Col(0)}, {{PH_A_COUNT_STAR, {}}});
        q->order = {{0, false}};
        q->ncols = 2;
        break;
    }
    case 5: {
        // Order <- Agg(n_name; sum(e * (1 - d))) over the six-table chain; the last join carries the two-column condition
        // (l_suppkey, c_nationkey) = (s_suppkey, s_nationkey)
        int reg = p.Scan(&db.region, {R_REGIONKEY}, {{R_NAME, PH_EQ, LStr("AMERICA")}});
        int nat = p.Scan(&db.nation, {N_NATIONKEY, N_NAME, N_REGIONKEY});
        int jn = p.Join(nat, reg, {2}, {0}, {0, 1});                       // n_nationkey, n_name
        int cust = p.Scan(&db.customer, {C_CUSTKEY, C_NATIONKEY});
        int jc = p.Join(cust, jn, {1}, {0}, {0, 1, 3});                    // c_custkey, c_nationkey, n_name
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_CUSTKEY}, {{O_ORDERDATE, PH_GE, LDate(1994, 1, 1)}, {O_ORDERDATE, PH_LT, LDate(1995, 1, 1)}});
        int jo = p.Join(ord, jc, {1}, {0}, {0, 3, 4});                     // o_orderkey, c_nationkey, n_name
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_SUPPKEY, L_EXTENDEDPRICE, L_DISCOUNT});
        int jl = p.Join(line, jo, {0}, {0}, {1, 2, 3, 5, 6});              // l_suppkey, ext, disc, c_nationkey, n_name
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY});
        int js = p.Join(jl, supp, {0, 3}, {0, 1}, {1, 2, 4});              // ext, disc, n_name
        p.Agg(js, {ProjExpr::Col(2)}, {{PH_A_SUM, DiscPrice(0, 1)}});
        q->order = {{1, true}};                                            // ORDER BY revenue DESC
        q->ncols = 2;
        break;
    }
    case 7: {
        // Order <- Agg(n1.n_name, n2.n_name, year(l_shipdate); sum(volume)) <- Project <- Filter((n1 = A and n2 = B) or (n1 = B and n2 = A))
        //   <- Join(o_custkey = c_custkey) <- Join(l_orderkey = o_orderkey) <- Join(l_suppkey = s_suppkey) probe Scan(lineitem, l_shipdate between);
        // the supplier and the customer side each arrive joined with their nation scan, which carries n_name IN (A, B) — implied by the
        // pair condition (the planner's derivation)
        const char *A = "FRANCE", *B = "ARGENTINA";
        int n1 = p.Scan(&db.nation, {N_NATIONKEY, N_NAME}, {}, BoolExpr::In(N_NAME, {LStr(A), LStr(B)}));
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY});
        int js = p.Join(supp, n1, {1}, {0}, {0, 3});                       // s_suppkey, n1.n_name
        int n2 = p.Scan(&db.nation, {N_NATIONKEY, N_NAME}, {}, BoolExpr::In(N_NAME, {LStr(A), LStr(B)}));
        int cust = p.Scan(&db.customer, {C_CUSTKEY, C_NATIONKEY});
        int jc = p.Join(cust, n2, {1}, {0}, {0, 3});                       // c_custkey, n2.n_name
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_SUPPKEY, L_SHIPDATE, L_EXTENDEDPRICE, L_DISCOUNT},
                          {{L_SHIPDATE, PH_GE, LDate(1995, 1, 1)}, {L_SHIPDATE, PH_LE, LDate(1996, 12, 31)}});
        int j1 = p.Join(line, js, {1}, {0}, {0, 2, 3, 4, 6});              // l_orderkey, l_shipdate, ext, disc, n1
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_CUSTKEY});
        int j2 = p.Join(j1, ord, {0}, {0}, {1, 2, 3, 4, 6});               // l_shipdate, ext, disc, n1, o_custkey
        int j3 = p.Join(j2, jc, {4}, {0}, {0, 1, 2, 3, 6});                // l_shipdate, ext, disc, n1, n2
        int f = p.Filter(j3, {}, BoolExpr::OrOf({BoolExpr::AndOf({BoolExpr::C(3, PH_EQ, LStr(A)), BoolExpr::C(4, PH_EQ, LStr(B))}),
                                                 BoolExpr::AndOf({BoolExpr::C(3, PH_EQ, LStr(B)), BoolExpr::C(4, PH_EQ, LStr(A))})}));
        int proj = p.Project(f, {ProjExpr::Col(3), ProjExpr::Col(4), ProjExpr::Year(0), ProjExpr::Dec(DiscPrice(1, 2))});
        p.Agg(proj, {ProjExpr::Col(0), ProjExpr::Col(1), ProjExpr::Col(2)}, {{PH_A_SUM, {XC(3)}}});
        q->order = {{0, false}, {1, false}, {2, false}};
        q->ncols = 4;
        break;
    }
    case 8: {
        // Order <- Agg(year(o_orderdate); sum(case when n2.n_name = 'ARGENTINA' then volume else 0 end), sum(volume)) over eight tables:
        // part[p_type] x lineitem x orders[date range] x customer x nation n1 x region[r_name] and supplier x nation n2;
        // select list: o_year, the quotient of the two sums (DECIMAL `/`, typed as the dividend)
        int reg = p.Scan(&db.region, {R_REGIONKEY}, {{R_NAME, PH_EQ, LStr("AMERICA")}});
        int n1 = p.Scan(&db.nation, {N_NATIONKEY, N_REGIONKEY});
        int jn = p.Join(n1, reg, {1}, {0}, {0});                           // n_nationkey (of the region)
        int cust = p.Scan(&db.customer, {C_CUSTKEY, C_NATIONKEY});
        int jc = p.Join(cust, jn, {1}, {0}, {0});                          // c_custkey
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE}, {{O_ORDERDATE, PH_GE, LDate(1995, 1, 1)}, {O_ORDERDATE, PH_LE, LDate(1996, 12, 31)}});
        int jo = p.Join(ord, jc, {1}, {0}, {0, 2});                        // o_orderkey, o_orderdate
        int part = p.Scan(&db.part, {P_PARTKEY}, {{P_TYPE, PH_EQ, LStr("ECONOMY BURNISHED TIN")}});
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_PARTKEY, L_SUPPKEY, L_EXTENDEDPRICE, L_DISCOUNT});
        int j1 = p.Join(line, part, {1}, {0}, {0, 2, 3, 4});               // l_orderkey, l_suppkey, ext, disc
        int j2 = p.Join(j1, jo, {0}, {0}, {1, 2, 3, 5});                   // l_suppkey, ext, disc, o_orderdate
        int n2 = p.Scan(&db.nation, {N_NATIONKEY, N_NAME});
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY});
        int js = p.Join(supp, n2, {1}, {0}, {0, 3});                       // s_suppkey, n2.n_name
        int j3 = p.Join(j2, js, {0}, {0}, {1, 2, 3, 5});                   // ext, disc, o_orderdate, n2.n_name
        ProjExpr mine = ProjExpr::CaseOf(BoolExpr::C(3, PH_EQ, LStr("ARGENTINA")), DiscPrice(0, 1), {XK(0)});
        p.Agg(j3, {ProjExpr::Year(2)}, {AggExpr::Of(PH_A_SUM, mine), {PH_A_SUM, DiscPrice(0, 1)}});
        q->outputs = {ProjExpr::Col(0), ProjExpr::DecQuo(1, 2)};
        q->order = {{0, false}};
        q->ncols = 2;
        break;
    }
    case 11: {
        // Order <- Agg(ps_partkey; sum(ps_supplycost * ps_availqty)) [HAVING sum > scalar] <- Join(ps_suppkey = s_suppkey) probe Scan(partsupp),
        //   build <- Join(s_nationkey = n_nationkey) probe Scan(supplier), build Scan(nation, n_name = 'JAPAN'); the scalar subquery is the same
        //   join under an ungrouped aggregate
        auto chain = [&](ResidentPlan &pl) {
            int nat = pl.Scan(&db.nation, {N_NATIONKEY}, {{N_NAME, PH_EQ, LStr("JAPAN")}});
            int supp = pl.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY});
            int js = pl.Join(supp, nat, {1}, {0}, {0});                    // s_suppkey
            int ps = pl.Scan(&db.partsupp, {PS_PARTKEY, PS_SUPPKEY, PS_SUPPLYCOST, PS_AVAILQTY});
            return pl.Join(ps, js, {1}, {0}, {0, 2, 3});                   // ps_partkey, ps_supplycost, ps_availqty
        };
        const std::vector<ph_rpn> value = {XC(1), XC(2), XO(PH_X_MUL)};
        p.Agg(chain(p), {ProjExpr::Col(0)}, {{PH_A_SUM, value}});
        q->scalar = std::make_shared<TpchQuery>();
        q->scalar->id = 11;
        q->scalar->plan.Agg(chain(q->scalar->plan), {}, {{PH_A_SUM, value}});
        q->scalar->ncols = 1;
        if (!q->scalar->plan.error.empty()) return q->scalar->plan.error;
        q->scalarFactor = 0.0001f;
        q->havingCol = 1;
        q->order = {{1, true}};                                            // ORDER BY value DESC
        q->ncols = 2;
        break;
    }
    case 12: {
        // Order <- Agg(l_shipmode; sum(case when prio = '1-URGENT' or prio = '2-HIGH' then 1 else 0 end), sum(case when prio <> .. and prio <> ..))
        //   <- Join(l_orderkey = o_orderkey) probe Scan(lineitem, shipmode IN ('FOB','TRUCK'), commit < receipt, ship < commit, receipt range)
        BoolExpr where = BoolExpr::AndOf({BoolExpr::In(L_SHIPMODE, {LStr("FOB"), LStr("TRUCK")}), BoolExpr::CC(L_COMMITDATE, PH_LT, L_RECEIPTDATE),
                                          BoolExpr::CC(L_SHIPDATE, PH_LT, L_COMMITDATE)});
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_SHIPMODE}, {{L_RECEIPTDATE, PH_GE, LDate(1996, 1, 1)}, {L_RECEIPTDATE, PH_LT, LDate(1997, 1, 1)}}, where);
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_ORDERPRIORITY});
        int j = p.Join(line, ord, {0}, {0}, {1, 3});                       // l_shipmode, o_orderpriority
        ProjExpr high = ProjExpr::CaseOf(BoolExpr::OrOf({BoolExpr::C(1, PH_EQ, LStr("1-URGENT")), BoolExpr::C(1, PH_EQ, LStr("2-HIGH"))}), {XK(1)}, {XK(0)}, true);
        ProjExpr low = ProjExpr::CaseOf(BoolExpr::AndOf({BoolExpr::C(1, PH_NE, LStr("1-URGENT")), BoolExpr::C(1, PH_NE, LStr("2-HIGH"))}), {XK(1)}, {XK(0)}, true);
        p.Agg(j, {ProjExpr::Col(0)}, {AggExpr::Of(PH_A_SUM, high), AggExpr::Of(PH_A_SUM, low)});
        q->order = {{0, false}};
        q->ncols = 3;
        break;
    }
    case 14: {
        // Agg(; sum(case when p_type like 'PROMO%' then e * (1 - d) else 0 end), sum(e * (1 - d))) <- Join(l_partkey = p_partkey)
        //   probe Scan(lineitem, shipdate in April 1996), build Scan(part); select list: 100.00 * a / b — FLOAT arithmetic
        int line = p.Scan(&db.lineitem, {L_PARTKEY, L_EXTENDEDPRICE, L_DISCOUNT}, {{L_SHIPDATE, PH_GE, LDate(1996, 4, 1)}, {L_SHIPDATE, PH_LT, LDate(1996, 5, 1)}});
        int part = p.Scan(&db.part, {P_PARTKEY, P_TYPE});
        int j = p.Join(line, part, {0}, {0}, {1, 2, 4});                   // ext, disc, p_type
        ProjExpr promo = ProjExpr::CaseOf(BoolExpr::C(2, PH_LIKE, LStr("PROMO%")), DiscPrice(0, 1), {XK(0)});
        p.Agg(j, {}, {AggExpr::Of(PH_A_SUM, promo), {PH_A_SUM, DiscPrice(0, 1)}});
        FloatOp hundred; hundred.op = FloatOp::Const; hundred.k = 100.00f;
        FloatOp a; a.op = FloatOp::Col; a.col = 0;
        FloatOp b; b.op = FloatOp::Col; b.col = 1;
        FloatOp mul; mul.op = FloatOp::Mul;
        FloatOp div; div.op = FloatOp::Div;
        q->outputs = {ProjExpr::Float({hundred, a, mul, b, div})};
        q->ncols = 1;
        break;
    }
    case 19: {
        // Agg(; sum(e * (1 - d))) <- Filter(OR of the three conjunctions) <- Join(l_partkey = p_partkey); the conjuncts common to all
        // three branches (join condition, l_shipmode IN ('AIR','AIR REG'), l_shipinstruct = 'DELIVER IN PERSON') are what
        // DistributivityRule (rule_distributivity.go) + filter push-down lift out of the OR and into the lineitem scan
        int line = p.Scan(&db.lineitem, {L_PARTKEY, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT}, {{L_SHIPINSTRUCT, PH_EQ, LStr("DELIVER IN PERSON")}},
                          BoolExpr::In(L_SHIPMODE, {LStr("AIR"), LStr("AIR REG")}));
        int part = p.Scan(&db.part, {P_PARTKEY, P_BRAND, P_SIZE, P_CONTAINER});
        int j = p.Join(line, part, {0}, {0}, {1, 2, 3, 5, 6, 7});          // qty, ext, disc, p_brand, p_size, p_container
        struct Br { const char *brand; std::vector<const char *> cntr; int q1, q2, sz; };
        const std::vector<Br> brs = {{"Brand#23", {"SM CASE", "SM BOX", "SM PACK", "SM PKG"}, 5, 15, 5},
                                     {"Brand#15", {"MED BAG", "MED BOX", "MED PKG", "MED PACK"}, 14, 24, 10},
                                     {"Brand#44", {"LG CASE", "LG BOX", "LG PACK", "LG PKG"}, 28, 38, 15}};
        std::vector<BoolExpr> ors;
        for (auto &b : brs) {
            std::vector<Literal> cn;
            for (auto c : b.cntr) cn.push_back(LStr(c));
            ors.push_back(BoolExpr::AndOf({BoolExpr::C(3, PH_EQ, LStr(b.brand)), BoolExpr::In(5, cn), BoolExpr::C(0, PH_GE, LInt(b.q1)), BoolExpr::C(0, PH_LE, LInt(b.q2)),
                                           BoolExpr::C(4, PH_GE, LInt(1)), BoolExpr::C(4, PH_LE, LInt(b.sz))}));
        }
        int f = p.Filter(j, {}, BoolExpr::OrOf(ors));
        p.Agg(f, {}, {{PH_A_SUM, DiscPrice(1, 2)}});
        q->ncols = 1;
        break;
    }
    case 15: {
        // Order(s_suppkey) <- Join(s_suppkey = supplier_no) probe Scan(supplier), build Join(total_revenue = max) probe CTE,
        //   build Agg(; max(total_revenue)) <- CTE;  CTE q15_revenue0 = Agg(l_suppkey; sum(l_extendedprice * (1 - l_discount))) <- Scan(lineitem,
        //   l_shipdate in [1995-12-01, + 3 months)). ONE resident plan whose ROOT is the final join (its rows come back: ph_plan_fetch_rows); the CTE
        //   node has two parents (lowered once per run); `=` on DECIMAL is the hash join the reference runs it as (executeSelect has no DECIMAL
        //   case for FuncEqual).
        int line = p.Scan(&db.lineitem, {L_SUPPKEY, L_EXTENDEDPRICE, L_DISCOUNT}, {{L_SHIPDATE, PH_GE, LDate(1995, 12, 1)}, {L_SHIPDATE, PH_LT, LDate(1996, 3, 1)}});
        int cte = p.Agg(line, {ProjExpr::Col(0)}, {{PH_A_SUM, DiscPrice(1, 2)}});
        int top = p.Agg(cte, {}, {{PH_A_MAX, {XC(1)}}});
        int j1 = p.Join(cte, top, {1}, {0}, {0, 1});
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NAME, S_ADDRESS, S_PHONE});
        p.Join(supp, j1, {0}, {0}, {0, 1, 2, 3, 5});
        q->order = {{0, false}};
        q->ncols = 5;
        break;
    }
    case 17: {
        // Project(sum / 7.0) <- Agg(; sum(l_extendedprice)) <- Filter(l_quantity < 0.2 * avg) <- Join(l_partkey = sub.l_partkey)
        //   probe <- Join(l_partkey = p_partkey) probe Scan(lineitem), build Scan(part, p_brand = 'Brand#54', p_container = 'LG BAG')
        //   build Agg(l_partkey; avg(l_quantity)) <- Scan(lineitem)      (the correlated subquery, by its correlation key)
        // avg(INTEGER) is DOUBLE and `*` has only (T, T) overloads: the FLOAT literal is widened and the predicate is DOUBLE arithmetic — a
        // PH_PE_FLOAT flag column (float64(l_quantity) < float64(0.2f) * (float64(sum) / float64(count)): the average travels as its SUM and
        // COUNT) under the Filter. The select list's `/ 7.0` is FLOAT arithmetic over the one result row (the aggregate's output phase).
        int subScan = p.Scan(&db.lineitem, {L_PARTKEY, L_QUANTITY});
        int sub = p.Agg(subScan, {ProjExpr::Col(0)}, {{PH_A_SUM, {XC(1)}}, {PH_A_COUNT, {XC(1)}}});   // l_partkey, sum, count
        int part = p.Scan(&db.part, {P_PARTKEY}, {{P_BRAND, PH_EQ, LStr("Brand#54")}, {P_CONTAINER, PH_EQ, LStr("LG BAG")}});
        int line = p.Scan(&db.lineitem, {L_PARTKEY, L_QUANTITY, L_EXTENDEDPRICE});
        int j1 = p.Join(line, part, {0}, {0}, {0, 1, 2});
        int j2 = p.Join(j1, sub, {0}, {0}, {1, 2, 4, 5});                      // l_quantity, l_extendedprice, sum(l_quantity), count(l_quantity)
        auto fcol = [](int c) { FloatOp o; o.op = FloatOp::Col; o.col = c; return o; };
        auto fk = [](float k) { FloatOp o; o.op = FloatOp::Const; o.k = k; return o; };
        auto fop = [](FloatOp::Op op) { FloatOp o; o.op = op; return o; };
        int pr = p.Project(j2, {ProjExpr::FloatTruth({fcol(0), fk(0.2f), fcol(2), fcol(3), fop(FloatOp::Div), fop(FloatOp::Mul), fop(FloatOp::Lt)}, true), ProjExpr::Col(1)});
        int f = p.Filter(pr, {{0, PH_EQ, LInt(1)}});
        p.Agg(f, {}, {{PH_A_SUM, {XC(1)}}});
        q->outputs = {ProjExpr::Float({fcol(0), fk(7.0f), fop(FloatOp::Div)})};
        q->ncols = 1;
        break;
    }
    case 18: {
        // Limit <- Order <- Agg(c_name, c_custkey, o_orderkey, o_orderdate, o_totalprice; sum(l_quantity))
        //   <- Join(l_orderkey = o_orderkey) probe Scan(lineitem)
        //        build <- Join(o_custkey = c_custkey) probe <- SEMI Join(o_orderkey = l_orderkey) probe Scan(orders)
        //                                                        build Filter(sum > 314) <- Agg(l_orderkey; sum(l_quantity)) <- Scan(lineitem)
        // (the IN subquery: an aggregate below the join, its HAVING the Filter above it)
        int subScan = p.Scan(&db.lineitem, {L_ORDERKEY, L_QUANTITY});
        int sub = p.Agg(subScan, {ProjExpr::Col(0)}, {{PH_A_SUM, {XC(1)}}});
        int having = p.Filter(sub, {{1, PH_GT, LInt(314)}});
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_TOTALPRICE});
        int j1 = p.Join(ord, having, {0}, {0}, {0, 1, 2, 3}, JoinSemi);
        int cust = p.Scan(&db.customer, {C_CUSTKEY, C_NAME});
        int j2 = p.Join(j1, cust, {1}, {0}, {0, 2, 3, 4, 5});                 // o_orderkey, o_orderdate, o_totalprice, c_custkey, c_name
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_QUANTITY});
        int j3 = p.Join(line, j2, {0}, {0}, {1, 6, 5, 2, 3, 4});              // l_quantity, c_name, c_custkey, o_orderkey, o_orderdate, o_totalprice
        p.Agg(j3, {ProjExpr::Col(1), ProjExpr::Col(2), ProjExpr::Col(3), ProjExpr::Col(4), ProjExpr::Col(5)}, {{PH_A_SUM, {XC(0)}}});
        q->order = {{4, true}, {3, false}};
        q->limit = 100;
        q->ncols = 6;
        break;
    }
    case 20: {
        // Order(s_name) <- Project(s_name, s_address) <- SEMI Join(s_suppkey = ps_suppkey) probe Join(s_nationkey = n_nationkey)[Scan(supplier),
        //   Scan(nation, n_name = 'VIETNAM')], build Filter(ps_availqty > 0.5 * sum [FLOAT]) <- Join((ps_partkey, ps_suppkey) = (l_partkey, l_suppkey))
        //   probe SEMI Join(ps_partkey = p_partkey)[Scan(partsupp), Scan(part, p_name like 'lime%')], build Agg(l_partkey, l_suppkey; sum(l_quantity))
        //   <- Scan(lineitem, 1993). ONE resident plan whose ROOT is the SEMI join (rows: s_name, s_address); the FLOAT predicate is a PH_PE_FLOAT flag
        //   column (float32(ps_availqty) > 0.5f * float32(sum)) under the Filter.
        int subScan = p.Scan(&db.lineitem, {L_PARTKEY, L_SUPPKEY, L_QUANTITY}, {{L_SHIPDATE, PH_GE, LDate(1993, 1, 1)}, {L_SHIPDATE, PH_LT, LDate(1994, 1, 1)}});
        int sub = p.Agg(subScan, {ProjExpr::Col(0), ProjExpr::Col(1)}, {{PH_A_SUM, {XC(2)}}});
        int part = p.Scan(&db.part, {P_PARTKEY}, {{P_NAME, PH_LIKE, LStr("lime%")}});
        int ps = p.Scan(&db.partsupp, {PS_PARTKEY, PS_SUPPKEY, PS_AVAILQTY});
        int j1 = p.Join(ps, part, {0}, {0}, {0, 1, 2}, JoinSemi);
        int j2 = p.Join(j1, sub, {0, 1}, {0, 1}, {1, 2, 5});                   // ps_suppkey, ps_availqty, sum(l_quantity)
        auto fcol = [](int c) { FloatOp o; o.op = FloatOp::Col; o.col = c; return o; };
        FloatOp half; half.op = FloatOp::Const; half.k = 0.5f;
        FloatOp mul; mul.op = FloatOp::Mul;
        FloatOp gt; gt.op = FloatOp::Gt;
        int pr = p.Project(j2, {ProjExpr::Col(0), ProjExpr::FloatTruth({fcol(1), half, fcol(2), mul, gt}, false)});
        int good = p.Filter(pr, {{1, PH_EQ, LInt(1)}});
        int nat = p.Scan(&db.nation, {N_NATIONKEY}, {{N_NAME, PH_EQ, LStr("VIETNAM")}});
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY, S_NAME, S_ADDRESS});
        int js = p.Join(supp, nat, {1}, {0}, {0, 2, 3});                       // s_suppkey, s_name, s_address
        p.Join(js, good, {0}, {0}, {1, 2}, JoinSemi);
        q->order = {{0, false}};                            // ORDER BY s_name
        q->ncols = 2;
        break;
    }
    case 21: {
        // Limit <- Order(numwait desc, s_name) <- Agg(s_name; count(*)) <- ANTI Join[l3] <- SEMI Join[l2] <- Join(orders[o_orderstatus = 'F'])
        //   <- Join(supplier x nation[BRAZIL]) probe Scan(lineitem l1, l_receiptdate > l_commitdate).
        // The EXISTS / NOT EXISTS joins carry l2.l_suppkey <> l1.l_suppkey beside the key: the INNER join on l_orderkey emits the key matches, a
        // Filter compares the two supplier columns, and the l1 rows that keep a pair — by lineitem's primary key (l_orderkey, l_linenumber) — are
        // an aggregate below the SEMI / ANTI join that closes the step (the l1 subtree has two parents per step).
        BoolExpr late = BoolExpr::CC(L_RECEIPTDATE, PH_GT, L_COMMITDATE);
        auto l1Side = [&]() {
            int nat = p.Scan(&db.nation, {N_NATIONKEY}, {{N_NAME, PH_EQ, LStr("BRAZIL")}});
            int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY, S_NAME});
            int js = p.Join(supp, nat, {1}, {0}, {0, 2});                       // s_suppkey, s_name
            int l1 = p.Scan(&db.lineitem, {L_ORDERKEY, L_SUPPKEY, L_LINENUMBER}, {}, late);
            int j1 = p.Join(l1, js, {1}, {0}, {0, 1, 2, 4});                    // l_orderkey, l_suppkey, l_linenumber, s_name
            int ord = p.Scan(&db.orders, {O_ORDERKEY}, {{O_ORDERSTATUS, PH_EQ, LStr("F")}});
            return p.Join(j1, ord, {0}, {0}, {0, 1, 2, 3});
        };
        auto withOtherSupplier = [&](int rows, bool onlyLate) {
            int other = p.Scan(&db.lineitem, {L_ORDERKEY, L_SUPPKEY}, {}, onlyLate ? late : BoolExpr());
            int pairs = p.Join(rows, other, {0}, {0}, {0, 2, 1, 5});            // l_orderkey, l_linenumber, l1.l_suppkey, other.l_suppkey
            int differ = p.Filter(pairs, {}, BoolExpr::CC(2, PH_NE, 3));
            return p.Agg(differ, {ProjExpr::Col(0), ProjExpr::Col(1)}, {{PH_A_COUNT_STAR, {}}});
        };
        int l1 = l1Side();
        int j4;
        if (!getenv("PH_Q21_NO_RESIDUAL")) {
            // the EXISTS / NOT EXISTS as joins with a RESIDUAL condition (the library filters the key matches and marks the l1 rows that keep one)
            BoolExpr otherSupplier = BoolExpr::CC(5, PH_NE, 1);                   // [l1: 0..3 | other: 4 l_orderkey, 5 l_suppkey]
            int l2 = p.Scan(&db.lineitem, {L_ORDERKEY, L_SUPPKEY});
            int j3 = p.Join(l1, l2, {0}, {0}, {0, 1, 2, 3}, JoinSemi, otherSupplier);
            int l3 = p.Scan(&db.lineitem, {L_ORDERKEY, L_SUPPKEY}, {}, late);
            j4 = p.Join(j3, l3, {0}, {0}, {3}, JoinAnti, otherSupplier);         // s_name
        } else {
            // (round 3's form, kept for the A/B: pairs + Filter + an aggregate by lineitem's primary key below a two-key SEMI / ANTI join; the l1
            // subtree has two parents per step and is lowered once per run)
            int e2 = withOtherSupplier(l1, false);
            int j3 = p.Join(l1, e2, {0, 2}, {0, 1}, {0, 1, 2, 3}, JoinSemi);
            int n3 = withOtherSupplier(j3, true);
            j4 = p.Join(j3, n3, {0, 2}, {0, 1}, {3}, JoinAnti);                   // s_name
        }
        p.Agg(j4, {ProjExpr::Col(0)}, {{PH_A_COUNT_STAR, {}}});
        q->order = {{1, true}, {0, false}};
        q->limit = 100;
        q->ncols = 2;
        q->topkAgg = 0; q->topkDesc = true;   // ORDER BY numwait DESC .. LIMIT 100: the groups that can be among the first hundred (ties kept)
        break;
    }
    case 22: {
        // Order(cntrycode) <- Agg(cntrycode; count(*), sum(c_acctbal)) <- ANTI Join(c_custkey = o_custkey) probe Filter(cntrycode IN (..)) <-
        //   Project(substring(c_phone from 1 for 2) as cntrycode, ..) <- Scan(customer, c_acctbal > scalar), build Scan(orders);
        //   scalar = Agg(; avg(c_acctbal)) <- Filter(cntrycode IN (..)) <- Project <- Scan(customer, c_acctbal > 0.00 [FLOAT literal: float32 compare])
        std::vector<Literal> codes;
        for (const char *c : {"10", "11", "26", "22", "19", "20", "27"}) codes.push_back(LStr(c));
        q->scalar = std::make_shared<TpchQuery>();
        q->scalar->id = 22;
        {
            ResidentPlan &s = q->scalar->plan;
            int cust = s.Scan(&db.customer, {C_PHONE, C_ACCTBAL}, {{C_ACCTBAL, PH_GT, LFloat(0.00f)}});
            int pr = s.Project(cust, {ProjExpr::Substr(0, 1, 2), ProjExpr::Col(1)});
            int f = s.Filter(pr, {}, BoolExpr::In(0, codes));
            s.Agg(f, {}, {{PH_A_AVG, {XC(1)}}});
            q->scalar->ncols = 1;
            if (!s.error.empty()) return s.error;
        }
        int cust = p.Scan(&db.customer, {C_CUSTKEY, C_PHONE, C_ACCTBAL}, {{C_ACCTBAL, PH_GT, LDec(0, 2)}});   // (the literal: the scalar, at run time)
        q->scalarScanNode = cust; q->scalarConjunct = 0;
        int pr = p.Project(cust, {ProjExpr::Col(0), ProjExpr::Substr(1, 1, 2), ProjExpr::Col(2)});
        int f = p.Filter(pr, {}, BoolExpr::In(1, codes));
        int ord = p.Scan(&db.orders, {O_CUSTKEY});
        int j = p.Join(f, ord, {0}, {0}, {1, 2}, JoinAnti);                    // cntrycode, c_acctbal
        p.Agg(j, {ProjExpr::Col(0)}, {{PH_A_COUNT_STAR, {}}, {PH_A_SUM, {XC(1)}}});
        q->order = {{0, false}};
        q->ncols = 3;
        break;
    }
    case 16: {
        // Order(supplier_cnt desc, p_brand, p_type, p_size) <- Agg(p_brand, p_type, p_size; count(DISTINCT ps_suppkey))
        //   <- ANTI Join(ps_suppkey = s_suppkey) [NOT IN] probe Join(ps_partkey = p_partkey) probe Scan(partsupp),
        //        build Scan(part, p_brand <> 'Brand#35' and p_type not like 'ECONOMY BURNISHED%' and p_size in (..));  build Scan(supplier, s_comment like ..)
        std::vector<Literal> sizes;
        for (int v : {14, 7, 21, 24, 35, 33, 2, 20}) sizes.push_back(LInt(v));
        int part = p.Scan(&db.part, {P_PARTKEY, P_BRAND, P_TYPE, P_SIZE}, {{P_BRAND, PH_NE, LStr("Brand#35")}},
                          BoolExpr::AndOf({BoolExpr::C(P_TYPE, PH_NOTLIKE, LStr("ECONOMY BURNISHED%")), BoolExpr::In(P_SIZE, sizes)}));
        int ps = p.Scan(&db.partsupp, {PS_PARTKEY, PS_SUPPKEY});
        int j1 = p.Join(ps, part, {0}, {0}, {1, 3, 4, 5});                     // ps_suppkey, p_brand, p_type, p_size
        int supp = p.Scan(&db.supplier, {S_SUPPKEY}, {{S_COMMENT, PH_LIKE, LStr("%Customer%Complaints%")}});
        int j2 = p.Join(j1, supp, {0}, {0}, {0, 1, 2, 3}, JoinAnti);
        p.Agg(j2, {ProjExpr::Col(1), ProjExpr::Col(2), ProjExpr::Col(3)}, {{PH_A_COUNT_DISTINCT, {XC(0)}}});
        q->order = {{3, true}, {0, false}, {1, false}, {2, false}};
        q->ncols = 4;
        break;
    }
    case 13: {
        // Order(custdist desc, c_count desc) <- Agg(c_count; count(*)) <- Agg(c_custkey; count(o_orderkey))
        //   <- LEFT Join(c_custkey = o_custkey) probe Scan(customer), build Scan(orders, o_comment not like '%pending%accounts%')
        // count() over the NULL-extended side is 0 for a customer without orders and finalises to NULL: the NULL group of the aggregate above
        int cust = p.Scan(&db.customer, {C_CUSTKEY});
        int ord = p.Scan(&db.orders, {O_CUSTKEY, O_ORDERKEY}, {{O_COMMENT, PH_NOTLIKE, LStr("%pending%accounts%")}});
        int j = p.Join(cust, ord, {0}, {0}, {0, 2}, JoinLeft);                 // c_custkey, o_orderkey
        int inner = p.Agg(j, {ProjExpr::Col(0)}, {{PH_A_COUNT, {XC(1)}}});     // c_custkey, c_count
        p.Agg(inner, {ProjExpr::Col(1)}, {{PH_A_COUNT_STAR, {}}});
        q->order = {{1, true}, {0, true}};
        q->ncols = 2;
        break;
    }
    case 2: {
        // Limit <- Order(s_acctbal desc, n_name, s_name, p_partkey) <- Join((ps_partkey, ps_supplycost) = (sub.ps_partkey, sub.min))
        //   probe Join(ps_partkey = p_partkey) [RS = partsupp x supplier x nation x region[MIDDLE EAST]] x part[p_size = 48, p_type like '%TIN']
        //   build Agg(ps_partkey; min(ps_supplycost)) <- RS      (the correlated subquery by its key; RS has two parents: lowered once per run)
        // ONE plan whose root is the final join (ph_plan_fetch_rows): four VARCHAR columns of supplier come back gathered on the device
        int reg = p.Scan(&db.region, {R_REGIONKEY}, {{R_NAME, PH_EQ, LStr("MIDDLE EAST")}});
        int nat = p.Scan(&db.nation, {N_NATIONKEY, N_NAME, N_REGIONKEY});
        int jn = p.Join(nat, reg, {2}, {0}, {0, 1});                           // n_nationkey, n_name
        int supp = p.Scan(&db.supplier, {S_SUPPKEY, S_NATIONKEY, S_ACCTBAL, S_NAME, S_ADDRESS, S_PHONE, S_COMMENT});
        int js = p.Join(supp, jn, {1}, {0}, {0, 2, 3, 4, 5, 6, 8});            // s_suppkey, s_acctbal, s_name, s_address, s_phone, s_comment, n_name
        int ps = p.Scan(&db.partsupp, {PS_PARTKEY, PS_SUPPKEY, PS_SUPPLYCOST});
        int rs = p.Join(ps, js, {1}, {0}, {0, 2, 4, 5, 6, 7, 8, 9});           // ps_partkey, ps_supplycost, s_acctbal, s_name, s_address, s_phone, s_comment, n_name
        int sub = p.Agg(rs, {ProjExpr::Col(0)}, {{PH_A_MIN, {XC(1)}}});        // ps_partkey, min(ps_supplycost)
        int part = p.Scan(&db.part, {P_PARTKEY, P_MFGR}, {{P_SIZE, PH_EQ, LInt(48)}}, BoolExpr::C(P_TYPE, PH_LIKE, LStr("%TIN")));
        int jp = p.Join(rs, part, {0}, {0}, {0, 1, 2, 3, 4, 5, 6, 7, 9});      // + p_mfgr
        p.Join(jp, sub, {0, 1}, {0, 1}, {2, 3, 7, 0, 8, 4, 5, 6});             // s_acctbal, s_name, n_name, p_partkey, p_mfgr, s_address, s_phone, s_comment
        q->order = {{0, true}, {2, false}, {1, false}, {3, false}};
        q->limit = 100;
        q->ncols = 8;
        q->rowsTopkCol = 0; q->rowsTopkDesc = true;   // ORDER BY s_acctbal DESC .. LIMIT 100: the rows that can be among the first hundred
        break;
    }
    case 10: {
        // Limit <- Order(revenue desc) <- Agg(c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment; sum(e * (1 - d)))
        //   <- Join(c_nationkey = n_nationkey) <- Join(o_custkey = c_custkey) <- Join(l_orderkey = o_orderkey) probe Scan(lineitem, l_returnflag = 'R'),
        //        build Scan(orders, o_orderdate in [1993-03-01, + 3 months))
        int ord = p.Scan(&db.orders, {O_ORDERKEY, O_CUSTKEY}, {{O_ORDERDATE, PH_GE, LDate(1993, 3, 1)}, {O_ORDERDATE, PH_LT, LDate(1993, 6, 1)}});
        int line = p.Scan(&db.lineitem, {L_ORDERKEY, L_EXTENDEDPRICE, L_DISCOUNT}, {{L_RETURNFLAG, PH_EQ, LStr("R")}});
        int j1 = p.Join(line, ord, {0}, {0}, {1, 2, 4});                       // ext, disc, o_custkey
        int cust = p.Scan(&db.customer, {C_CUSTKEY, C_NAME, C_ACCTBAL, C_PHONE, C_NATIONKEY, C_ADDRESS, C_COMMENT});
        int j2 = p.Join(j1, cust, {2}, {0}, {0, 1, 3, 4, 5, 6, 7, 8, 9});      // ext, disc, c_custkey, c_name, c_acctbal, c_phone, c_nationkey, c_address, c_comment
        int nat = p.Scan(&db.nation, {N_NATIONKEY, N_NAME});
        int j3 = p.Join(j2, nat, {6}, {0}, {0, 1, 2, 3, 4, 5, 10, 7, 8});      // ext, disc, c_custkey, c_name, c_acctbal, c_phone, n_name, c_address, c_comment
        p.Agg(j3, {ProjExpr::Col(2), ProjExpr::Col(3), ProjExpr::Col(4), ProjExpr::Col(5), ProjExpr::Col(6), ProjExpr::Col(7), ProjExpr::Col(8)},
              {{PH_A_SUM, DiscPrice(0, 1)}});
        // select list: c_custkey, c_name, revenue, c_acctbal, n_name, c_address, c_phone, c_comment; ORDER BY revenue DESC LIMIT 20
        q->outputs = {ProjExpr::Col(0), ProjExpr::Col(1), ProjExpr::Col(7), ProjExpr::Col(2), ProjExpr::Col(4), ProjExpr::Col(5), ProjExpr::Col(3), ProjExpr::Col(6)};
        q->order = {{2, true}};
        q->limit = 20;
        q->topkAgg = 0; q->topkDesc = true;
        q->ncols = 8;
        break;
    }
    default:
        return "no resident plan for TPC-H query " + std::to_string(id);
    }
    return p.error;
}

std::string RunTpchQuery(ph_ctx *ctx, const TpchQuery &q, std::vector<std::string> *lines, std::string *explain, ph_comm *comm) {
    std::vector<Compare> having = q.having;
    ResidentPlan mainPlan = q.plan;   // (Q22 patches a scan literal with its scalar subquery's value)
    if (q.scalar) {   // the uncorrelated scalar subquery first: one row, one DECIMAL value
        gpuResidentPlanExecutor sub(ctx, q.scalar->plan);
        if (comm) sub.SetComm(comm);
        std::string e = sub.Init();
        if (!e.empty()) return "Init (scalar subquery): " + e;
        Chunk out;
        std::string err;
        OperatorResult r = sub.Execute(nullptr, &out, &err);
        if (r == InvalidOpResult) return "Execute (scalar subquery): " + err;
        if (r == Done || out.Card() != 1 || out.Data[0]->_Typ.GetInternalType() != PT_DECIMAL) { sub.Close(); return "scalar subquery: expected one DECIMAL value"; }
        Vector::Unified u;
        out.Data[0]->ToUnifiedFormat(1, &u);
        const int64_t idx = u.sel->GetIndex(0);
        if (!u.mask->RowIsValid((uint64_t)idx)) { sub.Close(); lines->clear(); return ""; }   // NULL threshold: the comparison selects nothing
        const Decimal sv = reinterpret_cast<const Decimal *>(u.data)[idx];
        sub.Close();
        if (q.scalarScanNode >= 0) {
            Compare &c = mainPlan.nodes[(size_t)q.scalarScanNode].conjuncts[(size_t)q.scalarConjunct];
            int64_t fl = 0;
            if (!DecimalFloorUnscaled(sv, c.k.scale, &fl)) return "scalar subquery: value out of range";
            c.k.i = fl;
        } else {
            volatile float v = (float)DecimalToDouble(sv);
            volatile float t = v * q.scalarFactor;
            having.push_back(Compare{q.havingCol, PH_GT, LFloat((float)t)});
        }
    }
    static const bool timing = getenv("PH_HOST_TIMING") != nullptr;   // where a query's HOST time goes (stderr, one line per query)
    const double tq0 = now_s();
    gpuResidentPlanExecutor agg(ctx, mainPlan);
    if (comm) agg.SetComm(comm);
    if (!having.empty()) agg.SetHaving(having);
    if (!q.outputs.empty()) agg.SetOutputs(q.outputs);
    if (q.topkAgg >= 0 && q.limit > 0) agg.SetTopK(q.topkAgg, q.topkDesc, q.limit);
    if (q.rowsTopkCol >= 0 && q.limit > 0) agg.SetRowsTopK(q.rowsTopkCol, q.rowsTopkDesc, q.limit);
    std::string e = agg.Init();
    if (!e.empty()) return "Init: " + e;
    std::unique_ptr<gpuOrderExecutor> ord;
    std::unique_ptr<limitExecutor> lim;
    OperatorExec *root = &agg;
    if (!q.order.empty()) {
        ord.reset(new gpuOrderExecutor(ctx, q.order, root));
        e = ord->Init();
        if (!e.empty()) return "Init: " + e;
        root = ord.get();
    }
    if (q.limit >= 0) {
        lim.reset(new limitExecutor((uint64_t)q.limit, 0, root));
        lim->Init();
        root = lim.get();
    }
    lines->clear();
    const double tq1 = now_s();
    double tq2 = 0;
    for (;;) {   // the pull loop of execOps (executor.go:151-188)
        Chunk out;
        std::string err;
        OperatorResult r = root->Execute(nullptr, &out, &err);
        if (tq2 == 0) tq2 = now_s();
        if (r == InvalidOpResult) return "Execute: " + err;
        if (r == Done) break;
        std::string text;
        out.AppendText(&text);
        size_t pos = 0;
        while (pos < text.size()) {
            size_t nl = text.find('\n', pos);
            lines->push_back(text.substr(pos, nl - pos));
            pos = nl + 1;
        }
    }
    const double tq3 = now_s();
    if (explain) *explain = agg.Explain();
    if (lim) lim->Close();
    if (ord) ord->Close();
    agg.Close();
    if (timing) fprintf(stderr, "host timing Q%d: init %.1f us, first Execute %.1f us, rest of the pull loop + text %.1f us, close %.1f us\n", q.id, (tq1 - tq0) * 1e6,
                        (tq2 - tq1) * 1e6, (tq3 - tq2) * 1e6, (now_s() - tq3) * 1e6);
    return "";
}

}  // namespace plan

// ---------------------------------------------------------------- C entry points

static thread_local std::string g_host_err;

extern "C" const char *planhost_last_error(void) { return g_host_err.c_str(); }

extern "C" int planhost_tpch_load(ph_ctx *ctx, int64_t sf_num, int64_t sf_den, void **db_out) {
    if (!ctx || !db_out || sf_num <= 0 || sf_den <= 0) { g_host_err = "planhost_tpch_load: bad arguments"; return PH_EINVAL; }
    auto *db = new plan::TpchDatabase();
    std::string e = db->Load(ctx, sf_num, sf_den);
    if (!e.empty()) { g_host_err = e; delete db; return PH_EHIP; }
    *db_out = db;
    return PH_OK;
}

extern "C" int64_t planhost_tpch_rows(void *dbp, const char *table) {
    auto *db = (plan::TpchDatabase *)dbp;
    if (!db || !table) return -1;
    const std::string t = table;
    const plan::ResidentTable *rt = t == "lineitem" ? &db->lineitem : t == "orders" ? &db->orders : t == "customer" ? &db->customer : t == "part" ? &db->part :
                                    t == "partsupp" ? &db->partsupp : t == "supplier" ? &db->supplier : t == "nation" ? &db->nation : t == "region" ? &db->region : nullptr;
    return rt && rt->table ? ph_table_rows(rt->table) : -1;
}

extern "C" int planhost_tpch_load_shard(ph_ctx *ctx, int64_t sf_num, int64_t sf_den, int32_t rank, int32_t nranks, void **db_out) {
    if (!ctx || !db_out || sf_num <= 0 || sf_den <= 0 || nranks < 1 || rank < 0 || rank >= nranks) { g_host_err = "planhost_tpch_load_shard: bad arguments"; return PH_EINVAL; }
    auto *db = new plan::TpchDatabase();
    std::string e = db->Load(ctx, sf_num, sf_den, rank, nranks);
    if (!e.empty()) { g_host_err = e; delete db; return PH_EHIP; }
    *db_out = db;
    return PH_OK;
}

static int tpch_run(void *dbp, ph_comm *comm, int32_t query, int32_t repeat, int32_t warmup, double *ms_avg, double *ms_min, char *text_out, int64_t text_cap,
                    char *explain_out, int64_t explain_cap);

extern "C" int planhost_tpch_run(void *dbp, int32_t query, int32_t repeat, int32_t warmup, double *ms_avg, double *ms_min, char *text_out, int64_t text_cap,
                                 char *explain_out, int64_t explain_cap) {
    return tpch_run(dbp, nullptr, query, repeat, warmup, ms_avg, ms_min, text_out, text_cap, explain_out, explain_cap);
}

extern "C" int planhost_tpch_run_comm(void *dbp, ph_comm *comm, int32_t query, int32_t repeat, int32_t warmup, double *ms_avg, double *ms_min, char *text_out,
                                      int64_t text_cap, char *explain_out, int64_t explain_cap) {
    return tpch_run(dbp, comm, query, repeat, warmup, ms_avg, ms_min, text_out, text_cap, explain_out, explain_cap);
}

static int tpch_run(void *dbp, ph_comm *comm, int32_t query, int32_t repeat, int32_t warmup, double *ms_avg, double *ms_min, char *text_out, int64_t text_cap,
                    char *explain_out, int64_t explain_cap) {
    auto *db = (plan::TpchDatabase *)dbp;
    if (!db || repeat < 1 || warmup < 0) { g_host_err = "planhost_tpch_run: bad arguments"; return PH_EINVAL; }
    plan::TpchQuery q;
    std::string e = plan::BuildTpchQuery(*db, query, &q);
    if (!e.empty()) { g_host_err = e; return PH_EUNSUPPORTED; }
    std::vector<std::string> lines;
    std::string explain;
    double total = 0, best = 1e300;
    for (int i = 0; i < warmup + repeat; i++) {
        const double t0 = plan::now_s();
        e = plan::RunTpchQuery(db->ctx, q, &lines, &explain, comm);   // builds the executors, pulls every chunk, closes them: one whole query
        const double dt = plan::now_s() - t0;
        if (!e.empty()) { g_host_err = e; return PH_EHIP; }
        if (i >= warmup) { total += dt; best = std::min(best, dt); }
    }
    if (ms_avg) *ms_avg = total / repeat * 1e3;
    if (ms_min) *ms_min = best * 1e3;
    if (text_out && text_cap > 0) {
        std::string text = "#";
        for (int i = 1; i < q.ncols; i++) text += "\t";
        text += "\n";
        for (auto &l : lines) { text += l; text += "\n"; }
        snprintf(text_out, (size_t)text_cap, "%s", text.c_str());
    }
    if (explain_out && explain_cap > 0) snprintf(explain_out, (size_t)explain_cap, "%s", explain.c_str());
    return PH_OK;
}

extern "C" void planhost_tpch_free(void *db) { delete (plan::TpchDatabase *)db; }

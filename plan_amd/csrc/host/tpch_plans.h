// TPC-H as resident tables + the operator subtrees of its queries as ResidentPlans — the stand-in for what the
// reference's planner (pkg/compute builder_*.go / optimizer_*.go, out of scope: SURVEY.md §2 rows 12-13) hands
// buildOperatorExec for `tester tpch1g --query_id N` (cmd/tester/main.go:65-73). The join ORDER of each query is
// fixed here as a planner would fix it; every physical choice below that is the library's (ph_plan, planhip.h).
// Used by host_tester (`q3|q9 <sf> resident`), by tests/test_host_layer.py and — through the C entry points at the
// bottom — by bench.py's q3_operator_interface / q9_operator_interface companions.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "operator_exec.h"

namespace plan {

// column positions in the resident tables (cases/tpch/query/ddl.sql names; the pruned columns the queries read)
enum { L_ORDERKEY, L_PARTKEY, L_SUPPKEY, L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_RETURNFLAG, L_LINESTATUS, L_SHIPDATE,
       L_COMMITDATE, L_RECEIPTDATE, L_SHIPMODE, L_SHIPINSTRUCT, L_LINENUMBER };
enum { O_ORDERKEY, O_CUSTKEY, O_ORDERDATE, O_SHIPPRIORITY, O_ORDERPRIORITY, O_TOTALPRICE, O_ORDERSTATUS, O_COMMENT };
enum { C_CUSTKEY, C_NATIONKEY, C_MKTSEGMENT, C_NAME, C_PHONE, C_ACCTBAL, C_ADDRESS, C_COMMENT };
enum { P_PARTKEY, P_NAME, P_BRAND, P_TYPE, P_SIZE, P_CONTAINER, P_MFGR };
enum { PS_PARTKEY, PS_SUPPKEY, PS_SUPPLYCOST, PS_AVAILQTY };
enum { S_SUPPKEY, S_NATIONKEY, S_NAME, S_ADDRESS, S_PHONE, S_ACCTBAL, S_COMMENT };
enum { N_NATIONKEY, N_NAME, N_REGIONKEY };
enum { R_REGIONKEY, R_NAME };

struct TpchDatabase {
    ph_ctx *ctx = nullptr;
    int64_t num = 1, den = 1;
    ResidentTable lineitem, orders, customer, part, partsupp, supplier, nation, region;
    double generate_s = 0, load_s = 0;
    int64_t loaded_bytes = 0;
    // generates (clean-room dbgen equivalent, include/tpchgen.h) and loads every table; declares the primary keys.
    // rank / nranks: this rank's SHARD of a database split over nranks ranks — orders and lineitem by order ranges (a line lives with its order),
    // customer / supplier / part (+ its partsupp rows) by row ranges, NATION and REGION whole on every rank (ph_table_set_replicated)
    std::string Load(ph_ctx *ctx, int64_t sf_num, int64_t sf_den, int rank = 0, int nranks = 1);
    ~TpchDatabase();
};

struct TpchQuery {
    int id = 0;
    ResidentPlan plan;
    std::vector<Compare> having;          // aggExecutor's output phase (executor_aggr.go:143-263)
    std::vector<ProjExpr> outputs;        //   ... its output expressions, in select-list order
    std::vector<OrderKey> order;          // orderExecutor above the aggregate
    int64_t limit = -1;                   // limitExecutor above that (-1 = none)
    int topkAgg = -1;                     // ORDER BY's first key is aggregate topkAgg (index into the plan's aggregates)
    bool topkDesc = false;
    int rowsTopkCol = -1;                 // a join-rooted plan: ORDER BY's first key is this output column (with `limit`): ph_plan_set_rows_topk
    bool rowsTopkDesc = false;
    int ncols = 0;                        // result columns (the headline's tab count)
    // an UNCORRELATED scalar subquery in HAVING (Q11: sum > (select sum(..) * 0.0001 ..)): the reference plans it as a cross product
    // with a one-row relation and a Filter above; here its plan runs first and its one value, cast and multiplied as the binder types
    // the select list (a FLOAT literal: decimal -> float64 -> float32, `*` in float32), becomes the literal of one more HAVING conjunct
    // `result column havingCol > value` — a DECIMAL column against a FLOAT: compared as float32
    std::shared_ptr<TpchQuery> scalar;
    float scalarFactor = 0;
    int havingCol = -1;
    // ... or the scalar is the right side of a DECIMAL > DECIMAL conjunct pushed into a scan of the main plan (Q22: c_acctbal > (select
    // avg(c_acctbal) ..)): conjunct scalarConjunct of node scalarScanNode gets floor(value) at the column's scale as its literal
    // (greatDecimalOp is exact: for a column with `scale` digits, x > v  <=>  unscaled(x) > floor(v * 10^scale))
    int scalarScanNode = -1, scalarConjunct = -1;
};

// the operator subtree of cases/tpch/query/q<id>.sql over the resident database
std::string BuildTpchQuery(const TpchDatabase &db, int id, TpchQuery *out);

// one execution through the operator interface, exactly as execOps pulls it (executor.go:151-188):
// limitExecutor <- gpuOrderExecutor <- gpuResidentPlanExecutor; result rows as text lines (Chunk.SaveToFile format)
// comm != nullptr: multi-rank execution — every rank calls this over its shard with its communicator and receives the complete result (ph_plan_set_comm)
std::string RunTpchQuery(ph_ctx *ctx, const TpchQuery &q, std::vector<std::string> *lines, std::string *explain, ph_comm *comm = nullptr);

}  // namespace plan

// ---- C entry points (libplantpch.so) for harnesses without C++ (bench.py): load once, run a query `repeat` times
// behind `warmup` untimed runs, report the per-query wall time the way Run prints it (executor_bench.go:126-137)
extern "C" {
int planhost_tpch_load(ph_ctx *ctx, int64_t sf_num, int64_t sf_den, void **db_out);
// this rank's shard of a database split over nranks ranks; planhost_tpch_run_comm runs a query over the shards with the ranks' communicator
int planhost_tpch_load_shard(ph_ctx *ctx, int64_t sf_num, int64_t sf_den, int32_t rank, int32_t nranks, void **db_out);
int planhost_tpch_run_comm(void *db, ph_comm *comm, int32_t query, int32_t repeat, int32_t warmup, double *ms_avg, double *ms_min, char *text_out, int64_t text_cap,
                           char *explain_out, int64_t explain_cap);
int64_t planhost_tpch_rows(void *db, const char *table);
// text_out: headline + rows of the LAST run; explain_out: the library's account of the forms it chose
int planhost_tpch_run(void *db, int32_t query, int32_t repeat, int32_t warmup, double *ms_avg, double *ms_min, char *text_out, int64_t text_cap,
                      char *explain_out, int64_t explain_cap);
const char *planhost_last_error(void);
void planhost_tpch_free(void *db);
}

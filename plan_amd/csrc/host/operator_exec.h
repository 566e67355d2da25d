// Host-side mirror of the reference's operator interface for the hot path, driving libplanhip.so.
//
//   OperatorExec{Init, Execute, Close}      pkg/compute/executor_operator.go:52-56
//   OperatorResult values                   executor_operator.go:11-18
//   pull protocol                           execOps, executor.go:151-188: the parent passes an
//       EMPTY output chunk; the callee shapes it (ensureOutputChunk, executor.go:201-210) and
//       returns haveMoreOutput with Card() in [0,2048], or Done, or InvalidOpResult + error.
//   stubExecutor                            executor_stub.go:38-63 (replays Chunk.Serialize output)
//   gpuFilterExecutor  <- filterExecutor    executor_filter.go:27-114
//   gpuAggExecutor     <- aggExecutor       executor_aggr.go:37-265
//   gpuJoinExecutor    <- joinExecutor      executor_join.go:27-264
//   gpuOrderExecutor   <- orderExecutor     executor_order.go:56-138 (LocalSort sort_local.go:64-250)
//   gpuProjectExecutor <- projectExecutor   executor_project.go:24-85 (ExprExec over the child chunk)
//   limitExecutor      <- limitExecutor     executor_limit.go:27-238 (Limit.Sink / GetData / HandleOffset)
//   crossProductExecutor <- CrossProduct    join_cross.go:34-230
//
// The Go shim of INTEGRATION.md has exactly this shape; this C++ form exists because the build
// environment has no Go toolchain, and it is what the host-level tests drive.
// Errors: the reference returns (OperatorResult, error); here Execute returns the result and
// fills *err (empty = nil). Nothing throws or aborts.
#pragma once

#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "chunk.h"
#include "planhip.h"

namespace plan {

enum OperatorResult { InvalidOpResult = 0, NeedMoreInput = 1, haveMoreOutput = 2, Done = 3 };

class OperatorExec {
public:
    virtual ~OperatorExec() {}
    virtual std::string Init() = 0;                                                  // "" = nil
    virtual OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) = 0;
    virtual std::string Close() = 0;
    virtual std::vector<LType> OutputTypes() const = 0;
};

// ---- the pieces of a bound plan these executors need (what PhysicalOperator.Filters / Info carry)
struct Literal {
    enum Kind { Int, Float, DateDays, Str, Dec } kind = Int;
    int64_t i = 0;
    double f = 0;
    std::string s;
    int scale = 0;
};
struct Compare {  // one conjunct: child column OP literal
    int col;
    int op;  // ph_cmp
    Literal k;
};
// a boolean expression over a node's columns — the tree ExprExec.executeSelect walks (expr_exec.go:342-530):
// comparisons of a column with a literal or with another column, AND (narrows child by child), OR (unites the
// children's true rows; `a IN (x, y)` binds to in(a,x) OR in(a,y))
struct BoolExpr {
    enum Kind { Cmp, And, Or } kind = Cmp;
    int col = -1, op = 0;        // Cmp: col OP k, or — col2 >= 0 — col OP column col2
    Literal k;
    int col2 = -1;
    std::vector<BoolExpr> children;
    static BoolExpr C(int col, int op, Literal k) { BoolExpr b; b.col = col; b.op = op; b.k = std::move(k); return b; }
    static BoolExpr CC(int col, int op, int col2) { BoolExpr b; b.col = col; b.op = op; b.col2 = col2; return b; }
    static BoolExpr AndOf(std::vector<BoolExpr> c) { BoolExpr b; b.kind = And; b.children = std::move(c); return b; }
    static BoolExpr OrOf(std::vector<BoolExpr> c) { BoolExpr b; b.kind = Or; b.children = std::move(c); return b; }
    static BoolExpr In(int col, const std::vector<Literal> &vals) { BoolExpr b; b.kind = Or; for (auto &v : vals) b.children.push_back(C(col, PH_EQ, v)); return b; }
    bool empty() const { return kind == Cmp && col < 0; }
};

// FLOAT (float32) arithmetic over finalised aggregate rows — Q14's `100.00 * a / b`: the literal is FLOAT, both operators
// bind to their FLOAT overloads, DECIMAL operands are cast decimal -> float64 -> float32 (function_scalar.go:476-512,
// 960-1010; function_cast.go:349-354). RPN; evaluated on the host (the rows are group rows).
struct FloatOp {
    enum Op { Col, Const, Add, Sub, Mul, Div, Lt, Le, Gt, Ge } op = Const;   // (the comparisons: resident plans only — PH_PE_FLOAT)
    int col = -1;
    float k = 0;
};

struct AggExpr;

// one output expression of a Project / of the aggregate's output phase: a column reference (zero
// copy, like executeColumnRef), a decimal expression (RPN over child columns, evaluated on the
// device: executeFunc over the binary decimal operators), extract(year ...) or substring
struct ProjExpr {
    enum Kind { ColRef, Decimal, ExtractYear, Substring, Case, Float32, DecimalQuo } kind = ColRef;
    int col = -1;                 // ColRef / ExtractYear / Substring: child column; DecimalQuo: the dividend
    int col2 = -1;                // DecimalQuo: the divisor
    std::vector<ph_rpn> prog;     // Decimal: RPN over child columns; Case: the THEN branch
    int64_t offset = 1, length = 0;   // Substring(col FROM offset FOR length)
    // Case (resident plans only): CASE WHEN when THEN prog ELSE elseProg END (executeCase, expr_exec.go:144-246)
    std::shared_ptr<BoolExpr> when;
    std::vector<ph_rpn> elseProg;
    bool resultInt = false;       // both branches are INTEGER literals: the result is INTEGER
    std::vector<FloatOp> fprog;   // Float32
    // Float32 inside a resident plan (PH_PE_FLOAT, evaluated on the device): floatWide = DOUBLE arithmetic (a FLOAT literal beside a DOUBLE
    // operand is widened: Q17's 0.2 * avg(INTEGER)); floatTruth = the program ends in a comparison and the value is its INTEGER truth, which a
    // Filter above tests with `= 1`
    bool floatWide = false, floatTruth = false;
    static ProjExpr FloatTruth(std::vector<FloatOp> p, bool wide) { ProjExpr e; e.kind = Float32; e.fprog = std::move(p); e.floatTruth = true; e.floatWide = wide; return e; }
    static ProjExpr CaseOf(BoolExpr w, std::vector<ph_rpn> thenProg, std::vector<ph_rpn> elseProg, bool resultInt = false) {
        ProjExpr e; e.kind = Case; e.when = std::make_shared<BoolExpr>(std::move(w)); e.prog = std::move(thenProg); e.elseProg = std::move(elseProg); e.resultInt = resultInt; return e;
    }
    static ProjExpr Float(std::vector<FloatOp> p) { ProjExpr e; e.kind = Float32; e.fprog = std::move(p); return e; }
    // DECIMAL `/` over two DECIMAL columns, typed as the dividend (BindDecimalDivide, function_scalar.go:507-514); evaluated on the
    // host: in the queries that have it (Q8's market share) it runs over the aggregate's few result rows
    static ProjExpr DecQuo(int a, int b) { ProjExpr e; e.kind = DecimalQuo; e.col = a; e.col2 = b; return e; }
    static ProjExpr Col(int c) { ProjExpr e; e.kind = ColRef; e.col = c; return e; }
    static ProjExpr Dec(std::vector<ph_rpn> p) { ProjExpr e; e.kind = Decimal; e.prog = std::move(p); return e; }
    static ProjExpr Year(int c) { ProjExpr e; e.kind = ExtractYear; e.col = c; return e; }
    static ProjExpr Substr(int c, int64_t off, int64_t len) { ProjExpr e; e.kind = Substring; e.col = c; e.offset = off; e.length = len; return e; }
};

struct AggExpr {
    int kind;                  // ph_aggkind
    std::vector<ph_rpn> prog;  // argument over child columns (empty for count(*))
    std::shared_ptr<ProjExpr> expr;   // ... or any expression (a CASE: resident plans only); prog is ignored then
    AggExpr(int k = 0, std::vector<ph_rpn> p = {}) : kind(k), prog(std::move(p)) {}
    static AggExpr Of(int k, ProjExpr e) { AggExpr a(k); a.expr = std::make_shared<ProjExpr>(std::move(e)); return a; }
};

// replays serialized chunks (the reference's fixture mechanism)
class stubExecutor : public OperatorExec {
public:
    stubExecutor(std::vector<LType> types, std::string blob) : types_(std::move(types)), blob_(std::move(blob)) {}
    std::string Init() override { pos_ = 0; return ""; }
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override { return ""; }
    std::vector<LType> OutputTypes() const override { return types_; }
private:
    std::vector<LType> types_;
    std::string blob_;
    size_t pos_ = 0;
};

// produces chunks from a callback (stands in for scanExecutor over DataTable.Scan)
class sourceExecutor : public OperatorExec {
public:
    using Fn = std::function<bool(Chunk *out)>;  // fills <= 2048 rows, false at end
    sourceExecutor(std::vector<LType> types, Fn fn) : types_(std::move(types)), fn_(std::move(fn)) {}
    std::string Init() override { return ""; }
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override { return ""; }
    std::vector<LType> OutputTypes() const override { return types_; }
private:
    std::vector<LType> types_;
    Fn fn_;
};

// Stages columns of many 2048-row chunks into one device batch in the narrow encodings.
class DeviceBatch {
public:
    // asString[k]: stage VARCHAR column k as offsets + bytes (PH_STR: LIKE / substring operands)
    // instead of a <= 256-entry dictionary code (group keys, '=' on low-cardinality columns)
    DeviceBatch(ph_ctx *ctx, std::vector<LType> types, std::vector<int> cols, std::vector<bool> asString = {});
    ~DeviceBatch();
    std::string Append(const Chunk &c);      // copies rows of the selected columns (host staging)
    std::string Upload();                    // -> device columns
    void Reset();
    int64_t rows() const { return rows_; }
    ph_col col(int k) const { return dev_[(size_t)k]; }  // k-th selected column, device view
    const std::vector<std::string> &dict(int k) const { return dicts_[(size_t)k]; }
    int code_of(int k, const std::string &s) const;      // -1 when not in the dictionary
    // value range of the non-NULL INTEGER / BIGINT values staged in column k (column statistics for
    // ph_join_build_ex); false when the column has another type or no value yet
    bool key_range(int k, int64_t *lo, int64_t *hi) const {
        if ((size_t)k >= ranged_.size() || !ranged_[(size_t)k]) return false;
        *lo = lo_[(size_t)k]; *hi = hi_[(size_t)k];
        return true;
    }
private:
    void note_range(int k, int64_t v) {
        if (ranged_.size() <= (size_t)k) { ranged_.resize((size_t)k + 1, false); lo_.resize((size_t)k + 1, 0); hi_.resize((size_t)k + 1, 0); }
        if (!ranged_[(size_t)k]) { ranged_[(size_t)k] = true; lo_[(size_t)k] = hi_[(size_t)k] = v; }
        else { if (v < lo_[(size_t)k]) lo_[(size_t)k] = v; if (v > hi_[(size_t)k]) hi_[(size_t)k] = v; }
    }
    std::vector<bool> ranged_;
    std::vector<int64_t> lo_, hi_;
    ph_ctx *ctx_;
    std::vector<LType> types_;
    std::vector<int> cols_;
    std::vector<std::vector<uint8_t>> host_, valid_, bytes_;
    std::vector<bool> has_null_, as_string_;
    std::vector<std::vector<std::string>> dicts_;
    std::vector<std::map<std::string, int>> dict_index_;
    std::vector<ph_col> dev_;
    std::vector<void *> dev_data_, dev_valid_, dev_aux_;
    int64_t rows_ = 0;
};

class gpuFilterExecutor : public OperatorExec {
public:
    gpuFilterExecutor(ph_ctx *ctx, std::vector<Compare> conjuncts, OperatorExec *child, int batchChunks = 512);
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return child_->OutputTypes(); }
private:
    std::string fill();
    ph_ctx *ctx_;
    std::vector<Compare> conj_;
    OperatorExec *child_;
    int batchChunks_;
    std::vector<int> cols_;  // distinct filter columns
    std::unique_ptr<DeviceBatch> batch_;
    std::deque<std::pair<std::shared_ptr<Chunk>, std::shared_ptr<SelectVector>>> ready_;
    bool childDone_ = false;
};

class gpuAggExecutor : public OperatorExec {
public:
    gpuAggExecutor(ph_ctx *ctx, std::vector<int> groupCols, std::vector<AggExpr> aggs, OperatorExec *child,
                   int64_t batchRows = 1 << 20);
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return finalTypes_.empty() ? outTypes_ : finalTypes_; }
    // the output phase of aggExecutor.Execute (executor_aggr.go:143-263): HAVING conjuncts over the
    // [group columns | aggregate results] rows (filterExec.executeSelect), then the output
    // expressions over the surviving rows (outputExec.executeExprs). Both optional; call before Init.
    void SetHaving(std::vector<Compare> conjuncts) { having_ = std::move(conjuncts); }
    void SetOutputs(std::vector<ProjExpr> outputs) { outputs_ = std::move(outputs); }
private:
    std::string sinkBatch();
    std::string finalize();
    std::vector<Compare> having_;
    std::vector<ProjExpr> outputs_;
    std::vector<LType> finalTypes_;
    ph_ctx *ctx_;
    std::vector<int> groupCols_;
    std::vector<AggExpr> aggs_;
    OperatorExec *child_;
    int64_t batchRows_;
    std::vector<LType> childTypes_, outTypes_;
    std::vector<int> stagedCols_;          // child columns staged to the device
    std::vector<int> argScale_;            // scale of each aggregate argument
    std::vector<LType> argType_;
    std::unique_ptr<DeviceBatch> batch_;
    ph_agg *agg_ = nullptr;
    int64_t rowBase_ = 0;
    bool built_ = false;
    std::vector<std::shared_ptr<Chunk>> results_;
    size_t next_ = 0;
};

// join types of the hot path (LOT_JoinType*, join_scan.go:47-165); SEMI / ANTI emit the probe
// rows that have / lack a match (NextSemiOrAntiJoin :102-120), MARK is their building block;
// LEFT is the inner result followed by the unmatched probe rows with an all-NULL (PF_CONST)
// build side (NextLeftJoin :67-88)
enum JoinType { JoinInner, JoinSemi, JoinAnti, JoinLeft };

class gpuJoinExecutor : public OperatorExec {
public:
    // children[0] probes, children[1] is built (executor_join.go:237-264); inner: output = all
    // probe columns followed by buildPayload columns of the build side (also LEFT); semi/anti:
    // probe columns.
    gpuJoinExecutor(ph_ctx *ctx, OperatorExec *probe, OperatorExec *build, std::vector<int> probeKeys,
                    std::vector<int> buildKeys, std::vector<int> buildPayload, int batchChunks = 512,
                    JoinType type = JoinInner);
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return outTypes_; }
private:
    std::string buildTable();
    std::string probeBatch();
    ph_ctx *ctx_;
    OperatorExec *probe_, *build_;
    std::vector<int> probeKeys_, buildKeys_, buildPayload_;
    int batchChunks_;
    std::vector<LType> outTypes_;
    std::unique_ptr<DeviceBatch> buildBatch_, probeBatch_;
    std::vector<std::shared_ptr<Chunk>> buildChunks_;   // kept for payload gather
    std::vector<int64_t> buildStart_;                   // first row id of each build chunk
    ph_join *join_ = nullptr;
    bool built_ = false, probeDone_ = false;
    JoinType type_ = JoinInner;
    std::deque<std::shared_ptr<Chunk>> ready_;
};

// ORDER BY: drains the child (orderExecutor.Execute sinks every chunk into LocalSort, then scans
// the sorted rows, executor_order.go:56-138), sorts the row ids on the device by the ORDER BY
// columns (ph_sort_rows: the reference's key encoding, NULLs first) and emits the child's rows in
// that order, 2048 per chunk. VARCHAR keys must be dictionary columns whose codes are assigned in
// ascending byte order of the strings; the executor re-codes them so before uploading.
struct OrderKey {
    int col;          // child column
    bool descending;
};

class gpuOrderExecutor : public OperatorExec {
public:
    gpuOrderExecutor(ph_ctx *ctx, std::vector<OrderKey> keys, OperatorExec *child);
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return child_->OutputTypes(); }
private:
    std::string sortAll();
    ph_ctx *ctx_;
    std::vector<OrderKey> keys_;
    OperatorExec *child_;
    std::vector<std::shared_ptr<Chunk>> chunks_;
    std::vector<int64_t> start_;          // first row id of each chunk
    std::vector<int32_t> order_;          // sorted row ids
    std::vector<std::vector<Vector::Unified>> unified_;   // per input chunk and column (built with the first output chunk)
    size_t next_ = 0;
    bool sorted_ = false;
};

// Project: evaluates its expressions for every child chunk. Column references are zero-copy; decimal
// expressions, extract and substring run on the device over batches of child chunks.
class gpuProjectExecutor : public OperatorExec {
public:
    gpuProjectExecutor(ph_ctx *ctx, std::vector<ProjExpr> exprs, OperatorExec *child, int batchChunks = 512);
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return outTypes_; }
    // evaluates the expressions over already materialised chunks (used by the aggregate's output phase)
    static std::string Evaluate(ph_ctx *ctx, const std::vector<ProjExpr> &exprs, const std::vector<LType> &childTypes,
                                const std::vector<LType> &outTypes, const std::vector<std::shared_ptr<Chunk>> &in,
                                std::vector<std::shared_ptr<Chunk>> *out);
    static std::string Types(const std::vector<ProjExpr> &exprs, const std::vector<LType> &childTypes, std::vector<LType> *out);
private:
    ph_ctx *ctx_;
    std::vector<ProjExpr> exprs_;
    OperatorExec *child_;
    int batchChunks_;
    std::vector<LType> outTypes_;
    std::deque<std::shared_ptr<Chunk>> ready_;
    bool childDone_ = false;
};

// LIMIT / OFFSET (Limit.Sink + GetData + HandleOffset): passes child rows [offset, offset+limit)
class limitExecutor : public OperatorExec {
public:
    limitExecutor(uint64_t limit, uint64_t offset, OperatorExec *child) : limit_(limit), offset_(offset), child_(child) {}
    std::string Init() override { seen_ = 0; return ""; }
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override { return ""; }
    std::vector<LType> OutputTypes() const override { return child_->OutputTypes(); }
private:
    uint64_t limit_, offset_, seen_ = 0;
    OperatorExec *child_;
};

// Cross product (join_cross.go:34-230): the right child is collected first (Sink); then for every
// left chunk and every right ROW one output chunk: the left columns referenced as they are, the
// right row's values as constant vectors (ReferenceInPhyFormatConst) — no row is copied.
class crossProductExecutor : public OperatorExec {
public:
    crossProductExecutor(OperatorExec *left, OperatorExec *right) : left_(left), right_(right) {}
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override { rhs_.clear(); cur_.reset(); return ""; }
    std::vector<LType> OutputTypes() const override { return outTypes_; }
private:
    OperatorExec *left_, *right_;
    std::vector<LType> outTypes_;
    std::vector<std::shared_ptr<Chunk>> rhs_;
    std::shared_ptr<Chunk> cur_;       // current left chunk
    size_t rchunk_ = 0;
    int rrow_ = 0;
    bool collected_ = false, leftDone_ = false;
};

// Agg <- Scan(filter) over a RESIDENT table — the measured mode behind the operator interface
// (INTEGRATION.md's gpuScanAggExecutor): nothing is staged per chunk; Init builds the
// ph_scan_plan (fused kernel when the shape matches, operator chain otherwise), the first
// Execute runs it over the whole table, later calls hand out <= 2048 group rows each.
struct ResidentColumn {
    LType type;                      // SQL type of the column (what the scan would emit)
    std::vector<std::string> dict;   // VARCHAR dictionary columns: code -> string
};

class gpuScanAggExecutor : public OperatorExec {
public:
    gpuScanAggExecutor(ph_ctx *ctx, const ph_table *table, std::vector<ResidentColumn> columns,
                       std::vector<Compare> conjuncts, std::vector<int> groupCols, std::vector<AggExpr> aggs);
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return finalTypes_.empty() ? outTypes_ : finalTypes_; }
    const char *kind() const { return plan_ ? ph_scan_plan_kind(plan_) : ""; }
    void SetHaving(std::vector<Compare> conjuncts) { having_ = std::move(conjuncts); }
    void SetOutputs(std::vector<ProjExpr> outputs) { outputs_ = std::move(outputs); }
private:
    std::vector<Compare> having_;
    std::vector<ProjExpr> outputs_;
    std::vector<LType> finalTypes_;
    ph_ctx *ctx_;
    const ph_table *table_;
    std::vector<ResidentColumn> cols_;
    std::vector<Compare> conj_;
    std::vector<int> groupCols_;
    std::vector<AggExpr> aggs_;
    std::vector<LType> outTypes_, argType_;
    ph_scan_plan *plan_ = nullptr;
    std::vector<std::shared_ptr<Chunk>> results_;
    size_t next_ = 0;
    bool built_ = false;
};

// ---- resident plans: a whole operator SUBTREE over resident tables behind ONE OperatorExec
//
// The join analogue of gpuScanAggExecutor. buildOperatorExec (executor.go:305-350) builds its executors bottom-up;
// when the subtree under an Agg consists of Join / Filter / Project nodes whose leaves are scans of RESIDENT tables,
// the shim does not instantiate joinExecutor / filterExecutor / projectExecutor / aggExecutor objects at all: it
// writes the subtree down as a ResidentPlan (the PhysicalOperator fields each node carries: Filters, JoinOpInfo's
// OnConds as key column pairs, Outputs, AggOpInfo's GroupBys / Aggs) and gpuResidentPlanExecutor hands it to the
// library as a ph_plan (include/planhip.h). Every physical choice — table forms, lookups vs pairs, fused filters,
// merge / streaming forms — is the library's, from the tables' statistics; nothing here or in the planner hints it.
struct ResidentTable {
    const ph_table *table = nullptr;
    std::vector<ResidentColumn> cols;   // SQL type (and dictionary) of every column of the resident table
};

class ResidentPlan {
public:
    // every method returns the new node's index; children must exist already (bottom-up, like buildOperatorExec)
    // conjuncts: simple `column OP literal` ones (a build or a probe can absorb those); where: one more conjunct of any shape
    int Scan(const ResidentTable *t, std::vector<int> cols, std::vector<Compare> conjuncts = {}, BoolExpr where = BoolExpr());
    int Filter(int child, std::vector<Compare> conjuncts, BoolExpr where = BoolExpr());
    // output = the listed columns of [probe child's columns | build child's columns] (SEMI / ANTI: probe columns)
    // residual: the join's non-equi condition over [probe columns | build columns] (HashJoin's conditions beside the keys); INNER keeps the
    // pairs that satisfy it, SEMI / ANTI the probe rows with / without such a pair
    int Join(int probe, int build, std::vector<int> probeKeys, std::vector<int> buildKeys, std::vector<int> out, JoinType type = JoinInner,
             BoolExpr residual = BoolExpr());
    int Project(int child, std::vector<ProjExpr> exprs);
    // the root — or an aggregate BELOW other operators (a subquery's GROUP BY; its HAVING is a Filter above it): its output
    // columns are [group columns | aggregate results] with FinalizeStates' types (SUM(INTEGER) / COUNT are HUGEINT)
    int Agg(int child, std::vector<ProjExpr> groups, std::vector<AggExpr> aggs);
    struct Node {
        int kind = 0, child[2] = {-1, -1};
        const ResidentTable *table = nullptr;
        std::vector<int> cols, probeKeys, buildKeys, out;
        std::vector<Compare> conjuncts;
        BoolExpr where;
        JoinType joinType = JoinInner;
        std::vector<ProjExpr> exprs;    // Project; Agg: the group-by expressions
        std::vector<AggExpr> aggs;
        int ngroups = 0;                // Agg: group-by expressions; its output = [group columns | aggregate results]
        std::vector<LType> argTypes;    // Agg: type of every aggregate's argument
        std::vector<LType> types;       // output types of the node
        std::vector<const ResidentColumn *> source;   // per output column: the resident column it is an unchanged copy of (or null)
    };
    std::vector<Node> nodes;
    std::string error;                  // first construction error ("" = none); checked by the executor's Init
};

class gpuResidentPlanExecutor : public OperatorExec {
public:
    gpuResidentPlanExecutor(ph_ctx *ctx, ResidentPlan plan) : ctx_(ctx), rp_(std::move(plan)) {}
    std::string Init() override;
    OperatorResult Execute(Chunk *input, Chunk *output, std::string *err) override;
    std::string Close() override;
    std::vector<LType> OutputTypes() const override { return finalTypes_.empty() ? outTypes_ : finalTypes_; }
    void SetHaving(std::vector<Compare> conjuncts) { having_ = std::move(conjuncts); }
    void SetOutputs(std::vector<ProjExpr> outputs) { outputs_ = std::move(outputs); }
    // Order <- Limit above the aggregate with an aggregate as the first ORDER BY key (what the shim sees when it
    // walks up from the Agg): only the groups that can reach the first k rows come back from the device
    void SetTopK(int aggIndex, bool descending, int64_t k) { topkAgg_ = aggIndex; topkDesc_ = descending; topkK_ = k; }
    // a join-rooted plan below Order(first key = output column `col`) + Limit(k): ph_plan_set_rows_topk
    void SetRowsTopK(int col, bool descending, int64_t k) { rowsTopkCol_ = col; rowsTopkDesc_ = descending; rowsTopkK_ = k; }
    // multi-rank execution: every rank builds this executor over its shard of the resident tables and announces the ranks' communicator; the
    // library inserts the exchanges (ph_plan_set_comm) and every rank's executor emits the complete result
    void SetComm(ph_comm *comm) { comm_ = comm; }
    std::string Explain() const { return plan_ ? ph_plan_explain(plan_) : ""; }
private:
    ph_ctx *ctx_;
    ResidentPlan rp_;
    std::vector<Compare> having_;
    bool havingOnDevice_ = false;       // the plan applies the conjuncts where the groups are (ph_plan_set_having)
    bool rowsRoot_ = false;             // the plan's root is a join / filter / project: its ROWS are the output (ph_plan_fetch_rows)
    std::vector<ProjExpr> outputs_;
    std::vector<LType> outTypes_, finalTypes_, argType_;
    int topkAgg_ = -1;
    int rowsTopkCol_ = -1;
    bool rowsTopkDesc_ = false;
    int64_t rowsTopkK_ = 0;
    bool topkDesc_ = false;
    int64_t topkK_ = 0;
    ph_plan *plan_ = nullptr;
    ph_comm *comm_ = nullptr;
    std::vector<std::shared_ptr<Chunk>> results_;
    size_t next_ = 0;
    bool built_ = false;
};

// builds the output chunks of an aggregate from the device result arrays (FinalizeStates typing)
std::string BuildAggOutput(const std::vector<LType> &outTypes, const std::vector<LType> &keyTypes,
                           const std::vector<const std::vector<std::string> *> &keyDicts,
                           const std::vector<int> &aggKinds, const std::vector<LType> &argTypes,
                           const std::vector<int> &argScales, int64_t ngroups, const int64_t *keys,
                           const uint8_t *keyNull, const uint64_t *lo, const int64_t *hi, const uint64_t *cnt,
                           std::vector<std::shared_ptr<Chunk>> *out);

// HAVING + output expressions over finalised aggregate rows (executor_aggr.go:143-263)
std::string ApplyAggOutputPhase(ph_ctx *ctx, const std::vector<Compare> &having, const std::vector<ProjExpr> &outputs,
                                const std::vector<LType> &rowTypes, const std::vector<LType> &finalTypes,
                                std::vector<std::shared_ptr<Chunk>> *chunks);

// copies one cell (any supported type) between flat vectors; src may be any format
void CopyCell(const Vector &src, int srcRow, Vector *dst, int dstRow);

}  // namespace plan

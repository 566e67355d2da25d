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
struct AggExpr {
    int kind;                  // ph_aggkind
    std::vector<ph_rpn> prog;  // argument over child columns (empty for count(*))
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
    DeviceBatch(ph_ctx *ctx, std::vector<LType> types, std::vector<int> cols);
    ~DeviceBatch();
    std::string Append(const Chunk &c);      // copies rows of the selected columns (host staging)
    std::string Upload();                    // -> device columns
    void Reset();
    int64_t rows() const { return rows_; }
    ph_col col(int k) const { return dev_[(size_t)k]; }  // k-th selected column, device view
    const std::vector<std::string> &dict(int k) const { return dicts_[(size_t)k]; }
    int code_of(int k, const std::string &s) const;      // -1 when not in the dictionary
private:
    ph_ctx *ctx_;
    std::vector<LType> types_;
    std::vector<int> cols_;
    std::vector<std::vector<uint8_t>> host_, valid_;
    std::vector<bool> has_null_;
    std::vector<std::vector<std::string>> dicts_;
    std::vector<std::map<std::string, int>> dict_index_;
    std::vector<ph_col> dev_;
    std::vector<void *> dev_data_, dev_valid_;
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
    std::vector<LType> OutputTypes() const override { return outTypes_; }
private:
    std::string sinkBatch();
    std::string finalize();
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
    size_t next_ = 0;
    bool sorted_ = false;
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
    std::vector<LType> OutputTypes() const override { return outTypes_; }
    const char *kind() const { return plan_ ? ph_scan_plan_kind(plan_) : ""; }
private:
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

// builds the output chunks of an aggregate from the device result arrays (FinalizeStates typing)
std::string BuildAggOutput(const std::vector<LType> &outTypes, const std::vector<LType> &keyTypes,
                           const std::vector<const std::vector<std::string> *> &keyDicts,
                           const std::vector<int> &aggKinds, const std::vector<LType> &argTypes,
                           const std::vector<int> &argScales, int64_t ngroups, const int64_t *keys,
                           const uint8_t *keyNull, const uint64_t *lo, const int64_t *hi, const uint64_t *cnt,
                           std::vector<std::shared_ptr<Chunk>> *out);

// copies one cell (any supported type) between flat vectors; src may be any format
void CopyCell(const Vector &src, int srcRow, Vector *dst, int dstRow);

}  // namespace plan

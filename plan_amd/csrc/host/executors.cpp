// See operator_exec.h. GPU executors behind the reference's OperatorExec interface (compat mode:
// 2048-row chunks in and out, device work batched across many chunks).
#include "operator_exec.h"

#include <algorithm>
#include <chrono>
#include <string_view>
#include <unordered_map>
#include <cstdio>

namespace plan {

static std::string herr(const char *what) { return std::string(what) + ": " + ph_last_error(); }

static void ensureOutputChunk(const std::vector<LType> &types, Chunk *out) {  // executor.go:201-210
    if (out->ColumnCount() == 0) out->Init(types, DefaultVectorSize);
}

// ------------------------------------------------------------------ stub / source

OperatorResult stubExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (pos_ >= blob_.size()) return Done;
    if (!output->Deserialize(blob_, &pos_, err)) return InvalidOpResult;
    if (output->Card() == 0) return Done;
    return haveMoreOutput;
}

OperatorResult sourceExecutor::Execute(Chunk *, Chunk *output, std::string *) {
    ensureOutputChunk(types_, output);
    if (!fn_(output)) return Done;
    return haveMoreOutput;
}

void CopyCell(const Vector &src, int srcRow, Vector *dst, int dstRow) {
    Vector::Unified u;
    src.ToUnifiedFormat(srcRow + 1, &u);
    int64_t idx = u.sel->GetIndex(srcRow);
    if (!u.mask->RowIsValid((uint64_t)idx)) {
        dst->Mask.SetInvalid((uint64_t)dstRow, DefaultVectorSize);
        return;
    }
    size_t w = src._Typ.Size();
    if (src._Typ.GetInternalType() == PT_VARCHAR) {
        const String &s = reinterpret_cast<const String *>(u.data)[idx];
        dst->SetString(dstRow, s.Data, s.Len);
    } else {
        memcpy(dst->Data.data() + (size_t)dstRow * w, u.data + (size_t)idx * w, w);
    }
}

// ------------------------------------------------------------------ DeviceBatch

static int staged_width(const LType &t) {
    switch (t.GetInternalType()) {
    case PT_INT32: case PT_DATE: return 4;
    case PT_INT64: case PT_DECIMAL: return 8;
    case PT_INT128: return 8;   // HUGEINT aggregate results (HAVING operands): exact when they fit int64
    case PT_VARCHAR: return 1;
    default: return 0;
    }
}

static int staged_phtype(const LType &t) {
    switch (t.GetInternalType()) {
    case PT_INT32: return PH_I32;
    case PT_INT64: return PH_I64;
    case PT_DATE: return PH_DATE;
    case PT_DECIMAL: return PH_DEC64;
    // INT128 has exactly one comparison in selectOperation, '>' (function_operator_boolean.go:435-436),
    // which is also the one DECIMAL has: a scale-0 decimal column selects the same rows
    case PT_INT128: return PH_DEC64;
    case PT_VARCHAR: return PH_CODE8;
    default: return 0;
    }
}

DeviceBatch::DeviceBatch(ph_ctx *ctx, std::vector<LType> types, std::vector<int> cols, std::vector<bool> asString)
    : ctx_(ctx), types_(std::move(types)), cols_(std::move(cols)), as_string_(std::move(asString)) {
    size_t n = cols_.size();
    as_string_.resize(n, false);
    host_.resize(n); valid_.resize(n); bytes_.resize(n); has_null_.assign(n, false); dicts_.resize(n); dict_index_.resize(n);
    dev_.resize(n); dev_data_.assign(n, nullptr); dev_valid_.assign(n, nullptr); dev_aux_.assign(n, nullptr);
}

DeviceBatch::~DeviceBatch() { Reset(); }

void DeviceBatch::Reset() {
    for (size_t k = 0; k < cols_.size(); k++) {
        if (dev_data_[k]) ph_dev_free(ctx_, dev_data_[k]);
        if (dev_valid_[k]) ph_dev_free(ctx_, dev_valid_[k]);
        if (dev_aux_[k]) ph_dev_free(ctx_, dev_aux_[k]);
        dev_data_[k] = dev_valid_[k] = dev_aux_[k] = nullptr;
        host_[k].clear();
        bytes_[k].clear();
        valid_[k].clear();
        has_null_[k] = false;
    }
    std::fill(ranged_.begin(), ranged_.end(), false);
    rows_ = 0;
}

int DeviceBatch::code_of(int k, const std::string &s) const {
    auto it = dict_index_[(size_t)k].find(s);
    return it == dict_index_[(size_t)k].end() ? -1 : it->second;
}

std::string DeviceBatch::Append(const Chunk &c) {
    int card = c.Card();
    for (size_t k = 0; k < cols_.size(); k++) {
        const Vector &v = *c.Data[(size_t)cols_[k]];
        const LType &t = types_[(size_t)cols_[k]];
        int w = staged_width(t);
        if (w == 0) return "column type " + std::to_string(t.Id) + " cannot be staged to the device";
        Vector::Unified u;
        v.ToUnifiedFormat(card, &u);
        if (as_string_[k] && t.GetInternalType() == PT_VARCHAR) {
            // offsets + bytes (PH_STR): int32 offsets[rows+1], the first one written with the first row
            if (host_[k].empty()) host_[k].assign(4, 0);
            valid_[k].resize((size_t)(rows_ + card + 7) / 8 + 1, 0);
            for (int i = 0; i < card; i++) {
                int64_t idx = u.sel->GetIndex(i), row = rows_ + i;
                bool ok = u.mask->RowIsValid((uint64_t)idx);
                if (ok) {
                    valid_[k][(size_t)row >> 3] |= (uint8_t)(1u << (row & 7));
                    const String &sv = reinterpret_cast<const String *>(u.data)[idx];
                    bytes_[k].insert(bytes_[k].end(), (const uint8_t *)sv.Data, (const uint8_t *)sv.Data + sv.Len);
                } else has_null_[k] = true;
                if (bytes_[k].size() > 0x7fffffffu) return "VARCHAR batch exceeds 2 GiB of bytes";
                int32_t end = (int32_t)bytes_[k].size();
                host_[k].insert(host_[k].end(), (const uint8_t *)&end, (const uint8_t *)&end + 4);
            }
            continue;
        }
        size_t base = host_[k].size();
        host_[k].resize(base + (size_t)card * (size_t)w);
        valid_[k].resize((size_t)(rows_ + card + 7) / 8 + 1, 0);
        for (int i = 0; i < card; i++) {
            int64_t idx = u.sel->GetIndex(i);
            int64_t row = rows_ + i;
            bool ok = u.mask->RowIsValid((uint64_t)idx);
            if (ok) valid_[k][(size_t)row >> 3] |= (uint8_t)(1u << (row & 7)); else has_null_[k] = true;
            uint8_t *dst = host_[k].data() + base + (size_t)i * (size_t)w;
            if (!ok) { memset(dst, 0, (size_t)w); continue; }
            switch (t.GetInternalType()) {
            case PT_INT32: { int32_t v; memcpy(&v, u.data + (size_t)idx * 4, 4); memcpy(dst, &v, 4); note_range((int)k, v); break; }
            case PT_INT64: { int64_t v; memcpy(&v, u.data + (size_t)idx * 8, 8); memcpy(dst, &v, 8); note_range((int)k, v); break; }
            case PT_INT128: {
                const Hugeint &hg = reinterpret_cast<const Hugeint *>(u.data)[idx];
                if (hg.Upper != ((int64_t)hg.Lower >> 63)) return "HUGEINT value does not fit int64 on the device";
                memcpy(dst, &hg.Lower, 8);
                break;
            }
            case PT_DATE: { int32_t d = DaysFromDate(reinterpret_cast<const Date *>(u.data)[idx]); memcpy(dst, &d, 4); break; }
            case PT_DECIMAL: {
                int64_t x;
                if (!DecimalToUnscaled(reinterpret_cast<const Decimal *>(u.data)[idx], t.Scale, &x))
                    return "decimal value does not fit DECIMAL(18," + std::to_string(t.Scale) + ") on the device";
                memcpy(dst, &x, 8);
                break;
            }
            case PT_VARCHAR: {
                const String &s = reinterpret_cast<const String *>(u.data)[idx];
                std::string key(s.Data, (size_t)s.Len);
                auto it = dict_index_[k].find(key);
                int code;
                if (it == dict_index_[k].end()) {
                    code = (int)dicts_[k].size();
                    if (code > 255) return "VARCHAR column has more than 256 distinct values: not a dictionary-code column";
                    dicts_[k].push_back(key);
                    dict_index_[k][key] = code;
                } else code = it->second;
                *dst = (uint8_t)code;
                break;
            }
            default: break;
            }
        }
    }
    rows_ += card;
    return "";
}

std::string DeviceBatch::Upload() {
    for (size_t k = 0; k < cols_.size(); k++) {
        const LType &t = types_[(size_t)cols_[k]];
        if (dev_data_[k]) { ph_dev_free(ctx_, dev_data_[k]); dev_data_[k] = nullptr; }
        if (dev_valid_[k]) { ph_dev_free(ctx_, dev_valid_[k]); dev_valid_[k] = nullptr; }
        if (ph_dev_alloc(ctx_, (int64_t)host_[k].size() + 64, &dev_data_[k]) != PH_OK) return herr("ph_dev_alloc");
        if (ph_dev_upload(ctx_, dev_data_[k], host_[k].data(), (int64_t)host_[k].size()) != PH_OK) return herr("ph_dev_upload");
        ph_col c{};
        c.type = staged_phtype(t);
        c.scale = t.GetInternalType() == PT_INT128 ? 0 : t.Scale;
        c.data = dev_data_[k];
        if (as_string_[k] && t.GetInternalType() == PT_VARCHAR) {
            c.type = PH_STR;
            if (dev_aux_[k]) { ph_dev_free(ctx_, dev_aux_[k]); dev_aux_[k] = nullptr; }
            if (ph_dev_alloc(ctx_, (int64_t)bytes_[k].size() + 64, &dev_aux_[k]) != PH_OK) return herr("ph_dev_alloc");
            if (!bytes_[k].empty() && ph_dev_upload(ctx_, dev_aux_[k], bytes_[k].data(), (int64_t)bytes_[k].size()) != PH_OK) return herr("ph_dev_upload");
            c.aux = dev_aux_[k];
            c.aux_bytes = (int64_t)bytes_[k].size();
        }
        if (has_null_[k]) {
            int64_t nb = (rows_ + 7) / 8;
            if (ph_dev_alloc(ctx_, nb + 64, &dev_valid_[k]) != PH_OK) return herr("ph_dev_alloc");
            if (ph_dev_upload(ctx_, dev_valid_[k], valid_[k].data(), nb) != PH_OK) return herr("ph_dev_upload");
            c.validity = (const uint8_t *)dev_valid_[k];
        }
        dev_[k] = c;
    }
    return "";
}

// ------------------------------------------------------------------ filter

gpuFilterExecutor::gpuFilterExecutor(ph_ctx *ctx, std::vector<Compare> conjuncts, OperatorExec *child, int batchChunks)
    : ctx_(ctx), conj_(std::move(conjuncts)), child_(child), batchChunks_(batchChunks) {}

std::string gpuFilterExecutor::Init() {
    for (auto &c : conj_)
        if (std::find(cols_.begin(), cols_.end(), c.col) == cols_.end()) cols_.push_back(c.col);
    // a VARCHAR column under LIKE / NOT LIKE is staged as offsets + bytes, otherwise as dictionary codes
    std::vector<bool> asString(cols_.size(), false);
    for (auto &c : conj_)
        if (c.op == PH_LIKE || c.op == PH_NOTLIKE) asString[(size_t)(std::find(cols_.begin(), cols_.end(), c.col) - cols_.begin())] = true;
    batch_.reset(new DeviceBatch(ctx_, child_->OutputTypes(), cols_, asString));
    return "";
}

std::string gpuFilterExecutor::Close() { batch_.reset(); ready_.clear(); return ""; }

std::string gpuFilterExecutor::fill() {
    std::vector<std::shared_ptr<Chunk>> chunks;
    batch_->Reset();
    while ((int)chunks.size() < batchChunks_) {
        auto c = std::make_shared<Chunk>();
        std::string err;
        OperatorResult r = child_->Execute(nullptr, c.get(), &err);
        if (r == InvalidOpResult) return err.empty() ? "child failed" : err;
        if (r == Done) { childDone_ = true; break; }
        if (c->Card() == 0) continue;
        std::string e = batch_->Append(*c);
        if (!e.empty()) return e;
        chunks.push_back(c);
    }
    int64_t n = batch_->rows();
    if (n == 0) return "";
    std::string e = batch_->Upload();
    if (!e.empty()) return e;
    // execSelectAnd (expr_exec.go:444-486): each conjunct narrows the previous selection
    void *selA = nullptr, *selB = nullptr;
    if (ph_dev_alloc(ctx_, n * 4, &selA) != PH_OK || ph_dev_alloc(ctx_, n * 4, &selB) != PH_OK) return herr("ph_dev_alloc");
    const int32_t *cur = nullptr;
    int64_t cnt = n;
    for (size_t ci = 0; ci < conj_.size() && cnt > 0; ci++) {
        const Compare &cmp = conj_[ci];
        int k = (int)(std::find(cols_.begin(), cols_.end(), cmp.col) - cols_.begin());
        ph_col col = batch_->col(k);
        ph_const kc{};
        switch (cmp.k.kind) {
        case Literal::Int:   // against a HUGEINT / DECIMAL column the literal is cast to the column's type
            if (col.type == PH_DEC64) { kc.type = PH_DEC64; kc.i = cmp.k.i; kc.scale = 0; }
            else { kc.type = PH_I32; kc.i = cmp.k.i; }
            break;
        case Literal::Float: kc.type = PH_F32; kc.f = cmp.k.f; break;
        case Literal::DateDays: kc.type = PH_DATE; kc.i = cmp.k.i; break;
        case Literal::Dec: kc.type = PH_DEC64; kc.i = cmp.k.i; kc.scale = cmp.k.scale; break;
        case Literal::Str:
            if (col.type == PH_STR) { kc.type = PH_STR; kc.s = cmp.k.s.c_str(); }   // LIKE pattern / '=' operand
            else {  // VARCHAR =/!= on a dictionary column: literal -> code
                kc.type = PH_I32;
                int code = batch_->code_of(k, cmp.k.s);
                kc.i = code < 0 ? 999 : code;
            }
            break;
        }
        int32_t *out = (int32_t *)((ci & 1) ? selB : selA);
        int64_t m = 0;
        if (ph_filter_select(ctx_, &col, n, cmp.op, &kc, cur, cnt, out, &m) != PH_OK) {
            ph_dev_free(ctx_, selA); ph_dev_free(ctx_, selB);
            return herr("ph_filter_select");
        }
        cur = out;
        cnt = m;
    }
    std::vector<int32_t> sel((size_t)cnt);
    bool all = conj_.empty();
    if (!all && cnt > 0 && ph_dev_download(ctx_, sel.data(), cur, cnt * 4) != PH_OK) {
        ph_dev_free(ctx_, selA); ph_dev_free(ctx_, selB);
        return herr("ph_dev_download");
    }
    ph_dev_free(ctx_, selA);
    ph_dev_free(ctx_, selB);
    // split the batch selection back into per-chunk selection vectors
    size_t p = 0;
    int64_t start = 0;
    for (auto &c : chunks) {
        auto sv = std::make_shared<SelectVector>();
        if (all) { sv->identity = true; ready_.push_back({c, sv}); start += c->Card(); continue; }
        sv->identity = false;
        while (p < sel.size() && sel[p] < start + c->Card()) sv->SelVec.push_back(sel[p++] - start);
        start += c->Card();
        if (!sv->SelVec.empty()) ready_.push_back({c, sv});
    }
    return "";
}

OperatorResult gpuFilterExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    while (ready_.empty() && !childDone_) {
        std::string e = fill();
        if (!e.empty()) { *err = e; return InvalidOpResult; }
    }
    if (ready_.empty()) return Done;
    auto item = ready_.front();
    ready_.pop_front();
    ensureOutputChunk(OutputTypes(), output);
    std::vector<int> indice;
    for (int i = 0; i < item.first->ColumnCount(); i++) indice.push_back(i);
    int count = item.second->identity ? item.first->Card() : (int)item.second->SelVec.size();
    output->SliceIndice(*item.first, item.second, count, 0, indice);  // DICT views, no copy (executor_filter.go)
    return haveMoreOutput;
}

// ------------------------------------------------------------------ aggregate

gpuAggExecutor::gpuAggExecutor(ph_ctx *ctx, std::vector<int> groupCols, std::vector<AggExpr> aggs, OperatorExec *child,
                               int64_t batchRows)
    : ctx_(ctx), groupCols_(std::move(groupCols)), aggs_(std::move(aggs)), child_(child), batchRows_(batchRows) {}

std::string gpuAggExecutor::Init() {
    childTypes_ = child_->OutputTypes();
    auto stage = [&](int col) {
        auto it = std::find(stagedCols_.begin(), stagedCols_.end(), col);
        if (it == stagedCols_.end()) { stagedCols_.push_back(col); return (int)stagedCols_.size() - 1; }
        return (int)(it - stagedCols_.begin());
    };
    for (int g : groupCols_) stage(g);
    // remap expression columns from child columns to staged positions
    for (auto &a : aggs_)
        for (auto &o : a.prog)
            if (o.op == PH_X_COL) o.col = stage(o.col);
    batch_.reset(new DeviceBatch(ctx_, childTypes_, stagedCols_));
    // output: group columns (child types) then one column per aggregate (FinalizeStates types)
    for (int g : groupCols_) outTypes_.push_back(childTypes_[(size_t)g]);
    std::vector<ph_col> protos;
    for (int c : stagedCols_) { ph_col p{}; p.type = staged_phtype(childTypes_[(size_t)c]); p.scale = childTypes_[(size_t)c].Scale; protos.push_back(p); }
    std::vector<ph_aggspec> specs;
    for (size_t i = 0; i < aggs_.size(); i++) {
        const AggExpr &a = aggs_[i];
        int32_t scale = 0;
        LType at = IntegerType();
        if (a.kind != PH_A_COUNT_STAR) {
            if (a.prog.empty()) return "aggregate without an argument";
            if (a.prog.size() == 1 && a.prog[0].op == PH_X_COL) {
                at = childTypes_[(size_t)stagedCols_[(size_t)a.prog[0].col]];
                scale = at.Scale;
            } else {
                if (ph_expr_scale(protos.data(), a.prog.data(), (int32_t)a.prog.size(), &scale) != PH_OK) return herr("ph_expr_scale");
                at = DecimalType(38, scale);
            }
        }
        argScale_.push_back(scale);
        argType_.push_back(at);
        bool dec = at.Id == LTID_DECIMAL;
        switch (a.kind) {
        case PH_A_SUM: outTypes_.push_back(dec ? DecimalType(38, scale) : HugeintType()); break;   // BindDecimalSum / GetSumAggr
        case PH_A_AVG: outTypes_.push_back(dec ? DecimalType(38, scale) : DoubleType()); break;    // BindDecimalAvg / GetAvgAggr
        case PH_A_COUNT: case PH_A_COUNT_STAR: outTypes_.push_back(HugeintType()); break;
        case PH_A_MIN: case PH_A_MAX: outTypes_.push_back(dec ? DecimalType(at.Width, scale) : at); break;
        default: return "unknown aggregate kind";
        }
        specs.push_back(ph_aggspec{a.kind, (int32_t)i});
    }
    std::vector<int32_t> keyTypes;
    for (int g : groupCols_) keyTypes.push_back(staged_phtype(childTypes_[(size_t)g]));
    if (keyTypes.empty()) keyTypes.push_back(PH_I32);  // ungrouped: constant key (executor_aggr.go:37-48)
    if (ph_agg_create(ctx_, (int32_t)keyTypes.size(), keyTypes.data(), (int32_t)specs.size(), specs.data(), 1024, &agg_) != PH_OK)
        return herr("ph_agg_create");
    for (auto &h : having_) if (h.col < 0 || h.col >= (int)outTypes_.size()) return "HAVING column out of range";
    if (!outputs_.empty()) { std::string e = gpuProjectExecutor::Types(outputs_, outTypes_, &finalTypes_); if (!e.empty()) return e; }
    return "";
}

std::string gpuAggExecutor::Close() {
    if (agg_) { ph_agg_free(agg_); agg_ = nullptr; }
    batch_.reset();
    results_.clear();
    return "";
}

std::string gpuAggExecutor::sinkBatch() {
    int64_t n = batch_->rows();
    if (n == 0) return "";
    std::string e = batch_->Upload();
    if (!e.empty()) return e;
    std::vector<ph_col> staged;
    for (size_t k = 0; k < stagedCols_.size(); k++) staged.push_back(batch_->col((int)k));
    std::vector<void *> temps;
    auto cleanup = [&]() { for (void *p : temps) ph_dev_free(ctx_, p); };
    // keys
    std::vector<ph_col> keys;
    for (size_t g = 0; g < groupCols_.size(); g++) keys.push_back(staged[g]);
    if (keys.empty()) {
        void *zero = nullptr;
        if (ph_dev_alloc(ctx_, n * 4, &zero) != PH_OK || ph_dev_memset(ctx_, zero, 0, n * 4) != PH_OK) { cleanup(); return herr("constant key"); }
        temps.push_back(zero);
        ph_col c{}; c.type = PH_I32; c.data = zero;
        keys.push_back(c);
    }
    // arguments: a bare column is used as is, anything else is evaluated (executeExprs)
    std::vector<ph_col> args(aggs_.size());
    bool anyValidity = false;
    for (auto &c : staged) anyValidity |= c.validity != nullptr;
    for (size_t i = 0; i < aggs_.size(); i++) {
        const AggExpr &a = aggs_[i];
        if (a.kind == PH_A_COUNT_STAR) { args[i] = keys[0]; continue; }
        if (a.prog.size() == 1 && a.prog[0].op == PH_X_COL) { args[i] = staged[(size_t)a.prog[0].col]; continue; }
        void *out = nullptr, *val = nullptr;
        if (ph_dev_alloc(ctx_, n * 8, &out) != PH_OK) { cleanup(); return herr("ph_dev_alloc"); }
        temps.push_back(out);
        if (anyValidity) { if (ph_dev_alloc(ctx_, (n + 7) / 8 + 64, &val) != PH_OK) { cleanup(); return herr("ph_dev_alloc"); } temps.push_back(val); }
        if (ph_expr_eval(ctx_, staged.data(), (int32_t)staged.size(), a.prog.data(), (int32_t)a.prog.size(), nullptr, n,
                         (int64_t *)out, (uint8_t *)val) != PH_OK) { cleanup(); return herr("ph_expr_eval"); }
        ph_col c{}; c.type = PH_DEC64; c.scale = argScale_[i]; c.data = out; c.validity = (const uint8_t *)val;
        args[i] = c;
    }
    int rc = ph_agg_sink(agg_, keys.data(), args.data(), (int32_t)args.size(), nullptr, n, 0, rowBase_);
    std::string err = rc == PH_OK ? "" : herr("ph_agg_sink");
    if (rc == PH_OK && ph_ctx_sync(ctx_) != PH_OK) err = herr("ph_ctx_sync");
    cleanup();
    rowBase_ += n;
    batch_->Reset();   // dictionaries persist across batches; staged rows do not
    return err;
}

std::string BuildAggOutput(const std::vector<LType> &outTypes, const std::vector<LType> &keyTypes,
                           const std::vector<const std::vector<std::string> *> &keyDicts,
                           const std::vector<int> &aggKinds, const std::vector<LType> &argTypes,
                           const std::vector<int> &argScales, int64_t ng, const int64_t *keys,
                           const uint8_t *knull, const uint64_t *lo, const int64_t *hi, const uint64_t *cnt,
                           std::vector<std::shared_ptr<Chunk>> *results) {
    size_t nkOut = keyTypes.size();
    int nk = std::max<int>((int)nkOut, 1), na = (int)aggKinds.size();
    for (int64_t base = 0; base < ng; base += DefaultVectorSize) {
        int card = (int)std::min<int64_t>(DefaultVectorSize, ng - base);
        auto out = std::make_shared<Chunk>();
        out->Init(outTypes, ChunkCapacityFor(card));
        for (int r = 0; r < card; r++) {
            size_t gi = (size_t)(base + r);
            for (size_t c = 0; c < nkOut; c++) {
                Vector &v = *out->Data[c];
                if (knull && knull[gi * (size_t)nk + c]) { v.Mask.SetInvalid((uint64_t)r, DefaultVectorSize); continue; }
                int64_t kv = keys[gi * (size_t)nk + c];
                switch (v._Typ.GetInternalType()) {
                case PT_INT32: v.Slice<int32_t>()[r] = (int32_t)kv; break;
                case PT_INT64: v.Slice<int64_t>()[r] = kv; break;
                case PT_DATE: v.Slice<Date>()[r] = DateFromDays((int32_t)kv); break;
                case PT_DECIMAL: v.Slice<Decimal>()[r] = DecimalFromUnscaled(kv, v._Typ.Scale); break;
                case PT_INT128: v.Slice<Hugeint>()[r] = Hugeint{(uint64_t)kv, kv < 0 ? -1 : 0}; break;   // a SUM / COUNT of an aggregate below, grouped by above
                case PT_VARCHAR: { const std::string &s = (*keyDicts[c])[(size_t)kv]; v.SetString(r, s.data(), (int64_t)s.size()); break; }
                default: break;
                }
            }
            for (int a = 0; a < na; a++) {
                Vector &v = *out->Data[nkOut + (size_t)a];
                size_t si = gi * (size_t)na + (size_t)a;
                __int128 sum = ((__int128)hi[si] << 64) + (__int128)(unsigned __int128)lo[si];
                uint64_t n = cnt[si];
                bool dec = argTypes[(size_t)a].Id == LTID_DECIMAL;
                auto null = [&]() { v.Mask.SetInvalid((uint64_t)r, DefaultVectorSize); };
                switch (aggKinds[(size_t)a]) {
                case PH_A_SUM:  // SumOp.Finalize: NULL when never set (function_aggr.go:813-823)
                    if (n == 0) { null(); break; }
                    if (dec) { Decimal d; if (!DecimalFromInt128(sum, argScales[(size_t)a], &d)) return "decimal sum exceeds 19 digits"; v.Slice<Decimal>()[r] = d; }
                    else v.Slice<Hugeint>()[r] = Hugeint{lo[si], hi[si]};
                    break;
                case PH_A_AVG:  // AvgOp.Finalize (:873-900)
                    if (n == 0) { null(); break; }
                    if (dec) { Decimal d; if (!DecimalQuoCount(sum, argScales[(size_t)a], n, &d)) return "decimal average failed"; v.Slice<Decimal>()[r] = d; }
                    else v.Slice<double>()[r] = (double)sum / (double)n;
                    break;
                case PH_A_COUNT: case PH_A_COUNT_STAR: case PH_A_COUNT_DISTINCT:  // CountOp.Finalize: NULL when 0 (:950-962)
                    if (n == 0) { null(); break; }
                    v.Slice<Hugeint>()[r] = Hugeint{n, 0};
                    break;
                case PH_A_MIN: case PH_A_MAX:
                    if (n == 0) { null(); break; }
                    if (dec) v.Slice<Decimal>()[r] = DecimalFromUnscaled((int64_t)lo[si], argScales[(size_t)a]);
                    else if (v._Typ.GetInternalType() == PT_INT32) v.Slice<int32_t>()[r] = (int32_t)(int64_t)lo[si];
                    else v.Slice<int64_t>()[r] = (int64_t)lo[si];
                    break;
                default: break;
                }
            }
        }
        out->SetCard(card);
        results->push_back(out);
    }
    return "";
}

std::string gpuAggExecutor::finalize() {
    int64_t ng = 0;
    if (ph_agg_group_count(agg_, &ng) != PH_OK) return herr("ph_agg_group_count");
    int nk = std::max<int>((int)groupCols_.size(), 1), na = (int)aggs_.size();
    size_t g = (size_t)std::max<int64_t>(ng, 1);
    std::vector<int64_t> keys(g * (size_t)nk), hi(g * (size_t)std::max(na, 1));
    std::vector<uint8_t> knull(g * (size_t)nk);
    std::vector<uint64_t> lo(g * (size_t)std::max(na, 1)), cnt(g * (size_t)std::max(na, 1));
    if (ph_agg_finalize(agg_, (int64_t)g, nullptr, keys.data(), knull.data(), lo.data(), hi.data(), cnt.data()) != PH_OK)
        return herr("ph_agg_finalize");
    std::vector<LType> keyTypes;
    std::vector<const std::vector<std::string> *> dicts;
    for (size_t c = 0; c < groupCols_.size(); c++) { keyTypes.push_back(childTypes_[(size_t)groupCols_[c]]); dicts.push_back(&batch_->dict((int)c)); }
    std::vector<int> kinds;
    for (auto &a : aggs_) kinds.push_back(a.kind);
    std::string e = BuildAggOutput(outTypes_, keyTypes, dicts, kinds, argType_, argScale_, ng, keys.data(), knull.data(), lo.data(),
                                   hi.data(), cnt.data(), &results_);
    if (e.empty()) e = ApplyAggOutputPhase(ctx_, having_, outputs_, outTypes_, finalTypes_, &results_);
    return e;
}

// The output phase of aggExecutor.Execute (executor_aggr.go:143-263) over the finalised
// [group columns | aggregate results] chunks: HAVING = the conjuncts' selection per chunk
// (filterExec.executeSelect + SliceIndice), then the output expressions (outputExec.executeExprs).
std::string ApplyAggOutputPhase(ph_ctx *ctx, const std::vector<Compare> &having, const std::vector<ProjExpr> &outputs,
                                const std::vector<LType> &rowTypes, const std::vector<LType> &finalTypes,
                                std::vector<std::shared_ptr<Chunk>> *chunks) {
    if (!having.empty()) {
        size_t next = 0;
        sourceExecutor src(rowTypes, [&](Chunk *out) {
            if (next >= chunks->size()) return false;
            *out = *(*chunks)[next++];
            return true;
        });
        gpuFilterExecutor filt(ctx, having, &src);
        std::string e = filt.Init();
        if (!e.empty()) return e;
        std::vector<std::shared_ptr<Chunk>> kept;
        for (;;) {
            auto c = std::make_shared<Chunk>();
            std::string err;
            OperatorResult r = filt.Execute(nullptr, c.get(), &err);
            if (r == InvalidOpResult) return err.empty() ? "HAVING failed" : err;
            if (r == Done) break;
            if (c->Card() > 0) kept.push_back(c);
        }
        filt.Close();
        *chunks = kept;
    }
    if (!outputs.empty()) {
        std::vector<std::shared_ptr<Chunk>> out;
        std::string e = gpuProjectExecutor::Evaluate(ctx, outputs, rowTypes, finalTypes, *chunks, &out);
        if (!e.empty()) return e;
        *chunks = out;
    }
    return "";
}

OperatorResult gpuAggExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (!built_) {  // pipeline breaker: drain the child (HAS_INIT loop, executor_aggr.go:110-142)
        for (;;) {
            Chunk c;
            OperatorResult r = child_->Execute(nullptr, &c, err);
            if (r == InvalidOpResult) return InvalidOpResult;
            if (r == Done) break;
            if (c.Card() == 0) continue;
            std::string e = batch_->Append(c);
            if (e.empty() && batch_->rows() >= batchRows_) e = sinkBatch();
            if (!e.empty()) { *err = e; return InvalidOpResult; }
        }
        std::string e = sinkBatch();
        if (e.empty()) e = finalize();
        if (!e.empty()) { *err = e; return InvalidOpResult; }
        built_ = true;
    }
    if (next_ >= results_.size()) return Done;
    *output = *results_[next_++];
    return haveMoreOutput;
}

// ------------------------------------------------------------------ scan + aggregate over a resident table

gpuScanAggExecutor::gpuScanAggExecutor(ph_ctx *ctx, const ph_table *table, std::vector<ResidentColumn> columns,
                                       std::vector<Compare> conjuncts, std::vector<int> groupCols, std::vector<AggExpr> aggs)
    : ctx_(ctx), table_(table), cols_(std::move(columns)), conj_(std::move(conjuncts)), groupCols_(std::move(groupCols)),
      aggs_(std::move(aggs)) {}

std::string gpuScanAggExecutor::Init() {
    std::vector<ph_pred> preds;
    for (auto &c : conj_) {
        ph_pred p{};
        p.col = c.col;
        p.op = c.op;
        switch (c.k.kind) {
        case Literal::Int: p.k.type = PH_I32; p.k.i = c.k.i; break;
        case Literal::Float: p.k.type = PH_F32; p.k.f = c.k.f; break;
        case Literal::DateDays: p.k.type = PH_DATE; p.k.i = c.k.i; break;
        case Literal::Dec: p.k.type = PH_DEC64; p.k.i = c.k.i; p.k.scale = c.k.scale; break;
        case Literal::Str: p.k.type = PH_STR; p.k.s = c.k.s.c_str(); break;
        }
        preds.push_back(p);
    }
    std::vector<ph_col> protos(cols_.size());
    for (size_t c = 0; c < cols_.size(); c++) { protos[c].type = staged_phtype(cols_[c].type); protos[c].scale = cols_[c].type.Scale; }
    std::vector<ph_aggexpr> ax(aggs_.size());
    for (int g : groupCols_) outTypes_.push_back(cols_[(size_t)g].type);
    for (size_t i = 0; i < aggs_.size(); i++) {
        const AggExpr &a = aggs_[i];
        ax[i].kind = a.kind;
        ax[i].nprog = (int32_t)a.prog.size();
        if (a.prog.size() > 12) return "aggregate argument program too long";
        for (size_t k = 0; k < a.prog.size(); k++) ax[i].prog[k] = a.prog[k];
        LType at = IntegerType();
        int32_t scale = 0;
        if (a.kind != PH_A_COUNT_STAR) {
            if (a.prog.size() == 1 && a.prog[0].op == PH_X_COL) { at = cols_[(size_t)a.prog[0].col].type; scale = at.Scale; }
            else {
                if (ph_expr_scale(protos.data(), a.prog.data(), (int32_t)a.prog.size(), &scale) != PH_OK) return herr("ph_expr_scale");
                at = DecimalType(38, scale);
            }
        }
        argType_.push_back(at);
        bool dec = at.Id == LTID_DECIMAL;
        switch (a.kind) {
        case PH_A_SUM: outTypes_.push_back(dec ? DecimalType(38, scale) : HugeintType()); break;
        case PH_A_AVG: outTypes_.push_back(dec ? DecimalType(38, scale) : DoubleType()); break;
        case PH_A_COUNT: case PH_A_COUNT_STAR: outTypes_.push_back(HugeintType()); break;
        case PH_A_MIN: case PH_A_MAX: outTypes_.push_back(dec ? DecimalType(at.Width, scale) : at); break;
        default: return "unknown aggregate kind";
        }
    }
    std::vector<int32_t> groups(groupCols_.begin(), groupCols_.end());
    if (ph_scan_plan_create(ctx_, table_, preds.data(), (int32_t)preds.size(), groups.data(), (int32_t)groups.size(), ax.data(),
                            (int32_t)ax.size(), &plan_) != PH_OK)
        return herr("ph_scan_plan_create");
    for (auto &h : having_) if (h.col < 0 || h.col >= (int)outTypes_.size()) return "HAVING column out of range";
    if (!outputs_.empty()) { std::string e = gpuProjectExecutor::Types(outputs_, outTypes_, &finalTypes_); if (!e.empty()) return e; }
    return "";
}

std::string gpuScanAggExecutor::Close() {
    if (plan_) { ph_scan_plan_free(plan_); plan_ = nullptr; }
    results_.clear();
    return "";
}

OperatorResult gpuScanAggExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (!built_) {
        ph_agg_result *r = nullptr;
        if (ph_scan_plan_run(plan_, 0, ph_table_rows(table_)) != PH_OK || ph_scan_plan_fetch(plan_, &r) != PH_OK) {
            *err = herr("ph_scan_plan_run/fetch");
            return InvalidOpResult;
        }
        std::vector<LType> keyTypes;
        std::vector<const std::vector<std::string> *> dicts;
        for (int g : groupCols_) { keyTypes.push_back(cols_[(size_t)g].type); dicts.push_back(&cols_[(size_t)g].dict); }
        std::vector<int> kinds, scales;
        for (size_t i = 0; i < aggs_.size(); i++) { kinds.push_back(aggs_[i].kind); scales.push_back(r->scale[i]); }
        std::string e = BuildAggOutput(outTypes_, keyTypes, dicts, kinds, argType_, scales, r->ngroups, r->keys, nullptr, r->sum_lo,
                                       r->sum_hi, r->count, &results_);
        ph_agg_result_free(r);
        if (e.empty()) e = ApplyAggOutputPhase(ctx_, having_, outputs_, outTypes_, finalTypes_, &results_);
        if (!e.empty()) { *err = e; return InvalidOpResult; }
        built_ = true;
    }
    if (next_ >= results_.size()) return Done;
    *output = *results_[next_++];
    return haveMoreOutput;
}

// ------------------------------------------------------------------ join

gpuJoinExecutor::gpuJoinExecutor(ph_ctx *ctx, OperatorExec *probe, OperatorExec *build, std::vector<int> probeKeys,
                                 std::vector<int> buildKeys, std::vector<int> buildPayload, int batchChunks, JoinType type)
    : ctx_(ctx), probe_(probe), build_(build), probeKeys_(std::move(probeKeys)), buildKeys_(std::move(buildKeys)),
      buildPayload_(std::move(buildPayload)), batchChunks_(batchChunks), type_(type) {}

std::string gpuJoinExecutor::Init() {
    if (probeKeys_.size() != buildKeys_.size() || probeKeys_.empty()) return "join needs matching key lists";
    auto pt = probe_->OutputTypes(), bt = build_->OutputTypes();
    for (size_t i = 0; i < probeKeys_.size(); i++) {
        PhyType a = pt[(size_t)probeKeys_[i]].GetInternalType(), b = bt[(size_t)buildKeys_[i]].GetInternalType();
        if (a == PT_VARCHAR || b == PT_VARCHAR) return "VARCHAR join keys stay on the CPU executor";
        if (staged_width(pt[(size_t)probeKeys_[i]]) != staged_width(bt[(size_t)buildKeys_[i]])) return "join key widths differ";
    }
    outTypes_ = pt;
    if (type_ == JoinInner || type_ == JoinLeft) for (int c : buildPayload_) outTypes_.push_back(bt[(size_t)c]);
    buildBatch_.reset(new DeviceBatch(ctx_, bt, buildKeys_));
    probeBatch_.reset(new DeviceBatch(ctx_, pt, probeKeys_));
    return "";
}

std::string gpuJoinExecutor::Close() {
    if (join_) { ph_join_free(join_); join_ = nullptr; }
    buildBatch_.reset(); probeBatch_.reset(); buildChunks_.clear(); ready_.clear();
    return "";
}

std::string gpuJoinExecutor::buildTable() {  // joinBuildHashTable (executor_join.go:237-264)
    int64_t total = 0;
    for (;;) {
        auto c = std::make_shared<Chunk>();
        std::string err;
        OperatorResult r = build_->Execute(nullptr, c.get(), &err);
        if (r == InvalidOpResult) return err.empty() ? "build child failed" : err;
        if (r == Done) break;
        if (c->Card() == 0) continue;
        std::string e = buildBatch_->Append(*c);
        if (!e.empty()) return e;
        buildStart_.push_back(total);
        buildChunks_.push_back(c);
        total += c->Card();
    }
    std::string e = buildBatch_->Upload();
    if (!e.empty()) return e;
    std::vector<ph_col> keys;
    for (size_t k = 0; k < buildKeys_.size(); k++) keys.push_back(buildBatch_->col((int)k));
    // one INTEGER / BIGINT key: the key range noted while staging lets the library build a direct table for
    // dense keys (a primary key); anything else builds what ph_join_build builds
    int64_t lo = 0, hi = 0;
    const int32_t flags = keys.size() == 1 && buildBatch_->key_range(0, &lo, &hi) ? PH_JOIN_KEY_RANGE : 0;
    if (ph_join_build_ex(ctx_, keys.data(), (int32_t)keys.size(), nullptr, total, flags, lo, hi, &join_) != PH_OK) return herr("ph_join_build_ex");
    return "";
}

std::string gpuJoinExecutor::probeBatch() {
    std::vector<std::shared_ptr<Chunk>> chunks;
    std::vector<int64_t> starts;
    probeBatch_->Reset();
    while ((int)chunks.size() < batchChunks_) {
        auto c = std::make_shared<Chunk>();
        std::string err;
        OperatorResult r = probe_->Execute(nullptr, c.get(), &err);
        if (r == InvalidOpResult) return err.empty() ? "probe child failed" : err;
        if (r == Done) { probeDone_ = true; break; }
        if (c->Card() == 0) continue;
        starts.push_back(probeBatch_->rows());
        std::string e = probeBatch_->Append(*c);
        if (!e.empty()) return e;
        chunks.push_back(c);
    }
    int64_t n = probeBatch_->rows();
    if (n == 0) return "";
    if (ph_join_count(join_) == 0 && type_ != JoinAnti && type_ != JoinLeft) return "";
    std::string e = probeBatch_->Upload();
    if (!e.empty()) return e;
    std::vector<ph_col> keys;
    for (size_t k = 0; k < probeKeys_.size(); k++) keys.push_back(probeBatch_->col((int)k));
    if (type_ == JoinSemi || type_ == JoinAnti) {
        // ScanKeyMatches + NextSemiOrAntiJoin (join_scan.go:120-180): found flag per probe row,
        // then the probe chunk sliced by the rows with found == (type is SEMI)
        void *fd = nullptr;
        if (ph_dev_alloc(ctx_, n, &fd) != PH_OK) return herr("ph_dev_alloc");
        std::vector<uint8_t> found((size_t)n);
        if (ph_join_probe_mark(join_, keys.data(), nullptr, n, (uint8_t *)fd) != PH_OK ||
            ph_dev_download(ctx_, found.data(), fd, n) != PH_OK) { ph_dev_free(ctx_, fd); return herr("ph_join_probe_mark"); }
        ph_dev_free(ctx_, fd);
        uint8_t want = type_ == JoinSemi ? 1 : 0;
        for (size_t ci = 0; ci < chunks.size(); ci++) {
            auto sv = std::make_shared<SelectVector>();
            sv->identity = false;
            for (int i = 0; i < chunks[ci]->Card(); i++)
                if (found[(size_t)(starts[ci] + i)] == want) sv->SelVec.push_back(i);
            if (sv->SelVec.empty()) continue;
            auto out = std::make_shared<Chunk>();
            out->Init(outTypes_, DefaultVectorSize);
            std::vector<int> indice;
            for (int c = 0; c < chunks[ci]->ColumnCount(); c++) indice.push_back(c);
            out->SliceIndice(*chunks[ci], sv, (int)sv->SelVec.size(), 0, indice);
            ready_.push_back(out);
        }
        return "";
    }
    int64_t cap = n + 1024, m = 0;
    void *op = nullptr, *ob = nullptr;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (ph_dev_alloc(ctx_, cap * 4, &op) != PH_OK || ph_dev_alloc(ctx_, cap * 4, &ob) != PH_OK) return herr("ph_dev_alloc");
        int rc = ph_join_probe_inner(join_, keys.data(), nullptr, n, (int32_t *)op, (int32_t *)ob, cap, &m);
        if (rc == PH_OK) break;
        ph_dev_free(ctx_, op); ph_dev_free(ctx_, ob);
        op = ob = nullptr;
        if (rc != PH_ECAPACITY || attempt == 1) return herr("ph_join_probe_inner");
        cap = m;  // duplicate build keys: retry with the exact size
    }
    std::vector<int32_t> pr((size_t)m), br((size_t)m);
    if (m > 0 && (ph_dev_download(ctx_, pr.data(), op, m * 4) != PH_OK || ph_dev_download(ctx_, br.data(), ob, m * 4) != PH_OK)) {
        ph_dev_free(ctx_, op); ph_dev_free(ctx_, ob);
        return herr("ph_dev_download");
    }
    ph_dev_free(ctx_, op);
    ph_dev_free(ctx_, ob);
    // materialise <= 2048-row result chunks: probe columns, then the build payload (gatherResult)
    std::shared_ptr<Chunk> out;
    int np = (int)probe_->OutputTypes().size();
    for (int64_t i = 0; i < m; i++) {
        if (!out || out->Card() == DefaultVectorSize) {
            if (out) ready_.push_back(out);
            out = std::make_shared<Chunk>();
            out->Init(outTypes_, DefaultVectorSize);
        }
        size_t pc = (size_t)(std::upper_bound(starts.begin(), starts.end(), (int64_t)pr[(size_t)i]) - starts.begin()) - 1;
        size_t bc = (size_t)(std::upper_bound(buildStart_.begin(), buildStart_.end(), (int64_t)br[(size_t)i]) - buildStart_.begin()) - 1;
        int prow = (int)(pr[(size_t)i] - starts[pc]), brow = (int)(br[(size_t)i] - buildStart_[bc]);
        int r = out->Card();
        for (int c = 0; c < np; c++) CopyCell(*chunks[pc]->Data[(size_t)c], prow, out->Data[(size_t)c].get(), r);
        for (size_t c = 0; c < buildPayload_.size(); c++)
            CopyCell(*buildChunks_[bc]->Data[(size_t)buildPayload_[c]], brow, out->Data[(size_t)np + c].get(), r);
        out->SetCard(r + 1);
    }
    if (out && out->Card() > 0) ready_.push_back(out);
    if (type_ == JoinLeft) {
        // NextLeftJoin (join_scan.go:67-88): the probe rows no pair mentions, sliced out of their
        // chunk, with every build-side column a constant NULL
        std::vector<uint8_t> matched((size_t)n, 0);
        for (int64_t i = 0; i < m; i++) matched[(size_t)pr[(size_t)i]] = 1;
        for (size_t ci = 0; ci < chunks.size(); ci++) {
            auto sv = std::make_shared<SelectVector>();
            sv->identity = false;
            for (int i = 0; i < chunks[ci]->Card(); i++)
                if (!matched[(size_t)(starts[ci] + i)]) sv->SelVec.push_back(i);
            if (sv->SelVec.empty()) continue;
            auto rest = std::make_shared<Chunk>();
            rest->Init(outTypes_, DefaultVectorSize);
            std::vector<int> indice;
            for (int c = 0; c < np; c++) indice.push_back(c);
            rest->SliceIndice(*chunks[ci], sv, (int)sv->SelVec.size(), 0, indice);
            for (size_t c = 0; c < buildPayload_.size(); c++) rest->Data[(size_t)np + c]->SetConstNull();
            ready_.push_back(rest);
        }
    }
    return "";
}

OperatorResult gpuJoinExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (!built_) {
        std::string e = buildTable();
        if (!e.empty()) { *err = e; return InvalidOpResult; }
        built_ = true;
    }
    while (ready_.empty() && !probeDone_) {
        std::string e = probeBatch();
        if (!e.empty()) { *err = e; return InvalidOpResult; }
    }
    if (ready_.empty()) return Done;
    *output = *ready_.front();
    ready_.pop_front();
    return haveMoreOutput;
}

// ------------------------------------------------------------------ order

gpuOrderExecutor::gpuOrderExecutor(ph_ctx *ctx, std::vector<OrderKey> keys, OperatorExec *child)
    : ctx_(ctx), keys_(std::move(keys)), child_(child) {}

std::string gpuOrderExecutor::Init() {
    auto t = child_->OutputTypes();
    for (auto &k : keys_) {
        if (k.col < 0 || k.col >= (int)t.size()) return "order key column out of range";
        PhyType p = t[(size_t)k.col].GetInternalType();
        if (p != PT_INT32 && p != PT_DATE && p != PT_DECIMAL && p != PT_VARCHAR && p != PT_INT128)
            return "ORDER BY key type stays on the CPU executor";   // no RadixScatter case (sort_radix.go:257-321)
    }
    return "";
}

std::string gpuOrderExecutor::Close() { unified_.clear(); chunks_.clear(); order_.clear(); return ""; }

// inputs up to this many rows are ordered on the host (PH_ORDER_HOST_ROWS moves it; 0 = always the device sort, which the
// parity tests use to run both forms over the same rows)
static const int64_t kHostSortRows = getenv("PH_ORDER_HOST_ROWS") ? atoll(getenv("PH_ORDER_HOST_ROWS")) : (1 << 17);

std::string gpuOrderExecutor::sortAll() {
    static const bool timing = getenv("PH_HOST_TIMING") != nullptr;
    struct Tm { bool on; double t0; ~Tm() { if (on) fprintf(stderr, "  order: sortAll %.1f us (incl. the child's Execute)\n", (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0) * 1e6); } }
        tm{timing, std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()};
    // VARCHAR keys do not go through the batch (its VARCHAR staging is the <= 256-value dictionary of a group column): their rows
    // are ranked on the host (rank = position of the string among the distinct strings in byte order) and the ranks are the key
    std::vector<int> cols, batchPos(keys_.size(), -1);
    {
        auto types = child_->OutputTypes();
        for (size_t k = 0; k < keys_.size(); k++)
            if (types[(size_t)keys_[k].col].GetInternalType() != PT_VARCHAR) { batchPos[k] = (int)cols.size(); cols.push_back(keys_[k].col); }
    }
    DeviceBatch batch(ctx_, child_->OutputTypes(), cols);
    int64_t total = 0;
    for (;;) {   // SinkChunk for every child chunk (executor_order.go:75-99)
        auto c = std::make_shared<Chunk>();
        std::string err;
        OperatorResult r = child_->Execute(nullptr, c.get(), &err);
        if (r == InvalidOpResult) return err.empty() ? "child failed" : err;
        if (r == Done) break;
        if (c->Card() == 0) continue;
        start_.push_back(total);
        chunks_.push_back(c);
        total += c->Card();
    }
    if (total == 0) return "";
    // ranks of a VARCHAR key's rows: position of the row's string among the key's DISTINCT strings in byte order (the distinct strings are
    // found by hashing — a group column has few — and only they are sorted)
    auto rankStrings = [&](size_t k, std::vector<int32_t> *rank, std::vector<uint8_t> *valid, bool *anyNull) {
        rank->assign((size_t)total, 0);
        valid->assign((size_t)(total + 7) / 8 + 8, 0);
        *anyNull = false;
        std::unordered_map<std::string_view, int32_t> ids;
        std::vector<std::string_view> distinct;
        std::vector<int32_t> idOf((size_t)total, -1);
        int64_t row = 0;
        for (auto &ch : chunks_) {
            const Vector &v = *ch->Data[(size_t)keys_[k].col];
            Vector::Unified u;
            v.ToUnifiedFormat(ch->Card(), &u);
            for (int i = 0; i < ch->Card(); i++, row++) {
                const int64_t idx = u.sel->GetIndex(i);
                if (!u.mask->RowIsValid((uint64_t)idx)) { *anyNull = true; continue; }
                (*valid)[(size_t)row >> 3] |= (uint8_t)(1u << (row & 7));
                const String &sv = reinterpret_cast<const String *>(u.data)[idx];
                auto it = ids.emplace(std::string_view(sv.Data, (size_t)sv.Len), (int32_t)distinct.size());
                if (it.second) distinct.push_back(it.first->first);
                idOf[(size_t)row] = it.first->second;
            }
        }
        std::vector<int32_t> byOrder(distinct.size()), rankOfId(distinct.size());
        for (size_t i = 0; i < byOrder.size(); i++) byOrder[i] = (int32_t)i;
        std::sort(byOrder.begin(), byOrder.end(), [&](int32_t x, int32_t y) { return distinct[(size_t)x] < distinct[(size_t)y]; });
        for (size_t i = 0; i < byOrder.size(); i++) rankOfId[(size_t)byOrder[i]] = (int32_t)i;
        for (int64_t r = 0; r < total; r++) if (idOf[(size_t)r] >= 0) (*rank)[(size_t)r] = rankOfId[(size_t)idOf[(size_t)r]];
    };
    if (total <= kHostSortRows) {
        // The groups of an aggregate, a top-k preselection: LocalSort's order on the host — the same keys ph_sort_rows sorts by (NULLs first
        // whatever the direction, sort_layout.go:46; INTEGER / DATE by value, DECIMAL by dec.Int64(2) = the value rounded half-even to cents,
        // sort_encoder.go:65-70; VARCHAR bytewise, as ranks; DESC inverts the value order), ties in input order. Every key becomes an int64 per
        // row; when the keys' value ranges fit 64 bits together the rows are sorted by ONE packed word (Q16's 28 k groups: four keys in 30 bits).
        // Uploading rows, five launches and three read-backs cost ~0.1 ms for ten rows and 12 ms for 28 k (string ranks, allocations).
        std::vector<std::vector<int64_t>> kv(keys_.size(), std::vector<int64_t>((size_t)total, 0));
        std::vector<std::vector<uint8_t>> kn(keys_.size(), std::vector<uint8_t>((size_t)total, 0));
        auto types = child_->OutputTypes();
        for (size_t k = 0; k < keys_.size(); k++) {
            const LType &t = types[(size_t)keys_[k].col];
            if (t.GetInternalType() == PT_VARCHAR) {
                std::vector<int32_t> rank;
                std::vector<uint8_t> valid;
                bool anyNull = false;
                rankStrings(k, &rank, &valid, &anyNull);
                for (int64_t r = 0; r < total; r++) { kv[k][(size_t)r] = rank[(size_t)r]; kn[k][(size_t)r] = !((valid[(size_t)r >> 3] >> (r & 7)) & 1); }
                continue;
            }
            int64_t row = 0;
            for (auto &c : chunks_) {
                const Vector &v = *c->Data[(size_t)keys_[k].col];
                Vector::Unified u;
                v.ToUnifiedFormat(c->Card(), &u);
                for (int i = 0; i < c->Card(); i++, row++) {
                    int64_t idx = u.sel->GetIndex(i);
                    if (!u.mask->RowIsValid((uint64_t)idx)) { kn[k][(size_t)row] = 1; continue; }
                    int64_t &hv = kv[k][(size_t)row];
                    switch (t.GetInternalType()) {
                    case PT_INT32: hv = reinterpret_cast<const int32_t *>(u.data)[idx]; break;
                    case PT_DATE: hv = DaysFromDate(reinterpret_cast<const Date *>(u.data)[idx]); break;
                    case PT_DECIMAL: {
                        int64_t x;
                        if (!DecimalToUnscaled(reinterpret_cast<const Decimal *>(u.data)[idx], t.Scale, &x)) return "decimal ORDER BY key does not fit 18 digits";
                        if (t.Scale > 2) {   // round half-even to cents
                            int64_t p = 1;
                            for (int sc = 2; sc < t.Scale; sc++) p *= 10;
                            int64_t q = x / p, r = x % p, ar = r < 0 ? -r : r;
                            if (2 * ar > p || (2 * ar == p && (q & 1))) q += x < 0 ? -1 : 1;
                            x = q;
                        } else for (int sc = t.Scale; sc < 2; sc++) x *= 10;
                        hv = x;
                        break;
                    }
                    case PT_INT128: {   // hugeEncoder (sort_encoder.go:87-92): Upper then Lower = the signed 128-bit order; COUNT / SUM(INTEGER) results fit 64 bits
                        const Hugeint &h = reinterpret_cast<const Hugeint *>(u.data)[idx];
                        if (!((h.Upper == 0 && (int64_t)h.Lower >= 0) || (h.Upper == -1 && (int64_t)h.Lower < 0))) return "HUGEINT ORDER BY key beyond 64 bits";
                        hv = (int64_t)h.Lower;
                        break;
                    }
                    default: break;
                    }
                }
            }
        }
        order_.resize((size_t)total);
        for (int64_t i = 0; i < total; i++) order_[(size_t)i] = (int32_t)i;
        // one packed word: per key (value - min) or (max - value) for DESC, + 1 (0 = NULL, first either way), most significant key first
        std::vector<unsigned __int128> span(keys_.size());
        std::vector<int64_t> lo(keys_.size(), 0), hi(keys_.size(), 0);
        int bits = 0;
        bool packs = true;
        for (size_t k = 0; k < keys_.size() && packs; k++) {
            bool any = false;
            for (int64_t r = 0; r < total; r++) {
                if (kn[k][(size_t)r]) continue;
                const int64_t v = kv[k][(size_t)r];
                if (!any) { lo[k] = hi[k] = v; any = true; } else { lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
            }
            span[k] = (unsigned __int128)((__int128)hi[k] - (__int128)lo[k]) + 2;   // values 1 .. span-1, 0 = NULL
            int b = 0;
            while (b < 64 && ((unsigned __int128)1 << b) < span[k]) b++;
            if (((unsigned __int128)1 << b) < span[k]) packs = false;
            bits += b;
            if (bits > 64) packs = false;
        }
        if (packs) {
            std::vector<std::pair<uint64_t, int32_t>> rows((size_t)total);
            for (int64_t r = 0; r < total; r++) {
                uint64_t w = 0;
                for (size_t k = 0; k < keys_.size(); k++) {
                    int b = 0;
                    while (((unsigned __int128)1 << b) < span[k]) b++;
                    uint64_t f = 0;
                    if (!kn[k][(size_t)r]) f = (uint64_t)(keys_[k].descending ? (__int128)hi[k] - kv[k][(size_t)r] : (__int128)kv[k][(size_t)r] - lo[k]) + 1;
                    w = b == 64 ? f : (w << b) | f;
                }
                rows[(size_t)r] = {w, (int32_t)r};
            }
            std::sort(rows.begin(), rows.end());   // (word, input position): ties in input order
            for (int64_t r = 0; r < total; r++) order_[(size_t)r] = rows[(size_t)r].second;
            return "";
        }
        std::stable_sort(order_.begin(), order_.end(), [&](int32_t a, int32_t b) {
            for (size_t k = 0; k < keys_.size(); k++) {
                const bool xn = kn[k][(size_t)a], yn = kn[k][(size_t)b];
                if (xn != yn) return xn;            // NULLs first
                if (xn) continue;
                const int64_t x = kv[k][(size_t)a], y = kv[k][(size_t)b];
                if (x != y) return keys_[k].descending ? x > y : x < y;
            }
            return false;
        });
        return "";
    }
    for (auto &c : chunks_) { std::string e = batch.Append(*c); if (!e.empty()) return e; }
    std::string e = batch.Upload();
    if (!e.empty()) return e;
    std::vector<ph_col> kc;
    std::vector<int32_t> desc;
    std::vector<void *> recoded;
    auto freeRecoded = [&]() { for (void *d : recoded) ph_dev_free(ctx_, d); };
    for (size_t k = 0; k < keys_.size(); k++) {
        ph_col c{};
        if (batchPos[k] >= 0) c = batch.col(batchPos[k]);
        else {
            std::vector<int32_t> rank;
            std::vector<uint8_t> valid;
            bool anyNull = false;
            rankStrings(k, &rank, &valid, &anyNull);
            void *d = nullptr, *vd = nullptr;
            if (ph_dev_alloc(ctx_, total * 4, &d) != PH_OK) { freeRecoded(); return herr("ph_dev_alloc"); }
            recoded.push_back(d);
            if (ph_dev_upload(ctx_, d, rank.data(), total * 4) != PH_OK) { freeRecoded(); return herr("ph_dev_upload"); }
            if (anyNull) {
                if (ph_dev_alloc(ctx_, (int64_t)valid.size(), &vd) != PH_OK) { freeRecoded(); return herr("ph_dev_alloc"); }
                recoded.push_back(vd);
                if (ph_dev_upload(ctx_, vd, valid.data(), (int64_t)valid.size()) != PH_OK) { freeRecoded(); return herr("ph_dev_upload"); }
            }
            c.type = PH_I32; c.data = d; c.validity = (const uint8_t *)vd;
        }
        // (a HUGEINT key is staged as a scale-0 decimal of its low 64 bits: ph_sort_rows orders decimals by value)
        kc.push_back(c);
        desc.push_back(keys_[k].descending ? 1 : 0);
    }
    void *out = nullptr;
    std::string err;
    order_.resize((size_t)total);
    if (ph_dev_alloc(ctx_, total * 4, &out) != PH_OK) err = herr("ph_dev_alloc");
    else if (ph_sort_rows(ctx_, kc.data(), desc.data(), (int32_t)kc.size(), nullptr, total, (int32_t *)out) != PH_OK) err = herr("ph_sort_rows");
    else if (ph_dev_download(ctx_, order_.data(), out, total * 4) != PH_OK) err = herr("ph_dev_download");
    if (out) ph_dev_free(ctx_, out);
    for (void *d : recoded) ph_dev_free(ctx_, d);
    return err;
}

OperatorResult gpuOrderExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (!sorted_) {
        std::string e = sortAll();
        if (!e.empty()) { *err = e; return InvalidOpResult; }
        sorted_ = true;
    }
    if (next_ >= order_.size()) return Done;
    // PayloadScanner (executor_order.go:101-138): the next <= 2048 rows in sorted order
    int card = (int)std::min<size_t>((size_t)DefaultVectorSize, order_.size() - next_);
    output->Init(OutputTypes(), ChunkCapacityFor(card));
    int ncol = (int)OutputTypes().size();
    if (unified_.empty()) {   // every input vector's unified format once (CopyCell builds one per cell); sized first: a Unified may point into itself
        unified_.resize(chunks_.size());
        for (size_t ci = 0; ci < chunks_.size(); ci++) {
            unified_[ci] = std::vector<Vector::Unified>((size_t)ncol);
            for (int c = 0; c < ncol; c++) chunks_[ci]->Data[(size_t)c]->ToUnifiedFormat(chunks_[ci]->Card(), &unified_[ci][(size_t)c]);
        }
    }
    for (int r = 0; r < card; r++) {
        int64_t row = order_[next_ + (size_t)r];
        size_t ci = (size_t)(std::upper_bound(start_.begin(), start_.end(), row) - start_.begin()) - 1;
        int local = (int)(row - start_[ci]);
        for (int c = 0; c < ncol; c++) {
            const Vector &src = *chunks_[ci]->Data[(size_t)c];
            const Vector::Unified &u = unified_[ci][(size_t)c];
            Vector *dst = output->Data[(size_t)c].get();
            const int64_t idx = u.sel->GetIndex(local);
            if (!u.mask->RowIsValid((uint64_t)idx)) { dst->Mask.SetInvalid((uint64_t)r, DefaultVectorSize); continue; }
            // a VARCHAR cell references the input chunk's bytes: the input chunks live until Close, after every consumer of this output
            if (src._Typ.GetInternalType() == PT_VARCHAR) dst->Slice<String>()[r] = reinterpret_cast<const String *>(u.data)[idx];
            else { const size_t w = src._Typ.Size(); memcpy(dst->Data.data() + (size_t)r * w, u.data + (size_t)idx * w, w); }
        }
    }
    output->SetCard(card);
    next_ += (size_t)card;
    return haveMoreOutput;
}

}  // namespace plan

namespace plan {

// ------------------------------------------------------------------ project

std::string gpuProjectExecutor::Types(const std::vector<ProjExpr> &exprs, const std::vector<LType> &childTypes, std::vector<LType> *out) {
    out->clear();
    std::vector<ph_col> protos;
    for (auto &t : childTypes) { ph_col p{}; p.type = staged_phtype(t); p.scale = t.GetInternalType() == PT_INT128 ? 0 : t.Scale; protos.push_back(p); }
    for (auto &e : exprs) {
        switch (e.kind) {
        case ProjExpr::ColRef:
            if (e.col < 0 || e.col >= (int)childTypes.size()) return "project: column out of range";
            out->push_back(childTypes[(size_t)e.col]);
            break;
        case ProjExpr::Decimal: {
            for (auto &o : e.prog)
                if (o.op == PH_X_COL && (o.col < 0 || o.col >= (int)childTypes.size() || protos[(size_t)o.col].type == 0 || protos[(size_t)o.col].type == PH_CODE8))
                    return "project: expression column cannot be evaluated on the device";
            int32_t scale = 0;
            if (ph_expr_scale(protos.data(), e.prog.data(), (int32_t)e.prog.size(), &scale) != PH_OK) return herr("ph_expr_scale");
            out->push_back(DecimalType(38, scale));   // BindDecimalMultiply / AddSubstract widen to the cap
            break;
        }
        case ProjExpr::ExtractYear:
            if (e.col < 0 || e.col >= (int)childTypes.size() || childTypes[(size_t)e.col].GetInternalType() != PT_DATE) return "extract(year ...) needs a DATE column";
            out->push_back(IntegerType());            // ExtractFunc returns INTEGER (function_scalar.go:1509-1528)
            break;
        case ProjExpr::Substring:
            if (e.col < 0 || e.col >= (int)childTypes.size() || childTypes[(size_t)e.col].GetInternalType() != PT_VARCHAR) return "substring needs a VARCHAR column";
            out->push_back(VarcharType());
            break;
        case ProjExpr::Case: {
            if (e.resultInt) { out->push_back(IntegerType()); break; }
            int32_t scale = 0;
            if (ph_expr_scale(protos.data(), e.prog.data(), (int32_t)e.prog.size(), &scale) != PH_OK) return herr("ph_expr_scale");
            out->push_back(DecimalType(38, scale));
            break;
        }
        case ProjExpr::Float32:
            for (auto &o : e.fprog)
                if (o.op == FloatOp::Col && (o.col < 0 || o.col >= (int)childTypes.size())) return "float expression column out of range";
            out->push_back(FloatType());
            break;
        case ProjExpr::DecimalQuo:
            if (e.col < 0 || e.col >= (int)childTypes.size() || e.col2 < 0 || e.col2 >= (int)childTypes.size() ||
                childTypes[(size_t)e.col].GetInternalType() != PT_DECIMAL || childTypes[(size_t)e.col2].GetInternalType() != PT_DECIMAL)
                return "decimal division needs two DECIMAL columns";
            out->push_back(childTypes[(size_t)e.col]);
            break;
        }
    }
    return "";
}

// evaluates every non-reference expression over ALL rows of `in` in one device batch
std::string gpuProjectExecutor::Evaluate(ph_ctx *ctx, const std::vector<ProjExpr> &exprs, const std::vector<LType> &childTypes,
                                         const std::vector<LType> &outTypes, const std::vector<std::shared_ptr<Chunk>> &in,
                                         std::vector<std::shared_ptr<Chunk>> *out) {
    // columns the device needs
    std::vector<int> cols;
    std::vector<bool> asString;
    auto stage = [&](int c, bool str) {
        for (size_t i = 0; i < cols.size(); i++) if (cols[i] == c) { asString[i] = asString[i] || str; return (int)i; }
        cols.push_back(c); asString.push_back(str);
        return (int)cols.size() - 1;
    };
    std::vector<std::vector<ph_rpn>> progs(exprs.size());
    std::vector<int> arg(exprs.size(), -1);
    for (size_t i = 0; i < exprs.size(); i++) {
        const ProjExpr &e = exprs[i];
        if (e.kind == ProjExpr::Decimal) {
            progs[i] = e.prog;
            for (auto &o : progs[i]) if (o.op == PH_X_COL) o.col = stage(o.col, false);
        } else if (e.kind == ProjExpr::ExtractYear) arg[i] = stage(e.col, false);
        else if (e.kind == ProjExpr::Substring) arg[i] = stage(e.col, true);
    }
    int64_t total = 0;
    for (auto &c : in) total += c->Card();
    // device results, downloaded per expression
    std::vector<std::vector<int64_t>> dec(exprs.size());
    std::vector<std::vector<uint8_t>> decValid(exprs.size());
    std::vector<std::vector<int32_t>> i32(exprs.size()), soff(exprs.size());
    std::vector<std::vector<uint8_t>> sbytes(exprs.size());
    std::vector<bool> nullable(exprs.size(), false);
    if (!cols.empty() && total > 0) {
        DeviceBatch batch(ctx, childTypes, cols, asString);
        for (auto &c : in) { std::string e = batch.Append(*c); if (!e.empty()) return e; }
        std::string e = batch.Upload();
        if (!e.empty()) return e;
        std::vector<ph_col> staged;
        bool anyValidity = false;
        for (size_t k = 0; k < cols.size(); k++) { staged.push_back(batch.col((int)k)); anyValidity |= staged.back().validity != nullptr; }
        for (size_t i = 0; i < exprs.size(); i++) {
            const ProjExpr &ex = exprs[i];
            if (ex.kind == ProjExpr::Decimal) {
                void *o = nullptr, *v = nullptr;
                if (ph_dev_alloc(ctx, total * 8, &o) != PH_OK) return herr("ph_dev_alloc");
                if (anyValidity && ph_dev_alloc(ctx, (total + 7) / 8 + 64, &v) != PH_OK) { ph_dev_free(ctx, o); return herr("ph_dev_alloc"); }
                int rc = ph_expr_eval(ctx, staged.data(), (int32_t)staged.size(), progs[i].data(), (int32_t)progs[i].size(), nullptr, total,
                                      (int64_t *)o, (uint8_t *)v);
                dec[i].resize((size_t)total);
                if (rc == PH_OK) rc = ph_dev_download(ctx, dec[i].data(), o, total * 8);
                if (rc == PH_OK && v) { decValid[i].resize((size_t)(total + 7) / 8); rc = ph_dev_download(ctx, decValid[i].data(), v, (total + 7) / 8); nullable[i] = true; }
                ph_dev_free(ctx, o);
                if (v) ph_dev_free(ctx, v);
                if (rc != PH_OK) return herr("ph_expr_eval");
            } else if (ex.kind == ProjExpr::ExtractYear) {
                const ph_col &c = staged[(size_t)arg[i]];
                if (c.validity) return "extract over a NULL-able date stays on the CPU executor";
                void *o = nullptr;
                if (ph_dev_alloc(ctx, total * 4, &o) != PH_OK) return herr("ph_dev_alloc");
                i32[i].resize((size_t)total);
                int rc = ph_date_extract(ctx, PH_PART_YEAR, &c, nullptr, total, (int32_t *)o);
                if (rc == PH_OK) rc = ph_dev_download(ctx, i32[i].data(), o, total * 4);
                ph_dev_free(ctx, o);
                if (rc != PH_OK) return herr("ph_date_extract");
            } else if (ex.kind == ProjExpr::Substring) {
                const ph_col &c = staged[(size_t)arg[i]];
                void *o = nullptr, *b = nullptr;
                int64_t cap = c.aux_bytes + 64, nb = 0;
                if (ph_dev_alloc(ctx, (total + 1) * 4, &o) != PH_OK || ph_dev_alloc(ctx, cap, &b) != PH_OK) return herr("ph_dev_alloc");
                int rc = ph_substring(ctx, &c, ex.offset, ex.length, nullptr, total, (int32_t *)o, (uint8_t *)b, cap, &nb);
                soff[i].resize((size_t)total + 1);
                sbytes[i].resize((size_t)nb);
                if (rc == PH_OK) rc = ph_dev_download(ctx, soff[i].data(), o, (total + 1) * 4);
                if (rc == PH_OK && nb > 0) rc = ph_dev_download(ctx, sbytes[i].data(), b, nb);
                ph_dev_free(ctx, o); ph_dev_free(ctx, b);
                if (rc != PH_OK) return herr("ph_substring");
            }
        }
    }
    int64_t base = 0;
    for (auto &c : in) {
        auto oc = std::make_shared<Chunk>();
        oc->Init(outTypes, DefaultVectorSize);
        int card = c->Card();
        for (size_t i = 0; i < exprs.size(); i++) {
            const ProjExpr &ex = exprs[i];
            Vector &v = *oc->Data[i];
            if (ex.kind == ProjExpr::ColRef) { oc->Data[i] = c->Data[(size_t)ex.col]; continue; }   // Reference: no copy
            if (ex.kind == ProjExpr::Case) return "CASE expressions run inside resident plans only";
            if (ex.kind == ProjExpr::DecimalQuo) {
                Vector::Unified ua, ub;
                c->Data[(size_t)ex.col]->ToUnifiedFormat(card, &ua);
                c->Data[(size_t)ex.col2]->ToUnifiedFormat(card, &ub);
                for (int r = 0; r < card; r++) {
                    const int64_t ia = ua.sel->GetIndex(r), ib = ub.sel->GetIndex(r);
                    if (!ua.mask->RowIsValid((uint64_t)ia) || !ub.mask->RowIsValid((uint64_t)ib)) { v.Mask.SetInvalid((uint64_t)r, DefaultVectorSize); continue; }
                    Decimal q;
                    if (!DecimalQuo(reinterpret_cast<const Decimal *>(ua.data)[ia], reinterpret_cast<const Decimal *>(ub.data)[ib], &q)) return "decimal division failed (division by zero)";
                    v.Slice<Decimal>()[r] = q;
                }
                continue;
            }
            if (ex.kind == ProjExpr::Float32) {
                // FLOAT arithmetic, operand casts as the binder inserts them: DECIMAL -> float64 -> float32
                // (tryCastDecimalToFloat32), INTEGER -> float32, HUGEINT sums -> float32; every operation rounds to float32
                for (int r = 0; r < card; r++) {
                    std::vector<float> st;
                    bool null = false;
                    for (auto &o : ex.fprog) {
                        if (o.op == FloatOp::Const) { st.push_back(o.k); continue; }
                        if (o.op == FloatOp::Col) {
                            const Vector &src = *c->Data[(size_t)o.col];
                            Vector::Unified u;
                            src.ToUnifiedFormat(card, &u);
                            int64_t idx = u.sel->GetIndex(r);
                            if (!u.mask->RowIsValid((uint64_t)idx)) { null = true; st.push_back(0); continue; }
                            switch (src._Typ.GetInternalType()) {
                            case PT_DECIMAL: st.push_back((float)DecimalToDouble(reinterpret_cast<const Decimal *>(u.data)[idx])); break;
                            case PT_INT32: st.push_back((float)reinterpret_cast<const int32_t *>(u.data)[idx]); break;
                            case PT_FLOAT: st.push_back(reinterpret_cast<const float *>(u.data)[idx]); break;
                            case PT_DOUBLE: st.push_back((float)reinterpret_cast<const double *>(u.data)[idx]); break;
                            default: return "float expression over an unsupported column type";
                            }
                            continue;
                        }
                        if (st.size() < 2) return "malformed float expression";
                        volatile float b = st.back(); st.pop_back();
                        volatile float a = st.back(); st.pop_back();
                        volatile float res = o.op == FloatOp::Add ? a + b : o.op == FloatOp::Sub ? a - b : o.op == FloatOp::Mul ? a * b : a / b;
                        st.push_back((float)res);
                    }
                    if (st.size() != 1) return "malformed float expression";
                    if (null) v.Mask.SetInvalid((uint64_t)r, DefaultVectorSize); else v.Slice<float>()[r] = st[0];
                }
                continue;
            }
            for (int r = 0; r < card; r++) {
                size_t g = (size_t)(base + r);
                if (ex.kind == ProjExpr::Decimal) {
                    if (nullable[i] && !((decValid[i][g >> 3] >> (g & 7)) & 1)) { v.Mask.SetInvalid((uint64_t)r, DefaultVectorSize); continue; }
                    v.Slice<Decimal>()[r] = DecimalFromUnscaled(dec[i][g], outTypes[i].Scale);
                } else if (ex.kind == ProjExpr::ExtractYear) v.Slice<int32_t>()[r] = i32[i][g];
                else {
                    Vector::Unified u;
                    c->Data[(size_t)ex.col]->ToUnifiedFormat(card, &u);
                    if (!u.mask->RowIsValid((uint64_t)u.sel->GetIndex(r))) { v.Mask.SetInvalid((uint64_t)r, DefaultVectorSize); continue; }
                    v.SetString(r, (const char *)sbytes[i].data() + soff[i][g], soff[i][g + 1] - soff[i][g]);
                }
            }
        }
        oc->SetCard(card);
        out->push_back(oc);
        base += card;
    }
    return "";
}

gpuProjectExecutor::gpuProjectExecutor(ph_ctx *ctx, std::vector<ProjExpr> exprs, OperatorExec *child, int batchChunks)
    : ctx_(ctx), exprs_(std::move(exprs)), child_(child), batchChunks_(batchChunks) {}

std::string gpuProjectExecutor::Init() { return Types(exprs_, child_->OutputTypes(), &outTypes_); }
std::string gpuProjectExecutor::Close() { ready_.clear(); return ""; }

OperatorResult gpuProjectExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    while (ready_.empty() && !childDone_) {
        std::vector<std::shared_ptr<Chunk>> in, out;
        while ((int)in.size() < batchChunks_) {
            auto c = std::make_shared<Chunk>();
            OperatorResult r = child_->Execute(nullptr, c.get(), err);
            if (r == InvalidOpResult) return InvalidOpResult;
            if (r == Done) { childDone_ = true; break; }
            if (c->Card() > 0) in.push_back(c);
        }
        std::string e = Evaluate(ctx_, exprs_, child_->OutputTypes(), outTypes_, in, &out);
        if (!e.empty()) { *err = e; return InvalidOpResult; }
        for (auto &c : out) ready_.push_back(c);
    }
    if (ready_.empty()) return Done;
    *output = *ready_.front();
    ready_.pop_front();
    return haveMoreOutput;
}

// ------------------------------------------------------------------ limit

OperatorResult limitExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    // Limit.Sink caps what is collected at offset+limit rows, GetData / HandleOffset skips the
    // offset: rows [offset, offset+limit) of the child, in the child's order
    const uint64_t maxElement = limit_ == UINT64_MAX ? UINT64_MAX : limit_ + offset_;
    for (;;) {
        if (limit_ == 0 || seen_ >= maxElement) return Done;
        Chunk c;
        OperatorResult r = child_->Execute(nullptr, &c, err);
        if (r == InvalidOpResult) return InvalidOpResult;
        if (r == Done) return Done;
        uint64_t card = (uint64_t)c.Card(), start = seen_;
        seen_ += card;
        if (card == 0 || seen_ <= offset_) continue;
        uint64_t from = start < offset_ ? offset_ - start : 0;
        uint64_t to = std::min<uint64_t>(card, maxElement - start);
        if (from == 0 && to == card) { *output = c; return haveMoreOutput; }
        auto sv = std::make_shared<SelectVector>();
        sv->identity = false;
        for (uint64_t i = from; i < to; i++) sv->SelVec.push_back((int64_t)i);
        auto keep = std::make_shared<Chunk>(c);
        output->Init(OutputTypes(), DefaultVectorSize);
        std::vector<int> indice;
        for (int i = 0; i < keep->ColumnCount(); i++) indice.push_back(i);
        output->SliceIndice(*keep, sv, (int)(to - from), 0, indice);
        return haveMoreOutput;
    }
}

// ------------------------------------------------------------------ cross product

std::string crossProductExecutor::Init() {
    outTypes_ = left_->OutputTypes();
    for (auto &t : right_->OutputTypes()) outTypes_.push_back(t);
    return "";
}

OperatorResult crossProductExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (!collected_) {   // CrossProduct.Sink: the whole right side
        for (;;) {
            auto c = std::make_shared<Chunk>();
            OperatorResult r = right_->Execute(nullptr, c.get(), err);
            if (r == InvalidOpResult) return InvalidOpResult;
            if (r == Done) break;
            if (c->Card() > 0) rhs_.push_back(c);
        }
        collected_ = true;
    }
    if (rhs_.empty()) return Done;   // no RHS, empty result (join_cross.go:111-114)
    for (;;) {
        if (!cur_) {
            if (leftDone_) return Done;
            auto c = std::make_shared<Chunk>();
            OperatorResult r = left_->Execute(nullptr, c.get(), err);
            if (r == InvalidOpResult) return InvalidOpResult;
            if (r == Done) { leftDone_ = true; return Done; }
            if (c->Card() == 0) continue;
            cur_ = c;
            rchunk_ = 0;
            rrow_ = 0;
        }
        if (rchunk_ >= rhs_.size()) { cur_.reset(); continue; }   // RHS read over: next LHS chunk
        const Chunk &rc = *rhs_[rchunk_];
        int nl = cur_->ColumnCount();
        output->Init(outTypes_, DefaultVectorSize);
        for (int c = 0; c < nl; c++) output->Data[(size_t)c] = cur_->Data[(size_t)c];              // Reference
        for (int c = 0; c < rc.ColumnCount(); c++) {                                                   // ReferenceInPhyFormatConst
            Vector &v = *output->Data[(size_t)(nl + c)];
            v._PhyFormat = PF_CONST;
            CopyCell(*rc.Data[(size_t)c], rrow_, &v, 0);
        }
        output->SetCard(cur_->Card());
        if (++rrow_ >= rc.Card()) { rrow_ = 0; rchunk_++; }
        return haveMoreOutput;
    }
}

}  // namespace plan

namespace plan {

// ------------------------------------------------------------------ resident plans

static std::string exprType(const ProjExpr &e, const std::vector<LType> &childTypes, const std::vector<const ResidentColumn *> &childSrc,
                            LType *t, const ResidentColumn **src) {
    std::vector<LType> one;
    std::string err = gpuProjectExecutor::Types({e}, childTypes, &one);
    if (!err.empty()) return err;
    if (e.kind == ProjExpr::DecimalQuo) return "DECIMAL-division expressions are not part of a resident plan";
    if (e.kind == ProjExpr::Float32 && e.fprog.size() > 12) return "FLOAT expression program too long";
    *t = e.kind == ProjExpr::Float32 && e.floatTruth ? IntegerType() : one[0];
    *src = e.kind == ProjExpr::ColRef ? childSrc[(size_t)e.col] : nullptr;
    return "";
}

int ResidentPlan::Scan(const ResidentTable *t, std::vector<int> cols, std::vector<Compare> conjuncts, BoolExpr where) {
    Node n;
    n.kind = PH_PN_SCAN;
    n.table = t;
    for (int c : cols) {
        if (c < 0 || c >= (int)t->cols.size()) { if (error.empty()) error = "scan column out of range"; c = 0; }
        n.types.push_back(t->cols[(size_t)c].type);
        n.source.push_back(&t->cols[(size_t)c]);
    }
    for (auto &c : conjuncts) if ((c.col < 0 || c.col >= (int)t->cols.size()) && error.empty()) error = "scan conjunct column out of range";
    n.cols = std::move(cols);
    n.conjuncts = std::move(conjuncts);
    n.where = std::move(where);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
}

int ResidentPlan::Filter(int child, std::vector<Compare> conjuncts, BoolExpr where) {
    Node n;
    n.kind = PH_PN_FILTER;
    n.child[0] = child;
    n.types = nodes[(size_t)child].types;
    n.source = nodes[(size_t)child].source;
    n.conjuncts = std::move(conjuncts);
    n.where = std::move(where);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
}

int ResidentPlan::Join(int probe, int build, std::vector<int> probeKeys, std::vector<int> buildKeys, std::vector<int> out, JoinType type, BoolExpr residual) {
    Node n;
    n.kind = PH_PN_JOIN;
    n.child[0] = probe; n.child[1] = build;
    n.joinType = type;
    n.where = std::move(residual);
    std::vector<LType> all = nodes[(size_t)probe].types;
    std::vector<const ResidentColumn *> src = nodes[(size_t)probe].source;
    for (auto &t : nodes[(size_t)build].types) all.push_back(t);
    for (auto s : nodes[(size_t)build].source) src.push_back(s);
    for (int o : out) {
        if (o < 0 || o >= (int)all.size()) { if (error.empty()) error = "join output column out of range"; o = 0; }
        n.types.push_back(all[(size_t)o]);
        n.source.push_back(src[(size_t)o]);
    }
    n.probeKeys = std::move(probeKeys); n.buildKeys = std::move(buildKeys); n.out = std::move(out);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
}

int ResidentPlan::Project(int child, std::vector<ProjExpr> exprs) {
    Node n;
    n.kind = PH_PN_PROJECT;
    n.child[0] = child;
    for (auto &e : exprs) {
        LType t;
        const ResidentColumn *s = nullptr;
        std::string err = exprType(e, nodes[(size_t)child].types, nodes[(size_t)child].source, &t, &s);
        if (!err.empty() && error.empty()) error = err;
        n.types.push_back(t);
        n.source.push_back(s);
    }
    n.exprs = std::move(exprs);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
}

int ResidentPlan::Agg(int child, std::vector<ProjExpr> groups, std::vector<AggExpr> aggs) {
    Node n;
    n.kind = PH_PN_AGG;
    n.child[0] = child;
    for (auto &e : groups) {
        LType t;
        const ResidentColumn *s = nullptr;
        std::string err = exprType(e, nodes[(size_t)child].types, nodes[(size_t)child].source, &t, &s);
        if (!err.empty() && error.empty()) error = err;
        n.types.push_back(t);
        n.source.push_back(s);
    }
    n.ngroups = (int)groups.size();
    // aggregate results, typed as FinalizeStates types them (function_aggr.go:1330-1365)
    const std::vector<LType> &childTypes = nodes[(size_t)child].types;
    std::vector<ph_col> protos;
    for (auto &t : childTypes) { ph_col pc{}; pc.type = staged_phtype(t); pc.scale = t.Scale; protos.push_back(pc); }
    for (auto &a : aggs) {
        LType at = IntegerType();
        int32_t scale = 0;
        if (a.kind != PH_A_COUNT_STAR) {
            if (a.expr) {
                std::vector<LType> one;
                std::string e = gpuProjectExecutor::Types({*a.expr}, childTypes, &one);
                if (!e.empty()) { if (error.empty()) error = e; one = {IntegerType()}; }
                at = one[0]; scale = at.Scale;
            } else if (a.prog.size() == 1 && a.prog[0].op == PH_X_COL && a.prog[0].col >= 0 && a.prog[0].col < (int)childTypes.size()) {
                at = childTypes[(size_t)a.prog[0].col]; scale = at.Scale;
            } else if (a.prog.empty()) { if (error.empty()) error = "aggregate without an argument"; }
            else {
                if (ph_expr_scale(protos.data(), a.prog.data(), (int32_t)a.prog.size(), &scale) != PH_OK && error.empty()) error = herr("ph_expr_scale");
                at = DecimalType(38, scale);
            }
        }
        n.argTypes.push_back(at);
        const bool dec = at.Id == LTID_DECIMAL;
        switch (a.kind) {
        case PH_A_SUM: n.types.push_back(dec ? DecimalType(38, scale) : HugeintType()); break;      // BindDecimalSum / GetSumAggr
        case PH_A_AVG: n.types.push_back(dec ? DecimalType(38, scale) : DoubleType()); break;       // BindDecimalAvg / GetAvgAggr
        case PH_A_COUNT: case PH_A_COUNT_STAR: case PH_A_COUNT_DISTINCT: n.types.push_back(HugeintType()); break;
        case PH_A_MIN: case PH_A_MAX: n.types.push_back(dec ? DecimalType(at.Width, scale) : at); break;
        default: if (error.empty()) error = "unknown aggregate kind"; n.types.push_back(at);
        }
        n.source.push_back(nullptr);
    }
    n.exprs = std::move(groups);
    n.aggs = std::move(aggs);
    nodes.push_back(std::move(n));
    return (int)nodes.size() - 1;
}

static ph_pred lowerCompare(const Compare &c) {
    ph_pred p{};
    p.col = c.col;
    p.op = c.op;
    switch (c.k.kind) {
    case Literal::Int: p.k.type = PH_I32; p.k.i = c.k.i; break;
    case Literal::Float: p.k.type = PH_F32; p.k.f = c.k.f; break;
    case Literal::DateDays: p.k.type = PH_DATE; p.k.i = c.k.i; break;
    case Literal::Dec: p.k.type = PH_DEC64; p.k.i = c.k.i; p.k.scale = c.k.scale; break;
    case Literal::Str: p.k.type = PH_STR; p.k.s = c.k.s.c_str(); break;   // the plan outlives the call that copies it
    }
    return p;
}

static ph_const lowerLiteral(const Literal &k) {
    ph_const c{};
    switch (k.kind) {
    case Literal::Int: c.type = PH_I32; c.i = k.i; break;
    case Literal::Float: c.type = PH_F32; c.f = k.f; break;
    case Literal::DateDays: c.type = PH_DATE; c.i = k.i; break;
    case Literal::Dec: c.type = PH_DEC64; c.i = k.i; c.scale = k.scale; break;
    case Literal::Str: c.type = PH_STR; c.s = k.s.c_str(); break;
    }
    return c;
}

// BoolExpr tree -> the flat ph_bool array (node 0 = root, the children of a node contiguous)
static void flattenBool(const BoolExpr &b, size_t at, std::vector<ph_bool> *out) {
    ph_bool n{};
    if (b.kind == BoolExpr::Cmp) {
        n.kind = PH_B_CMP; n.col = b.col; n.op = b.op;
        if (b.col2 >= 0) { n.k.type = PH_COLREF; n.k.i = b.col2; } else n.k = lowerLiteral(b.k);
        (*out)[at] = n;
        return;
    }
    n.kind = b.kind == BoolExpr::And ? PH_B_AND : PH_B_OR;
    n.first_child = (int32_t)out->size();
    n.nchildren = (int32_t)b.children.size();
    (*out)[at] = n;
    const size_t base = out->size();
    out->resize(base + b.children.size());
    for (size_t c = 0; c < b.children.size(); c++) flattenBool(b.children[c], base + c, out);
}

static std::vector<ph_bool> lowerBool(const BoolExpr &b) {
    std::vector<ph_bool> out;
    if (b.empty()) return out;
    out.resize(1);
    flattenBool(b, 0, &out);
    return out;
}

static ph_plan_expr lowerExpr(const ProjExpr &e, std::vector<std::vector<ph_bool>> *whens) {
    ph_plan_expr x{};
    switch (e.kind) {
    case ProjExpr::Case:
        x.kind = PH_PE_CASE; x.col = -1;
        x.nprog = (int32_t)std::min<size_t>(e.prog.size(), 12);
        for (int i = 0; i < x.nprog; i++) x.prog[i] = e.prog[(size_t)i];
        x.nelse = (int32_t)std::min<size_t>(e.elseProg.size(), 12);
        for (int i = 0; i < x.nelse; i++) x.else_prog[i] = e.elseProg[(size_t)i];
        x.result_int = e.resultInt ? 1 : 0;
        whens->push_back(lowerBool(*e.when));
        x.nwhen = (int32_t)whens->back().size();
        x.when = whens->back().data();
        break;
    case ProjExpr::ColRef: x.kind = PH_PE_COL; x.col = e.col; break;
    case ProjExpr::ExtractYear: x.kind = PH_PE_YEAR; x.col = e.col; break;
    case ProjExpr::Substring: x.kind = PH_PE_SUBSTR; x.col = e.col; x.sub_offset = e.offset; x.sub_length = e.length; break;
    case ProjExpr::Float32: {
        x.kind = PH_PE_FLOAT; x.col = -1; x.result_int = e.floatTruth ? 1 : 0; x.float_wide = e.floatWide ? 1 : 0;
        x.nprog = (int32_t)std::min<size_t>(e.fprog.size(), 12);
        for (int i = 0; i < x.nprog; i++) {
            const FloatOp &o = e.fprog[(size_t)i];
            ph_rpn r{};
            switch (o.op) {
            case FloatOp::Col: r.op = PH_X_COL; r.col = o.col; break;
            case FloatOp::Const: { r.op = PH_X_CONST; uint32_t bits; memcpy(&bits, &o.k, 4); r.ival = bits; break; }
            case FloatOp::Add: r.op = PH_X_ADD; break;
            case FloatOp::Sub: r.op = PH_X_SUB; break;
            case FloatOp::Mul: r.op = PH_X_MUL; break;
            case FloatOp::Div: r.op = PH_X_DIV; break;
            case FloatOp::Lt: r.op = PH_X_LT; break;
            case FloatOp::Le: r.op = PH_X_LE; break;
            case FloatOp::Gt: r.op = PH_X_GT; break;
            case FloatOp::Ge: r.op = PH_X_GE; break;
            }
            x.prog[i] = r;
        }
        break;
    }
    default:
        x.kind = PH_PE_DECIMAL; x.col = -1; x.nprog = (int32_t)std::min<size_t>(e.prog.size(), 12);
        for (int i = 0; i < x.nprog; i++) x.prog[i] = e.prog[(size_t)i];
    }
    return x;
}

std::string gpuResidentPlanExecutor::Init() {
    if (!rp_.error.empty()) return rp_.error;
    if (rp_.nodes.empty()) return "empty resident plan";
    rowsRoot_ = rp_.nodes.back().kind != PH_PN_AGG;   // a join / filter / project root: the plan returns its rows
    if (rowsRoot_ && (!having_.empty() || !outputs_.empty() || topkAgg_ >= 0)) return "HAVING / output expressions / top-k belong to an aggregate root";
    // the descriptor arrays live until ph_plan_create has copied them
    size_t nn = rp_.nodes.size();
    std::vector<ph_plan_node> desc(nn);
    std::vector<std::vector<int32_t>> i32s;
    std::vector<std::vector<ph_pred>> preds(nn);
    std::vector<std::vector<ph_plan_expr>> exprs(nn);
    std::vector<std::vector<ph_plan_agg>> aggs(nn);
    std::vector<std::vector<ph_bool>> bools(nn), whens;
    whens.reserve(256);   // the WHEN arrays must not move while the descriptor points at them
    i32s.reserve(nn * 4);
    auto keep = [&](const std::vector<int> &v) { i32s.emplace_back(v.begin(), v.end()); if (i32s.back().empty()) i32s.back().push_back(0); return i32s.back().data(); };
    const ResidentPlan::Node &root = rp_.nodes.back();
    for (size_t i = 0; i < nn; i++) {
        const ResidentPlan::Node &n = rp_.nodes[i];
        ph_plan_node &d = desc[i];
        d = ph_plan_node{};
        d.kind = n.kind;
        d.child[0] = n.child[0]; d.child[1] = n.child[1];
        for (auto &c : n.conjuncts) preds[i].push_back(lowerCompare(c));
        d.npreds = (int32_t)preds[i].size();
        d.preds = preds[i].data();
        bools[i] = lowerBool(n.where);
        d.nbools = (int32_t)bools[i].size();
        d.bools = bools[i].data();
        switch (n.kind) {
        case PH_PN_SCAN:
            d.table = n.table->table;
            d.ncols = (int32_t)n.cols.size();
            d.cols = keep(n.cols);
            break;
        case PH_PN_JOIN:
            if (n.probeKeys.size() != n.buildKeys.size() || n.probeKeys.empty()) return "join needs matching key lists";
            d.join_type = n.joinType == JoinSemi ? PH_JT_SEMI : n.joinType == JoinAnti ? PH_JT_ANTI : n.joinType == JoinLeft ? PH_JT_LEFT : PH_JT_INNER;
            d.nkeys = (int32_t)n.probeKeys.size();
            d.probe_keys = keep(n.probeKeys);
            d.build_keys = keep(n.buildKeys);
            d.nout = (int32_t)n.out.size();
            d.out = keep(n.out);
            break;
        case PH_PN_PROJECT:
            for (auto &e : n.exprs) { if (e.prog.size() > 12 || e.elseProg.size() > 12) return "expression program too long"; exprs[i].push_back(lowerExpr(e, &whens)); }
            d.nexprs = (int32_t)exprs[i].size();
            d.exprs = exprs[i].data();
            break;
        case PH_PN_AGG:
            for (auto &e : n.exprs) { if (e.prog.size() > 12 || e.elseProg.size() > 12) return "expression program too long"; exprs[i].push_back(lowerExpr(e, &whens)); }
            for (auto &a : n.aggs) {
                ph_plan_agg pa{};
                pa.kind = a.kind;
                if (a.kind != PH_A_COUNT_STAR) {
                    if (a.expr) {
                        if (a.expr->prog.size() > 12 || a.expr->elseProg.size() > 12) return "aggregate argument program too long";
                        pa.arg = lowerExpr(*a.expr, &whens);
                    } else {
                        if (a.prog.empty()) return "aggregate without an argument";
                        if (a.prog.size() > 12) return "aggregate argument program too long";
                        ProjExpr e = a.prog.size() == 1 && a.prog[0].op == PH_X_COL ? ProjExpr::Col(a.prog[0].col) : ProjExpr::Dec(a.prog);
                        pa.arg = lowerExpr(e, &whens);
                    }
                }
                aggs[i].push_back(pa);
            }
            d.ngroups = (int32_t)exprs[i].size();
            d.groups = exprs[i].data();
            d.naggs = (int32_t)aggs[i].size();
            d.aggs = aggs[i].data();
            break;
        default: break;
        }
    }
    outTypes_ = root.types;       // [group columns | aggregate results], typed by ResidentPlan::Agg
    argType_ = root.argTypes;
    if (ph_plan_create(ctx_, desc.data(), (int32_t)nn, &plan_) != PH_OK) return herr("ph_plan_create");
    if (comm_ && ph_plan_set_comm(plan_, comm_) != PH_OK) return herr("ph_plan_set_comm");
    // HAVING runs in the aggregate's output phase, before Order and Limit (executor_aggr.go:143-263): with a HAVING the top-k
    // preselection is not announced — the k best groups could fail it while later ones pass — and the Order above sorts the survivors
    if (topkAgg_ >= 0 && having_.empty() && ph_plan_set_topk(plan_, topkAgg_, topkDesc_ ? 1 : 0, topkK_) != PH_OK) return herr("ph_plan_set_topk");
    if (rowsRoot_ && rowsTopkCol_ >= 0 && rowsTopkK_ > 0 && ph_plan_set_rows_topk(plan_, rowsTopkCol_, rowsTopkDesc_ ? 1 : 0, rowsTopkK_) != PH_OK) return herr("ph_plan_set_rows_topk");
    for (auto &h : having_) if (h.col < 0 || h.col >= (int)outTypes_.size()) return "HAVING column out of range";
    if (!having_.empty()) {   // numeric conjuncts over aggregate columns: filtered on the device, only the survivors are fetched
        std::vector<ph_pred> hp;
        bool numeric = true;
        for (auto &h : having_) { numeric = numeric && h.k.kind != Literal::Str && h.k.kind != Literal::DateDays; hp.push_back(lowerCompare(h)); }
        havingOnDevice_ = numeric && ph_plan_set_having(plan_, (int32_t)hp.size(), hp.data()) == PH_OK;
    }
    if (!outputs_.empty()) { std::string e = gpuProjectExecutor::Types(outputs_, outTypes_, &finalTypes_); if (!e.empty()) return e; }
    return "";
}

std::string gpuResidentPlanExecutor::Close() {
    if (plan_) { ph_plan_free(plan_); plan_ = nullptr; }
    results_.clear();
    return "";
}

OperatorResult gpuResidentPlanExecutor::Execute(Chunk *, Chunk *output, std::string *err) {
    if (!built_ && rowsRoot_) {
        ph_rows_result *r = nullptr;
        static const bool timing_rows = getenv("PH_HOST_TIMING") != nullptr;
        auto now_r = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tr0 = now_r();
        int prc = ph_plan_run(plan_);
        const double tr1 = now_r();
        if (prc == PH_OK) prc = ph_plan_fetch_rows(plan_, &r);
        if (timing_rows) fprintf(stderr, "  resident plan (rows): ph_plan_run %.1f us, ph_plan_fetch_rows %.1f us\n", (tr1 - tr0) * 1e6, (now_r() - tr1) * 1e6);
        if (prc != PH_OK) { *err = herr("ph_plan_run/fetch_rows"); return InvalidOpResult; }
        const ResidentPlan::Node &root = rp_.nodes.back();
        if (r->ncols != (int32_t)outTypes_.size()) { ph_rows_result_free(r); *err = "row plan: column count differs from the plan's typing"; return InvalidOpResult; }
        for (int64_t base = 0; base < r->nrows; base += DefaultVectorSize) {
            const int card = (int)std::min<int64_t>(DefaultVectorSize, r->nrows - base);
            auto c = std::make_shared<Chunk>();
            c->Init(outTypes_, DefaultVectorSize);
            for (int col = 0; col < r->ncols; col++) {
                Vector &v = *c->Data[(size_t)col];
                for (int i = 0; i < card; i++) {
                    const int64_t row = base + i;
                    if (r->type[col] == PH_STR) { v.SetString(i, r->bytes[col] + r->offsets[col][row], r->offsets[col][row + 1] - r->offsets[col][row]); continue; }
                    const int64_t x = r->values[col][row];
                    switch (v._Typ.GetInternalType()) {
                    case PT_INT32: v.Slice<int32_t>()[i] = (int32_t)x; break;
                    case PT_INT64: v.Slice<int64_t>()[i] = x; break;
                    case PT_DATE: v.Slice<Date>()[i] = DateFromDays((int32_t)x); break;
                    case PT_DECIMAL: v.Slice<Decimal>()[i] = DecimalFromUnscaled(x, r->scale[col]); break;
                    case PT_INT128: v.Slice<Hugeint>()[i] = Hugeint{(uint64_t)x, x < 0 ? -1 : 0}; break;
                    case PT_VARCHAR: {   // a dictionary-code column of a table: the code's string
                        const ResidentColumn *src = root.source[(size_t)col];
                        if (!src || x < 0 || (size_t)x >= src->dict.size()) { ph_rows_result_free(r); *err = "row plan: dictionary code without its dictionary"; return InvalidOpResult; }
                        v.SetString(i, src->dict[(size_t)x].data(), (int64_t)src->dict[(size_t)x].size());
                        break;
                    }
                    default: ph_rows_result_free(r); *err = "row plan: unsupported column type"; return InvalidOpResult;
                    }
                }
            }
            c->SetCard(card);
            results_.push_back(c);
        }
        ph_rows_result_free(r);
        if (timing_rows) fprintf(stderr, "  resident plan (rows): %zu chunks built in %.1f us\n", results_.size(), (now_r() - tr1) * 1e6);
        built_ = true;
    }
    if (!built_) {
        ph_agg_result *r = nullptr;
        static const bool timing = getenv("PH_HOST_TIMING") != nullptr;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        int prc = ph_plan_run(plan_);
        const double t1 = now();
        if (prc == PH_OK) prc = ph_plan_fetch(plan_, &r);
        if (timing) fprintf(stderr, "  resident plan: ph_plan_run %.1f us, ph_plan_fetch %.1f us\n", (t1 - t0) * 1e6, (now() - t1) * 1e6);
        if (prc != PH_OK) {
            *err = herr("ph_plan_run/fetch");
            return InvalidOpResult;
        }
        const double t2 = now();
        // (a sum beyond int64 makes the device hand every group back unfiltered: the HAVING is then applied below, like any other)
        const bool havingDone = havingOnDevice_ && ph_plan_having_applied(plan_) != 0;
        const ResidentPlan::Node &root = rp_.nodes.back();
        std::vector<LType> keyTypes(root.types.begin(), root.types.begin() + root.ngroups);
        std::vector<const std::vector<std::string> *> dicts;
        std::vector<std::vector<std::string>> fetched(keyTypes.size());
        const int nkw = std::max<int>((int)keyTypes.size(), 1);
        for (size_t k = 0; k < keyTypes.size(); k++) {
            dicts.push_back(root.source[k] ? &root.source[k]->dict : nullptr);
            int32_t kt = 0, ks = 0, kc = -1;
            const ph_table *tab = nullptr;
            if (ph_plan_key_info(plan_, (int32_t)k, &kt, &ks, &tab, &kc) != PH_OK) { ph_agg_result_free(r); *err = herr("ph_plan_key_info"); return InvalidOpResult; }
            if (kt != PH_STR) continue;
            // a VARCHAR key that is no small dictionary came back as ROWS of its column: fetch those strings, one per group, and let
            // the group's key value index them
            std::vector<int64_t> rows((size_t)r->ngroups);
            for (int64_t g = 0; g < r->ngroups; g++) rows[(size_t)g] = r->keys[g * nkw + (int64_t)k];
            std::vector<int32_t> off((size_t)r->ngroups + 1);
            std::vector<char> bytes((size_t)1 << 20);
            if (ph_table_strings(ctx_, tab, kc, rows.data(), r->ngroups, off.data(), bytes.data(), (int64_t)bytes.size()) != PH_OK) {
                ph_agg_result_free(r);
                *err = herr("ph_table_strings");
                return InvalidOpResult;
            }
            for (int64_t g = 0; g < r->ngroups; g++) {
                fetched[k].emplace_back(bytes.data() + off[(size_t)g], (size_t)(off[(size_t)g + 1] - off[(size_t)g]));
                r->keys[g * nkw + (int64_t)k] = g;
            }
            dicts[k] = &fetched[k];
        }
        std::vector<int> kinds, scales;
        for (size_t i = 0; i < root.aggs.size(); i++) { kinds.push_back(root.aggs[i].kind); scales.push_back(r->scale[i]); }
        std::string e = BuildAggOutput(outTypes_, keyTypes, dicts, kinds, argType_, scales, r->ngroups, r->keys, r->key_null, r->sum_lo, r->sum_hi,
                                       r->count, &results_);
        ph_agg_result_free(r);
        if (e.empty()) e = ApplyAggOutputPhase(ctx_, havingDone ? std::vector<Compare>{} : having_, outputs_, outTypes_, finalTypes_, &results_);
        if (!e.empty()) { *err = e; return InvalidOpResult; }
        if (timing) fprintf(stderr, "  resident plan: %lld groups into chunks %.1f us\n", (long long)results_.size(), (now() - t2) * 1e6);
        built_ = true;
    }
    if (next_ >= results_.size()) return Done;
    *output = *results_[next_++];
    return haveMoreOutput;
}

}  // namespace plan

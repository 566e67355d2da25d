// ph_hash (Chunk.Hash parity) and ph_expr_eval (decimal/integer expression programs).
#include <algorithm>

#include <sstream>

#include "common.h"
#include "scan_jit.h"
#include "device_util.h"
#include "ops.h"

namespace ph {

// ------------------------------------------------------------------ hash
// Chunk.Hash = HashTypeSwitch on the first key column, CombineHashTypeSwitch on the rest
// (reference pkg/chunk/chunk.go:160-166, hash.go:182-413). Values are bit-identical to the
// reference so that a mixed CPU/GPU plan could share hash columns; the device tables themselves
// use their own mixer for placement.

struct HashCol {
    int type, scale;
    const void *data;
    const uint8_t *validity;
    const uint8_t *bytes;          // PH_STR
    const uint64_t *dict_hashes;   // PH_CODE8
};

struct HashParams {
    int ncols;
    HashCol c[4];
};

__device__ __forceinline__ uint64_t hash_one(const HashCol &c, int64_t r) {
    if (!bit_valid(c.validity, r)) return NULL_HASH;
    switch (c.type) {
    case PH_I32:  // HashFuncInt32: uint32 zero-extended (hash.go:53-55)
        return murmurhash64((uint64_t)(uint32_t)((const int32_t *)c.data)[r]);
    case PH_I64:
        return murmurhash64((uint64_t)((const int64_t *)c.data)[r]);
    case PH_DATE: {  // h(Y) ^ h(M) ^ h(D) (hash.go:127-129)
        int32_t y, m, d;
        civil_from_days(((const int32_t *)c.data)[r], &y, &m, &d);
        return murmurhash64((uint64_t)(int64_t)y) ^ murmurhash64((uint64_t)(int64_t)m) ^
               murmurhash64((uint64_t)(int64_t)d);
    }
    case PH_DEC64: {
        // h(neg) ^ h(coef) ^ h(scale) (hash.go:144-151) of the value as the loader built it:
        // NewFromInt64 drops the fraction's trailing zeros (pkg/chunk/vector.go:257-264)
        long long v = ((const int64_t *)c.data)[r];
        uint64_t coef = v < 0 ? (uint64_t)0 - (uint64_t)v : (uint64_t)v;
        int s = c.scale;
        while (s > 0 && coef % 10 == 0) {
            coef /= 10;
            s--;
        }
        return murmurhash64(v < 0 ? 1 : 0) ^ murmurhash64(coef) ^ murmurhash64((uint64_t)s);
    }
    case PH_CODE8:
        return c.dict_hashes[((const uint8_t *)c.data)[r]];
    case PH_STR: {
        const int32_t *off = (const int32_t *)c.data;
        return hash_bytes(c.bytes + off[r], (uint64_t)(off[r + 1] - off[r]));
    }
    default:
        return 0;
    }
}

__global__ __launch_bounds__(256) void hash_kernel(HashParams P, int64_t n, uint64_t *__restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256) {
        uint64_t h = hash_one(P.c[0], r);
        for (int k = 1; k < P.ncols; k++) h = combine_hash(h, hash_one(P.c[k], r));
        out[r] = h;
    }
}

// ------------------------------------------------------------------ expression programs

constexpr int X_MAX_OPS = 24;
constexpr int X_MAX_COLS = 32;
constexpr int X_STACK = 8;

struct XInstr {
    int op;      // PH_X_*
    int col;
    long long k; // constant, already scaled
    long long ma, mb;  // ADD/SUB: multipliers aligning the two operands' scales
};

struct XParams {
    int nops, ncols;
    XInstr ins[X_MAX_OPS];
    struct { int type; const void *data; const uint8_t *validity; } c[X_MAX_COLS];
};

__global__ __launch_bounds__(256) void expr_kernel(XParams P, const int32_t *__restrict__ sel, int64_t n,
                                                   long long *__restrict__ out, uint8_t *__restrict__ out_valid,
                                                   int *__restrict__ overflow_flag) {
    // operand stack in LDS: one column per thread, so dynamic stack indexing costs no registers
    __shared__ long long stk[X_STACK][256];
    const int t = threadIdx.x;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < n; base += (int64_t)gridDim.x * 256) {
        int64_t i = base + t;
        bool live = i < n;
        int64_t r = live ? (sel ? sel[i] : i) : 0;
        bool null = false, ovf = false;
        int sp = 0;
        for (int p = 0; p < P.nops; p++) {
            const XInstr &o = P.ins[p];
            switch (o.op) {
            case PH_X_COL: {
                long long v = 0;
                if (live) {
                    if (!bit_valid(P.c[o.col].validity, r)) null = true;
                    else if (P.c[o.col].type == PH_I32) v = ((const int32_t *)P.c[o.col].data)[r];
                    else v = ((const int64_t *)P.c[o.col].data)[r];
                }
                stk[sp++][t] = v;
                break;
            }
            case PH_X_CONST: stk[sp++][t] = o.k; break;
            case PH_X_ADD: case PH_X_SUB: {
                long long b = stk[--sp][t], a = stk[sp - 1][t], x, y, z;
                ovf |= __builtin_mul_overflow(a, o.ma, &x);
                ovf |= __builtin_mul_overflow(b, o.mb, &y);
                ovf |= o.op == PH_X_ADD ? __builtin_add_overflow(x, y, &z) : __builtin_sub_overflow(x, y, &z);
                stk[sp - 1][t] = z;
                break;
            }
            case PH_X_MUL: {
                long long b = stk[--sp][t], a = stk[sp - 1][t], z;
                ovf |= __builtin_mul_overflow(a, b, &z);
                stk[sp - 1][t] = z;
                break;
            }
            default: break;
            }
        }
        if (live) {
            out[i] = null ? 0 : stk[0][t];
            if (ovf && !null) atomicOr(overflow_flag, 1);
        }
        if (out_valid) {  // one validity byte per 8 rows: lanes 0,8,16.. assemble it via ballot
            unsigned long long m = __ballot(live && !null);
            int lane = t & 63;
            if ((lane & 7) == 0 && base + (t & ~7) < n) out_valid[(base + t) >> 3] = (uint8_t)(m >> lane);
        }
    }
}

static bool pow10ll(int k, long long *out) {
    long long r = 1;
    for (int i = 0; i < k; i++)
        if (__builtin_mul_overflow(r, 10ll, &r)) return false;
    *out = r;
    return true;
}

// scale bookkeeping of the binder's rules: Mul adds scales, Add/Sub takes the max
// (BindDecimalMultiply function_scalar.go:429-475, BindDecimalAddSubstract :37-84)
static int compile_expr(const ph_col *cols, int32_t ncols, const ph_rpn *prog, int32_t nprog, XParams *X,
                        int32_t *result_scale) {
    if (nprog <= 0 || nprog > X_MAX_OPS) { set_error("expression program of %d ops (max %d)", nprog, X_MAX_OPS); return PH_EUNSUPPORTED; }
    int scales[X_STACK];
    int sp = 0;
    if (X) { X->nops = nprog; X->ncols = ncols; }
    for (int32_t p = 0; p < nprog; p++) {
        const ph_rpn &o = prog[p];
        XInstr ins{};
        ins.op = o.op;
        ins.col = o.col;
        ins.ma = ins.mb = 1;
        switch (o.op) {
        case PH_X_COL:
            if (o.col < 0 || o.col >= ncols || sp >= X_STACK) return PH_EUNSUPPORTED;
            if (cols[o.col].type == PH_DEC64) scales[sp++] = cols[o.col].scale;
            else if (cols[o.col].type == PH_I32 || cols[o.col].type == PH_I64) scales[sp++] = 0;
            else { set_error("expression over column type %d", cols[o.col].type); return PH_EUNSUPPORTED; }
            break;
        case PH_X_CONST:
            if (sp >= X_STACK) return PH_EUNSUPPORTED;
            ins.k = o.ival;
            scales[sp++] = o.scale;
            break;
        case PH_X_ADD: case PH_X_SUB: {
            if (sp < 2) return PH_EUNSUPPORTED;
            int sb = scales[--sp], sa = scales[sp - 1], s = std::max(sa, sb);
            if (!pow10ll(s - sa, &ins.ma) || !pow10ll(s - sb, &ins.mb)) return PH_EOVERFLOW;
            scales[sp - 1] = s;
            break;
        }
        case PH_X_MUL:
            if (sp < 2) return PH_EUNSUPPORTED;
            sp--;
            scales[sp - 1] += scales[sp];
            break;
        default:
            set_error("unknown expression op %d", o.op);
            return PH_EUNSUPPORTED;
        }
        if (X) X->ins[p] = ins;
    }
    if (sp != 1) { set_error("expression program leaves %d values on the stack", sp); return PH_EUNSUPPORTED; }
    if (scales[0] > 18) { set_error("result scale %d exceeds the int64 decimal domain", scales[0]); return PH_EOVERFLOW; }
    *result_scale = scales[0];
    return PH_OK;
}

}  // namespace ph

extern "C" uint64_t ph_hash_bytes(const void *p, uint64_t len) { return ph::hash_bytes((const uint8_t *)p, len); }

extern "C" int ph_hash(ph_ctx *ctx, const ph_col *cols, const uint64_t *const *dict_hashes, int32_t ncols,
                       int64_t n, uint64_t *out_dev) {
    PH_REQUIRE(ctx && cols && ncols >= 1 && ncols <= 4 && n >= 0, "ph_hash: bad arguments (1..4 key columns)");
    if (n == 0) return PH_OK;
    ph::HashParams P{};
    P.ncols = ncols;
    for (int c = 0; c < ncols; c++) {
        P.c[c].type = cols[c].type;
        P.c[c].scale = cols[c].scale;
        P.c[c].data = cols[c].data;
        P.c[c].validity = cols[c].validity;
        P.c[c].bytes = (const uint8_t *)cols[c].aux;
        P.c[c].dict_hashes = dict_hashes ? dict_hashes[c] : nullptr;
        if (cols[c].type == PH_CODE8 && !P.c[c].dict_hashes) { ph::set_error("ph_hash: column %d is PH_CODE8 but has no dict_hashes", c); return PH_EINVAL; }
        if (cols[c].type == PH_F32 || cols[c].type == PH_F64) { ph::set_error("ph_hash: no hash for float columns (the reference has none either)"); return PH_EUNSUPPORTED; }
    }
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 16);
    ph::hash_kernel<<<grid, 256, 0, ctx->stream>>>(P, n, out_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

extern "C" int ph_expr_scale(const ph_col *cols, const ph_rpn *prog, int32_t nprog, int32_t *scale) {
    PH_REQUIRE(cols && prog && scale, "ph_expr_scale: bad arguments");
    int32_t maxcol = 0;
    for (int32_t p = 0; p < nprog; p++) if (prog[p].op == PH_X_COL) maxcol = std::max(maxcol, prog[p].col + 1);
    return ph::compile_expr(cols, maxcol, prog, nprog, nullptr, scale);
}

// ---- plan-specialised expression kernel (hiprtc): the RPN unrolled into straight-line int64
// arithmetic with the same overflow checks, 4 rows per thread with all column reads issued first.
// The interpreter above keeps its operand stack in LDS and switches per operation and row: 70 us for
// 3.3 M rows of Q9's profit expression, where the bytes moved would take 20. Constants are kernel
// parameters, so a module is cached per expression SHAPE.
namespace {

struct ExprJitParams {
    const void *c[ph::X_MAX_COLS];
    const uint8_t *v[ph::X_MAX_COLS];
    const int32_t *sel;
    long long n;
    long long *out;
    uint8_t *out_valid;
    int *flag;
    long long k[ph::X_MAX_OPS * 2];
};

std::string expr_jit_key(const ph::XParams &X, bool has_sel, bool out_valid) {
    std::string k = "expr:";
    k += has_sel ? 's' : '-';
    k += out_valid ? 'v' : '-';
    for (int c = 0; c < X.ncols; c++) { k += '|'; k += std::to_string(X.c[c].type); if (X.c[c].validity) k += 'n'; }
    for (int p = 0; p < X.nops; p++) { k += ','; k += std::to_string(X.ins[p].op); k += ':'; k += std::to_string(X.ins[p].col); }
    return k;
}

std::string expr_jit_source(const ph::XParams &X, bool has_sel, bool out_valid, std::string *key) {
    std::ostringstream o;
    *key = expr_jit_key(X, has_sel, out_valid);
    constexpr int U = 4;
    o << "typedef long long i64;\n"
      << "struct EP { const void *c[" << ph::X_MAX_COLS << "]; const unsigned char *v[" << ph::X_MAX_COLS << "]; const int *sel; i64 n; i64 *out; "
      << "unsigned char *out_valid; int *flag; i64 k[" << ph::X_MAX_OPS * 2 << "]; };\n"
      << "__device__ __forceinline__ bool bitv(const unsigned char *m, i64 i) { return (m[i >> 3] >> (i & 7)) & 1; }\n"
      << "extern \"C\" __global__ __launch_bounds__(256) void expr_jit(EP p) {\n"
      << "  const i64 step = (i64)gridDim.x * 256 * " << U << ";\n"
      << "  for (i64 base = (i64)blockIdx.x * 256 * " << U << "; base < p.n; base += step) {\n";
    for (int u = 0; u < U; u++) {
        o << "    const i64 i" << u << " = base + " << u << " * 256 + threadIdx.x; const bool live" << u << " = i" << u << " < p.n;\n"
          << "    const i64 r" << u << " = live" << u << " ? " << (has_sel ? "(i64)p.sel[i" + std::to_string(u) + "]" : "i" + std::to_string(u)) << " : 0;\n";
    }
    // column reads of all rows first (each distinct column once)
    bool used[ph::X_MAX_COLS] = {};
    for (int p = 0; p < X.nops; p++) if (X.ins[p].op == PH_X_COL) used[X.ins[p].col] = true;
    for (int c = 0; c < X.ncols; c++) {
        if (!used[c]) continue;
        for (int u = 0; u < U; u++) {
            if (X.c[c].type == PH_I32) o << "    const i64 c" << c << "_" << u << " = ((const int *)p.c[" << c << "])[r" << u << "];\n";
            else o << "    const i64 c" << c << "_" << u << " = ((const i64 *)p.c[" << c << "])[r" << u << "];\n";
        }
    }
    for (int u = 0; u < U; u++) {
        o << "    bool null" << u << " = false, ovf" << u << " = false;\n";
        for (int c = 0; c < X.ncols; c++)
            if (used[c] && X.c[c].validity) o << "    null" << u << " = null" << u << " || !bitv(p.v[" << c << "], r" << u << ");\n";
        // the RPN as SSA values t<n>
        std::vector<std::string> st;
        int tn = 0;
        for (int p = 0; p < X.nops; p++) {
            const ph::XInstr &in = X.ins[p];
            std::string name = "t" + std::to_string(u) + "_" + std::to_string(tn++);
            if (in.op == PH_X_COL) { o << "    const i64 " << name << " = c" << in.col << "_" << u << ";\n"; st.push_back(name); }
            else if (in.op == PH_X_CONST) { o << "    const i64 " << name << " = p.k[" << 2 * p << "];\n"; st.push_back(name); }
            else if (in.op == PH_X_ADD || in.op == PH_X_SUB) {
                std::string b = st.back(); st.pop_back();
                std::string a = st.back(); st.pop_back();
                o << "    i64 " << name << "x, " << name << "y, " << name << ";\n"
                  << "    ovf" << u << " |= __builtin_mul_overflow(" << a << ", p.k[" << 2 * p << "], &" << name << "x);\n"
                  << "    ovf" << u << " |= __builtin_mul_overflow(" << b << ", p.k[" << 2 * p + 1 << "], &" << name << "y);\n"
                  << "    ovf" << u << " |= " << (in.op == PH_X_ADD ? "__builtin_add_overflow(" : "__builtin_sub_overflow(") << name << "x, " << name << "y, &" << name << ");\n";
                st.push_back(name);
            } else {   // PH_X_MUL
                std::string b = st.back(); st.pop_back();
                std::string a = st.back(); st.pop_back();
                o << "    i64 " << name << ";\n    ovf" << u << " |= __builtin_mul_overflow(" << a << ", " << b << ", &" << name << ");\n";
                st.push_back(name);
            }
        }
        o << "    if (live" << u << ") { p.out[i" << u << "] = null" << u << " ? 0 : " << st.back() << "; if (ovf" << u << " && !null" << u << ") atomicOr(p.flag, 1); }\n";
        if (out_valid)   // one validity byte per 8 rows: lanes 0, 8, 16, ... assemble it from the ballot
            o << "    { const unsigned long long m = __ballot(live" << u << " && !null" << u << "); const int lane = threadIdx.x & 63;\n"
              << "      if ((lane & 7) == 0 && (i" << u << " & ~7ll) < p.n) p.out_valid[i" << u << " >> 3] = (unsigned char)(m >> lane); }\n";
    }
    o << "  }\n}\n";
    return o.str();
}

int expr_jit_run(ph_ctx *ctx, const ph::XParams &X, const ph_col *cols, const int32_t *sel, int64_t n, long long *out,
                 uint8_t *out_valid, int *flag) {
    const char *e = getenv("PH_EXPR_JIT");
    if (e && atoi(e) == 0) return PH_EUNSUPPORTED;
    std::string key = expr_jit_key(X, sel != nullptr, out_valid != nullptr);
    ph::JitKernel kn;
    if (!ph::jit_cached(ctx, key, &kn)) {   // the source is only generated on a miss
        std::string src = expr_jit_source(X, sel != nullptr, out_valid != nullptr, &key);
        if (ph::jit_module(ctx, key, src, "expr_jit", &kn) != PH_OK) return PH_EUNSUPPORTED;
    }
    ExprJitParams P{};
    for (int c = 0; c < X.ncols; c++) { P.c[c] = cols[c].data; P.v[c] = cols[c].validity; }
    P.sel = sel; P.n = n; P.out = out; P.out_valid = out_valid; P.flag = flag;
    for (int p = 0; p < X.nops; p++) {
        if (X.ins[p].op == PH_X_CONST) P.k[2 * p] = X.ins[p].k;
        else { P.k[2 * p] = X.ins[p].ma; P.k[2 * p + 1] = X.ins[p].mb; }
    }
    size_t size = sizeof P;
    void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &P, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const int grid = (int)std::min<int64_t>((n + 1023) / 1024, (int64_t)ctx->cu_count * 8);
    PH_HIP(hipModuleLaunchKernel(kn.fn, (unsigned)grid, 1, 1, 256, 1, 1, 0, ctx->stream, nullptr, config));
    return PH_OK;
}

}  // namespace

// Build check without a device: Q9's two expressions (with a selection; positional with three
// operand types) and a NULL-able one compile for gfx950.
extern "C" int ph_expr_jit_selfcheck(int32_t which) {
    ph_col cols[3] = {};
    std::vector<ph_rpn> prog;
    auto col = [](int c) { return ph_rpn{PH_X_COL, c, 0, 0}; };
    auto op = [](int o) { return ph_rpn{o, -1, 0, 0}; };
    static const uint8_t dummy = 0xff;
    bool sel = false, outv = false;
    if (which == 0) {        // ext * (1 - disc), rows through a selection
        cols[0].type = PH_DEC64; cols[0].scale = 2; cols[1].type = PH_DEC64; cols[1].scale = 2;
        prog = {col(0), ph_rpn{PH_X_CONST, -1, 1, 0}, col(1), op(PH_X_SUB), op(PH_X_MUL)};
        sel = true;
    } else if (which == 1) { // rev - cost * qty, positional
        cols[0].type = PH_DEC64; cols[0].scale = 4; cols[1].type = PH_DEC64; cols[1].scale = 2; cols[2].type = PH_I32;
        prog = {col(0), col(1), col(2), op(PH_X_MUL), op(PH_X_SUB)};
    } else if (which == 2) { // NULL-able operands, validity out
        cols[0].type = PH_I64; cols[0].validity = &dummy; cols[1].type = PH_DEC64; cols[1].scale = 3; cols[1].validity = &dummy;
        prog = {col(0), col(1), op(PH_X_ADD), col(0), op(PH_X_MUL)};
        outv = true;
    } else { ph::set_error("ph_expr_jit_selfcheck: shapes 0..2"); return PH_EINVAL; }
    ph::XParams X{};
    int32_t scale = 0;
    PH_CHECK(ph::compile_expr(cols, 3, prog.data(), (int32_t)prog.size(), &X, &scale));
    for (int c = 0; c < 3; c++) { X.c[c].type = cols[c].type; X.c[c].validity = cols[c].validity; }
    std::string key, log;
    return ph::jit_compile_only(expr_jit_source(X, sel, outv, &key), "gfx950", &log);
}

extern "C" int ph_expr_eval(ph_ctx *ctx, const ph_col *cols, int32_t ncols, const ph_rpn *prog, int32_t nprog,
                            const int32_t *sel, int64_t n, int64_t *out_dev, uint8_t *out_validity_dev) {
    PH_REQUIRE(ctx && cols && prog && ncols >= 1 && ncols <= ph::X_MAX_COLS && n >= 0, "ph_expr_eval: bad arguments");
    ph::XParams X{};
    int32_t scale = 0;
    PH_CHECK(ph::compile_expr(cols, ncols, prog, nprog, &X, &scale));
    bool any_validity = false;
    for (int c = 0; c < ncols; c++) {
        X.c[c].type = cols[c].type;
        X.c[c].data = cols[c].data;
        X.c[c].validity = cols[c].validity;
        any_validity |= cols[c].validity != nullptr;
    }
    if (any_validity && !out_validity_dev) { ph::set_error("ph_expr_eval: inputs carry validity but out_validity_dev is NULL"); return PH_EINVAL; }
    if (n == 0) return PH_OK;
    // deferred errors: the overflow flag is one of the ctx's deferred words, read back by the next
    // call that synchronises anyway — no memset, no round trip here (~45 us of idle GPU per call)
    int *flag = nullptr;
    if (ctx->defer_errors) PH_CHECK(ctx->deferred_words(&flag));
    else {
        PH_CHECK(ctx->ensure_scratch(64));
        flag = (int *)ctx->scratch;
        PH_HIP(hipMemsetAsync(flag, 0, 4, ctx->stream));
    }
    // batches large enough to repay a one-time compile run the kernel generated for this expression
    if (n < (1 << 18) || expr_jit_run(ctx, X, cols, sel, n, (long long *)out_dev, out_validity_dev, flag) != PH_OK) {
        int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
        ph::expr_kernel<<<grid, 256, 0, ctx->stream>>>(X, sel, n, (long long *)out_dev, out_validity_dev, flag);
        PH_HIP(hipGetLastError());
    }
    if (ctx->defer_errors) { ctx->deferred_pending = true; return PH_OK; }
    int host_flag = 0;
    PH_CHECK(ctx->download(&host_flag, flag, 4));
    if (host_flag) { ph::set_error("ph_expr_eval: a row left the exact int64 decimal domain"); return PH_EOVERFLOW; }
    return PH_OK;
}

// ------------------------------------------------------------------ FLOAT / DOUBLE arithmetic and comparisons
namespace ph {

struct FParams {
    struct { int32_t type, scale; const void *data; const uint8_t *validity; } c[8];
    int32_t ncols, nprog, wide, truth;
    struct { int32_t op, col; float k; } prog[12];
};

// One row: the program in float32 (every operation rounds to float32: mulFloat32 and friends, function_scalar.go:1010-1025) or float64.
// Column operands are cast as the binder casts them: INTEGER -> float (tryCastInt32ToFloat32 / ..Float64), DECIMAL -> float64 ->
// float32 (tryCastDecimalToFloat32: the nearest double of the decimal, then rounded), a HUGEINT carried as a scale-0 decimal likewise.
// The comparisons are the ones selectOperation has for the type: FLOAT has > >= <=, DOUBLE has < (function_operator_boolean.go:431-490);
// the others select nothing there, and give 0 here.
__global__ __launch_bounds__(256) void float_eval_kernel(FParams F, const int32_t *__restrict__ sel, int64_t n, void *__restrict__ out,
                                                         uint8_t *__restrict__ out_valid) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool valid = true;
    double st[8];
    int sp = 0;
    if (i < n) {
        const int64_t r = sel ? (int64_t)sel[i] : i;
        for (int q = 0; q < F.nprog; q++) {
            const int op = F.prog[q].op;
            if (op == PH_X_COL) {
                const auto &c = F.c[F.prog[q].col];
                if (c.validity && !((c.validity[r >> 3] >> (r & 7)) & 1)) valid = false;
                double v;
                if (c.type == PH_I32 || c.type == PH_DATE) v = (double)((const int32_t *)c.data)[r];
                else {
                    v = (double)((const long long *)c.data)[r];
                    if (c.type == PH_DEC64 && c.scale > 0) { double p10 = 1.0; for (int s = 0; s < c.scale; s++) p10 *= 10.0; v = v / p10; }
                }
                st[sp++] = F.wide ? v : (double)(float)v;
            } else if (op == PH_X_CONST) {
                st[sp++] = (double)F.prog[q].k;      // a FLOAT literal (widened for DOUBLE arithmetic: tryCastFloat32ToFloat64)
            } else {
                const double b = st[--sp], a = st[--sp];
                double x;
                if (F.wide) {
                    x = op == PH_X_ADD ? a + b : op == PH_X_SUB ? a - b : op == PH_X_MUL ? a * b : op == PH_X_DIV ? a / b
                        : op == PH_X_LT ? (a < b ? 1.0 : 0.0) : 0.0;
                } else {
                    const float fa = (float)a, fb = (float)b;
                    float fx;
                    if (op == PH_X_ADD) fx = __fadd_rn(fa, fb);
                    else if (op == PH_X_SUB) fx = __fsub_rn(fa, fb);
                    else if (op == PH_X_MUL) fx = __fmul_rn(fa, fb);
                    else if (op == PH_X_DIV) fx = __fdiv_rn(fa, fb);
                    else fx = (op == PH_X_GT ? fa > fb : op == PH_X_GE ? fa >= fb : op == PH_X_LE ? fa <= fb : false) ? 1.0f : 0.0f;
                    x = (double)fx;
                }
                st[sp++] = x;
            }
        }
        if (F.truth) ((int32_t *)out)[i] = valid && st[0] != 0.0 ? 1 : 0;     // a NULL operand: the comparison is not true
        else ((float *)out)[i] = (float)st[0];
    }
    if (out_valid && !F.truth) {
        const unsigned long long m = __ballot(i < n && valid);
        if ((threadIdx.x & 63) == 0 && i < n) reinterpret_cast<unsigned long long *>(out_valid)[i >> 6] = m;
    }
}

}  // namespace ph

extern "C" int ph_float_eval(ph_ctx *ctx, const ph_col *cols, int32_t ncols, const ph_rpn *prog, int32_t nprog, int32_t wide, const int32_t *sel,
                             int64_t n, int32_t out_type, void *out_dev, uint8_t *out_validity_dev) {
    PH_REQUIRE(ctx && cols && prog && ncols >= 1 && ncols <= 8 && nprog >= 1 && nprog <= 12 && n >= 0 && (out_type == PH_I32 || out_type == PH_F32),
               "ph_float_eval: bad arguments (1..8 columns, 1..12 program steps, out_type PH_I32 or PH_F32)");
    if (out_type == PH_F32 && wide) { ph::set_error("ph_float_eval: DOUBLE values have no column type here: only the truth value of a DOUBLE comparison"); return PH_EUNSUPPORTED; }
    ph::FParams F{};
    F.ncols = ncols; F.nprog = nprog; F.wide = wide ? 1 : 0; F.truth = out_type == PH_I32 ? 1 : 0;
    bool any_validity = false;
    for (int c = 0; c < ncols; c++) {
        const int t = cols[c].type;
        if (t != PH_I32 && t != PH_I64 && t != PH_DEC64 && t != PH_DATE) { ph::set_error("ph_float_eval: column %d has type %d", c, t); return PH_EUNSUPPORTED; }
        F.c[c].type = t; F.c[c].scale = cols[c].scale; F.c[c].data = cols[c].data; F.c[c].validity = cols[c].validity;
        any_validity |= cols[c].validity != nullptr;
    }
    int depth = 0;
    for (int q = 0; q < nprog; q++) {
        const int op = prog[q].op;
        F.prog[q].op = op; F.prog[q].col = prog[q].col;
        if (op == PH_X_COL) { if (prog[q].col < 0 || prog[q].col >= ncols) { ph::set_error("ph_float_eval: column %d out of range", prog[q].col); return PH_EINVAL; } depth++; }
        else if (op == PH_X_CONST) { uint32_t bits = (uint32_t)prog[q].ival; memcpy(&F.prog[q].k, &bits, 4); depth++; }
        else if (op == PH_X_ADD || op == PH_X_SUB || op == PH_X_MUL || op == PH_X_DIV || op == PH_X_LT || op == PH_X_LE || op == PH_X_GT || op == PH_X_GE) {
            if (depth < 2) { ph::set_error("ph_float_eval: malformed program"); return PH_EINVAL; }
            depth--;
        } else { ph::set_error("ph_float_eval: operation %d", op); return PH_EINVAL; }
        if (depth > 8) { ph::set_error("ph_float_eval: program too deep"); return PH_EUNSUPPORTED; }
    }
    if (depth != 1) { ph::set_error("ph_float_eval: malformed program"); return PH_EINVAL; }
    if (any_validity && !F.truth && !out_validity_dev) { ph::set_error("ph_float_eval: inputs carry validity but out_validity_dev is NULL"); return PH_EINVAL; }
    if (n == 0) return PH_OK;
    PH_REQUIRE(out_dev, "ph_float_eval: out_dev is NULL");
    ph::float_eval_kernel<<<(int)((n + 255) / 256), 256, 0, ctx->stream>>>(F, sel, n, out_dev, any_validity ? out_validity_dev : nullptr);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ------------------------------------------------------------------ extract(part from date)
namespace ph {
__global__ __launch_bounds__(256) void date_extract_kernel(int part, const int32_t *__restrict__ days,
                                                           const int32_t *__restrict__ sel, int64_t n,
                                                           int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int32_t y, m, d;
        const int64_t r = sel ? (int64_t)sel[i] : i;
        civil_from_days(days[r < 0 ? 0 : r], &y, &m, &d);   // a negative row id (a strict lookup's miss, reported later) reads row 0
        out[i] = part == PH_PART_YEAR ? y : (part == PH_PART_MONTH ? m : d);
    }
}
}  // namespace ph

extern "C" int ph_date_extract(ph_ctx *ctx, int32_t part, const ph_col *col, const int32_t *sel, int64_t n,
                               int32_t *out_dev) {
    PH_REQUIRE(ctx && col && n >= 0 && (n == 0 || out_dev), "ph_date_extract: bad arguments");
    PH_REQUIRE(col->type == PH_DATE, "ph_date_extract: column type %d is not PH_DATE", col->type);
    PH_REQUIRE(part >= PH_PART_YEAR && part <= PH_PART_DAY, "ph_date_extract: unknown part %d", part);
    if (col->validity) { ph::set_error("ph_date_extract: NULL-able dates are not supported on the device"); return PH_EUNSUPPORTED; }
    if (n == 0) return PH_OK;
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 8);
    ph::date_extract_kernel<<<grid, 256, 0, ctx->stream>>>(part, (const int32_t *)col->data, sel, n, out_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

// ------------------------------------------------------------------ substring(s from offset for length)
// substringFunc / substringStartEnd (pkg/compute/function_operator_binary.go:553-625, registered by
// SubstringFunc function_scalar.go:1530-1563): byte positions, 1-based offset, negative offsets count
// from the end, a negative length reads leftwards, offset 0 shortens the length by one. Two passes:
// every row's (start, length) -> exclusive scan of the lengths -> byte copy.
namespace ph {
__device__ __forceinline__ bool substr_range(long long slen, long long offset, long long length, long long *start, long long *end) {
    if (length == 0) return false;
    if (offset > 0) *start = slen < offset - 1 ? slen : offset - 1;
    else if (offset < 0) *start = slen + offset > 0 ? slen + offset : 0;
    else {
        *start = 0;
        length--;
        if (length <= 0) return false;
    }
    if (length > 0) *end = slen < *start + length ? slen : *start + length;
    else {
        *end = *start;
        *start = *start + length > 0 ? *start + length : 0;
    }
    return *start != *end;
}

__global__ __launch_bounds__(256) void substr_len_kernel(const int32_t *__restrict__ off, const uint8_t *validity,
                                                         const int32_t *__restrict__ sel, int64_t n, long long offset,
                                                         long long length, int32_t *__restrict__ out_len,
                                                         int32_t *__restrict__ out_start, unsigned long long *__restrict__ total64) {
    // total64: the result bytes summed in 64 bits (one add per wave) — the int32 offsets of the result wrap from 2^31 bytes on (row ids
    // that repeat below a join can ask for more than the column holds), and the host refuses such a result instead of returning it
    unsigned long long mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = sel ? sel[i] : i;
        long long s = 0, e = 0;
        const bool ok = bit_valid(validity, r) && substr_range(off[r + 1] - off[r], offset, length, &s, &e);
        out_len[i] = ok ? (int32_t)(e - s) : 0;
        out_start[i] = off[r] + (int32_t)s;
        mine += ok ? (unsigned long long)(e - s) : 0ull;
    }
    // one add per WORKGROUP: adds on one address execute one after the other at the memory side (~10 ns each) — one per wave was 10 600 adds =
    // the whole 102 us of this kernel over Q22's 680 k phone numbers
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d, 64);
    __shared__ unsigned long long wsum[4];
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long all = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (all) atomicAdd(total64, all);
    }
}

__global__ __launch_bounds__(256) void substr_copy_kernel(const uint8_t *__restrict__ bytes, const int32_t *__restrict__ start,
                                                          const int32_t *__restrict__ out_off, int64_t n,
                                                          uint8_t *__restrict__ out_bytes) {
    // one wave per row batch: results are short (a few bytes), so a thread copies its own row
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int32_t o = out_off[i], len = out_off[i + 1] - o, s = start[i];
        for (int32_t k = 0; k < len; k++) out_bytes[o + k] = bytes[s + k];
    }
}

__global__ void substr_total_kernel(const int64_t *total, int32_t *out_off, int64_t n) { out_off[n] = (int32_t)*total; }

__global__ __launch_bounds__(256) void cross_pairs_kernel(int64_t nl, int64_t nr, int32_t *__restrict__ out_l, int32_t *__restrict__ out_r) {
    const int64_t total = nl * nr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        out_l[i] = (int32_t)(i % nl);   // for every right row, all left rows: the order CrossProductExec emits
        out_r[i] = (int32_t)(i / nl);
    }
}
}  // namespace ph

extern "C" int ph_substring(ph_ctx *ctx, const ph_col *col, int64_t offset, int64_t length, const int32_t *sel, int64_t n,
                            int32_t *out_offsets_dev, uint8_t *out_bytes_dev, int64_t out_bytes_capacity, int64_t *out_bytes) {
    PH_REQUIRE(ctx && col && n >= 0 && out_bytes && (n == 0 || out_offsets_dev), "ph_substring: bad arguments");
    PH_REQUIRE(col->type == PH_STR && col->data && (col->aux || col->aux_bytes == 0), "ph_substring: column type %d is not PH_STR", col->type);
    *out_bytes = 0;
    if (n == 0) return PH_OK;
    int32_t *start = nullptr;
    int64_t *total = nullptr;   // [0]: the scan's total, [1]: the 64-bit byte count
    PH_CHECK(ctx->pool_alloc(n * 4, (void **)&start));
    int rc = ctx->pool_alloc(16, (void **)&total);
    if (rc != PH_OK) { ctx->pool_release(start); return rc; }
    if (hipMemsetAsync(total, 0, 16, ctx->stream) != hipSuccess) { ctx->pool_release(start); ctx->pool_release(total); ph::set_error("ph_substring: memset failed"); return PH_EHIP; }
    int grid = (int)std::min<int64_t>((n + 255) / 256, 256 * 2);   // (few workgroups: few adds on the one total)
    ph::substr_len_kernel<<<grid, 256, 0, ctx->stream>>>((const int32_t *)col->data, col->validity, sel, n, (long long)offset,
                                                         (long long)length, out_offsets_dev, start, (unsigned long long *)(total + 1));
    rc = ph::exclusive_scan_i32(ctx, out_offsets_dev, n, total);
    long long tots[2] = {0, 0};
    if (rc == PH_OK) {
        ph::substr_total_kernel<<<1, 1, 0, ctx->stream>>>(total, out_offsets_dev, n);
        rc = ctx->download(tots, total, 16);
    }
    long long tot = tots[1];
    if (rc == PH_OK && tot >= (1ll << 31)) {
        ph::set_error("ph_substring: %lld result bytes do not fit the int32 offsets of a PH_STR column", tot);
        *out_bytes = tot;
        ctx->pool_release(start);
        ctx->pool_release(total);
        return PH_EUNSUPPORTED;
    }
    if (rc == PH_OK && tot > out_bytes_capacity) {
        ph::set_error("ph_substring: %lld result bytes, room for %lld", tot, (long long)out_bytes_capacity);
        rc = PH_ECAPACITY;
    }
    *out_bytes = tot;
    if (rc == PH_OK && tot > 0) {
        ph::substr_copy_kernel<<<grid, 256, 0, ctx->stream>>>((const uint8_t *)col->aux, start, out_offsets_dev, n, out_bytes_dev);
        if (hipGetLastError() != hipSuccess) rc = PH_EHIP;
    }
    ctx->pool_release(start);
    ctx->pool_release(total);
    return rc;
}

extern "C" int ph_cross_pairs(ph_ctx *ctx, int64_t n_left, int64_t n_right, int32_t *out_left_dev, int32_t *out_right_dev) {
    PH_REQUIRE(ctx && n_left >= 0 && n_right >= 0 && n_left < (1ll << 31) && n_right < (1ll << 31) && n_left * n_right < (1ll << 31),
               "ph_cross_pairs: bad arguments (the product must stay below 2^31 rows)");
    const int64_t total = n_left * n_right;
    if (total == 0) return PH_OK;
    PH_REQUIRE(out_left_dev && out_right_dev, "ph_cross_pairs: output is NULL");
    int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
    ph::cross_pairs_kernel<<<grid, 256, 0, ctx->stream>>>(n_left, n_right, out_left_dev, out_right_dev);
    PH_HIP(hipGetLastError());
    return PH_OK;
}

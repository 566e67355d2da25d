// See chunk.h. Product-side code: independent of oracle/ (which tests use to check it).
#include "chunk.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace plan {

typedef unsigned __int128 u128;

PhyType LType::GetInternalType() const {
    switch (Id) {
    case LTID_BOOLEAN: return PT_BOOL;
    case LTID_INTEGER: return PT_INT32;
    case LTID_BIGINT: return PT_INT64;
    case LTID_UBIGINT: return PT_UINT64;
    case LTID_DATE: return PT_DATE;
    case LTID_DECIMAL: return PT_DECIMAL;
    case LTID_FLOAT: return PT_FLOAT;
    case LTID_DOUBLE: return PT_DOUBLE;
    case LTID_VARCHAR: return PT_VARCHAR;
    case LTID_HUGEINT: return PT_INT128;
    default: return PT_INVALID;
    }
}

size_t LType::Size() const {
    switch (GetInternalType()) {
    case PT_BOOL: return 1;
    case PT_INT32: case PT_FLOAT: return 4;
    case PT_INT64: case PT_UINT64: case PT_DOUBLE: return 8;
    case PT_DATE: return sizeof(Date);
    case PT_DECIMAL: return sizeof(Decimal);
    case PT_VARCHAR: return sizeof(String);
    case PT_INT128: return sizeof(Hugeint);
    default: return 0;
    }
}

Vector::Vector(LType t, int cap) : _Typ(t) { Data.assign(t.Size() * (size_t)cap, 0); }

void Vector::ToUnifiedFormat(int count, Unified *u) const {
    switch (_PhyFormat) {
    case PF_DICT:
        u->data = Child->Data.data();
        u->sel = Sel.get();
        u->mask = &Child->Mask;
        break;
    case PF_CONST:  // ZeroSelectVectorInPhyFormatConst: every row reads slot 0
        u->ident.identity = false;
        u->ident.SelVec.assign((size_t)(count > 0 ? count : 1), 0);
        u->data = Data.data();
        u->sel = &u->ident;
        u->mask = &Mask;
        break;
    case PF_SEQUENCE: {  // Flatten (vector.go:85-92): materialise start + i*incr
        const int64_t *seq = reinterpret_cast<const int64_t *>(Data.data());
        size_t w = _Typ.Size();
        u->flat.assign(w * (size_t)(count > 0 ? count : 1), 0);
        for (int i = 0; i < count; i++) {
            int64_t v = seq[0] + seq[1] * (int64_t)i;
            if (w == 4) { int32_t x = (int32_t)v; memcpy(u->flat.data() + (size_t)i * 4, &x, 4); }
            else memcpy(u->flat.data() + (size_t)i * 8, &v, 8);
        }
        u->ident.identity = true;
        u->data = u->flat.data();
        u->sel = &u->ident;
        u->mask = &Mask;
        break;
    }
    default:
        u->ident.identity = true;
        u->data = Data.data();
        u->sel = &u->ident;
        u->mask = &Mask;
        break;
    }
}

void Vector::SetConstNull() {
    _PhyFormat = PF_CONST;
    if (Data.size() < _Typ.Size()) Data.assign(_Typ.Size(), 0);
    Mask.Bits.assign(1, 0xFE);  // slot 0 invalid
}

void Vector::Sequence(int64_t start, int64_t incr, int64_t count) {
    _PhyFormat = PF_SEQUENCE;
    Data.assign(3 * sizeof(int64_t), 0);
    int64_t seq[3] = {start, incr, count};
    memcpy(Data.data(), seq, sizeof seq);
}

void Vector::SetString(int idx, const char *s, int64_t len) {
    std::unique_ptr<char[]> buf(new char[(size_t)len + 1]);
    memcpy(buf.get(), s, (size_t)len);
    buf[(size_t)len] = 0;
    Slice<String>()[idx] = String{len, buf.get()};
    _heap.push_back(std::move(buf));
}

void Chunk::Init(const std::vector<LType> &types, int cap) {
    Data.clear();
    for (auto &t : types) Data.push_back(std::make_shared<Vector>(t, cap));
    _cap = cap;
    _count = 0;
}

void Chunk::SliceIndice(const Chunk &other, const std::shared_ptr<SelectVector> &sel, int count, int colOffset,
                        const std::vector<int> &indice) {
    for (size_t i = 0; i < indice.size(); i++) {
        auto v = std::make_shared<Vector>();
        std::shared_ptr<Vector> src = other.Data[(size_t)indice[i]];
        v->_Typ = src->_Typ;
        if (src->_PhyFormat == PF_CONST) {  // Slice of a constant stays that constant (vector.go Slice)
            *v = Vector(src->_Typ, 1);
            v->_PhyFormat = PF_CONST;
            memcpy(v->Data.data(), src->Data.data(), std::min(v->Data.size(), src->Data.size()));
            v->Mask = src->Mask;
            if (src->_Typ.GetInternalType() == PT_VARCHAR && src->Mask.RowIsValid(0)) {
                const String &str = src->Slice<String>()[0];
                v->SetString(0, str.Data, str.Len);
            }
            Data[(size_t)colOffset + i] = v;
            continue;
        }
        v->_PhyFormat = PF_DICT;
        if (src->_PhyFormat == PF_DICT) {  // selection of a selection: merge, share the flat child
            auto merged = std::make_shared<SelectVector>();
            merged->identity = false;
            merged->SelVec.resize((size_t)count);
            for (int r = 0; r < count; r++) merged->SelVec[(size_t)r] = src->Sel->GetIndex(sel->GetIndex(r));
            v->Sel = merged;
            v->Child = src->Child;
        } else if (src->_PhyFormat == PF_SEQUENCE) {  // flatten first (Flatten, vector.go:85-92)
            Vector::Unified u;
            int64_t n = reinterpret_cast<const int64_t *>(src->Data.data())[2];
            src->ToUnifiedFormat((int)n, &u);
            auto flat = std::make_shared<Vector>();
            flat->_Typ = src->_Typ;
            flat->Data = std::move(u.flat);
            v->Sel = sel;
            v->Child = flat;
        } else {
            v->Sel = sel;
            v->Child = src;
        }
        Data[(size_t)colOffset + i] = v;
    }
    _count = count;
}

// ------------------------------------------------------------------ serialization

template <typename T> static void put(std::string *o, T v) { o->append(reinterpret_cast<const char *>(&v), sizeof v); }
template <typename T> static bool get(const std::string &in, size_t *pos, T *v) {
    if (*pos + sizeof(T) > in.size()) return false;
    memcpy(v, in.data() + *pos, sizeof(T));
    *pos += sizeof(T);
    return true;
}

void Chunk::Serialize(std::string *out) const {
    put<uint32_t>(out, (uint32_t)Card());
    put<uint32_t>(out, (uint32_t)ColumnCount());
    for (auto &v : Data) {  // LType.Serialize: three Go ints (ltype.go:31-45)
        put<int64_t>(out, v->_Typ.Id);
        put<int64_t>(out, v->_Typ.Width);
        put<int64_t>(out, v->_Typ.Scale);
    }
    for (auto &v : Data) {
        Vector::Unified u;
        v->ToUnifiedFormat(Card(), &u);
        bool writeValidity = Card() > 0 && !u.mask->AllValid();
        put<uint8_t>(out, writeValidity ? 1 : 0);
        if (writeValidity) {
            std::vector<uint8_t> flat((size_t)(Card() + 7) / 8, 0xFF);
            for (int i = 0; i < Card(); i++)
                if (!u.mask->RowIsValid((uint64_t)u.sel->GetIndex(i))) flat[(size_t)i >> 3] &= (uint8_t)~(1u << (i & 7));
            out->append(reinterpret_cast<const char *>(flat.data()), flat.size());
        }
        size_t w = v->_Typ.Size();
        if (v->_Typ.GetInternalType() == PT_VARCHAR) {
            const String *s = reinterpret_cast<const String *>(u.data);
            for (int i = 0; i < Card(); i++) {
                int64_t idx = u.sel->GetIndex(i);
                bool valid = u.mask->RowIsValid((uint64_t)idx);
                uint32_t len = valid ? (uint32_t)s[idx].Len : 0;
                put<uint32_t>(out, len);
                if (len) out->append(s[idx].Data, len);
            }
        } else {
            for (int i = 0; i < Card(); i++) out->append(reinterpret_cast<const char *>(u.data) + (size_t)u.sel->GetIndex(i) * w, w);
        }
    }
}

bool Chunk::Deserialize(const std::string &in, size_t *pos, std::string *err) {
    uint32_t rows = 0, cols = 0;
    if (!get(in, pos, &rows) || !get(in, pos, &cols)) { *err = "chunk header truncated"; return false; }
    std::vector<LType> types(cols);
    for (auto &t : types) {
        int64_t id, w, s;
        if (!get(in, pos, &id) || !get(in, pos, &w) || !get(in, pos, &s)) { *err = "chunk types truncated"; return false; }
        t = LType{(LTypeId)id, (int)w, (int)s};
        if (t.Size() == 0) { *err = "unsupported column type " + std::to_string(id); return false; }
    }
    Init(types, std::max<int>((int)rows, DefaultVectorSize));
    for (auto &v : Data) {
        uint8_t has = 0;
        if (!get(in, pos, &has)) { *err = "validity flag truncated"; return false; }
        if (has) {
            size_t nb = ((size_t)rows + 7) / 8;
            if (*pos + nb > in.size()) { *err = "validity truncated"; return false; }
            v->Mask.Bits.assign(in.begin() + (long)*pos, in.begin() + (long)(*pos + nb));
            *pos += nb;
        }
        if (v->_Typ.GetInternalType() == PT_VARCHAR) {
            for (uint32_t i = 0; i < rows; i++) {
                uint32_t len = 0;
                if (!get(in, pos, &len) || *pos + len > in.size()) { *err = "string truncated"; return false; }
                v->SetString((int)i, in.data() + *pos, len);
                *pos += len;
            }
        } else {
            size_t nb = v->_Typ.Size() * rows;
            if (*pos + nb > in.size()) { *err = "column data truncated"; return false; }
            memcpy(v->Data.data(), in.data() + *pos, nb);
            *pos += nb;
        }
    }
    _count = (int)rows;
    return true;
}

// ------------------------------------------------------------------ decimal (product side)

static u128 p10(int k) { u128 r = 1; while (k-- > 0) r *= 10; return r; }
// decimal digits of v: values below 2^64 (every coefficient the device path produces) by comparison with the powers of ten, no division
static int ndig(u128 v) {
    static const uint64_t P[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull, 10000000000ull,
                                   100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull, 1000000000000000ull, 10000000000000000ull,
                                   100000000000000000ull, 1000000000000000000ull, 10000000000000000000ull};
    if ((v >> 64) == 0) {
        const uint64_t x = (uint64_t)v;
        int n = 1;
        while (n < 20 && x >= P[n]) n++;
        return n;
    }
    int n = 0;
    while (v) { v /= 10; n++; }
    return n;
}

Decimal DecimalFromUnscaled(int64_t unscaled, int scale) {
    Decimal d{};
    d.neg = unscaled < 0;
    uint64_t c = d.neg ? (uint64_t)0 - (uint64_t)unscaled : (uint64_t)unscaled;
    while (scale > 0 && c % 10 == 0) { c /= 10; scale--; }  // NewFromInt64 drops trailing zeros
    d.coef = c;
    d.scale = (int8_t)scale;
    if (c == 0) d.neg = false;
    return d;
}

bool DecimalFromInt128(__int128 v, int scale, Decimal *out) {
    bool neg = v < 0;
    u128 c = neg ? (u128)(-v) : (u128)v;
    if (ndig(c) > 19) return false;
    Decimal d{};
    d.neg = neg && c != 0;
    d.coef = (uint64_t)c;
    d.scale = (int8_t)scale;
    *out = d;
    return true;
}

// round c / 10^drop half-to-even, `sticky` = non-zero digits below c
static u128 rsh_half_even(u128 c, int drop, bool sticky) {
    if (drop <= 0) return c;
    u128 d = p10(drop), q = c / d, rem = c % d, half = d / 2;
    if (rem > half || (rem == half && (sticky || (q & 1)))) q++;
    return q;
}

bool DecimalQuoCount(__int128 sum, int scale, uint64_t count, Decimal *out) {
    // AvgOp.Finalize: sum.Quo(MustNew(count, 0)) (function_aggr.go:886-895)
    if (count == 0) return false;
    bool neg = sum < 0;
    u128 n = neg ? (u128)(-sum) : (u128)sum;
    if (ndig(n) > 19) return false;  // the reference's sum would already have overflowed
    int pref = scale;               // preferred scale max(0, sd - se) with se = 0
    if (n == 0) { *out = Decimal{false, 0, (int8_t)pref}; return true; }
    u128 d = count, c = n / d, r = n % d;
    int s = scale;
    while (r != 0 && ndig(c) <= 19 && s <= 19) {
        r *= 10;
        c = c * 10 + r / d;
        r %= d;
        s++;
    }
    bool sticky = r != 0;
    for (;;) {
        int drop = std::max(ndig(c) - 19, s - 19);
        if (drop <= 0) break;
        if (s - drop < 0) return false;
        c = rsh_half_even(c, drop, sticky);
        sticky = false;
        s -= drop;
    }
    while (s > pref && c % 10 == 0) { c /= 10; s--; }
    *out = Decimal{neg && c != 0, (uint64_t)c, (int8_t)s};
    return true;
}

bool DecimalQuo(const Decimal &a, const Decimal &b, Decimal *out) {
    // binDecimalDivOp: left.Quo(right) (function_operator_binary.go:199-207) — the quotient to 19 significant digits, half-even,
    // trailing zeros trimmed down to the preferred scale max(0, scale(a) - scale(b)); the same digit loop as DecimalQuoCount
    if (b.coef == 0) return false;
    int pref = std::max(0, (int)a.scale - (int)b.scale);
    if (a.coef == 0) { *out = Decimal{false, 0, (int8_t)pref}; return true; }
    u128 n = a.coef;
    int s = (int)a.scale - (int)b.scale;
    if (s < 0) { n *= p10(-s); s = 0; }
    u128 d = b.coef, c = n / d, r = n % d;
    while (r != 0 && ndig(c) <= 19 && s <= 19) {
        r *= 10;
        c = c * 10 + r / d;
        r %= d;
        s++;
    }
    bool sticky = r != 0;
    for (;;) {
        int drop = std::max(ndig(c) - 19, s - 19);
        if (drop <= 0) break;
        if (s - drop < 0) return false;
        c = rsh_half_even(c, drop, sticky);
        sticky = false;
        s -= drop;
    }
    while (s > pref && c % 10 == 0) { c /= 10; s--; }
    *out = Decimal{(a.neg != b.neg) && c != 0, (uint64_t)c, (int8_t)s};
    return true;
}

std::string DecimalString(const Decimal &d) {
    char digs[32];
    int n = snprintf(digs, sizeof digs, "%llu", (unsigned long long)d.coef);
    std::string s = d.neg ? "-" : "";
    if (d.scale == 0) s += digs;
    else if (n > d.scale) { s.append(digs, (size_t)(n - d.scale)); s += '.'; s.append(digs + n - d.scale); }
    else { s += "0."; s.append((size_t)(d.scale - n), '0'); s += digs; }
    return s;
}

std::string DecimalValueString(const Decimal &d, int typeScale) {
    // Vector.GetValue: Int64(Typ.Scale) (vector.go:121-137) then NewFromInt64(w,f,scale).String()
    u128 c = d.coef;
    if (typeScale < d.scale) c = rsh_half_even(c, d.scale - typeScale, false);
    else c *= p10(typeScale - d.scale);
    if (ndig(c) > 19) return DecimalString(d);
    u128 y = p10(typeScale);
    u128 w = c / y, f = c % y;
    int sc = typeScale;
    if (f == 0) sc = 0; else while (f % 10 == 0) { f /= 10; sc--; }
    Decimal r{d.neg && (w != 0 || f != 0), (uint64_t)(w * p10(sc) + f), (int8_t)sc};
    return DecimalString(r);
}

bool DecimalToUnscaled(const Decimal &d, int scale, int64_t *out) {
    u128 c = d.coef;
    if (scale >= d.scale) c *= p10(scale - d.scale);
    else { if (c % p10(d.scale - scale) != 0) return false; c /= p10(d.scale - scale); }
    if (c > (u128)INT64_MAX) return false;
    *out = d.neg ? -(int64_t)c : (int64_t)c;
    return true;
}

bool DecimalFloorUnscaled(const Decimal &d, int scale, int64_t *out) {
    u128 c = d.coef;
    bool inexact = false;
    if (scale >= d.scale) c *= p10(scale - d.scale);
    else { inexact = c % p10(d.scale - scale) != 0; c /= p10(d.scale - scale); }
    if (c >= (u128)INT64_MAX) return false;
    *out = d.neg ? -(int64_t)c - (inexact ? 1 : 0) : (int64_t)c;
    return true;
}

// ------------------------------------------------------------------ dates

int32_t DaysFromDate(const Date &dt) {
    int32_t y = dt.Year, m = dt.Month, d = dt.Day;
    y -= m <= 2;
    int32_t era = (y >= 0 ? y : y - 399) / 400;
    uint32_t yoe = (uint32_t)(y - era * 400);
    uint32_t doy = (153u * (uint32_t)(m + (m > 2 ? -3 : 9)) + 2u) / 5u + (uint32_t)d - 1u;
    uint32_t doe = yoe * 365u + yoe / 4u - yoe / 100u + doy;
    return era * 146097 + (int32_t)doe - 719468;
}

Date DateFromDays(int32_t z) {
    z += 719468;
    int32_t era = (z >= 0 ? z : z - 146096) / 146097;
    uint32_t doe = (uint32_t)(z - era * 146097);
    uint32_t yoe = (doe - doe / 1460u + doe / 36524u - doe / 146096u) / 365u;
    int32_t y = (int32_t)yoe + era * 400;
    uint32_t doy = doe - (365u * yoe + yoe / 4u - yoe / 100u);
    uint32_t mp = (5u * doy + 2u) / 153u;
    Date r;
    r.Day = (int32_t)(doy - (153u * mp + 2u) / 5u + 1u);
    r.Month = (int32_t)(mp < 10 ? mp + 3 : mp - 9);
    r.Year = y + (r.Month <= 2);
    return r;
}

// ------------------------------------------------------------------ text

static std::string go_float(double v) {
    // fmt "%v" of a float64: shortest round-trip digits, %e form when exp < -4 || exp >= 21
    // is JSON's rule; Go's %v uses 21 only for the 'g' shortest form threshold of... 6. Keep 6.
    if (v != v) return "NaN";
    if (v == 0) return "0";
    char tmp[64];
    // the shortest digit string that parses back to v. Fifteen digits first: if they do, every shorter string that does is those digits with
    // their trailing zeros cut (a shorter round-tripping string differs from v by less than half an ulp, far below the fifteenth digit) — the
    // zeros are stripped below; else sixteen, else seventeen (which always do). One to three conversions instead of a search over 1..17.
    for (int prec = 15; prec <= 17; prec++) {
        snprintf(tmp, sizeof tmp, "%.*e", prec - 1, v);
        if (prec == 17 || strtod(tmp, nullptr) == v) break;
    }
    std::string digits;
    const char *p = tmp;
    bool neg = false;
    if (*p == '-') { neg = true; p++; }
    for (; *p && *p != 'e'; p++) if (*p != '.') digits += *p;
    int exp = atoi(p + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    std::string o = neg ? "-" : "";
    int nd = (int)digits.size();
    if (exp < -4 || exp >= 6) {
        o += digits[0];
        if (nd > 1) { o += '.'; o += digits.substr(1); }
        char e[16];
        snprintf(e, sizeof e, "e%c%02d", exp < 0 ? '-' : '+', exp < 0 ? -exp : exp);
        o += e;
    } else if (exp < 0) {
        o += "0.";
        o.append((size_t)(-exp - 1), '0');
        o += digits;
    } else {
        for (int i = 0; i <= exp; i++) o += i < nd ? digits[(size_t)i] : '0';
        if (nd > exp + 1) { o += '.'; o += digits.substr((size_t)exp + 1); }
    }
    return o;
}

double DecimalToDouble(const Decimal &d) { return strtod(DecimalString(d).c_str(), nullptr); }   // correctly rounded, like the Go parse

// one value's text appended to *out; u = the vector's unified format over at least row + 1 rows
static void AppendValue(std::string *out, const Vector &v, const Vector::Unified &u, int row) {
    int64_t idx = u.sel->GetIndex(row);
    if (!u.mask->RowIsValid((uint64_t)idx)) { *out += "NULL"; return; }
    char buf[64];
    switch (v._Typ.Id) {
    case LTID_INTEGER: out->append(buf, (size_t)snprintf(buf, sizeof buf, "%d", reinterpret_cast<const int32_t *>(u.data)[idx])); return;
    case LTID_BIGINT: out->append(buf, (size_t)snprintf(buf, sizeof buf, "%lld", (long long)reinterpret_cast<const int64_t *>(u.data)[idx])); return;
    case LTID_VARCHAR: { const String &s = reinterpret_cast<const String *>(u.data)[idx]; out->append(s.Data, (size_t)s.Len); return; }
    case LTID_DECIMAL: *out += DecimalValueString(reinterpret_cast<const Decimal *>(u.data)[idx], v._Typ.Scale); return;
    case LTID_DATE: { const Date &d = reinterpret_cast<const Date *>(u.data)[idx]; out->append(buf, (size_t)snprintf(buf, sizeof buf, "%04d-%02d-%02d", d.Year, d.Month, d.Day)); return; }
    case LTID_DOUBLE: *out += go_float(reinterpret_cast<const double *>(u.data)[idx]); return;
    case LTID_FLOAT: *out += go_float((double)reinterpret_cast<const float *>(u.data)[idx]); return;   // Value.F64 = float64(float32), printed %v
    case LTID_HUGEINT: {
        const Hugeint &h = reinterpret_cast<const Hugeint *>(u.data)[idx];
        __int128 x = ((__int128)h.Upper << 64) + (__int128)(u128)h.Lower;
        bool neg = x < 0;
        u128 a = neg ? (u128)(-x) : (u128)x;
        int n = 0;
        do { buf[n++] = (char)('0' + (int)(a % 10)); a /= 10; } while (a);
        if (neg) buf[n++] = '-';
        std::reverse(buf, buf + n);
        out->append(buf, (size_t)n);
        return;
    }
    default: *out += '?';
    }
}

std::string ValueString(const Vector &v, int row) {
    Vector::Unified u;
    v.ToUnifiedFormat(row + 1, &u);
    std::string s;
    AppendValue(&s, v, u, row);
    return s;
}

void Chunk::AppendText(std::string *out) const {
    // the unified format once per column (a constant vector's is a selection of Card() zeros), then row by row
    std::vector<Vector::Unified> us((size_t)ColumnCount());
    for (int j = 0; j < ColumnCount(); j++) Data[(size_t)j]->ToUnifiedFormat(Card(), &us[(size_t)j]);
    for (int i = 0; i < Card(); i++) {
        for (int j = 0; j < ColumnCount(); j++) {
            AppendValue(out, *Data[(size_t)j], us[(size_t)j], i);
            if (j + 1 < ColumnCount()) *out += '\t';
        }
        *out += '\n';
    }
}

}  // namespace plan

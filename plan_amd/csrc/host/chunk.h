// Host-side mirror of the reference's columnar data plane (pkg/chunk, pkg/common, pkg/util) — the
// boundary TYPES the GPU executors consume and produce. Layouts and method names follow the Go
// code so the executors and their tests read like the reference's:
//   Chunk            pkg/chunk/chunk.go:16-20    Init :22-28, Card/SetCard :49-56, SliceIndice :82-93
//   Vector           pkg/chunk/vector.go:15-22   formats FLAT/CONST/DICT/SEQUENCE (phy_format.go)
//   SelectVector     pkg/chunk/select_vector.go:7-9   ([]int)
//   Bitmap           pkg/util/bitmap.go          1 bit/row LSB first, empty = all valid
//   LType/PhyType    pkg/common/ltype.go, phy_type.go:67-114 (sizes via unsafe.Sizeof, types.go:22-38)
//   Date/Decimal/Hugeint/String   pkg/common/date.go:8-12, decimal.go:7-9, hugeint.go:8-11, string.go:10-13
//   Serialize        chunk.go:168-194, vector_serialize.go:9-64, ltype.go:31-45 (the stub fixture format)
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace plan {

constexpr int DefaultVectorSize = 2048;  // pkg/util/util.go:123-125
// Buffer capacity of a chunk that is known to hold `card` rows and is never appended to (an aggregate's result rows, a sorted page): the
// reference allocates every chunk at 2048 rows (ensureOutputChunk, executor.go:201-210) — ~400 KB of zeroed memory for Q1's four group
// rows, per chunk and query: 10 us each beside a 320 us kernel. Same rows, same Card(); only the allocation is right-sized.
inline int ChunkCapacityFor(int card) { int c = (card < 1 ? 1 : card); c = (c + 63) / 64 * 64; return c > DefaultVectorSize ? DefaultVectorSize : c; }

enum LTypeId : int {  // pkg/common/type_id.go
    LTID_INVALID = 0, LTID_BOOLEAN = 10, LTID_INTEGER = 13, LTID_BIGINT = 14, LTID_DATE = 15,
    LTID_DECIMAL = 21, LTID_FLOAT = 22, LTID_DOUBLE = 23, LTID_VARCHAR = 25, LTID_UBIGINT = 31,
    LTID_HUGEINT = 50
};

enum PhyType { PT_BOOL, PT_INT32, PT_INT64, PT_UINT64, PT_FLOAT, PT_DOUBLE, PT_DATE, PT_DECIMAL, PT_VARCHAR, PT_INT128, PT_INVALID };

struct Date { int32_t Year, Month, Day; };                 // 12 bytes
struct Decimal { bool neg; uint64_t coef; int8_t scale; }; // 24 bytes, like the Go struct
struct Hugeint { uint64_t Lower; int64_t Upper; };         // 16 bytes
struct String { int64_t Len; char *Data; };                // 16 bytes; bytes are malloc'd

static_assert(sizeof(Date) == 12 && sizeof(Decimal) == 24 && sizeof(Hugeint) == 16 && sizeof(String) == 16, "layout");

struct LType {
    LTypeId Id = LTID_INVALID;
    int Width = 0, Scale = 0;
    PhyType GetInternalType() const;  // ltype.go:272-330
    size_t Size() const;              // phy_type.go:67-114
    bool operator==(const LType &o) const { return Id == o.Id && Width == o.Width && Scale == o.Scale; }
};
inline LType IntegerType() { return {LTID_INTEGER, 0, 0}; }
inline LType BigintType() { return {LTID_BIGINT, 0, 0}; }
inline LType DateType() { return {LTID_DATE, 0, 0}; }
inline LType DecimalType(int w, int s) { return {LTID_DECIMAL, w, s}; }
inline LType VarcharType() { return {LTID_VARCHAR, 0, 0}; }
inline LType DoubleType() { return {LTID_DOUBLE, 0, 0}; }
inline LType FloatType() { return {LTID_FLOAT, 0, 0}; }
inline LType HugeintType() { return {LTID_HUGEINT, 0, 0}; }

struct Bitmap {
    std::vector<uint8_t> Bits;  // empty = all valid
    bool AllValid() const { return Bits.empty(); }
    bool RowIsValid(uint64_t i) const { return Bits.empty() || ((Bits[i >> 3] >> (i & 7)) & 1); }
    void Init(int count) { Bits.assign((size_t)(count + 7) / 8, 0xFF); }
    void SetInvalid(uint64_t i, int cap) { if (Bits.empty()) Init(cap); Bits[i >> 3] &= (uint8_t)~(1u << (i & 7)); }
};

struct SelectVector {
    std::vector<int64_t> SelVec;  // Go `[]int`; empty + identity = incremental
    bool identity = true;
    int64_t GetIndex(int64_t i) const { return identity ? i : SelVec[(size_t)i]; }
};

enum PhyFormat { PF_FLAT, PF_CONST, PF_DICT, PF_SEQUENCE };

struct Vector {
    PhyFormat _PhyFormat = PF_FLAT;
    LType _Typ;
    std::vector<uint8_t> Data;
    Bitmap Mask;
    // PF_DICT: selection + child (vector.go:401-409)
    std::shared_ptr<SelectVector> Sel;
    std::shared_ptr<Vector> Child;
    std::vector<std::unique_ptr<char[]>> _heap;  // owns VARCHAR bytes (the reference leaks them)

    Vector() = default;
    Vector(LType t, int cap);
    template <typename T> T *Slice() { return reinterpret_cast<T *>(Data.data()); }
    template <typename T> const T *Slice() const { return reinterpret_cast<const T *>(Data.data()); }
    // ToUnifiedFormat (vector_format.go:64-97): data pointer + selection + mask for any format
    // `ident` backs the incremental / all-zero selection, `flat` the materialised SEQUENCE
    struct Unified { const uint8_t *data; const SelectVector *sel; const Bitmap *mask; SelectVector ident; std::vector<uint8_t> flat; };
    void ToUnifiedFormat(int count, Unified *u) const;
    // PF_CONST: one value (slot 0) for every row (vector.go:189-200); null = every row NULL
    void SetConstNull();
    // PF_SEQUENCE: start, start+incr, ... (vector.go:303-313); INTEGER / BIGINT only
    void Sequence(int64_t start, int64_t incr, int64_t count);
    void SetString(int idx, const char *s, int64_t len);
};

struct Chunk {
    std::vector<std::shared_ptr<Vector>> Data;
    int _count = 0, _cap = 0;
    void Init(const std::vector<LType> &types, int cap);  // chunk.go:22-28
    int Card() const { return _count; }
    void SetCard(int c) { _count = c; }
    int ColumnCount() const { return (int)Data.size(); }
    // SliceIndice (chunk.go:82-93): output columns become DICT views over `other`'s vectors
    void SliceIndice(const Chunk &other, const std::shared_ptr<SelectVector> &sel, int count, int colOffset,
                     const std::vector<int> &indice);
    // stub fixture format
    void Serialize(std::string *out) const;
    bool Deserialize(const std::string &in, size_t *pos, std::string *err);
    // SaveToFile text (chunk.go:196-220 via Vector.GetValue / Value.String)
    void AppendText(std::string *out) const;
};

// Value.String for one cell (vector.go:76-186, value.go:26-70)
std::string ValueString(const Vector &v, int row);

// ---- decimal helpers of the product side (govalues semantics needed at the boundary) ----
// NewFromInt64(whole, frac, scale): trims the fraction's trailing zeros
Decimal DecimalFromUnscaled(int64_t unscaled, int scale);
// exact 128-bit unscaled sum at `scale` -> Decimal (fails > 19 digits)
bool DecimalFromInt128(__int128 v, int scale, Decimal *out);
// sum.Quo(count): 19 significant digits, half-even, trailing zeros trimmed (AvgOp.Finalize)
bool DecimalQuoCount(__int128 sum, int scale, uint64_t count, Decimal *out);
// a.Quo(b): DECIMAL `/` (binDecimalDivOp); false on division by zero / a quotient beyond 19 digits
bool DecimalQuo(const Decimal &a, const Decimal &b, Decimal *out);
std::string DecimalString(const Decimal &d);
// Float64(): the nearest double of the decimal's text (what tryCastDecimalToFloat32 / ToFloat64 start from)
double DecimalToDouble(const Decimal &d);
// Int64(scale) rounding + NewFromInt64 + String: the text the reference prints for a DECIMAL cell
std::string DecimalValueString(const Decimal &d, int typeScale);
// unscaled int64 of a Decimal at `scale` (exact) — what the staging code uploads
bool DecimalToUnscaled(const Decimal &d, int scale, int64_t *out);
// floor of the decimal at `scale` as an unscaled int64: for an x with exactly `scale` digits, x > d  <=>  unscaled(x) > floor
bool DecimalFloorUnscaled(const Decimal &d, int scale, int64_t *out);

int32_t DaysFromDate(const Date &d);
Date DateFromDays(int32_t days);

}  // namespace plan

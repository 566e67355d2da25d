// ph_table_create_arrow: Arrow C data interface -> resident table (see planhip.h). Host code: the Arrow buffers are
// narrowed / re-packed where needed and handed to ph_table_create, which stages them through pinned memory into HBM.
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct Staged {                       // what keeps the temporaries of one column alive until ph_table_create returns
    std::vector<int64_t> i64;
    std::vector<int32_t> i32;
    std::vector<uint8_t> u8, validity, bytes;
    std::string dict_blob;
};

// validity bitmap of rows [offset, offset + n): the array's own bytes when the offset is byte-aligned, else re-packed
const uint8_t *validity_of(const ArrowArray *a, int64_t n, Staged *st) {
    if (a->null_count == 0 || a->n_buffers < 1 || !a->buffers[0]) return nullptr;
    const uint8_t *bits = (const uint8_t *)a->buffers[0];
    if (a->offset % 8 == 0) return bits + a->offset / 8;
    st->validity.assign((size_t)(n + 7) / 8, 0);
    for (int64_t i = 0; i < n; i++) {
        const int64_t s = a->offset + i;
        if ((bits[s >> 3] >> (s & 7)) & 1) st->validity[(size_t)i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    return st->validity.data();
}

bool valid_at(const uint8_t *v, int64_t i) { return !v || ((v[i >> 3] >> (i & 7)) & 1); }

// strings of a column (any of utf8 / large_utf8, plain or dictionary-encoded) as (pointer, length) per row
struct StrView { const char *p; int64_t len; };

int string_views(const ArrowSchema *sc, const ArrowArray *a, int64_t n, const uint8_t *validity, std::vector<StrView> *out) {
    auto values = [&](const ArrowSchema *vs, const ArrowArray *va, int64_t row, StrView *v) -> bool {
        const bool large = vs->format[0] == 'U';
        if (va->n_buffers < 3) return false;
        const int64_t r = va->offset + row;
        int64_t b, e;
        if (large) { const int64_t *o = (const int64_t *)va->buffers[1]; b = o[r]; e = o[r + 1]; }
        else { const int32_t *o = (const int32_t *)va->buffers[1]; b = o[r]; e = o[r + 1]; }
        v->p = (const char *)va->buffers[2] + b;
        v->len = e - b;
        return true;
    };
    out->resize((size_t)n);
    if (sc->dictionary) {
        if (!a->dictionary || a->n_buffers < 2) return PH_EINVAL;
        const char f = sc->format[0];
        for (int64_t i = 0; i < n; i++) {
            (*out)[(size_t)i] = StrView{"", 0};
            if (!valid_at(validity, i)) continue;
            const int64_t r = a->offset + i;
            int64_t code;
            switch (f) {
            case 'c': code = ((const int8_t *)a->buffers[1])[r]; break;
            case 'C': code = ((const uint8_t *)a->buffers[1])[r]; break;
            case 's': code = ((const int16_t *)a->buffers[1])[r]; break;
            case 'S': code = ((const uint16_t *)a->buffers[1])[r]; break;
            case 'i': code = ((const int32_t *)a->buffers[1])[r]; break;
            case 'I': code = ((const uint32_t *)a->buffers[1])[r]; break;
            case 'l': code = ((const int64_t *)a->buffers[1])[r]; break;
            default: return PH_EUNSUPPORTED;
            }
            if (code < 0 || code >= a->dictionary->length || !values(sc->dictionary, a->dictionary, code, &(*out)[(size_t)i])) return PH_EINVAL;
        }
        return PH_OK;
    }
    for (int64_t i = 0; i < n; i++) {
        (*out)[(size_t)i] = StrView{"", 0};
        if (valid_at(validity, i) && !values(sc, a, i, &(*out)[(size_t)i])) return PH_EINVAL;
    }
    return PH_OK;
}

}  // namespace

extern "C" int ph_table_create_arrow(ph_ctx *ctx, const ArrowSchema *schema, const ArrowArray *batch, const int32_t *cols, int32_t ncols,
                                     ph_table **out) {
    PH_REQUIRE(ctx && schema && batch && out && schema->format, "ph_table_create_arrow: bad arguments");
    PH_REQUIRE(!strcmp(schema->format, "+s") && schema->n_children == batch->n_children && batch->n_children > 0,
               "ph_table_create_arrow: a record batch is a struct array (\"+s\") whose children are the columns");
    PH_REQUIRE(batch->offset == 0, "ph_table_create_arrow: a sliced struct array is not supported (slice the children)");
    std::vector<int32_t> pick;
    if (cols && ncols > 0) pick.assign(cols, cols + ncols);
    else for (int64_t c = 0; c < batch->n_children; c++) pick.push_back((int32_t)c);
    const int64_t n = batch->length;
    std::vector<ph_col> hc(pick.size());
    std::vector<Staged> st(pick.size());
    for (size_t k = 0; k < pick.size(); k++) {
        PH_REQUIRE(pick[k] >= 0 && pick[k] < batch->n_children, "ph_table_create_arrow: column %d out of range", pick[k]);
        const ArrowSchema *sc = schema->children[pick[k]];
        const ArrowArray *a = batch->children[pick[k]];
        PH_REQUIRE(sc && a && sc->format && a->length == n, "ph_table_create_arrow: column %d: length differs from the batch's", pick[k]);
        ph_col &h = hc[k];
        h = ph_col{};
        const uint8_t *val = validity_of(a, n, &st[k]);
        h.validity = val;
        const char *f = sc->format;
        const bool is_string = !strcmp(f, "u") || !strcmp(f, "U") || (sc->dictionary && sc->dictionary->format && (!strcmp(sc->dictionary->format, "u") || !strcmp(sc->dictionary->format, "U")));
        if (is_string) {
            std::vector<StrView> sv;
            int rc = string_views(sc, a, n, val, &sv);
            if (rc != PH_OK) { ph::set_error("ph_table_create_arrow: column %d: malformed or unsupported string array (format %s)", pick[k], f); return rc; }
            // <= 256 distinct strings: dictionary codes in byte order (code order = string order)
            std::map<std::string, int> dict;
            bool small = true;
            for (int64_t i = 0; i < n && small; i++) {
                if (!valid_at(val, i)) continue;
                dict.emplace(std::string(sv[(size_t)i].p, (size_t)sv[(size_t)i].len), 0);
                small = dict.size() <= 256;
            }
            if (small) {
                int code = 0;
                for (auto &kv : dict) { kv.second = code++; st[k].dict_blob += kv.first; st[k].dict_blob.push_back('\0'); }
                st[k].u8.assign((size_t)std::max<int64_t>(n, 1), 0);
                for (int64_t i = 0; i < n; i++)
                    if (valid_at(val, i)) st[k].u8[(size_t)i] = (uint8_t)dict[std::string(sv[(size_t)i].p, (size_t)sv[(size_t)i].len)];
                h.type = PH_CODE8; h.data = st[k].u8.data(); h.aux = st[k].dict_blob.data(); h.aux_bytes = (int64_t)st[k].dict_blob.size();
            } else {
                st[k].i32.assign((size_t)n + 1, 0);
                int64_t total = 0;
                for (int64_t i = 0; i < n; i++) total += sv[(size_t)i].len;
                PH_REQUIRE(total < (1ll << 31), "ph_table_create_arrow: column %d holds %lld string bytes (int32 offsets)", pick[k], (long long)total);
                st[k].bytes.resize((size_t)std::max<int64_t>(total, 1));
                int64_t pos = 0;
                for (int64_t i = 0; i < n; i++) {
                    st[k].i32[(size_t)i] = (int32_t)pos;
                    if (sv[(size_t)i].len) memcpy(st[k].bytes.data() + pos, sv[(size_t)i].p, (size_t)sv[(size_t)i].len);
                    pos += sv[(size_t)i].len;
                }
                st[k].i32[(size_t)n] = (int32_t)pos;
                h.type = PH_STR; h.data = st[k].i32.data(); h.aux = st[k].bytes.data(); h.aux_bytes = total;
            }
            continue;
        }
        PH_REQUIRE(!sc->dictionary, "ph_table_create_arrow: column %d: only string dictionaries are supported", pick[k]);
        PH_REQUIRE(a->n_buffers >= 2 && (a->buffers[1] || n == 0), "ph_table_create_arrow: column %d has no data buffer", pick[k]);
        // fixed-width values: the array's own buffer — unless the column has NULLs, whose slots hold arbitrary bytes in
        // Arrow and must not reach the min / max statistics: those are copied with the NULL slots zeroed
        auto fixed32 = [&](int32_t type) {
            const int32_t *src = (const int32_t *)a->buffers[1] + a->offset;
            h.type = type; h.data = src;
            if (!val) return;
            st[k].i32.assign((size_t)std::max<int64_t>(n, 1), 0);
            for (int64_t i = 0; i < n; i++) if (valid_at(val, i)) st[k].i32[(size_t)i] = src[i];
            h.data = st[k].i32.data();
        };
        if (!strcmp(f, "i")) fixed32(PH_I32);
        else if (!strcmp(f, "tdD")) fixed32(PH_DATE);
        else if (!strcmp(f, "l")) {
            const int64_t *src = (const int64_t *)a->buffers[1] + a->offset;
            h.type = PH_I64; h.data = src;
            if (val) {
                st[k].i64.assign((size_t)std::max<int64_t>(n, 1), 0);
                for (int64_t i = 0; i < n; i++) if (valid_at(val, i)) st[k].i64[(size_t)i] = src[i];
                h.data = st[k].i64.data();
            }
        }
        else if (!strncmp(f, "d:", 2)) {
            int prec = 0, scale = 0, bits = 128;
            if (sscanf(f + 2, "%d,%d,%d", &prec, &scale, &bits) < 2 || bits != 128 || scale < 0 || scale > 18) {
                ph::set_error("ph_table_create_arrow: column %d: decimal format %s (decimal128 with a scale of 0..18 only)", pick[k], f);
                return PH_EUNSUPPORTED;
            }
            const int64_t *w = (const int64_t *)a->buffers[1] + 2 * a->offset;   // 16-byte little-endian two's complement
            st[k].i64.assign((size_t)std::max<int64_t>(n, 1), 0);
            for (int64_t i = 0; i < n; i++) {
                if (!valid_at(val, i)) continue;
                if (w[2 * i + 1] != (w[2 * i] >> 63)) { ph::set_error("ph_table_create_arrow: column %d row %lld: decimal outside the int64 range", pick[k], (long long)i); return PH_EOVERFLOW; }
                st[k].i64[(size_t)i] = w[2 * i];
            }
            h.type = PH_DEC64; h.scale = scale; h.data = st[k].i64.data();
        } else {
            ph::set_error("ph_table_create_arrow: column %d: arrow format \"%s\" has no device encoding", pick[k], f);
            return PH_EUNSUPPORTED;
        }
    }
    return ph_table_create(ctx, (int32_t)hc.size(), hc.data(), n, out);
}

extern "C" int32_t ph_table_dict_size(const ph_table *t, int32_t c) {
    if (!t || c < 0 || c >= (int32_t)t->cols.size()) return -1;
    return (int32_t)t->cols[(size_t)c].dict.size();
}

extern "C" const char *ph_table_dict_entry(const ph_table *t, int32_t c, int32_t code) {
    if (!t || c < 0 || c >= (int32_t)t->cols.size() || code < 0 || code >= (int32_t)t->cols[(size_t)c].dict.size()) return nullptr;
    return t->cols[(size_t)c].dict[(size_t)code].c_str();
}

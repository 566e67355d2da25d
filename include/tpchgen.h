/*
 * tpchgen.h — clean-room TPC-H data generator (numeric / key / date / code columns only)
 * for the tables the hot path reads: lineitem, orders, customer, part, partsupp,
 * supplier, nation.
 *
 * Why it exists: the reference pins its compute path only through SF1 text goldens
 * (cases/tpch/1g/plan/q{1,3,6,9}.txt) produced from official `dbgen -s 1` data, which is
 * not in this environment. This generator follows the public TPC-H specification's
 * data-generation rules (per-column Lehmer streams, a = 16807, m = 2^31-1, with per-row
 * stream advance) so that the same rows come out, and the goldens become usable as
 * third-party known answers (tests/test_golden_tpch.py checks that they do).
 * It is also the synthetic-data source of bench.py ("data": "synthetic").
 *
 * Every generator can start at an arbitrary row ("first") — streams are advanced by
 * modular exponentiation — so N ranks can each generate their own shard.
 *
 * Column encodings are the device encodings of SURVEY.md §8(d):
 *   INTEGER -> int32, BIGINT -> int64, DECIMAL(15,2) -> int64 unscaled (scale 2),
 *   DATE -> int32 days since 1970-01-01, VARCHAR(1)/c_mktsegment -> uint8 dictionary code.
 */
#ifndef TPCHGEN_H
#define TPCHGEN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Scale factor is passed as a rational sf_num/sf_den (e.g. 1/100 for SF0.01, 10/1 for SF10). */

/* Table cardinalities at a scale factor. orders = 1,500,000*sf, customer = 150,000*sf,
 * part = 200,000*sf, supplier = 10,000*sf, partsupp = 4*part, nation = 25. */
int64_t tpchgen_orders_count(int64_t sf_num, int64_t sf_den);
int64_t tpchgen_customer_count(int64_t sf_num, int64_t sf_den);
int64_t tpchgen_part_count(int64_t sf_num, int64_t sf_den);
int64_t tpchgen_supplier_count(int64_t sf_num, int64_t sf_den);

/* Number of lineitem rows belonging to orders [first_order, first_order+n_orders)
 * (0-based order ordinals). */
int64_t tpchgen_lineitem_count(int64_t sf_num, int64_t sf_den,
                               int64_t first_order, int64_t n_orders);

/* Dictionaries (code -> string). Codes are positions in these arrays. */
extern const char *const TPCHGEN_RETURNFLAG_DICT[3];  /* "A","N","R" */
extern const char *const TPCHGEN_LINESTATUS_DICT[2];  /* "F","O" */
extern const char *const TPCHGEN_MKTSEGMENT_DICT[5];  /* AUTOMOBILE,BUILDING,FURNITURE,HOUSEHOLD,MACHINERY */
extern const char *const TPCHGEN_NATION_NAMES[25];
extern const int32_t TPCHGEN_NATION_REGION[25];        /* n_regionkey of nation n */
extern const char *const TPCHGEN_REGION_NAMES[5];      /* AFRICA, AMERICA, ASIA, EUROPE, MIDDLE EAST */
extern const char *const TPCHGEN_COLORS[92];
/* dictionaries of the VARCHAR columns generated as codes (all in ascending byte order, so code order = string order) */
extern const char *const TPCHGEN_SHIPMODE_DICT[7];     /* AIR, FOB, MAIL, RAIL, REG AIR, SHIP, TRUCK */
extern const char *const TPCHGEN_SHIPINSTRUCT_DICT[4]; /* COLLECT COD, DELIVER IN PERSON, NONE, TAKE BACK RETURN */
extern const char *const TPCHGEN_ORDERPRIORITY_DICT[5];/* 1-URGENT .. 5-LOW */
/* p_type (150 strings "<size> <finish> <metal>"), p_container (40 "<size> <kind>"), p_brand (25 "Brand#MN"):
 * code -> string through these; the strings are built once by tpchgen_part_dicts */
const char *const *tpchgen_part_type_dict(void);      /* 150 */
const char *const *tpchgen_part_container_dict(void); /* 40 */
const char *const *tpchgen_part_brand_dict(void);     /* 25 */

typedef struct {
    int64_t *l_orderkey;      /* BIGINT */
    int32_t *l_partkey;       /* INTEGER */
    int32_t *l_suppkey;       /* INTEGER */
    int32_t *l_linenumber;    /* INTEGER */
    int32_t *l_quantity;      /* INTEGER (ddl.sql:71) */
    int64_t *l_extendedprice; /* DECIMAL(15,2) unscaled */
    int64_t *l_discount;      /* DECIMAL(15,2) unscaled */
    int64_t *l_tax;           /* DECIMAL(15,2) unscaled */
    uint8_t *l_returnflag;    /* code into TPCHGEN_RETURNFLAG_DICT */
    uint8_t *l_linestatus;    /* code into TPCHGEN_LINESTATUS_DICT */
    int32_t *l_shipdate;      /* days since epoch */
    int32_t *l_commitdate;
    int32_t *l_receiptdate;
    uint8_t *l_shipinstruct;  /* code into TPCHGEN_SHIPINSTRUCT_DICT */
    uint8_t *l_shipmode;      /* code into TPCHGEN_SHIPMODE_DICT */
} tpchgen_lineitem_cols; /* any pointer may be NULL = column not wanted */

/* Generates the lineitem rows of orders [first_order, first_order+n_orders).
 * Returns the number of rows written (== tpchgen_lineitem_count for the same range). */
int64_t tpchgen_lineitem(int64_t sf_num, int64_t sf_den, int64_t first_order,
                         int64_t n_orders, const tpchgen_lineitem_cols *out);

typedef struct {
    int64_t *o_orderkey;
    int32_t *o_custkey;
    int32_t *o_orderdate;
    int32_t *o_shippriority;
    int64_t *o_totalprice; /* DECIMAL(15,2) unscaled */
    uint8_t *o_orderstatus; /* 'F','O','P' raw byte */
    uint8_t *o_orderpriority; /* code into TPCHGEN_ORDERPRIORITY_DICT */
    /* round 4 (appended) */
    char *o_comment;          /* TPCHGEN_O_COMMENT_STRIDE bytes per row, zero padded: 19..78 characters of the text pool */
    uint8_t *o_comment_len;
} tpchgen_orders_cols;

int64_t tpchgen_orders(int64_t sf_num, int64_t sf_den, int64_t first_order,
                       int64_t n_orders, const tpchgen_orders_cols *out);

#define TPCHGEN_S_PHONE_LEN 15        /* s_phone / c_phone: "CC-AAA-EEE-NNNN" */
#define TPCHGEN_O_COMMENT_STRIDE 80   /* o_comment: 19..78 characters */
#define TPCHGEN_C_COMMENT_STRIDE 120  /* c_comment: 29..116 */
#define TPCHGEN_S_COMMENT_STRIDE 104  /* s_comment: 25..100 */
/* The COMMENT columns are substrings of one pregenerated 300 MiB text (sentences of the specification's grammar, clause 4.2.2.14), built
 * on first use (~3 s, kept for the process lifetime, thread-safe). Returns the pool and its size. */
const char *tpchgen_text_pool(int64_t *size);
/* n_comment / r_comment of one row of the fixed NATION / REGION tables (28..115 characters into dest, which needs 116 bytes); returns the length */
int32_t tpchgen_nation_comment(int32_t nation, char *dest);
int32_t tpchgen_region_comment(int32_t region, char *dest);
typedef struct {
    int32_t *c_custkey;
    int32_t *c_nationkey;
    uint8_t *c_mktsegment; /* code into TPCHGEN_MKTSEGMENT_DICT */
    /* round 3 (appended) */
    char *c_phone;         /* 15 bytes per row: "CC-AAA-EEE-NNNN", CC = 10 + nation */
    int64_t *c_acctbal;    /* DECIMAL(15,2) unscaled, -999.99 .. 9999.99 */
    /* round 4 (appended) */
    char *c_address;          /* TPCHGEN_S_ADDRESS_STRIDE bytes per row (10..40 characters), like s_address */
    uint8_t *c_address_len;
    char *c_comment;          /* TPCHGEN_C_COMMENT_STRIDE bytes per row */
    uint8_t *c_comment_len;
} tpchgen_customer_cols;

int64_t tpchgen_customer(int64_t sf_num, int64_t sf_den, int64_t first, int64_t n,
                         const tpchgen_customer_cols *out);

typedef struct {
    int32_t *p_partkey;
    uint8_t *p_name_colors; /* 5 bytes per part: indices into TPCHGEN_COLORS; p_name is the
                               5 words joined by single blanks */
    uint8_t *p_brand;       /* code into tpchgen_part_brand_dict() */
    uint8_t *p_type;        /* code into tpchgen_part_type_dict() */
    int32_t *p_size;        /* INTEGER 1..50 */
    uint8_t *p_container;   /* code into tpchgen_part_container_dict() */
    uint8_t *p_mfgr;        /* round 4 (appended): 0..4 = "Manufacturer#1" .. "#5" */
} tpchgen_part_cols;

int64_t tpchgen_part(int64_t sf_num, int64_t sf_den, int64_t first, int64_t n,
                     const tpchgen_part_cols *out);

typedef struct {
    int32_t *ps_partkey;
    int32_t *ps_suppkey;
    int64_t *ps_supplycost; /* DECIMAL(15,2) unscaled */
    int32_t *ps_availqty;  /* INTEGER 1..9999 (round 3; appended: older callers that zero-initialise the struct keep working) */
} tpchgen_partsupp_cols;

/* first/n count PARTS; 4 partsupp rows are produced per part. Returns rows written. */
int64_t tpchgen_partsupp(int64_t sf_num, int64_t sf_den, int64_t first_part, int64_t n_parts,
                         const tpchgen_partsupp_cols *out);

#define TPCHGEN_S_ADDRESS_STRIDE 40   /* s_address: 10..40 characters, zero padded to the stride */
typedef struct {
    int32_t *s_suppkey;
    int32_t *s_nationkey;
    /* round 3 (appended: callers that zero-initialise the struct keep working) */
    char *s_address;          /* TPCHGEN_S_ADDRESS_STRIDE bytes per row */
    uint8_t *s_address_len;   /* its length */
    char *s_phone;            /* TPCHGEN_S_PHONE_LEN bytes per row, no terminator */
    /* round 4 (appended) */
    int64_t *s_acctbal;       /* DECIMAL(15,2) unscaled, -999.99 .. 9999.99 */
    char *s_comment;          /* TPCHGEN_S_COMMENT_STRIDE bytes per row, "Customer ... Complaints / Recommends" injected for 10 in 10 000 */
    uint8_t *s_comment_len;
    uint8_t *s_complaint;     /* 1 = the injected text is "Customer ... Complaints" (what Q16's LIKE selects); needs no text pool */
} tpchgen_supplier_cols;

int64_t tpchgen_supplier(int64_t sf_num, int64_t sf_den, int64_t first, int64_t n,
                         const tpchgen_supplier_cols *out);

/* days-since-epoch <-> civil date helpers (proleptic Gregorian). */
int32_t tpchgen_days_from_civil(int32_t y, int32_t m, int32_t d);
void tpchgen_civil_from_days(int32_t days, int32_t *y, int32_t *m, int32_t *d);

#ifdef __cplusplus
}
#endif
#endif

"""The oracle's result text of every reproduced TPC-H query with the constants of cases/tpch/query/q*.sql — one table, so that the SF1 golden
tests, the SF10 parity tests and bench.py's parity stamps all ask the same question. Test infrastructure (imports oracle_lib)."""
import numpy as np

import oracle_lib as O
from plan_amd import tpchgen

D = tpchgen.days
QUERIES = (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22)


def text(q, t):
    """result file text (headline + rows) of query q over the numpy tables t (tpch_data.load with text=True: Q2 / Q10 / Q13 / Q16 read COMMENT columns)"""
    if q == 1:
        return O.q1_text(O.q1(t["lineitem"], D(1998, 12, 1) - 112))
    if q == 2:
        return O.q2_text(t)
    if q == 10:
        return O.q10_text(t)
    if q == 13:
        return O.q13_text(t)
    if q == 16:
        return O.q16_text(t)
    if q == 3:
        n, rows = O.q3(t, "HOUSEHOLD", D(1995, 3, 29), cap=1 << 25)
        return O.q3_text(rows, n, 10)
    if q == 4:
        return O.q4_text(t, D(1997, 7, 1), D(1997, 10, 1))
    if q == 5:
        return O.q5_text(t, "AMERICA", D(1994, 1, 1), D(1995, 1, 1))
    if q == 6:
        lo = np.float32(0.03) - np.float32(0.01)
        hi = np.float32(0.03) + np.float32(0.01)
        rc, d = O.q6(t["lineitem"], D(1994, 1, 1), D(1995, 1, 1), lo, hi, 24)
        return O.q6_text(rc, d)
    if q == 7:
        return O.q7_text(t, "FRANCE", "ARGENTINA", D(1995, 1, 1), D(1996, 12, 31))
    if q == 8:
        return O.q8_text(t, "ARGENTINA", "AMERICA", "ECONOMY BURNISHED TIN", D(1995, 1, 1), D(1996, 12, 31))
    if q == 9:
        n, rows = O.q9(t, "%pink%")
        return O.q9_text(rows, n, tpchgen.nation_names())
    if q == 11:
        return O.q11_text(t)
    if q == 12:
        return O.q12_text(t, "FOB", "TRUCK", D(1996, 1, 1), D(1997, 1, 1))
    if q == 14:
        return O.q14_text(t, "PROMO%", D(1996, 4, 1), D(1996, 5, 1))
    if q == 15:
        return O.q15_text(t, D(1995, 12, 1), D(1996, 3, 1))
    if q == 17:
        return O.q17_text(t)
    if q == 18:
        return O.q18_text(t)
    if q == 19:
        return O.q19_text(t)
    if q == 20:
        return O.q20_text(t)
    if q == 21:
        return O.q21_text(t)
    if q == 22:
        return O.q22_text(t)
    raise ValueError(q)

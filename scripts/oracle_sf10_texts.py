"""The oracle's result text of every reproduced TPC-H query at a scale factor (default SF10), written to <outdir>/q<N>.txt with the time each
took. CPU only (test infrastructure: it runs the oracle). tests/golden/sf10/oracle_q*.txt were made by
    python scripts/oracle_sf10_texts.py tests/golden/sf10 10 1 --prefix oracle_
and tests/test_gpu_sf10_parity.py runs it again on the GPU box, next to the device runs, so the fixtures cannot go stale unnoticed.
usage: python scripts/oracle_sf10_texts.py <outdir> [sf_num sf_den] [--jobs N] [--prefix P]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_queries as OQ  # noqa: E402
import tpch_data  # noqa: E402

T = None


def one(q):
    t0 = time.time()
    text = OQ.text(q, T)
    return q, text, time.time() - t0


def main():
    global T
    args = [a for a in sys.argv[1:]]
    jobs, prefix = 1, "q"
    if "--jobs" in args:
        i = args.index("--jobs"); jobs = int(args[i + 1]); del args[i:i + 2]
    if "--prefix" in args:
        i = args.index("--prefix"); prefix = args[i + 1] + "q"; del args[i:i + 2]
    out = args[0]
    num, den = (int(args[1]), int(args[2])) if len(args) > 2 else (10, 1)
    os.makedirs(out, exist_ok=True)
    t0 = time.time()
    T = tpch_data.load(num, den, text=True)
    print(f"generated SF{num}/{den} in {time.time() - t0:.1f} s", flush=True)
    order = sorted(OQ.QUERIES, key=lambda q: q != 1)   # Q1 (a minute of chunked decimal arithmetic at SF10) first
    if jobs > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(jobs) as pool:   # the tables are shared copy-on-write; no GPU in this process
            results = pool.imap_unordered(one, order)
            for q, text, dt in results:
                print(f"q{q}: {dt:.1f} s, {text.count(chr(10)) - 1} rows", flush=True)
                open(os.path.join(out, f"{prefix}{q}.txt.tmp"), "w").write(text)
                os.replace(os.path.join(out, f"{prefix}{q}.txt.tmp"), os.path.join(out, f"{prefix}{q}.txt"))
    else:
        for q in order:
            q, text, dt = one(q)
            print(f"q{q}: {dt:.1f} s, {text.count(chr(10)) - 1} rows", flush=True)
            open(os.path.join(out, f"{prefix}{q}.txt"), "w").write(text)
    print(f"done in {time.time() - t0:.1f} s", flush=True)


if __name__ == "__main__":
    main()

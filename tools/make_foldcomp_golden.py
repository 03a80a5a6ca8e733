"""Golden vectors for the Foldcomp decoder (tests/test_foldcomp.py).

Runs the REFERENCE's own decoder — oracle/_ref/libfoldcomp_ref.so, built by oracle/Makefile from the vendored sources under
/root/reference/lib/foldcomp where they lie — over the fixture entries (tests/golden/foldcomp: the data files the reference's
own tests read, data/foldcomp/7m0y.fcz and data/foldcomp/example_db) and stores the atom records it returns.  Only runs where the
reference tree exists; the .npz it writes is committed and is what travels.

    python tools/make_foldcomp_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def main():
    oracle.build()
    if not oracle.foldcomp_ref_available():
        sys.exit("oracle/_ref/libfoldcomp_ref.so is not built (no reference tree here)")
    g = os.path.join(ROOT, "tests", "golden", "foldcomp")
    out = {}
    out["fcz_7m0y"] = oracle.foldcomp_ref_decode(open(os.path.join(g, "7m0y.fcz"), "rb").read())
    db = open(os.path.join(g, "example_db"), "rb").read()
    for line in open(os.path.join(g, "example_db.index")):
        k, s, l = (int(t) for t in line.split())
        out[f"db_{k}"] = oracle.foldcomp_ref_decode(db[s:s + l])
    np.savez_compressed(os.path.join(g, "ref_atoms.npz"), **out)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()

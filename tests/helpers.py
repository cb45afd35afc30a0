"""Shared test helpers: load golden fixtures, build the oracle for a fixture's parameters."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)

HMAT_CASES = sorted(k for k, v in MANIFEST.items() if v["mode"] == "hmat" and v.get("prec", "f64") == "f64")
F32_CASES = sorted(k for k, v in MANIFEST.items() if v["mode"] == "hmat" and v.get("prec") == "f32")
Z_CASES = sorted(k for k, v in MANIFEST.items() if v["mode"] == "hmat" and v.get("prec") in ("z64", "c32"))
LRMAT_CASES = sorted(k for k, v in MANIFEST.items() if v["mode"] == "lrmat")

DEFAULTS = dict(nsrc=0, geom="ellipse", sz=0.0, leaf=100, children=2, partitions=2, partitioning="pca_regular", eps=1e-4,
                eta=10.0, sym="N", uplo="N", compressor="partialACA", delta=1e-5, scale=1.0, mindepth=0, rank=-1,
                reqrank=-1, alpha=3.0, beta=2.0, consistent=1, prec="f64", local=-1, recompress=0, cre=1.0, cim=1.0, given="none", complete=0)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def params(name):
    p = dict(DEFAULTS)
    p.update(MANIFEST[name])
    p.setdefault("sgeom", p["geom"])
    p["dim"] = 2 if p["geom"] == "disk2d" else 3
    if p["compressor"] == "default":  # hmatrix/tree_builder/tree_builder.hpp:384-386
        p["compressor"] = "sympartialACA"
    return p


def rel_err(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)

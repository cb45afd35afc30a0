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
                reqrank=-1, alpha=3.0, beta=2.0, consistent=1, prec="f64", local=-1, recompress=0, cre=1.0, cim=1.0, given="none", complete=0,
                kernel="invdist", wavenumber=0.0)


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


def device_generator(p, T, S):
    """The fixture's generator as the device-evaluable family it belongs to (include/hmx.h hmx_kernel)."""
    import htool_amd as hm
    cplx = p["prec"] in ("z64", "c32")
    if p["kernel"] == "helmholtz":
        return hm.HelmholtzGenerator(p["dim"], T.coordinates, S.coordinates, p["wavenumber"], p["delta"], p["scale"])
    if p["kernel"] == "laplace":
        return hm.LaplaceGenerator(p["dim"], T.coordinates, S.coordinates, p["delta"], p["cre"], p["cim"] if cplx else 0.0)
    if cplx:
        return hm.InvDistGenerator(p["dim"], T.coordinates, S.coordinates, p["delta"], p["scale"], p["cre"], p["cim"], p["sym"] == "H")
    return hm.InvDistGenerator(p["dim"], T.coordinates, S.coordinates, p["delta"], p["scale"])


def rel_err(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def native_inv_dist_generator(target_coordinates, source_coordinates, delta=1e-5, scale=1.0, cre=1.0, cim=1.0, hermitian=False, dtype=np.float64, threads=0):
    """examples/host_generator.c (built by __graft_entry__.build() into examples/libhostgen.so) as an htool_amd.NativeGenerator: the user's
    VirtualGenerator as compiled host code, called by libhmx from `threads` host threads (0: all cores)."""
    import ctypes as C

    import htool_amd as hm

    class _InvDist(C.Structure):
        _fields_ = [("dim", C.c_int32), ("pad", C.c_int32), ("target", C.c_void_p), ("source", C.c_void_p), ("delta", C.c_double), ("scale", C.c_double),
                    ("cre", C.c_double), ("cim", C.c_double), ("hermitian", C.c_int32), ("pad2", C.c_int32)]

    so = os.path.join(ROOT, "examples", "libhostgen.so")
    if not os.path.exists(so):  # normally built by __graft_entry__.build()
        import subprocess
        subprocess.check_call(["gcc", "-O3", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "examples", "host_generator.c"), "-lm"])
    lib = C.CDLL(so)
    xt = np.ascontiguousarray(target_coordinates, dtype=np.float64)
    xs = np.ascontiguousarray(source_coordinates, dtype=np.float64)
    dim = xt.shape[1] if xt.ndim == 2 else 3
    g = _InvDist(dim, 0, xt.ctypes.data, xs.ctypes.data, delta, scale, cre, cim, int(bool(hermitian)), 0)
    fn = {np.dtype(np.float64): lib.hostgen_inv_dist_f64, np.dtype(np.float32): lib.hostgen_inv_dist_f32,
          np.dtype(np.complex128): lib.hostgen_inv_dist_z64, np.dtype(np.complex64): lib.hostgen_inv_dist_c32}[np.dtype(dtype)]
    return hm.NativeGenerator(fn, C.addressof(g), threads=threads, keep=(lib, g, xt, xs))

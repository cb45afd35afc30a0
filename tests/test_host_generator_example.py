"""examples/host_generator.c -- the user's VirtualGenerator as compiled host code that bench.py and the GPU tests hand to libhmx -- against the
formula it states (examples/use_hmatrix.cpp:24-35: 1 / (delta + |x - y|), column-major block), without a GPU; and hmx_host_cores, the
core count every host-side thread pool of the library uses."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

import htool_amd as hm
from helpers import ROOT, native_inv_dist_generator


def test_compiled_generator_is_the_stated_formula_bit_for_bit():
    rng = np.random.default_rng(0)
    xt, xs = rng.standard_normal((300, 3)), rng.standard_normal((200, 3))
    rows, cols = rng.permutation(300)[:37].astype(np.int32), rng.permutation(200)[:23].astype(np.int32)
    for dt, cplx in ((np.float64, False), (np.float32, False), (np.complex128, True), (np.complex64, True)):
        g = native_inv_dist_generator(xt, xs, 1e-5, 1.5, 0.3, -1.2, hermitian=cplx, dtype=dt)
        out = np.zeros((len(cols), len(rows)), dtype=dt)  # column-major M x N
        fn_t = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
        C.cast(g.function, fn_t)(g.user, len(rows), len(cols), rows.ctypes.data, cols.ctypes.data, out.ctypes.data)
        d = xt[rows][:, None, :] - xs[cols][None, :, :]
        s = np.zeros((len(rows), len(cols)))
        for p in range(3):  # squared differences summed in coordinate order from 0: the sequence the device kernel follows
            s = s + d[:, :, p] * d[:, :, p]
        den = 1e-5 + 1.5 * np.sqrt(s)
        if cplx:
            ref = 0.3 / den + 1j * ((-1.2 * np.sign(d[:, :, 0])) / den)
        else:
            ref = 1.0 / den
        assert np.array_equal(out.T, ref.astype(dt))


def test_host_cores_follow_the_cgroup_quota_and_the_override():
    n = hm.lib().hmx_host_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, -(-int(q) // int(period)))
    except OSError:
        pass
    if quota is not None:
        assert n == min(os.cpu_count() or 1, quota)
    code = "import htool_amd as hm; print(hm.lib().hmx_host_cores())"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, HMX_HOST_CORES="3", PYTHONPATH=ROOT))
    assert out.stdout.strip() == "3", out.stdout + out.stderr

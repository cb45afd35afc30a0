"""The C-level DistributedOperator (hmx_dist_*, include/hmx.h) without torch: libhmx + the system RCCL, one rank (the only size one
GPU offers; with the option force_collectives the collectives are really issued on a one-rank communicator, through both the
all-gather and the grouped-broadcast route).  Device memory through the HIP runtime directly.  Subprocess: HMX_NO_TORCH must be set
before libhmx is loaded."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    os.environ["HMX_NO_TORCH"] = "1"
    sys.path.insert(0, %r)
    import htool_amd as hm
    from htool_amd._lib import lib, check, DIST_OPTIONS
    L = lib()
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    rccl = C.CDLL("/opt/rocm/lib/librccl.so", mode=C.RTLD_GLOBAL)
    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]
    uid, comm = Uid(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    def dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0
        return p
    def host(p, like):
        out = np.empty_like(like)
        assert hip.hipMemcpy(C.c_void_p(out.ctypes.data), p, C.c_size_t(out.nbytes), 2) == 0
        return out
    n = 5000
    x3 = hm.create_geometry("ellipse", n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(64)
    T = b.create_cluster_tree(n, 3, x3, 2, 1)
    rng = np.random.default_rng(0)
    for dtype in (np.float64, np.complex128, np.float32):
        cplx = dtype == np.complex128
        tb = hm.HMatrixTreeBuilder(1e-5, 10.0, "N", "N"); tb.set_low_rank_generator("partialACA")
        H = tb.build(hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0, 1.0, 0.5 if cplx else 0.0), T, T, 0, 0, dtype=dtype)
        D = C.c_void_p()
        check(L.hmx_dist_create(H._h, T._h, T._h, comm, 0, 1, None, C.byref(D)))
        forced = "force_collectives" in OPTIONS
        for name in OPTIONS:  # per-operator switches (hmx_dist_set_option), not process state
            check(L.hmx_dist_set_option(D, DIST_OPTIONS[name], 1))
        xin = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
        y0 = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
        ab = np.array([1.5 - (0.5j if cplx else 0), 0.25 + (1j if cplx else 0)], dtype=dtype)
        pa, pb = C.c_void_p(ab.ctypes.data), C.c_void_p(ab.ctypes.data + ab.itemsize)
        for trans in ("N", "T"):
            ref = y0.copy()
            hm.internal_add_hmatrix_vector_product(trans, ab[0], H, xin, ab[1], ref)
            for fn in (L.hmx_dist_matvec_global_to_global, L.hmx_dist_matvec_local_to_local):
                dx, dy = dev(xin), dev(y0)
                check(fn(D, trans.encode(), pa, dx, pb, dy, None))
                assert hip.hipDeviceSynchronize() == 0
                y = host(dy, y0)
                err = np.linalg.norm(y - ref) / np.linalg.norm(ref)
                assert err < (1e-5 if dtype == np.float32 else 1e-13), (np.dtype(dtype).name, trans, fn.__name__, err)
                hip.hipFree(dx); hip.hipFree(dy)
        # the overlapped exchange (row-chunked expand stage, every chunk's rows exchanged on the side stream): with
        # the option force_collectives: through the real RCCL (all-gather of the chunk bounds, grouped broadcasts, events)
        ref = y0.copy()
        hm.internal_add_hmatrix_vector_product("N", ab[0], H, xin, ab[1], ref)
        for chunks in (2, 3, 0):
            check(L.hmx_dist_set_overlap(D, chunks, None))
            used = L.hmx_dist_overlap_chunks(D)
            assert used == (chunks if forced else 0), (chunks, used)
            dx, dy = dev(xin), dev(y0)
            check(L.hmx_dist_matvec_global_to_global(D, b"N", pa, dx, pb, dy, None))
            assert hip.hipDeviceSynchronize() == 0
            err = np.linalg.norm(host(dy, y0) - ref) / np.linalg.norm(ref)
            assert err < (1e-5 if dtype == np.float32 else 1e-13), (np.dtype(dtype).name, "overlap", chunks, err)
            hip.hipFree(dx); hip.hipFree(dy)
        L.hmx_dist_destroy(D)
        print("ok", np.dtype(dtype).name)
''') % ROOT


@pytest.mark.parametrize("options", [[], ["force_collectives"], ["force_collectives", "no_allgather"]])
def test_c_level_distributed_operator(options):
    e = dict(os.environ, HMX_NO_TORCH="1")
    out = subprocess.run([sys.executable, "-c", "OPTIONS = %r\n" % (options,) + SCRIPT], capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("ok ") == 3, out.stdout


MOCK_SCRIPT = textwrap.dedent('''
    import ctypes as C, os, sys, threading
    import numpy as np
    os.environ["HMX_NO_TORCH"] = "1"
    sys.path.insert(0, %r)
    import htool_amd as hm
    from htool_amd._lib import lib, check
    L = lib()
    hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    WORLD = int(sys.argv[1])
    ESZ = {7: 4, 8: 8}  # ncclFloat32, ncclFloat64 (rccl.h)

    # ---- an in-process communicator: WORLD threads, one per rank, all on this GPU.  The collectives have the shapes of
    # ---- hmx_rccl_api; `comm` carries rank + 1.  Stream order is kept by synchronising the stream first.
    barrier = threading.Barrier(WORLD)
    slots = [None] * WORLD
    calls = dict(all_gather=0, all_reduce=0, broadcast=0)
    def all_gather(send, recv, count, dtype, comm, stream):
        r, nbytes = comm - 1, count * ESZ[dtype]
        calls["all_gather"] += r == 0
        hip.hipStreamSynchronize(stream)
        slots[r] = send
        barrier.wait()
        for k in range(WORLD):
            assert hip.hipMemcpy(recv + k * nbytes, slots[k], nbytes, 3) == 0
        hip.hipStreamSynchronize(None)
        barrier.wait()
        return 0
    def all_reduce(send, recv, count, dtype, op, comm, stream):
        r, nbytes = comm - 1, count * ESZ[dtype]
        assert op == 0  # ncclSum
        calls["all_reduce"] += r == 0
        hip.hipStreamSynchronize(stream)
        slots[r] = send
        barrier.wait()
        acc = np.zeros(count, dtype=np.float64 if dtype == 8 else np.float32)
        tmp = np.empty_like(acc)
        for k in range(WORLD):  # every rank sums in the same order: identical results everywhere
            assert hip.hipMemcpy(tmp.ctypes.data, slots[k], nbytes, 2) == 0
            acc += tmp
        barrier.wait()  # everybody has read the inputs (send may alias recv)
        assert hip.hipMemcpy(recv, acc.ctypes.data, nbytes, 1) == 0
        barrier.wait()
        return 0
    def broadcast(send, recv, count, dtype, root, comm, stream):
        r, nbytes = comm - 1, count * ESZ[dtype]
        calls["broadcast"] += r == 0
        hip.hipStreamSynchronize(stream)
        slots[r] = send
        barrier.wait()
        assert hip.hipMemcpy(recv, slots[root], nbytes, 3) == 0
        hip.hipStreamSynchronize(None)
        barrier.wait()
        return 0
    def reduce_scatter(send, recv, recvcount, dtype, op, comm, stream):
        r, nbytes = comm - 1, recvcount * ESZ[dtype]
        assert op == 0
        calls["reduce_scatter"] = calls.get("reduce_scatter", 0) + (r == 0)
        hip.hipStreamSynchronize(stream)
        slots[r] = send
        barrier.wait()
        acc = np.zeros(recvcount, dtype=np.float64 if dtype == 8 else np.float32)
        tmp = np.empty_like(acc)
        for k in range(WORLD):
            assert hip.hipMemcpy(tmp.ctypes.data, slots[k] + r * nbytes, nbytes, 2) == 0
            acc += tmp
        barrier.wait()
        assert hip.hipMemcpy(recv, acc.ctypes.data, nbytes, 1) == 0
        barrier.wait()
        return 0
    # ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd: the copies happen at the group's end, when every rank has posted
    tl = threading.local()
    mail, pend = {}, [[] for _ in range(WORLD)]
    def send(buf, count, dtype, peer, comm, stream):
        r = comm - 1
        tl.rank, tl.posted = r, True
        calls["send"] = calls.get("send", 0) + (r == 0)
        hip.hipStreamSynchronize(stream)
        mail[(r, peer)] = (buf, count * ESZ[dtype])
        return 0
    def recv(buf, count, dtype, peer, comm, stream):
        r = comm - 1
        tl.rank, tl.posted = r, True
        pend[r].append((peer, buf, count * ESZ[dtype]))
        return 0
    def group_end():
        if not getattr(tl, "posted", False):
            return 0  # a group of broadcasts: they were carried out one by one
        r = tl.rank
        barrier.wait()
        for peer, buf, nbytes in pend[r]:
            src, nb = mail[(peer, r)]
            assert nb == nbytes, (peer, r, nb, nbytes)
            assert hip.hipMemcpy(buf, src, nbytes, 3) == 0
        hip.hipStreamSynchronize(None)
        barrier.wait()
        del pend[r][:]
        for key in [k for k in mail if k[0] == r]:
            del mail[key]
        tl.posted = False
        barrier.wait()
        return 0
    SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    send_fn, recv_fn = SR(send), SR(recv)
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
    RS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    rs_fn = RS(reduce_scatter)
    AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    BC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    GR = C.CFUNCTYPE(C.c_int)
    class Api(C.Structure):
        _fields_ = [("all_gather", AG), ("all_reduce", AR), ("broadcast", BC), ("group_start", GR), ("group_end", GR)]
    api = Api(AG(all_gather), AR(all_reduce), BC(broadcast), GR(lambda: 0), GR(group_end))

    def dev(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p
    def host(p, like):
        out = np.empty_like(like)
        assert hip.hipMemcpy(out.ctypes.data, p, out.nbytes, 2) == 0
        return out

    n = 3001 if WORLD == 3 else 4000  # 3 ranks: unequal parts (grouped broadcasts); 4 ranks: equal parts (all-gather)
    geom, leaf = "ball", 50
    if len(sys.argv) > 2:  # a random operator instead: size, geometry, leaf size from the seed (unequal parts whatever the rank count)
        r0 = np.random.default_rng(int(sys.argv[2]))
        n, geom, leaf = int(r0.integers(1500, 7000)), str(r0.choice(["ball", "ellipse", "disk"])), int(r0.integers(20, 120))
        print("random operator: n = %%d, %%s, leaf %%d" %% (n, geom, leaf))
    x3 = hm.create_geometry(geom, n)
    b = hm.ClusterTreeBuilder(); b.set_maximal_leaf_size(leaf)
    T = b.create_cluster_tree(n, 3, x3, 2, WORLD)
    parts = T.get_clusters_on_partition()
    assert len(parts) == WORLD
    rng = np.random.default_rng(1)
    for dtype in (np.float64, np.complex128):
        cplx = dtype == np.complex128
        tb = hm.HMatrixTreeBuilder(1e-6, 10.0, "N", "N"); tb.set_low_rank_generator("partialACA")
        gen = hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0, 1.0, 0.5 if cplx else 0.0)
        Hfull = tb.build(gen, T, T, dtype=dtype)
        Hloc = [tb.build(gen, T, T, k, k, dtype=dtype) for k in range(WORLD)]
        Ds = []
        for k in range(WORLD):
            D = C.c_void_p()
            check(L.hmx_dist_create(Hloc[k]._h, T._h, T._h, C.c_void_p(k + 1), k, WORLD, C.byref(api), C.byref(D)))
            check(L.hmx_dist_set_reduce_scatter(D, C.cast(rs_fn, C.c_void_p)))
            Ds.append(D)
        xin = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
        y0 = (rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)).astype(dtype)
        ab = np.array([1.5 - (0.5j if cplx else 0), 0.25 + (1j if cplx else 0)], dtype=dtype)
        pa, pb = C.c_void_p(ab.ctypes.data), C.c_void_p(ab.ctypes.data + ab.itemsize)
        # p2p: the output slices exchanged pairwise (grouped send / recv) instead of all-gather / grouped broadcasts
        # last field: the trans = 'N' output exchange as ncclAllReduce of the zero-padded vector (hmx_dist_set_output_collective)
        for trans, overlap, p2p, allred in (("N", 0, 0, 0), ("T", 0, 0, 0), ("N", 3, 0, 0), ("N", 2, 0, 0), ("N", 0, 1, 0), ("N", 3, 1, 0), ("T", 0, 1, 0), ("N", 0, 0, 1)):
            for D in Ds:
                check(L.hmx_dist_set_point_to_point(D, C.cast(send_fn, C.c_void_p), C.cast(recv_fn, C.c_void_p), p2p))
                check(L.hmx_dist_set_output_collective(D, allred))
                check(L.hmx_dist_set_profiling(D, 1))
            ref = y0.copy()
            hm.internal_add_hmatrix_vector_product(trans, ab[0], Hfull, xin, ab[1], ref)
            for local in ((False, True) if overlap == 0 else (False,)):
                errs, fails = [None] * WORLD, []
                def rank_main(k):
                    try:
                        off, sz = int(parts[k][0]), int(parts[k][1])
                        xi, yi = (xin[off:off + sz].copy(), y0[off:off + sz].copy()) if local else (xin, y0)
                        dx, dy = dev(xi), dev(yi)
                        fn = L.hmx_dist_matvec_local_to_local if local else L.hmx_dist_matvec_global_to_global
                        # overlap: expand stage in row chunks, each chunk exchanged on the side stream (grouped broadcasts)
                        check(L.hmx_dist_set_overlap(Ds[k], overlap, None))
                        assert L.hmx_dist_overlap_chunks(Ds[k]) == overlap
                        check(fn(Ds[k], trans.encode(), pa, dx, pb, dy, None))
                        assert hip.hipDeviceSynchronize() == 0
                        if trans == "N" and not local:  # measured: local part and exposed exchange of this product (HIP events)
                            lm, em = C.c_float(-1), C.c_float(-1)
                            check(L.hmx_dist_last_exchange_ms(Ds[k], C.byref(lm), C.byref(em)))
                            assert lm.value > 0 and em.value >= 0, (lm.value, em.value)
                        y = host(dy, yi)
                        want = ref[off:off + sz] if local else ref
                        errs[k] = np.linalg.norm(y - want) / np.linalg.norm(want)
                    except BaseException as e:
                        fails.append(e)
                        barrier.abort()
                th = [threading.Thread(target=rank_main, args=(k,)) for k in range(WORLD)]
                [t.start() for t in th]
                [t.join() for t in th]
                assert not fails, fails
                assert max(errs) < 1e-12, (np.dtype(dtype).name, trans, local, errs)
        for D in Ds:
            check(L.hmx_dist_set_output_collective(D, 0))
            check(L.hmx_dist_set_profiling(D, 0))
        # row-major multi-RHS global-to-global products (mu = 3): exchange of mu-interleaved row slices / all-reduce of the whole matrix
        # last field: the overlapped exchange for several right-hand sides (expand stage of ALL groups of right-hand sides in row chunks,
        # every chunk's mu-interleaved rows exchanged on the side stream; mu = 19: two groups, 16 + a ragged 3)
        for mu, variants in ((3, (("N", 0, 0), ("T", 0, 0), ("N", 1, 0), ("N", 0, 3), ("N", 1, 2))), (19, (("N", 0, 2), ("N", 0, 0)))):
            X = (rng.standard_normal((n, mu)) + (1j * rng.standard_normal((n, mu)) if cplx else 0)).astype(dtype)
            Y0 = (rng.standard_normal((n, mu)) + (1j * rng.standard_normal((n, mu)) if cplx else 0)).astype(dtype)
            for trans, p2p, overlap in variants:
                for D in Ds:
                    check(L.hmx_dist_set_point_to_point(D, None, None, p2p))
                ref = Y0.copy()
                hm.internal_add_hmatrix_matrix_product_row_major(trans, ab[0], Hfull, X, ab[1], ref, mu)
                errs, fails = [None] * WORLD, []
                def mm_rank(k):
                    try:
                        dx, dy = dev(X), dev(Y0)
                        check(L.hmx_dist_set_overlap(Ds[k], overlap, None))
                        for rep in range(2):  # the second call reuses the chunk rows agreed on by the first
                            dy = dev(Y0)
                            check(L.hmx_dist_matmat_row_major_global_to_global(Ds[k], trans.encode(), pa, dx, pb, dy, mu, None))
                            assert hip.hipDeviceSynchronize() == 0
                        assert L.hmx_dist_overlap_chunks_multi(Ds[k]) == overlap, (L.hmx_dist_overlap_chunks_multi(Ds[k]), overlap)
                        errs[k] = np.linalg.norm(host(dy, Y0) - ref) / np.linalg.norm(ref)
                    except BaseException as e:
                        fails.append(e)
                        barrier.abort()
                th = [threading.Thread(target=mm_rank, args=(k,)) for k in range(WORLD)]
                [t.start() for t in th]
                [t.join() for t in th]
                assert not fails, fails
                assert max(errs) < 1e-12, (np.dtype(dtype).name, "matmat", trans, mu, overlap, errs)
        for D in Ds:
            check(L.hmx_dist_set_overlap(D, 0, None))
        mu = 3
        X = (rng.standard_normal((n, mu)) + (1j * rng.standard_normal((n, mu)) if cplx else 0)).astype(dtype)
        Y0 = (rng.standard_normal((n, mu)) + (1j * rng.standard_normal((n, mu)) if cplx else 0)).astype(dtype)
        # row-major multi-RHS LOCAL-TO-LOCAL product (HPDDMOperator::GMV's call for mu != 1): all-gather of the mu-interleaved rows of X /
        # reduce-scatter of mu * n rows; against the rows of the whole operator's product and against mu single-vector local-to-local products
        for trans in ("N", "T"):
            ref = Y0.copy()
            hm.internal_add_hmatrix_matrix_product_row_major(trans, ab[0], Hfull, X, ab[1], ref, mu)
            errs, fails = [None] * WORLD, []
            def l2l_rank(k):
                try:
                    off, sz = int(parts[k][0]), int(parts[k][1])
                    Xl, Yl = np.ascontiguousarray(X[off:off + sz]), np.ascontiguousarray(Y0[off:off + sz])
                    dx, dy = dev(Xl), dev(Yl)
                    check(L.hmx_dist_matmat_row_major_local_to_local(Ds[k], trans.encode(), pa, dx, pb, dy, mu, None))
                    assert hip.hipDeviceSynchronize() == 0
                    got = host(dy, Yl)
                    e1 = np.linalg.norm(got - ref[off:off + sz]) / np.linalg.norm(ref[off:off + sz])
                    cols = np.empty_like(Yl)
                    for c in range(mu):  # the same through mu single-vector products
                        dxc, dyc = dev(np.ascontiguousarray(Xl[:, c])), dev(np.ascontiguousarray(Yl[:, c]))
                        check(L.hmx_dist_matvec_local_to_local(Ds[k], trans.encode(), pa, dxc, pb, dyc, None))
                        assert hip.hipDeviceSynchronize() == 0
                        cols[:, c] = host(dyc, np.ascontiguousarray(Yl[:, c]))
                    errs[k] = max(e1, np.linalg.norm(got - cols) / np.linalg.norm(cols))
                except BaseException as e:
                    fails.append(e)
                    barrier.abort()
            th = [threading.Thread(target=l2l_rank, args=(k,)) for k in range(WORLD)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not fails, fails
            assert max(errs) < 1e-12, (np.dtype(dtype).name, "matmat l2l", trans, errs)
            # column-major front end, partition numbering (transposition on the device around the same product)
            errs, fails = [None] * WORLD, []
            def cm_rank(k):
                try:
                    dx, dy = dev(np.asfortranarray(X).ravel("K")), dev(np.asfortranarray(Y0).ravel("K"))
                    check(L.hmx_dist_matmat_global_to_global(Ds[k], trans.encode(), pa, dx, pb, dy, mu, 0, None))
                    assert hip.hipDeviceSynchronize() == 0
                    got = host(dy, np.asfortranarray(Y0).ravel("K")).reshape(mu, n).T
                    errs[k] = np.linalg.norm(got - ref) / np.linalg.norm(ref)
                except BaseException as e:
                    fails.append(e)
                    barrier.abort()
            th = [threading.Thread(target=cm_rank, args=(k,)) for k in range(WORLD)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not fails, fails
            assert max(errs) < 1e-12, (np.dtype(dtype).name, "column-major g2g", trans, errs)
        # error behaviour of the front ends: the local USER numbering needs a permutation that is local to the partitions
        # (cluster_node.hpp:126,138 log exactly this); create_cluster_tree with several partitions does not give one.  Refused before
        # anything is exchanged, so one rank may ask alone.
        dxe, dye = dev(np.ascontiguousarray(X[:int(parts[0][1])])), dev(np.ascontiguousarray(Y0[:int(parts[0][1])]))
        rc = L.hmx_dist_matmat_local_to_local(Ds[0], b"N", pa, dxe, pb, dye, mu, 1, None)
        assert rc == -1 and b"not local" in L.hmx_last_error(), (rc, L.hmx_last_error())
        assert L.hmx_dist_matmat_global_to_global(Ds[0], b"N", pa, dxe, pb, dye, mu, 7, None) == -1  # no such numbering
        assert L.hmx_dist_gmv(Ds[0], dxe, dye, mu, int(parts[0][1]) - 1, None) == -1                  # dof below the local size
        # a DistributedOperator with BOTH kinds of operators (distributed_operator.hpp:47-53): the rank's block rows as global-to-local
        # operator plus its block-diagonal H-matrix as local-to-local operator -- every product is the sum of the two operators' products
        Hdiag = [tb.build(gen, T, T, dtype=dtype, local_partitions=(k, k)) for k in range(WORLD)]
        Dd = []
        for k in range(WORLD):
            D = C.c_void_p()
            check(L.hmx_dist_create(None, T._h, T._h, C.c_void_p(k + 1), k, WORLD, C.byref(api), C.byref(D)))
            check(L.hmx_dist_add_local_to_local_operator(D, Hdiag[k]._h))
            Dd.append(D)
            check(L.hmx_dist_add_local_to_local_operator(Ds[k], Hdiag[k]._h))
        one = np.array([1.0, 0.0], dtype=dtype)
        p1, p0 = C.c_void_p(one.ctypes.data), C.c_void_p(one.ctypes.data + one.itemsize)
        for trans in ("N", "T"):
            outs, fails = {}, []
            def both_rank(k):
                try:
                    res = []
                    for Dk in (Dd[k], None, Ds[k]):  # block-diagonal only, (rows only: from the whole operator below), both
                        if Dk is None:
                            continue
                        dx, dy = dev(xin), dev(np.zeros_like(y0))
                        check(L.hmx_dist_matvec_global_to_global(Dk, trans.encode(), p1, dx, p0, dy, None))
                        assert hip.hipDeviceSynchronize() == 0
                        res.append(host(dy, y0))
                    outs[k] = res
                except BaseException as e:
                    fails.append(e)
                    barrier.abort()
            th = [threading.Thread(target=both_rank, args=(k,)) for k in range(WORLD)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not fails, fails
            rows = np.zeros_like(y0)
            hm.internal_add_hmatrix_vector_product(trans, 1.0, Hfull, xin, 0.0, rows)
            diag = np.zeros_like(y0)
            for k in range(WORLD):
                off, sz = int(parts[k][0]), int(parts[k][1])
                yk = np.zeros(sz, dtype=dtype)
                hm.internal_add_hmatrix_vector_product(trans, 1.0, Hdiag[k], np.ascontiguousarray(xin[off:off + sz]), 0.0, yk)
                diag[off:off + sz] = yk
            for k in range(WORLD):
                e1 = np.linalg.norm(outs[k][0] - diag) / np.linalg.norm(diag)
                e2 = np.linalg.norm(outs[k][1] - (rows + diag)) / np.linalg.norm(rows + diag)
                assert e1 < 1e-12 and e2 < 1e-12, (np.dtype(dtype).name, "two operators", trans, k, e1, e2)
        # ... and VECTORS of both kinds, as the reference holds them (distributed_operator.hpp:47-53): the same two operators registered a second
        # time (hmx_dist_add_global_to_local_operator, a second hmx_dist_add_local_to_local_operator): every product doubles
        for k in range(WORLD):
            check(L.hmx_dist_add_global_to_local_operator(Ds[k], Hloc[k]._h))
            check(L.hmx_dist_add_local_to_local_operator(Ds[k], Hdiag[k]._h))
            assert L.hmx_dist_add_global_to_local_operator(Ds[k], Hdiag[k]._h) == -1 and b"whole source" in L.hmx_last_error()  # not a block row
        for trans in ("N", "T"):
            outs, fails = {}, []
            def twice_rank(k):
                try:
                    dx, dy = dev(xin), dev(y0.copy())
                    check(L.hmx_dist_matvec_global_to_global(Ds[k], trans.encode(), pa, dx, pb, dy, None))
                    assert hip.hipDeviceSynchronize() == 0
                    outs[k] = host(dy, y0)
                except BaseException as e:
                    fails.append(e)
                    barrier.abort()
            th = [threading.Thread(target=twice_rank, args=(k,)) for k in range(WORLD)]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not fails, fails
            rows = np.zeros_like(y0)
            hm.internal_add_hmatrix_vector_product(trans, 1.0, Hfull, xin, 0.0, rows)
            diag = np.zeros_like(y0)
            for k in range(WORLD):
                off, sz = int(parts[k][0]), int(parts[k][1])
                yk = np.zeros(sz, dtype=dtype)
                hm.internal_add_hmatrix_vector_product(trans, 1.0, Hdiag[k], np.ascontiguousarray(xin[off:off + sz]), 0.0, yk)
                diag[off:off + sz] = yk
            want = ab[0] * 2 * (rows + diag) + ab[1] * y0
            for k in range(WORLD):
                e = np.linalg.norm(outs[k] - want) / np.linalg.norm(want)
                assert e < 1e-12, (np.dtype(dtype).name, "vectors of operators", trans, k, e)
        for D in Ds + Dd:
            L.hmx_dist_destroy(D)
        print("ok", np.dtype(dtype).name)
    # symmetric storage: the diagonal block of every rank is fused (mirrored contributions reach rows after the expand stage), so the
    # operators cannot be chunked -- hmx_dist_set_overlap must make ALL ranks keep the single exchange, and the product stays right
    tb = hm.HMatrixTreeBuilder(1e-6, 10.0, "S", "L"); tb.set_low_rank_generator("sympartialACA")
    gen = hm.InvDistGenerator(3, x3, x3, 1e-5, 1.0)
    Hfull = tb.build(gen, T, T)
    Hloc = [tb.build(gen, T, T, k, k) for k in range(WORLD)]
    Ds = []
    for k in range(WORLD):
        D = C.c_void_p()
        check(L.hmx_dist_create(Hloc[k]._h, T._h, T._h, C.c_void_p(k + 1), k, WORLD, C.byref(api), C.byref(D)))
        Ds.append(D)
    xin, y0 = rng.standard_normal(n), rng.standard_normal(n)
    ab = np.array([1.5, 0.25])
    pa, pb = C.c_void_p(ab.ctypes.data), C.c_void_p(ab.ctypes.data + ab.itemsize)
    ref = y0.copy()
    hm.internal_add_hmatrix_vector_product("N", ab[0], Hfull, xin, ab[1], ref)
    errs, fails = [None] * WORLD, []
    def sym_rank(k):
        try:
            check(L.hmx_dist_set_overlap(Ds[k], 3, None))
            assert L.hmx_dist_overlap_chunks(Ds[k]) == 0
            dx, dy = dev(xin), dev(y0)
            check(L.hmx_dist_matvec_global_to_global(Ds[k], b"N", pa, dx, pb, dy, None))
            assert hip.hipDeviceSynchronize() == 0
            errs[k] = np.linalg.norm(host(dy, y0) - ref) / np.linalg.norm(ref)
        except BaseException as e:
            fails.append(e)
            barrier.abort()
    th = [threading.Thread(target=sym_rank, args=(k,)) for k in range(WORLD)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not fails, fails
    assert max(errs) < 1e-12, errs
    # ... and so does the product with SEVERAL right-hand sides: it runs on the stored triangle too (round 6: the default), rows receive mirrored
    # contributions after the E pass, the exchange stays whole
    mu = 5
    X, Y0 = rng.standard_normal((n, mu)), rng.standard_normal((n, mu))
    ref = Y0.copy()
    hm.internal_add_hmatrix_matrix_product_row_major("N", ab[0], Hfull, X, ab[1], ref, mu)
    errs, fails = [None] * WORLD, []
    def sym_mm_stored_rank(k):
        try:
            dx, dy = dev(X), dev(Y0)
            check(L.hmx_dist_matmat_row_major_global_to_global(Ds[k], b"N", pa, dx, pb, dy, mu, None))
            assert hip.hipDeviceSynchronize() == 0
            assert L.hmx_dist_overlap_chunks(Ds[k]) == 0 and L.hmx_dist_overlap_chunks_multi(Ds[k]) <= 1
            assert Hloc[k].stats()["expanded_bytes"] == 0
            errs[k] = np.linalg.norm(host(dy, Y0) - ref) / np.linalg.norm(ref)
        except BaseException as e:
            fails.append(e)
            barrier.abort()
    th = [threading.Thread(target=sym_mm_stored_rank, args=(k,)) for k in range(WORLD)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not fails, fails
    assert max(errs) < 1e-12, errs
    # ... but on the expanded VIEW of every rank's operator (HMX_OPT_SYM_MULTI_RHS = 0), which has an expand stage like any other, the product is
    # chunked (its own row chunks, agreed on inside the first product) and the chunks' rows are exchanged on the side stream
    for k in range(WORLD):
        Hloc[k].set_option("sym_multi_rhs", 0)
    def reset_rank(k):
        try:
            check(L.hmx_dist_set_overlap(Ds[k], 3, None))
        except BaseException as e:
            fails.append(e)
            barrier.abort()
    th = [threading.Thread(target=reset_rank, args=(k,)) for k in range(WORLD)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not fails, fails
    X, Y0 = rng.standard_normal((n, mu)), rng.standard_normal((n, mu))
    ref = Y0.copy()
    hm.internal_add_hmatrix_matrix_product_row_major("N", ab[0], Hfull, X, ab[1], ref, mu)
    errs, fails = [None] * WORLD, []
    def sym_mm_rank(k):
        try:
            dx = dev(X)
            for rep in range(2):
                dy = dev(Y0)
                check(L.hmx_dist_matmat_row_major_global_to_global(Ds[k], b"N", pa, dx, pb, dy, mu, None))
                assert hip.hipDeviceSynchronize() == 0
            assert L.hmx_dist_overlap_chunks(Ds[k]) == 0 and L.hmx_dist_overlap_chunks_multi(Ds[k]) == 3
            errs[k] = np.linalg.norm(host(dy, Y0) - ref) / np.linalg.norm(ref)
        except BaseException as e:
            fails.append(e)
            barrier.abort()
    th = [threading.Thread(target=sym_mm_rank, args=(k,)) for k in range(WORLD)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not fails, fails
    assert max(errs) < 1e-12, errs
    for D in Ds:
        L.hmx_dist_destroy(D)
    print("ok symmetric")
    print("calls", calls)
    equal = len(set(int(p[1]) for p in parts)) == 1
    assert calls["all_reduce"] > 0 and calls["broadcast"] > 0 and calls["all_gather"] > 0 and calls.get("send", 0) > 0, calls
    assert (calls.get("reduce_scatter", 0) > 0) == equal, calls  # transposed local-to-local: reduce-scatter for equal parts
''') % ROOT


@pytest.mark.parametrize("world", [3, 4])
def test_c_level_distributed_operator_multi_rank_with_mock_collectives(world):
    """hmx_dist_* with SEVERAL ranks on the one GPU a box has: every rank is a thread with its own row-restricted operator and
    hmx_dist handle, and the hmx_rccl_api table is filled with in-process collectives (barrier + device copies) instead of
    RCCL's.  Exercises what one rank cannot: the partition offsets, the all-gather route (4 equal parts) and the grouped
    broadcast route (3 unequal parts), the all-reduce of the transposed products, the reduce-scatter of the transposed local-to-local
    product (equal parts), local-to-local slices, and the overlapped exchange (hmx_dist_set_overlap: expand stage in 2 and 3 row chunks,
    every chunk's rows exchanged on the side stream); real and complex.
    Reference: the single-process product of the whole operator."""
    out = subprocess.run([sys.executable, "-c", MOCK_SCRIPT, str(world)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HMX_NO_TORCH="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("ok ") == 3 and "calls" in out.stdout, out.stdout
    print(out.stdout)


@pytest.mark.parametrize("world,seed", [(2, 5), (3, 6), (4, 7), (4, 8), (3, 9)])
def test_c_level_distributed_operator_multi_rank_on_random_operators(world, seed):
    """The same program on operators drawn from a seed (size 1 500 - 7 000, geometry, leaf size; 2, 3 and 4 ranks with unequal parts)."""
    out = subprocess.run([sys.executable, "-c", MOCK_SCRIPT, str(world), str(seed)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, HMX_NO_TORCH="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert out.stdout.count("ok ") == 3 and "calls" in out.stdout, out.stdout
    print(out.stdout)

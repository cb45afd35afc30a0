"""Host-side mirror of htool's operator interface for the hot path, over the C ABI (include/hmx.h).

Names, argument meaning and error behaviour follow the reference (paths relative to htool's
include/htool/):

  ClusterTreeBuilder            clustering/tree_builder/tree_builder.hpp:22-49
  Cluster                       clustering/cluster_node.hpp:17-82
  HMatrixTreeBuilder            hmatrix/tree_builder/tree_builder.hpp:27-274
  HMatrix                       hmatrix/hmatrix.hpp:28-245
  add_hmatrix_vector_product    hmatrix/linalg/add_hmatrix_vector_product.hpp:173-206 (user numbering)
  internal_add_hmatrix_vector_product            same file :107-170 (cluster numbering)
  internal_add_hmatrix_matrix_product_row_major  hmatrix/linalg/add_hmatrix_matrix_product_row_major.hpp:112-178

Vectors may be numpy arrays (host memory; staged over PCIe by the library) or torch CUDA tensors
(device memory; only their data_ptr() crosses the C ABI).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HmxError, check, lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def trim_device_cache():
    """Return the device buffers libhmx keeps parked for reuse (at most HMX_CACHE_GB, default 48 GB per process) to the driver,
    e.g. before another framework needs the memory."""
    check(lib().hmx_device_trim_cache())


def create_geometry(name, n, z=0.0):
    """testing/geometry.hpp: "ellipse" (create_rotated_ellipse 4:1), "disk", "ball" -- seeded mt19937(0)."""
    out = np.empty((n, 2 if name == "disk2d" else 3), dtype=np.float64)
    check(lib().hmx_geometry(name.encode(), n, z, _dp(out)))
    return out


class Cluster:
    """Root of a cluster tree (read-only view of the host structure)."""

    def __init__(self, handle, coords):
        self._h = handle
        self.coordinates = coords
        L = lib()
        self._n = L.hmx_cluster_tree_size(handle)
        nn = L.hmx_cluster_tree_num_nodes(handle)
        nodes = (_lib.ClusterNode * nn)()
        check(L.hmx_cluster_tree_nodes(handle, nodes))
        self.nodes = np.ctypeslib.as_array(nodes).copy() if nn else None
        self._nodes_struct = nodes
        p = L.hmx_cluster_tree_permutation(handle)
        self._perm = np.ctypeslib.as_array(p, shape=(self._n,)).copy()
        npart = L.hmx_cluster_tree_num_partitions(handle)
        part = np.zeros((npart, 2), dtype=np.int32)
        check(L.hmx_cluster_tree_partition(handle, part.ctypes.data_as(C.POINTER(C.c_int32))))
        self._partition = part

    # htool getters
    def get_size(self):
        return self._n

    def get_offset(self):
        return 0

    def get_permutation(self):
        return self._perm

    def _depths(self):
        d = np.zeros(4, dtype=np.int32)
        check(lib().hmx_cluster_tree_depths(self._h, d.ctypes.data_as(C.POINTER(C.c_int32))))
        return d

    def get_maximal_depth(self):
        return int(self._depths()[0])

    def get_minimal_depth(self):
        return int(self._depths()[1])

    def get_maximal_leaf_size(self):
        return int(self._depths()[2])

    def is_permutation_local(self):
        return bool(self._depths()[3])

    def get_clusters_on_partition(self):
        """(offset, size) of every partition cluster (Cluster::get_clusters_on_partition)."""
        return self._partition

    def nodes_int(self):
        s = self._nodes_struct
        return np.array([[c.depth, c.offset, c.size, c.rank, c.counter, c.n_children] for c in s], dtype=np.int32)

    def nodes_real(self):
        s = self._nodes_struct
        return np.array([[c.radius, c.center[0], c.center[1], c.center[2]] for c in s], dtype=np.float64)

    def __del__(self):
        try:  # (at interpreter shutdown the modules `lib` needs may be gone already: nothing to release then)
            if getattr(self, "_h", None):
                lib().hmx_cluster_tree_destroy(self._h)
                self._h = None
        except Exception:
            pass


def save_cluster_tree(cluster, filename):
    """clustering/cluster_output.hpp:33-84: writes <filename>_cluster_tree_properties.csv and <filename>_cluster_tree.csv,
    byte-identical to htool's files for the same tree."""
    check(lib().hmx_cluster_tree_save(cluster._h, str(filename).encode()))


def read_cluster_tree(file_cluster_tree_properties, file_cluster_tree, coordinates=None):
    """clustering/cluster_output.hpp:87-179: a cluster tree from htool's (or save_cluster_tree's) files.  The files hold
    radii and centers with 6 significant digits; like htool, the loaded tree carries those rounded values."""
    h = C.c_void_p()
    check(lib().hmx_cluster_tree_load(str(file_cluster_tree_properties).encode(), str(file_cluster_tree).encode(), C.byref(h)))
    return Cluster(h, None if coordinates is None else np.ascontiguousarray(coordinates, dtype=np.float64))


def matrix_to_bytes(mat, file):
    """matrix/utils/output.hpp:41-55: int32 rows, int32 cols, then the column-major coefficients."""
    a = np.asarray(mat)
    with open(file, "wb") as f:
        np.array(a.shape, dtype=np.int32).tofile(f)
        np.asfortranarray(a).T.tofile(f)  # .T of an F-ordered array is C-contiguous: column-major bytes of `a`


def bytes_to_matrix(file, dtype=np.float64):
    """matrix/utils/output.hpp:57-75."""
    with open(file, "rb") as f:
        rows, cols = np.fromfile(f, dtype=np.int32, count=2)
        data = np.fromfile(f, dtype=dtype, count=int(rows) * int(cols))
    return data.reshape(cols, rows).T


def save_leaves_with_rank(hmatrix, filename):
    """hmatrix/hmatrix_output.hpp:39-55: writes <filename>.csv ("nt,ns", then t_off,t_size,s_off,s_size,rank per leaf in
    htool's preorder; rank -1 = dense)."""
    if isinstance(hmatrix, BlockTree):  # structure only: pass (block_tree, ranks) via the `ranks` attribute if set
        r = getattr(hmatrix, "ranks", None)
    else:
        hmatrix.refresh_leaves()
        r = hmatrix.ranks
    ptr = None if r is None else np.ascontiguousarray(r, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int32))
    check(lib().hmx_block_tree_save_leaves_with_rank(hmatrix._bt, ptr, str(filename).encode()))


def cluster_tree_from_nodes(permutation, nodes_int, nodes_real, partition_nodes, maximal_leaf_size, coordinates, permutation_is_local=False):
    """An existing cluster tree as the engine's Cluster (hmx_cluster_tree_from_nodes): `nodes_int` / `nodes_real` are the preorder
    walk of the tree -- per node (depth, offset, size, rank, counter, number of children) and (radius, center[3]) -- the layout
    Cluster.nodes_int() / nodes_real() return; `partition_nodes[k]` is the preorder index of the cluster of partition k."""
    ni = np.ascontiguousarray(nodes_int, dtype=np.int32)
    nr = np.ascontiguousarray(nodes_real, dtype=np.float64)
    arr = (_lib.ClusterNode * len(ni))()
    for k in range(len(ni)):
        c = arr[k]
        c.depth, c.offset, c.size, c.rank, c.counter, c.n_children = (int(v) for v in ni[k])
        c.radius = float(nr[k, 0])
        for p in range(3):
            c.center[p] = float(nr[k, 1 + p])
    perm = np.ascontiguousarray(permutation, dtype=np.int32)
    parts = np.ascontiguousarray(partition_nodes, dtype=np.int32)
    x = np.ascontiguousarray(coordinates, dtype=np.float64)
    h = C.c_void_p()
    check(lib().hmx_cluster_tree_from_nodes(len(perm), x.shape[1], perm.ctypes.data_as(C.POINTER(C.c_int32)), len(ni), C.cast(arr, C.c_void_p), len(parts),
                                            parts.ctypes.data_as(C.POINTER(C.c_int32)), int(maximal_leaf_size), 1 if permutation_is_local else 0, C.byref(h)))
    return Cluster(h, x)


class ClusterTreeBuilder:
    def __init__(self):
        self._leaf = 10  # htool default (tree_builder.hpp:25)
        self._direction, self._splitting, self._n = "largest_extent", "regular", False
        self._complete = False

    def set_is_complete(self, is_complete):
        self._complete = bool(is_complete)

    def set_maximal_leaf_size(self, n):
        self._leaf = int(n)

    def set_partitioning_strategy(self, direction="largest_extent", splitting="regular", partitioning_n=False):
        """Partitioning<Direction,Splitting> or Partitioning_N<...> (clustering/implementations/partitioning.hpp)."""
        if direction not in _lib.DIRECTIONS or splitting not in _lib.SPLITTINGS:
            raise HmxError("unknown partitioning strategy")
        self._direction, self._splitting, self._n = direction, splitting, bool(partitioning_n)

    def create_cluster_tree(self, number_of_points, spatial_dimension, coordinates, number_of_children, size_of_partition,
                            radii=None, weights=None, partition=None, is_given_partition_local=False):
        x = np.ascontiguousarray(coordinates, dtype=np.float64).reshape(number_of_points, spatial_dimension)
        r = None if radii is None else np.ascontiguousarray(radii, dtype=np.float64)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        part = None if partition is None else np.ascontiguousarray(partition, dtype=np.int32)
        kind = 0 if part is None else (2 if is_given_partition_local else 1)
        if part is not None and part.size != (2 * size_of_partition if kind == 2 else number_of_points):
            raise HmxError("partition must hold one part number per point (global) or (offset, size) per part (local)")
        h = C.c_void_p()
        check(lib().hmx_cluster_tree_create_ex(number_of_points, spatial_dimension, _dp(x), None if r is None else _dp(r),
                                               None if w is None else _dp(w), self._leaf, number_of_children,
                                               size_of_partition, _lib.DIRECTIONS[self._direction],
                                               _lib.SPLITTINGS[self._splitting], int(self._n), int(self._complete),
                                               None if part is None else part.ctypes.data_as(C.POINTER(C.c_int32)), kind, C.byref(h)))
        return Cluster(h, x)

    def create_cluster_tree_from_global_partition(self, number_of_points, spatial_dimension, coordinates, number_of_children,
                                                  size_of_partition, partition):
        """tree_builder.hpp:46: partition[i] = part of point i."""
        return self.create_cluster_tree(number_of_points, spatial_dimension, coordinates, number_of_children, size_of_partition,
                                        partition=partition, is_given_partition_local=False)

    def create_cluster_tree_from_local_partition(self, number_of_points, spatial_dimension, coordinates, number_of_children,
                                                 size_of_partition, partition):
        """tree_builder.hpp:48: partition[2p], partition[2p+1] = offset and size of part p (points already grouped)."""
        return self.create_cluster_tree(number_of_points, spatial_dimension, coordinates, number_of_children, size_of_partition,
                                        partition=partition, is_given_partition_local=True)


class InvDistGenerator:
    """Device-evaluable VirtualGenerator: K(x,y) = 1 / (delta + scale * |x - y|)
    (examples/use_hmatrix.cpp:33 has delta=1e-5, scale=1; testing/generator_test.hpp:159 delta=0, scale=4*pi)."""

    def __init__(self, spatial_dimension, target_coordinates, source_coordinates, delta=1e-5, scale=1.0, cre=1.0, cim=0.0,
                 hermitian=False):
        """With complex coefficients (HMatrixTreeBuilder.build(dtype=np.complex128 | np.complex64)) the numerator is
        cre + i*cim*sgn, sgn = 1 or -- hermitian=True -- sign(x_target[0] - x_source[0]): the complex symmetric / Hermitian
        generators of testing/generator_test.hpp:163-205."""
        self.dim = spatial_dimension
        self.xt = np.ascontiguousarray(target_coordinates, dtype=np.float64)
        self.xs = np.ascontiguousarray(source_coordinates, dtype=np.float64)
        self.delta, self.scale = float(delta), float(scale)
        self.cre, self.cim, self.hermitian = float(cre), float(cim), bool(hermitian)

    kernel = _lib.HMX_KERNEL_INV_DIST

    def kernel_params(self):
        return [self.delta, self.scale, self.cre, self.cim, float(self.hermitian)]


class HelmholtzGenerator(InvDistGenerator):
    """Device-evaluable VirtualGenerator: K(x,y) = exp(i k |x - y|) / (delta + scale * |x - y|) -- the Helmholtz single layer
    exp(i k r) / (4 pi r) for delta = 0, scale = 4 pi (hmx.h: HMX_KERNEL_HELMHOLTZ).  Complex symmetric; with real coefficient
    types the real part cos(k r) / (...)."""

    kernel = _lib.HMX_KERNEL_HELMHOLTZ

    def __init__(self, spatial_dimension, target_coordinates, source_coordinates, wavenumber, delta=0.0, scale=4.0 * np.pi):
        super().__init__(spatial_dimension, target_coordinates, source_coordinates, delta, scale)
        self.wavenumber = float(wavenumber)

    def kernel_params(self):
        return [self.delta, self.scale, self.wavenumber]


class LaplaceGenerator(InvDistGenerator):
    """Device-evaluable VirtualGenerator: K(x,y) = (cre + i cim) / (4 pi (delta + |x - y|)) -- the Laplace single layer 1 / (4 pi r)
    for delta = 0 (hmx.h: HMX_KERNEL_LAPLACE_SL)."""

    kernel = _lib.HMX_KERNEL_LAPLACE_SL

    def __init__(self, spatial_dimension, target_coordinates, source_coordinates, delta=0.0, cre=1.0, cim=0.0):
        super().__init__(spatial_dimension, target_coordinates, source_coordinates, delta, 1.0, cre, cim)

    def kernel_params(self):
        return [self.delta, self.cre, self.cim]


class VirtualGenerator:
    """User-defined generator, mirror of htool's VirtualGenerator (hmatrix/interfaces/virtual_generator.hpp:17-31):
    subclass and implement copy_submatrix(M, N, rows, cols) -> array of shape (M, N) holding A[rows[j], cols[k]]
    (rows / cols are numpy int32 arrays in USER numbering).  The generator runs on the host; the engine calls it for
    one cross row / column per block and ACA iteration and for the dense leaves, everything else runs on the GPU.

    A Python generator is called from the calling thread only (htool does the same for its Python interface:
    HTOOL_WITH_PYTHON_INTERFACE, hmatrix/tree_builder/tree_builder.hpp:606).  `parallel = True` on the subclass lets the
    engine call copy_submatrix from all host threads -- worth it only when it spends its time in code that releases the
    GIL.  NativeGenerator wraps a C function pointer instead, which runs on all cores like a C++ VirtualGenerator."""

    parallel = False

    def copy_submatrix(self, M, N, rows, cols):
        raise NotImplementedError

    def _as_callback(self, prec):
        dt = _PREC[prec]["np"]
        fn_t = _lib.GENERATOR_FN_S if prec in (_lib.HMX_PREC_F32, _lib.HMX_PREC_C32) else _lib.GENERATOR_FN
        cplx = _PREC[prec]["complex"]
        rdt = np.float32 if prec in (_lib.HMX_PREC_F32, _lib.HMX_PREC_C32) else np.float64

        self._callback_error = None

        def trampoline(_user, M, N, rows, cols, out):
            r = np.ctypeslib.as_array(rows, shape=(M,))
            c = np.ctypeslib.as_array(cols, shape=(N,))
            try:
                block = np.asarray(self.copy_submatrix(M, N, r, c), dtype=dt).reshape(M, N)
            except BaseException as e:  # noqa: B902 -- cannot cross the C frames: zero block now, raised by build() afterwards
                if self._callback_error is None:
                    self._callback_error = e
                block = np.zeros((M, N), dtype=dt)
            if cplx:  # interleaved (re, im), column-major M x N
                np.ctypeslib.as_array(out, shape=(N, M, 2))[:] = np.ascontiguousarray(block.T).view(rdt).reshape(N, M, 2)
            else:
                np.ctypeslib.as_array(out, shape=(N, M))[:] = block.T  # column-major M x N
        return fn_t(trampoline)


class NativeGenerator:
    """A VirtualGenerator that is compiled code: `function` is the address of (or a ctypes pointer to) a C function
    void f(void *user, int M, int N, const int32_t *rows, const int32_t *cols, T *out) with copy_submatrix semantics
    (hmatrix/interfaces/virtual_generator.hpp:24: column-major M x N block of user-numbered rows / cols; T = the
    operator's real type, complex entries interleaved), `user` its first argument.  It is called concurrently from
    `threads` host threads (0: all cores), as htool calls a C++ generator from its OpenMP build loop.  `keep` holds
    whatever must outlive the build (the ctypes library, coordinate arrays)."""

    def __init__(self, function, user=None, threads=0, keep=None):
        self.function = C.cast(function, C.c_void_p)
        self.user = C.cast(user, C.c_void_p) if user is not None else None
        self.threads = threads
        self.keep = keep


# coefficient types: htool's HMatrix<double>, <float>, <std::complex<double>>, <std::complex<float>>
_PREC = {
    _lib.HMX_PREC_F64: dict(np=np.float64, torch="float64", sfx="", complex=False),
    _lib.HMX_PREC_F32: dict(np=np.float32, torch="float32", sfx="_s", complex=False),
    _lib.HMX_PREC_Z64: dict(np=np.complex128, torch="complex128", sfx="_z", complex=True),
    _lib.HMX_PREC_C32: dict(np=np.complex64, torch="complex64", sfx="_c", complex=True),
}


def _vec_ptr(v, A):
    """(pointer, mem kind) for a numpy array or a torch tensor of the operator's coefficient type."""
    want = _PREC[A.prec]
    if isinstance(v, np.ndarray):
        if v.dtype != want["np"] or not v.flags["C_CONTIGUOUS"]:
            raise HmxError("host vectors must be C-contiguous %s" % np.dtype(want["np"]).name)
        return v.ctypes.data, _lib.HMX_MEM_HOST
    if hasattr(v, "data_ptr"):
        import torch
        if v.dtype != getattr(torch, want["torch"]) or not v.is_contiguous():
            raise HmxError("device vectors must be contiguous %s" % want["torch"])
        return v.data_ptr(), (_lib.HMX_MEM_DEVICE if v.is_cuda else _lib.HMX_MEM_HOST)
    raise HmxError("unsupported vector type %r" % type(v))


def _coef_args(A, alpha, beta):
    """alpha / beta as the C ABI wants them: by value for real coefficients, pointers to one (re, im) pair for complex ones.
    Returns (alpha_arg, beta_arg, keepalive)."""
    if not A.complex:
        return float(alpha), float(beta), None
    ab = np.array([alpha, beta], dtype=A.dtype)
    return ab.ctypes.data, ab.ctypes.data + ab.itemsize, ab


def _fn(A, name):
    return getattr(lib(), name + _PREC[A.prec]["sfx"])


def _stream_ptr(v):
    if hasattr(v, "is_cuda") and v.is_cuda:
        import torch
        return C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
    return None


class _OwnedVector:
    """__cuda_array_interface__ view of a vector hmx_hmatrix_alloc_vector handed out (torch.as_tensor keeps this object alive with the tensor)."""

    def __init__(self, owner, ptr, shape, dt):
        self._owner, self._ptr = owner, ptr
        self.__cuda_array_interface__ = dict(shape=shape, typestr=dt.str, data=(ptr, False), version=2, strides=None)

    def __del__(self):
        try:
            if self._ptr and getattr(self._owner, "_h", None):
                lib().hmx_hmatrix_free_vector(self._owner._h, self._ptr)
        except Exception:
            pass
        self._ptr = None


class HMatrix:
    """Compressed operator resident in HBM."""

    def __init__(self, bt_handle, hm_handle, target_cluster, source_cluster):
        self._bt, self._h = bt_handle, hm_handle
        self.prec = lib().hmx_hmatrix_precision(hm_handle)
        self.f32 = self.prec == _lib.HMX_PREC_F32
        self.complex = _PREC[self.prec]["complex"]
        self.dtype = _PREC[self.prec]["np"]
        self._keep = (target_cluster, source_cluster)
        L = lib()
        r = np.zeros(4, dtype=np.int32)
        sym, uplo = C.create_string_buffer(1), C.create_string_buffer(1)
        check(L.hmx_block_tree_root(bt_handle, r.ctypes.data_as(C.POINTER(C.c_int32)), sym, uplo))
        self.target_offset, self.target_size, self.source_offset, self.source_size = [int(v) for v in r]
        self._sym, self._uplo = sym.raw.decode(), uplo.raw.decode()
        self.refresh_leaves()

    def set_option(self, name, value):
        """hmx_hmatrix_set_option: `name` is a key of htool_amd._lib.OPTIONS (the hmx_option enumerators of include/hmx.h in lower case
        without the prefix).  Product options may change between any two products; layout / build options only before the build
        (HMatrixTreeBuilder.set_option)."""
        if name not in _lib.OPTIONS:
            raise HmxError("unknown option %r (known: %s)" % (name, ", ".join(sorted(_lib.OPTIONS))))
        check(lib().hmx_hmatrix_set_option(self._h, _lib.OPTIONS[name], float(value)))

    def get_option(self, name):
        v = C.c_double(0.0)
        check(lib().hmx_hmatrix_get_option(self._h, _lib.OPTIONS[name], C.byref(v)))
        return v.value

    def release_factors(self, with_transposed=False):
        """Give the compression pool back to the device (products only need the streams); see hmx_hmatrix_release_factors."""
        check(lib().hmx_hmatrix_release_factors(self._h, int(with_transposed)))

    def prepare(self, trans="N", mu=1):
        """Build / allocate now everything products with this `trans` and this many right-hand sides need (second stream layouts, work
        vectors, staging buffers): afterwards they allocate nothing (hmx_hmatrix_prepare)."""
        check(lib().hmx_hmatrix_prepare(self._h, trans.encode(), int(mu)))

    def empty_output(self, shape, trans="N"):
        """A zero-filled torch tensor (this operator's coefficient type, on its device) for products of this operator to WRITE: any device tensor
        is a valid output, one from here lies where the operator's sweeps write fastest (hmx_hmatrix_alloc_vector).  The memory belongs to the
        operator and lives as long as the tensor or the operator, whichever ends first -- keep the operator alive while the tensor is in use."""
        import torch
        shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        dt = np.dtype(self.dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        check(lib().hmx_hmatrix_alloc_vector(self._h, trans.encode(), nbytes, C.byref(p)))
        return torch.as_tensor(_OwnedVector(self, p.value, shape, dt), device="cuda:%d" % getattr(self, "_device", 0))

    def save(self, path):
        """Binary dump of the compressed operator (hmx_hmatrix_save); reload with HMatrixTreeBuilder.load()."""
        check(lib().hmx_hmatrix_save(self._h, str(path).encode()))

    def refresh_leaves(self):
        """The leaf table and the ranks changed (build, recompression, upload): read again when next asked for (a download of the ranks --
        12 ms at N = 1e6 -- that a caller who only multiplies never needs)."""
        self._leaves = self._ranks = None

    def _load_leaves(self):
        L = lib()
        n = L.hmx_block_tree_num_leaves(self._bt)
        leaves = (_lib.Leaf * n)()
        check(L.hmx_block_tree_leaves(self._bt, leaves))
        arr = np.ctypeslib.as_array(leaves).copy() if n else np.zeros(0, dtype=[("t_offset", "<i4")])
        ranks = np.zeros(n, dtype=np.int32)
        check(L.hmx_hmatrix_leaf_ranks(self._h, ranks.ctypes.data_as(C.POINTER(C.c_int32))))
        self._leaves, self._ranks = arr, ranks

    @property
    def leaves(self):
        if self._leaves is None:
            self._load_leaves()
        return self._leaves

    @property
    def ranks(self):
        if self._ranks is None:
            self._load_leaves()
        return self._ranks

    # htool getters
    def nb_rows(self):
        return self.target_size

    def nb_cols(self):
        return self.source_size

    def get_symmetry_for_leaves(self):
        return self._sym

    def get_UPLO_for_leaves(self):
        return self._uplo

    def leaf_table(self):
        """n x 6 int table in htool's leaf order: t_off t_size s_off s_size rank(-1 dense) mirror
        (save_leaves_with_rank format, hmatrix/hmatrix_output.hpp:43-54, plus the mirror flag)."""
        a = self.leaves
        return np.stack([a["t_offset"], a["t_size"], a["s_offset"], a["s_size"], self.ranks, a["mirror"]], axis=1).astype(np.int32)

    def get_block(self, leaf):
        """(U, V) with U M x r and V r x N, or the dense M x N block."""
        a = self.leaves[leaf]
        M, N, r = int(a["t_size"]), int(a["s_size"]), int(self.ranks[leaf])
        get = _fn(self, "hmx_hmatrix_get_block")
        ptr = self._ptr
        if r >= 0:
            U, V = np.empty((r, M), dtype=self.dtype), np.empty((N, r), dtype=self.dtype)
            check(get(self._h, leaf, ptr(U), ptr(V)))
            return U.T, V.T
        D = np.empty((N, M), dtype=self.dtype)
        check(get(self._h, leaf, ptr(D), None))
        return D.T

    def get_blocks(self, leaves=None):
        """Many leaves in one bulk download (hmx_hmatrix_get_blocks: gathered on the device, a few large copies): list of (U, V) /
        dense blocks in the order of `leaves` (default: every leaf)."""
        idx = np.arange(len(self.leaves), dtype=np.int64) if leaves is None else np.ascontiguousarray(leaves, dtype=np.int64)
        keep, pu, pv = [], (C.c_void_p * len(idx))(), (C.c_void_p * len(idx))()
        for k, b in enumerate(idx):
            a = self.leaves[int(b)]
            M, N, r = int(a["t_size"]), int(a["s_size"]), int(self.ranks[int(b)])
            if r >= 0:
                U, V = np.empty((r, M), dtype=self.dtype), np.empty((N, r), dtype=self.dtype)
                keep.append((U.T, V.T))
                pu[k], pv[k] = U.ctypes.data or None, V.ctypes.data or None
                if r == 0:  # empty factors have no address: any valid pointer will do, nothing is written
                    pu[k] = pv[k] = idx.ctypes.data
            else:
                D = np.empty((N, M), dtype=self.dtype)
                keep.append(D.T)
                pu[k], pv[k] = D.ctypes.data, None
        check(_fn(self, "hmx_hmatrix_get_blocks")(self._h, len(idx), idx.ctypes.data, C.cast(pu, C.c_void_p), C.cast(pv, C.c_void_p)))
        return keep

    def _ptr(self, arr):
        if self.complex:
            return arr.ctypes.data  # void*: interleaved (re, im)
        return arr.ctypes.data_as(C.POINTER(C.c_float if self.f32 else C.c_double))

    def copy_to_dense(self):
        """copy_to_dense (hmatrix/hmatrix.hpp): the operator as a dense matrix in cluster numbering, local to the root clusters --
        every stored leaf (U V for low-rank ones) plus, for symmetric / Hermitian storage, the mirrored (conjugate) transposes;
        diagonal symmetric leaves are completed from the stored triangle like symv / hemv read them.  Host-side, for checks."""
        self.refresh_leaves()
        D = np.zeros((self.target_size, self.source_size), dtype=self.dtype)
        sym, herm, lower = self._sym != "N", self._sym == "H", self._uplo == "L"
        for b, (to, m, so, n, r, mirror) in enumerate(self.leaf_table()):
            blk = self.get_block(b)
            blk = blk[0] @ blk[1] if r >= 0 else np.array(blk)
            if sym and r < 0 and to == so and m == n:
                tri = np.tril(blk) if lower else np.triu(blk)
                off = tri - np.diag(np.diag(tri))
                blk = tri + (off.conj().T if herm else off.T)
            i, j = to - self.target_offset, so - self.source_offset
            D[i:i + m, j:j + n] = blk
            if mirror:
                i2, j2 = so - self.target_offset, to - self.source_offset
                if 0 <= i2 and i2 + n <= self.target_size and 0 <= j2 and j2 + m <= self.source_size:
                    D[i2:i2 + n, j2:j2 + m] = blk.conj().T if herm else blk.T
        return D

    def copy_to_dense_in_user_numbering(self):
        """copy_to_dense_in_user_numbering (hmatrix/hmatrix.hpp): only for an operator on the whole cluster trees."""
        tgt, src = self._keep
        if self.target_offset != 0 or self.source_offset != 0 or self.target_size != tgt.get_size() or self.source_size != src.get_size():
            raise HmxError("copy_to_dense_in_user_numbering needs an operator on the root clusters")
        D = self.copy_to_dense()
        out = np.empty_like(D)
        out[np.ix_(tgt.get_permutation(), src.get_permutation())] = D
        return out

    def set_block_lowrank(self, leaf, U, V):
        U = np.asfortranarray(U, dtype=self.dtype)
        V = np.asfortranarray(V, dtype=self.dtype)
        check(_fn(self, "hmx_hmatrix_set_block_lowrank")(self._h, leaf, U.shape[1], self._ptr(U), self._ptr(V)))

    def set_block_dense(self, leaf, D):
        D = np.asfortranarray(D, dtype=self.dtype)
        check(_fn(self, "hmx_hmatrix_set_block_dense")(self._h, leaf, self._ptr(D)))

    def recompress(self, epsilon=-1.0):
        """recompression(hmatrix) (hmatrix/utils/recompression.hpp:8-13): SVD recompression of every low-rank leaf."""
        check(lib().hmx_hmatrix_recompress(self._h, float(epsilon)))
        self.refresh_leaves()

    def finalize(self):
        check(lib().hmx_hmatrix_finalize(self._h))
        self.refresh_leaves()

    def stats(self):
        s = _lib.Stats()
        check(lib().hmx_hmatrix_stats_sized(self._h, C.byref(s), C.sizeof(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    def set_profiling(self, on):
        check(lib().hmx_hmatrix_set_profiling(self._h, int(on)))

    def last_kernel_times(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = lib().hmx_hmatrix_last_kernel_times(self._h, 32, names, ms)
        return [(names[i].decode(), float(ms[i])) for i in range(n)]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().hmx_hmatrix_destroy(self._h)
                self._h = None
            if getattr(self, "_bt", None):
                lib().hmx_block_tree_destroy(self._bt)
                self._bt = None
        except Exception:
            pass


class BlockTree:
    """Structure-only view (no GPU needed): the leaf list HMatrixTreeBuilder would build."""

    def __init__(self, handle):
        self._bt = handle
        L = lib()
        n = L.hmx_block_tree_num_leaves(handle)
        leaves = (_lib.Leaf * n)()
        check(L.hmx_block_tree_leaves(handle, leaves))
        self.leaves = np.ctypeslib.as_array(leaves).copy() if n else None
        r = np.zeros(4, dtype=np.int32)
        sym, uplo = C.create_string_buffer(1), C.create_string_buffer(1)
        check(L.hmx_block_tree_root(handle, r.ctypes.data_as(C.POINTER(C.c_int32)), sym, uplo))
        self.root = r
        self.symmetry_for_leaves, self.uplo_for_leaves = sym.raw.decode(), uplo.raw.decode()

    def __del__(self):
        try:
            if getattr(self, "_bt", None):
                lib().hmx_block_tree_destroy(self._bt)
                self._bt = None
        except Exception:
            pass


class HMatrixTreeBuilder:
    def __init__(self, epsilon, eta, symmetry, UPLO, reqrank=-1, low_rank_strategy=None):
        self._eps, self._eta, self._sym, self._uplo, self._reqrank = float(epsilon), float(eta), symmetry, UPLO, int(reqrank)
        # default compressor is sympartialACA (tree_builder.hpp:384-386)
        self._compressor = low_rank_strategy or "sympartialACA"
        self._mint = self._mins = 0
        self._consistent = True
        self._recompressed = False  # RecompressedLowRankGenerator form of the compressor
        self._adm_error = None
        self._adm = None            # user admissibility condition (ctypes thunk), None: Rjasanow-Steinbach
        self._options = {}          # engine options applied to every operator this builder creates (set_option)

    def set_low_rank_generator(self, name, recompressed=False):
        """One of the device compressors; recompressed=True wraps it like htool's RecompressedLowRankGenerator
        (hmatrix/lrmat/recompressed_low_rank_generator.hpp:12-31): every compressed block goes through SVD_recompression
        with the builder's epsilon -- the same result as HMatrix.recompress() after the build."""
        if name not in _lib.COMPRESSORS:
            raise HmxError("unknown compressor %r" % name)
        self._compressor = name
        self._recompressed = bool(recompressed)

    def set_option(self, name, value):
        """An engine option (htool_amd._lib.OPTIONS; include/hmx.h hmx_option) for every operator this builder creates: applied right after
        the operator is created, i.e. before compression and layout -- the place for layout and build options."""
        if name not in _lib.OPTIONS:
            raise HmxError("unknown option %r (known: %s)" % (name, ", ".join(sorted(_lib.OPTIONS))))
        self._options[name] = float(value)

    def set_minimal_target_depth(self, d):
        self._mint = int(d)

    def set_minimal_source_depth(self, d):
        self._mins = int(d)

    def set_block_tree_consistency(self, c):
        self._consistent = bool(c)

    def set_admissibility_condition(self, condition):
        """HMatrixTreeBuilder::set_admissibility_condition (tree_builder.hpp:243-246): `condition(target, source, eta) -> bool` with
        target / source objects exposing depth, offset, size, rank, radius and center (the fields of htool's Cluster); None restores
        the default Rjasanow-Steinbach condition.  Called on the host while the block tree is built."""
        if condition is None:
            self._adm = None
            return
        self._adm_error = None

        def thunk(_user, t, s, eta):  # an exception cannot cross the C frames: keep it and raise it after the build
            try:
                return int(bool(condition(t.contents, s.contents, eta)))
            except BaseException as e:  # noqa: B902
                if self._adm_error is None:
                    self._adm_error = e
                return 0

        self._adm = _lib.ADMISSIBILITY_FN(thunk)

    def _block_tree(self, target, source, target_partition_number, partition_number_for_symmetry):
        h = C.c_void_p()
        if self._adm is not None:
            self._adm_error = None
            check(lib().hmx_block_tree_create_adm(target._h, source._h, self._eta, self._sym.encode(), self._uplo.encode(),
                                                  self._mint, self._mins, target_partition_number, partition_number_for_symmetry,
                                                  int(self._consistent), self._adm, None, C.byref(h)))
            if self._adm_error is not None:
                lib().hmx_block_tree_destroy(h)
                err, self._adm_error = self._adm_error, None
                raise err
            return h
        check(lib().hmx_block_tree_create(target._h, source._h, self._eta, self._sym.encode(), self._uplo.encode(),
                                          self._mint, self._mins, target_partition_number, partition_number_for_symmetry,
                                          int(self._consistent), C.byref(h)))
        return h

    def _local_block_tree(self, target, source, target_partition, source_partition):
        h = C.c_void_p()
        self._adm_error = None
        check(lib().hmx_block_tree_create_local_adm(target._h, source._h, self._eta, self._sym.encode(), self._uplo.encode(), self._mint,
                                                    self._mins, target_partition, source_partition, int(self._consistent),
                                                    self._adm if self._adm is not None else _lib.ADMISSIBILITY_FN(), None, C.byref(h)))
        if self._adm_error is not None:
            lib().hmx_block_tree_destroy(h)
            err, self._adm_error = self._adm_error, None
            raise err
        return h

    def build_local_block_tree(self, target, source, target_partition, source_partition):
        return BlockTree(self._local_block_tree(target, source, target_partition, source_partition))

    def build_block_tree(self, target, source, target_partition_number=-1, partition_number_for_symmetry=-1):
        return BlockTree(self._block_tree(target, source, target_partition_number, partition_number_for_symmetry))

    def build(self, generator, target_root_cluster_tree, source_root_cluster_tree, target_partition_number=-1,
              partition_number_for_symmetry=-1, device=0, compress=True, dtype=np.float64, local_partitions=None):
        """HMatrixTreeBuilder::build (tree_builder.hpp:199-210).  With compress=False only the structure is
        created on the device and blocks are expected through HMatrix.set_block_*() + finalize()."""
        import time
        t_bt = time.perf_counter()
        if local_partitions is not None:  # rooted at (target partition, source partition): block-diagonal operator
            bt = self._local_block_tree(target_root_cluster_tree, source_root_cluster_tree, *local_partitions)
        else:
            bt = self._block_tree(target_root_cluster_tree, source_root_cluster_tree, target_partition_number,
                                  partition_number_for_symmetry)
        t_bt = time.perf_counter() - t_bt
        h = C.c_void_p()
        # HMatrix<T,double>: coefficients of type `dtype`, fp64 geometry
        prec = [k for k, v in _PREC.items() if np.dtype(v["np"]) == np.dtype(dtype)]
        if not prec:
            raise HmxError("dtype must be float64, float32, complex128 or complex64")
        t_create = time.perf_counter()
        check(getattr(lib(), "hmx_hmatrix_create" + _PREC[prec[0]]["sfx"])(bt, device, C.byref(h)))
        H = HMatrix(bt, h, target_root_cluster_tree, source_root_cluster_tree)
        H._device = int(device)
        t_create = time.perf_counter() - t_create
        H._tree_parameters = dict(eta=self._eta, epsilon=self._eps, min_target_depth=self._mint, min_source_depth=self._mins)
        H._block_tree_walltime = t_bt
        for name, value in self._options.items():
            H.set_option(name, value)
        if isinstance(generator, NativeGenerator):
            H._callback = generator  # keeps the library and the user data alive as long as the operator
            fn_t = _lib.GENERATOR_FN_S if H.prec in (_lib.HMX_PREC_F32, _lib.HMX_PREC_C32) else _lib.GENERATOR_FN
            check(_fn(H, "hmx_hmatrix_set_callback")(h, C.cast(generator.function, fn_t), generator.user))
            check(lib().hmx_hmatrix_set_callback_threads(h, int(generator.threads)))
        elif isinstance(generator, VirtualGenerator):
            H._callback = generator._as_callback(H.prec)  # keep the ctypes thunk alive as long as the operator
            check(_fn(H, "hmx_hmatrix_set_callback")(h, H._callback, None))
            check(lib().hmx_hmatrix_set_callback_threads(h, 0 if generator.parallel else 1))
        elif generator is not None:
            if not isinstance(generator, InvDistGenerator):
                raise HmxError("generator must be an InvDistGenerator / HelmholtzGenerator / LaplaceGenerator (evaluated on the device), a "
                               "VirtualGenerator subclass or a NativeGenerator (evaluated on the host through a callback)")
            params = np.array(generator.kernel_params(), dtype=np.float64)
            check(lib().hmx_hmatrix_set_kernel(h, generator.kernel, _dp(params), len(params), generator.dim, _dp(generator.xt), _dp(generator.xs)))
        H._build_walltimes = dict(block_tree=t_bt, create=t_create)
        if compress:
            t_c = time.perf_counter()
            check(lib().hmx_hmatrix_compress(h, _lib.COMPRESSORS[self._compressor], self._eps, self._reqrank))
            if getattr(generator, "_callback_error", None) is not None:  # raised inside copy_submatrix during the build
                err, generator._callback_error = generator._callback_error, None
                raise err
            H._build_walltimes["compress_call"] = time.perf_counter() - t_c
            t_c = time.perf_counter()
            H.refresh_leaves()
            H._build_walltimes["refresh_leaves"] = time.perf_counter() - t_c
            if self._recompressed:
                H.recompress()
        return H


    def load(self, path, target_root_cluster_tree, source_root_cluster_tree, target_partition_number=-1,
             partition_number_for_symmetry=-1, device=0, local_partitions=None):
        """An operator written by HMatrix.save(): the block tree is rebuilt from this builder's parameters (it must be the
        one the file was written for), the blocks come from the file instead of a compressor."""
        if local_partitions is not None:
            bt = self._local_block_tree(target_root_cluster_tree, source_root_cluster_tree, *local_partitions)
        else:
            bt = self._block_tree(target_root_cluster_tree, source_root_cluster_tree, target_partition_number,
                                  partition_number_for_symmetry)
        h = C.c_void_p()
        try:
            check(lib().hmx_hmatrix_load(bt, device, str(path).encode(), C.byref(h)))
        except HmxError:
            lib().hmx_block_tree_destroy(bt)
            raise
        H = HMatrix(bt, h, target_root_cluster_tree, source_root_cluster_tree)
        H._device = int(device)
        H._tree_parameters = dict(eta=self._eta, epsilon=self._eps, min_target_depth=self._mint, min_source_depth=self._mins)
        return H


def get_tree_parameters(hmatrix):
    """get_tree_parameters (hmatrix/hmatrix_output.hpp:85-99): the same keys, values formatted as std::to_string does."""
    p = getattr(hmatrix, "_tree_parameters", None)
    if p is None:
        raise HmxError("this HMatrix was not created by HMatrixTreeBuilder.build / load")
    T, S = hmatrix._keep
    return {
        "Eta": "%f" % p["eta"], "Epsilon": "%f" % p["epsilon"],
        "MinTargetDepth": str(p["min_target_depth"]), "MinSourceDepth": str(p["min_source_depth"]),
        "MaxClusterLeafSizeTarget": str(T.get_maximal_leaf_size()), "MaxClusterDepthTarget": str(T.get_maximal_depth()),
        "MinClusterDepthTarget": str(T.get_minimal_depth()),
        "MaxClusterLeafSizeSource": str(S.get_maximal_leaf_size()), "MaxClusterDepthSource": str(S.get_maximal_depth()),
        "MinClusterDepthSource": str(S.get_minimal_depth()),
    }


def get_hmatrix_information(hmatrix):
    """get_hmatrix_information (hmatrix/hmatrix_output.hpp:133-216): block-size / rank statistics, compression ratio and space
    saving of the stored leaves, the builder's false-positive count and the device build times.  Same keys and number
    formatting as the reference (its minima start from max(rows, columns), :145); Number_of_threads is an OpenMP figure and is
    not reported."""
    nr, nc = hmatrix.nb_rows(), hmatrix.nb_cols()
    lt = hmatrix.leaf_table()
    size = lt[:, 1].astype(np.int64) * lt[:, 3].astype(np.int64) if len(lt) else np.zeros(0, dtype=np.int64)
    lr = lt[:, 4] >= 0 if len(lt) else np.zeros(0, dtype=bool)
    dn = ~lr
    cap = max(nr, nc)
    generated = float((lt[lr, 4].astype(np.int64) * (lt[lr, 1].astype(np.int64) + lt[lr, 3])).sum() + size[dn].sum()) if len(lt) else 0.0

    def mx(v):
        return int(v.max()) if len(v) else 0

    def mn(v):
        return min(cap, int(v.min())) if len(v) else 0

    def mean(v):
        return float(v.sum()) / len(v) if len(v) else 0.0

    st = hmatrix.stats()
    info = {
        "Target_size": str(nr), "Source_size": str(nc),
        "Dense_block_size_max": str(mx(size[dn])), "Dense_block_size_mean": "%f" % mean(size[dn]), "Dense_block_size_min": str(mn(size[dn])),
        "Low_rank_block_size_max": str(mx(size[lr])), "Low_rank_block_size_mean": "%f" % mean(size[lr]), "Low_rank_block_size_min": str(mn(size[lr])),
        "Rank_max": str(mx(lt[lr, 4]) if len(lt) else 0), "Rank_mean": "%f" % (mean(lt[lr, 4].astype(np.int64)) if len(lt) else 0.0),
        "Rank_min": str(mn(lt[lr, 4]) if len(lt) else 0),
        "Number_of_low_rank_blocks": str(int(lr.sum())), "Number_of_dense_blocks": str(int(dn.sum())),
        "Compression_ratio": "%f" % ((nr * nc) / generated if generated else float("inf")),
        "Space_saving": "%f" % (1 - generated / (nr * nc)) if nr * nc else "%f" % 0.0,
        "Number_of_false_positive": str(int(st["n_false_positive"])),
        "Blocks_computation_walltime": "%f second(s)" % (st["t_compress_s"] + st["t_assemble_s"] + st["t_pack_s"]),
        "Block_tree_walltime": "%f second(s)" % getattr(hmatrix, "_block_tree_walltime", 0.0),
    }
    return info


def _print_map(title, entries, width, file):
    import sys
    out = sys.stdout if file is None else file
    out.write(title + "\n")
    for k in sorted(entries):  # std::map order
        out.write(k.ljust(width, "_") + entries[k] + "\n")


def print_tree_parameters(hmatrix, file=None):
    """print_tree_parameters (hmatrix/hmatrix_output.hpp:101-118): the text use_hmatrix.cpp prints, byte for byte."""
    _print_map("Block tree parameters", get_tree_parameters(hmatrix), 25, file)
    (__import__("sys").stdout if file is None else file).write("\n")


def print_hmatrix_information(hmatrix, file=None):
    """print_hmatrix_information (hmatrix/hmatrix_output.hpp:218-236)."""
    info = get_hmatrix_information(hmatrix)
    _print_map("Hmatrix information", info, 2 + max(len(k) for k in info), file)


def internal_add_hmatrix_vector_product(trans, alpha, A, x, beta, y):
    """y = alpha*op(A)*x + beta*y in cluster numbering (vectors local to A's root clusters)."""
    px, mx = _vec_ptr(x, A)
    py, my = _vec_ptr(y, A)
    if mx != my:
        raise HmxError("in and out must live in the same memory space")
    a, b, _keep = _coef_args(A, alpha, beta)
    check(_fn(A, "hmx_hmatrix_matvec")(A._h, trans.encode(), a, px, b, py, mx, _stream_ptr(x)))
    return y


def add_hmatrix_vector_product(trans, alpha, A, x, beta, y):
    """User-numbering front end (permutations on the device)."""
    px, mx = _vec_ptr(x, A)
    py, my = _vec_ptr(y, A)
    if mx != my:
        raise HmxError("in and out must live in the same memory space")
    a, b, _keep = _coef_args(A, alpha, beta)
    check(_fn(A, "hmx_hmatrix_matvec_user")(A._h, trans.encode(), a, px, b, py, mx, _stream_ptr(x)))
    return y


def internal_add_hmatrix_matrix_product_row_major(trans, alpha, A, X, beta, Y, mu):
    """Row-major (mu fastest) multi-RHS product in cluster numbering."""
    px, mx = _vec_ptr(X, A)
    py, my = _vec_ptr(Y, A)
    if mx != my:
        raise HmxError("in and out must live in the same memory space")
    a, b, _keep = _coef_args(A, alpha, beta)
    check(_fn(A, "hmx_hmatrix_matmat_row_major")(A._h, trans.encode(), a, px, b, py, mu, mx, _stream_ptr(X)))
    return Y


def add_hmatrix_matrix_product(transa, alpha, A, B, beta, Cm):
    """Column-major multi-RHS front end in USER numbering (hmatrix/linalg/add_hmatrix_matrix_product.hpp:176-205):
    Cm = alpha * op(A) * B + beta * Cm with B (n x mu) and Cm (m x mu) column-major (numpy Fortran order, or torch tensors whose
    .T is contiguous).  As in the reference (same file :26-77) every column is permuted to cluster numbering and the operands are
    transposed to row-major (mu fastest) -- here by one gather kernel each way on the device --, the fused row-major kernels run,
    and the result is transposed and permuted back (hmx_hmatrix_matmat_user)."""
    if isinstance(B, np.ndarray):
        if not (B.flags["F_CONTIGUOUS"] and Cm.flags["F_CONTIGUOUS"]) or B.dtype != A.dtype or Cm.dtype != A.dtype:
            raise HmxError("add_hmatrix_matrix_product takes column-major (Fortran-ordered) %s matrices" % np.dtype(A.dtype).name)
        mu = B.shape[1] if B.ndim == 2 else 1
        pb, pc, mem, stream = B.ctypes.data, Cm.ctypes.data, _lib.HMX_MEM_HOST, None
    else:  # torch: column-major n x mu == the transpose of a contiguous mu x n tensor
        if not (B.T.is_contiguous() and Cm.T.is_contiguous()):
            raise HmxError("device matrices must be column-major (the transpose of a contiguous mu x n tensor)")
        _vec_ptr(B.T, A)  # dtype check
        mu = B.shape[1]
        pb, pc = B.data_ptr(), Cm.data_ptr()
        mem, stream = (_lib.HMX_MEM_DEVICE if B.is_cuda else _lib.HMX_MEM_HOST), _stream_ptr(B)
    a, b, _keep = _coef_args(A, alpha, beta)
    check(_fn(A, "hmx_hmatrix_matmat_user")(A._h, transa.encode(), a, pb, b, pc, mu, mem, stream))
    return Cm

"""Row-partitioned DistributedOperator over torch.distributed (RCCL over xGMI on MI355X; gloo in CPU tests).

Mirror of the reference's MPI layer for the hot path (paths relative to htool's include/htool/):

  VirtualPartition / PartitionFromCluster   distributed_operator/implementations/partition_from_cluster.hpp:11-42
  VirtualGlobalToLocalOperator              distributed_operator/interfaces/virtual_global_to_local_operator.hpp:16-33
  RestrictedGlobalToLocalHMatrix            distributed_operator/implementations/global_to_local_operators/hmatrix.hpp:15-35
  DistributedOperator                       distributed_operator/distributed_operator.hpp:20-61
  DefaultApproximationBuilder               distributed_operator/utility.hpp:38-61
  internal_add_..._global_to_global         distributed_operator/linalg/add_distributed_operator_vector_product_global_to_global.hpp:18-85
  add_..._global_to_global                  same file :97-118
  internal_add_..._local_to_local           distributed_operator/linalg/add_distributed_operator_vector_product_local_to_local.hpp:19-89
  add_..._local_to_local                    same file :99-125
  ..._matrix_product_row_major_{global_to_global,local_to_local}   linalg/add_distributed_operator_matrix_product_row_major_*.hpp
  ..._matrix_product_{global_to_global,local_to_local} (column-major, user and partition numbering)
                                            linalg/add_distributed_operator_matrix_product_{global_to_global,local_to_local}.hpp
  ..._vector_sub_product_global_to_local    linalg/add_distributed_operator_vector_sub_product_global_to_local.hpp:11-21

One process per GPU; rank k owns the block rows of partition cluster k.  The MPI collectives map to
  MPI_Allgatherv (trans='N', C1/C3)  -> all_gather_into_tensor on a max-size padded buffer (RCCL has no allgatherv;
                                        partition sizes differ by at most the split remainder)
  MPI_Allreduce  (trans!='N', C2)    -> all_reduce(SUM)
  MPI_Alltoallv + axpys (C4)         -> reduce_scatter_tensor on the max-size padded layout (straight from the product's
                                        buffer when the partitions are equal); HMX_DIST_NO_REDUCE_SCATTER=1: all_reduce + slice
Vectors are torch tensors that stay on the device; only pointers cross the C ABI.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


def _rv(t):
    """Complex tensors go through the collectives as their (re, im) float view (same storage): RCCL / gloo reduce and
    gather real element types."""
    return torch.view_as_real(t) if t.is_complex() else t


class PartitionFromCluster:
    """Partition k = cluster on partition k of the cluster tree; "partition numbering" is the tree's
    cluster numbering (global_to_partition_numbering = global_to_root_cluster, cluster_node.hpp:100-119)."""

    def __init__(self, cluster):
        self._cluster = cluster
        self._part = np.asarray(cluster.get_clusters_on_partition())
        self._perm = torch.from_numpy(np.asarray(cluster.get_permutation()).astype(np.int64))
        self._perm_dev = {}

    def get_size_of_partition(self, k):
        return int(self._part[k, 1])

    def get_offset_of_partition(self, k):
        return int(self._part[k, 0])

    def get_global_size(self):
        return int(self._cluster.get_size())

    def number_of_partitions(self):
        return len(self._part)

    def _perm_on(self, device):
        key = str(device)
        if key not in self._perm_dev:
            self._perm_dev[key] = self._perm.to(device)
        return self._perm_dev[key]

    def global_to_partition_numbering(self, x):
        return x.index_select(0, self._perm_on(x.device))

    def partition_to_global_numbering(self, x, out=None):
        out = torch.empty_like(x) if out is None else out
        out.index_copy_(0, self._perm_on(x.device), x)
        return out

    def is_renumbering_local(self):
        return bool(self._cluster.is_permutation_local())

    def _local_perm_on(self, k, device):
        """perm[off_k + i] - off_k, i < size_k: positions inside partition k (local_to_local_cluster, cluster_node.hpp:136-146)."""
        if not self.is_renumbering_local():
            raise ValueError("Permutation is not local to partition, local numbering cannot be used")  # cluster_node.hpp:126,138
        key = ("loc", k, str(device))
        if key not in self._perm_dev:
            off, n = self.get_offset_of_partition(k), self.get_size_of_partition(k)
            self._perm_dev[key] = (self._perm[off:off + n] - off).to(device)
        return self._perm_dev[key]

    def local_to_local_partition_numbering(self, k, x):
        """Rows of partition k from the rank's local user numbering to partition numbering (partition_from_cluster.hpp:34-36)."""
        return x.index_select(0, self._local_perm_on(k, x.device))

    def local_partition_to_local_numbering(self, k, x, out=None):
        out = torch.empty_like(x) if out is None else out
        out.index_copy_(0, self._local_perm_on(k, x.device), x)
        return out


class RestrictedGlobalToLocalHMatrix:
    """Local (N/p) x N H-matrix of this rank as a global-to-local operator.  `in` is the whole input
    vector in partition numbering for trans='N' (the local slice for trans!='N'); `out` the local slice
    (the whole vector for trans!='N').  Calls the HIP engine through the C ABI."""

    def __init__(self, hmatrix):
        self.hmatrix = hmatrix

    def add_vector_product(self, trans, alpha, x, beta, y):
        from . import api
        api.internal_add_hmatrix_vector_product(trans, alpha, self.hmatrix, x, beta, y)

    def add_matrix_product_row_major(self, trans, alpha, X, beta, Y, mu):
        from . import api
        api.internal_add_hmatrix_matrix_product_row_major(trans, alpha, self.hmatrix, X, beta, Y, mu)

    def add_sub_matrix_product_to_local(self, X, Y, mu, offset, size):
        add_sub_matrix_product_to_local(self, self.hmatrix.source_offset, self.hmatrix.source_size, X, Y, mu, offset, size)


class LocalToLocalHMatrix:
    """H-matrix on (target partition k) x (source partition k) as a local-to-local operator
    (distributed_operator/implementations/local_to_local_operators/hmatrix.hpp:15-56): local slices in and out."""

    def __init__(self, hmatrix):
        self.hmatrix = hmatrix

    def add_vector_product(self, trans, alpha, x, beta, y):
        from . import api
        api.internal_add_hmatrix_vector_product(trans, alpha, self.hmatrix, x, beta, y)

    def add_matrix_product_row_major(self, trans, alpha, X, beta, Y, mu):
        from . import api
        api.internal_add_hmatrix_matrix_product_row_major(trans, alpha, self.hmatrix, X, beta, Y, mu)

    def add_sub_matrix_product_to_local(self, X, Y, mu, offset, size):
        add_sub_matrix_product_to_local(self, self.hmatrix.source_offset, self.hmatrix.source_size, X, Y, mu, offset, size)


def add_sub_matrix_product_to_local(op, source_offset, source_size, X, Y, mu, offset, size):
    """Y += op * (X zero-extended): X holds rows [offset, offset+size) of the source numbering, row-major with mu
    columns (restricted_operator.hpp:170-193, local_to_local_operators/hmatrix.hpp:33-51)."""
    lo, hi = max(offset, source_offset), min(offset + size, source_offset + source_size)
    if offset == source_offset and hi == source_offset + source_size and size == source_size:
        op.add_matrix_product_row_major("N", 1.0, X, 1.0, Y, mu)
        return
    ext = torch.zeros((source_size, mu), dtype=X.dtype, device=X.device)
    if hi > lo:
        ext[lo - source_offset:hi - source_offset] = X[lo - offset:hi - offset]
    op.add_matrix_product_row_major("N", 1.0, ext, 1.0, Y, mu)


class DistributedOperator:
    def __init__(self, target_partition, source_partition, group=None, output_collective=None):
        """output_collective: how the disjoint output slices of a trans='N' product reach every rank -- "allgather" (default;
        the reference's MPI_Allgatherv: every rank sends its N/p slice) or "allreduce" (a zero-padded length-N vector summed
        over the ranks: p times the bytes, one collective fewer to tune; HMX_DIST_COLLECTIVE sets the default)."""
        self.target_partition, self.source_partition = target_partition, source_partition
        self.group = group
        import os
        self.output_collective = output_collective or os.environ.get("HMX_DIST_COLLECTIVE", "allgather")
        if self.output_collective not in ("allgather", "allreduce"):
            raise ValueError("output_collective must be 'allgather' or 'allreduce'")
        self.global_to_local_operators = []
        self.local_to_local_operators = []
        self._pad = {}

    def add_global_to_local_operator(self, op):
        self.global_to_local_operators.append(op)

    def add_local_to_local_operator(self, op):
        self.local_to_local_operators.append(op)

    # communicator
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _gather_slices(self, local, partition, out):
        """out[offset_k : offset_k + size_k] = rank k's `local` (MPI_Allgatherv)."""
        p = self.size()
        if p == 1:
            out.copy_(local)
            return
        if self.output_collective == "allreduce":
            off = partition.get_offset_of_partition(self.rank())
            out.zero_()
            out[off:off + local.shape[0]].copy_(local)
            dist.all_reduce(_rv(out), op=dist.ReduceOp.SUM, group=self.group)
            return
        sizes = [partition.get_size_of_partition(k) for k in range(p)]
        m = max(sizes)
        if min(sizes) == m and out.is_contiguous() and partition.get_offset_of_partition(0) == 0:
            # equal partitions (e.g. N = 1e6 over 8 GPUs): gather straight into the output, no staging copies
            dist.all_gather_into_tensor(_rv(out), _rv(local.contiguous()), group=self.group)
            return
        key = (m, p, local.device, local.dtype, tuple(local.shape[1:]))
        if key not in self._pad:
            self._pad[key] = (torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device),
                              torch.empty((p * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device))
        send, recv = self._pad[key]
        send[:local.shape[0]].copy_(local)
        dist.all_gather_into_tensor(_rv(recv), _rv(send), group=self.group)
        for k in range(p):
            out[partition.get_offset_of_partition(k):partition.get_offset_of_partition(k) + sizes[k]].copy_(recv[k * m:k * m + sizes[k]])


    def _reduce_scatter_slices(self, full, partition):
        """Sum of the ranks' `full` (global length, partition numbering; first dimension = rows), returned restricted to THIS
        rank's slice: the reference's MPI_Alltoallv + p axpys (add_distributed_operator_vector_product_local_to_local.hpp:77,
        ..._matrix_product_row_major_local_to_local.hpp:84) as ONE reduce-scatter -- every rank receives N/p rows instead of N."""
        import os
        p, rank = self.size(), self.rank()
        off, n = partition.get_offset_of_partition(rank), partition.get_size_of_partition(rank)
        if p == 1:
            return full[off:off + n]
        if os.environ.get("HMX_DIST_NO_REDUCE_SCATTER"):
            dist.all_reduce(_rv(full), op=dist.ReduceOp.SUM, group=self.group)
            return full[off:off + n]
        sizes = [partition.get_size_of_partition(k) for k in range(p)]
        m = max(sizes)
        tail = tuple(full.shape[1:])
        out = torch.empty((m,) + tail, dtype=full.dtype, device=full.device)
        if min(sizes) == m and partition.get_offset_of_partition(0) == 0 and full.is_contiguous():
            dist.reduce_scatter_tensor(_rv(out), _rv(full), op=dist.ReduceOp.SUM, group=self.group)
            return out
        key = ("rs", m, p, full.device, full.dtype, tail)
        if key not in self._pad:
            self._pad[key] = torch.zeros((p * m,) + tail, dtype=full.dtype, device=full.device)
        send = self._pad[key]
        for k in range(p):  # rows beyond size_k stay zero
            send[k * m:k * m + sizes[k]].copy_(full[partition.get_offset_of_partition(k):partition.get_offset_of_partition(k) + sizes[k]])
        dist.reduce_scatter_tensor(_rv(out), _rv(send), op=dist.ReduceOp.SUM, group=self.group)
        return out[:n]


def internal_add_distributed_operator_vector_product_global_to_global(trans, alpha, A, x, beta, y):
    """Partition numbering; x and y are whole vectors replicated on every rank."""
    rank = A.rank()
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    off_in, n_in = in_part.get_offset_of_partition(rank), in_part.get_size_of_partition(rank)
    off_out, n_out = out_part.get_offset_of_partition(rank), out_part.get_size_of_partition(rank)
    if trans == "N":
        local = y[off_out:off_out + n_out].clone() if beta != 0 else torch.zeros(n_out, dtype=y.dtype, device=y.device)
        apply_beta = True
        for op in A.global_to_local_operators:
            op.add_vector_product(trans, alpha, x, beta if apply_beta else 1.0, local)
            apply_beta = False
        for op in A.local_to_local_operators:
            op.add_vector_product(trans, alpha, x[off_in:off_in + n_in], beta if apply_beta else 1.0, local)
            apply_beta = False
        A._gather_slices(local, out_part, y)
    else:
        y_old = y.clone() if beta != 0 else None
        y.zero_()
        x_loc = x[off_in:off_in + n_in].contiguous()
        apply_beta = True
        for op in A.global_to_local_operators:
            op.add_vector_product(trans, alpha, x_loc, beta if apply_beta else 1.0, y)
            apply_beta = False
        for op in A.local_to_local_operators:
            op.add_vector_product(trans, alpha, x_loc, beta if apply_beta else 1.0, y[off_out:off_out + n_out])
            apply_beta = False
        if A.size() > 1:
            dist.all_reduce(_rv(y), op=dist.ReduceOp.SUM, group=A.group)
        if beta != 0:
            y.add_(y_old, alpha=beta)
    return y


class GraphedGlobalToGlobalProduct:
    """y = alpha * A * x (trans='N', beta = 0, partition numbering) with this rank's kernels captured ONCE in a HIP graph.

    At 8 GPUs a rank's share of an N=1e6 product is a few hundred microseconds spread over half a dozen launches, so the
    step is launch-bound; replaying a graph removes the per-launch cost.  Only the LOCAL work is captured (zeroing the
    local slice + the operators' kernels); the collective (all-gather of the slices) is issued eagerly after the replay,
    so no rank ever replays a graph that holds a collective.  x and y are bound at construction (graphs replay fixed
    pointers): write new input into `self.x` and read `self.y`."""

    def __init__(self, A, x, y, alpha=1.0):
        self.A, self.x, self.y = A, x, y
        rank = A.rank()
        self.out_part = A.target_partition
        off_in, n_in = A.source_partition.get_offset_of_partition(rank), A.source_partition.get_size_of_partition(rank)
        self.local = torch.zeros(self.out_part.get_size_of_partition(rank), dtype=y.dtype, device=y.device)

        def local_work():
            first = True  # the first operator overwrites the slice (beta = 0): no memset, no read of the old values
            for op in A.global_to_local_operators:
                op.add_vector_product("N", alpha, x, 0.0 if first else 1.0, self.local)
                first = False
            for op in A.local_to_local_operators:
                op.add_vector_product("N", alpha, x[off_in:off_in + n_in], 0.0 if first else 1.0, self.local)
                first = False
            if first:
                self.local.zero_()

        self._eager = local_work
        self.graph = None
        local_work()  # warm-up outside the capture (lazy allocations inside the engine)
        torch.cuda.synchronize()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                local_work()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: only this thread's calls are checked against the capture, so the process group's
            # watchdog thread (event queries on the RCCL streams) cannot invalidate it
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                local_work()
            torch.cuda.synchronize()
            ref = self.local.clone()
            self.local.fill_(7)
            g.replay()
            torch.cuda.synchronize()
            if torch.equal(self.local, ref):
                self.graph = g
        except Exception:  # capture not available: keep launching eagerly
            self.graph = None
            torch.cuda.synchronize()

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._eager()
        self.A._gather_slices(self.local, self.out_part, self.y)
        return self.y


def add_distributed_operator_vector_product_global_to_global(trans, alpha, A, x, beta, y):
    """User numbering: permute to partition numbering, multiply, permute back."""
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    xp = in_part.global_to_partition_numbering(x)
    yp = out_part.global_to_partition_numbering(y) if beta != 0 else torch.zeros_like(y)
    internal_add_distributed_operator_vector_product_global_to_global(trans, alpha, A, xp, beta, yp)
    out_part.partition_to_global_numbering(yp, out=y)
    return y


def internal_add_distributed_operator_vector_product_local_to_local(trans, alpha, A, x_loc, beta, y_loc):
    """The Krylov-side contract (wrappers/wrapper_hpddm.hpp:121): local slices in and out, partition numbering.
    trans='N': all-gather the input, local product, output stays local.  trans!='N': local product into a
    zeroed global buffer, reduce-scatter of that buffer (DistributedOperator._reduce_scatter_slices)."""
    rank = A.rank()
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    if trans == "N":
        apply_beta = True
        for op in A.local_to_local_operators:
            op.add_vector_product(trans, alpha, x_loc, beta if apply_beta else 1.0, y_loc)
            apply_beta = False
        if A.global_to_local_operators:
            x = torch.empty(in_part.get_global_size(), dtype=x_loc.dtype, device=x_loc.device)
            A._gather_slices(x_loc, in_part, x)  # local_to_global, linalg/utility.hpp:11-28
            for op in A.global_to_local_operators:
                op.add_vector_product(trans, alpha, x, beta if apply_beta else 1.0, y_loc)
                apply_beta = False
    else:
        if beta != 1:
            y_loc.mul_(beta)
        for op in A.local_to_local_operators:
            op.add_vector_product(trans, alpha, x_loc, 1.0, y_loc)
        if A.global_to_local_operators:
            buf = torch.zeros(out_part.get_global_size(), dtype=x_loc.dtype, device=x_loc.device)
            for op in A.global_to_local_operators:
                op.add_vector_product(trans, alpha, x_loc, 1.0, buf)
            y_loc.add_(A._reduce_scatter_slices(buf, out_part))  # MPI_Alltoallv + p axpys == a reduce-scatter
    return y_loc


# ---- the same products below Python: libhmx's hmx_dist_* (include/hmx.h) over an RCCL communicator of its own ------------------------
# At 8 GPUs a rank's share of an N=1e6 product is ~0.4 ms of kernels: interpreter + torch.distributed dispatch per collective is then a
# visible part of every step.  NativeDistributedOperator issues ONE C call per product; the local kernels, the events and the
# collectives (optionally overlapped on a side stream, hmx_dist_set_overlap) are enqueued by libhmx.

_NCCL_DTYPE_SIZE = {7: 4, 8: 8}  # ncclFloat32, ncclFloat64 (rccl.h)


class _RcclApi(C.Structure):  # hmx_rccl_api
    _fields_ = [("all_gather", C.c_void_p), ("all_reduce", C.c_void_p), ("broadcast", C.c_void_p), ("group_start", C.c_void_p), ("group_end", C.c_void_p)]


class NativeCommunicator:
    """ncclComm_t created through librccl directly (ncclGetUniqueId on rank 0, shipped over the existing torch.distributed group,
    ncclCommInitRank everywhere).  `api` is the hmx_rccl_api table of THAT library, so libhmx calls the instance the communicator
    belongs to.  backend="gloo": no RCCL at all -- the table holds host-staged collectives over the torch.distributed group (several
    ranks sharing one GPU in tests; every call synchronises the stream it is given)."""

    def __init__(self, group=None, backend="rccl", lib_path=None):
        import os
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = backend
        self.comm = C.c_void_p(self.rank + 1)
        self.api = _RcclApi()
        self.reduce_scatter = None
        self.send = self.recv = None  # ncclSend / ncclRecv (hmx_dist_set_point_to_point)
        self._keep = []
        if backend == "rccl":
            path = lib_path or os.environ.get("HMX_RCCL_LIB")
            if not path:  # the RCCL torch itself loaded, when there is one: one instance in the process
                cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
                path = cand if os.path.exists(cand) else "librccl.so"
            self.lib = C.CDLL(path, mode=C.RTLD_GLOBAL)

            class Uid(C.Structure):
                _fields_ = [("b", C.c_char * 128)]
            uid = Uid()
            if self.rank == 0 and self.lib.ncclGetUniqueId(C.byref(uid)) != 0:
                raise RuntimeError("ncclGetUniqueId failed")
            box = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
            C.memmove(C.addressof(uid), box[0], 128)
            self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
            comm = C.c_void_p()
            rc = self.lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)
            if rc != 0:
                raise RuntimeError("ncclCommInitRank failed: %d" % rc)
            self.comm = comm
            for field, sym in (("all_gather", "ncclAllGather"), ("all_reduce", "ncclAllReduce"), ("broadcast", "ncclBroadcast"),
                               ("group_start", "ncclGroupStart"), ("group_end", "ncclGroupEnd")):
                setattr(self.api, field, C.cast(getattr(self.lib, sym), C.c_void_p).value)
            self.reduce_scatter = C.cast(self.lib.ncclReduceScatter, C.c_void_p)
            self.send, self.recv = C.cast(self.lib.ncclSend, C.c_void_p), C.cast(self.lib.ncclRecv, C.c_void_p)
        else:
            self._make_host_staged_table()

    def count(self):
        """Number of ranks AS THE COMMUNICATOR REPORTS IT (ncclCommCount on the native RCCL communicator; the process group's size
        for the host-staged table)."""
        if self.backend != "rccl":
            return self.world
        n = C.c_int(-1)
        self.lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        if self.lib.ncclCommCount(self.comm, C.byref(n)) != 0:
            raise RuntimeError("ncclCommCount failed")
        return int(n.value)

    def destroy(self):
        if self.backend == "rccl" and self.comm:
            self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None

    def _make_host_staged_table(self):
        hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        group = self.group

        def np_dtype(dt):
            return np.float64 if dt == 8 else np.float32

        def fetch(ptr, count, dt):
            a = np.empty(count, dtype=np_dtype(dt))
            assert hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2) == 0
            return torch.from_numpy(a)

        def store(ptr, t):
            a = np.ascontiguousarray(t.numpy())
            assert hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, 1) == 0
            # a copy from pageable memory may return once the data is staged: the kernels that read `ptr` run on non-blocking streams, which
            # the null stream does not order -- wait for the transfer itself (a rare wrong product in the two-process tests otherwise)
            hip.hipStreamSynchronize(None)

        def all_gather(send, recv, count, dt, comm, stream):
            hip.hipStreamSynchronize(stream)
            mine = fetch(send, count, dt)
            out = torch.empty(count * self.world, dtype=mine.dtype)
            dist.all_gather_into_tensor(out, mine, group=group)
            store(recv, out)
            return 0

        def all_reduce(send, recv, count, dt, op, comm, stream):
            hip.hipStreamSynchronize(stream)
            t = fetch(send, count, dt)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            store(recv, t)
            return 0

        def broadcast(send, recv, count, dt, root, comm, stream):
            hip.hipStreamSynchronize(stream)
            t = fetch(send, count, dt) if self.rank == root else torch.empty(count, dtype=torch.float64 if dt == 8 else torch.float32)
            dist.broadcast(t, src=dist.get_global_rank(group, root) if group is not None else root, group=group)
            store(recv, t)
            return 0

        def reduce_scatter(send, recv, recvcount, dt, op, comm, stream):
            hip.hipStreamSynchronize(stream)
            t = fetch(send, recvcount * self.world, dt)
            out = torch.empty(recvcount, dtype=t.dtype)
            dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM, group=group)
            store(recv, out)
            return 0

        # ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd: posted at the group's end (batch_isend_irecv), so that every rank
        # can name all its sends before any receive completes
        pending = []

        def send(buf, count, dt, peer, comm, stream):
            hip.hipStreamSynchronize(stream)
            pending.append((dist.isend, fetch(buf, count, dt), peer, None))
            return 0

        def recv(buf, count, dt, peer, comm, stream):
            pending.append((dist.irecv, torch.empty(count, dtype=torch.float64 if dt == 8 else torch.float32), peer, buf))
            return 0

        def group_end():
            if pending:
                ops = [dist.P2POp(op, t, dist.get_global_rank(group, peer) if group is not None else peer, group) for op, t, peer, _ in pending]
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
                for op, t, _, buf in pending:
                    if buf is not None:
                        store(buf, t)
                del pending[:]
            return 0

        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
        SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
        GR = C.CFUNCTYPE(C.c_int)
        fns = [AG(all_gather), AR(all_reduce), AR(broadcast), GR(lambda: 0), GR(group_end), AR(reduce_scatter), SR(send), SR(recv)]
        self.send, self.recv = C.cast(fns[6], C.c_void_p), C.cast(fns[7], C.c_void_p)
        self._keep = fns
        for field, f in zip(("all_gather", "all_reduce", "broadcast", "group_start", "group_end"), fns):
            setattr(self.api, field, C.cast(f, C.c_void_p).value)
        self.reduce_scatter = C.cast(fns[5], C.c_void_p)


class NativeDistributedOperator:
    """hmx_dist_* (include/hmx.h): the row-restricted H-matrix of this rank + a communicator; products take torch tensors on the
    device, partition numbering.  set_overlap(chunks) is collective (hmx_dist_set_overlap)."""

    def __init__(self, hmatrix, target_cluster, source_cluster, communicator, block_diagonal_hmatrix=None):
        """hmatrix: the rank's row-restricted H-matrix (global-to-local operator) or None; block_diagonal_hmatrix: an H-matrix on
        (target partition rank) x (source partition rank) registered as a local-to-local operator
        (hmx_dist_add_local_to_local_operator; DefaultLocalApproximationBuilder's operator)."""
        from ._lib import check, lib
        self.hmatrix, self.comm, self.block_diagonal_hmatrix = hmatrix, communicator, block_diagonal_hmatrix
        self._L = lib()
        self._h = C.c_void_p()
        check(self._L.hmx_dist_create(hmatrix._h if hmatrix is not None else None, target_cluster._h, source_cluster._h, communicator.comm, communicator.rank, communicator.world,
                                      C.byref(communicator.api), C.byref(self._h)))
        if block_diagonal_hmatrix is not None:
            check(self._L.hmx_dist_add_local_to_local_operator(self._h, block_diagonal_hmatrix._h))
        if communicator.reduce_scatter is not None:
            check(self._L.hmx_dist_set_reduce_scatter(self._h, communicator.reduce_scatter))
        if communicator.send is not None:  # known, not in use: set_point_to_point(True) switches the slice exchange over
            check(self._L.hmx_dist_set_point_to_point(self._h, communicator.send, communicator.recv, 0))
        self._scal = {}
        self._more = []  # operators registered later: kept alive with this object (the C layer keeps references)

    def add_global_to_local_operator(self, hmatrix):
        """DistributedOperator::add_global_to_local_operator (distributed_operator.hpp:47-49): one more (target partition rank) x (whole source)
        operator; every product adds it after the first (hmx_dist_add_global_to_local_operator)."""
        from ._lib import check
        check(self._L.hmx_dist_add_global_to_local_operator(self._h, hmatrix._h))
        self._more.append(hmatrix)

    def add_local_to_local_operator(self, block_diagonal_hmatrix):
        """DistributedOperator::add_local_to_local_operator (distributed_operator.hpp:50-53): may be called several times."""
        from ._lib import check
        check(self._L.hmx_dist_add_local_to_local_operator(self._h, block_diagonal_hmatrix._h))
        self._more.append(block_diagonal_hmatrix)

    def __del__(self):
        try:
            if self._h:
                self._L.hmx_dist_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _scalars(self, alpha, beta, dtype):
        key = (complex(alpha), complex(beta), dtype)
        if key not in self._scal:
            npdt = {torch.float64: np.float64, torch.float32: np.float32, torch.complex128: np.complex128, torch.complex64: np.complex64}[dtype]
            ab = np.array([alpha, beta], dtype=npdt)
            self._scal[key] = (ab, C.c_void_p(ab.ctypes.data), C.c_void_p(ab.ctypes.data + ab.itemsize))
        return self._scal[key][1:]

    @staticmethod
    def _stream(t):
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def set_overlap(self, chunks, like=None):
        from ._lib import check
        st = self._stream(like) if like is not None else C.c_void_p(torch.cuda.current_stream().cuda_stream)
        check(self._L.hmx_dist_set_overlap(self._h, int(chunks), st))
        return int(self._L.hmx_dist_overlap_chunks(self._h))

    def overlap_chunks_multi(self):
        """Row chunks in use for trans='N' products with SEVERAL right-hand sides (hmx_dist_overlap_chunks_multi): agreed on inside the first
        such product after set_overlap; 0 before that, or when some rank's multi-RHS layout cannot be chunked."""
        return int(self._L.hmx_dist_overlap_chunks_multi(self._h))

    def set_output_collective(self, all_reduce):
        """trans='N' global-to-global products: exchange of the output slices (False, default) or ncclAllReduce of the zero-padded
        output vector (True): hmx_dist_set_output_collective.  Every rank must choose the same."""
        from ._lib import check
        check(self._L.hmx_dist_set_output_collective(self._h, 1 if all_reduce else 0))

    def set_profiling(self, on):
        from ._lib import check
        check(self._L.hmx_dist_set_profiling(self._h, 1 if on else 0))

    def last_exchange_ms(self):
        """(local_ms, exposed_ms) of the last profiled trans='N' global-to-global product: HIP events on the caller's stream
        (hmx_dist_last_exchange_ms)."""
        from ._lib import check
        a, b = C.c_float(0), C.c_float(0)
        check(self._L.hmx_dist_last_exchange_ms(self._h, C.byref(a), C.byref(b)))
        return float(a.value), float(b.value)

    def set_point_to_point(self, enable):
        """Output slices exchanged pairwise (grouped ncclSend / ncclRecv) instead of all-gather / grouped broadcasts
        (hmx_dist_set_point_to_point); every rank must choose the same."""
        from ._lib import check
        check(self._L.hmx_dist_set_point_to_point(self._h, None, None, 1 if enable else 0))

    def matvec_global_to_global(self, trans, alpha, x, beta, y):
        from ._lib import check
        pa, pb = self._scalars(alpha, beta, y.dtype)
        check(self._L.hmx_dist_matvec_global_to_global(self._h, trans.encode(), pa, C.c_void_p(x.data_ptr()), pb, C.c_void_p(y.data_ptr()), self._stream(y)))
        return y

    def matmat_row_major_global_to_global(self, trans, alpha, X, beta, Y, mu):
        from ._lib import check
        pa, pb = self._scalars(alpha, beta, Y.dtype)
        check(self._L.hmx_dist_matmat_row_major_global_to_global(self._h, trans.encode(), pa, C.c_void_p(X.data_ptr()), pb, C.c_void_p(Y.data_ptr()), int(mu), self._stream(Y)))
        return Y

    def matvec_local_to_local(self, trans, alpha, x_loc, beta, y_loc):
        from ._lib import check
        pa, pb = self._scalars(alpha, beta, y_loc.dtype)
        check(self._L.hmx_dist_matvec_local_to_local(self._h, trans.encode(), pa, C.c_void_p(x_loc.data_ptr()), pb, C.c_void_p(y_loc.data_ptr()), self._stream(y_loc)))
        return y_loc

    def matmat_row_major_local_to_local(self, trans, alpha, X_loc, beta, Y_loc, mu):
        """hmx_dist_matmat_row_major_local_to_local: the Krylov-side block product (HPDDMOperator::GMV for mu != 1)."""
        from ._lib import check
        pa, pb = self._scalars(alpha, beta, Y_loc.dtype)
        check(self._L.hmx_dist_matmat_row_major_local_to_local(self._h, trans.encode(), pa, C.c_void_p(X_loc.data_ptr()), pb, C.c_void_p(Y_loc.data_ptr()), int(mu), self._stream(Y_loc)))
        return Y_loc

    def gmv(self, x, y, mu, dof):
        """hmx_dist_gmv: HPDDMOperator::GMV's body (wrappers/wrapper_hpddm.hpp:102-142) without HPDDM's overlap exchange.  x, y: 1-D device
        tensors of dof * mu coefficients, column-major with leading dimension dof."""
        from ._lib import check
        check(self._L.hmx_dist_gmv(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), int(mu), int(dof), self._stream(y)))
        return y

    @staticmethod
    def _column_major(M, name):
        """(pointer, mu) of a column-major n x mu device matrix: a 1-D tensor (mu = 1) or the transpose of a contiguous mu x n tensor."""
        if M.dim() == 1:
            if not M.is_contiguous():
                raise ValueError(name + " must be contiguous")
            return M.data_ptr(), 1
        if M.dim() != 2 or not M.T.is_contiguous():
            raise ValueError(name + " must be column-major (the transpose of a contiguous mu x n tensor)")
        return M.data_ptr(), int(M.shape[1])

    def matmat_global_to_global(self, trans, alpha, X, beta, Y, user_numbering=True):
        """hmx_dist_matmat_global_to_global: column-major X (n x mu), Y (m x mu), user or partition numbering."""
        from ._lib import check
        px, mu = self._column_major(X, "X")
        py, mu_y = self._column_major(Y, "Y")
        if mu != mu_y:
            raise ValueError("X and Y must have the same number of columns")
        pa, pb = self._scalars(alpha, beta, Y.dtype)
        check(self._L.hmx_dist_matmat_global_to_global(self._h, trans.encode(), pa, C.c_void_p(px), pb, C.c_void_p(py), mu, 1 if user_numbering else 0, self._stream(Y)))
        return Y

    def matmat_local_to_local(self, trans, alpha, X_loc, beta, Y_loc, user_numbering=True):
        """hmx_dist_matmat_local_to_local: column-major local slices, the rank's local user numbering or partition numbering."""
        from ._lib import check
        px, mu = self._column_major(X_loc, "X_loc")
        py, mu_y = self._column_major(Y_loc, "Y_loc")
        if mu != mu_y:
            raise ValueError("X_loc and Y_loc must have the same number of columns")
        pa, pb = self._scalars(alpha, beta, Y_loc.dtype)
        check(self._L.hmx_dist_matmat_local_to_local(self._h, trans.encode(), pa, C.c_void_p(px), pb, C.c_void_p(py), mu, 1 if user_numbering else 0, self._stream(Y_loc)))
        return Y_loc


class DefaultApproximationBuilder:
    """Builds this rank's block rows on its GPU and wires them into a DistributedOperator
    (distributed_operator/utility.hpp:38-61).  Holds `hmatrix`, `distributed_operator`."""

    def __init__(self, generator, target_cluster, source_cluster, hmatrix_tree_builder, group=None, device=None, dtype=np.float64):
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if device is None:
            device = torch.cuda.current_device()
        self.target_partition = PartitionFromCluster(target_cluster)
        self.source_partition = PartitionFromCluster(source_cluster)
        self.hmatrix = hmatrix_tree_builder.build(generator, target_cluster, source_cluster, rank, rank, device=device, dtype=dtype)
        self.local_hmatrix = RestrictedGlobalToLocalHMatrix(self.hmatrix)
        self.distributed_operator = DistributedOperator(self.target_partition, self.source_partition, group)
        self.distributed_operator.add_global_to_local_operator(self.local_hmatrix)


def internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, alpha, A, X, beta, Y, mu):
    """Multi-RHS, row-major (mu fastest), partition numbering
    (distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_global_to_global.hpp:18-85):
    same collectives as the vector product on mu-interleaved rows (C5)."""
    rank = A.rank()
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    off_in, n_in = in_part.get_offset_of_partition(rank), in_part.get_size_of_partition(rank)
    off_out, n_out = out_part.get_offset_of_partition(rank), out_part.get_size_of_partition(rank)
    if trans == "N":
        local = Y[off_out:off_out + n_out].clone() if beta != 0 else torch.zeros((n_out, mu), dtype=Y.dtype, device=Y.device)
        apply_beta = True
        for op in A.global_to_local_operators:
            op.add_matrix_product_row_major(trans, alpha, X, beta if apply_beta else 1.0, local, mu)
            apply_beta = False
        for op in A.local_to_local_operators:  # :68-71 of the reference: the block-diagonal operators see the local rows of X
            op.add_matrix_product_row_major(trans, alpha, X[off_in:off_in + n_in], beta if apply_beta else 1.0, local, mu)
            apply_beta = False
        A._gather_slices(local, out_part, Y)
    else:
        Y_old = Y.clone() if beta != 0 else None
        Y.zero_()
        X_loc = X[off_in:off_in + n_in].contiguous()
        for op in A.global_to_local_operators:
            op.add_matrix_product_row_major(trans, alpha, X_loc, 1.0, Y, mu)
        for op in A.local_to_local_operators:
            op.add_matrix_product_row_major(trans, alpha, X_loc, 1.0, Y[off_out:off_out + n_out], mu)
        if A.size() > 1:
            dist.all_reduce(_rv(Y), op=dist.ReduceOp.SUM, group=A.group)
        if beta != 0:
            Y.add_(Y_old, alpha=beta)
    return Y


def internal_add_distributed_operator_matrix_product_row_major_local_to_local(trans, alpha, A, X_loc, beta, Y_loc, mu):
    """Multi-RHS Krylov-side contract: what HPDDMOperator::GMV calls for mu != 1 (wrappers/wrapper_hpddm.hpp:126).  Local row
    slices in and out, row-major (mu fastest), partition numbering
    (distributed_operator/linalg/add_distributed_operator_matrix_product_row_major_local_to_local.hpp:19-95).
    trans='N': local-to-local operators on the local rows, all-gather of the mu-interleaved rows of X (local_to_global,
    linalg/utility.hpp:11-28), global-to-local operators.  trans!='N': local product into a zeroed global matrix, then the
    reference's MPI_Alltoallv + p axpys (:64-93) as one reduce-scatter (DistributedOperator._reduce_scatter_slices)."""
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    if trans == "N":
        apply_beta = True
        for op in A.local_to_local_operators:
            op.add_matrix_product_row_major(trans, alpha, X_loc, beta if apply_beta else 1.0, Y_loc, mu)
            apply_beta = False
        if A.global_to_local_operators:
            X = torch.empty((in_part.get_global_size(), mu), dtype=X_loc.dtype, device=X_loc.device)
            A._gather_slices(X_loc, in_part, X)
            for op in A.global_to_local_operators:
                op.add_matrix_product_row_major(trans, alpha, X, beta if apply_beta else 1.0, Y_loc, mu)
                apply_beta = False
    else:
        if beta != 1:
            Y_loc.mul_(beta)
        for op in A.local_to_local_operators:
            op.add_matrix_product_row_major(trans, alpha, X_loc, 1.0, Y_loc, mu)
        if A.global_to_local_operators:
            buf = torch.zeros((out_part.get_global_size(), mu), dtype=X_loc.dtype, device=X_loc.device)
            for op in A.global_to_local_operators:
                op.add_matrix_product_row_major(trans, alpha, X_loc, 1.0, buf, mu)
            Y_loc.add_(A._reduce_scatter_slices(buf, out_part))
    return Y_loc


def hpddm_gmv(A, x, y, mu, dof):
    """HPDDMOperator::GMV (wrappers/wrapper_hpddm.hpp:102-142) up to HPDDM's own overlap exchange, over the torch.distributed layer:
    x, y are 1-D tensors of dof * mu coefficients, column-major with leading dimension dof (local size + overlap); the first local_size
    rows of every column go through the local-to-local product (alpha = 1, beta = 0), the overlap rows of y are zeroed."""
    n = A.target_partition.get_size_of_partition(A.rank())
    X, Y = x.view(mu, dof), y.view(mu, dof)
    if mu == 1:
        out = torch.zeros(n, dtype=y.dtype, device=y.device)
        internal_add_distributed_operator_vector_product_local_to_local("N", 1.0, A, X[0, :n].contiguous(), 0.0, out)
        Y[0, :n] = out
    else:
        out = torch.zeros((n, mu), dtype=y.dtype, device=y.device)
        internal_add_distributed_operator_matrix_product_row_major_local_to_local("N", 1.0, A, X[:, :n].t().contiguous(), 0.0, out, mu)
        Y[:, :n] = out.t()
    Y[:, n:] = 0
    return y


def _rows_by_mu(M):
    """A column-major n x mu matrix as the library sees it: any 2-D tensor of shape (n, mu) (or a vector: mu = 1)."""
    return M.reshape(-1, 1) if M.dim() == 1 else M


def internal_add_distributed_operator_matrix_product_global_to_global(trans, alpha, A, X, beta, Y):
    """Column-major multi-RHS product in partition numbering: transposition to the row-major layout, row-major product,
    transposition back (distributed_operator/linalg/add_distributed_operator_matrix_product_global_to_global.hpp:18-117).
    X (n x mu) and Y (m x mu) are 2-D tensors of those shapes with any strides (column-major = the transpose of a contiguous
    mu x n tensor); Y is updated in place."""
    X2, Y2 = _rows_by_mu(X), _rows_by_mu(Y)
    mu = X2.shape[1]
    Xr = X2.contiguous()
    Yr = Y2.contiguous().clone() if beta != 0 else torch.zeros(Y2.shape, dtype=Y2.dtype, device=Y2.device)
    internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, alpha, A, Xr, beta, Yr, mu)
    Y2.copy_(Yr)
    return Y


def add_distributed_operator_matrix_product_global_to_global(trans, alpha, A, X, beta, Y):
    """The same in USER numbering: every column through global_to_partition_numbering on the way in and
    partition_to_global_numbering on the way out (same file :132-279)."""
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    X2, Y2 = _rows_by_mu(X), _rows_by_mu(Y)
    mu = X2.shape[1]
    Xr = in_part.global_to_partition_numbering(X2).contiguous()
    Yr = out_part.global_to_partition_numbering(Y2).contiguous() if beta != 0 else torch.zeros(Y2.shape, dtype=Y2.dtype, device=Y2.device)
    internal_add_distributed_operator_matrix_product_row_major_global_to_global(trans, alpha, A, Xr, beta, Yr, mu)
    Y2.copy_(out_part.partition_to_global_numbering(Yr))
    return Y


def internal_add_distributed_operator_matrix_product_local_to_local(trans, alpha, A, X_loc, beta, Y_loc):
    """Column-major local slices, partition numbering
    (distributed_operator/linalg/add_distributed_operator_matrix_product_local_to_local.hpp:20-49)."""
    X2, Y2 = _rows_by_mu(X_loc), _rows_by_mu(Y_loc)
    mu = X2.shape[1]
    Xr = X2.contiguous()
    Yr = Y2.contiguous().clone() if beta != 0 else torch.zeros(Y2.shape, dtype=Y2.dtype, device=Y2.device)
    internal_add_distributed_operator_matrix_product_row_major_local_to_local(trans, alpha, A, Xr, beta, Yr, mu)
    Y2.copy_(Yr)
    return Y_loc


def add_distributed_operator_matrix_product_local_to_local(trans, alpha, A, X_loc, beta, Y_loc):
    """Column-major local slices in the rank's LOCAL user numbering (same file :66-120): needs cluster trees whose permutation
    is local to the partitions (create_cluster_tree_from_local_partition)."""
    rank = A.rank()
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    X2, Y2 = _rows_by_mu(X_loc), _rows_by_mu(Y_loc)
    mu = X2.shape[1]
    Xr = in_part.local_to_local_partition_numbering(rank, X2).contiguous()
    Yr = out_part.local_to_local_partition_numbering(rank, Y2).contiguous() if beta != 0 else torch.zeros(Y2.shape, dtype=Y2.dtype, device=Y2.device)
    internal_add_distributed_operator_matrix_product_row_major_local_to_local(trans, alpha, A, Xr, beta, Yr, mu)
    Y2.copy_(out_part.local_partition_to_local_numbering(rank, Yr))
    return Y_loc


def add_distributed_operator_vector_product_local_to_local(trans, alpha, A, x_loc, beta, y_loc):
    """Local slices in the rank's local user numbering
    (distributed_operator/linalg/add_distributed_operator_vector_product_local_to_local.hpp:99-125)."""
    rank = A.rank()
    in_part = A.source_partition if trans == "N" else A.target_partition
    out_part = A.target_partition if trans == "N" else A.source_partition
    xp = in_part.local_to_local_partition_numbering(rank, x_loc)
    yp = out_part.local_to_local_partition_numbering(rank, y_loc) if beta != 0 else torch.zeros_like(y_loc)
    internal_add_distributed_operator_vector_product_local_to_local(trans, alpha, A, xp, beta, yp)
    out_part.local_partition_to_local_numbering(rank, yp, out=y_loc)
    return y_loc


def internal_add_distributed_operator_vector_sub_product_global_to_local(A, X, Y_loc, mu, offset, size):
    """Y_loc += A_loc * (X zero-extended): X holds rows [offset, offset + size) of the source partition numbering, row-major
    with mu columns (distributed_operator/linalg/add_distributed_operator_vector_sub_product_global_to_local.hpp:11-21)."""
    for op in A.global_to_local_operators:
        op.add_sub_matrix_product_to_local(X, Y_loc, mu, offset, size)
    for op in A.local_to_local_operators:
        op.add_sub_matrix_product_to_local(X, Y_loc, mu, offset, size)
    return Y_loc


class DefaultLocalApproximationBuilder:
    """Block-diagonal operator: rank k compresses only (target partition k) x (source partition k) and registers it as a
    local-to-local operator (distributed_operator/utility.hpp:64-88).  `block_diagonal_hmatrix` is that H-matrix."""

    def __init__(self, generator, target_cluster, source_cluster, hmatrix_tree_builder, group=None, device=None, rank=None, dtype=np.float64):
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if device is None:
            device = torch.cuda.current_device()
        self.target_partition = PartitionFromCluster(target_cluster)
        self.source_partition = PartitionFromCluster(source_cluster)
        self.hmatrix = hmatrix_tree_builder.build(generator, target_cluster, source_cluster, device=device, local_partitions=(rank, rank), dtype=dtype)
        self.block_diagonal_hmatrix = self.hmatrix
        self.local_hmatrix = LocalToLocalHMatrix(self.hmatrix)
        self.distributed_operator = DistributedOperator(self.target_partition, self.source_partition, group)
        self.distributed_operator.add_local_to_local_operator(self.local_hmatrix)


def get_distributed_hmatrix_information(hmatrix, group=None):
    """get_distributed_hmatrix_information (hmatrix/hmatrix_distributed_output.hpp:30-214): block-size / rank statistics of the
    rank-local H-matrices reduced over the ranks (max / min / sum), compression ratio and space saving of the whole
    operator, summed integer build information, max / mean / min of the build times.  Same keys and number formatting as
    the reference; the map is filled on rank 0 only (empty elsewhere), the OpenMP thread-count line is not reported and
    Number_of_procs is the number of ranks (one GPU each).  `hmatrix` needs nb_rows(), nb_cols(), leaf_table() and stats()."""
    ini = dist.is_initialized()
    rank = dist.get_rank(group) if ini else 0
    world = dist.get_world_size(group) if ini else 1
    nr, nc = int(hmatrix.nb_rows()), int(hmatrix.nb_cols())
    local_size = nr * nc
    lt = np.asarray(hmatrix.leaf_table())
    lt = lt.reshape(-1, lt.shape[1] if lt.ndim == 2 else 6)
    size = lt[:, 1].astype(np.int64) * lt[:, 3].astype(np.int64)
    lr = lt[:, 4] >= 0
    dn = ~lr
    ranks = lt[lr, 4].astype(np.int64)
    generated = int((ranks * (lt[lr, 1].astype(np.int64) + lt[lr, 3])).sum() + size[dn].sum())

    def mx(v):
        return int(v.max()) if len(v) else 0

    def mn(v):  # the reference starts its minima from the local size (:47)
        return min(local_size, int(v.min())) if len(v) else local_size

    st = hmatrix.stats()
    timings = {"Blocks_computation_walltime": float(st.get("t_compress_s", 0.0)) + float(st.get("t_assemble_s", 0.0)) + float(st.get("t_pack_s", 0.0)),
               "Block_tree_walltime": float(getattr(hmatrix, "_block_tree_walltime", 0.0))}
    names = sorted(timings)
    maxi = torch.tensor([mx(size[dn]), mx(size[lr]), mx(ranks), nr, nc] + [0] * len(names), dtype=torch.float64)
    mini = torch.tensor([mn(size[dn]), mn(size[lr]), mn(ranks), nr, nc] + [0] * len(names), dtype=torch.float64)
    sums = torch.tensor([float(size[dn].sum()), float(size[lr].sum()), float(ranks.sum()), float(nr), float(nc),
                         float(dn.sum()), float(lr.sum()), float(local_size), float(generated), float(st["n_false_positive"])] +
                        [timings[k] for k in names], dtype=torch.float64)
    for k, name in enumerate(names):
        maxi[5 + k] = mini[5 + k] = timings[name]
    if ini and world > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        maxi, mini, sums = maxi.to(dev), mini.to(dev), sums.to(dev)
        dist.all_reduce(maxi, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(mini, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        maxi, mini, sums = maxi.cpu(), mini.cpu(), sums.cpu()
    if rank != 0:
        return {}
    nd, nl, total_size, total_generated = int(sums[5]), int(sums[6]), int(sums[7]), int(sums[8])
    info = {
        "Target_size_max": str(int(maxi[3])), "Target_size_mean": "%f" % (float(sums[3]) / world), "Target_size_min": str(int(mini[3])),
        "Source_size_max": str(int(maxi[4])), "Source_size_mean": "%f" % (float(sums[4]) / world), "Source_size_min": str(int(mini[4])),
        "Dense_block_size_max": str(int(maxi[0])), "Dense_block_size_mean": "%f" % (float(sums[0]) / nd if nd else 0.0),
        "Dense_block_size_min": str(int(mini[0]) if nd else 0),
        "Low_rank_block_size_max": str(int(maxi[1])), "Low_rank_block_size_mean": "%f" % (float(sums[1]) / nl if nl else 0.0),
        "Low_rank_block_size_min": str(int(mini[1]) if nl else 0),
        "Rank_max": str(int(maxi[2])), "Rank_mean": "%f" % (float(sums[2]) / nl if nl else 0.0), "Rank_min": str(int(mini[2]) if nl else 0),
        "Number_of_low_rank_blocks": str(nl), "Number_of_dense_blocks": str(nd),
        "Compression_ratio": "%f" % (total_size / float(total_generated)) if total_generated else "inf",
        "Space_saving": "%f" % (1 - float(total_generated) / total_size) if total_size else "%f" % 0.0,
        "Number_of_MPI_tasks": str(world), "Number_of_procs": str(world),
        "Number_of_false_positive": str(int(sums[9])),
    }
    for k, name in enumerate(names):
        info[name + "_max"] = "%f second(s)" % float(maxi[5 + k])
        info[name + "_mean"] = "%f second(s)" % (float(sums[10 + k]) / world)
        info[name + "_min"] = "%f second(s)" % float(mini[5 + k])
    return info


def print_distributed_hmatrix_information(hmatrix, file=None, group=None):
    """print_distributed_hmatrix_information (hmatrix/hmatrix_distributed_output.hpp:218-243): rank 0 writes the text
    use_distributed_operator.cpp prints; every rank must call it (it reduces over the group)."""
    import sys
    info = get_distributed_hmatrix_information(hmatrix, group)
    if not info:
        return
    out = sys.stdout if file is None else file
    width = 2 + max(len(k) for k in info)
    out.write("Distributed Hmatrix information\n")
    for k in sorted(info):
        out.write(k.ljust(width, "_") + info[k] + "\n")

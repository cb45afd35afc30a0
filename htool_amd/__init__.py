"""htool_amd -- MI355X-native H-matrix compression + H-matvec engine behind htool's plugin surface.

Only the hot path lives here: csrc/ holds the HIP kernels and the C ABI
(include/hmx.h); api.py mirrors htool's ClusterTreeBuilder / HMatrixTreeBuilder / add_hmatrix_vector_product
operator interface over that ABI; distributed.py is the row-partitioned DistributedOperator.
"""
from ._lib import HmxError, lib  # noqa: F401
from .api import (ClusterTreeBuilder, Cluster, HMatrixTreeBuilder, HMatrix, InvDistGenerator, HelmholtzGenerator, LaplaceGenerator, VirtualGenerator, NativeGenerator,  # noqa: F401
                  add_hmatrix_vector_product, internal_add_hmatrix_vector_product,
                  internal_add_hmatrix_matrix_product_row_major, add_hmatrix_matrix_product, create_geometry,
                  save_cluster_tree, read_cluster_tree, save_leaves_with_rank, matrix_to_bytes, bytes_to_matrix, trim_device_cache,
                  get_tree_parameters, get_hmatrix_information, print_tree_parameters, print_hmatrix_information, cluster_tree_from_nodes)

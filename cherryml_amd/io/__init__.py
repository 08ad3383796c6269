"""Text formats on either side of the hot path (same files as the reference:
cherryml/io/_count_matrices.py:8-81, cherryml/io/_rate_matrix.py:37-77)."""
from ._formats import (  # noqa: F401
    read_count_matrices,
    read_count_matrices_arrays,
    read_mask_matrix,
    read_probability_distribution,
    read_rate_matrix,
    write_count_matrices,
    write_probability_distribution,
    write_rate_matrix,
)
from ._tree import Tree, convert_newick_to_CherryML_Tree, read_tree, write_tree  # noqa: F401,E402

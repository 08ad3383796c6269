"""Counting stage: trees + MSAs (+ site rates / contact maps) -> the
count-matrix bank, with the reference's stage-function signatures
(cherryml/counting/_count_transitions.py:210, _count_co_transitions.py:238)."""
from ._stage import count_co_transitions, count_transitions  # noqa: F401
from ._host import (  # noqa: F401
    build_pairs,
    encode_msa,
    read_contact_map,
    read_msa,
    read_site_rates,
    read_tree_arrays,
)

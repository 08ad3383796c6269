from ._cherry import (  # noqa: F401
    coevolution_end_to_end_with_cherryml_optimizer,
    create_maximal_matching_contact_map,
    lg_end_to_end_with_cherryml_optimizer,
)

from ._cherry import (  # noqa: F401
    coevolution_end_to_end_with_cherryml_optimizer,
    create_maximal_matching_contact_map,
    lg_end_to_end_with_cherryml_optimizer,
)
from ._resident import coevolution_fit_resident, jtt_ipw_from_reduced_statistics  # noqa: F401,E402

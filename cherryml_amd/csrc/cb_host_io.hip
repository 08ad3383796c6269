// libcherrybank: host-side text formats and FastCherries pairing (no device code).
#include "cb_internal.hip.h"

#include "host_io.hip.h"

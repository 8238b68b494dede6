// k_walks.hip -- home translation unit of: chain walks and crown steps (k_walks.hpp).
// Nothing but the explicit instantiations: the templates are in the headers, the list is generated (tools/gen_instantiations.py),
// rapidnet_capi.hip declares the same list `extern`.
#include "k_walks.hpp"

#define RN_LINKAGE
#include "instantiations/walks.inc"

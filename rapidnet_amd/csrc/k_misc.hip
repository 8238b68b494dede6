// k_misc.hip -- home translation unit of: probes, factor step, affine terms, utilities (k_misc.hpp).
// Nothing but the explicit instantiations: the templates are in the headers, the list is generated (tools/gen_instantiations.py),
// rapidnet_capi.hip declares the same list `extern`.
#include "k_misc.hpp"

#define RN_LINKAGE
#include "instantiations/misc.inc"

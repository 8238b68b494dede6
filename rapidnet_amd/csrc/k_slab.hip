// k_slab.hip -- home translation unit of: shared-operator products on the matrix cores (k_slab.hpp).
// Nothing but the explicit instantiations: the templates are in the headers, the list is generated (tools/gen_instantiations.py),
// rapidnet_capi.hip declares the same list `extern`.
#include "k_slab.hpp"

#define RN_LINKAGE
#include "instantiations/slab.inc"

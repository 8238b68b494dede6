// k_dual.hip -- home translation unit of: fused dual update, bookkeeping, step-wise element-wise kernels (k_dual.hpp).
// Nothing but the explicit instantiations: the templates are in the headers, the list is generated (tools/gen_instantiations.py),
// rapidnet_capi.hip declares the same list `extern`.
#include "k_dual.hpp"

#define RN_LINKAGE
#include "instantiations/dual.inc"

// k_fbe.hip -- home translation unit of: global-FBE / NAMA kernels (fbe_kernels.hpp).
// Nothing but the explicit instantiations: the templates are in the headers, the list is generated (tools/gen_instantiations.py),
// rapidnet_capi.hip declares the same list `extern`.
#include "fbe_kernels.hpp"

#define RN_LINKAGE
#include "instantiations/fbe.inc"

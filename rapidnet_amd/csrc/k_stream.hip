// k_stream.hip -- home translation unit of: k_stream_gemv / k_struct_prep (k_stream.hpp).
// Nothing but the explicit instantiations: the templates are in the headers, the list is generated (tools/gen_instantiations.py),
// rapidnet_capi.hip declares the same list `extern`.
#include "k_stream.hpp"

#define RN_LINKAGE
#include "instantiations/stream.inc"

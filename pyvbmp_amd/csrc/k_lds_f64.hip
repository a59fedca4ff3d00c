// the smoother keeps the hazard pad in front of every DPP group: tools/check_dpp_hazards.py finds register copies placed
// right in front of some of its groups when the pads are dropped
#define VBMP_GROUP_PAD "s_nop 1\n\t"
#define VBMP_REAL double
#define VBMP_SUF f64
#include "k_lds_impl.inc"

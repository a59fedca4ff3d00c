// the smoother keeps the hazard pad in front of the groups that read their own destination row through the DPP operand:
// tools/check_dpp_hazards.py finds register copies placed right in front of some of them when those pads are dropped
#define VBMP_GROUP_PAD ""
#define VBMP_SELF_PAD "s_nop 1\n\t"
#define VBMP_REAL float
#define VBMP_SUF f32
#include "k_lds_impl.inc"

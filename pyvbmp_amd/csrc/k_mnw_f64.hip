// the fp64 instances spill into accumulator registers, and tools/check_dpp_hazards.py finds their reloads (v_accvgpr_read)
// placed right in front of DPP groups: this unit keeps the hazard pad in front of every group
#define VBMP_GROUP_PAD "s_nop 1\n\t"
#define VBMP_REAL double
#define VBMP_SUF f64
#include "k_mnw_impl.inc"

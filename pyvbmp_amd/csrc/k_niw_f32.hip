// K1 / K2 are HBM-bound (their vector unit is a third busy): the hazard pads in front of the DPP groups cost nothing
// measurable here and stay -- tools/check_dpp_hazards.py finds constants materialised right in front of groups of the
// padded-size instances when they are dropped
#define VBMP_GROUP_PAD "s_nop 1\n\t"
#define VBMP_REAL float
#define VBMP_SUF f32
#include "k_niw_impl.inc"

#define VBMP_REAL float
#define VBMP_SUF f32
#include "k_mnw_impl.inc"

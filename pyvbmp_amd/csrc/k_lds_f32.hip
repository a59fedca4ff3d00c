#define VBMP_REAL float
#define VBMP_SUF f32
#include "k_lds_impl.inc"

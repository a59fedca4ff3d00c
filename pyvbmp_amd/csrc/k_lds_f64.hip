#define VBMP_REAL double
#define VBMP_SUF f64
#include "k_lds_impl.inc"

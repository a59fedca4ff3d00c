#define VBMP_REAL double
#define VBMP_SUF f64
#include "k_niw_impl.inc"

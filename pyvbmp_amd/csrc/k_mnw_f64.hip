#define VBMP_REAL double
#define VBMP_SUF f64
#include "k_mnw_impl.inc"

// Kernel instantiations for P256 (see k_weierstrass.inc).
#define ECCX_CURVE P256
#define ECCX_CURVE_U P256U
#define ECCX_OPS_NAME ops_P256
#include "k_weierstrass.inc"

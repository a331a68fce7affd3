// Kernel instantiations for P384 (see k_weierstrass.inc).
#define ECCX_CURVE P384
#define ECCX_CURVE_U P384U
#define ECCX_OPS_NAME ops_P384
#include "k_weierstrass.inc"

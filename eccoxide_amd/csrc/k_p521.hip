// Kernel instantiations for P521 (see k_weierstrass.inc).
#define ECCX_CURVE P521
#define ECCX_CURVE_U P521U
#define ECCX_OPS_NAME ops_P521
#include "k_weierstrass.inc"

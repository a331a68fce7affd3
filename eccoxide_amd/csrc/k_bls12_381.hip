// Kernel instantiations for BLS12_381 (see k_weierstrass.inc).
#define ECCX_CURVE BLS12_381
#define ECCX_CURVE_U BLS12_381U
#define ECCX_OPS_NAME ops_BLS12_381
#define ECCX_CODEC_FORMAT FORMAT_ZCASH
#include "k_weierstrass.inc"

// zr_tu_di_e.hip -- translation unit of libzetaray_amd.so holding the material-class permutation (PLAIN = true) of K5 - K8 and K10 (zr_kernels_di.h)
#include "zr_kernels_di.h"
ZR_DI_GROUP(template, true)

// zr_tu_rpt_e.hip -- translation unit of libzetaray_amd.so holding the material-class permutation of K11 / K14 / K16 (PLAIN = true) (ZR_RPT_GROUP_E, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_E(template)

// zr_tu_rpt_d.hip -- translation unit of libzetaray_amd.so holding the textured K16 spatial reconnect kernels (ZR_RPT_GROUP_D, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_D(template)

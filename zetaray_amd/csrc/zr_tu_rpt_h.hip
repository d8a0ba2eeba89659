// zr_tu_rpt_h.hip -- translation unit of libzetaray_amd.so holding the untextured K14 temporal and K16 spatial reconnect kernels (ZR_RPT_GROUP_H, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_H(template)

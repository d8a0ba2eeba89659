// zr_tu_rpt_g.hip -- translation unit of libzetaray_amd.so holding the textured K11 path-tracing and K14 temporal reconnect kernels (ZR_RPT_GROUP_G, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_G(template)

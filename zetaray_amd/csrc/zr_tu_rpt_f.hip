// zr_tu_rpt_f.hip -- translation unit of libzetaray_amd.so holding the K13 replay kernels of the spatial pass (ZR_RPT_GROUP_F, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_F(template)

// zr_tu_rpt_b.hip -- translation unit of libzetaray_amd.so holding the K13 replay kernels of the temporal pass, emissive lighting (ZR_RPT_GROUP_B, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_B(template)

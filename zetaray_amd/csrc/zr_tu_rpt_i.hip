// zr_tu_rpt_i.hip -- translation unit of libzetaray_amd.so holding the K13 replay kernels of the temporal pass, sun + sky lighting (ZR_RPT_GROUP_I, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_I(template)

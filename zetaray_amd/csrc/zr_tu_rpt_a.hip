// zr_tu_rpt_a.hip -- translation unit of libzetaray_amd.so holding the K11 / K14 ReSTIR PT kernels (see zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_A(template)

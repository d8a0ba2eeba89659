// zr_tu_rpt_a.hip -- translation unit of libzetaray_amd.so holding the untextured K11 ReSTIR PT path-tracing kernels (ZR_RPT_GROUP_A, zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_A(template)

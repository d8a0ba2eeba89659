// zr_tu_rpt_c.hip -- translation unit of libzetaray_amd.so holding K11 with pooled traces (k_rpt_pathtrace_coop; see zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_C(template)

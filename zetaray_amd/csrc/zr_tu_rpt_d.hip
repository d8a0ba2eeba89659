// zr_tu_rpt_d.hip -- translation unit of libzetaray_amd.so holding the K16 ReSTIR PT spatial reconnect kernel (see zr_kernels.h)
#include "zr_kernels.h"
ZR_RPT_GROUP_D(template)

// zr_tu_rpt_c.hip -- translation unit of the EXPERIMENTS build only (libzetaray_amd_exp.so, `make experiments`): K11's park / compact / trip / pool
// forms (zr_kernels_exp.h)
#include "zr_kernels.h"
#ifndef ZR_EXPERIMENTS
#error "zr_tu_rpt_c.hip belongs to the experiments build (-DZR_EXPERIMENTS)"
#endif
ZR_RPT_GROUP_C(template)

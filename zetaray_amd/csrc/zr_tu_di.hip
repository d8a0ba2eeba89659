// zr_tu_di.hip -- translation unit of libzetaray_amd.so holding the direct-lighting kernels (K5 / K6 emissive ReSTIR DI, K7 / K8 sun + sky ReSTIR DI) and
// the ReSTIR GI kernel (K10) of the general material class + the TEXTURED K10; definitions in zr_kernels_di.h, launches in zr_api.hip
#include "zr_kernels_di.h"
ZR_DI_GROUP(template, false)
__global__ void __launch_bounds__(kRgiBlock) ZR_WAVES(6) k_rgi_tex(rgi::GiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters)
{ ZR_RGI_KERNEL_BODY(true, false) }

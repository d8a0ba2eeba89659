// zr_kernels_di.h -- block sizes and prototypes of the direct-lighting (K5 - K8) and ReSTIR GI (K10) kernels, shared by zr_api.hip (launches)
// and zr_tu_di.hip (definitions)
#pragma once
#include "zr_kernels.h"
#include "zr_rdi.h"
#include "zr_sdi.h"
#include "zr_rgi.h"
static constexpr int kSdiBlock = 256, kDiBlock = 64;
__global__ void k_sdi_temporal(sdi::SkyFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);
__global__ void k_sdi_spatial(sdi::SkyFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);
__global__ void k_rdi_temporal(rdi::DiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);
__global__ void k_rdi_spatial(rdi::DiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);
__global__ void k_rgi(rgi::GiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);
__global__ void k_rgi_tex(rgi::GiFrame F, zr_frame_constants g, uint32_t tilesX, unsigned long long* counters);

# Builds the REFERENCE's own code, IN PLACE from /root/reference, into oracle/_ref/ (git-ignored; travels to the GPU box with the
# snapshot).  g++ directly on the few source files -- the reference's CMake build is not run.  Usage: make -C oracle -f _ref.mk
#
#   _ref/libzref.so        the reference's CPU sources for the alias table / Kahan sum / oct32 / half (ZetaCore/Math/*.cpp) and the C++
#                          side of its shared C++/HLSL headers (RtCommon.h, Material.h, Vertex.h, FrameConstants.h: layout pins)
#   _ref/libzref_hlsl.so   the reference's own HLSL shader headers (ZetaRenderPass/Common/*.hlsli ...) compiled as C++: hlsl2cpp.py
#                          rewrites them lexically into _ref/gen/ (never committed), ref_hlsl/hlsl_shim.h supplies the HLSL language
#                          surface over the ABI's arithmetic contract, ref_hlsl/ref_hlsl_driver.cpp exports the probes
REF ?= /root/reference
CXX ?= g++
FLAGS := -std=c++20 -O2 -mavx2 -mfma -mf16c -fPIC -w -include ref_shim.h -I$(REF)/Source/ZetaCore -I$(REF)/Source -I$(REF)/External -DNDEBUG
# The shader code is compiled with clang++ (the ROCm toolchain's) and -ftrivial-auto-var-init=zero: HLSL leaves locals such as
# `Reconnection ret;` partly unset (Shift.hlsli:16-36 never sets x_k_in_motion / lobes / ID) and later READS them
# (ReSTIR_PT_Reconnect_StC.hlsl:275); on the GPU an undefined value is 0 in practice, and 0 is what the ABI defines for them.
HLSL_CXX ?= /opt/rocm/lib/llvm/bin/clang++
HLSL_FLAGS := -std=c++17 -O2 -ffp-contract=off -fno-fast-math -mavx2 -mfma -mf16c -fPIC -w -ftrivial-auto-var-init=zero -Iref_hlsl -I_ref/gen
HLSL_ROOTS := ZetaRenderPass/Common/BSDFSampling.hlsli ZetaRenderPass/Common/RT.hlsli ZetaRenderPass/Common/GBuffers.hlsli ZetaRenderPass/Common/FrameConstants.h \
    ZetaRenderPass/GBuffer/GBufferRT_Inline.hlsl ZetaRenderPass/GBuffer/GBufferRT_Common.h \
    ZetaRenderPass/IndirectLighting/PathTracer/PathTracer.hlsl ZetaRenderPass/IndirectLighting/IndirectLighting_Common.h \
    $(addprefix ZetaRenderPass/IndirectLighting/ReSTIR_PT/,ReSTIR_PT_PathTrace.hlsl ReSTIR_PT_Replay.hlsl ReSTIR_PT_Reconnect_CtT.hlsl \
        ReSTIR_PT_Reconnect_TtC.hlsl ReSTIR_PT_Reconnect_CtS.hlsl ReSTIR_PT_Reconnect_StC.hlsl ReSTIR_PT_SpatialSearch.hlsl ReSTIR_PT_Sort.hlsl) \
    ZetaRenderPass/IndirectLighting/ReSTIR_GI/ReSTIR_GI.hlsl \
    $(addprefix ZetaRenderPass/DirectLighting/,Emissive/ReSTIR_DI_Temporal.hlsl Emissive/ReSTIR_DI_Spatial.hlsl Emissive/DirectLighting_Common.h \
        Sky/SkyDI_Temporal.hlsl Sky/SkyDI_Spatial.hlsl Sky/SkyDI_Common.h) \
    $(addprefix ZetaRenderPass/AutoExposure/,AutoExposure_Histogram.hlsl AutoExposure_WeightedAvg.hlsl AutoExposure_Common.h) \
    ZetaRenderPass/Display/Display.hlsl ZetaRenderPass/Display/Display_Common.h \
    $(addprefix ZetaRenderPass/PreLighting/,EstimateTriEmissivePower.hlsl PresampleEmissives.hlsl BuildLightVoxelGrid.hlsl PreLighting_Common.h) \
    ZetaRenderPass/Sky/SkyViewLUT.hlsl ZetaRenderPass/Sky/Sky_Common.h \
    $(addprefix ZetaRenderPass/Compositing/,Compositing.hlsl FireflyFilter.hlsl Compositing_Common.h) ZetaRenderPass/TAA/TAA.hlsl ZetaRenderPass/TAA/TAA_Common.h

# the reference's shader PASSES compiled as C++ (one shared object per shader permutation, like the reference's .cso files):
#   _ref/libzref_k1.so                      GBufferRT_Inline.hlsl
#   _ref/libzref_k9_{e0,e1,e1p}.so          PathTracer.hlsl with NEE_EMISSIVE = 0 / 1 / 1 + USE_PRESAMPLED_SETS (PathTracer, _WoPS, _WPS)
#   _ref/libzref_rpt_{e0,e1,e1p}.so         ReSTIR PT: the 10 shaders of Variants/*.hlsl per NEE permutation + the restated host sequence (ref_rpt_host.cpp)
PASS_LIBS := _ref/libzref_k1.so _ref/libzref_k9_e0.so _ref/libzref_k9_e1.so _ref/libzref_k9_e1p.so _ref/libzref_rpt_e0.so _ref/libzref_rpt_e1.so _ref/libzref_rpt_e1p.so \
    _ref/libzref_gi_e0.so _ref/libzref_gi_e1.so _ref/libzref_gi_e1p.so _ref/libzref_di_e1.so _ref/libzref_di_e1p.so _ref/libzref_di_e1h.so _ref/libzref_di_sky.so _ref/libzref_post.so _ref/libzref_aux.so _ref/libzref_gi_e1l.so
PASS_HDRS := ref_hlsl/ref_pass_common.h ref_hlsl/hlsl_shim.h ref_hlsl/hlsl_resources.h ref_hlsl/hlsl_rt.h ref_hlsl/hlsl_group.h zro_scene.h

all: _ref/libzref.so _ref/libzref_hlsl.so $(PASS_LIBS)

_ref/libzref_k1.so: _ref/gen/.stamp ref_hlsl/ref_pass_gbuffer.cpp $(PASS_HDRS)
	$(HLSL_CXX) $(HLSL_FLAGS) -shared -o $@ ref_hlsl/ref_pass_gbuffer.cpp
_ref/libzref_k9_e0.so: _ref/gen/.stamp ref_hlsl/ref_pass_pathtracer.cpp $(PASS_HDRS)
	$(HLSL_CXX) $(HLSL_FLAGS) -DNEE_EMISSIVE=0 -shared -o $@ ref_hlsl/ref_pass_pathtracer.cpp
_ref/libzref_k9_e1.so: _ref/gen/.stamp ref_hlsl/ref_pass_pathtracer.cpp $(PASS_HDRS)
	$(HLSL_CXX) $(HLSL_FLAGS) -DNEE_EMISSIVE=1 -shared -o $@ ref_hlsl/ref_pass_pathtracer.cpp
_ref/libzref_k9_e1p.so: _ref/gen/.stamp ref_hlsl/ref_pass_pathtracer.cpp $(PASS_HDRS)
	$(HLSL_CXX) $(HLSL_FLAGS) -DNEE_EMISSIVE=1 -DUSE_PRESAMPLED_SETS -shared -o $@ ref_hlsl/ref_pass_pathtracer.cpp

_ref/libzref.so: ref_driver.cpp ref_shim.h
	mkdir -p _ref
	$(CXX) $(FLAGS) -shared -o $@ $(REF)/Source/ZetaCore/Math/Common.cpp $(REF)/Source/ZetaCore/Math/Sampling.cpp ref_driver.cpp

_ref/gen/.stamp: ref_hlsl/hlsl2cpp.py _ref.mk
	mkdir -p _ref/gen
	python3 ref_hlsl/hlsl2cpp.py $(REF)/Source _ref/gen $(HLSL_ROOTS)
	touch $@

_ref/libzref_hlsl.so: _ref/gen/.stamp ref_hlsl/ref_hlsl_driver.cpp ref_hlsl/hlsl_shim.h ref_hlsl/hlsl_resources.h zro_kat_layout.h
	$(HLSL_CXX) $(HLSL_FLAGS) -shared -o $@ ref_hlsl/ref_hlsl_driver.cpp

# ---- ReSTIR PT: one object per shader permutation (private namespace through -Dhlsl=...), linked with the host sequence
RPT := ZetaRenderPass/IndirectLighting/ReSTIR_PT
RPT_CB := -include ZetaRenderPass/IndirectLighting/IndirectLighting_Common.h
# $(call rpt_obj,<perm tag>,<shader tag>,<file>,<g_local type>,<has scene>,<has lights>,<permutation macros>)
define rpt_obj
_ref/obj/rpt_$(1)_$(2).o: _ref/gen/.stamp ref_hlsl/ref_pass_shader.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $$@ ref_hlsl/ref_pass_shader.cpp -Dhlsl=hlsl_rpt_$(2) '-DZR_SHADER="$(RPT)/$(3)"' \
	    -DZR_ENTRY=zrefp_shader_rpt_$(2) -DZR_LOCAL_CB=$(4) -DZR_HAS_SCENE=$(5) -DZR_HAS_LIGHTS=$(6) $(7)
RPT_OBJS_$(1) += _ref/obj/rpt_$(1)_$(2).o
endef
# $(call rpt_perm,<perm tag>,<NEE_EMISSIVE>,<extra path-trace macros>)
define rpt_perm
$(call rpt_obj,$(1),pathtrace,ReSTIR_PT_PathTrace.hlsl,cb_ReSTIR_PT_PathTrace,1,$(2),-DNEE_EMISSIVE=$(2) $(3))
$(call rpt_obj,$(1),replay_ctt,ReSTIR_PT_Replay.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),replay_ttc,ReSTIR_PT_Replay.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2) -DTEMPORAL_TO_CURRENT)
$(call rpt_obj,$(1),replay_cts,ReSTIR_PT_Replay.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2) -DCURRENT_TO_SPATIAL)
$(call rpt_obj,$(1),replay_stc,ReSTIR_PT_Replay.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2) -DSPATIAL_TO_CURRENT)
$(call rpt_obj,$(1),reconnect_ctt,ReSTIR_PT_Reconnect_CtT.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),reconnect_ttc,ReSTIR_PT_Reconnect_TtC.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),reconnect_cts,ReSTIR_PT_Reconnect_CtS.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),reconnect_stc,ReSTIR_PT_Reconnect_StC.hlsl,cb_ReSTIR_PT_Reuse,1,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),spatial_search,ReSTIR_PT_SpatialSearch.hlsl,cb_ReSTIR_PT_SpatialSearch,0,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),sort_ctt,ReSTIR_PT_Sort.hlsl,cb_ReSTIR_PT_Sort,0,0,-DNEE_EMISSIVE=$(2))
$(call rpt_obj,$(1),sort_ttc,ReSTIR_PT_Sort.hlsl,cb_ReSTIR_PT_Sort,0,0,-DNEE_EMISSIVE=$(2) -DTEMPORAL_TO_CURRENT)
$(call rpt_obj,$(1),sort_cts,ReSTIR_PT_Sort.hlsl,cb_ReSTIR_PT_Sort,0,0,-DNEE_EMISSIVE=$(2) -DCURRENT_TO_SPATIAL)
$(call rpt_obj,$(1),sort_stc,ReSTIR_PT_Sort.hlsl,cb_ReSTIR_PT_Sort,0,0,-DNEE_EMISSIVE=$(2) -DSPATIAL_TO_CURRENT)
_ref/obj/rpt_$(1)_host.o: _ref/gen/.stamp ref_hlsl/ref_rpt_host.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -c -o $$@ ref_hlsl/ref_rpt_host.cpp
_ref/libzref_rpt_$(1).so: $$(RPT_OBJS_$(1)) _ref/obj/rpt_$(1)_host.o
	$(HLSL_CXX) -shared -o $$@ $$(RPT_OBJS_$(1)) _ref/obj/rpt_$(1)_host.o
endef
$(eval $(call rpt_perm,e0,0,))
$(eval $(call rpt_perm,e1,1,))
$(eval $(call rpt_perm,e1p,1,-DUSE_PRESAMPLED_SETS))

# ---- ReSTIR GI (K10): ReSTIR_GI.hlsl / Variants/ReSTIR_GI_WoPS.hlsl / ReSTIR_GI_WPS.hlsl + the restated host (ref_gi_host.cpp)
# $(call gi_perm,<tag>,<NEE_EMISSIVE>,<extra macros>)
define gi_perm
_ref/obj/gi_$(1)_shader.o: _ref/gen/.stamp ref_hlsl/ref_pass_shader.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $$@ ref_hlsl/ref_pass_shader.cpp -Dhlsl=hlsl_gi '-DZR_SHADER="ZetaRenderPass/IndirectLighting/ReSTIR_GI/ReSTIR_GI.hlsl"' \
	    -DZR_ENTRY=zrefp_shader_gi -DZR_LOCAL_CB=cb_ReSTIR_GI -DZR_HAS_SCENE=1 -DZR_HAS_LIGHTS=$(2) -DNEE_EMISSIVE=$(2) $(3)
_ref/obj/gi_$(1)_host.o: _ref/gen/.stamp ref_hlsl/ref_gi_host.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -c -o $$@ ref_hlsl/ref_gi_host.cpp
_ref/libzref_gi_$(1).so: _ref/obj/gi_$(1)_shader.o _ref/obj/gi_$(1)_host.o
	$(HLSL_CXX) -shared -o $$@ _ref/obj/gi_$(1)_shader.o _ref/obj/gi_$(1)_host.o
endef
$(eval $(call gi_perm,e0,0,))
$(eval $(call gi_perm,e1,1,))
$(eval $(call gi_perm,e1p,1,-DUSE_PRESAMPLED_SETS))
# Variants/ReSTIR_GI_LVG.hlsl: presampled sets + the light voxel grid (g_lvg : register(t8))
$(eval $(call gi_perm,e1l,1,-DUSE_PRESAMPLED_SETS -DUSE_LVG))

# ---- ReSTIR DI: emissive (K5 / K6: ReSTIR_DI_Temporal{,_WPS}.hlsl + ReSTIR_DI_Spatial.hlsl) and sun + sky (K7 / K8: SkyDI_Temporal.hlsl + SkyDI_Spatial.hlsl)
DL := ZetaRenderPass/DirectLighting
# $(call di_lib,<tag>,<sky 0/1>,<temporal file>,<spatial file>,<cb type>,<temporal globals mode>,<spatial globals mode>,<extra macros>)
define di_lib
_ref/obj/di_$(1)_temporal.o: _ref/gen/.stamp ref_hlsl/ref_pass_shader.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $$@ ref_hlsl/ref_pass_shader.cpp -Dhlsl=hlsl_di_t '-DZR_SHADER="$(DL)/$(3)"' \
	    -DZR_ENTRY=zrefp_shader_di_temporal -DZR_LOCAL_CB=$(5) -DZR_HAS_SCENE=0 -DZR_HAS_LIGHTS=0 -DZR_DI_GLOBALS=$(6) $(8)
_ref/obj/di_$(1)_spatial.o: _ref/gen/.stamp ref_hlsl/ref_pass_shader.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $$@ ref_hlsl/ref_pass_shader.cpp -Dhlsl=hlsl_di_s '-DZR_SHADER="$(DL)/$(4)"' \
	    -DZR_ENTRY=zrefp_shader_di_spatial -DZR_LOCAL_CB=$(5) -DZR_HAS_SCENE=0 -DZR_HAS_LIGHTS=0 -DZR_DI_GLOBALS=$(7) $(8)
_ref/obj/di_$(1)_host.o: _ref/gen/.stamp ref_hlsl/ref_di_host.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -DZR_DI_SKY=$(2) -c -o $$@ ref_hlsl/ref_di_host.cpp
_ref/libzref_di_$(1).so: _ref/obj/di_$(1)_temporal.o _ref/obj/di_$(1)_spatial.o _ref/obj/di_$(1)_host.o
	$(HLSL_CXX) -shared -o $$@ _ref/obj/di_$(1)_temporal.o _ref/obj/di_$(1)_spatial.o _ref/obj/di_$(1)_host.o
endef
$(eval $(call di_lib,e1,0,Emissive/ReSTIR_DI_Temporal.hlsl,Emissive/ReSTIR_DI_Spatial.hlsl,cb_ReSTIR_DI,1,2,))
$(eval $(call di_lib,e1p,0,Emissive/ReSTIR_DI_Temporal.hlsl,Emissive/ReSTIR_DI_Spatial.hlsl,cb_ReSTIR_DI,1,2,-DUSE_PRESAMPLED_SETS))
# (round 6) the emissive DI shaders with the reference's half-vector copy shift compiled in (Params.hlsli: USE_HALF_VECTOR_COPY_SHIFT, 0 in the reference's tree)
$(eval $(call di_lib,e1h,0,Emissive/ReSTIR_DI_Temporal.hlsl,Emissive/ReSTIR_DI_Spatial.hlsl,cb_ReSTIR_DI,1,2,-DUSE_HALF_VECTOR_COPY_SHIFT=1))
$(eval $(call di_lib,sky,1,Sky/SkyDI_Temporal.hlsl,Sky/SkyDI_Spatial.hlsl,cb_SkyDI,3,4,))

# ---- post stack: AutoExposure_Histogram.hlsl + AutoExposure_WeightedAvg.hlsl (compute; g_hist is a root UAV) and Display.hlsl (pixel shader)
AE := ZetaRenderPass/AutoExposure
_ref/obj/post_ae_hist.o: _ref/gen/.stamp ref_hlsl/ref_pass_shader.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $@ ref_hlsl/ref_pass_shader.cpp -Dhlsl=hlsl_ae_hist '-DZR_SHADER="$(AE)/AutoExposure_Histogram.hlsl"' \
	    -DZR_ENTRY=zrefp_shader_ae_hist -DZR_LOCAL_CB=cbAutoExposureHist -DZR_HAS_SCENE=0 -DZR_HAS_LIGHTS=0 -DZR_ROOT_UAV=g_hist
_ref/obj/post_ae_avg.o: _ref/gen/.stamp ref_hlsl/ref_pass_shader.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $@ ref_hlsl/ref_pass_shader.cpp -Dhlsl=hlsl_ae_avg '-DZR_SHADER="$(AE)/AutoExposure_WeightedAvg.hlsl"' \
	    -DZR_ENTRY=zrefp_shader_ae_avg -DZR_LOCAL_CB=cbAutoExposureHist -DZR_HAS_SCENE=0 -DZR_HAS_LIGHTS=0 -DZR_ROOT_UAV=g_hist
_ref/obj/post_display.o: _ref/gen/.stamp ref_hlsl/ref_pass_display.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $@ ref_hlsl/ref_pass_display.cpp -Dhlsl=hlsl_display
_ref/obj/post_host.o: _ref/gen/.stamp ref_hlsl/ref_post_host.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -c -o $@ ref_hlsl/ref_post_host.cpp
_ref/libzref_post.so: _ref/obj/post_ae_hist.o _ref/obj/post_ae_avg.o _ref/obj/post_display.o _ref/obj/post_host.o
	$(HLSL_CXX) -shared -o $@ $^

# ---- auxiliary passes: PreLighting (K2 / K3 / K4), SkyViewLUT (K17), Compositing + FireflyFilter, TAA -- ref_pass_aux.cpp per shader + ref_aux_host.cpp
RP := ZetaRenderPass
# $(call aux_obj,<tag>,<ZR_AUX>,<file>,<extra macros>)
define aux_obj
_ref/obj/aux_$(1).o: _ref/gen/.stamp ref_hlsl/ref_pass_aux.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -fvisibility=hidden -c -o $$@ ref_hlsl/ref_pass_aux.cpp -Dhlsl=hlsl_aux_$(1) '-DZR_SHADER="$(RP)/$(3)"' -DZR_ENTRY=zrefp_shader_$(1) -DZR_AUX=$(2) $(4)
AUX_OBJS += _ref/obj/aux_$(1).o
endef
$(eval $(call aux_obj,estimate_power,1,PreLighting/EstimateTriEmissivePower.hlsl,))
$(eval $(call aux_obj,presample,2,PreLighting/PresampleEmissives.hlsl,-DZR_LOCAL_CB=cbPresampling))
$(eval $(call aux_obj,build_lvg,3,PreLighting/BuildLightVoxelGrid.hlsl,-DZR_LOCAL_CB=cbLVG))
$(eval $(call aux_obj,sky_lut,4,Sky/SkyViewLUT.hlsl,-DZR_LOCAL_CB=cbSky))
$(eval $(call aux_obj,compositing,5,Compositing/Compositing.hlsl,-DZR_LOCAL_CB=cbCompositing))
$(eval $(call aux_obj,firefly,6,Compositing/FireflyFilter.hlsl,-DZR_LOCAL_CB=cbFireflyFilter))
$(eval $(call aux_obj,taa,7,TAA/TAA.hlsl,-DZR_LOCAL_CB=cbTAA))
_ref/obj/aux_host.o: _ref/gen/.stamp ref_hlsl/ref_aux_host.cpp ref_hlsl/ref_dispatch.h $(PASS_HDRS)
	mkdir -p _ref/obj
	$(HLSL_CXX) $(HLSL_FLAGS) -c -o $@ ref_hlsl/ref_aux_host.cpp
_ref/libzref_aux.so: $(AUX_OBJS) _ref/obj/aux_host.o
	$(HLSL_CXX) -shared -o $@ $^

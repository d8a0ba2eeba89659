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
HLSL_FLAGS := -std=c++17 -O2 -ffp-contract=off -fno-fast-math -mavx2 -mfma -mf16c -fPIC -fpermissive -w -Iref_hlsl -I_ref/gen
HLSL_ROOTS := ZetaRenderPass/Common/BSDFSampling.hlsli ZetaRenderPass/Common/RT.hlsli ZetaRenderPass/Common/GBuffers.hlsli ZetaRenderPass/Common/FrameConstants.h

all: _ref/libzref.so _ref/libzref_hlsl.so

_ref/libzref.so: ref_driver.cpp ref_shim.h
	mkdir -p _ref
	$(CXX) $(FLAGS) -shared -o $@ $(REF)/Source/ZetaCore/Math/Common.cpp $(REF)/Source/ZetaCore/Math/Sampling.cpp ref_driver.cpp

_ref/gen/.stamp: ref_hlsl/hlsl2cpp.py
	mkdir -p _ref/gen
	python3 ref_hlsl/hlsl2cpp.py $(REF)/Source _ref/gen $(HLSL_ROOTS)
	touch $@

_ref/libzref_hlsl.so: _ref/gen/.stamp ref_hlsl/ref_hlsl_driver.cpp ref_hlsl/hlsl_shim.h ref_hlsl/hlsl_resources.h zro_kat_layout.h
	$(CXX) $(HLSL_FLAGS) -shared -o $@ ref_hlsl/ref_hlsl_driver.cpp

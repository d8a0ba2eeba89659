# Builds the reference's own CPU sources for the alias table / Kahan sum / oct32 / half, IN PLACE from /root/reference,
# into oracle/_ref/libzref.so (git-ignored; travels to the GPU box with the snapshot).  g++ directly on the few source
# files -- the reference's CMake build is not run.  Usage: make -C oracle -f _ref.mk
REF ?= /root/reference
CXX ?= g++
FLAGS := -std=c++20 -O2 -mavx2 -mfma -mf16c -fPIC -w -include ref_shim.h -I$(REF)/Source/ZetaCore -I$(REF)/Source -I$(REF)/External -DNDEBUG

_ref/libzref.so: ref_driver.cpp ref_shim.h
	mkdir -p _ref
	$(CXX) $(FLAGS) -shared -o $@ $(REF)/Source/ZetaCore/Math/Common.cpp $(REF)/Source/ZetaCore/Math/Sampling.cpp ref_driver.cpp

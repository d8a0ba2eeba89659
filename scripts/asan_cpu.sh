# The CPU test suite with AddressSanitizer builds of the oracle and of the host executor (which compiles the HIP stage functions for the host): out-of-bounds
# reads / writes in the per-pixel code show up here without a GPU.  The two scene-loader error-path tests are skipped: they throw C++ exceptions inside a
# shared object loaded under an LD_PRELOADed libasan, whose __cxa_throw interceptor aborts ("real___cxa_throw != 0") -- an artefact of the preload.
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
FL="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mfma -mavx2 -fsanitize=address -fno-omit-frame-pointer -Wno-unused-function -Wno-unused-variable"
(cd $R/tests/hostexec && g++ $FL -shared -o $T/libzhx.so hostexec.cpp)
(cd $R/oracle && g++ $FL -shared -o $T/libzro.so zro_render.cpp)
cp $R/tests/hostexec/libzhx.so $T/libzhx.orig; cp $R/oracle/libzro.so $T/libzro.orig
trap "cp $T/libzhx.orig $R/tests/hostexec/libzhx.so; cp $T/libzro.orig $R/oracle/libzro.so" EXIT
cp $T/libzhx.so $R/tests/hostexec/libzhx.so; cp $T/libzro.so $R/oracle/libzro.so
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 python -m pytest tests -q -m "not gpu" -p no:cacheprovider -n 4 \
  --deselect tests/test_scene_io.py::test_native_loader_reports_errors --deselect tests/test_scene_io.py::test_native_loader_rejects_malformed_files

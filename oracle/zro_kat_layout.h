// ORACLE tooling -- test infrastructure only.
// Row layouts (floats per row; uint32 values travel as float bit patterns) of the function-level known-answer probes that are
// implemented three times and compared bit for bit (tests/test_ref_pins.py):
//   reference code compiled from /root/reference (oracle/ref_hlsl/ref_hlsl_driver.cpp, zrefh_kat_*),
//   the oracle restatement (oracle/zro_kat.h, zro_kat2_*), the HIP stage functions run on the host (tests/hostexec, zhx_kat_*).
#pragma once
#define ZR_KAT_SAMPLING_IN  4
#define ZR_KAT_SAMPLING_OUT 32
#define ZR_KAT_MATH_IN      21
#define ZR_KAT_MATH_OUT     48
#define ZR_KAT_RT_IN        17
#define ZR_KAT_RT_OUT       12
#define ZR_KAT_BSDF_IN      26
#define ZR_KAT_BSDF_OUT     61
#define ZR_KAT_GBUFFER_IN   2
#define ZR_KAT_GBUFFER_OUT  4

// Force-included when compiling the reference's own sources (in place, under /root/reference) with g++ on Linux.
// No reference source is copied or modified: this header only supplies the MSVC-isms the sources assume.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#define __forceinline inline __attribute__((always_inline))
#define __vectorcall
#define _aligned_malloc(size, al) aligned_alloc((al), (((size) + (al) - 1) / (al)) * (al))
#define _aligned_free free
#define __debugbreak() __builtin_trap()
// LP64 vs LLP64: the reference calls Math::Max(size_t, 4llu); on Linux size_t is `unsigned long`, so overload
// resolution needs an exact match.
namespace ZetaRay { namespace Math {
    constexpr unsigned long Max(unsigned long a, unsigned long long b) { return a > b ? a : (unsigned long)b; }
    constexpr unsigned long Min(unsigned long a, unsigned long long b) { return a < b ? a : (unsigned long)b; }
} }

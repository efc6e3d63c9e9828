// Hand-written stand-in for the one generated header the reference needs
// (native/src/seal/util/config.h.in is normally filled in by CMake). It selects the
// same options the survey's oracle build used: Release, HEXL OFF,
// no MSGSL/ZSTD, system zlib, __int128 intrinsics, Blake2xb default PRNG.
// Variant for the DROP-IN build (integration/Makefile compiles native/src/seal/util/mempool.cpp against it): the same options as
// oracle/ref_config, except that the reference's allocator hook SEAL_MALLOC / SEAL_FREE (util/defines.h:170-179) points at
// integration/seal_evaluator_hip.cpp, which hands the pool page-aligned chunks and forgets the device mirrors of a chunk that is
// freed.  SEAL_USE_ALIGNED_ALLOC stays off because util/gcc.h would define the two macros itself.
#pragma once

#define SEAL_VERSION "4.4.3"
#define SEAL_VERSION_MAJOR 4
#define SEAL_VERSION_MINOR 4
#define SEAL_VERSION_PATCH 3

// C++17 features
#define SEAL_USE_STD_BYTE
#define SEAL_USE_SHARED_MUTEX
#define SEAL_USE_IF_CONSTEXPR
#define SEAL_USE_MAYBE_UNUSED
#define SEAL_USE_NODISCARD
#define SEAL_USE_STD_FOR_EACH_N

// Security
#define SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT
#define SEAL_DEFAULT_PRNG Blake2xb

// Intrinsics (GCC/Clang on x86-64)
#define SEAL_INTRIN_HEADER_FOUND
#define SEAL_USE_INTRIN
#define SEAL_USE___INT128
#define SEAL_USE___BUILTIN_CLZLL
#define SEAL_USE__ADDCARRY_U64
#define SEAL_USE__SUBBORROW_U64

// Zero memory functions
#define SEAL_USE_EXPLICIT_BZERO

// Third-party dependencies: zlib only (the system's <zlib.h>: lets the wire-format tests check compr_mode_type::zlib streams
// against the reference); HEXL OFF, no MSGSL, no ZSTD (its header is not in the image)
#define SEAL_USE_ZLIB

// allocator hook (device-resident drop-in)
#include <cstddef>
extern "C" void *sealhip_host_alloc(std::size_t size);
extern "C" void sealhip_host_free(void *ptr);
#define SEAL_MALLOC(size) static_cast<seal_byte *>(::sealhip_host_alloc(size))
#define SEAL_FREE(ptr) ::sealhip_host_free(ptr)

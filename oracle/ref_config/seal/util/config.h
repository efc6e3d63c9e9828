// Hand-written stand-in for the one generated header the reference needs
// (native/src/seal/util/config.h.in is normally filled in by CMake). It selects the
// same options the survey's oracle build used: Release, HEXL OFF,
// no MSGSL/ZSTD, system zlib, __int128 intrinsics, Blake2xb default PRNG.
// TEST INFRASTRUCTURE ONLY: used by oracle/Makefile to compile the read-only reference
// tree into oracle/_ref/. Nothing under seal_amd/ includes it.
#pragma once

#define SEAL_VERSION "4.4.3"
#define SEAL_VERSION_MAJOR 4
#define SEAL_VERSION_MINOR 4
#define SEAL_VERSION_PATCH 3

// C++17 features
#define SEAL_USE_STD_BYTE
#define SEAL_USE_ALIGNED_ALLOC
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

#pragma once
// Shared by the capi_*.cpp files: the extern "C" layer of libsealhip.so — see include/sealhip.h.  Every body is closed by the same
// exception-to-HRESULT ladder as the reference's C export layer (SEAL_C_CATCH_ALL,
// native/src/seal/c/defines.h:75-97); null handles return E_POINTER like IfNullRet
// (native/src/seal/c/utilities.h).
#include "../../include/sealhip.h"
#include "evaluator.h"
#include "ckks_encoder.h"
#include "decryptor.h"
#include "keygen.h"
#include "serial.h"
#include "xof.h"
#include <atomic>
#include <cstring>
#include <new>
#include <string>

namespace sealhip
{
    // message of the calling thread's last failed call (SealHip_LastError); one instance for all capi_*.cpp files (capi_core.cpp)
    std::string &capi_last_error();
    // SEALContext_Destroy: the ContextData handles given out for this context die with it (capi_containers.cpp)
    void capi_forget_context(const class Context *context);
}
using namespace sealhip;

namespace
{

    struct EncParams
    {
        uint8_t scheme = 0;
        uint64_t n = 0;
        std::vector<uint64_t> coeff_modulus;
        uint64_t plain_modulus = 0;
    };
    struct Timer
    {
        hipEvent_t e0 = nullptr, e1 = nullptr;
    };

#define SHL_TRY \
    try         \
    {
#define SHL_CATCH                               \
    }                                           \
    catch (const std::invalid_argument &e)      \
    {                                           \
        sealhip::capi_last_error() = e.what();                \
        return SHL_E_INVALIDARG;                \
    }                                           \
    catch (const std::out_of_range &e)          \
    {                                           \
        sealhip::capi_last_error() = e.what();                \
        return SHL_E_INVALID_INDEX;             \
    }                                           \
    catch (const std::logic_error &e)           \
    {                                           \
        sealhip::capi_last_error() = e.what();                \
        return SHL_COR_E_INVALIDOPERATION;      \
    }                                           \
    catch (const std::runtime_error &e)         \
    {                                           \
        sealhip::capi_last_error() = e.what();                \
        return SHL_COR_E_IO;                    \
    }                                           \
    catch (const std::bad_alloc &)              \
    {                                           \
        sealhip::capi_last_error() = "out of device memory";  \
        return SHL_E_OUTOFMEMORY;               \
    }                                           \
    catch (...)                                 \
    {                                           \
        sealhip::capi_last_error() = "unexpected exception";  \
        return SHL_E_UNEXPECTED;                \
    }                                           \
    return SHL_S_OK;

#define IfNullRet(p, r) \
    if (!(p))           \
    return (r)

    template <typename T>
    T *as(void *p)
    {
        return static_cast<T *>(p);
    }

    void hip_ok(hipError_t e, const char *what)
    {
        if (e != hipSuccess)
            throw std::runtime_error(std::string("HIP failure in ") + what + ": " + hipGetErrorString(e));
    }

    // dest = src unless they are the same object (the sealc "destination" convention)
    Ciphertext &prepare_dest(void *encrypted, void *destination)
    {
        Ciphertext *src = as<Ciphertext>(encrypted), *dst = as<Ciphertext>(destination);
        if (src != dst)
            *dst = *src;
        return *dst;
    }
} // namespace

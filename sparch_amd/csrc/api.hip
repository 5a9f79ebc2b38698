// ABI bookkeeping entry points of libsparch_hip.so.
#include "common.h"

static thread_local int g_last_hip_error = 0;
void sparch_note_hip_error(int e) { g_last_hip_error = e; }

// 2: bf16 spike planes (s16_out, s_prev16, spike16 GEMMs), adam
// 3: step variants of the recurrent cells (any hidden size), BatchNorm sums out of the cell backward kernels,
//    optional bf16 saved states, device-side skip words on adam / bn_finalize, readout up to 256 classes
extern "C" int sparch_abi_version(void) { return 3; }

extern "C" const char* sparch_last_hip_error(void) {
    return hipGetErrorString((hipError_t)g_last_hip_error);
}

extern "C" const char* sparch_strerror(int code) {
    switch (code) {
        case SPARCH_OK: return "ok";
        case SPARCH_EINVAL: return "invalid argument (shape, null pointer or unsupported size)";
        case SPARCH_EALIGN: return "pointer or leading dimension is not 16-byte aligned";
        case SPARCH_EWORKSPACE: return "workspace too small";
        case SPARCH_ELAUNCH: return "HIP kernel launch failed";
        case SPARCH_ETIMEOUT: return "in-kernel wait for a neighbouring workgroup gave up";
        default: return "unknown error";
    }
}

extern "C" int sparch_device_cus(void) {
    SPARCH_ENTER();
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return cus;
}

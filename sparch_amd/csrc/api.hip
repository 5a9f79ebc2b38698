// ABI bookkeeping entry points of libsparch_hip.so.
#include "common.h"

static thread_local int g_last_hip_error = 0;
void sparch_note_hip_error(int e) { g_last_hip_error = e; }

// 2: bf16 spike planes (s16_out, s_prev16, spike16 GEMMs), adam
// 3: step variants of the recurrent cells (any hidden size), BatchNorm sums out of the cell backward kernels,
//    optional bf16 saved states, device-side skip words on adam / bn_finalize, readout up to 256 classes
// 4: sparch_set_operand_precision (bf16 operands, fp32 accumulation) for the GEMMs and the recurrent cells
// 5: s_out optional in the cell forwards; the status word is SPARCH_STATUS_WORDS uint32 (raised, skipped optimizer
//    steps, kernel id, time step); sparch_bn_finalize takes BatchNorm's num_batches_tracked; sparch_adam_step
//    counts the steps it skips; operand precision is a per-call argument (sparch_set_operand_precision is gone)
extern "C" int sparch_abi_version(void) { return 5; }

// Operand precision of the matrix products: a PER-CALL argument of every entry point that multiplies (ABI v5; rounds
// 1-2 had a process-wide switch, sparch_set_operand_precision).  The entry point opens a PrecisionScope for the
// duration of the call — a thread-local the dispatch helpers below it read — so two models of different precision in
// one process, or two host threads, never see each other's setting, and nothing outlives the call.
static thread_local int tl_operand_precision = SPARCH_PRECISION_FP32_EXACT;
int sparch_operand_bf16(void) { return tl_operand_precision == SPARCH_PRECISION_BF16; }
PrecisionScope::PrecisionScope(int precision)
    : prev(tl_operand_precision), ok(precision == SPARCH_PRECISION_FP32_EXACT || precision == SPARCH_PRECISION_BF16) {
    if (ok) tl_operand_precision = precision;
}
PrecisionScope::~PrecisionScope() { tl_operand_precision = prev; }

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

// ---- host side: the reference's initial-state draws.
// Every forward the reference draws u0, [w0,] s0 of every layer with torch.rand from the global CPU generator
// (snns.py:286-287, 423-425, 558-559, 700-702, 812): MT19937, one 32-bit output per element,
// x = (y & (2^24 - 1)) * 2^-24.  At the headline shape that is 1.6 M numbers per step, and torch's serial loop
// (~5 ns per number) made the HOST the bound of the training step: 8.5 ms of draws against 7.6 ms of GPU work.
// This routine produces the same stream from the same state (the caller parses / writes back the generator's
// serialized state) with the block regeneration and the tempering as vectorisable loops.
namespace {
inline void mt_regenerate(uint32_t* s) {
    constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, A = 0x9908b0dfu;
    int i = 0;
    for (; i < 624 - 397; ++i) {
        const uint32_t y = (s[i] & UPPER) | (s[i + 1] & LOWER);
        s[i] = s[i + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    for (; i < 623; ++i) {
        const uint32_t y = (s[i] & UPPER) | (s[i + 1] & LOWER);
        s[i] = s[i + 397 - 624] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    const uint32_t y = (s[623] & UPPER) | (s[0] & LOWER);
    s[623] = s[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
}
}  // namespace

extern "C" int sparch_mt19937_uniform_f32(uint32_t* key, int* pos, size_t n, float* out) {
    if (!key || !pos || (!out && n) || *pos < 0 || *pos > 624) return SPARCH_EINVAL;
    int p = *pos;
    while (n) {
        if (p == 624) { mt_regenerate(key); p = 0; }
        const size_t m = n < (size_t)(624 - p) ? n : (size_t)(624 - p);
        const uint32_t* s = key + p;
        for (size_t j = 0; j < m; ++j) {
            uint32_t y = s[j];
            y ^= y >> 11;
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= y >> 18;
            out[j] = (float)(y & 0xFFFFFFu) * (1.0f / 16777216.0f);
        }
        out += m; n -= m; p += (int)m;
    }
    *pos = p;
    return SPARCH_OK;
}

// f-3: event lists -> dense binned spike counts on the device.
//
// Replaces SpikingDataset.__getitem__ (spiking_datasets.py:66-78): per sample
//     times = np.digitize(firing_times, np.linspace(0, max_time, nb_steps))     # 1-based bin of each event
//     x     = sparse(idx=[times, units], val=1, size=(nb_steps, nb_units)).to_dense()   # duplicates add up
// for a whole batch at once.  np.digitize(t, bins) (right=False) = number of edges <= t, so an event at
// t in [0, bins[1]) lands in row 1 and row 0 stays empty (a quirk of the reference, preserved); an event
// with t >= max_time would index row nb_steps, which the reference's sparse constructor rejects: such
// events (and negative times / out-of-range units) are counted in *n_dropped and skipped.
// Edges are evaluated in fp64 exactly as np.linspace does (start + j*step, last edge = stop).
// Counts are accumulated with float atomics of integer values: exact and order-independent.
#include "common.h"

namespace {

__global__ void bin_events_kernel(long long n_events, const float* __restrict__ times,
                                  const int* __restrict__ units, const long long* __restrict__ offsets,
                                  int n_samples, int nb_steps, int nb_units, double max_time,
                                  float* __restrict__ out, unsigned* __restrict__ n_dropped) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_events) return;
    // sample of event i: largest b with offsets[b] <= i (binary search over the batch)
    int lo = 0, hi = n_samples;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const double t = (double)times[i];
    const int u = units[i];
    const double step = max_time / (double)(nb_steps - 1);
    auto edge = [&](int j) { return j == nb_steps - 1 ? max_time : (double)j * step; };
    int k = (int)floor(t / step);             // candidate: edges 0..k are <= t
    k = max(-1, min(k, nb_steps - 1));
    while (k + 1 < nb_steps && edge(k + 1) <= t) ++k;   // fix rounding of the division, either way
    while (k >= 0 && edge(k) > t) --k;
    const int bin = k + 1;                    // np.digitize
    if (t < 0.0 || bin >= nb_steps || u < 0 || u >= nb_units) {
        atomicAdd(n_dropped, 1u);
        return;
    }
    atomicAdd(out + ((size_t)lo * nb_steps + bin) * nb_units + u, 1.0f);
}

// ---- a11 (exp.py:355-356): a batch of binned spike counts uploaded as ONE BYTE per element and expanded on the
// device.  The reference copies the dense float batch over PCIe every step ((256, 250, 700) fp32 = 179 MB against
// 45 MB as uint8); counts are small non-negative integers (spiking_datasets.py:71-78), exact in uint8 up to 255
// and in bf16 up to 256.  One pass writes the bf16 plane the first layer's GEMMs read (rows padded to ldp
// elements, zeros behind column K — the layout of sparch_plane_bf16_exact) and, on request, the fp32 tensor.
__global__ __launch_bounds__(256) void expand_counts_kernel(long long M, int K, const uint8_t* __restrict__ c,
                                                            uint16_t* __restrict__ plane, int ldp,
                                                            float* __restrict__ x, int ldx) {
    const long long n = M * (long long)ldp;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / ldp;
        const int k = (int)(i - m * ldp);
        const unsigned v = k < K ? (unsigned)c[m * K + k] : 0u;
        const float f = (float)v;                       // <= 255: exact in bf16 (8 significant bits)
        plane[i] = (uint16_t)(__float_as_uint(f) >> 16);
        if (x && k < K) x[m * ldx + k] = f;
    }
}

}  // namespace

extern "C" int sparch_expand_counts_u8(long long M, int K, const uint8_t* counts, uint16_t* plane, int ldp,
                                       float* x, int ldx, void* stream) {
    SPARCH_ENTER();
    if (M <= 0 || K <= 0 || !counts || !plane || ldp < K || (ldp % 8) != 0 || (x && ldx < K)) return SPARCH_EINVAL;
    if (!aligned16(plane)) return SPARCH_EALIGN;
    const long long n = M * (long long)ldp;
    const unsigned grid = (unsigned)((n + 256 * 8 - 1) / (256 * 8) < 65535 * 16 ? (n + 256 * 8 - 1) / (256 * 8) : 65535 * 16);
    hipLaunchKernelGGL(expand_counts_kernel, dim3(grid ? grid : 1), dim3(256), 0, (hipStream_t)stream, M, K, counts,
                       plane, ldp, x, ldx);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

extern "C" int sparch_bin_events(long long n_events, const float* times, const int* units,
                                 const long long* sample_offsets, int n_samples, int nb_steps, int nb_units,
                                 double max_time, float* out, uint32_t* n_dropped, void* stream) {
    SPARCH_ENTER();
    if (n_events < 0 || n_samples <= 0 || nb_steps < 2 || nb_units <= 0 || !(max_time > 0.0) || !sample_offsets ||
        !out || !n_dropped || (n_events > 0 && (!times || !units)))
        return SPARCH_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, (size_t)n_samples * nb_steps * nb_units * sizeof(float), st) != hipSuccess)
        return SPARCH_ELAUNCH;
    if (hipMemsetAsync(n_dropped, 0, sizeof(uint32_t), st) != hipSuccess) return SPARCH_ELAUNCH;
    if (n_events == 0) return SPARCH_OK;
    hipLaunchKernelGGL(bin_events_kernel, dim3((unsigned)((n_events + 255) / 256)), dim3(256), 0, st, n_events,
                       times, units, sample_offsets, n_samples, nb_steps, nb_units, max_time, out, n_dropped);
    SPARCH_CHECK_LAUNCH();
    return SPARCH_OK;
}

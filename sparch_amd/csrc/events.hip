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

}  // namespace

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

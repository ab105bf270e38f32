// kvm_ed_tiled.hip -- LDS-tiled ED kernel (placeholder: routes to the wave kernel until implemented).
#include "kvm_internal.h"
hipError_t kvm_launch_ed_tiled(const double *series, const double2 *prefix, int64_t n, const kvm_query_dev &qd,
                               const kvm_tile *tiles, int64_t n_tiles, int32_t *out_off, double *out_dist,
                               int64_t cap, unsigned long long *counters, hipStream_t s) {
    return kvm_launch_ed_wave(series, prefix, n, qd, tiles, n_tiles, out_off, out_dist, cap, counters, s);
}

// kq_pruned.hip -- pruned forward path of the pre-detection filter (placeholder: not built yet).
#include "kq_device.hpp"

namespace kq {

bool pruned_supported(const Geom &) { return false; }
size_t pruned_table_elems(const Geom &) { return 1; }
void launch_pruned_tables(hipStream_t, const Geom &, const ChanDev &, float2 *, int) {}
void launch_filter_pruned(hipStream_t, const Geom &, const ChanDev &, const Planes &, const float2 *, const float2 *,
                          const float2 *, int, int) {}

}  // namespace kq

// find_kernels_counted.hip -- the find kernels once more, keeping the request counters
// (FindArgs::stats; blurrily_storage_set_stats), in namespace blurrily::counted.  See find_kernels.hip.
#define BLURRILY_COUNTED 1
#include "find_kernels.hip"

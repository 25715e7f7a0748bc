// msda_f64.hip — C-ABI entry points msda_fwd_f64 / msda_bwd_f64 (storage type double).
#include "msda_launch.hpp"

MSDA_DEFINE_ENTRY_POINTS(f64, double)

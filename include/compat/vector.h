/* Forwarding header: the reference's include/vector.h, served by the MI355X engine's compat layer.
 * Build the reference's main.cpp with -Iinclude/compat instead of the reference's -I include. */
#include "../arm_spmv_compat.hpp"

/* Forwarding header: the reference's main.cpp includes "mmio.h" (main.cpp:4) without using anything from it;
 * the harness-side Matrix Market parsing lives in arm-spmv_amd/host/mm_banner.h + mtx_io.cpp. */
#include "../../arm-spmv_amd/host/mm_banner.h"

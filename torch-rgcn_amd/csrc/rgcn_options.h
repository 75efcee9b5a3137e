// Tuning options shared by every translation unit of librgcn_hip.so (no HIP dependency: rgcn_host.cpp includes it too).
#pragma once
#include <stdint.h>

// ---- tuning options (rgcn_set_option / rgcn_get_option, include/rgcn_hip.h): plain integers set by the host side -- the library never
// reads the environment.  Index = position in RGCN_OPTION_NAMES (rgcn_host.cpp).
enum RgcnOpt {
  RGCN_OPT_BWD_NW, RGCN_OPT_GEMM_BM, RGCN_OPT_SPMM_U, RGCN_OPT_WGRAD_RG, RGCN_OPT_WGRAD_U,
  RGCN_OPT_BWD_ABL,      // honoured by the ablation build only (-DRGCN_ABLATIONS)
  RGCN_OPT_COUNT
};
extern "C" int32_t rgcn_option_value(int index);

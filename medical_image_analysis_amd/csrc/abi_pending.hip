// abi_pending.hip -- entry points declared in include/mxvl.h whose kernels are not written yet.
// They fail loudly (MXVL_ERR_UNSUPPORTED); nothing falls back to another implementation.
#include "mxvl_common.h"
extern "C" {
int mxvl_conv1d_fwd(const mxvl_conv1d_desc*, void*) { return MXVL_ERR_UNSUPPORTED; }
int mxvl_conv1d_bwd(const mxvl_conv1d_bwd_desc*, void*) { return MXVL_ERR_UNSUPPORTED; }
int mxvl_conv1d_update(const void*, void*, const void*, const void*, void*, int, int, int, int, int, void*) {
  return MXVL_ERR_UNSUPPORTED;
}
int mxvl_state_update(void*, const void*, const void*, const void*, const void*, const void*, const void*,
                      const void*, const void*, void*, int, int, int, int, int, void*) {
  return MXVL_ERR_UNSUPPORTED;
}
}

// MFMA (v_mfma_f64_16x16x4_f64) variant of the IALM streaming pass.  Placeholder until the
// kernel lands: reports "unsupported" so the driver keeps using the LDS/VALU kernel.
#include "swk_internal.h"

namespace swk {

bool ialm_v2_supported(int) { return false; }
void launch_ialm_pass_v2(hipStream_t, const IalmBuffers &, int) {}

}  // namespace swk

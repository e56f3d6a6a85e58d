// scratch: phase timing of kp_finish_kernel (wall_clock64 = 100 MHz)
#define MV_KP_PROFILE 1
#include "../../mac-vo_amd/csrc/kp_select.hip"
extern "C" int kp_probe_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kp_stamps), sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}

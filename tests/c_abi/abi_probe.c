/* Plain-C consumer of include/macvo_hip.h: proves the boundary is a C ABI (C99, no C++ in the header) and that struct
 * layouts are what the ctypes binding in mac-vo_amd/_lib.py assumes.  Built and run by tests/test_abi_and_host.py with gcc;
 * it only calls host-side entry points (no GPU needed). */
#include "macvo_hip.h"
#include <stddef.h>
#include <stdio.h>
#include <string.h>

int main(void) {
    mvLMParams lm;
    mvFramePipeConfig c;
    memset(&c, 0, sizeof c);
    mv_lm_default_params(&lm);
    c.H = 480; c.W = 640; c.C = 256; c.pairs = 2; c.iters = 12; c.radius = 4;
    c.selector_mode = MV_KP_NODEPTH; c.graph_type = MV_GRAPH_DISP; c.num_point = 200; c.lm = lm;
    printf("abi=%d\n", mv_abi_version());
    printf("sizeof mvLMParams=%zu mvFramePipeConfig=%zu mvFrameInputs=%zu mvKpSelectParams=%zu mvMatchCovParams=%zu\n",
           sizeof(mvLMParams), sizeof(mvFramePipeConfig), sizeof(mvFrameInputs), sizeof(mvKpSelectParams),
           sizeof(mvMatchCovParams));
    printf("offsetof lm=%zu fx=%zu\n", offsetof(mvFramePipeConfig, lm), offsetof(mvFramePipeConfig, fx));
    printf("lm huber=%.3f steps=%d reject=%d\n", lm.huber_delta, lm.max_steps, lm.reject);
    printf("arena=%zu ws=%zu\n", mv_frame_pipe_arena_bytes(&c), mv_kp_select_workspace_bytes(480, 640));
    printf("err=%s\n", mv_error_string(MV_ERR_UNSUPPORTED));
    printf("MV_FB_POSE=%d MV_BF16X2=%d\n", MV_FB_POSE, MV_BF16X2);
    return 0;
}

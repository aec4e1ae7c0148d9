// Arguments of the fused rollout forward (ppo_mlp_fused.hip), shared with the C-ABI layer.
#pragma once
#include "ppo_device.h"

struct MlpArgs {
    const float *in[2];            // [M][dims[z][0]] observations of net z
    float *out[2];                 // [M][dims[z][nl]] head outputs
    const float *params;           // fp32 parameters (biases)
    const uint16_t *wfrag;         // fragment-order weight image (ppo_mlp_fused.hip), frag_off[z][l] elements in
    int M, nl, act;                // rows, linear layers, activation code of the hidden layers (as k_gemm)
    // PPO.act epilogue in the same launch (sample = 1): a ~ N(mu, sigma), log-prob, transition store of step t
    int sample, t, inject;
    int64_t act_count;
    int dims[2][LG_PPO_MAX_LAYERS + 1];
    int64_t frag_off[2][LG_PPO_MAX_LAYERS], w_off[2][LG_PPO_MAX_LAYERS], b_off[2][LG_PPO_MAX_LAYERS];
};

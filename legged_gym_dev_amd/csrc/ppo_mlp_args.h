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
    // Rollout epilogue of the PREVIOUS policy step in the same launch (lg_ppo_attach_env): blockIdx.y == 2 workgroups run
    // process_env_step of step `pp_t` for 256 envs each and, one of them, the env's deferred single-workgroup epilogue.
    int pp, pp_t, pp_use_tos;                     // pp: 0 none, 1 process step only, 2 process step + env epilogue
    int64_t pp_counter;                           // the env's step counter of that step
    const DevParams *pp_env;                      // device copy of the env's parameters (buffers: rew, reset, time_out, extras_time_outs)
    int dims[2][LG_PPO_MAX_LAYERS + 1];
    int64_t frag_off[2][LG_PPO_MAX_LAYERS], w_off[2][LG_PPO_MAX_LAYERS], b_off[2][LG_PPO_MAX_LAYERS];
};

// Device-side structs for the PPO kernels.
#pragma once
#include "lg_device.h"

#define LG_PPO_MAX_A 16
#define LG_PPO_MAX_LAYERS (LG_MAX_HIDDEN + 1)

struct GemmArgs {                  // up to 2 problems per launch (actor, critic) on blockIdx.z
    const float *A[2], *B[2], *bias[2], *aux[2];
    float *C[2], *colsum[2];
    int M[2], N[2], K[2], lda[2], ldb[2], ldc[2], ldaux[2];
    int elu;
    // B operand pre-split into three bf16 planes [n][k] (reduction contiguous), plane p at Bpl + p * pl_stride
    const uint16_t *Bpl[2];
    int64_t pl_stride;
    // first layer on a padded minibatch (observation width not a multiple of 8: 235, 169, 65): the plane image of W_0 and the
    // gathered observations both have rows of Kpl = ldbpl = width rounded up to 32 with zero pad columns, so the plane path runs on
    // whole 16-byte chunks; K / ldb above stay the true width for the fp32-operand fallback.  0 = same as K / ldb.
    int Kpl[2], ldbpl[2];
    // weight gradient against the padded observations: N = padded width is computed, nstore (true width, row length of C) is stored
    int nstore[2];
    // deterministic accumulation (lg_ppo_set_deterministic): C / colsum addresses inside [det_base, det_base + det_n) accumulate into
    // det64 as 2^-40 fixed point instead (acc_add below); nullptr = float atomics
    const float *det_base;
    long long *det64;
    int64_t det_n;
};

#define LG_PPO_MAX_SEG (2 * LG_PPO_MAX_LAYERS)

struct PpoDev {                    // passed by value to kernels
    int N, T, A, O, OC, mb_rows, world, env_offset;
    int adaptive, clipped_value;
    uint64_t seed;
    float gamma, lam, clip, value_coef, entropy_coef, desired_kl, max_grad_norm;
    int64_t num_params;
    int off_std, off_bias_actor_head, off_bias_critic_head;
    float *params, *grads, *adam_m, *adam_v;
    float *st_obs, *st_critic_obs, *st_actions, *st_rewards, *st_values, *st_returns, *st_adv, *st_log_prob, *st_mu, *st_sigma;
    uint8_t *st_dones;
    float *act_actions, *act_values, *act_log_prob, *act_mu;
    float *stats;                  // [lr, kl, value_loss_sum, surrogate_sum, adam_t, n_updates, adv_mean, adv_std]
    float *loss_acc;               // [value_loss, surrogate, grad_norm_sq]
    float *noise;
    int32_t *perm;
    float *adv_partial;
    float *mb_obs, *mb_critic_obs, *mb_actions, *mb_mu, *mb_scalars;   // minibatch gathers; scalars = [v_old, ret, adv, logp_old]
    int Op, OCp;                   // row length of mb_obs / mb_critic_obs: O / OC rounded up to 32 = one k-tile (pad columns stay zero)
    float *cur_reward_sum, *cur_episode_len, *ep_stats, *ep_ring;      // ep_ring (2, 100): last finished episodes' return / length
    int32_t *ep_ring_count;
    float *head_part;              // scratch rows of k_head_net (one per workgroup), folded by k_head_finish
    // split-bf16 image of the weight matrices W [n][k], maintained by the optimiser step: three planes (h, m, l)
    // pl_stride elements apart; segment s = one weight matrix (rows x cols at flat offset seg_off)
    uint16_t *wpl;
    int32_t *pl_dest;              // per parameter: element index inside a plane, -1 for parameters that are not weights
    int64_t pl_stride;
    int nseg;
    int64_t seg_off[LG_PPO_MAX_SEG], seg_pl[LG_PPO_MAX_SEG];
    int seg_rows[LG_PPO_MAX_SEG], seg_cols[LG_PPO_MAX_SEG];
    // Deterministic mode (lg_ppo_set_deterministic; nullptr = off).  Every sum that many workgroups contribute to -- weight-gradient
    // slices, bias column sums, the head's row sums, loss statistics, the gradient norm, the advantage moments -- is accumulated with
    // float atomics, whose order differs from run to run.  With det64 set the same contributions are added as 2^-40 fixed-point
    // 64-bit integers (integer addition is associative: any order gives the same bits) into a shadow of
    // [grads (num_params + 2) | loss_acc (4) | adv_partial (4)], and k_det_fold adds the shadow into the float buffers before they are read.
    long long *det64;
};

#define LG_DET_SCALE 1099511627776.0          // 2^40: |sum| < 2^23, resolution 9.1e-13
__device__ __forceinline__ void det_add64(long long *q, float v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(q), (unsigned long long)__double2ll_rn((double)v * LG_DET_SCALE));
}
// atomicAdd(p, v) for p inside grads / loss_acc / adv_partial
__device__ __forceinline__ void acc_add(const PpoDev &P, float *p, float v) {
    if (!P.det64) { atomicAdd(p, v); return; }
    long long i;
    if (p >= P.grads && p < P.grads + P.num_params + 2) i = p - P.grads;
    else if (p >= P.loss_acc && p < P.loss_acc + 4) i = P.num_params + 2 + (p - P.loss_acc);
    else i = P.num_params + 6 + (p - P.adv_partial);
    det_add64(P.det64 + i, v);
}
__device__ __forceinline__ void acc_add(const GemmArgs &g, float *p, float v) {
    if (g.det64 && p >= g.det_base && p < g.det_base + g.det_n) det_add64(g.det64 + (p - g.det_base), v);
    else atomicAdd(p, v);
}

// PPO.process_env_step for env i at rollout step t: rewards += gamma * V * time_outs ; store ; runner bookkeeping
// (OnPolicyRunner.learn: cur_reward_sum / cur_episode_length / rewbuffer).  to: the env's extras["time_outs"] entry, or false.
__device__ __forceinline__ void process_step_body(const PpoDev &P, float rew, bool done, bool to, int t, int i) {
    float r = rew;
    r += P.gamma * (P.st_values[(size_t)t * P.N + i] * (to ? 1.0f : 0.0f));
    P.st_rewards[(size_t)t * P.N + i] = r;
    P.st_dones[(size_t)t * P.N + i] = done ? 1 : 0;
    const float cr = P.cur_reward_sum[i] + rew, cl = P.cur_episode_len[i] + 1.0f;
    if (done) {
        atomicAdd(&P.ep_stats[0], cr);
        atomicAdd(&P.ep_stats[1], cl);
        atomicAdd(&P.ep_stats[2], 1.0f);
        const int slot = (int)((unsigned)atomicAdd(P.ep_ring_count, 1) % 100u);   // rewbuffer / lenbuffer = deque(maxlen=100)
        P.ep_ring[slot] = cr;
        P.ep_ring[100 + slot] = cl;
        P.cur_reward_sum[i] = 0.f;
        P.cur_episode_len[i] = 0.f;
    } else {
        P.cur_reward_sum[i] = cr;
        P.cur_episode_len[i] = cl;
    }
}

// standard normal of (seed, env, act() call, action index): Box-Muller on one Philox4x32-10 block (k_act_sample, k_mlp_fwd)
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t env, uint64_t step, uint32_t a) {
    uint32_t c[4] = {env, (uint32_t)step, a, 0x5eedu};
    philox4x32((uint32_t)seed, (uint32_t)(seed >> 32), c);
    float u1 = ((float)(c[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
    float u2 = (float)(c[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

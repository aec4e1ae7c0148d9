// The env step's single-workgroup epilogue, shared by k_finalize (env_kernels.hip) and the fused rollout epilogue of the learner's
// act launch (ppo_mlp_fused.hip).
#pragma once
#include "lg_device.h"

__device__ __forceinline__ float term_scale(const lg_cfg &c, int k) {
    return k < LG_NUM_REWARDS ? c.rew_scale[k] : (k - LG_NUM_REWARDS < c.num_xterms ? c.xterms[k - LG_NUM_REWARDS].scale : 0.0f);
}

// Single-workgroup epilogue: extras["episode"], extras["time_outs"] (only refreshed when >=1 env
// reset this step: the early return of LR:156-157 keeps the previous, stale values), terrain level mean.
// accumulate = 0 (lg_reset_ids: a reset made outside step()): the per-step logging sums and their step count stay untouched.
// One workgroup of 256 threads: k_finalize, or one block of the learner's act launch (ppo_mlp_fused.hip) when the env defers it.
__device__ __forceinline__ void finalize_body(const DevParams *__restrict__ P, int accumulate) {
    const lg_cfg &c = P->cfg;
    const int N = c.num_envs, tid = threadIdx.x;
    const int n = *P->reset_count;
    __shared__ float s_red[256];
    if (n > 0) {
        if (tid < LG_NUM_TERMS)
            P->buf.extras_episode[tid] = term_scale(c, tid) != 0.0f ? (P->ep_accum[tid] / (float)n) / c.episode_length_s : 0.0f;

        if (c.send_timeouts) {                       // byte masks, 16 per lane and trip (hipMalloc'ed: 256-byte aligned)
            const int n16 = N >> 4;
            for (int i = tid; i < n16; i += 256)
                reinterpret_cast<uint4 *>(P->buf.extras_time_outs)[i] = reinterpret_cast<const uint4 *>(P->buf.time_out)[i];
            for (int i = (n16 << 4) + tid; i < N; i += 256) P->buf.extras_time_outs[i] = P->buf.time_out[i];
        }
        if (c.curriculum) {
            float s = 0.0f;
            for (int i = tid; i < N; i += 256) s += (float)P->buf.terrain_levels[i];
            s_red[tid] = s;
            __syncthreads();
            for (int w = 128; w > 0; w >>= 1) {
                if (tid < w) s_red[tid] += s_red[tid + w];
                __syncthreads();
            }
            if (tid == 0) P->buf.extras_terrain_level[0] = s_red[0] / (float)N;
        }
    }
    __syncthreads();
    if (n > 0 || c.traj.enabled) {
        const int n16 = N >> 4;
        for (int i = tid; i < n16; i += 256) reinterpret_cast<uint4 *>(P->reset_mark)[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = (n16 << 4) + tid; i < N; i += 256) P->reset_mark[i] = 0;
    }
    if (accumulate && tid < LG_NUM_TERMS) P->buf.extras_episode_acc[tid] += P->buf.extras_episode[tid];   // OnPolicyRunner.log: mean over the steps
    if (tid == 0) {
        if (accumulate) {
            P->buf.extras_episode_acc[LG_NUM_TERMS] += P->buf.extras_terrain_level[0];
            P->buf.extras_episode_acc[LG_NUM_TERMS + 1] += 1.0f;
        }
        P->buf.n_reset[0] = n; *P->reset_count = 0;
        const int nf = *P->fault_count;
        P->buf.n_fault[0] = nf; P->buf.fault_total[0] += nf; *P->fault_count = 0;
        const int nc = *P->clamp_count;
        P->buf.n_vel_clamp[0] = nc; P->buf.vel_clamp_total[0] += nc; *P->clamp_count = 0;
    }
    if (tid < LG_NUM_TERMS) P->ep_accum[tid] = 0.0f;
}


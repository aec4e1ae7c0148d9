// C-ABI (include/legged_hip.h, lg_ppo_*) for the PPO learner: parameter/storage allocation in
// HBM, layer-by-layer GEMM scheduling of the ActorCritic forward/backward, update orchestration.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ppo_device.h"
#include "ppo_mlp_args.h"

void lg_set_error(const std::string &s);

extern "C" {
void ppok_gemm_fwd(const GemmArgs *g, int nz, hipStream_t s);
void ppok_gemm_dx(const GemmArgs *g, int nz, hipStream_t s);
void ppok_sync_planes(const PpoDev *P, hipStream_t s);
int ppok_mlp_fwd(const MlpArgs *g, const PpoDev *P, int mask, hipStream_t s);
int ppok_mlp_supported(const MlpArgs *g);
int64_t ppok_mlp_frag_elems(int K, int N);
void ppok_mlp_frag_build(const MlpArgs *g, hipStream_t s);
void ppok_gemm_dw(const GemmArgs *g, int nz, int splits, hipStream_t s);
void ppok_debug_set_xcd_remap(int v);
void ppok_debug_set_dw_t(int v);
void ppok_act_sample(const PpoDev *P, const float *obs, const float *cobs, const float *mu, const float *val, int t,
                     int64_t cnt, int inject, hipStream_t s);
void ppok_process_step(const PpoDev *P, const float *rew, const uint8_t *dones, const uint8_t *tos, int t, hipStream_t s);
void ppok_gae(const PpoDev *P, const float *last_values, hipStream_t s);
void ppok_det_fold(const PpoDev *P, hipStream_t s);
void ppok_adv_normalize(const PpoDev *P, hipStream_t s);
void ppok_gather(const PpoDev *P, int mb, hipStream_t s);
void ppok_randperm(const PpoDev *P, int n, uint64_t update_idx, hipStream_t s);
void ppok_loss(const PpoDev *P, const float *mu, const float *v, float *dmu, float *dval, hipStream_t s);
int ppok_step(const PpoDev *P, int par, const PpoDev *G, int gather_mb, hipStream_t s);
size_t ppok_head_part_floats();
int ppok_head_fused(const PpoDev *P, int H3, const float *xa, const float *xc, float *dza, float *dzc, int64_t w_a, int64_t b_a,
                    int64_t w_c, int64_t b_c, int64_t b_prev_a, int64_t b_prev_c, hipStream_t s);
}

struct Net {
    int nl;                                  // linear layers (hidden + head)
    int dims[LG_PPO_MAX_LAYERS + 1];         // dims[0] = input, dims[nl] = output
    int64_t w_off[LG_PPO_MAX_LAYERS], b_off[LG_PPO_MAX_LAYERS];
    int64_t pl_off[LG_PPO_MAX_LAYERS];       // offset of this layer's weight matrix inside a bf16 plane (multiple of 8)
    int pl_ld0;                              // row length of W_0's plane image: dims[0] rounded up to 32 = one k-tile (pad columns zero)
    float *act[LG_PPO_MAX_LAYERS + 1];       // act[l], l >= 1: output of layer l-1 (workspace, Mmax rows)
    float *dz[LG_PPO_MAX_LAYERS + 1];        // gradient wrt act[l] pre-activation
};

struct lg_comm;
struct lg_ctx;
extern "C" {
// env side of the fused rollout epilogue (env_api.hip; same library, not in the header)
void lg_internal_defer_finalize(lg_ctx *c, int on);
hipStream_t lg_internal_stream(lg_ctx *c);
int lg_internal_finalize_pending(lg_ctx *c);
const DevParams *lg_internal_take_finalize(lg_ctx *c, int64_t *counter);
const DevParams *lg_internal_host_params(lg_ctx *c);
int lg_finalize(lg_ctx *c);
hipStream_t lg_comm_stream_(lg_comm *c);
hipEvent_t lg_comm_event_(lg_comm *c);
int lg_comm_group_allreduce_(lg_comm *c, float *const *bufs, const int64_t *counts, int n, hipStream_t s);
int lg_comm_allreduce_sum(lg_comm *c, float *buf, int64_t n, void *stream);
int lg_comm_broadcast(lg_comm *c, float *buf, int64_t n, int root, void *stream);
}

struct lg_ppo {
    lg_comm *comm;                           // when set: gradients are reduced per layer inside the backward pass
    hipEvent_t ev_bucket;                    // a layer's weight gradients are complete on the side stream
    int comm_rc;
    int comm_timing;                         // lg_ppo_comm_timing: event pairs around the wait for the last bucket
    std::vector<hipEvent_t> comm_ev;         // [2 k], [2 k + 1]: before / after the wait of recorded minibatch k
    size_t comm_ev_used;
    lg_ppo_cfg cfg;
    PpoDev dev;
    Net net[2];                              // 0 actor, 1 critic
    hipStream_t stream;
    hipStream_t side;                        // weight-gradient GEMMs run here, overlapping the input-gradient chain
    hipEvent_t ev_dz, ev_side;
    int overlap;
    long long *det_buf;                      // fixed-point shadow of the accumulators (lg_ppo_set_deterministic); dev.det64 = it when on
    int act_code;                            // kernels' activation code = cfg.activation + 1 (0 is 'none')
    int grads_dirty;
    // fused rollout epilogue (lg_ppo_attach_env): a process_env_step recorded for the next act launch
    lg_ctx *env;
    int pp_pending, pp_t, pp_use_tos;
    int params_dirty;                        // the fp32 parameters changed since the rollout images (fragment-order weights / bf16 planes)
                                             // were derived from them: optimiser step, broadcast, lg_ppo_params_changed()
    int fused_act;                           // rollout forward through k_mlp_fwd when the network shape allows (else per-layer GEMMs)
    MlpArgs mlp;                             // its arguments (fragment-order weight image allocated at create)
                         // gradients hold a backward pass that no optimiser step has consumed (and zeroed)
    int step, inject;
    int64_t act_count, update_count;
    int Mmax;
    std::vector<void *> allocs;
    int64_t perm_count;                      // updates begun: keys the device-side minibatch permutation
    // minibatch gathers are double-buffered: the optimiser step of minibatch mb gathers mb + 1 into the other set in the spare
    // workgroups of its norm-reduction launch (the gather depends on the rollout storage and the permutation only)
    struct MbSet { float *obs, *critic_obs, *actions, *mu, *scalars; } mbset[2];
    int mb_cur, mb_ready, mb_last;           // set the kernels read now; minibatch held by the other set (-1: none); last backward
    int gather_ahead;
    lg_ppo_buffers pub;
};

template <typename T>
static bool palloc(lg_ppo *p, T **out, size_t n) {
    void *q = nullptr;
    size_t bytes = (n ? n : 1) * sizeof(T);
    if (hipMalloc(&q, bytes) != hipSuccess || hipMemset(q, 0, bytes) != hipSuccess) return false;
    p->allocs.push_back(q);
    *out = (T *)q;
    return true;
}
#define PA(ptr, n) do { if (!palloc(p, &(ptr), (n))) { lg_set_error("hipMalloc failed in lg_ppo_create"); lg_ppo_destroy(p); return -100; } } while (0)

static int launch_ok() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { lg_set_error(std::string("ppo kernel launch: ") + hipGetErrorString(e)); return -100; }
    return 0;
}

// forward of the selected nets on M rows.  in[z] = input of net z.  mask bit z selects the net.
// in_pad: the inputs are the minibatch gathers, rows padded to a multiple of 32 floats (PpoDev::Op / OCp); otherwise rows of dims[0]
static void forward(lg_ppo *p, int M, const float *in0, const float *in1, int mask, int skip_head = 0, bool planes = false,
                    bool in_pad = false, hipStream_t on = nullptr, int l_begin = 0, int l_end = 1 << 30) {
    const hipStream_t fs = on ? on : p->stream;
    const float *in[2] = {in0, in1};
    const int in_ld[2] = {in_pad ? p->dev.Op : p->net[0].dims[0], in_pad ? p->dev.OCp : p->net[1].dims[0]};
    int sel[2], nz = 0;
    for (int z = 0; z < 2; ++z) if (mask & (1 << z)) sel[nz++] = z;
    const int nl = p->net[sel[0]].nl;
    for (int l = l_begin; l < nl - skip_head && l < l_end; ++l) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        for (int k = 0; k < nz; ++k) {
            Net &n = p->net[sel[k]];
            g.A[k] = l == 0 ? in[sel[k]] : n.act[l];
            g.B[k] = p->dev.params + n.w_off[l];
            g.bias[k] = p->dev.params + n.b_off[l];
            g.Bpl[k] = planes ? p->dev.wpl + n.pl_off[l] : nullptr;      // valid only inside an update (lg_ppo_begin_update syncs them)
            g.C[k] = n.act[l + 1];
            g.M[k] = M; g.N[k] = n.dims[l + 1]; g.K[k] = n.dims[l];
            g.lda[k] = l == 0 ? in_ld[sel[k]] : n.dims[l]; g.ldb[k] = n.dims[l]; g.ldc[k] = n.dims[l + 1];
            // W_0's plane image has rows as long as the padded gather rows: with both padded the plane path reduces over whole chunks
            if (l == 0 && in_pad && planes && n.pl_ld0 != n.dims[0]) { g.Kpl[k] = n.pl_ld0; g.ldbpl[k] = n.pl_ld0; }
            else if (l == 0 && n.pl_ld0 != n.dims[0]) g.Bpl[k] = nullptr;   // unpadded input against padded plane rows: fp32 operand
        }
        g.elu = l < nl - 1 ? p->act_code : 0;
        g.pl_stride = p->dev.pl_stride;
        ppok_gemm_fwd(&g, nz, fs);
    }
}

// Gradient bucket of layer l (both nets: W_l and b_l are adjacent in the flat buffer), all-reduced on the communicator's stream
// as soon as the layer's weight-gradient GEMM is done.  b_l was summed by the input-gradient GEMM of layer l+1 (or by the loss
// kernel for the head), which the weight-gradient GEMM of layer l already waited for.  The head's bucket also carries std and
// the [KL sum | pad] tail.  The collectives of one bucket are one RCCL group.
static int bucket_extents(lg_ppo *p, int l, float **bufs, int64_t *counts) {
    int n = 0;
    for (int z = 0; z < 2; ++z) {
        Net &net = p->net[z];
        bufs[n] = p->dev.grads + net.w_off[l];
        counts[n++] = (int64_t)net.dims[l + 1] * net.dims[l] + net.dims[l + 1];
    }
    if (l == p->net[0].nl - 1) {
        bufs[n] = p->dev.grads + p->dev.off_std; counts[n++] = p->cfg.num_actions;
        bufs[n] = p->dev.grads + p->dev.num_params; counts[n++] = 2;
    }
    return n;
}
static void reduce_layer_bucket(lg_ppo *p, int l, hipStream_t ready_on) {
    float *bufs[4];
    int64_t counts[4];
    const int n = bucket_extents(p, l, bufs, counts);
    hipStream_t cs = lg_comm_stream_(p->comm);
    (void)hipEventRecord(p->ev_bucket, ready_on);
    (void)hipStreamWaitEvent(cs, p->ev_bucket, 0);
    const int rc = lg_comm_group_allreduce_(p->comm, bufs, counts, n, cs);
    if (rc && !p->comm_rc) p->comm_rc = rc;
}

// backward of both nets on M rows given dz[nl] (head output gradients) already filled
static void backward(lg_ppo *p, int M, const float *in0, const float *in1, int skip_head = 0) {
    const float *in[2] = {in0, in1};
    auto det_args = [&](GemmArgs &g) {                // deterministic mode: gradient atomics go to the fixed-point shadow
        g.det_base = p->dev.grads; g.det64 = p->dev.det64; g.det_n = p->dev.num_params + 2;
    };
    const int nl = p->net[0].nl;
    for (int l = nl - 1 - skip_head; l >= 0; --l) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        det_args(g);
        long tiles = 0;
        for (int z = 0; z < 2; ++z) {                // dW_l += dz[l+1]^T . act[l]
            Net &n = p->net[z];
            g.A[z] = n.dz[l + 1]; g.lda[z] = n.dims[l + 1];
            g.B[z] = l == 0 ? in[z] : n.act[l]; g.ldb[z] = n.dims[l];
            g.C[z] = p->dev.grads + n.w_off[l]; g.ldc[z] = n.dims[l];
            g.M[z] = n.dims[l + 1]; g.N[z] = n.dims[l]; g.K[z] = M;
            if (l == 0) {                            // the minibatch gathers: padded rows, the pad columns computed but not stored
                const int ldp = z == 0 ? p->dev.Op : p->dev.OCp;
                g.ldb[z] = ldp;
                if (ldp != n.dims[0]) { g.N[z] = ldp; g.nstore[z] = n.dims[0]; }
            }
            const int tile = (g.M[z] > 64 && g.N[z] > 64) ? 128 : 64;
            long t = (long)((g.M[z] + tile - 1) / tile) * ((g.N[z] + tile - 1) / tile);
            tiles = t > tiles ? t : tiles;
        }
        // ~256 workgroups per net: enough to fill the chip, few enough that the split-M float atomics
        // (splits x output floats) stay well below the MFMA time
        // workgroups per net over (output tiles x reduction splits); swept 64..384 inside the update (side stream beside the
        // input-gradient chain): 0.637 / 0.585 / 0.597 / 0.574 / 0.560 ms per minibatch at 64 / 128 / 192 / 256 / 384
        static const int dw_target = getenv("LG_DW_WGS") ? atoi(getenv("LG_DW_WGS")) : 384;
        int splits = (int)((dw_target + tiles - 1) / tiles);
        int max_splits = M / 256 > 0 ? M / 256 : 1;
        if (splits > max_splits) splits = max_splits;
        if (splits >= 8) splits &= ~7;                 // whole groups of 8 slices: one per XCD (xcd_tile)
        if (splits < 1) splits = 1;
        // the weight gradient of layer l runs beside the input-gradient chain on the side stream -- except the last one
        // (l = 0), which has nothing left to overlap with: on the main stream it starts without the cross-stream event wait
        static const int dw0_main = getenv("LG_DW0_MAIN") ? atoi(getenv("LG_DW0_MAIN")) : 1;
        const bool on_side = p->overlap && (l > 0 || !dw0_main);
        hipStream_t dw_stream = on_side ? p->side : p->stream;
        if (on_side) {                               // dz[l+1] is complete on the main stream at this point
            (void)hipEventRecord(p->ev_dz, p->stream);
            (void)hipStreamWaitEvent(p->side, p->ev_dz, 0);
        }
        ppok_gemm_dw(&g, 2, splits, dw_stream);
        if (p->comm) reduce_layer_bucket(p, l, dw_stream);
        if (l > 0) {                                 // dz[l] = (dz[l+1] . W_l) * act'(act[l]); db_{l-1} = colsum(dz[l])
            memset(&g, 0, sizeof(g));
            det_args(g);
            for (int z = 0; z < 2; ++z) {
                Net &n = p->net[z];
                g.A[z] = n.dz[l + 1]; g.lda[z] = n.dims[l + 1];
                g.B[z] = p->dev.params + n.w_off[l]; g.ldb[z] = n.dims[l];
                g.Bpl[z] = p->dev.wpl + n.pl_off[l];         // current inside an update (begin_update / k_opt_adam keep them)
                g.C[z] = n.dz[l]; g.ldc[z] = n.dims[l];
                g.aux[z] = n.act[l]; g.ldaux[z] = n.dims[l];
                g.colsum[z] = p->dev.grads + n.b_off[l - 1];
                g.M[z] = M; g.N[z] = n.dims[l]; g.K[z] = n.dims[l + 1];
            }
            g.elu = p->act_code;                         // derivative of the hidden activation, through its output act[l]
            g.pl_stride = p->dev.pl_stride;
            ppok_gemm_dx(&g, 2, p->stream);
        }
    }
    if (p->overlap) {                                // join: the optimiser step (main stream) needs every dW
        (void)hipEventRecord(p->ev_side, p->side);
        (void)hipStreamWaitEvent(p->stream, p->ev_side, 0);
    }
    if (p->comm) {                                   // ... and every reduced bucket
        const bool rec = p->comm_timing && p->comm_ev_used + 2 <= 2 * 4096;
        if (rec) {
            while (p->comm_ev.size() < p->comm_ev_used + 2) { hipEvent_t e; (void)hipEventCreate(&e); p->comm_ev.push_back(e); }
            (void)hipEventRecord(p->comm_ev[p->comm_ev_used], p->stream);
        }
        (void)hipEventRecord(lg_comm_event_(p->comm), lg_comm_stream_(p->comm));
        (void)hipStreamWaitEvent(p->stream, lg_comm_event_(p->comm), 0);
        if (rec) { (void)hipEventRecord(p->comm_ev[p->comm_ev_used + 1], p->stream); p->comm_ev_used += 2; }
    }
}

// What a deferred process_env_step (and the env's epilogue before it) still owes, the ordinary way: env epilogue first, then the
// process kernel on the refreshed extras["time_outs"] -- the order of the unfused loop.
static void flush_rollout_epilogue(lg_ppo *p) {
    if (!p->env) return;
    (void)lg_finalize(p->env);
    if (p->pp_pending) {
        const lg_buffers &b = lg_internal_host_params(p->env)->buf;
        ppok_process_step(&p->dev, b.rew, b.reset, p->pp_use_tos ? b.extras_time_outs : nullptr, p->pp_t, p->stream);
        p->pp_pending = 0;
    }
}

extern "C" {

int lg_ppo_destroy(lg_ppo *p) {
    if (!p) return 0;
    if (p->env) { flush_rollout_epilogue(p); lg_internal_defer_finalize(p->env, 0); p->env = nullptr; }
    if (p->side) { (void)hipStreamSynchronize(p->side); (void)hipStreamDestroy(p->side); }
    if (p->ev_dz) (void)hipEventDestroy(p->ev_dz);
    if (p->ev_side) (void)hipEventDestroy(p->ev_side);
    if (p->ev_bucket) (void)hipEventDestroy(p->ev_bucket);
    for (hipEvent_t e : p->comm_ev) (void)hipEventDestroy(e);
    for (void *q : p->allocs) (void)hipFree(q);
    delete p;
    return 0;
}

int lg_ppo_create(const lg_ppo_cfg *cfg, lg_ppo **out) {
    if (!cfg || !out) { lg_set_error("null argument"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { lg_set_error("no HIP device: no CPU fallback"); return -2; }
    if (cfg->activation < 0 || cfg->activation > 5) { lg_set_error("activation must be one of 0 elu, 1 selu, 2 relu (also rsl_rl's crelu), 3 lrelu, 4 tanh, 5 sigmoid"); return -3; }
    if (cfg->num_hidden < 1 || cfg->num_hidden > LG_MAX_HIDDEN || cfg->num_actions > LG_PPO_MAX_A) {
        lg_set_error("unsupported network size"); return -4;
    }
    const int N = cfg->num_envs, T = cfg->num_steps, A = cfg->num_actions, O = cfg->num_obs;
    const int OC = cfg->num_critic_obs > 0 ? cfg->num_critic_obs : O;
    if ((long)N * T % cfg->num_mini_batches != 0 && (long)N * T / cfg->num_mini_batches == 0) { lg_set_error("bad minibatch count"); return -5; }
    lg_ppo *p = new lg_ppo();
    p->cfg = *cfg;
    p->stream = nullptr;
    p->act_code = cfg->activation + 1;
    p->fused_act = getenv("LG_FUSED_ACT") ? atoi(getenv("LG_FUSED_ACT")) : 1;   // one-launch rollout forward (ppo_mlp_fused.hip) when the shape allows
    p->overlap = getenv("LG_PPO_OVERLAP") ? atoi(getenv("LG_PPO_OVERLAP")) : 1;
    if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_dz, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_side, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&p->ev_bucket, hipEventDisableTiming) != hipSuccess) {
        lg_set_error("stream/event creation failed"); delete p; return -100;
    }
    p->step = 0; p->inject = 0; p->act_count = 0; p->update_count = 0; p->params_dirty = 1;
    p->comm = nullptr; p->comm_rc = 0; p->comm_timing = 0; p->comm_ev_used = 0;
    p->env = nullptr; p->pp_pending = 0;
    if (getenv("LG_XCD_REMAP")) ppok_debug_set_xcd_remap(atoi(getenv("LG_XCD_REMAP")));
    if (getenv("LG_DW_T")) ppok_debug_set_dw_t(atoi(getenv("LG_DW_T")));
    p->perm_count = 0;
    const int R = (int)((long)N * T / cfg->num_mini_batches);
    p->Mmax = R > N ? R : N;

    // parameter layout: std, actor (W,b)*, critic (W,b)*  -- the order of ActorCritic.parameters()
    int64_t off = 0;
    PpoDev &d = p->dev;
    memset(&d, 0, sizeof(d));
    d.off_std = (int)off; off += A;
    for (int z = 0; z < 2; ++z) {
        Net &n = p->net[z];
        n.nl = cfg->num_hidden + 1;
        n.dims[0] = z == 0 ? O : OC;
        for (int l = 0; l < cfg->num_hidden; ++l) n.dims[l + 1] = z == 0 ? cfg->actor_hidden[l] : cfg->critic_hidden[l];
        n.dims[n.nl] = z == 0 ? A : 1;
        for (int l = 0; l < n.nl; ++l) {
            n.w_off[l] = off; off += (int64_t)n.dims[l + 1] * n.dims[l];
            n.b_off[l] = off; off += n.dims[l + 1];
        }
    }
    d.num_params = off;
    int seg_ld[LG_PPO_MAX_SEG];
    {   // bf16 plane layout: one segment per weight matrix, 16-byte aligned
        int64_t po = 0;
        d.nseg = 0;
        for (int z = 0; z < 2; ++z)
            for (int l = 0; l < p->net[z].nl; ++l) {
                Net &n = p->net[z];
                n.pl_off[l] = po;
                d.seg_off[d.nseg] = n.w_off[l]; d.seg_pl[d.nseg] = po;
                d.seg_rows[d.nseg] = n.dims[l + 1]; d.seg_cols[d.nseg] = n.dims[l];
                // rows of the first layer's image padded to whole k-tiles of 32 (the observation width need not be one: 48, 235, 169, 65):
                // the LDS-DMA forward (ppo_gemm_glds.h) moves whole k-tiles, the plane path of k_gemm whole 16-byte chunks
                const int ld = l == 0 ? (n.dims[0] + 31) / 32 * 32 : n.dims[l];
                if (l == 0) n.pl_ld0 = ld;
                seg_ld[d.nseg] = ld;
                ++d.nseg;
                po += ((int64_t)n.dims[l + 1] * ld + 7) / 8 * 8;
            }
        d.pl_stride = po;
    }
    d.off_bias_actor_head = (int)p->net[0].b_off[p->net[0].nl - 1];
    d.off_bias_critic_head = (int)p->net[1].b_off[p->net[1].nl - 1];
    d.N = N; d.T = T; d.A = A; d.O = O; d.OC = OC; d.mb_rows = R;
    d.Op = (O + 31) / 32 * 32; d.OCp = (OC + 31) / 32 * 32;
    d.world = cfg->world_size > 0 ? cfg->world_size : 1;
    d.env_offset = 0;
    d.adaptive = cfg->adaptive_schedule; d.clipped_value = cfg->use_clipped_value_loss;
    d.seed = cfg->seed;
    d.gamma = cfg->gamma; d.lam = cfg->lam; d.clip = cfg->clip_param; d.value_coef = cfg->value_loss_coef;
    d.entropy_coef = cfg->entropy_coef; d.desired_kl = cfg->desired_kl; d.max_grad_norm = cfg->max_grad_norm;

    PA(d.params, off + 2); PA(d.grads, off + 2); PA(d.adam_m, off + 2); PA(d.adam_v, off + 2);
    {
        const int64_t pls = d.pl_stride;            // PA memsets through d.*: keep the stride across the allocation macros
        PA(d.wpl, (size_t)3 * pls + 8);
        PA(d.pl_dest, (size_t)off + 2);
        std::vector<int32_t> dest((size_t)off + 2, -1);
        for (int sg = 0; sg < d.nseg; ++sg)
            for (int64_t e = 0; e < (int64_t)d.seg_rows[sg] * d.seg_cols[sg]; ++e)
                dest[d.seg_off[sg] + e] = (int32_t)(d.seg_pl[sg] + (e / d.seg_cols[sg]) * seg_ld[sg] + e % d.seg_cols[sg]);
        (void)hipMemcpy(d.pl_dest, dest.data(), dest.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    const size_t TN = (size_t)T * N;
    PA(d.st_obs, TN * O);
    if (cfg->num_critic_obs > 0) PA(d.st_critic_obs, TN * OC); else d.st_critic_obs = d.st_obs;
    PA(d.st_actions, TN * A); PA(d.st_rewards, TN); PA(d.st_values, TN); PA(d.st_returns, TN); PA(d.st_adv, TN);
    PA(d.st_log_prob, TN); PA(d.st_mu, TN * A); PA(d.st_sigma, A); PA(d.st_dones, TN);
    PA(d.act_actions, (size_t)N * A); PA(d.act_values, N); PA(d.act_log_prob, N); PA(d.act_mu, (size_t)N * A);
    PA(d.stats, 8); PA(d.loss_acc, 4); PA(d.noise, (size_t)N * A); PA(d.perm, TN); PA(d.adv_partial, 4);
    for (int k = 0; k < 2; ++k) {
        lg_ppo::MbSet &m = p->mbset[k];
        PA(m.obs, (size_t)R * d.Op);                // pad columns: zeroed here, never written
        if (cfg->num_critic_obs > 0) PA(m.critic_obs, (size_t)R * d.OCp); else m.critic_obs = m.obs;
        PA(m.actions, (size_t)R * A); PA(m.mu, (size_t)R * A); PA(m.scalars, (size_t)R * 4);
    }
    p->mb_cur = 0; p->mb_ready = -1; p->mb_last = -1;
    p->gather_ahead = getenv("LG_GATHER_AHEAD") ? atoi(getenv("LG_GATHER_AHEAD")) : 1;
    d.mb_obs = p->mbset[0].obs; d.mb_critic_obs = p->mbset[0].critic_obs; d.mb_actions = p->mbset[0].actions;
    d.mb_mu = p->mbset[0].mu; d.mb_scalars = p->mbset[0].scalars;
    PA(d.head_part, ppok_head_part_floats());
    PA(d.cur_reward_sum, N); PA(d.cur_episode_len, N); PA(d.ep_stats, 4); PA(d.ep_ring, 200); PA(d.ep_ring_count, 1);
    for (int z = 0; z < 2; ++z) {
        Net &n = p->net[z];
        n.act[0] = nullptr; n.dz[0] = nullptr;
        for (int l = 1; l <= n.nl; ++l) { PA(n.act[l], (size_t)p->Mmax * n.dims[l]); PA(n.dz[l], (size_t)p->Mmax * n.dims[l]); }
    }
    if (p->fused_act) {                                         // arguments + weight image of the one-launch rollout forward
        MlpArgs &g = p->mlp;
        memset(&g, 0, sizeof(g));
        g.params = d.params; g.M = N; g.nl = p->net[0].nl; g.act = p->act_code;
        int64_t tot = 0;
        if (p->net[0].nl != p->net[1].nl) p->fused_act = 0;
        for (int z = 0; z < 2 && p->fused_act; ++z) {
            Net &n = p->net[z];
            for (int l = 0; l <= n.nl; ++l) g.dims[z][l] = n.dims[l];
            for (int l = 0; l < n.nl; ++l) {
                g.w_off[z][l] = n.w_off[l]; g.b_off[z][l] = n.b_off[l]; g.frag_off[z][l] = tot;
                tot += ppok_mlp_frag_elems(n.dims[l], n.dims[l + 1]);
            }
        }
        if (p->fused_act && ppok_mlp_supported(&g) != 0) p->fused_act = 0;
        if (p->fused_act) { uint16_t *wf = nullptr; PA(wf, (size_t)tot); g.wfrag = wf; }
    }
    {   // std = init_noise_std, lr = learning_rate
        std::vector<float> h(A, cfg->init_noise_std);
        (void)hipMemcpy(d.params + d.off_std, h.data(), sizeof(float) * A, hipMemcpyHostToDevice);
        float lr = cfg->learning_rate;
        (void)hipMemcpy(d.stats, &lr, sizeof(float), hipMemcpyHostToDevice);
    }
    lg_ppo_buffers &b = p->pub;
    memset(&b, 0, sizeof(b));
    b.params = d.params; b.grads = d.grads; b.adam_m = d.adam_m; b.adam_v = d.adam_v;
    b.obs = d.st_obs; b.critic_obs = d.st_critic_obs; b.actions = d.st_actions; b.rewards = d.st_rewards;
    b.values = d.st_values; b.returns = d.st_returns; b.advantages = d.st_adv; b.log_prob = d.st_log_prob;
    b.mu = d.st_mu; b.sigma = d.st_sigma; b.dones = d.st_dones;
    b.act_actions = d.act_actions; b.act_values = d.act_values; b.act_log_prob = d.act_log_prob; b.act_mu = d.act_mu;
    b.stats = d.stats; b.noise = d.noise; b.perm = d.perm; b.adv_partial = d.adv_partial;
    b.cur_reward_sum = d.cur_reward_sum; b.cur_episode_len = d.cur_episode_len; b.ep_stats = d.ep_stats;
    b.ep_ring = d.ep_ring; b.ep_ring_count = d.ep_ring_count;
    b.num_params = off; b.num_reduce = off + 2;
    if (hipDeviceSynchronize() != hipSuccess) { lg_set_error("device sync failed in lg_ppo_create"); lg_ppo_destroy(p); return -100; }
    *out = p;
    return 0;
}

int lg_ppo_get_buffers(lg_ppo *p, lg_ppo_buffers *out) { *out = p->pub; return 0; }
int lg_ppo_set_stream(lg_ppo *p, void *s) { p->stream = (hipStream_t)s; return 0; }
int lg_ppo_inject_noise(lg_ppo *p, int enable) { p->inject = enable; return 0; }
int lg_ppo_debug_set_overlap(lg_ppo *p, int v) { p->overlap = v; return 0; }
int lg_ppo_debug_set_fused_act(lg_ppo *p, int v) { p->fused_act = v && p->mlp.wfrag; return 0; }
int lg_ppo_debug_get_fused_act(lg_ppo *p) { return p->fused_act; }          // 1: lg_ppo_act runs the one-launch forward for this shape
int lg_ppo_debug_set_act_count(lg_ppo *p, long long v) { p->act_count = v; return 0; }   // replay the same Philox draws (tests)

// entries: std, then per net per layer (W, b).  offsets[i]; shapes[2i] = rows, shapes[2i+1] = cols (0 for vectors)
int lg_ppo_param_layout(lg_ppo *p, int64_t *offsets, int64_t *shapes, int max_entries) {
    int k = 0;
    auto put = [&](int64_t o, int64_t r, int64_t c) {
        if (k < max_entries) { offsets[k] = o; shapes[2 * k] = r; shapes[2 * k + 1] = c; }
        ++k;
    };
    put(p->dev.off_std, p->cfg.num_actions, 0);
    for (int z = 0; z < 2; ++z)
        for (int l = 0; l < p->net[z].nl; ++l) {
            put(p->net[z].w_off[l], p->net[z].dims[l + 1], p->net[z].dims[l]);
            put(p->net[z].b_off[l], p->net[z].dims[l + 1], 0);
        }
    return k;
}

int lg_ppo_attach_env(lg_ppo *p, lg_ctx *env) {
    flush_rollout_epilogue(p);
    if (p->env) lg_internal_defer_finalize(p->env, 0);
    p->env = nullptr;
    if (env) {
        if (lg_internal_host_params(env)->cfg.num_envs != p->cfg.num_envs) { lg_set_error("lg_ppo_attach_env: env and learner differ in num_envs"); return -12; }
        // the fused epilogue reads the env's rew / reset / time_out inside the learner's next act launch: only stream order makes that safe
        if (lg_internal_stream(env) != p->stream) { lg_set_error("lg_ppo_attach_env: env and learner must run on the same stream (lg_set_stream / lg_ppo_set_stream)"); return -12; }
        p->env = env;
        lg_internal_defer_finalize(env, 1);
    }
    return launch_ok();
}

int lg_ppo_act(lg_ppo *p, const float *obs, const float *critic_obs) {
    if (p->step >= p->cfg.num_steps) { lg_set_error("Rollout buffer overflow"); return -10; }
    const float *cobs = critic_obs ? critic_obs : obs;
    int fused = -1;
    // rollout images of the weights (fragment-order image / bf16 planes) are rebuilt when the parameters changed since they were built AND
    // at the first act of every rollout: a write through the zero-copy parameter views without lg_ppo_params_changed() is then stale for
    // at most the rollout in progress.  The flag is cleared once the launch that consumed it has gone out.
    const bool dirty = p->params_dirty != 0 || p->step == 0;
    if (p->fused_act) {
        // the image is rebuilt from the fp32 parameters whenever they changed since it was built (optimiser step, broadcast, or a
        // host write announced through lg_ppo_params_changed: checkpoint load, load_state_dict) -- also in the middle of a rollout
        MlpArgs &g = p->mlp;
        g.in[0] = obs; g.in[1] = cobs;
        g.out[0] = p->net[0].act[p->net[0].nl]; g.out[1] = p->net[1].act[p->net[1].nl];
        if (dirty) ppok_mlp_frag_build(&g, p->stream);
        static const int fuse_sample = getenv("LG_FUSED_SAMPLE") ? atoi(getenv("LG_FUSED_SAMPLE")) : 1;
        g.sample = fuse_sample; g.t = p->step; g.inject = p->inject; g.act_count = p->act_count;
        g.pp = 0;
        const int N = p->cfg.num_envs;
        const bool fits = (N + 255) / 256 + 1 <= (N + 31) / 32;  // epilogue workgroups within the launch's grid.x
        if (p->env && p->pp_pending && g.sample && fits) {        // the previous step's epilogue rides on this launch
            g.pp = lg_internal_finalize_pending(p->env) ? 2 : 1;
            g.pp_t = p->pp_t; g.pp_use_tos = p->pp_use_tos;
            g.pp_env = lg_internal_take_finalize(p->env, &g.pp_counter);
            p->pp_pending = 0;
        } else flush_rollout_epilogue(p);
        fused = ppok_mlp_fwd(&g, &p->dev, 3, p->stream);
        if (fused != 0 && g.pp) { lg_set_error("fused act launch refused with a rollout epilogue attached"); return -13; }
        if (fused == 0 && g.sample) {                                             // sampled and stored by the same launch
            p->act_count++;
            const int rc = launch_ok();
            if (rc == 0) p->params_dirty = 0;
            return rc;
        }
    }
    if (!p->fused_act) flush_rollout_epilogue(p);
    if (fused != 0) {
        // per-layer GEMMs on the optimiser's weight planes (no re-split of W per tile); same freshness rule as above
        static const int act_planes = getenv("LG_ACT_PLANES") ? atoi(getenv("LG_ACT_PLANES")) : 1;
        if (act_planes && dirty) ppok_sync_planes(&p->dev, p->stream);
        forward(p, p->cfg.num_envs, obs, cobs, 3, 0, act_planes != 0);
    }
    ppok_act_sample(&p->dev, obs, cobs, p->net[0].act[p->net[0].nl], p->net[1].act[p->net[1].nl], p->step, p->act_count,
                    p->inject, p->stream);
    p->act_count++;
    const int rc = launch_ok();
    if (rc == 0) p->params_dirty = 0;
    return rc;
}

int lg_ppo_process_env_step(lg_ppo *p, const float *rew, const uint8_t *dones, const uint8_t *time_outs) {
    if (p->step >= p->cfg.num_steps) { lg_set_error("Rollout buffer overflow"); return -10; }
    if (p->env) {
        // attached env: when the arguments are that env's own step outputs, record the call for the next act launch
        const lg_buffers &b = lg_internal_host_params(p->env)->buf;
        if (p->fused_act && !p->pp_pending && rew == b.rew && dones == b.reset && (!time_outs || time_outs == b.extras_time_outs)) {
            p->pp_pending = 1; p->pp_t = p->step; p->pp_use_tos = time_outs != nullptr;
            p->step++;
            return 0;
        }
        flush_rollout_epilogue(p);
    }
    ppok_process_step(&p->dev, rew, dones, time_outs, p->step, p->stream);
    p->step++;
    return launch_ok();
}

int lg_ppo_compute_returns(lg_ppo *p, const float *last_critic_obs) {
    flush_rollout_epilogue(p);
    forward(p, p->cfg.num_envs, nullptr, last_critic_obs, 2);
    ppok_gae(&p->dev, p->net[1].act[p->net[1].nl], p->stream);
    return launch_ok();
}

int lg_ppo_normalize_advantages(lg_ppo *p) {
    ppok_adv_normalize(&p->dev, p->stream);
    return launch_ok();
}

int lg_ppo_begin_update(lg_ppo *p) {
    flush_rollout_epilogue(p);
    // randperm(num_mini_batches * mini_batch_size), drawn once per update and reused by every epoch (Appendix B): on the device
    const size_t n = (size_t)p->dev.mb_rows * p->cfg.num_mini_batches;
    ppok_randperm(&p->dev, (int)n, (uint64_t)p->perm_count++, p->stream);
    p->mb_ready = -1;                           // a set gathered ahead belongs to the previous permutation
    (void)hipMemsetAsync(p->dev.stats + 2, 0, 2 * sizeof(float), p->stream);
    (void)hipMemsetAsync(p->dev.stats + 5, 0, sizeof(float), p->stream);
    (void)hipMemsetAsync(p->dev.loss_acc, 0, 4 * sizeof(float), p->stream);
    p->grads_dirty = 1;                      // first backward of this update clears the gradient buffer
    ppok_sync_planes(&p->dev, p->stream);    // weight planes follow whatever the fp32 parameters hold now
    return launch_ok();
}

int lg_ppo_minibatch_backward(lg_ppo *p, int epoch, int mb) {
    (void)epoch;
    PpoDev &d = p->dev;
    const int R = d.mb_rows;
    // k_opt_adam leaves the gradient buffer zeroed; only a backward pass that was never stepped needs a clear
    if (p->grads_dirty) (void)hipMemsetAsync(d.grads, 0, (size_t)(d.num_params + 2) * sizeof(float), p->stream);
    p->grads_dirty = 1;
    auto use_set = [&](PpoDev &dev, int k) {
        const lg_ppo::MbSet &m = p->mbset[k];
        dev.mb_obs = m.obs; dev.mb_critic_obs = m.critic_obs; dev.mb_actions = m.actions; dev.mb_mu = m.mu; dev.mb_scalars = m.scalars;
    };
    if (p->mb_ready == mb) {                                    // gathered ahead by the previous optimiser step
        p->mb_cur ^= 1;
        use_set(d, p->mb_cur);
    } else {
        ppok_gather(&d, mb, p->stream);
    }
    p->mb_ready = -1;
    p->mb_last = mb;
    Net &na = p->net[0], &nc = p->net[1];
    const int nl = na.nl, H3 = na.dims[nl - 1];
    // fused head (forward + loss + backward of the two thin head layers) when both nets end in the same
    // supported width; otherwise head GEMMs + k_loss
    // at 128 wide the fused head was 18 us slower per minibatch while the weight-gradient stream had slack; with that stream the
    // long pole (profiles/r02_timelines.txt) it is 9 us faster than head GEMM + k_loss + two head-gradient GEMMs (A/B on one box)
    static const int fuse128 = getenv("LG_HEAD_FUSE128") ? atoi(getenv("LG_HEAD_FUSE128")) : 1;
    const bool fuse = p->act_code == 1 && nl >= 2 && nc.dims[nl - 1] == H3 && (H3 == 64 || H3 == 32 || (H3 == 128 && fuse128));
    forward(p, R, d.mb_obs, d.mb_critic_obs, 3, fuse ? 1 : 0, true, true);
    if (fuse) {
        ppok_head_fused(&d, H3, na.act[nl - 1], nc.act[nl - 1], na.dz[nl - 1], nc.dz[nl - 1], na.w_off[nl - 1], na.b_off[nl - 1],
                        nc.w_off[nl - 1], nc.b_off[nl - 1], na.b_off[nl - 2], nc.b_off[nl - 2], p->stream);
    } else {
        ppok_loss(&d, na.act[na.nl], nc.act[nc.nl], na.dz[na.nl], nc.dz[nc.nl], p->stream);
    }
    if (fuse && p->comm) reduce_layer_bucket(p, nl - 1, p->stream);   // the fused head produced the head layer's gradients itself
    backward(p, R, d.mb_obs, d.mb_critic_obs, fuse ? 1 : 0);
    if (d.det64) ppok_det_fold(&d, p->stream);       // gradients, KL sum and loss sums of this minibatch, order-independent
    if (p->comm_rc) { const int rc = p->comm_rc; p->comm_rc = 0; return rc; }
    return launch_ok();
}

int lg_ppo_minibatch_step(lg_ppo *p) {
    int next = -1;
    PpoDev other = p->dev;
    if (p->gather_ahead && p->cfg.num_mini_batches > 1 && p->mb_last >= 0) {
        // the next minibatch of the generator's order (the permutation is fixed for the whole update)
        next = (p->mb_last + 1) % p->cfg.num_mini_batches;
        const lg_ppo::MbSet &m = p->mbset[p->mb_cur ^ 1];
        other.mb_obs = m.obs; other.mb_critic_obs = m.critic_obs; other.mb_actions = m.actions; other.mb_mu = m.mu; other.mb_scalars = m.scalars;
    }
    if (ppok_step(&p->dev, (int)(p->update_count & 1), &other, next, p->stream)) p->mb_ready = next;
    p->mb_last = -1;
    p->grads_dirty = 0;
    p->params_dirty = 1;
    p->update_count++;
    return launch_ok();
}

int lg_ppo_end_update(lg_ppo *p) { p->step = 0; return 0; }

int lg_ppo_set_comm(lg_ppo *p, lg_comm *c) {
    if (c && p->dev.det64) { lg_set_error("gradient buckets inside the backward pass cannot be combined with deterministic mode"); return -14; }
    p->comm = c;
    return 0;
}
int lg_ppo_comm_timing(lg_ppo *p, int enable) { p->comm_timing = enable ? 1 : 0; return 0; }
int lg_ppo_comm_wait_ms(lg_ppo *p, double *ms_total, int64_t *minibatches) {
    if (!ms_total || !minibatches) { lg_set_error("null argument"); return -1; }
    (void)hipStreamSynchronize(p->stream);
    double sum = 0.0;
    for (size_t k = 0; k + 1 < p->comm_ev_used; k += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->comm_ev[k], p->comm_ev[k + 1]) == hipSuccess) sum += ms;
    }
    *ms_total = sum; *minibatches = (int64_t)(p->comm_ev_used / 2);
    p->comm_ev_used = 0;
    return 0;
}
int lg_ppo_allreduce_adv_moments(lg_ppo *p, lg_comm *c) { return lg_comm_allreduce_sum(c, p->dev.adv_partial, 4, p->stream); }
int lg_ppo_broadcast_params(lg_ppo *p, lg_comm *c, int root) {
    p->params_dirty = 1;
    return lg_comm_broadcast(c, p->dev.params, p->dev.num_params + 2, root, p->stream);
}
int lg_ppo_params_changed(lg_ppo *p) { p->params_dirty = 1; return 0; }

int lg_ppo_set_deterministic(lg_ppo *p, int on) {
    if (on && p->comm) { lg_set_error("deterministic mode cannot be combined with gradient buckets reduced inside the backward pass (lg_ppo_set_comm)"); return -14; }
    if (on && !p->det_buf) {
        const size_t n = (size_t)p->dev.num_params + 10;
        void *q = nullptr;
        if (hipMalloc(&q, n * sizeof(long long)) != hipSuccess || hipMemset(q, 0, n * sizeof(long long)) != hipSuccess) {
            lg_set_error("hipMalloc failed in lg_ppo_set_deterministic"); return -100;
        }
        p->allocs.push_back(q);
        p->det_buf = (long long *)q;
    }
    if (!on && p->dev.det64) ppok_det_fold(&p->dev, p->stream);    // nothing stays behind in the shadow
    p->dev.det64 = on ? p->det_buf : nullptr;
    return launch_ok();
}

// The extents reduce_layer_bucket() hands to RCCL for layer l, as offsets into the flat gradient buffer (host-side, for tests:
// the buckets of all layers must tile [0, num_reduce) exactly once).  Returns the number of extents (<= 4).
int lg_ppo_debug_bucket_extents(lg_ppo *p, int l, int64_t *offsets, int64_t *counts) {
    if (l < 0 || l >= p->net[0].nl) return -1;
    float *bufs[4];
    const int n = bucket_extents(p, l, bufs, counts);
    for (int k = 0; k < n; ++k) offsets[k] = bufs[k] - p->dev.grads;
    return n;
}

int lg_ppo_act_inference(lg_ppo *p, const float *obs, float *actions_out, int64_t rows) {
    if (rows > p->Mmax) { lg_set_error("too many rows for act_inference"); return -11; }
    forward(p, (int)rows, obs, nullptr, 1);
    Net &na = p->net[0];
    if (hipMemcpyAsync(actions_out, na.act[na.nl], (size_t)rows * p->cfg.num_actions * sizeof(float), hipMemcpyDeviceToDevice,
                       p->stream) != hipSuccess) { lg_set_error("copy failed"); return -100; }
    return launch_ok();
}

}  // extern "C"

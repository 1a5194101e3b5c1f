// Parameter-arena and workspace layout shared by host launchers and kernels.
#pragma once
#include <stdint.h>
#include "../../include/ivae_amd.h"

#define IVAE_MAX_LAYERS 8
#define IVAE_MLP_LOGL 0
#define IVAE_MLP_F 1
#define IVAE_MLP_G 2
#define IVAE_MLP_LOGV 3

size_t pairwise_ws_floats(int n_rows, int n_cols, int d);     // pairwise.hip
static inline size_t ivae_pair_ws_floats(int n_rows, int n_cols, int d) { return pairwise_ws_floats(n_rows, n_cols, d); }

// Everything a kernel needs about one nn.Linear sits in ONE 64-byte line of the kernel-argument segment: a product's
// set-up costs one scalar-cache miss, not one per field (kernel arguments are read through the scalar cache, ~500 cycles a
// cold line).
struct alignas(64) LayerDesc {
    int w_off, b_off;       // float offset of the layer's weight (out, in) / bias inside a model's arena
    int din, dout;
    int64_t pre_off;        // workspace offset of the hidden activations act(pre), feature-major (hidden, Bp)
    int64_t der_off;        // ... and of the activation derivatives act'(pre), same layout
    int64_t dlt_off;        // dLoss/d(pre-activation output) of the layer, feature-major (dout, Bp)
    int pad[6];
};
struct MlpDesc {
    LayerDesc lay[IVAE_MAX_LAYERS];
    int present;
    int in_dim, out_dim;
};

struct Layout {
    int B, dd, dl, M, H, L, act, n_models;
    float slope;
    int n_params;
    MlpDesc mlp[4];
    int maxw;            // widest activation row (inputs included)
    int maxh;            // widest row among hidden / output tensors
    // workspace (float offsets per model)
    IvaeWsLayout ws;
    int64_t ws_partial;  // (n_wtiles, slab_ld) per-tile gradient slabs (16-byte aligned rows)
    int slab_ld;         // n_params rounded up to a multiple of 4
    int64_t ws_pair;     // scratch of the pairwise kernels for a (B, B, dl) problem
    // row tiling of k_fwd_enc / k_dec / k_bwd_enc (elbo_mfma.hip)
    int tr, n_tiles;     // rows per workgroup (16, 32 or 64), workgroups per model
    int trw, n_wtiles;   // rows per workgroup of k_wgrad (the contraction length of a weight gradient: 64 .. 256), its tiles
    int nw;              // waves per workgroup (4, 8 or 16): a streamed operand chunk holds 16 nw output columns
    int Bp;              // batch rounded up to 64: leading dimension of the feature-major pre-activations
    int uoff;            // column of u inside the [x | pad | u | 1] input tile (multiple of 4)
    int ldx, ldh, lds;   // LDS row strides (4 * odd floats): input tile, hidden tiles, latent-width tiles
    int ldd, ldr, nT;    // delta tiles (row-major stride), transposed delta (stride, rows)
    int smem_enc, smem_dec, smem_benc, smem_wg;   // dynamic LDS bytes of k_fwd_enc, k_dec, k_bwd_enc, k_wgrad
    int smem_chain;      // ... of the fused row-chain kernel k_chain; 0: its tiles do not fit, the three kernels above run
    int64_t ws_dz;       // (B, dl) dLoss/dz through the decoder (k_dec -> k_bwd_enc)
    // k_wgrad's jobs (4 x 4 tile blocks) per layer, MLP order f, g, logv, logl: one 64-byte line to decode blockIdx.z from
    alignas(64) unsigned short wg_jobs[4][IVAE_MAX_LAYERS];
    int wg_total;        // their sum = k_wgrad's gridDim.z
};

#define IVAE_LDS_LIMIT (160 * 1024)
#ifndef IVAE_CHAIN_MIN_TILES
#define IVAE_CHAIN_MIN_TILES 128
#endif
#ifndef IVAE_CHAIN_MAX_COLS
#define IVAE_CHAIN_MAX_COLS 64
#endif
#ifndef IVAE_CHAIN_LDS_LIMIT
#define IVAE_CHAIN_LDS_LIMIT (128 * 1024)
#endif    // larger row-chain tiles leave the three-kernel path (one chain per workgroup)

static inline int ivae_up16(int x) { return (x + 15) & ~15; }
static inline int ivae_ld4o(int x) { return (((x + 3) >> 2) | 1) << 2; }     // smallest 4*odd >= x

static inline bool ivae_lds_over(const Layout& L);
static inline void ivae_tile_geometry(Layout& L, int tr, int nw) {
    const int wide = L.maxh;
    L.tr = tr;
    L.nw = nw;
    L.ldx = ivae_ld4o(ivae_up16(L.uoff + L.M + 1));
    L.ldh = ivae_ld4o(ivae_up16(L.H + 1));
    L.lds = ivae_ld4o(ivae_up16(L.dl + 1));
    L.ldd = ivae_ld4o(ivae_up16(wide));
    L.nT = ivae_up16(wide);
    // +64 floats: k-steps past a tile's last column read (finite) neighbours; the last tile needs room for that
    const int ldt = L.ldh > L.ldd ? L.ldh : L.ldd;
    L.smem_enc = 4 * (tr * L.ldx + 2 * tr * L.ldh + 64);
    L.smem_dec = 4 * (2 * tr * ldt + 2 * tr * L.lds + 64);
    L.smem_benc = 4 * (2 * tr * L.ldd + 64 + 256);             // (+ the loss reduction's 256 partials)
    L.smem_wg = 0;
    {   // k_chain: input tile, two tiles per concurrent chain, seven row tensors (g, v, l, z, dz, eps; f)
        const int ngrp = L.M > 0 ? 3 : 2;
        const int64_t fl = (int64_t)tr * L.ldx + (int64_t)ngrp * 2 * tr * ldt + 6 * (int64_t)tr * L.lds + (int64_t)tr * (L.dd | 1) + 64;
        L.smem_chain = 4 * fl <= IVAE_CHAIN_LDS_LIMIT ? (int)(4 * ((fl + 3) & ~(int64_t)3)) : 0;
        // a launch that leaves most CUs empty runs faster as one chain per workgroup (three times the CUs, a third of
        // the instruction stream on each): the fused kernel is for launches that fill the chip anyway
        if (IVAE_CHAIN_MIN_TILES && (int64_t)((L.B + tr - 1) / tr) * L.n_models < IVAE_CHAIN_MIN_TILES) L.smem_chain = 0;
        if (tr > 32) L.smem_chain = 0;            // (its 12-wave workgroups have 170 registers per lane: two row tiles at most)
        // its chains are groups of 4 waves = 64 output columns per pass: wider layers run faster as 8-wave workgroups of
        // their own (measured, H = 100: 64 packed models 104 against 111 us per step, batch 16384 170 against 189 us)
        if (IVAE_CHAIN_MAX_COLS && L.maxh > IVAE_CHAIN_MAX_COLS) L.smem_chain = 0;      // (maxh: widest OUTPUT of any layer)
    }
}

static inline bool ivae_lds_over(const Layout& L) {
    return L.smem_enc > IVAE_LDS_LIMIT || L.smem_dec > IVAE_LDS_LIMIT || L.smem_benc > IVAE_LDS_LIMIT || L.smem_wg > IVAE_LDS_LIMIT;
}

static inline int ivae_check_dims(const IvaeDims* d) {
    if (!d) return IVAE_E_NULLPTR;
    if (d->batch <= 0 || d->data_dim <= 0 || d->latent_dim <= 0 || d->aux_dim < 0 || d->n_models <= 0)
        return IVAE_E_BADDIMS;
    if (d->n_layers < 1 || d->n_layers > IVAE_MAX_LAYERS) return IVAE_E_BADDIMS;
    if (d->n_layers > 1 && d->hidden_dim <= 0) return IVAE_E_BADDIMS;
    if (d->activation < IVAE_ACT_NONE || d->activation > IVAE_ACT_SIGMOID) return IVAE_E_BADACT;
    if (d->latent_dim > 64 || d->data_dim > 1024 || d->aux_dim > 4096 || d->hidden_dim > 1024)
        return IVAE_E_TOOLARGE;
    {   // the kernels form byte offsets into one (rows, features) / (features, rows) tensor in 32-bit integers
        int w = d->hidden_dim > d->data_dim + d->aux_dim ? d->hidden_dim : d->data_dim + d->aux_dim;
        if (d->latent_dim > w) w = d->latent_dim;
        if (((int64_t)d->batch + 64) * (w + 4) * 4 >= ((int64_t)1 << 31)) return IVAE_E_TOOLARGE;
    }
    const int tr = d->tile_rows, nw = d->waves, wr = d->wgrad_rows;      // tuning fields: 0 = the library picks
    if (tr != 0 && tr != 16 && tr != 32 && tr != 64) return IVAE_E_BADDIMS;
    if (nw != 0 && nw != 4 && nw != 8 && nw != 16) return IVAE_E_BADDIMS;
    if (wr != 0 && wr != 16 && wr != 32 && wr != 64 && wr != 128 && wr != 256) return IVAE_E_BADDIMS;
    return IVAE_OK;
}

// with_pair: the workspace also holds the pairwise kernels' scratch for a (B, B, dl) problem (ivae_elbo_fwd_bwd with
// coef.pairwise); without it every part of the workspace is linear in the batch.
static inline int ivae_make_layout(const IvaeDims* d, bool with_pair, Layout* Lo) {
    int rc = ivae_check_dims(d);
    if (rc) return rc;
    Layout& L = *Lo;
    L.B = d->batch; L.dd = d->data_dim; L.dl = d->latent_dim; L.M = d->aux_dim;
    L.H = d->n_layers > 1 ? d->hidden_dim : 0; L.L = d->n_layers; L.act = d->activation;
    L.slope = d->slope; L.n_models = d->n_models;
    L.Bp = (((L.B + 63) >> 6) | 1) << 6;      // batch rounded up to an ODD multiple of 64: feature-major planes then start
                                              // 256 B apart modulo 512 B instead of all on the same memory channels
    L.uoff = L.M > 0 ? (L.dd + 3) & ~3 : L.dd;
    int off = 0;
    int64_t wso = 0;
    for (int m = 0; m < 4; ++m) {
        MlpDesc& D = L.mlp[m];
        D.present = (m != IVAE_MLP_LOGL) || (L.M > 0);
        if (m == IVAE_MLP_LOGL) { D.in_dim = L.M; D.out_dim = L.dl; }
        else if (m == IVAE_MLP_F) { D.in_dim = L.dl; D.out_dim = L.dd; }
        else { D.in_dim = L.dd + L.M; D.out_dim = L.dl; }
        for (int li = 0; li < IVAE_MAX_LAYERS; ++li) D.lay[li] = LayerDesc{};
        if (!D.present) continue;
        for (int li = 0; li < L.L; ++li) {
            D.lay[li].din = li == 0 ? D.in_dim : L.H;
            D.lay[li].dout = li == L.L - 1 ? D.out_dim : L.H;
            D.lay[li].w_off = off; off += D.lay[li].din * D.lay[li].dout;
            D.lay[li].b_off = off; off += D.lay[li].dout;
            if (li < L.L - 1) { D.lay[li].pre_off = wso; wso += (int64_t)L.Bp * L.H; D.lay[li].der_off = wso; wso += (int64_t)L.Bp * L.H; }
            D.lay[li].dlt_off = wso; wso += (int64_t)L.Bp * D.lay[li].dout;
        }
    }
    L.n_params = off;
    L.maxh = L.H; if (L.dd > L.maxh) L.maxh = L.dd; if (L.dl > L.maxh) L.maxh = L.dl;
    L.maxw = L.maxh; if (L.dd + L.M > L.maxw) L.maxw = L.dd + L.M;
    auto take = [&](int64_t n) { int64_t o = wso; wso += n; return o; };
    int64_t Bl = (int64_t)L.B * L.dl;
    L.ws.g = take(Bl); L.ws.v = take(Bl); L.ws.l = take(Bl); L.ws.z = take(Bl);
    L.ws.f = take((int64_t)L.B * L.dd);
    L.ws.lpx = take(L.B); L.ws.lqc = take(L.B); L.ws.lp = take(L.B); L.ws.lq = take(L.B); L.ws.lqi = take(L.B);
    L.ws.rz = take(Bl); L.ws.cg = take(Bl); L.ws.cv = take(Bl);
    L.ws.lse_s = take(L.B); L.ws.lse_k = take(Bl);
    // rows per workgroup: 32 when that still gives every CU two row tiles, else 16 (measured: 64-row tiles lose to 32-row
    // ones even on chip-filling launches -- their LDS footprint leaves one workgroup per CU and nothing to hide its
    // chunk round trips behind; the knob below still selects them), subject to the LDS fit further down
    int tr = 32;
    while (tr > 16 && (int64_t)((L.B + tr - 1) / tr) * L.n_models < 512) tr >>= 1;
    if (d->tile_rows) tr = d->tile_rows;                           // caller's choice (validated by ivae_check_dims)
    // waves per workgroup: 4 when no product streams more than 64 output columns; wider layers take 8 (128 columns per
    // pass; measured faster than 4 and than 16 on chip-filling launches too: H = 200, batch 16384: 0.40 against 0.43 /
    // 0.43 ms), and 16 when the launch leaves CUs empty and a layer is wider than 128, so that a short dependent chain of
    // layers passes fewer barriers
    const int jmax = L.maxh;                                       // widest output (streamed columns) over all products
    int nw = 4;
    if (jmax > 64) nw = ((int64_t)((L.B + tr - 1) / tr) * L.n_models < 256 && jmax > 128) ? 16 : 8;
    if (d->waves) nw = d->waves;
    ivae_tile_geometry(L, tr, nw);
    while (ivae_lds_over(L) && (nw > 4 || tr > 16)) {
        if (nw > 4) nw >>= 1; else tr >>= 1;
        ivae_tile_geometry(L, tr, nw);
    }
    if (ivae_lds_over(L)) return IVAE_E_TOOLARGE;
    L.n_tiles = (L.B + tr - 1) / tr;
    // k_wgrad's own row tiling: a weight gradient contracts over the batch rows, so its one-wave jobs take as many rows
    // (16 .. 256) as still gives every SIMD of the chip a job: longer contractions per job and fewer slabs for k_reduce
    {
        int64_t jobs = 0;                         // (MLP, layer, 4 x 4 tile block) jobs per row tile and model: k_wgrad's blockIdx.z
        const int order[4] = {IVAE_MLP_F, IVAE_MLP_G, IVAE_MLP_LOGV, IVAE_MLP_LOGL};
        for (int mi = 0; mi < 4; ++mi)
            for (int li = 0; li < IVAE_MAX_LAYERS; ++li) {
                const MlpDesc& D = L.mlp[order[mi]];
                int nj = 0;
                if (D.present && li < L.L) nj = (((D.lay[li].din + 16) / 16 + 3) / 4) * (((D.lay[li].dout + 15) / 16 + 3) / 4);
                L.wg_jobs[mi][li] = (unsigned short)nj;        // (din, dout <= 4096: at most 65 * 64 blocks... checked below)
                jobs += nj;
            }
        if (jobs > 60000) return IVAE_E_TOOLARGE;
        L.wg_total = (int)jobs;
        jobs *= L.n_models;
        int trw = 256;
        while (trw > 16 && (trw / 2 >= L.B + 15 || (int64_t)((L.B + trw - 1) / trw) * jobs < 1024)) trw >>= 1;
        if (d->wgrad_rows) trw = d->wgrad_rows;
        L.trw = trw;
        L.ldr = trw + 4;
        L.smem_wg = 0;
        L.n_wtiles = (L.B + trw - 1) / trw;
    }
    L.slab_ld = (L.n_params + 3) & ~3;
    wso = (wso + 3) & ~(int64_t)3;
    L.ws_partial = take((int64_t)L.n_wtiles * L.slab_ld);
    L.ws_dz = take(Bl);
    L.ws_pair = take(with_pair ? (int64_t)ivae_pair_ws_floats(L.B, L.B, L.dl) : 0);
    wso = (wso + 63) & ~(int64_t)63;
    L.ws.per_model = wso;
    // callers form per_model * n_models * sizeof(float) in size_t: refuse what no memory holds before that can wrap
    if (wso > ((int64_t)1 << 44) / L.n_models) return IVAE_E_TOOLARGE;
    return IVAE_OK;
}

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

struct MlpDesc {
    int present;
    int in_dim, out_dim;
    int w_off[IVAE_MAX_LAYERS];    // float offset of layer weight (out,in) inside a model's arena
    int b_off[IVAE_MAX_LAYERS];
    int din[IVAE_MAX_LAYERS];
    int dout[IVAE_MAX_LAYERS];
    int64_t pre_off[IVAE_MAX_LAYERS];  // workspace offset of the hidden activations act(pre), feature-major (hidden, Bp)
    int64_t der_off[IVAE_MAX_LAYERS];  // ... and of the activation derivatives act'(pre), same layout
    int64_t dlt_off[IVAE_MAX_LAYERS];  // dLoss/d(pre-activation output) of every layer, feature-major (dout, Bp)
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
    int64_t ws_dz;       // (B, dl) dLoss/dz through the decoder (k_dec -> k_bwd_enc)
};

#define IVAE_CHUNK_FLOATS(nw) (16 * (nw) * 68)      // one streamed operand chunk (elbo_mfma.hip)
#define IVAE_LDS_LIMIT (160 * 1024)

static inline int ivae_up16(int x) { return (x + 15) & ~15; }
static inline int ivae_ld4o(int x) { return (((x + 3) >> 2) | 1) << 2; }     // smallest 4*odd >= x

static inline bool ivae_lds_over(const Layout& L);
static inline void ivae_tile_geometry(Layout& L, int tr, int nw) {
    const int wide = L.maxh;
    L.tr = tr;
    L.nw = nw;
    const int chunk = IVAE_CHUNK_FLOATS(nw);
    L.ldx = ivae_ld4o(ivae_up16(L.uoff + L.M + 1));
    L.ldh = ivae_ld4o(ivae_up16(L.H + 1));
    L.lds = ivae_ld4o(ivae_up16(L.dl + 1));
    L.ldd = ivae_ld4o(ivae_up16(wide));
    L.nT = ivae_up16(wide);
    // +64 floats: k-steps past a tile's last column read (finite) neighbours; the last tile needs room for that
    const int ldt = L.ldh > L.ldd ? L.ldh : L.ldd;
    L.smem_enc = 4 * (2 * chunk + tr * L.ldx + 2 * tr * L.ldh + 64);
    L.smem_dec = 4 * (2 * chunk + 2 * tr * ldt + 2 * tr * L.lds + 64);
    L.smem_benc = 4 * (2 * chunk + 2 * tr * L.ldd + 64);
    L.smem_wg = 4 * (2 * chunk + L.nT * (16 + 4) + 64);      // with k_wgrad's smallest row tile; final value: ivae_make_layout
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
    L.Bp = (L.B + 63) & ~63;
    L.uoff = L.M > 0 ? (L.dd + 3) & ~3 : L.dd;
    int off = 0;
    int64_t wso = 0;
    for (int m = 0; m < 4; ++m) {
        MlpDesc& D = L.mlp[m];
        D.present = (m != IVAE_MLP_LOGL) || (L.M > 0);
        if (m == IVAE_MLP_LOGL) { D.in_dim = L.M; D.out_dim = L.dl; }
        else if (m == IVAE_MLP_F) { D.in_dim = L.dl; D.out_dim = L.dd; }
        else { D.in_dim = L.dd + L.M; D.out_dim = L.dl; }
        for (int li = 0; li < IVAE_MAX_LAYERS; ++li) { D.w_off[li] = D.b_off[li] = D.din[li] = D.dout[li] = 0; D.pre_off[li] = D.der_off[li] = D.dlt_off[li] = 0; }
        if (!D.present) continue;
        for (int li = 0; li < L.L; ++li) {
            D.din[li] = li == 0 ? D.in_dim : L.H;
            D.dout[li] = li == L.L - 1 ? D.out_dim : L.H;
            D.w_off[li] = off; off += D.din[li] * D.dout[li];
            D.b_off[li] = off; off += D.dout[li];
            if (li < L.L - 1) { D.pre_off[li] = wso; wso += (int64_t)L.Bp * L.H; D.der_off[li] = wso; wso += (int64_t)L.Bp * L.H; }
            D.dlt_off[li] = wso; wso += (int64_t)L.Bp * D.dout[li];
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
    // waves per workgroup: 4 when the launch fills the chip by itself (several workgroups share a CU and hide each
    // other's latencies) or no product streams more than 64 output columns; otherwise a wider workgroup takes
    // 128 / 256 columns per chunk, so that a short dependent chain of layers needs fewer chunk round trips
    int jmax = L.H + 1;                                            // widest streamed extent over all products
    if (L.dd + L.M + 1 > jmax) jmax = L.dd + L.M + 1;
    if (L.dl + 1 > jmax) jmax = L.dl + 1;
    int nw = 4;
    if ((int64_t)((L.B + tr - 1) / tr) * L.n_models < 256 && jmax > 64) nw = jmax > 128 ? 16 : 8;
    if (d->waves) nw = d->waves;
    ivae_tile_geometry(L, tr, nw);
    while (ivae_lds_over(L) && (nw > 4 || tr > 16)) {
        if (nw > 4) nw >>= 1; else tr >>= 1;
        ivae_tile_geometry(L, tr, nw);
    }
    if (ivae_lds_over(L)) return IVAE_E_TOOLARGE;
    L.n_tiles = (L.B + tr - 1) / tr;
    // k_wgrad's own row tiling: a weight gradient contracts over the batch rows, so its workgroups take as many rows
    // (16 .. 256) as still leaves the chip two workgroups per CU (and fits 64 KB of LDS): fewer, longer contractions per
    // chunk round trip and fewer slabs for k_reduce; launches too small to fill the chip keep 16-row tiles (measured)
    {
        const int64_t prods = (int64_t)(L.M > 0 ? 4 : 3) * L.L * L.n_models;
        int trw = L.Bp < 256 ? L.Bp : 256;       // (Bp is a multiple of 64)
        if (trw == 192) trw = 128;
        auto smem = [&](int t) { return 4 * (2 * IVAE_CHUNK_FLOATS(L.nw) + L.nT * (t + 4) + 64); };
        while (trw > 16 && ((int64_t)((L.B + trw - 1) / trw) * prods < 512 || smem(trw) > 64 * 1024)) trw >>= 1;
        if (d->wgrad_rows) trw = d->wgrad_rows;
        while (trw > 16 && smem(trw) > IVAE_LDS_LIMIT) trw >>= 1;
        if (smem(trw) > IVAE_LDS_LIMIT) return IVAE_E_TOOLARGE;
        L.trw = trw;
        L.ldr = trw + 4;
        L.smem_wg = smem(trw);
        L.n_wtiles = (L.B + trw - 1) / trw;
    }
    L.slab_ld = (L.n_params + 3) & ~3;
    wso = (wso + 3) & ~(int64_t)3;
    L.ws_partial = take((int64_t)L.n_wtiles * L.slab_ld);
    L.ws_dz = take(Bl);
    L.ws_pair = take(with_pair ? (int64_t)ivae_pair_ws_floats(L.B, L.B, L.dl) : 0);
    wso = (wso + 63) & ~(int64_t)63;
    L.ws.per_model = wso;
    return IVAE_OK;
}

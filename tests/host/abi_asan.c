/* Host side of libivae_amd under AddressSanitizer + UBSan (SURVEY.md §5 "race detection / sanitizers": the kernels are
 * validated by parity, the host code -- argument checks, layout arithmetic, launch preparation -- by an instrumented
 * build).  `make asan` compiles the library's sources with the sanitizers on the HOST side only and links this driver.
 * It needs no GPU: the size / layout entry points run over a grid and a random walk of shapes (valid, degenerate, huge),
 * every compute entry point is called with NULL arguments, and -- only when NO device is visible -- with fake device
 * pointers, which must come back as IVAE_E_HIP after the whole host-side preparation ran (nothing on the host may
 * dereference a device pointer).  Test infrastructure; prints "abi_asan: OK" and exits 0 when every check held. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ivae_amd.h"

int hipGetDeviceCount(int* n);        /* from libamdhip64 (linked anyway) */

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { ++fails; fprintf(stderr, "abi_asan: " __VA_ARGS__); fputc('\n', stderr); } } while (0)

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 11); }
static int pick(const int* v, int n) { return v[rnd() % n]; }

static int64_t closed_form_params(const IvaeDims* d) {
    int64_t tot = 0;
    for (int m = 0; m < 4; ++m) {          /* logl, f, g, logv (models/nets.py:422-430) */
        if (m == 0 && d->aux_dim == 0) continue;
        int in = m == 0 ? d->aux_dim : m == 1 ? d->latent_dim : d->data_dim + d->aux_dim;
        int out = m == 1 ? d->data_dim : d->latent_dim;
        for (int l = 0; l < d->n_layers; ++l) {
            int di = l == 0 ? in : d->hidden_dim, dout = l == d->n_layers - 1 ? out : d->hidden_dim;
            tot += (int64_t)di * dout + dout;
        }
    }
    return tot;
}

static void check_shape(const IvaeDims* d) {
    int64_t np = -1;
    size_t b0 = 0, b1 = 0;
    IvaeWsLayout w0, w1;
    memset(&w0, 0, sizeof w0);
    memset(&w1, 0, sizeof w1);
    const int rc = ivae_param_count(d, &np);
    CHECK(rc <= 0 && rc >= IVAE_E_TOOLARGE, "ivae_param_count -> %d", rc);
    const int r0 = ivae_workspace_bytes(d, 0, &b0), r1 = ivae_workspace_bytes(d, 1, &b1);
    const int l0 = ivae_workspace_layout(d, 0, &w0), l1 = ivae_workspace_layout(d, 1, &w1);
    CHECK(r0 <= 0 && r0 >= IVAE_E_TOOLARGE && r1 <= 0 && r1 >= IVAE_E_TOOLARGE, "ivae_workspace_bytes -> %d / %d", r0, r1);
    CHECK((r0 == 0) == (l0 == 0) && (r1 == 0) == (l1 == 0), "size and layout queries disagree: %d %d %d %d", r0, l0, r1, l1);
    (void)ivae_train_steps_supported(d);
    if (rc == 0) CHECK(np == closed_form_params(d), "param count %lld, closed form %lld", (long long)np, (long long)closed_form_params(d));
    if (r0 == 0 && l0 == 0) {
        const int64_t B = d->batch, dl = d->latent_dim, dd = d->data_dim;
        CHECK((int64_t)(b0 / 4) >= w0.per_model * d->n_models, "workspace %zu B < %lld floats x %d models", b0, (long long)w0.per_model, d->n_models);
        const int64_t off[] = {w0.g, w0.v, w0.l, w0.z, w0.f, w0.lpx, w0.lqc, w0.lp, w0.lq, w0.lqi, w0.rz, w0.cg, w0.cv, w0.lse_s, w0.lse_k};
        const int64_t len[] = {B * dl, B * dl, B * dl, B * dl, B * dd, B, B, B, B, B, B * dl, B * dl, B * dl, B, B * dl};
        for (int i = 0; i < 15; ++i) {
            CHECK(off[i] >= 0 && off[i] + len[i] <= w0.per_model, "row tensor %d outside the model's workspace", i);
            if (i) CHECK(off[i] >= off[i - 1] + len[i - 1], "row tensors %d and %d overlap", i - 1, i);
        }
        if (r1 == 0 && l1 == 0) CHECK(b1 >= b0 && w1.g == w0.g && w1.lse_k == w0.lse_k,
                                     "pairwise workspace is not a superset: B %d dd %d dl %d M %d H %d L %d K %d knobs %d %d %d: %zu / %zu bytes, g %lld / %lld",
                                     d->batch, d->data_dim, d->latent_dim, d->aux_dim, d->hidden_dim, d->n_layers, d->n_models, d->tile_rows,
                                     d->waves, d->wgrad_rows, b0, b1, (long long)w0.g, (long long)w1.g);
    }
}

int main(void) {
    /* ---- grid of the shapes the reference's configs use, with every tuning knob */
    const int tr[] = {0, 16, 32, 64}, nw[] = {0, 4, 8, 16}, wr[] = {0, 16, 64, 256};
    const int shapes[][6] = {{64, 10, 10, 40, 50, 3}, {64, 5, 5, 40, 100, 3}, {1024, 5, 5, 0, 100, 3}, {4096, 20, 20, 200, 50, 3},
                             {256, 10, 10, 40, 200, 3}, {5, 3, 2, 4, 8, 1}, {200, 4, 3, 0, 7, 4}, {64, 40, 40, 40, 50, 3}};
    for (unsigned s = 0; s < sizeof shapes / sizeof shapes[0]; ++s)
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) for (int k = 1; k <= 256; k *= 16) {
            IvaeDims d = {shapes[s][0], shapes[s][1], shapes[s][2], shapes[s][3], shapes[s][4], shapes[s][5], IVAE_ACT_LRELU, .1f, k, tr[a], nw[b], wr[c]};
            check_shape(&d);
        }
    /* ---- random walk: valid, degenerate and huge values in every field */
    const int dims[] = {-7, 0, 1, 2, 3, 5, 15, 16, 17, 31, 50, 63, 64, 65, 100, 200, 1000, 1024, 1025, 4096, 4097, 65536, 1 << 20, 0x7fffffff};
    const int small[] = {-1, 0, 1, 2, 3, 4, 8, 9, 100}, acts[] = {-1, 0, 1, 2, 3, 4, 99};
    const int knobs[] = {0, 16, 32, 64, 4, 8, 128, 256, 3, -16};
    for (int it = 0; it < 200000; ++it) {
        IvaeDims d = {pick(dims, 24), pick(dims, 24), pick(dims, 24), pick(dims, 24), pick(dims, 24), pick(small, 9), pick(acts, 7),
                      (float)(rnd() % 5) * .1f - .1f, pick(dims, 24), pick(knobs, 10), pick(knobs, 10), pick(knobs, 10)};
        if (it % 3 == 0) { d.batch = 1 + rnd() % 5000; d.data_dim = 1 + rnd() % 64; d.latent_dim = 1 + rnd() % 64; d.aux_dim = rnd() % 300;
                           d.hidden_dim = 1 + rnd() % 300; d.n_layers = 1 + rnd() % 8; d.activation = rnd() % 4; d.n_models = 1 + rnd() % 300;
                           if (it % 6 == 0) d.tile_rows = d.waves = d.wgrad_rows = 0; }
        check_shape(&d);
        size_t o = 0;
        int r = ivae_pairwise_workspace_bytes(pick(dims, 24), pick(dims, 24), pick(dims, 24), &o);
        CHECK(r <= 0 && r >= IVAE_E_TOOLARGE, "ivae_pairwise_workspace_bytes -> %d", r);
        r = ivae_corr_workspace_bytes((int64_t)pick(dims, 24) * (it & 1 ? 1 : 4099), pick(dims, 24), &o);
        CHECK(r <= 0 && r >= IVAE_E_TOOLARGE, "ivae_corr_workspace_bytes -> %d", r);
    }
    /* ---- NULL arguments: an error code, never a crash */
    IvaeDims ok = {64, 10, 10, 40, 50, 3, IVAE_ACT_LRELU, .1f, 2, 0, 0, 0};
    IvaeCoef co = {1.f, 1.f, 1.f, 1.f, 40000.f, .01f, 0, 64};
    CHECK(ivae_param_count(NULL, NULL) < 0 && ivae_workspace_bytes(&ok, 0, NULL) < 0 && ivae_workspace_layout(NULL, 0, NULL) < 0, "NULL size queries");
    CHECK(ivae_elbo_fwd_bwd(&ok, &co, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) < 0, "ivae_elbo_fwd_bwd(NULL)");
    CHECK(ivae_forward(&ok, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) < 0, "ivae_forward(NULL)");
    CHECK(ivae_backward(NULL, &co, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) < 0, "ivae_backward(NULL)");
    CHECK(ivae_pairwise(64, 64, 10, &co, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL) < 0, "ivae_pairwise(NULL)");
    CHECK(ivae_elbo_adam_steps(&ok, &co, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 4, 0, 1e-3, NULL, .9, .999, 1e-8, NULL, NULL, NULL, NULL, 0, NULL) < 0, "ivae_elbo_adam_steps(NULL)");
    CHECK(ivae_train_steps(&ok, &co, NULL, NULL, NULL, NULL, NULL, 100, NULL, 0, NULL, 1, 4, 0, 1e-3, NULL, .9, .999, 1e-8, NULL, NULL, NULL, NULL) < 0, "ivae_train_steps(NULL)");
    CHECK(ivae_adam_step(NULL, NULL, NULL, NULL, 10, 1e-3, .9, .999, 1e-8, 1, NULL) < 0, "ivae_adam_step(NULL)");
    CHECK(ivae_corr(NULL, NULL, 100, 10, NULL, NULL, 0, NULL) < 0 && ivae_corr_static(NULL, NULL, 100, 10, NULL, NULL) < 0 &&
          ivae_corr_batched(NULL, NULL, 4, 64, 10, NULL, NULL) < 0, "ivae_corr*(NULL)");
    CHECK(ivae_version() && ivae_last_hip_error(), "diagnostics");
    /* ---- no device visible: the whole host-side launch preparation with fake device pointers ends in IVAE_E_HIP */
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != 0) ndev = 0;
    if (ndev == 0) {
        float* p = (float*)(uintptr_t)0x7f0000001000ull;      /* never dereferenced on the host */
        size_t wb = 0, pb = 0, cb = 0;
        const int shp[][6] = {{64, 10, 10, 40, 50, 3}, {64, 5, 5, 40, 100, 3}, {4096, 20, 20, 200, 50, 3}, {70, 4, 3, 0, 7, 2}};
        for (int s = 0; s < 4; ++s) for (int pw = 0; pw < 2; ++pw) for (int k = 1; k <= 300; k += 299) {
            IvaeDims d = {shp[s][0], shp[s][1], shp[s][2], shp[s][3], shp[s][4], shp[s][5], s == 1 ? IVAE_ACT_XTANH : IVAE_ACT_LRELU, .1f, k, 0, 0, 0};
            IvaeCoef c = co;
            c.pairwise = pw; c.global_batch = d.batch;
            if (pw && k > 1 && d.batch > 64) continue;
            if (ivae_workspace_bytes(&d, pw, &wb)) continue;
            int r = ivae_elbo_fwd_bwd(&d, &c, p, p, d.aux_dim ? p : NULL, p, p, p, p, p, wb, NULL);
            CHECK(r == IVAE_E_HIP, "ivae_elbo_fwd_bwd without a device -> %d", r);
            r = ivae_elbo_fwd_bwd(&d, &c, p, p, d.aux_dim ? p : NULL, p, p, p, p, p, wb / 2, NULL);
            CHECK(r == IVAE_E_WORKSPACE || r == IVAE_E_HIP, "short workspace -> %d", r);
            IvaeCoef cs[5] = {c, c, c, c, c};
            double mlr[300];
            for (int i = 0; i < 300; ++i) mlr[i] = 1e-3 * (1 + i % 3);
            r = ivae_elbo_adam_steps(&d, &c, cs, p, p, p, p, d.aux_dim ? p : NULL, p, 5, 7, 1e-3, mlr, .9, .999, 1e-8, p, p, p, p, wb, NULL);
            CHECK(r == IVAE_E_HIP, "ivae_elbo_adam_steps without a device -> %d", r);
            if (ivae_train_steps_supported(&d) == 1 && !pw) {
                r = ivae_train_steps(&d, &c, p, p, p, p, p, 40000, (const int32_t*)p, 0, NULL, 1, 128, 3, 1e-2, mlr, .9, .999, 1e-8, p, p, NULL, NULL);
                CHECK(r == IVAE_E_HIP, "ivae_train_steps without a device -> %d", r);
                r = ivae_train_steps(&d, &c, p, p, p, p, p, 40000, (const int32_t*)p, 40000, p, 1, 129, 3, 1e-2, NULL, .9, .999, 1e-8, p, p, p, NULL);
                CHECK(r < 0, "ivae_train_steps beyond its step limit -> %d", r);
            }
        }
        if (!ivae_pairwise_workspace_bytes(4096, 4096, 20, &pb)) {
            int r = ivae_pairwise(4096, 4096, 20, &co, p, p, p, p, p, p, p, p, p, p, p, pb, NULL);
            CHECK(r == IVAE_E_HIP, "ivae_pairwise without a device -> %d", r);
        }
        if (!ivae_corr_workspace_bytes(1000000, 20, &cb)) {
            int r = ivae_corr(p, p, 1000000, 20, (double*)p, p, cb, NULL);
            CHECK(r == IVAE_E_HIP, "ivae_corr without a device -> %d", r);
        }
        CHECK(ivae_adam_step(p, p, p, p, 100000, 1e-3, .9, .999, 1e-8, 5, NULL) == IVAE_E_HIP, "ivae_adam_step without a device");
        CHECK(ivae_corr_batched(p, p, 625, 64, 10, (double*)p, NULL) == IVAE_E_HIP, "ivae_corr_batched without a device");
    } else {
        printf("abi_asan: %d device(s) visible, fake-pointer launches skipped\n", ndev);
    }
    if (fails) { fprintf(stderr, "abi_asan: %d check(s) failed\n", fails); return 1; }
    printf("abi_asan: OK\n");
    return 0;
}

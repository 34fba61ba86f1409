/*
 * selftest.c -- drives every part of the CPU oracle on one stereo pair; meant to be built with
 * -fsanitize=address,undefined (make sanitize) so that out-of-bounds accesses and undefined
 * operations in the restatement abort the run.  TEST INFRASTRUCTURE ONLY.
 * Usage: oracle_selftest left.npy right.npy   (2-D uint8 arrays in NumPy .npy format, version 1.0)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "orb_oracle.h"

static uint8_t *load_npy_u8(const char *path, int *w, int *h)
{
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    unsigned char pre[10];
    if (fread(pre, 1, 10, f) != 10 || memcmp(pre, "\x93NUMPY", 6) != 0) { fprintf(stderr, "%s: not an npy file\n", path); exit(2); }
    const int hlen = pre[8] | (pre[9] << 8);
    char *hdr = (char *)malloc((size_t)hlen + 1);
    if (fread(hdr, 1, (size_t)hlen, f) != (size_t)hlen) exit(2);
    hdr[hlen] = 0;
    if (!strstr(hdr, "'|u1'") && !strstr(hdr, "'u1'")) { fprintf(stderr, "%s: not uint8\n", path); exit(2); }
    const char *sp = strstr(hdr, "'shape': (");
    if (!sp || sscanf(sp + 10, "%d, %d", h, w) != 2) { fprintf(stderr, "%s: not 2-D\n", path); exit(2); }
    free(hdr);
    uint8_t *img = (uint8_t *)malloc((size_t)*w * *h);
    if (fread(img, 1, (size_t)*w * *h, f) != (size_t)*w * *h) { fprintf(stderr, "%s: short file\n", path); exit(2); }
    fclose(f);
    return img;
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest: check failed at line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s left.npy right.npy\n", argv[0]); return 2; }
    int w, h, w2, h2;
    uint8_t *L = load_npy_u8(argv[1], &w, &h), *R = load_npy_u8(argv[2], &w2, &h2);
    CHECK(w == w2 && h == h2);
    oo_extractor *exl = oo_create(1000, 1.2f, 8, 20, 7), *exr = oo_create(1000, 1.2f, 8, 20, 7);
    const int cap = 2000;
    oo_keypoint *kl = calloc(cap, sizeof(*kl)), *kr = calloc(cap, sizeof(*kr)), *k2 = calloc(cap, sizeof(*k2));
    uint8_t *dl = calloc(cap, 32), *dr = calloc(cap, 32), *d2 = calloc(cap, 32);
    /* literal and derived formulations agree */
    const int nl = oo_extract(exl, L, w, h, w, kl, dl, cap, 1);
    const int n2 = oo_extract(exl, L, w, h, w, k2, d2, cap, 0);
    CHECK(nl > 500 && nl == n2 && memcmp(kl, k2, sizeof(*kl) * nl) == 0 && memcmp(dl, d2, 32 * (size_t)nl) == 0);
    const int nr = oo_extract(exr, R, w, h, w, kr, dr, cap, 1);
    CHECK(nr > 500);
    /* a small image with levels that vanish, an odd stride, a constant image */
    uint8_t *small = malloc(61 * 100);
    for (int i = 0; i < 61 * 100; i++) small[i] = (uint8_t)((i * 2654435761u) >> 24);
    oo_extractor *exs = oo_create(300, 1.2f, 8, 20, 7);
    CHECK(oo_extract(exs, small, 57, 100, 61, k2, d2, cap, 1) >= 0);
    memset(small, 90, 61 * 100);
    CHECK(oo_extract(exs, small, 61, 100, 61, k2, d2, cap, 0) == 0);
    oo_destroy(exs);
    free(small);
    /* matching */
    int32_t *bi = calloc(cap, 4), *bd = calloc(cap, 4), *sd = calloc(cap, 4), *bl = calloc(cap, 4), *sl = calloc(cap, 4);
    oo_match_bruteforce((const uint32_t *)dl, nl, (const uint32_t *)dr, nr, bi, bd, sd);
    for (int i = 0; i < nl; i++) CHECK(bi[i] >= 0 && bi[i] < nr && bd[i] <= sd[i]);
    int32_t *off = calloc((size_t)nl + 1, 4), *idx = calloc((size_t)nl * 8, 4), *toct = calloc(cap, 4);
    for (int i = 0; i < nl; i++) { off[i + 1] = off[i] + (i % 9); }
    for (int k = 0; k < off[nl]; k++) idx[k] = (k * 37) % nr;
    for (int i = 0; i < nr; i++) toct[i] = kr[i].octave;
    CHECK(off[nl] <= nl * 8);
    oo_match_windowed((const uint32_t *)dl, nl, (const uint32_t *)dr, nr, off, idx, toct, bi, bd, sd, bl, sl);
    uint16_t *dist = calloc((size_t)off[nl] + 1, 2);
    oo_hamming_csr((const uint32_t *)dl, nl, (const uint32_t *)dr, nr, off, idx, dist);
    /* stereo: one piece, classic with SAD, delayed (online, one point, offline) */
    float sf[OO_MAX_LEVELS], dummy[OO_MAX_LEVELS];
    int nf[OO_MAX_LEVELS], um[16];
    oo_get_tables(exl, sf, dummy, dummy, dummy, nf, um);
    const float mbf = 47.9f, mb = 0.11f;
    float *ur = calloc(cap, 4), *dp = calloc(cap, 4);
    oo_stereo_search(kl, (const uint32_t *)dl, nl, kr, (const uint32_t *)dr, nr, sf, h, 0.0f, mbf / mb, 0.0f, bi, bd);
    const int nm = oo_stereo_finish(kl, nl, kr, bi, bd, mbf, 0.0f, mbf / mb, ur, dp);
    CHECK(nm > 0);
    CHECK(oo_stereo_classic(exl, exr, kl, (const uint32_t *)dl, nl, kr, (const uint32_t *)dr, nr, mbf, mb, bi, bd, ur, dp) >= 0);
    oo_stereo_state st;
    st.u_right = ur; st.depth = dp;
    st.stereo_matched = calloc(cap, 1);
    st.dist_idx_dist = calloc(cap, 4); st.dist_idx_il = calloc(cap, 4);
    oo_stereo_prepare(&st, nl);
    uint8_t *mp = calloc(cap, 1);
    float *prior = calloc(cap, 4);
    for (int i = 0; i < nl; i++) { mp[i] = (uint8_t)(i % 3 == 0); prior[i] = (i % 2) ? 30.0f + (float)(i % 50) : -1.0f; }
    const int on = oo_stereo_undistorted(kl, (const uint32_t *)dl, nl, kr, (const uint32_t *)dr, nr, sf, h, mbf, mb, 0.0f, 50.0f,
                                         mp, prior, 1, &st, bi, bd);
    mp[1] = 1;
    (void)oo_stereo_one_point(kl, (const uint32_t *)dl, nl, kr, (const uint32_t *)dr, nr, sf, h, mbf, mp, 1, 75, 0.0f, 200.0f, &st, bi, bd);
    (void)oo_stereo_one_point(kl, (const uint32_t *)dl, nl, kr, (const uint32_t *)dr, nr, sf, h, mbf, mp, -5, 75, 0.0f, 200.0f, &st, bi, bd);
    const int offn = oo_stereo_undistorted(kl, (const uint32_t *)dl, nl, kr, (const uint32_t *)dr, nr, sf, h, mbf, mb, 0.0f, 50.0f,
                                           mp, prior, 0, &st, bi, bd);
    CHECK(on >= 0 && offn >= -nl && st.n_dist_idx <= nl);
    /* frame grid + undistortion */
    float qx[64], qy[64], qr[64];
    int32_t qmin[64], qmax[64], goff[65];
    for (int q = 0; q < 64; q++) { qx[q] = (float)(q * 13 - 40); qy[q] = (float)(q * 9 - 30); qr[q] = (float)(3 + q); qmin[q] = q % 4 - 1; qmax[q] = (q % 3) ? -1 : q % 8; }
    const int tot = oo_features_in_area(kl, nl, 0.f, 0.f, 64.f / (float)w, 48.f / (float)h, qx, qy, qr, qmin, qmax, 64, goff, NULL, 0);
    int32_t *gidx = calloc((size_t)tot + 1, 4);
    CHECK(oo_features_in_area(kl, nl, 0.f, 0.f, 64.f / (float)w, 48.f / (float)h, qx, qy, qr, qmin, qmax, 64, goff, gidx, tot) == tot);
    const double K[9] = {458.654, 0, 367.215, 0, 457.296, 248.375, 0, 0, 1};
    const double dist4[4] = {-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05};
    float *xy = calloc((size_t)nl * 2, 4), *xyo = calloc((size_t)nl * 2, 4);
    for (int i = 0; i < nl; i++) { xy[2 * i] = kl[i].x; xy[2 * i + 1] = kl[i].y; }
    oo_undistort_points(xy, nl, K, dist4, 4, NULL, K, 3, 0, xyo);
    oo_undistort_points(xy, nl, K, dist4, 4, NULL, K, 3, 1, xyo);
    uint64_t counts[2], tested;
    oo_count_trig_differences(0, 100003, counts, &tested);
    CHECK(tested > 10000);
    printf("selftest ok: %d / %d keypoints, %d stereo matches, %d grid candidates\n", nl, nr, nm, tot);
    free(kl); free(kr); free(k2); free(dl); free(dr); free(d2); free(bi); free(bd); free(sd); free(bl); free(sl);
    free(off); free(idx); free(toct); free(dist); free(ur); free(dp); free(st.stereo_matched); free(st.dist_idx_dist);
    free(st.dist_idx_il); free(mp); free(prior); free(gidx); free(xy); free(xyo); free(L); free(R);
    oo_destroy(exl); oo_destroy(exr);
    return 0;
}

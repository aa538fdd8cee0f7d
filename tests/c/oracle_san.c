/* Sanitizer run of the C oracle (CPU only): built by tests/test_oracle_golden.py with
 * -fsanitize=address,undefined and executed on ragged sizes, including the degenerate ones
 * (one ray, one pixel, a ray parallel to the target plane).  Exits 0 when no report fires. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

void oracle_geometry(int B, int N, const float* helios, const float* sun, const float* action, const float* trig,
                     const float* target_pos, const float* target_normal, float* actual, float* refl, float* inter,
                     float* mask);
void oracle_splat(int B, int N, int R, const float* inter, const float* mask, const float* helios, const float* origin,
                  const float* u, const float* v, const float* xs, const float* ys, float sigma_scale, float* image);
void oracle_ideal_normals(int B, int N, const float* helios, const float* sun, const float* target, float* out);

static unsigned long long s = 88172645463325252ull;
static float rnd(void) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 16777216.0f; }

static int one(int B, int N, int R) {
    float *helios = malloc(12u * N), *sun = malloc(12u * B), *action = malloc(12u * B * N), *trig = malloc(16u * B * N);
    float *actual = malloc(12u * B * N), *refl = malloc(12u * B * N), *inter = malloc(12u * B * N), *mask = malloc(4u * B * N);
    float *xs = malloc(4u * R), *ys = malloc(4u * R), *img = malloc(4u * (size_t)B * R * R), *ideal = malloc(12u * B * N);
    const float tp[3] = {0.f, -5.f, 0.f}, tn[3] = {0.f, 1.f, 0.f}, u[3] = {1.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 1.f};
    for (int i = 0; i < 3 * N; ++i) helios[i] = 80.f + 10.f * rnd();
    for (int i = 0; i < 3 * B; ++i) sun[i] = 1e4f * (rnd() + 0.1f);
    for (int i = 0; i < 3 * B * N; ++i) action[i] = rnd() - 0.3f;
    action[0] = 1.f; action[1] = 0.f; action[2] = 0.f;           /* may reflect parallel to the plane */
    for (int i = 0; i < B * N; ++i) { const float e = 0.1f * (rnd() - 0.5f); trig[4 * i] = cosf(e); trig[4 * i + 1] = sinf(e); trig[4 * i + 2] = cosf(-e); trig[4 * i + 3] = sinf(-e); }
    for (int i = 0; i < R; ++i) xs[i] = ys[i] = R > 1 ? -7.5f + 15.f * i / (R - 1) : -7.5f;
    oracle_ideal_normals(B, N, helios, sun, tp, ideal);
    oracle_geometry(B, N, helios, sun, action, trig, tp, tn, actual, refl, inter, mask);
    oracle_splat(B, N, R, inter, mask, helios, tp, u, v, xs, ys, 0.02f, img);
    int bad = 0;
    for (size_t p = 0; p < (size_t)B * R * R; ++p) bad |= !(img[p] >= 0.0f);     /* also catches NaN */
    for (int i = 0; i < 3 * B * N; ++i) bad |= !isfinite(actual[i]) || !isfinite(ideal[i]);
    free(helios); free(sun); free(action); free(trig); free(actual); free(refl); free(inter); free(mask);
    free(xs); free(ys); free(img); free(ideal);
    return bad;
}

int main(void) {
    const int sizes[][3] = {{1, 1, 1}, {1, 1, 2}, {2, 3, 5}, {3, 7, 33}, {5, 65, 17}, {2, 130, 64}};
    int bad = 0;
    for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); ++k) bad |= one(sizes[k][0], sizes[k][1], sizes[k][2]);
    printf(bad ? "ORACLE SAN FAILED\n" : "ORACLE SAN OK\n");
    return bad;
}

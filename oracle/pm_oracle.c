/*
 * TEST INFRASTRUCTURE -- not part of the product.
 *
 * Single-thread CPU restatement of the particle loops of the reference's PM
 * step.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this; the product path never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here bit for bit against the .npz files under tests/golden, which were written by running
 * the reference's own functions (tools/make_golden.py).
 *
 * Arithmetic contract (what numba compiles the reference loops to):
 *   - box_size is an int64 and grid_size/box_size a true divide, so wrapping,
 *     scaling, the fractional parts and all eight weights are float64, also
 *     for float32 particle positions (they are widened, value preserving);
 *   - the accumulators (the density mesh, the force rows) are float32: every
 *     `acc += term` widens acc to float64, adds, and rounds back to float32;
 *   - no fused multiply-add (build with -ffp-contract=off), matching the
 *     plain-Python execution that produced the golden vectors.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* Python-style float modulo for a positive divisor: result in [0, L]. */
static inline double pymod(double x, double L)
{
    double r = fmod(x, L);
    if (r != 0.0) {
        if (r < 0.0) r += L;
    } else {
        r = 0.0;                    /* copysign(0, L), L > 0 */
    }
    return r;
}

/* float32 np.remainder(x, L), L > 0 (main.py:60): may return exactly L. */
static inline float npremf(float x, float L)
{
    float r = fmodf(x, L);
    if (r != 0.0f) {
        if (r < 0.0f) r += L;
    } else {
        r = 0.0f;
    }
    return r;
}

typedef struct { int64_t i, j, k, i1, j1, k1; double wx0, wx1, wy0, wy1, wz0, wz1; } corner_t;

/* functions.py:27-58 and :108-137 (identical in both loops). */
static inline corner_t corners(const double *p, int64_t G, double L, double inv)
{
    corner_t c;
    double sx = pymod(p[0], L) * inv, sy = pymod(p[1], L) * inv, sz = pymod(p[2], L) * inv;
    c.i = (int64_t)sx; c.j = (int64_t)sy; c.k = (int64_t)sz;
    double dx = sx - (double)c.i, dy = sy - (double)c.j, dz = sz - (double)c.k;
    c.wx0 = 1.0 - dx; c.wx1 = dx; c.wy0 = 1.0 - dy; c.wy1 = dy; c.wz0 = 1.0 - dz; c.wz1 = dz;
    /* A coordinate whose modulo rounds up to L itself (a tiny negative one: -1e-35 % 100 == 100.0) scales to exactly G.
     * The reference then indexes plane G of its mesh: an IndexError in plain Python, a write past the array under
     * numba.  There is nothing to restate; such a particle is put into cell 0 with offset 0, where the periodic box
     * has it (the library does the same). */
    if (c.i >= G) c.i -= G;
    if (c.j >= G) c.j -= G;
    if (c.k >= G) c.k -= G;
    c.i1 = (c.i + 1) % G; c.j1 = (c.j + 1) % G; c.k1 = (c.k + 1) % G;
    return c;
}

/* cell (i,j,k) of every particle: functions.py:27-39.  out is (n,3) int32. */
void oracle_cell_index(const double *pos, int64_t n, int64_t G, int64_t box, int32_t *out)
{
    double L = (double)box, inv = (double)G / (double)box;
    for (int64_t p = 0; p < n; ++p) {
        corner_t c = corners(pos + 3 * p, G, L, inv);
        out[3 * p] = (int32_t)c.i; out[3 * p + 1] = (int32_t)c.j; out[3 * p + 2] = (int32_t)c.k;
    }
}

#define DEP(I, J, K, W) do { float *d = density + (((I) * G + (J)) * G + (K)); \
        *d = (float)((double)*d + (W)); } while (0)

/* CIC mass deposition: functions.py:19-70.  density is (G,G,G) float32, zeroed here. */
void oracle_cic(const double *pos, const float *mass, int64_t n, int64_t G, int64_t box, float *density)
{
    double L = (double)box, inv = (double)G / (double)box;
    memset(density, 0, sizeof(float) * (size_t)(G * G * G));
    for (int64_t p = 0; p < n; ++p) {
        corner_t c = corners(pos + 3 * p, G, L, inv);
        double m = (double)mass[p];
        DEP(c.i,  c.j,  c.k,  m * c.wx0 * c.wy0 * c.wz0);     /* functions.py:61-68, same order */
        DEP(c.i1, c.j,  c.k,  m * c.wx1 * c.wy0 * c.wz0);
        DEP(c.i,  c.j1, c.k,  m * c.wx0 * c.wy1 * c.wz0);
        DEP(c.i,  c.j,  c.k1, m * c.wx0 * c.wy0 * c.wz1);
        DEP(c.i1, c.j1, c.k,  m * c.wx1 * c.wy1 * c.wz0);
        DEP(c.i1, c.j,  c.k1, m * c.wx1 * c.wy0 * c.wz1);
        DEP(c.i,  c.j1, c.k1, m * c.wx0 * c.wy1 * c.wz1);
        DEP(c.i1, c.j1, c.k1, m * c.wx1 * c.wy1 * c.wz1);
    }
}

/* The same deposit with float64 accumulation: the "truth" that the accuracy
 * tests rank the reference-order and the GPU results against. */
void oracle_cic_f64(const double *pos, const float *mass, int64_t n, int64_t G, int64_t box, double *density)
{
    double L = (double)box, inv = (double)G / (double)box;
    memset(density, 0, sizeof(double) * (size_t)(G * G * G));
#define DEP64(I, J, K, W) density[((I) * G + (J)) * G + (K)] += (W)
    for (int64_t p = 0; p < n; ++p) {
        corner_t c = corners(pos + 3 * p, G, L, inv);
        double m = (double)mass[p];
        DEP64(c.i,  c.j,  c.k,  m * c.wx0 * c.wy0 * c.wz0);
        DEP64(c.i1, c.j,  c.k,  m * c.wx1 * c.wy0 * c.wz0);
        DEP64(c.i,  c.j1, c.k,  m * c.wx0 * c.wy1 * c.wz0);
        DEP64(c.i,  c.j,  c.k1, m * c.wx0 * c.wy0 * c.wz1);
        DEP64(c.i1, c.j1, c.k,  m * c.wx1 * c.wy1 * c.wz0);
        DEP64(c.i1, c.j,  c.k1, m * c.wx1 * c.wy0 * c.wz1);
        DEP64(c.i,  c.j1, c.k1, m * c.wx0 * c.wy1 * c.wz1);
        DEP64(c.i1, c.j1, c.k1, m * c.wx1 * c.wy1 * c.wz1);
    }
}

/* CIC force gather: functions.py:99-180.  forces is (n,3) float32. */
void oracle_force_cic(const float *gx, const float *gy, const float *gz, const double *pos,
                      int64_t n, int64_t G, int64_t box, float *forces)
{
    double L = (double)box, inv = (double)G / (double)box;
    for (int64_t p = 0; p < n; ++p) {
        corner_t c = corners(pos + 3 * p, G, L, inv);
        float fx = 0.0f, fy = 0.0f, fz = 0.0f;
#define GAT(I, J, K, WX, WY, WZ) do { int64_t q = ((I) * G + (J)) * G + (K); double w = (WX) * (WY) * (WZ); \
        fx = (float)((double)fx + (double)gx[q] * w); \
        fy = (float)((double)fy + (double)gy[q] * w); \
        fz = (float)((double)fz + (double)gz[q] * w); } while (0)
        GAT(c.i,  c.j,  c.k,  c.wx0, c.wy0, c.wz0);           /* functions.py:140-178, same order */
        GAT(c.i1, c.j,  c.k,  c.wx1, c.wy0, c.wz0);
        GAT(c.i,  c.j1, c.k,  c.wx0, c.wy1, c.wz0);
        GAT(c.i,  c.j,  c.k1, c.wx0, c.wy0, c.wz1);
        GAT(c.i1, c.j1, c.k,  c.wx1, c.wy1, c.wz0);
        GAT(c.i1, c.j,  c.k1, c.wx1, c.wy0, c.wz1);
        GAT(c.i,  c.j1, c.k1, c.wx0, c.wy1, c.wz1);
        GAT(c.i1, c.j1, c.k1, c.wx1, c.wy1, c.wz1);
        forces[3 * p] = fx; forces[3 * p + 1] = fy; forces[3 * p + 2] = fz;
    }
}

/* np.gradient(potential) with unit spacing and edge_order=1 (call site main.py:54):
 * interior (f[n+1]-f[n-1])*0.5, one-sided first differences on the two faces of
 * every axis, no periodic wrap.  float32 throughout. */
void oracle_gradient(const float *f, int64_t G, float *gx, float *gy, float *gz)
{
    int64_t s0 = G * G, s1 = G;
    for (int64_t i = 0; i < G; ++i)
        for (int64_t j = 0; j < G; ++j)
            for (int64_t k = 0; k < G; ++k) {
                int64_t q = i * s0 + j * s1 + k;
                gx[q] = (i == 0) ? f[q + s0] - f[q] : (i == G - 1) ? f[q] - f[q - s0] : (f[q + s0] - f[q - s0]) * 0.5f;
                gy[q] = (j == 0) ? f[q + s1] - f[q] : (j == G - 1) ? f[q] - f[q - s1] : (f[q + s1] - f[q - s1]) * 0.5f;
                gz[q] = (k == 0) ? f[q + 1] - f[q] : (k == G - 1) ? f[q] - f[q - 1] : (f[q + 1] - f[q - 1]) * 0.5f;
            }
}

/* main.py:58-60 with float32 positions and velocities:
 *   velocities += forces*dt   (forces*dt is float32: dt is a weak Python float)
 *   positions  += velocities*dt
 *   positions   = positions.astype(float32) % box        (float32 remainder) */
void oracle_kick_drift_f32(float *pos, float *vel, const float *forces, int64_t n3, double dt, int64_t box)
{
    float dtf = (float)dt, L = (float)box;
    for (int64_t q = 0; q < n3; ++q) {
        float dv = forces[q] * dtf;
        vel[q] = vel[q] + dv;
        float dx = vel[q] * dtf;
        pos[q] = npremf(pos[q] + dx, L);
    }
}

/* The shipped default run: float64 velocities; positions float64 on entry of
 * step 0 and float32 afterwards (pos_in is double either way, pos_out float). */
void oracle_kick_drift_f64v(const double *pos_in, float *pos_out, double *vel, const float *forces,
                            int64_t n3, double dt, int64_t box)
{
    float dtf = (float)dt, L = (float)box;
    for (int64_t q = 0; q < n3; ++q) {
        float dv = forces[q] * dtf;                 /* float32 product, then widened */
        vel[q] = vel[q] + (double)dv;
        double x = pos_in[q] + vel[q] * dt;         /* float64 loop of the in-place add */
        pos_out[q] = npremf((float)x, L);           /* f64 array + astype(f32), or f32 array: cast on store */
    }
}

// Mesh-side kernels of the PM step for gfx950: the Green's-function multiply
// on the half spectrum and the np.gradient stencil.
#include "pm_internal.hpp"

namespace {

constexpr int BLOCK = 256;

// np.fft.fftfreq(G)[a] as float32 (functions.py:75): integer frequency times 1/G in float64, then rounded.
__device__ __forceinline__ float freq(int a, int G, double invG)
{
    int f = (a < (G + 1) / 2) ? a : a - G;
    return (float)((double)f * invG);
}

// potential_k = density_k / k2 with the DC term zeroed (functions.py:85-86), times the 1/M of
// the inverse transform (functions.py:89).  NumPy divides complex64 by a real float32 by
// multiplying with the float32 reciprocal; k2 = (kx^2 + ky^2) + kz^2 in float32 (functions.py:77).
__global__ void k_green(float2 *__restrict__ spec, pm_geom g, float invM)
{
    const int G = g.G, Gh = g.Gh;
    int64_t total = (int64_t)G * G * Gh;
    int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= total) return;
    int c = (int)(q % Gh);
    int b = (int)((q / Gh) % G);
    int a = (int)(q / ((int64_t)Gh * G));
    double invG = 1.0 / (double)G;
    float kx = freq(a, G, invG), ky = freq(b, G, invG), kz = freq(c, G, invG);
    float k2 = (kx * kx + ky * ky) + kz * kz;
    float2 v = spec[q];
    if ((a | b | c) == 0) {
        v.x = 0.0f; v.y = 0.0f;
    } else {
        float scl = 1.0f / k2;
        if (g.soft != 0.0f) scl *= expf(-g.soft * k2);
        v.x = (v.x * scl) * invM;
        v.y = (v.y * scl) * invM;
    }
    spec[q] = v;
}

// np.gradient(potential), unit spacing, edge_order 1 (main.py:54): central difference times 0.5
// inside, one-sided first differences on the two faces of each axis (no periodic wrap).
__global__ void k_gradient(const float *__restrict__ f, float *__restrict__ gx, float *__restrict__ gy,
                           float *__restrict__ gz, pm_geom g)
{
    const int G = g.G;
    int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= g.M) return;
    int k = (int)(q % G), j = (int)((q / G) % G), i = (int)(q / ((int64_t)G * G));
    int64_t s0 = (int64_t)G * G, s1 = G;
    float c = f[q];
    gx[q] = (i == 0) ? f[q + s0] - c : (i == G - 1) ? c - f[q - s0] : (f[q + s0] - f[q - s0]) * 0.5f;
    gy[q] = (j == 0) ? f[q + s1] - c : (j == G - 1) ? c - f[q - s1] : (f[q + s1] - f[q - s1]) * 0.5f;
    gz[q] = (k == 0) ? f[q + 1] - c : (k == G - 1) ? c - f[q - 1] : (f[q + 1] - f[q - 1]) * 0.5f;
}

// The same multiply on a transposed slab of the half spectrum: (G, ny, Ghp) with rows y0 .. y0 + ny - 1.
__global__ void k_green_t(float2 *__restrict__ spec, pm_geom g, float invM, int y0, int ny)
{
    const int G = g.G, Gh = g.Gh, P = g.Ghp;
    int64_t total = (int64_t)G * ny * Gh;
    int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= total) return;
    int c = (int)(q % Gh);
    int bl = (int)((q / Gh) % ny), b = bl + y0;
    int a = (int)(q / ((int64_t)Gh * ny));
    double invG = 1.0 / (double)G;
    float kx = freq(a, G, invG), ky = freq(b, G, invG), kz = freq(c, G, invG);
    float k2 = (kx * kx + ky * ky) + kz * kz;
    float2 *e = spec + ((int64_t)a * ny + bl) * P + c;
    float2 v = *e;
    if ((a | b | c) == 0) {
        v.x = 0.0f; v.y = 0.0f;
    } else {
        float scl = 1.0f / k2;
        if (g.soft != 0.0f) scl *= expf(-g.soft * k2);
        v.x = (v.x * scl) * invM;
        v.y = (v.y * scl) * invM;
    }
    *e = v;
}

// Slab (Gs, G, Ghp) of the half spectrum <-> its blocks for the all-to-all: block p holds rows
// y = p*Gs .. (p+1)*Gs - 1 of every owned plane, (Gs planes, Gs rows, Ghp), planes first.
__global__ void k_transpose_blocks(const float2 *__restrict__ in, float2 *__restrict__ out, int Gs, int G, int P, bool unpack)
{
    int64_t total = (int64_t)Gs * G * P;
    int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q >= total) return;
    int z = (int)(q % P);
    int y = (int)((q / P) % G);
    int x = (int)(q / ((int64_t)P * G));
    int64_t blocked = (((int64_t)(y / Gs) * Gs + x) * Gs + (y % Gs)) * P + z;
    if (unpack) out[q] = in[blocked]; else out[blocked] = in[q];
}

__global__ void k_add(float *__restrict__ mesh, const float *__restrict__ other, int64_t n)
{
    int64_t q = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (q < n) mesh[q] = mesh[q] + other[q];
}

}  // namespace

hipError_t pm_launch_add(pm_ctx *c, float *mesh, const float *other, int64_t n)
{
    if (n == 0) return hipSuccess;
    k_add<<<dim3((unsigned)((n + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(mesh, other, n);
    return hipGetLastError();
}

hipError_t pm_launch_green_t(pm_ctx *c, float2 *spec_t, int y0, int ny)
{
    int64_t total = (int64_t)c->g.G * ny * c->g.Gh;
    float invM = (float)(1.0 / (double)c->g.M);
    k_green_t<<<dim3((unsigned)((total + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(spec_t, c->g, invM, y0, ny);
    return hipGetLastError();
}

hipError_t pm_launch_transpose_pack(pm_ctx *c, const float2 *in, float2 *out, bool unpack)
{
    int64_t total = (int64_t)c->slabn * c->g.G * c->g.Ghp;        // the slab's own planes
    k_transpose_blocks<<<dim3((unsigned)((total + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(in, out, c->slabn, c->g.G, c->g.Ghp, unpack);
    return hipGetLastError();
}

hipError_t pm_launch_green(pm_ctx *c)
{
    int64_t total = (int64_t)c->g.G * c->g.G * c->g.Gh;
    float invM = (float)(1.0 / (double)c->g.M);
    k_green<<<dim3((unsigned)((total + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(c->spectrum, c->g, invM);
    return hipGetLastError();
}

hipError_t pm_launch_gradient(pm_ctx *c)
{
    k_gradient<<<dim3((unsigned)((c->g.M + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(c->potential, c->grad[0], c->grad[1], c->grad[2], c->g);
    return hipGetLastError();
}

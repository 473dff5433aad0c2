// Synthetic particle sets drawn on the device, straight into a context's particle records: the uniform set, the
// two-galaxy collision (two truncated Plummer spheres on a collision course) and the exponential disk.
//
// The distributions are those of the reference's generators -- galaxy_generation_functions.py:97-136 (disk: radius
// Gamma(2, Rd), height Laplace(0, z0), both truncated by rejection, uniform azimuth) and :267-288 (Plummer profile
// rho ~ (1 + r^2/a^2)^-5/2) -- and the placement of two galaxies about the box centre with a zero-momentum
// relative velocity is that of functions.py:320-338.  The random stream is not the reference's (NumPy's PCG64 consumed
// array by array cannot be drawn particle by particle): every particle draws from Philox4x32-10 with its own index as
// the counter, so a set is a pure function of (kind, n, seed, parameters) whatever the launch shape.
#include "pm_internal.hpp"
#include "pm_philox_dev.hpp"
#include "pm_cell_dev.hpp"

namespace {

constexpr int BLOCK = 256;
constexpr int MAX_ATTEMPTS = 64;        // rejection loops end here (acceptance is > 95 % per attempt)

using pmrand::u32x4;
using pmrand::unit;
using pmrand::draw_t;

// float32 np.remainder(x, L), L > 0: the wrap of main.py:60
__device__ __forceinline__ float wrap32(float x, float L)
{
    if (x >= 0.0f && x < L) return (x == 0.0f) ? 0.0f : x;
    float r = fmodf(x, L);
    if (r != 0.0f) { if (r < 0.0f) r += L; } else r = 0.0f;
    return r;
}

struct ic_params { double p[16]; };

constexpr double TWO_PI = 6.283185307179586476925286766559;

// particle number idx of the set (kind, n, seed, q): its records
// purposes (third counter word): 0 radius / position, 1 velocity, 2 height of the disk
__device__ __forceinline__ void draw_particle(int64_t idx, int64_t n, int kind, uint64_t seed, const ic_params &q, const pm_geom &g,
                                              pm_prec<float> &p, pm_vrec<float> &v)
{
    const draw_t draw{(uint32_t)idx, (uint32_t)((uint64_t)idx >> 32), (uint32_t)seed, (uint32_t)(seed >> 32)};
    double x, y, z, sigma, bx = 0.0, by = 0.0, bz = 0.0;     // position, velocity dispersion, bulk velocity
    if (kind == PM_IC_UNIFORM) {
        const u32x4 b = draw(0, 0);
        x = unit(b.v[0]) * g.L; y = unit(b.v[1]) * g.L; z = unit(b.v[2]) * g.L;
        sigma = q.p[0];
    } else if (kind == PM_IC_TWO_GALAXY) {
        // params: a1, a2, rmax, centre1 (3), centre2 (3), relative velocity (3), sigma
        const int64_t n1 = n / 2;
        const bool second = idx >= n1;
        const double a = q.p[second ? 1 : 0], rmax = q.p[2];
        const double *c = q.p + (second ? 6 : 3);
        double r = rmax;
        u32x4 b{};
        for (int t = 0; t < MAX_ATTEMPTS; ++t) {
            b = draw(0, (uint32_t)t);
            // inverse of the Plummer mass profile M(<r)/M = (1 + a^2/r^2)^-3/2
            const double rr = a / sqrt(pow(unit(b.v[0]), -2.0 / 3.0) - 1.0);
            if (rr < rmax) { r = rr; break; }
        }
        const double cz = 1.0 - 2.0 * unit(b.v[1]), phi = TWO_PI * unit(b.v[2]);
        const double s = sqrt(fmax(0.0, 1.0 - cz * cz));
        x = c[0] + r * (s * cos(phi)); y = c[1] + r * (s * sin(phi)); z = c[2] + r * cz;
        // functions.py:329-338 with equal total masses: each galaxy takes half of the relative velocity
        const double h = second ? 0.5 : -0.5;
        bx = h * q.p[9]; by = h * q.p[10]; bz = h * q.p[11];
        sigma = q.p[12];
    } else {
        // params: Rd, z0, Rmax, zmax, centre (3), sigma
        const double Rd = q.p[0], z0 = q.p[1], Rmax = q.p[2], zmax = q.p[3];
        double R = Rmax, h = 0.0;
        u32x4 b{};
        for (int t = 0; t < MAX_ATTEMPTS; ++t) {
            b = draw(0, (uint32_t)t);
            const double rr = -Rd * log(unit(b.v[0]) * unit(b.v[1]));       // Gamma(2, Rd): two exponentials
            if (rr <= Rmax) { R = rr; break; }
        }
        const double phi = TWO_PI * unit(b.v[2]);
        for (int t = 0; t < MAX_ATTEMPTS; ++t) {
            const double w = unit(draw(2, (uint32_t)t).v[0]);
            const double hh = w < 0.5 ? z0 * log(2.0 * w) : -z0 * log(2.0 * (1.0 - w));    // Laplace(0, z0)
            if (fabs(hh) <= zmax) { h = hh; break; }
        }
        x = q.p[4] + R * cos(phi); y = q.p[5] + R * sin(phi); z = q.p[6] + h;
        sigma = q.p[7];
    }
    // isotropic Maxwellian about the bulk velocity (Box-Muller on two pairs)
    const u32x4 b = draw(1, 0);
    const double m0 = sqrt(-2.0 * log(unit(b.v[0]))), a0 = TWO_PI * unit(b.v[1]);
    const double m1 = sqrt(-2.0 * log(unit(b.v[2]))), a1 = TWO_PI * unit(b.v[3]);
    p.x = wrap32((float)x, g.Lf); p.y = wrap32((float)y, g.Lf); p.z = wrap32((float)z, g.Lf);
    p.m = 1.0f;
    v.x = (float)(bx + sigma * (m0 * cos(a0))); v.y = (float)(by + sigma * (m0 * sin(a0))); v.z = (float)(bz + sigma * (m1 * cos(a1)));
    v.id = (uint32_t)idx;
}

__global__ void __launch_bounds__(BLOCK)
k_generate(pm_prec<float> *__restrict__ P, pm_vrec<float> *__restrict__ V, int64_t n, int kind, uint64_t seed, ic_params q, pm_geom g)
{
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    pm_prec<float> p;
    pm_vrec<float> v;
    draw_particle(idx, n, kind, seed, q, g, p, v);
    P[idx] = p;
    V[idx] = v;
}

// plane of the mesh a position lies in (functions.py:27-39, float64 like the particle kernels)
__device__ __forceinline__ int plane_of(float x, const pm_geom &g) { return pmcell::locate((double)x, 0.0, 0.0, g).i; }

// A slab keeps the particles of the set whose cell lies in its window [g.i0, g.i0 + g.Gs), in the order of their numbers:
// MODE 0 counts them per block of BLOCK numbers, MODE 1 writes them, block b's from slot block_first[b] on.
template <int MODE>
__global__ void __launch_bounds__(BLOCK)
k_generate_window(pm_prec<float> *__restrict__ P, pm_vrec<float> *__restrict__ V, int64_t n, int kind, uint64_t seed, ic_params q, pm_geom g,
                  uint32_t *__restrict__ block_counts, const uint32_t *__restrict__ block_first)
{
    __shared__ uint32_t wave_keeps[BLOCK / 64];
    const int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    pm_prec<float> p{};
    pm_vrec<float> v{};
    bool keep = false;
    if (idx < n) {
        draw_particle(idx, n, kind, seed, q, g, p, v);
        const int il = plane_of(p.x, g) - g.i0;
        keep = g.Gs == g.G || (il >= 0 && il < g.Gs);
    }
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) wave_keeps[w] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int k = 0; k < BLOCK / 64; ++k) { if (k < w) before += wave_keeps[k]; total += wave_keeps[k]; }
    if (MODE == 0) {
        if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
    } else if (keep) {
        const uint32_t slot = block_first[blockIdx.x] + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        P[slot] = p;
        V[slot] = v;
    }
}

__global__ void __launch_bounds__(BLOCK)
k_generate_planes(int64_t n, int kind, uint64_t seed, ic_params q, pm_geom g, unsigned long long *__restrict__ counts)
{
    extern __shared__ uint32_t planes[];                // G counters
    for (int k = threadIdx.x; k < g.G; k += BLOCK) planes[k] = 0u;
    __syncthreads();
    for (int64_t idx = (int64_t)blockIdx.x * BLOCK + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * BLOCK) {
        pm_prec<float> p;
        pm_vrec<float> v;
        draw_particle(idx, n, kind, seed, q, g, p, v);
        atomicAdd(planes + plane_of(p.x, g), 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < g.G; k += BLOCK)
        if (planes[k]) atomicAdd(counts + k, (unsigned long long)planes[k]);
}

}  // namespace

int pm_generate_param_count(int kind)
{
    return kind == PM_IC_UNIFORM ? 1 : kind == PM_IC_TWO_GALAXY ? 13 : kind == PM_IC_DISK ? 8 : -1;
}

// writes float32 records for particles 0 .. n-1 into the live buffers (pf[cur], vf[vcur])
hipError_t pm_launch_generate(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params)
{
    if (n == 0) return hipSuccess;
    ic_params q{};
    for (int k = 0; k < pm_generate_param_count(kind); ++k) q.p[k] = params[k];
    k_generate<<<dim3((unsigned)((n + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(c->pf[c->cur], c->vf[c->vcur], n, kind, seed, q, c->g);
    return hipGetLastError();
}

static ic_params params_of(int kind, const double *params)
{
    ic_params q{};
    for (int k = 0; k < pm_generate_param_count(kind); ++k) q.p[k] = params[k];
    return q;
}

hipError_t pm_launch_generate_count(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params, uint32_t *block_counts)
{
    if (n == 0) return hipSuccess;
    k_generate_window<0><<<dim3((unsigned)((n + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(nullptr, nullptr, n, kind, seed, params_of(kind, params), c->g,
                                                                                             block_counts, nullptr);
    return hipGetLastError();
}

hipError_t pm_launch_generate_keep(pm_ctx *c, int kind, int64_t n, uint64_t seed, const double *params, const uint32_t *block_first)
{
    if (n == 0) return hipSuccess;
    k_generate_window<1><<<dim3((unsigned)((n + BLOCK - 1) / BLOCK)), BLOCK, 0, c->stream>>>(c->pf[c->cur], c->vf[c->vcur], n, kind, seed,
                                                                                             params_of(kind, params), c->g, nullptr, block_first);
    return hipGetLastError();
}

hipError_t pm_launch_generate_planes(const pm_geom &g, hipStream_t stream, int kind, int64_t n, uint64_t seed, const double *params,
                                     unsigned long long *plane_counts)
{
    if (n == 0) return hipSuccess;
    const int64_t want = (n + BLOCK - 1) / BLOCK;
    k_generate_planes<<<(unsigned)(want < 8192 ? want : 8192), BLOCK, (size_t)g.G * sizeof(uint32_t), stream>>>(n, kind, seed, params_of(kind, params), g, plane_counts);
    return hipGetLastError();
}

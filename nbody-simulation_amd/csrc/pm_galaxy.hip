// The reference's importance-sampled galaxy generators on the device, in float64:
//   generate_ellipse_importance          galaxy_generation_functions.py:181-244, velocities by virial_ellipsoid :9-94
//   generate_diffuse_sphere_importance   galaxy_generation_functions.py:267-383
// Everything behind the random draws is deterministic -- importance weights normalised to the total mass, enclosed
// mass from the particles sorted by radius, circular speeds, tangential directions -- and is computed here particle by
// particle, with reductions (maxima, sums, weighted means), one radix sort of the radii and one prefix sum.  The
// standard normal deviates either come from the caller (the host mirror passes what NumPy's Generator hands out for
// the seed, so that a seed names the same galaxy as in the reference) or from Philox4x32-10 on the device.
#include "pm_internal.hpp"
#include "pm_philox_dev.hpp"

#include <cmath>
#include <vector>

namespace {

constexpr int BLOCK = 256;
constexpr int PARTIALS = 1024;          // partial results of a reduction, finished on the host in index order
constexpr double EPS = 1e-12;
constexpr double TWO_PI = 6.283185307179586476925286766559;

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + BLOCK - 1) / BLOCK)); }

// ---- reductions: PARTIALS blocks stride over the input, a fixed tree inside the block ------------------------------
enum { OP_SUM = 0, OP_MAX = 1 };

template <int OP> __device__ __forceinline__ double combine(double a, double b) { return OP == OP_SUM ? a + b : fmax(a, b); }

// value of element i: in[i * stride] (* weight[i] if weight) for i in [first, last)
template <int OP>
__global__ void __launch_bounds__(BLOCK)
k_reduce(const double *__restrict__ in, int stride, const double *__restrict__ weight, int64_t first, int64_t last, double *__restrict__ partial)
{
    __shared__ double sh[BLOCK];
    double acc = OP == OP_SUM ? 0.0 : -INFINITY;
    for (int64_t i = first + (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < last; i += (int64_t)gridDim.x * BLOCK) {
        const double v = weight ? in[i * stride] * weight[i] : in[i * stride];
        acc = combine<OP>(acc, v);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int d = BLOCK / 2; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) sh[threadIdx.x] = combine<OP>(sh[threadIdx.x], sh[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// ---- inclusive prefix sum of doubles: blocks of SCAN_TILE, block sums scanned on the host ---------------------------
constexpr int SCAN_ITEMS = 8, SCAN_TILE = BLOCK * SCAN_ITEMS;

__global__ void __launch_bounds__(BLOCK)
k_scan_tiles(const double *__restrict__ in, double *__restrict__ out, double *__restrict__ tile_sum, int64_t n)
{
    __shared__ double sh[BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    double v[SCAN_ITEMS], run = 0.0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        run += base + k < n ? in[base + k] : 0.0;
        v[k] = run;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int d = 1; d < BLOCK; d <<= 1) {               // Hillis-Steele over the threads' totals
        const double add = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    const double before = threadIdx.x ? sh[threadIdx.x - 1] : 0.0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] = before + v[k];
    if (threadIdx.x == BLOCK - 1) tile_sum[blockIdx.x] = sh[BLOCK - 1];
}

// out[order[r]] = scanned[r] + offset of r's tile: the enclosed mass of every particle, in the particles' own order
__global__ void k_scan_scatter(const double *__restrict__ scanned, const double *__restrict__ tile_before, const uint32_t *__restrict__ order,
                               double *__restrict__ out, int64_t n)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r < n) out[order[r]] = scanned[r] + tile_before[r / SCAN_TILE];
}

__global__ void k_gather_by(const double *__restrict__ in, const uint32_t *__restrict__ order, double *__restrict__ out, int64_t n)
{
    const int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (r < n) out[r] = in[order[r]];
}

// ---- the deviates ---------------------------------------------------------------------------------------------------
// Standard normal deviates of particle i for `purpose`: Box-Muller on the four words of one Philox block gives two
// pairs; words 0..2 of the result are used.
__global__ void k_normals(double *__restrict__ out, int64_t n, int per, uint32_t purpose, uint64_t seed)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const pmrand::draw_t draw{(uint32_t)i, (uint32_t)((uint64_t)i >> 32), (uint32_t)seed, (uint32_t)(seed >> 32)};
    const pmrand::u32x4 b = draw(purpose, 0);
    const double m0 = sqrt(-2.0 * log(pmrand::unit(b.v[0]))), a0 = TWO_PI * pmrand::unit(b.v[1]);
    const double m1 = sqrt(-2.0 * log(pmrand::unit(b.v[2]))), a1 = TWO_PI * pmrand::unit(b.v[3]);
    const double z[3] = {m0 * cos(a0), m0 * sin(a0), m1 * cos(a1)};
    for (int d = 0; d < per; ++d) out[i * per + d] = z[d];
}

// ---- ellipse --------------------------------------------------------------------------------------------------------
struct gal_params { double a, b, c, M_tot, disp, cbrt_abc, R, R_dm, M_star, M_dark; int64_t n, n_star; };

// positions = 0 + sigma * deviate (Generator.normal(loc, scale), galaxy_generation_functions.py:215) and the logarithm
// of the importance weight, log q - log p = -m^2/2 + m (:222-230)
__global__ void k_ellipse_positions(const double *__restrict__ z, double *__restrict__ pos, double *__restrict__ logw, gal_params q)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n) return;
    const double x = 0.0 + q.a * z[3 * i], y = 0.0 + q.b * z[3 * i + 1], w = 0.0 + q.c * z[3 * i + 2];
    pos[3 * i] = x; pos[3 * i + 1] = y; pos[3 * i + 2] = w;
    const double m2 = x * (1.0 / (q.a * q.a)) * x + y * (1.0 / (q.b * q.b)) * y + w * (1.0 / (q.c * q.c)) * w;
    logw[i] = -0.5 * m2 - (-sqrt(m2));
}

// w = exp(logw - max) over [first, last)
__global__ void k_exp_shift(double *__restrict__ logw, double top, int64_t first, int64_t last)
{
    const int64_t i = first + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < last) logw[i] = exp(logw[i] - top);
}

// masses = M * (w / sum) (:233-236) or M * w / sum (:313, 322)
__global__ void k_masses(const double *__restrict__ w, double *__restrict__ mass, double M, double sum, bool quotient_first, int64_t first, int64_t last)
{
    const int64_t i = first + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < last) mass[i] = quotient_first ? M * (w[i] / sum) : M * w[i] / sum;
}

// virial_ellipsoid, :29-37: positions about the centre of mass, ellipsoidal radius m, r_eff = m (abc)^(1/3); the
// sort key is r_eff's bit pattern (positive doubles order like their bits)
__global__ void k_virial_radius(const double *__restrict__ pos, double cx, double cy, double cz, double *__restrict__ x, double *__restrict__ r,
                                uint64_t *__restrict__ rkey, gal_params q)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n) return;
    const double a = pos[3 * i] - cx, b = pos[3 * i + 1] - cy, c = pos[3 * i + 2] - cz;
    x[3 * i] = a; x[3 * i + 1] = b; x[3 * i + 2] = c;
    const double s = a * a * (1.0 / (q.a * q.a)) + b * b * (1.0 / (q.b * q.b)) + c * c * (1.0 / (q.c * q.c));
    const double m = sqrt(fmax(s, EPS));
    const double rr = fmax(m * q.cbrt_abc, EPS);
    r[i] = rr;
    rkey[i] = (uint64_t)__double_as_longlong(rr);
}

// :44-86: circular speed, shell normal, tangent basis about the z axis, mean tangential motion plus in-plane
// dispersion, the radial part projected out.  vc2[i] = v_c^2 for the final rescale.
__global__ void k_virial_velocity(const double *__restrict__ x, const double *__restrict__ r, const double *__restrict__ Menc,
                                  const double *__restrict__ z1, const double *__restrict__ z2, double *__restrict__ vel,
                                  double *__restrict__ vc2, gal_params q)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n) return;
    const double v_c = sqrt(1.0 * Menc[i] / r[i]);
    double n0 = x[3 * i] * (1.0 / (q.a * q.a)), n1 = x[3 * i + 1] * (1.0 / (q.b * q.b)), n2 = x[3 * i + 2] * (1.0 / (q.c * q.c));
    const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2) + EPS;
    n0 /= nn; n1 /= nn; n2 /= nn;
    const double kz = 1.0 / (1.0 + EPS);                // k = (0, 0, 1) / (|k| + eps)
    // t1 = k x n
    double t0 = 0.0 * n2 - kz * n1, t1 = kz * n0 - 0.0 * n2, t2 = 0.0 * n1 - 0.0 * n0;
    double tn = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    if (tn < 1e-8) {                                    // k ~ n: the alternate axis (0, 1, 0), :64-70
        double u0 = 1.0 * n2 - 0.0 * n1, u1 = 0.0 * n0 - 0.0 * n2, u2 = 0.0 * n1 - 1.0 * n0;
        const double un = sqrt(u0 * u0 + u1 * u1 + u2 * u2) + EPS;
        t0 = u0 / un; t1 = u1 / un; t2 = u2 / un;
        tn = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    }
    t0 /= tn + EPS; t1 /= tn + EPS; t2 /= tn + EPS;
    // t2 = n x t1
    const double s0 = n1 * t2 - n2 * t1, s1 = n2 * t0 - n0 * t2, s2 = n0 * t1 - n1 * t0;
    const double d = q.disp * v_c, a1 = z1[i], a2 = z2[i];
    double v0 = t0 * v_c + (t0 * a1 + s0 * a2) * d, v1 = t1 * v_c + (t1 * a1 + s1 * a2) * d, v2 = t2 * v_c + (t2 * a1 + s2 * a2) * d;
    const double radial = v0 * n0 + v1 * n1 + v2 * n2;
    v0 -= n0 * radial; v1 -= n1 * radial; v2 -= n2 * radial;
    vel[3 * i] = v0; vel[3 * i + 1] = v1; vel[3 * i + 2] = v2;
    vc2[i] = v_c * v_c;
}

// v -= mean; speed2[i] = |v|^2 afterwards
__global__ void k_shift_velocity(double *__restrict__ vel, double mx, double my, double mz, double *__restrict__ speed2, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double a = vel[3 * i] - mx, b = vel[3 * i + 1] - my, c = vel[3 * i + 2] - mz;
    vel[3 * i] = a; vel[3 * i + 1] = b; vel[3 * i + 2] = c;
    speed2[i] = a * a + b * b + c * c;
}

__global__ void k_scale(double *__restrict__ v, double f, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < count) v[i] *= f;
}

// ---- diffuse sphere -------------------------------------------------------------------------------------------------
// :302-322: stars from N(0, R^2), dark matter from N(0, R_dm^2); radius, and the logarithm of the weight: Plummer against
// the Gaussian for the stars, NFW against the Gaussian for the dark matter
__global__ void k_sphere_positions(const double *__restrict__ z, double *__restrict__ pos, double *__restrict__ r, uint64_t *__restrict__ rkey,
                                   double *__restrict__ logw, gal_params q)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= q.n) return;
    const bool star = i < q.n_star;
    const double scale = star ? q.R : q.R_dm;
    const double x = 0.0 + scale * z[3 * i], y = 0.0 + scale * z[3 * i + 1], w = 0.0 + scale * z[3 * i + 2];
    pos[3 * i] = x; pos[3 * i + 1] = y; pos[3 * i + 2] = w;
    const double rr = sqrt(x * x + y * y + w * w);
    r[i] = rr;
    rkey[i] = (uint64_t)__double_as_longlong(rr);
    if (star) {
        const double s = rr / q.R;
        logw[i] = -2.5 * log1p(s * s) + 0.5 * (s * s);
    } else {
        const double s = fmax(rr / q.R_dm, 1e-12);
        logw[i] = -log(s) - 2.0 * log1p(s) + 0.5 * (s * s);
    }
}

// :336-381: circular speed from the enclosed mass, direction = r_hat x q (q a fresh normal vector), (1, 0, 0) in its
// place when the two are parallel; r = 0 has no direction and no speed
__global__ void k_sphere_velocity(const double *__restrict__ pos, const double *__restrict__ r, const double *__restrict__ Menc,
                                  const double *__restrict__ qn, double *__restrict__ vel, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const double rr = r[i];
    if (!(rr > 0.0)) { vel[3 * i] = 0.0; vel[3 * i + 1] = 0.0; vel[3 * i + 2] = 0.0; return; }
    const double vmag = sqrt(Menc[i] / rr);
    const double h0 = pos[3 * i] / rr, h1 = pos[3 * i + 1] / rr, h2 = pos[3 * i + 2] / rr;
    double q0 = qn[3 * i], q1 = qn[3 * i + 1], q2 = qn[3 * i + 2];
    double t0 = h1 * q2 - h2 * q1, t1 = h2 * q0 - h0 * q2, t2 = h0 * q1 - h1 * q0;
    double tn = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    if (tn < 1e-12) {
        q0 = 1.0; q1 = 0.0; q2 = 0.0;
        t0 = h1 * q2 - h2 * q1; t1 = h2 * q0 - h0 * q2; t2 = h0 * q1 - h1 * q0;
        tn = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
    }
    vel[3 * i] = t0 / tn * vmag; vel[3 * i + 1] = t1 / tn * vmag; vel[3 * i + 2] = t2 / tn * vmag;
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct workspace {
    int device = 0;
    hipStream_t s = nullptr;
    std::vector<void *> owned;
    double *partial = nullptr;
    std::string err;
    hipError_t e = hipSuccess;

    template <typename T> T *alloc(size_t count)
    {
        void *p = nullptr;
        if (e == hipSuccess) e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return nullptr;
        owned.push_back(p);
        return (T *)p;
    }
    ~workspace()
    {
        for (void *p : owned) hipFree(p);
        if (s) hipStreamDestroy(s);
    }
    bool ok() const { return e == hipSuccess; }
    void check() { if (e == hipSuccess) e = hipGetLastError(); }

    template <int OP> double reduce(const double *in, int stride, const double *weight, int64_t first, int64_t last)
    {
        double host[PARTIALS];
        if (!ok()) return 0.0;
        k_reduce<OP><<<PARTIALS, BLOCK, 0, s>>>(in, stride, weight, first, last, partial);
        check();
        if (ok()) e = hipMemcpyAsync(host, partial, sizeof(host), hipMemcpyDeviceToHost, s);
        if (ok()) e = hipStreamSynchronize(s);
        if (!ok()) return 0.0;
        double acc = host[0];
        for (int k = 1; k < PARTIALS; ++k) acc = OP == OP_SUM ? acc + host[k] : std::fmax(acc, host[k]);
        return acc;
    }
    double sum(const double *in, int stride, const double *weight, int64_t n) { return reduce<OP_SUM>(in, stride, weight, 0, n); }
};

// Menc[i] = total mass of the particles whose radius is not larger than particle i's (itself included)
void enclosed_mass(workspace &w, const uint64_t *rkey, const double *mass, double *Menc, int64_t n)
{
    const size_t tmp_bytes = pm_sort_outbox_tmp_bytes(n);
    void *tmp = w.alloc<char>(tmp_bytes + 16);
    uint64_t *sorted = w.alloc<uint64_t>(n);
    uint32_t *order = w.alloc<uint32_t>(n);
    double *m_sorted = w.alloc<double>(n), *scanned = w.alloc<double>(n);
    const int64_t tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    double *tile_sum = w.alloc<double>(tiles);
    if (!w.ok()) return;
    w.e = pm_sort_u64(tmp, tmp_bytes, rkey, sorted, order, (size_t)n, w.device, w.s);
    if (!w.ok()) return;
    k_gather_by<<<grid_for(n), BLOCK, 0, w.s>>>(mass, order, m_sorted, n);
    k_scan_tiles<<<dim3((unsigned)tiles), BLOCK, 0, w.s>>>(m_sorted, scanned, tile_sum, n);
    w.check();
    std::vector<double> host(tiles);
    if (w.ok()) w.e = hipMemcpyAsync(host.data(), tile_sum, tiles * sizeof(double), hipMemcpyDeviceToHost, w.s);
    if (w.ok()) w.e = hipStreamSynchronize(w.s);
    if (!w.ok()) return;
    double run = 0.0;
    for (int64_t t = 0; t < tiles; ++t) { const double v = host[t]; host[t] = run; run += v; }
    w.e = hipMemcpyAsync(tile_sum, host.data(), tiles * sizeof(double), hipMemcpyHostToDevice, w.s);
    if (!w.ok()) return;
    k_scan_scatter<<<grid_for(n), BLOCK, 0, w.s>>>(scanned, tile_sum, order, Menc, n);
    w.check();
    if (w.ok()) w.e = hipStreamSynchronize(w.s);     // (host[] goes out of scope)
}

// the deviates: from the caller (host, `count` doubles at `from`) or drawn here
double *deviates(workspace &w, const double *host, int64_t n, int per, uint32_t purpose, uint64_t seed)
{
    double *d = w.alloc<double>((size_t)n * per);
    if (!w.ok()) return nullptr;
    if (host) w.e = hipMemcpyAsync(d, host, (size_t)n * per * sizeof(double), hipMemcpyHostToDevice, w.s);
    else { k_normals<<<grid_for(n), BLOCK, 0, w.s>>>(d, n, per, purpose, seed); w.check(); }
    return d;
}

}  // namespace

extern "C" int pm_generate_galaxy(int device, int kind, int64_t n, const double *params, int nparams, uint64_t seed, const double *normals,
                                  double *positions, double *velocities, double *masses)
{
    auto fail = [](int code, const std::string &msg) { pm_set_create_error(msg); return code; };
    const int want = kind == PM_GALAXY_ELLIPSE ? 5 : kind == PM_GALAXY_DIFFUSE_SPHERE ? 4 : -1;
    if (want < 0) return fail(PM_EINVAL, "pm_generate_galaxy: unknown kind (PM_GALAXY_ELLIPSE or PM_GALAXY_DIFFUSE_SPHERE)");
    if (!params || nparams != want) return fail(PM_EINVAL, "pm_generate_galaxy: this kind takes " + std::to_string(want) + " parameters");
    if (n < 1 || n >= ((int64_t)1 << 31)) return fail(PM_EINVAL, "pm_generate_galaxy: n must be in [1, 2^31)");
    if (!positions || !velocities || !masses) return fail(PM_EINVAL, "pm_generate_galaxy: positions, velocities and masses must not be NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(PM_EHIP, "pm_generate_galaxy: no such HIP device");
    workspace w;
    w.device = device;
    w.e = hipSetDevice(device);
    if (w.ok()) w.e = hipStreamCreateWithFlags(&w.s, hipStreamNonBlocking);
    w.partial = w.alloc<double>(PARTIALS);
    gal_params q{};
    q.n = n;
    double *pos = w.alloc<double>(3 * n), *vel = w.alloc<double>(3 * n), *mass = w.alloc<double>(n), *wgt = w.alloc<double>(n);
    double *r = w.alloc<double>(n), *Menc = w.alloc<double>(n);
    uint64_t *rkey = w.alloc<uint64_t>(n);
    if (kind == PM_GALAXY_ELLIPSE) {
        q.a = params[0]; q.b = params[1]; q.c = params[2]; q.M_tot = params[3]; q.disp = params[4];
        if (!(q.a > 0.0) || !(q.b > 0.0) || !(q.c > 0.0)) return fail(PM_EINVAL, "pm_generate_galaxy: the axes a, b, c must be positive");
        q.cbrt_abc = std::pow(std::fabs(q.a * q.b * q.c), 1.0 / 3.0);
        // deviates: (n,3) positions, then z1 (n), then z2 (n)
        double *zp = deviates(w, normals, n, 3, 0, seed);
        double *z1 = deviates(w, normals ? normals + 3 * n : nullptr, n, 1, 1, seed);
        double *z2 = deviates(w, normals ? normals + 4 * n : nullptr, n, 1, 2, seed);
        double *x = w.alloc<double>(3 * n), *vc2 = w.alloc<double>(n);
        if (!w.ok()) return fail(PM_EHIP, std::string("pm_generate_galaxy: ") + hipGetErrorString(w.e));
        k_ellipse_positions<<<grid_for(n), BLOCK, 0, w.s>>>(zp, pos, wgt, q);
        const double top = w.reduce<OP_MAX>(wgt, 1, nullptr, 0, n);
        k_exp_shift<<<grid_for(n), BLOCK, 0, w.s>>>(wgt, top, 0, n);
        const double total = w.sum(wgt, 1, nullptr, n);
        // w /= sum(w); masses = M_tot * w   (:233-236)
        k_masses<<<grid_for(n), BLOCK, 0, w.s>>>(wgt, mass, q.M_tot, total, true, 0, n);
        // virial_ellipsoid (:29): np.average(pos, weights=mass)
        const double msum = w.sum(mass, 1, nullptr, n);
        const double cx = w.sum(pos, 3, mass, n) / msum, cy = w.sum(pos + 1, 3, mass, n) / msum, cz = w.sum(pos + 2, 3, mass, n) / msum;
        k_virial_radius<<<grid_for(n), BLOCK, 0, w.s>>>(pos, cx, cy, cz, x, r, rkey, q);
        w.check();
        enclosed_mass(w, rkey, mass, Menc, n);
        if (w.ok()) k_virial_velocity<<<grid_for(n), BLOCK, 0, w.s>>>(x, r, Menc, z1, z2, vel, vc2, q);
        // bulk drift out (:89), kinetic energy rescaled to sum m v_c^2 (:92-95)
        const double mx = w.sum(vel, 3, mass, n) / msum, my = w.sum(vel + 1, 3, mass, n) / msum, mz = w.sum(vel + 2, 3, mass, n) / msum;
        if (w.ok()) k_shift_velocity<<<grid_for(n), BLOCK, 0, w.s>>>(vel, mx, my, mz, wgt, n);
        const double num = w.sum(vc2, 1, mass, n), den = w.sum(wgt, 1, mass, n);
        if (w.ok() && den > EPS) k_scale<<<grid_for(3 * n), BLOCK, 0, w.s>>>(vel, std::sqrt(num / den), 3 * n);
    } else {
        q.R = params[0]; q.M_tot = params[1];
        const double stellar_fraction = params[2];
        q.R_dm = params[3];
        if (!(q.R > 0.0) || !(q.R_dm > 0.0)) return fail(PM_EINVAL, "pm_generate_galaxy: the scale lengths R and R_dm must be positive");
        q.M_star = stellar_fraction * q.M_tot; q.M_dark = (1 - stellar_fraction) * q.M_tot;
        q.n_star = (int64_t)(0.5 * (double)n);
        // deviates: (n,3) positions (stars first), then (n,3) directions
        double *zp = deviates(w, normals, n, 3, 0, seed);
        double *zq = deviates(w, normals ? normals + 3 * n : nullptr, n, 3, 1, seed);
        if (!w.ok()) return fail(PM_EHIP, std::string("pm_generate_galaxy: ") + hipGetErrorString(w.e));
        k_sphere_positions<<<grid_for(n), BLOCK, 0, w.s>>>(zp, pos, r, rkey, wgt, q);
        const int64_t cut[3] = {0, q.n_star, n};
        const double M[2] = {q.M_star, q.M_dark};
        for (int part = 0; part < 2; ++part) {
            const int64_t first = cut[part], last = cut[part + 1];
            if (first == last) continue;
            const double top = w.reduce<OP_MAX>(wgt, 1, nullptr, first, last);
            k_exp_shift<<<grid_for(last - first), BLOCK, 0, w.s>>>(wgt, top, first, last);
            const double total = w.reduce<OP_SUM>(wgt, 1, nullptr, first, last);
            k_masses<<<grid_for(last - first), BLOCK, 0, w.s>>>(wgt, mass, M[part], total, false, first, last);
        }
        w.check();
        enclosed_mass(w, rkey, mass, Menc, n);
        if (w.ok()) k_sphere_velocity<<<grid_for(n), BLOCK, 0, w.s>>>(pos, r, Menc, zq, vel, n);
    }
    w.check();
    if (w.ok()) w.e = hipMemcpyAsync(positions, pos, 3 * n * sizeof(double), hipMemcpyDeviceToHost, w.s);
    if (w.ok()) w.e = hipMemcpyAsync(velocities, vel, 3 * n * sizeof(double), hipMemcpyDeviceToHost, w.s);
    if (w.ok()) w.e = hipMemcpyAsync(masses, mass, n * sizeof(double), hipMemcpyDeviceToHost, w.s);
    if (w.ok()) w.e = hipStreamSynchronize(w.s);
    if (!w.ok()) return fail(PM_EHIP, std::string("pm_generate_galaxy: ") + hipGetErrorString(w.e));
    return PM_OK;
}

// Tuning probe: how fast can workgroups bring stretches of mesh rows into LDS?  One workgroup per stretch of J rows of a
// plane of a G^3 float mesh: (4 J + 8) rows of G floats from four neighbouring planes, 16 bytes per thread and turn,
// then a barrier and a token read.  hipcc --offload-arch=gfx950 -O3 -o fill_probe fill_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int J, int T, int DEPTH>
__global__ void __launch_bounds__(T) k_fill(const float *__restrict__ phi, float *__restrict__ out, int G, int tpp)
{
    extern __shared__ __align__(16) float tile[];
    constexpr int ROWS = 4 * J + 8;
    const int quads = G / 4, total = ROWS * quads;
    const int t = blockIdx.x, il = t / tpp, j0 = (t % tpp) * J;
    const int i = il < 1 ? 1 : il > G - 3 ? G - 3 : il;
    for (int base = 0; base < total; base += T * DEPTH) {
        float4 part[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int q = base + d * T + (int)threadIdx.x;
            part[d] = make_float4(0, 0, 0, 0);
            if (q < total) {
                const int row = q / quads, col = q % quads;
                int pa, jr;
                if (row < 2 * (J + 3)) { const int a = row >= J + 3; pa = i + a; jr = j0 - 1 + row - a * (J + 3); }
                else { const int rr = row - 2 * (J + 3), o = rr >= J + 1; pa = o ? i + 2 : i - 1; jr = j0 + rr - o * (J + 1); }
                jr = jr < 0 ? 0 : jr >= G ? G - 1 : jr;
                part[d] = *(const float4 *)(phi + ((size_t)pa * G + jr) * G + 4 * col);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int q = base + d * T + (int)threadIdx.x;
            if (q < total) ((float4 *)tile)[q] = part[d];
        }
    }
    __syncthreads();
    float s = 0;
    for (int k = 0; k < 8; ++k) s += tile[(threadIdx.x * 37 + k * 1031) % (ROWS * G)];
    if (s == 12345.678f) out[t] = s;
}

template <int J, int T, int DEPTH> void run(const float *phi, float *out, int G, const char *tag)
{
    const int tpp = (G + J - 1) / J, tiles = G * tpp;
    const size_t lds = (size_t)(4 * J + 8) * G * 4;
    hipFuncSetAttribute((const void *)k_fill<J, T, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_fill<J, T, DEPTH>, T, lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_fill<J, T, DEPTH><<<tiles, T, lds>>>(phi, out, G, tpp);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) k_fill<J, T, DEPTH><<<tiles, T, lds>>>(phi, out, G, tpp);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-14s J=%2d T=%4d depth=%2d lds=%6zu occ/CU=%d tiles=%6d  %.3f ms  %.2f TB/s into LDS  (%s)\n", tag, J, T, DEPTH, lds, occ, tiles, ms,
           (double)tiles * lds / ms / 1e9, hipGetErrorString(hipGetLastError()));
}

int main()
{
    const int G = 512;
    float *phi, *out;
    hipMalloc(&phi, (size_t)G * G * G * 4); hipMemset(phi, 0, (size_t)G * G * G * 4);
    hipMalloc(&out, 1 << 22);
    run<8, 640, 10>(phi, out, G, "as built");
    run<8, 640, 5>(phi, out, G, "half depth");
    run<8, 1024, 5>(phi, out, G, "1024 thr");
    run<8, 256, 10>(phi, out, G, "256 thr");
    run<6, 448, 10>(phi, out, G, "J6");
    run<4, 320, 10>(phi, out, G, "J4");
    run<4, 256, 6>(phi, out, G, "J4 256");
    run<2, 192, 11>(phi, out, G, "J2");
    run<2, 128, 8>(phi, out, G, "J2 128");
    run<1, 64, 12>(phi, out, G, "J1 wave");
    return 0;
}

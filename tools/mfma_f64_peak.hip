// mfma_f64_peak.hip — micro-benchmark: sustained issue interval of v_mfma_f64_16x16x4_f64 on
// gfx950, measured IN-KERNEL with the shader cycle counter (s_memtime) and the constant
// 100 MHz wall clock, so that the result separates "cycles per MFMA" from the DVFS clock.
// Exactly one workgroup per CU is forced by a 160 KiB dynamic-LDS request; waves per SIMD =
// blockDim / 256.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ void k(double* out, const double* in, int iters, long long* cyc, long long* wall) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64 * 8; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    d4 acc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = d4{0, 0, 0, 0};
    double a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
    double v = 1.0;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) {
            if (MODE == 1) {  // two LDS operand reads per MFMA (like fsnap_syrk_lds)
                a = lds[((it + u) & 7) * 512 + lane];
                b = lds[((it + u + 3) & 7) * 512 + 64 + lane];
            }
            if (MODE == 2) v = __builtin_fma(v, 1.0000001, a);  // one fp64 VALU fma per MFMA
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = v;
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {
        const int wid = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        cyc[wid] = t1 - t0;
        wall[wid] = w1 - w0;
    }
}

template <int NACC, int MODE>
void run(int waves_per_simd, const char* tag, double* out, double* in, long long* dcyc, long long* dwall) {
    const int iters = 3000;
    dim3 grid(256), block(256 * waves_per_simd);
    const size_t lds = 160 * 1024 - 1024;
    hipFuncSetAttribute((const void*)k<NACC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), grid, block, lds, 0, out, in, 10, dcyc, dwall);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, MODE>), grid, block, lds, 0, out, in, iters, dcyc, dwall);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const int nw = 256 * 4 * waves_per_simd;
    std::vector<long long> c(nw), w(nw);
    (void)hipMemcpy(c.data(), dcyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipMemcpy(w.data(), dwall, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double csum = 0, wsum = 0;
    for (int i = 0; i < nw; ++i) { csum += c[i]; wsum += w[i]; }
    const double per_wave_mfma = (double)iters * NACC;
    const double cyc_per_mfma_simd = (csum / nw) / (per_wave_mfma * waves_per_simd);  // pipe interval per SIMD
    const double ghz = (csum / nw) / ((wsum / nw) / 100e6) / 1e9;
    const double nmfma = (double)nw * per_wave_mfma;
    const double tf = nmfma * 2048.0 / (best * 1e-3) / 1e12;
    printf("%-24s nacc=%2d waves/SIMD=%d  %.3f ms  %.1f TF/s  %.1f shader-cycles/MFMA/SIMD  clock %.2f GHz\n", tag, NACC,
           waves_per_simd, best, tf, cyc_per_mfma_simd, ghz);
}

int main() {
    double *out, *in;
    long long *dcyc, *dwall;
    (void)hipMalloc(&out, 256 * 1024 * sizeof(double));
    (void)hipMalloc(&in, 4096 * sizeof(double));
    (void)hipMalloc(&dcyc, 4096 * sizeof(long long));
    (void)hipMalloc(&dwall, 4096 * sizeof(long long));
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1e-3 * ((i * 37) % 97) - 0.04;
    (void)hipMemcpy(in, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
    run<4, 0>(1, "mfma only", out, in, dcyc, dwall);
    run<8, 0>(1, "mfma only", out, in, dcyc, dwall);
    run<18, 0>(1, "mfma only", out, in, dcyc, dwall);
    run<8, 0>(2, "mfma only", out, in, dcyc, dwall);
    run<18, 0>(2, "mfma only", out, in, dcyc, dwall);
    run<5, 0>(4, "mfma only", out, in, dcyc, dwall);
    run<8, 0>(4, "mfma only", out, in, dcyc, dwall);
    run<5, 1>(4, "mfma + 2 ds_read_b64", out, in, dcyc, dwall);
    run<18, 1>(2, "mfma + 2 ds_read_b64", out, in, dcyc, dwall);
    run<5, 2>(4, "mfma + 1 v_fma_f64", out, in, dcyc, dwall);
    run<18, 2>(2, "mfma + 1 v_fma_f64", out, in, dcyc, dwall);
    return 0;
}

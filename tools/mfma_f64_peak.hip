// mfma_f64_peak.hip — micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 on gfx950
// (how close to the 78.6 TF vendor peak can ANY instruction stream get, and what do
// interleaved LDS reads / fp64 VALU cost).  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak tools/mfma_f64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(double* out, const double* in, int iters) {
    __shared__ double lds[8 * 64 * 8];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 64 * 8; i += blockDim.x) lds[i] = in[i & 1023];
    __syncthreads();
    d4 acc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = d4{0, 0, 0, 0};
    double a = in[threadIdx.x], b = in[threadIdx.x + 256];
    double v = 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) {
            if (MODE == 1) {  // two LDS operand reads per MFMA (like fsnap_syrk_lds)
                a = lds[((it + u) & 7) * 512 + lane];
                b = lds[((it + u + 3) & 7) * 512 + 64 + lane];
            }
            if (MODE == 2) {  // one fp64 VALU fma per MFMA
                v = __builtin_fma(v, 1.0000001, a);
            }
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
    }
    double s = v;
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(int wg_per_cu, const char* tag, double* out, double* in) {
    const int iters = 4000;
    dim3 grid(256 * wg_per_cu), block(256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), grid, block, 0, 0, out, in, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, MODE>), grid, block, 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double nmfma = (double)grid.x * 4 * iters * NACC;
    const double tf = nmfma * 2048.0 / (best * 1e-3) / 1e12;
    // cycles per MFMA per SIMD if the clock were 2.4 GHz
    const double cyc24 = (best * 1e-3) * 2.4e9 / (nmfma / 1024.0);
    printf("%-28s nacc=%2d waves/SIMD=%d  %.3f ms  %.1f TF/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", tag, NACC, wg_per_cu, best, tf, cyc24);
}

int main() {
    double *out, *in;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    hipMalloc(&in, 4096 * sizeof(double));
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 1e-3 * (i % 97) - 0.04;
    hipMemcpy(in, h.data(), 4096 * sizeof(double), hipMemcpyHostToDevice);
    run<4, 0>(1, "mfma only", out, in);
    run<8, 0>(1, "mfma only", out, in);
    run<18, 0>(1, "mfma only", out, in);
    run<8, 0>(2, "mfma only", out, in);
    run<18, 0>(2, "mfma only", out, in);
    run<5, 0>(4, "mfma only", out, in);
    run<8, 0>(4, "mfma only", out, in);
    run<5, 1>(4, "mfma + 2 ds_read_b64", out, in);
    run<18, 1>(2, "mfma + 2 ds_read_b64", out, in);
    run<5, 2>(4, "mfma + 1 v_fma_f64", out, in);
    run<18, 2>(2, "mfma + 1 v_fma_f64", out, in);
    return 0;
}

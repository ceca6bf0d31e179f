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

typedef unsigned u4 __attribute__((ext_vector_type(4)));

// MODE 3: MFMA stream + an HBM stream issued by the SAME waves at kernel 1L's ratio (one 4-row x 128-column
// chunk = 4 x 16 B per lane per 36 MFMAs), consumed one iteration later by integer VALU only: no LDS, no
// barrier, no fp64 VALU.  Separates "what concurrent HBM traffic costs the matrix pipe" from kernel structure.
template <int NACC>
__global__ void kstream(double* out, const double* in, const u4* big, size_t big_vec, int iters, long long* cyc,
                        long long* wall) {
    const int lane = threadIdx.x & 63;
    d4 acc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) acc[u] = d4{0, 0, 0, 0};
    const double a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
    const size_t wid = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t nwave = (size_t)gridDim.x * (blockDim.x / 64);
    u4 x0, x1, x2, x3, sink = u4{0, 0, 0, 0};
    const unsigned nchunk = (unsigned)(big_vec / 256);   // 256 vectors of 16 B = one 4 KiB chunk per wave and iteration
    const unsigned step = (unsigned)(nwave % nchunk);
    unsigned chunk = __builtin_amdgcn_readfirstlane((unsigned)(wid % nchunk));
    size_t pos = (size_t)chunk * 256 + lane;
    x0 = __builtin_nontemporal_load(big + pos);
    x1 = __builtin_nontemporal_load(big + pos + 64);
    x2 = __builtin_nontemporal_load(big + pos + 128);
    x3 = __builtin_nontemporal_load(big + pos + 192);
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        sink ^= x0 ^ x1 ^ x2 ^ x3;
        chunk += step;
        if (chunk >= nchunk) chunk -= nchunk;
        pos = (size_t)chunk * 256 + lane;
        x0 = __builtin_nontemporal_load(big + pos);
        x1 = __builtin_nontemporal_load(big + pos + 64);
        x2 = __builtin_nontemporal_load(big + pos + 128);
        x3 = __builtin_nontemporal_load(big + pos + 192);
#pragma unroll
        for (int r = 0; r < 36 / NACC; ++r)
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    sink ^= x0 ^ x1 ^ x2 ^ x3;
    double s = (double)(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
#pragma unroll
    for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {
        cyc[wid] = t1 - t0;
        wall[wid] = w1 - w0;
    }
}

template <int NACC>
void run_stream(int waves_per_simd, bool with_loads, double* out, double* in, const u4* big, size_t big_vec, long long* dcyc,
                long long* dwall) {
    const int iters = 600;
    dim3 grid(256), block(256 * waves_per_simd);
    const size_t lds = 160 * 1024 - 1024;
    hipFuncSetAttribute((const void*)kstream<NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // without loads: a 64 KiB window that stays in L2 (same instruction stream, no HBM traffic)
    const size_t window = with_loads ? big_vec : 4096;
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kstream<NACC>), grid, block, lds, 0, out, in, big, window, iters, dcyc, dwall);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    const int nw = 256 * 4 * waves_per_simd;
    std::vector<long long> c(nw), w(nw);
    (void)hipMemcpy(c.data(), dcyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipMemcpy(w.data(), dwall, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double csum = 0, wsum = 0;
    for (int i = 0; i < nw; ++i) { csum += c[i]; wsum += w[i]; }
    const double nmfma = (double)nw * iters * 36.0;
    const double tf = nmfma * 2048.0 / (best * 1e-3) / 1e12;
    const double gbs = (double)nw * iters * 4096.0 / (best * 1e-3) / 1e9;
    printf("mfma + %-18s nacc=%2d waves/SIMD=%d  %.3f ms  %.1f TF/s  %.0f GB/s  in-loop wall %.3f ms  clock %.2f GHz\n",
           with_loads ? "HBM stream (4KiB/36)" : "L2-resident loads", NACC, waves_per_simd, best, tf, with_loads ? gbs : 0.0,
           (wsum / nw) / 100e6 * 1e3, (csum / nw) / ((wsum / nw) / 100e6) / 1e9);
}


// Step-by-step approach to the real kernel's per-chunk work (no LDS, no barrier): every wave streams its own
// 4 KiB chunks (like kstream) and feeds 36 MFMAs per chunk from
//   STEP 1: the loaded values themselves (fresh operand registers every chunk, no VALU),
//   STEP 2: + 8 v_mul_f64 (row weighting),  STEP 3: + 16 v_cndmask (row mask),
//   STEP 4: + 8 v_fma_f64 + 3 scalar updates (c = A^T W^2 b and the scalars).
typedef double d2 __attribute__((ext_vector_type(2)));
template <int STEP, int NA>
__device__ __forceinline__ void chunk_work(d4 (&acc)[NA], const u4 (&x)[4], double w, bool keep, double wb, double (&cacc)[8],
                                           double& bb, double& sb, double& cnt) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const d2 t = __builtin_bit_cast(d2, x[j]);
        v[2 * j] = t[0];
        v[2 * j + 1] = t[1];
    }
    if (STEP >= 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= w;
    }
    if (STEP >= 3) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = keep ? v[j] : 0.0;
    }
    if (STEP >= 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cacc[j] = __builtin_fma(v[j], wb, cacc[j]);
        bb = __builtin_fma(wb, wb, bb);
        sb += wb;
        cnt += keep ? 1.0 : 0.0;
    }
    // 36 tiles of the 8-block triangle on 5 rotating accumulators (register budget of 4 waves/SIMD, no spills)
    int t = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int q = p; q < 8; ++q) {
            acc[t % NA] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[p], v[q], acc[t % NA], 0, 0, 0);
            ++t;
        }
}

template <int STEP, int NA>
__global__ __launch_bounds__(NA > 9 ? 512 : 1024) void kstep(double* out, const double* in, const u4* big, size_t big_vec, int iters, long long* cyc,
                      long long* wall) {
    const int lane = threadIdx.x & 63;
    d4 acc[NA];
#pragma unroll
    for (int u = 0; u < NA; ++u) acc[u] = d4{0, 0, 0, 0};
    double cacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, bb = 0, sb = 0, cnt = 0;
    const double w = in[threadIdx.x & 255] + 1.5, wb = in[(threadIdx.x & 255) + 256];
    const bool keep = in[(threadIdx.x & 255) + 512] > -1.0;   // true, but not known at compile time
    const size_t wid = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const size_t nwave = (size_t)gridDim.x * (blockDim.x / 64);
    const unsigned nchunk = (unsigned)(big_vec / 256);
    const unsigned step = (unsigned)(nwave % nchunk);
    unsigned chunk = __builtin_amdgcn_readfirstlane((unsigned)(wid % nchunk));
    u4 x[4], y[4];
    auto issue = [&](u4 (&r)[4]) {
        const size_t pos = (size_t)chunk * 256 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = __builtin_nontemporal_load(big + pos + 64 * j);
        chunk += step;
        if (chunk >= nchunk) chunk -= nchunk;
    };
    issue(x);
    issue(y);
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; it += 2) {
        chunk_work<STEP, NA>(acc, x, w, keep, wb, cacc, bb, sb, cnt);
        issue(x);
        chunk_work<STEP, NA>(acc, y, w, keep, wb, cacc, bb, sb, cnt);
        issue(y);
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    double s = bb + sb + cnt + (double)(x[0][0] ^ y[0][0]);
#pragma unroll
    for (int u = 0; u < NA; ++u) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) s += cacc[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) {
        cyc[wid] = t1 - t0;
        wall[wid] = w1 - w0;
    }
}

template <int STEP, int NA = 5>
void run_step(int waves_per_simd, double* out, double* in, const u4* big, size_t big_vec, long long* dcyc, long long* dwall) {
    const int iters = 600;
    dim3 grid(256), block(256 * waves_per_simd);
    const size_t lds = 160 * 1024 - 1024;
    hipFuncSetAttribute((const void*)kstep<STEP, NA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kstep<STEP, NA>), grid, block, lds, 0, out, in, big, big_vec, iters, dcyc, dwall);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    const int nw = 256 * 4 * waves_per_simd;
    std::vector<long long> c(nw), w(nw);
    (void)hipMemcpy(c.data(), dcyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
    (void)hipMemcpy(w.data(), dwall, nw * sizeof(long long), hipMemcpyDeviceToHost);
    double csum = 0, wsum = 0;
    for (int i = 0; i < nw; ++i) { csum += c[i]; wsum += w[i]; }
    const double nmfma = (double)nw * iters * 36.0;
    static const char* names[] = {"", "fresh operands", "+ 8 v_mul_f64", "+ 16 v_cndmask", "+ 8 v_fma_f64 + scalars"};
    printf("stream step %d nacc=%2d %-24s waves/SIMD=%d  %.3f ms  %.1f TF/s  %.0f GB/s  clock %.2f GHz\n", STEP, NA, names[STEP],
           waves_per_simd, best, nmfma * 2048.0 / (best * 1e-3) / 1e12, (double)nw * iters * 4096.0 / (best * 1e-3) / 1e9,
           (csum / nw) / ((wsum / nw) / 100e6) / 1e9);
}

// fill the streamed buffer with pseudo-random doubles in [-0.5, 0.5): the power drawn by the matrix pipe (and
// with it the clock the chip sustains) depends on how many operand bits toggle between MFMAs
__global__ void fill_random(double* p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        x ^= x >> 29;
        x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 32;
        p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
}

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
    int iv = threadIdx.x, iw = 7;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC; ++u) {
            if (MODE == 1) {  // two LDS operand reads per MFMA (like fsnap_syrk_lds)
                a = lds[((it + u) & 7) * 512 + lane];
                b = lds[((it + u + 3) & 7) * 512 + 64 + lane];
            }
            if (MODE == 2) v = __builtin_fma(v, 1.0000001, a);  // one fp64 VALU fma per MFMA
            if (MODE == 4) {   // one 32-bit integer VALU op per MFMA (the cndmask class)
                iv = iv * 3 + (int)u;
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 7) {   // two 32-bit integer VALU ops per MFMA
                iv = iv * 3 + (int)u;
                iw = iw ^ (iv >> 3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE == 5) {   // software-pipelined operands: the LDS reads for the NEXT MFMA are in flight during this one
                const double an = lds[((it + u) & 7) * 512 + lane];
                const double bn = lds[((it + u + 3) & 7) * 512 + 64 + lane];
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
                a = an;
                b = bn;
                continue;
            }
            if (MODE == 6) {   // one pipelined LDS operand read per MFMA (the other operand stays in a register)
                const double an = lds[((it + u) & 7) * 512 + lane];
                acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
                a = an;
                continue;
            }
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
        }
    }
    v += (double)(iv ^ iw);
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
    run<5, 4>(4, "mfma + 1 int VALU", out, in, dcyc, dwall);
    run<5, 7>(4, "mfma + 2 int VALU", out, in, dcyc, dwall);
    run<5, 5>(4, "mfma + 2 ds_read pipelined", out, in, dcyc, dwall);
    run<5, 6>(4, "mfma + 1 ds_read pipelined", out, in, dcyc, dwall);
    run<8, 4>(2, "mfma + 1 int VALU", out, in, dcyc, dwall);
    run<8, 5>(2, "mfma + 2 ds_read pipelined", out, in, dcyc, dwall);
    run<8, 6>(2, "mfma + 1 ds_read pipelined", out, in, dcyc, dwall);
    u4* big;
    const size_t big_bytes = (size_t)1 << 30;
    (void)hipMalloc(&big, big_bytes);
    (void)hipMemset(big, 1, big_bytes);
    for (int wps = 2; wps <= 4; wps += 2) {
        run_stream<9>(wps, false, out, in, big, big_bytes / 16, dcyc, dwall);
        run_stream<9>(wps, true, out, in, big, big_bytes / 16, dcyc, dwall);
        run_stream<4>(wps, false, out, in, big, big_bytes / 16, dcyc, dwall);
        run_stream<4>(wps, true, out, in, big, big_bytes / 16, dcyc, dwall);
    }
    printf("-- streamed data = constant bytes (0x01)\n");
    for (int wps = 2; wps <= 4; wps = wps * 2) {
        run_step<1>(wps, out, in, big, big_bytes / 16, dcyc, dwall);
        run_step<4>(wps, out, in, big, big_bytes / 16, dcyc, dwall);
    }
    fill_random<<<4096, 256>>>((double*)big, big_bytes / 8);
    (void)hipDeviceSynchronize();
    printf("-- streamed data = pseudo-random doubles in [-0.5, 0.5)\n");
    for (int wps = 1; wps <= 4; wps = wps * 2) {
        run_step<1>(wps, out, in, big, big_bytes / 16, dcyc, dwall);
        run_step<2>(wps, out, in, big, big_bytes / 16, dcyc, dwall);
        run_step<3>(wps, out, in, big, big_bytes / 16, dcyc, dwall);
        run_step<4>(wps, out, in, big, big_bytes / 16, dcyc, dwall);
    }
    run_step<4, 18>(1, out, in, big, big_bytes / 16, dcyc, dwall);
    run_step<4, 18>(2, out, in, big, big_bytes / 16, dcyc, dwall);
    run_step<1, 18>(2, out, in, big, big_bytes / 16, dcyc, dwall);
    run_step<4, 9>(2, out, in, big, big_bytes / 16, dcyc, dwall);
    run_step<4, 9>(4, out, in, big, big_bytes / 16, dcyc, dwall);
    return 0;
}

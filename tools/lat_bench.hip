// lat_bench.hip — single-wave instruction latencies on gfx950, in shader cycles (s_memtime) and wall-clock ticks.
// The serial kernels of the device Cholesky (fsnap_chol.hip) run ONE wave through long dependent chains; this tool
// measures what a link of such a chain costs: dependent / independent fp64 FMAs, v_rsq_f64, dependent fp64 MFMAs,
// an MFMA whose result is broadcast with v_readlane into the next MFMA's operand (the pivot pattern), selects.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/lat_bench tools/lat_bench.hip ; run: tools/lat_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

constexpr int N = 256;

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, const double* in, long long* cyc, long long* wall) {
    const int lane = threadIdx.x;
    double x = in[lane], y = in[lane + 64], z = in[lane + 128], w = in[lane + 192];
    d4 acc = {x, y, z, w}, acc2 = {y, z, w, x};
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    const long long t0 = clock64(), w0 = wall_clock64();
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x), "+v"(y), "+v"(z), "+v"(acc), "+v"(acc2)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 0) {          // dependent v_fma_f64
#pragma unroll
        for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, z);
    } else if constexpr (MODE == 1) {   // four independent v_fma_f64 chains
#pragma unroll
        for (int i = 0; i < N / 4; ++i) {
            x = __builtin_fma(x, y, z);
            acc[0] = __builtin_fma(acc[0], y, z);
            acc[1] = __builtin_fma(acc[1], y, z);
            acc[2] = __builtin_fma(acc[2], y, z);
        }
    } else if constexpr (MODE == 2) {   // dependent v_rsq_f64
#pragma unroll
        for (int i = 0; i < N; ++i) x = __builtin_amdgcn_rsq(x);
    } else if constexpr (MODE == 3) {   // dependent MFMA (accumulator chain)
#pragma unroll
        for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
    } else if constexpr (MODE == 4) {   // two independent MFMA chains
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, acc2, 0, 0, 0);
        }
    } else if constexpr (MODE == 5) {   // MFMA -> readlane of the result -> one multiply -> operand of the next MFMA
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double d = readlane_f64(acc[i & 3], (i * 17) & 63);
            const double a = y * d;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, z, acc, 0, 0, 0);
        }
    } else if constexpr (MODE == 6) {   // the whole pivot link: readlane, rsq + 3 Newton steps, scale, select, MFMA
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double d = readlane_f64(acc[i & 3], (i * 17) & 63);
            double r = __builtin_amdgcn_rsq(d);
            const double h = 0.5 * d;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const double e = __builtin_fma(-h * r, r, 0.5);
                r = __builtin_fma(r, e, r);
            }
            const double u = acc[i & 3] * r;
            const bool own = ((lane >> 4) == (i & 3));
            acc[i & 3] = own ? u : acc[i & 3];
            const double a = (own && (lane & 15) > (i & 15)) ? -u : 0.0;
            const double b = own ? u : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    } else if constexpr (MODE == 7) {   // dependent 64-bit selects
#pragma unroll
        for (int i = 0; i < N; ++i) x = ((lane + i) & 1) ? x : y + x;
    } else if constexpr (MODE == 8) {   // readlane -> VALU -> readlane chain
#pragma unroll
        for (int i = 0; i < N; ++i) x = x + readlane_f64(x, (i * 17) & 63);
    } else if constexpr (MODE == 9) {   // the pivot link with the MFMA replaced by nothing (VALU part alone)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double d = readlane_f64(x, (i * 17) & 63);
            double r = __builtin_amdgcn_rsq(d);
            const double h = 0.5 * d;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const double e = __builtin_fma(-h * r, r, 0.5);
                r = __builtin_fma(r, e, r);
            }
            x = x * r + y;
        }
    } else if constexpr (MODE == 10) {  // 1/sqrt from an f32 seed: cvt, v_rsq_f32, cvt, 2 Newton steps (24 -> 48 -> 96 bits)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double r = (double)__builtin_amdgcn_rsqf((float)x);
            const double h = 0.5 * x;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const double e = __builtin_fma(-h * r, r, 0.5);
                r = __builtin_fma(r, e, r);
            }
            x = r + y;
        }
    } else if constexpr (MODE == 11) {  // v_rsq_f64 + 2 Newton steps
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double r = __builtin_amdgcn_rsq(x);
            const double h = 0.5 * x;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const double e = __builtin_fma(-h * r, r, 0.5);
                r = __builtin_fma(r, e, r);
            }
            x = r + y;
        }
    } else if constexpr (MODE == 12) {  // v_rsq_f64 + 3 Newton steps (the chain of kernel 8b)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double r = __builtin_amdgcn_rsq(x);
            const double h = 0.5 * x;
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const double e = __builtin_fma(-h * r, r, 0.5);
                r = __builtin_fma(r, e, r);
            }
            x = r + y;
        }
    } else if constexpr (MODE == 13) {  // LDS round trip: ds_write_b64 -> ds_read_b64 of the same word -> add
        __shared__ double sh[64];
        typedef __attribute__((address_space(3))) double lds_f64;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            *(volatile lds_f64*)&sh[lane] = x;
            x = *(volatile lds_f64*)&sh[lane] + y;
        }
    } else if constexpr (MODE == 15 || MODE == 16 || MODE == 17) {
        // the owner's pivot loop of kernel 8b4 (side chain for the next pivot, one MFMA per pivot), 16 tiles of 16 pivots;
        // 16: with the LDS publication; 17: publication + write-back of the scaled row deferred behind the MFMA
        __shared__ double slot[16][16], invs[16], dummy[64];
        typedef __attribute__((address_space(3))) double lds_f64;
        const int e = lane & 15, kr = lane >> 4;
#pragma unroll 1
        for (int rep = 0; rep < N / 16; ++rep) {
            d4 D = acc;
            double dcur = readlane_f64(D[0], 0);
            double inv = __builtin_amdgcn_rsq(dcur);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int q = j >> 2, kk = j & 3;
                double t = 0.0, pn = 0.0;
                if (j < 15) {
                    const int q1 = (j + 1) >> 2, k1 = (j + 1) & 3;
                    t = readlane_f64(D[q], kk * 16 + j + 1);
                    pn = readlane_f64(D[q1], k1 * 16 + j + 1);
                }
                const bool own = (kr == kk);
                const double ud = D[q] * inv;
                const bool keep = own && e >= j;
                const double aop = (own && e > j) ? -ud : 0.0;
                if constexpr (MODE != 17) D[q] = keep ? ud : D[q];
                if constexpr (MODE == 16) {
                    *(volatile lds_f64*)(own ? &slot[j][e] : &dummy[lane]) = aop;
                    *(volatile lds_f64*)&invs[j] = inv;
                }
                if (j < 15) {
                    const double bop = keep ? ud : 0.0;
                    d4 Dn = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, D, 0, 0, 0);
                    if constexpr (MODE == 17) {
                        *(volatile lds_f64*)(own ? &slot[j][e] : &dummy[lane]) = aop;
                        *(volatile lds_f64*)&invs[j] = inv;
                        Dn[q] = keep ? ud : Dn[q];
                    }
                    D = Dn;
                    const double u = t * inv;
                    dcur = __builtin_fma(-u, u, pn);
                    double r = __builtin_amdgcn_rsq(dcur);
                    const double h = 0.5 * dcur;
#pragma unroll
                    for (int it = 0; it < 3; ++it) {
                        const double ee = __builtin_fma(-h * r, r, 0.5);
                        r = __builtin_fma(r, ee, r);
                    }
                    inv = r;
                }
            }
            acc[0] += D[0] * 1e-30 + 1.0;      // keeps the tiles apart (and the diagonal positive)
            acc[1] += D[1] * 1e-30;
        }
    } else if constexpr (MODE >= 18 && MODE <= 22) {
        // the rank-4 pivot group of kernel 8b4 and its parts: 18 whole group, 19 the ten v_readlane pairs alone, 20 the 4 x 4
        // factorisation + W on uniform values alone, 21 the operand selects + the two MFMAs alone, 22 whole group without readlanes
        const int e = lane & 15, kr = lane >> 4;
#pragma unroll 1
        for (int rep = 0; rep < N / 4; ++rep) {
            constexpr int Q = 1;
            d4 D = acc;
            double b00, b01, b02, b03, b11, b12, b13, b22, b23, b33;
            if constexpr (MODE == 18 || MODE == 19) {
                b00 = readlane_f64(D[Q], 4 * Q); b01 = readlane_f64(D[Q], 4 * Q + 1); b02 = readlane_f64(D[Q], 4 * Q + 2); b03 = readlane_f64(D[Q], 4 * Q + 3);
                b11 = readlane_f64(D[Q], 16 + 4 * Q + 1); b12 = readlane_f64(D[Q], 16 + 4 * Q + 2); b13 = readlane_f64(D[Q], 16 + 4 * Q + 3);
                b22 = readlane_f64(D[Q], 32 + 4 * Q + 2); b23 = readlane_f64(D[Q], 32 + 4 * Q + 3); b33 = readlane_f64(D[Q], 48 + 4 * Q + 3);
            } else {
                b00 = x + 4.0; b01 = y * 0.01; b02 = z * 0.01; b03 = w * 0.01; b11 = x + 5.0; b12 = y * 0.02; b13 = z * 0.02; b22 = x + 6.0; b23 = w * 0.02; b33 = x + 7.0;
            }
            if constexpr (MODE == 19) {
                acc[0] += ((b00 + b01) + (b02 + b03)) + ((b11 + b12) + (b13 + b22)) + (b23 + b33);
                continue;
            }
            double aW = y, un = z;
            if constexpr (MODE != 21) {
                auto rs = [](double d) { double r = __builtin_amdgcn_rsq(d); const double h = 0.5 * d;
                                         for (int it = 0; it < 2; ++it) { const double ee = __builtin_fma(-h * r, r, 0.5); r = __builtin_fma(r, ee, r); } return r; };
                const double i0 = rs(b00);
                const double u01 = b01 * i0, u02 = b02 * i0, u03 = b03 * i0;
                const double d1 = __builtin_fma(-u01, u01, b11);
                const double i1 = rs(d1);
                const double u12 = __builtin_fma(-u01, u02, b12) * i1, u13 = __builtin_fma(-u01, u03, b13) * i1;
                const double d2 = __builtin_fma(-u12, u12, __builtin_fma(-u02, u02, b22));
                const double i2 = rs(d2);
                const double u23 = __builtin_fma(-u12, u13, __builtin_fma(-u02, u03, b23)) * i2;
                const double d3 = __builtin_fma(-u23, u23, __builtin_fma(-u13, u13, __builtin_fma(-u03, u03, b33)));
                const double i3 = rs(d3);
                const double w00 = i0, w10 = -i1 * u01 * w00, w11 = i1, w20 = i2 * (-u02 * w00 - u12 * w10), w21 = -i2 * u12 * w11, w22 = i2;
                const double w30 = i3 * (-u03 * w00 - u13 * w10 - u23 * w20), w31 = i3 * (-u13 * w11 - u23 * w21), w32 = -i3 * u23 * w22, w33 = i3;
                if constexpr (MODE == 20) {
                    x = x * 0.5 + 1e-3 * (((w00 + w10) + (w11 + w20)) + ((w21 + w22) + (w30 + w31)) + (w32 + w33));
                    continue;
                }
                const int i = e - 4 * Q;
                const double c0 = i == 0 ? w00 : i == 1 ? w10 : i == 2 ? w20 : i == 3 ? w30 : 0.0;
                const double c1 = i == 1 ? w11 : i == 2 ? w21 : i == 3 ? w31 : 0.0;
                const double c2 = i == 2 ? w22 : i == 3 ? w32 : 0.0;
                const double c3 = i == 3 ? w33 : 0.0;
                aW = kr == 0 ? c0 : kr == 1 ? c1 : kr == 2 ? c2 : c3;
            }
            const d4 zero = {0.0, 0.0, 0.0, 0.0};
            const d4 Z0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aW, D[Q], zero, 0, 0, 0);
            un = Z0[Q];
            D[Q] = un;
            const double aU = (e > 4 * Q + 3) ? -un : 0.0;
            D = __builtin_amdgcn_mfma_f64_16x16x4f64(aU, un, D, 0, 0, 0);
            acc[0] = D[0] * 1e-30 + 4.0;
            acc[1] = D[1] * 1e-30 + 4.0 + 1e-3 * lane;
            acc[2] = D[2] * 1e-30 + 3.0;
            acc[3] = D[3] * 1e-30 + 5.0;
        }
    } else if constexpr (MODE == 14) {  // MFMA -> VALU read of the accumulator -> MFMA operand (no readlane)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double a = y * acc[i & 3];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, z, acc, 0, 0, 0);
        }
    }
    asm volatile("s_nop 0" : "+v"(x), "+v"(acc), "+v"(acc2)::"memory");     // the results exist here
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = clock64(), w1 = wall_clock64();
    __builtin_amdgcn_sched_barrier(0);
    out[lane] = x + acc[0] + acc[1] + acc[2] + acc[3] + acc2[0];
    if (lane == 0) {
        cyc[MODE] = t1 - t0;
        wall[MODE] = w1 - w0;
    }
}

int main() {
    double *in, *out;
    long long *cyc, *wall;
    hipMalloc(&in, 256 * 8);
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 32 * 8);
    hipMalloc(&wall, 32 * 8);
    double h[256];
    for (int i = 0; i < 256; ++i) h[i] = 1.0 + 1e-3 * i;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const char* names[] = {"dependent v_fma_f64", "4 independent v_fma_f64", "dependent v_rsq_f64", "dependent MFMA f64 16x16x4",
                           "2 independent MFMA chains", "MFMA -> readlane -> mul -> MFMA", "full pivot link (rsq + Newton + selects + MFMA)",
                           "dependent 64-bit select + add", "readlane -> add chain", "pivot link without the MFMA",
                           "rsq from f32 seed + 2 Newton + add", "v_rsq_f64 + 2 Newton + add", "v_rsq_f64 + 3 Newton + add",
                           "LDS write -> read -> add", "MFMA -> mul of the accumulator -> MFMA", "8b4 owner pivot (no LDS)", "8b4 owner pivot + LDS publication",
                           "8b4 owner pivot, publication + write-back behind the MFMA",
                           "rank-4 group (4 pivots): whole", "rank-4 group: the ten v_readlane pairs alone", "rank-4 group: 4 x 4 factorisation + W alone",
                           "rank-4 group: two MFMAs + masks alone", "rank-4 group without the readlanes"};
    for (int rep = 0; rep < 3; ++rep) {
        k<0><<<1, 64>>>(out, in, cyc, wall);
        k<1><<<1, 64>>>(out, in, cyc, wall);
        k<2><<<1, 64>>>(out, in, cyc, wall);
        k<3><<<1, 64>>>(out, in, cyc, wall);
        k<4><<<1, 64>>>(out, in, cyc, wall);
        k<5><<<1, 64>>>(out, in, cyc, wall);
        k<6><<<1, 64>>>(out, in, cyc, wall);
        k<7><<<1, 64>>>(out, in, cyc, wall);
        k<8><<<1, 64>>>(out, in, cyc, wall);
        k<9><<<1, 64>>>(out, in, cyc, wall);
        k<10><<<1, 64>>>(out, in, cyc, wall);
        k<11><<<1, 64>>>(out, in, cyc, wall);
        k<12><<<1, 64>>>(out, in, cyc, wall);
        k<13><<<1, 64>>>(out, in, cyc, wall);
        k<14><<<1, 64>>>(out, in, cyc, wall);
        k<15><<<1, 64>>>(out, in, cyc, wall);
        k<16><<<1, 64>>>(out, in, cyc, wall);
        k<17><<<1, 64>>>(out, in, cyc, wall);
        k<18><<<1, 64>>>(out, in, cyc, wall);
        k<19><<<1, 64>>>(out, in, cyc, wall);
        k<20><<<1, 64>>>(out, in, cyc, wall);
        k<21><<<1, 64>>>(out, in, cyc, wall);
        k<22><<<1, 64>>>(out, in, cyc, wall);
        hipDeviceSynchronize();
    }
    long long hc[32], hw[32];
    hipMemcpy(hc, cyc, sizeof hc, hipMemcpyDeviceToHost);
    hipMemcpy(hw, wall, sizeof hw, hipMemcpyDeviceToHost);
    for (int m = 0; m < 23; ++m)
        printf("%-60s %8.1f s_memtime ticks / link   %7.1f ns / link (100 MHz wall clock)\n", names[m], (double)hc[m] / N,
               (double)hw[m] * 10.0 / N);
    return 0;
}

// tools/weight_rows_variants.hip -- micro-benchmark behind the launch shape of fsnap_weight_rows_k (kernel 3):
// aw[i,:] = w[i] * A[i,:] over 10^6 x 128 fp64 (1.024 GB read + 1.024 GB written), variants of the cache policy
// (nontemporal loads / stores), rows per wave iteration, work distribution (grid-stride waves or one contiguous
// row range per workgroup) and grid size, next to a plain 16-byte copy of the same bytes (the practical HBM ceiling).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/weight_rows_variants.hip -o /tmp/wrv && /tmp/wrv
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

template <bool NT>
__device__ __forceinline__ d2 ld(const d2* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st(d2* p, d2 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// K = 128 dense: a row is 64 16-byte units, one per lane.  R rows per wave iteration.
// MODE 0: waves grid-stride over R-row groups; MODE 1: every workgroup owns one contiguous row range.
template <int R, bool NTL, bool NTS, int MODE, bool COPY>
__global__ __launch_bounds__(256) void wr_k(const double* __restrict__ A, const double* __restrict__ b,
                                            const double* __restrict__ w, const unsigned char* __restrict__ mask,
                                            int64_t m, double* __restrict__ aw, double* __restrict__ bw) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int64_t g0, g1, gs;           // groups of R rows
    const int64_t ngroups = (m + R - 1) / R;
    if (MODE == 0) {
        g0 = (int64_t)blockIdx.x * 4 + wv;
        g1 = ngroups;
        gs = (int64_t)gridDim.x * 4;
    } else {
        const int64_t per = (ngroups + gridDim.x - 1) / gridDim.x;
        g0 = (int64_t)blockIdx.x * per + wv;
        g1 = g0 - wv + per;
        if (g1 > ngroups) g1 = ngroups;
        gs = 4;
    }
    for (int64_t g = g0; g < g1; g += gs) {
        const int64_t row0 = g * R;
        d2 x[R];
        double wv_[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r < m ? row0 + r : m - 1;
            x[r] = ld<NTL>(reinterpret_cast<const d2*>(A + row * 128) + lane);
            if (!COPY) wv_[r] = mask[row] ? w[row] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (row0 + r < m) {
                d2 y = x[r];
                if (!COPY) {
                    y[0] *= wv_[r];
                    y[1] *= wv_[r];
                }
                st<NTS>(reinterpret_cast<d2*>(aw + (row0 + r) * 128) + lane, y);
            }
        }
        if (!COPY && lane < R && row0 + lane < m) {
            const int64_t row = row0 + lane;
            bw[row] = mask[row] ? w[row] * b[row] : 0.0;
        }
    }
}

struct Bufs {
    double *A, *b, *w, *aw, *bw;
    unsigned char* mask;
    int64_t m;
};

template <int R, bool NTL, bool NTS, int MODE, bool COPY>
void run(const Bufs& B, int grid, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((wr_k<R, NTL, NTS, MODE, COPY>), dim3(grid), dim3(256), 0, 0, B.A, B.b, B.w, B.mask, B.m, B.aw, B.bw);
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((wr_k<R, NTL, NTS, MODE, COPY>), dim3(grid), dim3(256), 0, 0, B.A, B.b, B.w, B.mask, B.m, B.aw, B.bw);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double bytes = COPY ? 2.0 * 1024 * B.m : (16.0 * 128 + 24) * B.m;
    printf("%-44s grid %5d  %.4f ms  %.0f GB/s\n", name, grid, ms, bytes / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    Bufs B;
    B.m = 1000000;
    const size_t ab = (size_t)B.m * 128 * 8;
    CK(hipMalloc(&B.A, ab));
    CK(hipMalloc(&B.aw, ab));
    CK(hipMalloc(&B.b, B.m * 8));
    CK(hipMalloc(&B.w, B.m * 8));
    CK(hipMalloc(&B.bw, B.m * 8));
    CK(hipMalloc(&B.mask, B.m));
    CK(hipMemset(B.A, 0, ab));
    CK(hipMemset(B.mask, 1, B.m));
    std::vector<double> h(B.m, 1.5);
    CK(hipMemcpy(B.w, h.data(), B.m * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(B.b, h.data(), B.m * 8, hipMemcpyHostToDevice));
    // clock pre-heat
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((wr_k<4, true, true, 0, true>), dim3(2048), dim3(256), 0, 0, B.A, B.b, B.w, B.mask, B.m, B.aw, B.bw);
    CK(hipDeviceSynchronize());
    for (int grid : {1024, 2048, 4096, 8192}) {
        run<4, true, true, 0, true>(B, grid, "copy R=4 ntl nts grid-stride");
        run<4, false, false, 0, true>(B, grid, "copy R=4 plain grid-stride");
        run<4, true, true, 0, false>(B, grid, "weight R=4 ntl nts grid-stride (current)");
        run<4, false, true, 0, false>(B, grid, "weight R=4 nts grid-stride");
        run<4, true, false, 0, false>(B, grid, "weight R=4 ntl grid-stride");
        run<4, false, false, 0, false>(B, grid, "weight R=4 plain grid-stride");
        run<8, true, true, 0, false>(B, grid, "weight R=8 ntl nts grid-stride");
        run<8, false, false, 0, false>(B, grid, "weight R=8 plain grid-stride");
        run<2, true, true, 0, false>(B, grid, "weight R=2 ntl nts grid-stride");
        run<4, true, true, 1, false>(B, grid, "weight R=4 ntl nts contiguous");
        run<8, true, true, 1, false>(B, grid, "weight R=8 ntl nts contiguous");
        run<4, false, false, 1, false>(B, grid, "weight R=4 plain contiguous");
    }
    run<4, true, true, 0, false>(B, 62500, "weight R=4 ntl nts one group per wave");
    run<8, true, true, 0, false>(B, 31250, "weight R=8 ntl nts one group per wave");
    run<4, false, false, 0, false>(B, 62500, "weight R=4 plain one group per wave");
    return 0;
}

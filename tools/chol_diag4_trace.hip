// tools/chol_diag4_trace.hip -- where do the cycles of the four-wave diagonal block (kernel 8b4) go?
// Includes the product kernels with -DFSNAP_CHOL_TRACE (shader-clock stamps per wave), factorises one 64 x 64 block
// and prints, per wave: entry, end of the consumer steps, owner start / end, exit -- in cycles from the first entry --
// next to the kernel's duration by HIP events, and the single-wave kernel 8b on the same block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFSNAP_CHOL_TRACE=1 -I include tools/chol_diag4_trace.hip -o tools/chol_diag4_trace
#include "../fitsnap_amd/csrc/fsnap_chol.hip"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

int main() {
    const int n = 64, ld = n + 32;
    std::vector<double> A(256 * n), G((size_t)n * ld, 0.0);
    unsigned x = 12345;
    for (auto& v : A) {
        x = x * 1664525u + 1013904223u;
        v = (double)(x >> 8) / (1 << 24) - 0.5;
    }
    std::vector<double> d(n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int r = 0; r < 256; ++r) s += A[r * n + i] * A[r * n + j];
            G[(size_t)i * ld + j] = s;
        }
    for (int i = 0; i < n; ++i) d[i] = 1.0 / std::sqrt(G[(size_t)i * ld + i]);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) G[(size_t)i * ld + j] *= d[i] * d[j];
    double *dS, *dU, *dY, *dmin;
    int* dst;
    CK(hipMalloc(&dS, G.size() * 8));
    CK(hipMalloc(&dU, G.size() * 8));
    CK(hipMalloc(&dY, 1024 * 8));
    CK(hipMalloc(&dmin, 64));
    CK(hipMalloc(&dst, 64));
    CK(hipMemset(dst, 0, 64));
    const double big = 1e300;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int form = 0; form < 3; form += 2) {
        float best = 1e9f, sum = 0;
        const int reps = 50;
        for (int it = 0; it < reps; ++it) {
            CK(hipMemcpy(dS, G.data(), G.size() * 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(dmin, &big, 8, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, 0));
            if (form == 2)
                hipLaunchKernelGGL(fsnap_chol_diag4_twice_k, dim3(1), dim3(256), 0, 0, (const double*)dS, dU, ld, 0, dY, dst, dmin);
            else
                hipLaunchKernelGGL(fsnap_chol_diag4_k, dim3(1), dim3(256), 0, 0, (const double*)dS, dU, ld, 0, dY, dst, dmin, (int*)nullptr);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it >= 5) {
                best = ms < best ? ms : best;
                sum += ms;
            }
        }
        printf("%s: events min %.2f us, mean %.2f us\n", form == 0 ? "four-wave 8b4" : form == 1 ? "single-wave 8b (chain 2)" : "four-wave 8b4 TWICE in one launch (stamps: second pass)", best * 1e3, sum / (reps - 5) * 1e3);
        if (form != 1) {
            long long t[4][8];
            CK(hipMemcpyFromSymbol(t, HIP_SYMBOL(chol_trace_buf), sizeof(t)));
            long long t0 = t[0][0];
            for (int w = 1; w < 4; ++w) t0 = t[w][0] < t0 ? t[w][0] : t0;
            printf("wave  entry  cons0  cons1  cons2  own_start  own_end  exit   (shader cycles from the first entry)\n");
            for (int w = 0; w < 4; ++w) {
                printf("%4d %6lld", w, t[w][0] - t0);
                for (int a = 0; a < 3; ++a) printf(" %6lld", a < w ? t[w][1 + a] - t0 : -1LL);
                printf(" %10lld %8lld %6lld\n", t[w][4] - t0, t[w][5] - t0, t[w][6] - t0);
            }
            std::vector<double> U(G.size());
            CK(hipMemcpy(U.data(), dU, G.size() * 8, hipMemcpyDeviceToHost));
            double err = 0;
            for (int i = 0; i < n; ++i)
                for (int j = i; j < n; ++j) {
                    double s = 0;
                    for (int r = 0; r <= i; ++r) s += U[(size_t)r * ld + i] * U[(size_t)r * ld + j];
                    err = std::fmax(err, std::fabs(s - G[(size_t)i * ld + j]));
                }
            printf("max |U^T U - S| = %.2e\n", err);
        }
    }
    return 0;
}

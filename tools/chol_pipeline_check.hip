// tools/chol_pipeline_check.hip -- the whole blocked device Cholesky solve (fsnap::launch_chol_large) on one system, checked
// against a host Cholesky solve in long double; with -DFSNAP_CHOL_TRACE=1 the cycle stamps of the last panel launch:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include [-DFSNAP_CHOL_TRACE=1] tools/chol_pipeline_check.hip -o tools/bin/chol_pipeline_check
#include "../fitsnap_amd/csrc/fsnap_chol.hip"

#include <chrono>
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

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256;
    const int m = 4 * n;
    std::vector<double> A((size_t)m * n), packed((size_t)n * n + n + 3, 0.0);
    unsigned x = 2024;
    for (auto& v : A) {
        x = x * 1664525u + 1013904223u;
        v = (double)(x >> 8) / (1 << 24) - 0.5;
    }
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = 0;
            for (int r = 0; r < m; ++r) s += A[(size_t)r * n + i] * A[(size_t)r * n + j];
            packed[(size_t)i * n + j] = packed[(size_t)j * n + i] = s;
        }
    for (int i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        packed[(size_t)n * n + i] = (double)(x >> 8) / (1 << 24) - 0.5;
    }
    // reference: Cholesky in long double
    std::vector<long double> L((size_t)n * n, 0.0L), y(n), ref(n);
    for (int j = 0; j < n; ++j) {
        long double d = packed[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        L[(size_t)j * n + j] = sqrtl(d);
        for (int i = j + 1; i < n; ++i) {
            long double s = packed[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / L[(size_t)j * n + j];
        }
    }
    for (int i = 0; i < n; ++i) {
        long double s = packed[(size_t)n * n + i];
        for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * y[k];
        y[i] = s / L[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        long double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * ref[k];
        ref[i] = s / L[(size_t)i * n + i];
    }
    const int np = (n + 63) / 64 * 64;
    double *dp, *work, *dsc, *z, *beta, *minpiv;
    int* status;
    CK(hipMalloc(&dp, packed.size() * 8));
    CK(hipMalloc(&work, fsnap::chol_large_work_doubles(n) * 8));
    CK(hipMalloc(&dsc, np * 8));
    CK(hipMalloc(&z, np * 8));
    CK(hipMalloc(&beta, np * 8));
    CK(hipMalloc(&minpiv, (np / 64 + 1) * 8));
    CK(hipMalloc(&status, 64));
    CK(hipMemcpy(dp, packed.data(), packed.size() * 8, hipMemcpyHostToDevice));
    printf("n = %d\n", n);
    {
        const int form = 5;
        std::vector<double> got(n);
        double best = 1e9;
        int st = 0;
        for (int it = 0; it < 12; ++it) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            CK(fsnap::launch_chol_large(dp, nullptr, n, 0.0, work, dsc, z, beta, status, minpiv, nullptr, true, nullptr, 0));
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (it >= 2 && us < best) best = us;
        }
        CK(hipMemcpy(got.data(), beta, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
        long double num = 0, den = 0;
        for (int i = 0; i < n; ++i) {
            num += (got[i] - ref[i]) * (got[i] - ref[i]);
            den += ref[i] * ref[i];
        }
        printf("form %d: status %d, rel err %.3e, %.1f us (launch + sync, best of 10)\n", form, st, (double)sqrtl(num / den), best);
#ifdef FSNAP_CHOL_TRACE
        if (form == 5) {
            // the LAST panel launch of the solve (kernel 8s, workgroup 0): where its ~17 us go, in shader cycles from the entry
            long long ts[4][4], tb[4][8];
            CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(chol_trace_step), sizeof(ts)));
            CK(hipMemcpyFromSymbol(tb, HIP_SYMBOL(chol_trace_buf), sizeof(tb)));
            long long t0 = ts[0][0];
            for (int w = 1; w < 4; ++w) t0 = ts[w][0] < t0 ? ts[w][0] : t0;
            printf("  wave  entry  strip_substituted  tiles_formed  own_start  own_end  pipeline_exit   (cycles)\n");
            long long tl[4];
            CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(chol_trace_last), sizeof(tl)));
            printf("  last pivot of block step 0 / 1 / 2 in a consumer's hands at %lld / %lld / %lld\n", tl[0] - t0, tl[1] - t0, tl[2] - t0);
            for (int w = 0; w < 4; ++w)
                printf("  %4d %6lld %18lld %13lld %10lld %8lld %14lld\n", w, ts[w][0] - t0, ts[w][1] - t0, ts[w][2] - t0, tb[w][4] - t0,
                       tb[w][5] - t0, ts[w][3] - t0);
        }
#endif
    }
    return 0;
}

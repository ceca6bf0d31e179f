// tools/trsm_check.hip — stand-alone check + timing of the row-space pass kernels (fsnap_trsm.hip): Q = X R^-1 for a random
// well-conditioned upper triangular R against a host substitution, error per 16-column block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/trsm_check.hip -o tools/trsm_check
//   tools/trsm_check m K [first=0|1] [reps]
#include "../fitsnap_amd/csrc/fsnap_trsm.hip"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
    const int64_t m = argc > 1 ? atoll(argv[1]) : 6001;
    const int K = argc > 2 ? atoi(argv[2]) : 208;
    const int first = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    const int only_rb = argc > 5 ? atoi(argv[5]) : -1;      // >= 0: off-diagonal entries only in block row only_rb (diagnosis)
    const int K16 = (K + 15) & ~15;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> X((size_t)m * K), R(fsnap::trsm_factor_doubles(K16), 0.0), wp((size_t)2 * m + 8, 0.0);
    for (auto& v : X) v = nd(rng);
    for (int i = 0; i < K16; ++i) R[(size_t)i * K16 + i] = 1.0;
    for (int i = 0; i < K; ++i) {
        R[(size_t)i * K16 + i] = 2.0 + 0.1 * nd(rng);
        for (int j = i + 1; j < K; ++j) {
            const double v = 0.1 * nd(rng) / std::sqrt((double)K);
            if (only_rb < 0 || i / 16 == only_rb) R[(size_t)i * K16 + j] = v;
        }
    }
    fsnap::trsm_invert_diagonal_blocks(R.data(), K16);
    for (int64_t r = 0; r < m; ++r) wp[2 * r] = (r % 7 == 3) ? 0.0 : 0.5 + (r % 5) * 0.25;
    double *dX, *dQ, *dR, *dW;
    hipMalloc(&dX, X.size() * 8 + 256);
    hipMalloc(&dQ, X.size() * 8 + 256);
    hipMalloc(&dR, R.size() * 8);
    hipMalloc(&dW, wp.size() * 8);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dW, wp.data(), wp.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < reps; ++it) {
        if (!first) hipMemcpy(dQ, X.data(), X.size() * 8, hipMemcpyHostToDevice);     // in place on Q
        hipEventRecord(e0, 0);
        hipError_t e = first ? fsnap::launch_trsm_rows(dX, K, dW, dQ, K, m, K, dR, K16, 0)
                             : fsnap::launch_trsm_rows(dQ, K, nullptr, dQ, K, m, K, dR, K16, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (e != hipSuccess || hipGetLastError() != hipSuccess) {
            printf("launch failed\n");
            return 1;
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    std::vector<double> Q(X.size());
    hipMemcpy(Q.data(), dQ, Q.size() * 8, hipMemcpyDeviceToHost);
    // host reference on a sample of rows
    std::vector<double> errblk((K16 / 16), 0.0);
    double emax = 0.0;
    const int64_t step = m > 4096 ? m / 4096 : 1;
    std::vector<double> x(K);
    for (int64_t r = 0; r < m; r += step) {
        const double w = first ? wp[2 * r] : 1.0;
        for (int j = 0; j < K; ++j) x[j] = (w != 0.0) ? w * X[(size_t)r * K + j] : 0.0;
        for (int i = 0; i < K; ++i) {
            const double q = x[i] / R[(size_t)i * K16 + i];
            x[i] = q;
            for (int j = i + 1; j < K; ++j) x[j] -= q * R[(size_t)i * K16 + j];
        }
        for (int j = 0; j < K; ++j) {
            const double d = std::fabs(Q[(size_t)r * K + j] - x[j]);
            errblk[j / 16] = d > errblk[j / 16] ? d : errblk[j / 16];
            emax = d > emax ? d : emax;
        }
    }
    printf("m %lld K %d first %d: %.3f ms (%.1f TF/s), max |Q - ref| = %.3e\n", (long long)m, K, first, best,
           (double)m * K * K / (best * 1e-3) / 1e12, emax);
    if (emax > 1e-9) {
        printf("  per 16-column block:");
        for (size_t b = 0; b < errblk.size(); ++b) printf(" %zu:%.1e", b, errblk[b]);
        printf("\n");
    }
    return emax > 1e-9;
}

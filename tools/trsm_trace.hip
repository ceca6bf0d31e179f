// tools/trsm_trace.hip -- where does the time of a row tile of kernel 13C go?  Includes the product kernels with
// -DFSNAP_TRSM_TRACE (wall-clock stamps per 64-row tile: entry, panel loaded + left-looking updates done, every 16-column block
// done, exit) and prints the average time between the stamps, next to the pass time by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFSNAP_TRSM_TRACE=1 -I include tools/trsm_trace.hip -o tools/bin/trsm_trace
//   tools/bin/trsm_trace [rows [K [first]]]
#include "../fitsnap_amd/csrc/fsnap_trsm.hip"

#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
    const int64_t m = argc > 1 ? atoll(argv[1]) : 1000000;
    const int K = argc > 2 ? atoi(argv[2]) : 128;
    const int first = argc > 3 ? atoi(argv[3]) : 0;
    const int K16 = (K + 15) & ~15;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> X((size_t)m * K), R(fsnap::trsm_factor_doubles(K16), 0.0), wp((size_t)2 * m + 8, 0.0);
    for (auto& v : X) v = nd(rng);
    for (int i = 0; i < K16; ++i) R[(size_t)i * K16 + i] = 1.0;
    for (int i = 0; i < K; ++i) {
        R[(size_t)i * K16 + i] = 2.0 + 0.1 * nd(rng);
        for (int j = i + 1; j < K; ++j) R[(size_t)i * K16 + j] = 0.1 * nd(rng) / std::sqrt((double)K);
    }
    fsnap::trsm_invert_diagonal_blocks(R.data(), K16);
    for (int64_t r = 0; r < m; ++r) wp[2 * r] = 0.5 + (r % 5) * 0.25;
    double *dX, *dQ, *dR, *dW;
    hipMalloc(&dX, X.size() * 8 + 256);
    hipMalloc(&dQ, X.size() * 8 + 256);
    hipMalloc(&dR, R.size() * 8);
    hipMalloc(&dW, wp.size() * 8);
    hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dQ, X.data(), X.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dW, wp.data(), wp.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0, 0);
        hipError_t e = first ? fsnap::launch_trsm_rows(dX, K, dW, dQ, K, m, K, dR, K16, 0)
                             : fsnap::launch_trsm_rows(dQ, K, nullptr, dQ, K, m, K, dR, K16, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (e != hipSuccess) return 1;
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const int ntile = (int)((m + 63) / 64 < 8192 ? (m + 63) / 64 : 8192);
    std::vector<unsigned long long> tr(8192 * 16);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(fsnap_trsm_trace), tr.size() * 8);
    printf("%lld x %d, first %d: pass %.3f ms (%.1f TF/s); %d tiles traced\n", (long long)m, K, first, best,
           (double)m * K * K / (best * 1e-3) / 1e12, ntile);
    const int NB = K16 / 16, NP = (NB + 3) / 4;
    std::vector<int> idx = {0};
    for (int P = 0; P < NP; ++P) {
        idx.push_back(1 + 5 * P);
        const int nbp = NB - 4 * P < 4 ? NB - 4 * P : 4;
        for (int J = 0; J < nbp; ++J) idx.push_back(2 + 5 * P + J);
    }
    idx.push_back(15);
    double total = 0;
    for (size_t k = 1; k < idx.size(); ++k) {
        double av = 0, mx = 0;
        for (int t = 0; t < ntile; ++t) {
            const double d = (double)(tr[t * 16 + idx[k]] - tr[t * 16 + idx[k - 1]]) * 0.01;
            av += d;
            mx = d > mx ? d : mx;
        }
        av /= ntile;
        total += av;
        const int i = idx[k];
        if (i == 15) printf("  -> exit                                   %7.2f us (max %.2f)\n", av, mx);
        else if ((i - 1) % 5 == 0) printf("  -> panel %d loaded, left-looking updates done %7.2f us (max %.2f)\n", (i - 1) / 5, av, mx);
        else printf("  -> block %d of panel %d done                  %7.2f us (max %.2f)\n", (i - 2) % 5, (i - 2) / 5, av, mx);
    }
    printf("  entry -> exit of a tile, average %.2f us\n", total);
    if (NB >= 3) {
        // inside block 1 of panel 0: stamps 2 (block 0 done) -> 11 (block and R_JJ in LDS) -> 12 (reciprocals) -> 13 (substitution,
        // solved block back in LDS) -> 14 (stores issued, A operands read) -> 3 (updates done)
        const int seq[6] = {2, 11, 12, 13, 14, 3};
        const char* nm[5] = {"block + R_JJ into LDS (waits for R_JJ from L2)", "reciprocals of the diagonal", "substitution + block back to LDS",
                             "stores issued, A operands read", "updates (32 MFMAs)"};
        for (int k = 0; k < 5; ++k) {
            double av = 0;
            for (int t = 0; t < ntile; ++t) av += (double)(tr[t * 16 + seq[k + 1]] - tr[t * 16 + seq[k]]) * 0.01;
            printf("     block 1 of panel 0: %-50s %6.2f us\n", nm[k], av / ntile);
        }
    }
    return 0;
}

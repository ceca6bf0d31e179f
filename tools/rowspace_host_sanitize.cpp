// Sanitizer driver for the host K x K end of the row-space solve (fsnap_rowspace_host.cpp): triangular, deflated and Jacobi forms of
// FactorSolver, certified and multiplied-out FactorChain, ragged sizes (n % 4, % 8, % 32, % 64) and the threaded phases (n >= 384).
// CPU only (GPU AddressSanitizer is not available on the pool):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -mavx2 -mfma -pthread -I fitsnap_amd/csrc -I include \
//       tools/rowspace_host_sanitize.cpp fitsnap_amd/csrc/fsnap_rowspace_host.cpp fitsnap_amd/csrc/fsnap_solve.cpp -o /tmp/rs_san && FSNAP_HOST_THREADS=6 /tmp/rs_san
// Round 6 (after the tiled product / inverse, the Scratch pool and the vectorised substitutions): no report.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "fsnap_rowspace_host.h"
using namespace fsnap_rs;
static std::vector<double> upper(int n, std::mt19937_64& g, int ndep, double grade) {
    std::normal_distribution<double> N(0, 1);
    // R = qr-like upper factor: random upper triangle with a graded diagonal; ndep tiny pivots
    std::vector<double> R((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) {
        for (int j = i + 1; j < n; ++j) R[(size_t)i * n + j] = 0.5 * N(g) / n;
        R[(size_t)i * n + i] = std::pow(10.0, -grade * i / std::max(1, n - 1)) * (1.0 + 0.1 * std::fabs(N(g)));
    }
    for (int d = 0; d < ndep; ++d) R[(size_t)(n - 1 - 3 * d) * n + (n - 1 - 3 * d)] = 1e-17;
    return R;
}
int main() {
    std::mt19937_64 g(7);
    std::normal_distribution<double> N(0, 1);
    for (int n : {1, 3, 31, 67, 130, 200, 259, 450, 1100}) {
        for (int ndep : {0, 2, 6}) {
            if (3 * ndep >= n) continue;
            std::vector<double> R1 = upper(n, g, ndep, 3.0), R2 = upper(n, g, 0, 0.0), z(n), beta(n);
            for (auto& v : z) v = N(g);
            // single factor through FactorSolver
            {
                FactorSolver fs;
                fs.prepare(n, R1.data(), 1e-13);
                fs.apply(z.data(), beta.data());
                double s = 0; for (double v : beta) s += v;
                printf("n=%d ndep=%d solver rank %d deflated %d tri %d sum %.3e\n", n, ndep, fs.rank, (int)fs.deflated, (int)fs.triangular, s);
            }
            // chain of two factors
            {
                FactorChain ch;
                std::vector<double> G((size_t)n * n, 0.0);
                for (int i = 0; i < n; ++i) G[(size_t)i * n + i] = 1.0;
                ch.start(n, G.data());
                ch.push(R1.data());
                ch.push(R2.data());
                double nrm, inv, bound;
                const bool ok = ch.certified(1e-13, &nrm, &inv, &bound);
                if (ok) ch.solve(z.data(), beta.data());
                else {
                    Scratch Rh((size_t)n * n);
                    ch.product(Rh.data());
                    FactorSolver fs;
                    fs.prepare(n, Rh.data(), 1e-13);
                    fs.apply(z.data(), beta.data());
                }
                printf("   chain certified %d bound %.3e\n", (int)ok, bound);
            }
        }
    }
    return 0;
}

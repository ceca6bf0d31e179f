// tools/short_trace.hip -- where does the time of kernel 1S go?  Includes the product kernel with -DFSNAP_SHORT_TRACE
// (wall-clock stamps per workgroup: entry, pairs in LDS, rows staged, products done, tiles stored), runs it on a random
// m x K system and prints the stamps relative to the first entry, next to the launch's duration by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFSNAP_SHORT_TRACE=1 -I include tools/short_trace.hip -o tools/bin/short_trace
//   tools/bin/short_trace [rows [K [rows_per_chunk]]]
#include "../fitsnap_amd/csrc/fsnap_syrk_short.hip"
namespace fsnap {
int syrk_num_blocks(int K) { return (K + 15) / 16; }
}

#include <algorithm>
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
    const int64_t m = argc > 1 ? atoll(argv[1]) : 13035;
    const int K = argc > 2 ? atoi(argv[2]) : 142;
    int64_t rpc = argc > 3 ? atoll(argv[3]) : ((m + 127) / 128 + 3) / 4 * 4;
    if (rpc < 32) rpc = 32;
    const int nchunk = (int)((m + rpc - 1) / rpc);
    const int NB = (K + 15) / 16, NT = NB * (NB + 1) / 2;
    std::vector<double> A((size_t)m * K), b(m), w(m);
    unsigned x = 12345;
    auto rnd = [&] {
        x = x * 1664525u + 1013904223u;
        return (double)(x >> 8) / (1 << 24) - 0.5;
    };
    for (auto& v : A) v = rnd();
    for (auto& v : b) v = rnd();
    for (auto& v : w) v = 1.0 + rnd();
    std::vector<unsigned char> mask(m, 1);
    double *dA, *db, *dw, *part, *cpart, *spart;
    unsigned char* dm;
    CK(hipMalloc(&dA, A.size() * 8 + 256));
    CK(hipMalloc(&db, m * 8));
    CK(hipMalloc(&dw, m * 8));
    CK(hipMalloc(&dm, m));
    CK(hipMalloc(&part, (size_t)nchunk * NT * 256 * 8));
    CK(hipMalloc(&cpart, (size_t)nchunk * NB * 16 * 8));
    CK(hipMalloc(&spart, (size_t)nchunk * 4 * 8));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), m * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, w.data(), m * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dm, mask.data(), m, hipMemcpyHostToDevice));
    fsnap::SyrkArgs a;
    a.A = dA; a.lda = K; a.b = db; a.w = dw; a.mask = dm; a.m = m; a.K = K; a.nblocks = nchunk; a.split = 1;
    a.chunks_per_wave = rpc; a.nontemporal = false; a.part = part; a.cpart = cpart; a.spart = spart; a.fused_pack = true;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) CK(fsnap::launch_syrk_short(a, 0));
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    const int reps = 50;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, 0));
        CK(fsnap::launch_syrk_short(a, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
        sum += ms;
    }
    // back-to-back launches: the launch rate the stream sustains
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 100; ++i) CK(fsnap::launch_syrk_short(a, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms100;
    CK(hipEventElapsedTime(&ms100, e0, e1));
    const int nwg = 16 * ((nchunk + 7) / 8);
    std::vector<unsigned long long> tr(1024 * 8);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(fsnap_short_trace), tr.size() * 8));
    unsigned long long t00 = ~0ull, tend = 0;
    for (int g = 0; g < nwg && g < 1024; ++g) {
        const int chunk = (g & 7) + 8 * (g >> 4);
        if (chunk >= nchunk) continue;
        t00 = std::min(t00, tr[g * 8]);
        tend = std::max(tend, tr[g * 8 + 4]);
    }
    printf("%lld x %d, %d chunks of %lld rows, %d workgroups: events avg %.2f us, min %.2f us; 100 back-to-back launches %.2f us each\n",
           (long long)m, K, nchunk, (long long)rpc, nwg, sum / reps * 1e3, best * 1e3, ms100 * 10.0);
    const char* names[5] = {"entry", "pairs in LDS", "rows staged", "products done", "tiles stored"};
    for (int s = 0; s < 5; ++s) {
        double mn = 1e30, mx = 0, av = 0;
        int n = 0;
        for (int g = 0; g < nwg && g < 1024; ++g) {
            const int chunk = (g & 7) + 8 * (g >> 4);
            if (chunk >= nchunk) continue;
            const double t = (double)(tr[g * 8 + s] - t00) * 0.01;
            mn = std::min(mn, t);
            mx = std::max(mx, t);
            av += t;
            ++n;
        }
        printf("  %-14s us after the first entry: min %6.2f  avg %6.2f  max %6.2f\n", names[s], mn, av / n, mx);
    }
    printf("  first entry -> last tile stored %.2f us\n", (double)(tend - t00) * 0.01);
    return 0;
}

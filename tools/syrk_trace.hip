// tools/syrk_trace.hip — where and when do the workgroups of kernel 1L run?
// Includes the product kernels with -DFSNAP_TRACE (per-workgroup start/end wall clock, HW_ID,
// XCC_ID), launches fsnap_syrk_lds_static on a synthetic 10^6 x 128 problem with the same
// geometry the C-ABI layer plans, and prints per-workgroup lifetimes grouped by compute unit.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFSNAP_TRACE=1 tools/syrk_trace.hip -o tools/syrk_trace
//   (-DFSNAP_TRACE=2 adds per-wave phase clocks inside the stage loop; they perturb the kernel by ~20 %)
//   usage: syrk_trace [rows] [workgroups] [verbose] [waves per workgroup: 8|4|16] [variant]
#include "../fitsnap_amd/csrc/fsnap_syrk.hip"
#include "../fitsnap_amd/csrc/fsnap_rows.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

__global__ void fill(double* p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15;
        x *= 2246822519u;
        x ^= x >> 13;
        p[i] = (double)(x & 0xffff) / 65536.0 - 0.5;
    }
}

int main(int argc, char** argv) {
    const int64_t m = argc > 1 ? atoll(argv[1]) : 1000000;
    const int K = 128;
    int nblocks = argc > 2 ? atoi(argv[2]) : 512;
    const int verbose = argc > 3 ? atoi(argv[3]) : 0;
    const int nw = argc > 4 ? atoi(argv[4]) : 8;        // waves per workgroup (8 default, 4, 16)
    const int ablate = argc > 5 ? atoi(argv[5]) : 0;    // kernel variant (option "ablate")
    const int gap_us = argc > 6 ? atoi(argv[6]) : 0;    // idle time between launches
    double *A, *b, *w, *part, *cpart, *spart;
    unsigned char* mask;
    CK(hipMalloc(&A, (size_t)m * K * 8 + 256));
    CK(hipMalloc(&b, m * 8));
    CK(hipMalloc(&w, m * 8));
    CK(hipMalloc(&mask, m));
    CK(hipMemset(mask, 1, m));
    fill<<<2048, 256>>>(A, (size_t)m * K, 1u);
    fill<<<256, 256>>>(b, m, 2u);
    fill<<<256, 256>>>(w, m, 3u);
    const int64_t nchunks = (m + 3) / 4;
    int64_t cpwg = (nchunks + nblocks - 1) / nblocks;
    if (nw == 1) {   // kernel 1A: nblocks workgroups of 4 independent row-waves, cpwg = chunks per row-wave
        cpwg = (nchunks + (int64_t)nblocks * 4 - 1) / ((int64_t)nblocks * 4);
        nblocks = (int)((nchunks + cpwg * 4 - 1) / (cpwg * 4));
    } else {
        cpwg = (cpwg + nw - 1) / nw * nw;
        nblocks = (int)((nchunks + cpwg - 1) / cpwg);
    }
    CK(hipMalloc(&part, (size_t)nblocks * 36 * 256 * 8));
    CK(hipMalloc(&cpart, (size_t)nblocks * 4 * 128 * 8));
    CK(hipMalloc(&spart, (size_t)nblocks * 4 * 4 * 8));
    fsnap::SyrkArgs a{};
    a.A = A; a.lda = K; a.b = b; a.w = w; a.mask = mask; a.m = m; a.K = K;
    a.nblocks = nblocks; a.split = nw; a.chunks_per_wave = cpwg; a.nontemporal = true; a.ablate = ablate;
    a.part = part; a.cpart = cpart; a.spart = spart;
    double *wpack, *wpack_spart;                      // kernel 1A: packed (w_eff, w_eff b) per row
    CK(hipMalloc(&wpack, (size_t)m * 16 + 64));
    CK(hipMalloc(&wpack_spart, (size_t)fsnap::pack_weights_num_blocks(m) * 32));
    CK(fsnap::launch_pack_weights(b, w, mask, m, wpack, wpack_spart, 0));
    a.wpack = wpack;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0;
    // ~200 launches first: an idle MI355X needs ~35 ms of load to reach its steady clock (scripts/ramptest.py)
    for (int it = 0; it < 200; ++it) {
        CK(hipEventRecord(e0, 0));
        if (nw == 1) CK(fsnap::launch_syrk_acc(a, 0));
        else CK(fsnap::launch_syrk_lds(a, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (gap_us > 0) {      // idle gap between launches (a fit loop's host solve), busy-waited
            const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(gap_us);
            while (std::chrono::steady_clock::now() < t_end) {}
        }
    }
    std::vector<unsigned long long> tr((size_t)nblocks * 8);
    CK(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(fsnap_trace_buf), tr.size() * 8));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int g = 0; g < nblocks; ++g) {
        tmin = std::min(tmin, tr[g * 8]);
        tmax = std::max(tmax, tr[g * 8 + 1]);
    }
    printf("nw=%d variant=%d gap=%dus ", nw, ablate, gap_us);
    printf("m=%lld workgroups=%d chunks/wg=%lld kernel %.1f us (events), trace span %.1f us\n", (long long)m, nblocks,
           (long long)cpwg, ms * 1e3, (tmax - tmin) * 0.01);
    std::map<unsigned, std::vector<int>> bycu;
    for (int g = 0; g < nblocks; ++g) {
        const unsigned hw = (unsigned)tr[g * 8 + 2], xcc = (unsigned)tr[g * 8 + 3] & 15u;
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        bycu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(g);
    }
    // lifetime statistics by dispatch layer (k-th workgroup placed on a compute unit)
    double lsum[8] = {}, esum[8] = {}, ssum[8] = {};
    int lcnt[8] = {};
    std::map<int, int> occupancy;
    for (auto& kv : bycu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end());   // dispatch order = age
        occupancy[(int)v.size()]++;
        for (size_t i = 0; i < v.size() && i < 8; ++i) {
            const double ts = (tr[v[i] * 8] - tmin) * 0.01, te = (tr[v[i] * 8 + 1] - tmin) * 0.01;
            lsum[i] += te - ts; esum[i] += te; ssum[i] += ts; ++lcnt[i];
            if (verbose) printf("  cu %05x wg %4d start %7.2f end %7.2f us\n", kv.first, v[i], ts, te);
        }
    }
    {
        double cyc = 0, wall = 0;
        for (int g = 0; g < nblocks; ++g) {
            cyc += (double)tr[g * 8 + 4];
            wall += (double)(tr[g * 8 + 1] - tr[g * 8]) * 10e-9;
        }
        printf("shader clock while the workgroups ran: %.3f GHz (s_memtime cycles / 100 MHz wall clock)\n", cyc / wall / 1e9);
        if (nw == 1) {
            double ep = 0, last = 0;
            for (int g = 0; g < nblocks; ++g) {
                ep += (double)(tr[g * 8 + 5] - tr[g * 8 + 1]) * 0.01;
                last = std::max(last, (double)(tr[g * 8 + 5] - tmin) * 0.01);
            }
            printf("kernel 1A epilogue (accumulator read-out, 4-wave fold, partial stores): mean %.2f us per workgroup, last one done at %.1f us\n",
                   ep / nblocks, last);
        }
    }
    printf("compute units used %zu; workgroups per CU histogram:", bycu.size());
    for (auto& o : occupancy) printf(" %d:%d", o.first, o.second);
    printf("\n");
    for (int i = 0; i < 8; ++i)
        if (lcnt[i]) printf("workgroup #%d of a CU: n=%d mean start %.1f us, mean end %.1f us, mean life %.1f us\n", i, lcnt[i],
                            ssum[i] / lcnt[i], esum[i] / lcnt[i], lsum[i] / lcnt[i]);
    // where a wave's time goes inside the stage loop (shader-clock cycles per stage)
#if FSNAP_TRACE >= 2
    {
        std::vector<unsigned long long> tw((size_t)nblocks * 16 * 4);
        CK(hipMemcpyFromSymbol(tw.data(), HIP_SYMBOL(fsnap_trace_wave), tw.size() * 8));
        for (int layer = 0; layer < 4; ++layer) {
            double a[16][3] = {};
            int cnt = 0;
            for (auto& kv : bycu) {
                auto& v = kv.second;
                if ((int)v.size() <= layer || v.size() < 2) continue;
                const int g = v[layer];
                for (int wv = 0; wv < nw; ++wv) {
                    const unsigned long long* o = &tw[((size_t)g * 16 + wv) * 4];
                    const double ns = (double)o[3];
                    for (int k = 0; k < 3; ++k) a[wv][k] += o[k] / ns;
                }
                ++cnt;
            }
            if (!cnt) continue;
            printf("workgroup #%d of a CU (by start time), cycles per stage by wave [mfma phase | park+issue | barrier]:\n", layer);
            for (int wv = 0; wv < nw; ++wv)
                printf("  wave %d: %8.0f %8.0f %8.0f  (sum %8.0f)\n", wv, a[wv][0] / cnt, a[wv][1] / cnt, a[wv][2] / cnt,
                       (a[wv][0] + a[wv][1] + a[wv][2]) / cnt);
        }
    }
#endif
    // distribution of end times
    std::vector<double> ends;
    for (int g = 0; g < nblocks; ++g) ends.push_back((tr[g * 8 + 1] - tmin) * 0.01);
    std::sort(ends.begin(), ends.end());
    printf("end-time percentiles (us): p0 %.1f p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f p100 %.1f\n", ends[0],
           ends[ends.size() / 10], ends[ends.size() / 4], ends[ends.size() / 2], ends[ends.size() * 3 / 4],
           ends[ends.size() * 9 / 10], ends.back());
    return 0;
}

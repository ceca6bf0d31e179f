// tools/graph_probe.hip -- is a hipGraph launch of the per-fit kernel sequence faster than plain launches?
// A short fit is launch-bound: SYRK kernel (10 ... 20 us) -> reduction kernel (5 ... 8 us) -> event -> the host polls the event
// and solves.  This probe times, from the host's point of view (steady_clock around submit ... event complete), the same two
// dependent kernels of about those durations (a) as two hipLaunchKernelGGL + hipEventRecord on a stream, (b) as one
// hipGraphLaunch of the captured pair + hipEventRecord, (c) the graph with the event record captured as a node as well.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/graph_probe.hip -o tools/bin/graph_probe && tools/bin/graph_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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

// streams n doubles per workgroup-strided thread and spins `spin` dependent FMAs: a kernel of a chosen length
__global__ void busy_k(const double* __restrict__ in, double* __restrict__ out, int64_t n, int spin) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    double s = 0.0;
    for (int64_t i = tid; i < n; i += nt) s += in[i];
    for (int k = 0; k < spin; ++k) s = __builtin_fma(s, 1.0000001, 1e-9);
    out[tid] = s;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const int64_t n1 = 13035ll * 142, n2 = 126ll * 45 * 256;      // the bytes of the 13 035 x 142 SYRK and of its partials
    double *a, *p, *o1, *o2;
    CK(hipMalloc(&a, n1 * 8));
    CK(hipMalloc(&p, n2 * 8));
    CK(hipMalloc(&o1, 256 * 512 * 8));
    CK(hipMalloc(&o2, 370 * 256 * 8));
    CK(hipMemset(a, 0, n1 * 8));
    CK(hipMemset(p, 0, n2 * 8));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    auto launch_pair = [&](hipStream_t s) {
        hipLaunchKernelGGL(busy_k, dim3(256), dim3(512), 0, s, (const double*)a, o1, n1, 1500);
        hipLaunchKernelGGL(busy_k, dim3(370), dim3(256), 0, s, (const double*)p, o2, n2, 300);
    };
    // the kernels' own durations
    {
        hipEvent_t e0, e1, e2;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventCreate(&e2));
        for (int i = 0; i < 50; ++i) launch_pair(st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(busy_k, dim3(256), dim3(512), 0, st, (const double*)a, o1, n1, 1500);
        CK(hipEventRecord(e1, st));
        hipLaunchKernelGGL(busy_k, dim3(370), dim3(256), 0, st, (const double*)p, o2, n2, 300);
        CK(hipEventRecord(e2, st));
        CK(hipStreamSynchronize(st));
        float t1, t2;
        CK(hipEventElapsedTime(&t1, e0, e1));
        CK(hipEventElapsedTime(&t2, e1, e2));
        printf("kernels by events: %.1f us + %.1f us\n", t1 * 1e3, t2 * 1e3);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    launch_pair(st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipGraph_t g2;
    hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    launch_pair(st);
    CK(hipEventRecord(ev, st));
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    auto run = [&](int mode, const char* name) {
        std::vector<double> t(reps), sub(reps);
        for (int i = -100; i < reps; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            if (mode == 0) {
                launch_pair(st);
                CK(hipEventRecord(ev, st));
            } else if (mode == 1) {
                CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(ev, st));
            } else {
                CK(hipGraphLaunch(ge2, st));
            }
            const auto t1 = std::chrono::steady_clock::now();
            while (hipEventQuery(ev) == hipErrorNotReady) {
            }
            const auto t2 = std::chrono::steady_clock::now();
            if (i >= 0) {
                t[i] = std::chrono::duration<double, std::micro>(t2 - t0).count();
                sub[i] = std::chrono::duration<double, std::micro>(t1 - t0).count();
            }
        }
        std::sort(t.begin(), t.end());
        std::sort(sub.begin(), sub.end());
        printf("%-58s submit -> event seen: median %.1f us (p10 %.1f, p90 %.1f); host time in the submit calls: median %.1f us\n", name,
               t[reps / 2], t[reps / 10], t[reps * 9 / 10], sub[reps / 2]);
    };
    for (int r = 0; r < 2; ++r) {
        run(0, "two launches + event record");
        run(1, "hipGraphLaunch (two kernel nodes) + event record");
        run(2, "hipGraphLaunch (two kernel nodes + event record node)");
    }
    return 0;
}

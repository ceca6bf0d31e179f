import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from fitsnap_amd import _capi
from fitsnap_amd import synthetic as orc  # input data only
A, b, w = orc.synth_problem(1000000, 128)
ctx = _capi.HipContext(0); ctx.upload_rows(A, b); ctx.set_weights(w)
dev = torch.device("cuda", 0)
packed = torch.zeros(128*128+128+3, dtype=torch.float64, device=dev)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
def run(label, resident, timing_each, n=40):
    ks = []
    def one():
        if resident:
            ptr = ctx.normal_eq_resident()
        else:
            ptr = packed.data_ptr(); ctx.normal_eq_async(ptr)
        beta, _, _ = ctx.solve_device(_capi.SOLVE_RIDGE, 1e-8, 128, ptr)
        if timing_each:
            ks.append(ctx.timing(2)["syrk_ms"])
    for _ in range(5): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
    last = ctx.timing(2)["syrk_ms"]
    print(f"{label:55s} {dt:.4f} ms/step  kernel(last) {last:.4f}" + (f"  kernel(avg) {np.mean(ks[-n:]):.4f}" if ks else ""))
for rep in range(2):
    run("async + solve_device(torch ptr), no per-step timing", False, False)
    run("async + solve_device(torch ptr), timing each step", False, True)
    run("resident (mirror), no per-step timing", True, False)
    run("resident (mirror), timing each step", True, True)

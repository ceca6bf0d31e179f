"""Cost of one candidate of a re-weighting (GA) loop: perform_fit + error_analysis on resident rows
(examples/library/genetic_algorithm/libmod_optimize.py:461-488 in the reference)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from fitsnap_amd.config import Config
from fitsnap_amd.parallel_tools import ParallelTools
from fitsnap_amd.solvers import solver_factory
from fitsnap_amd import synthetic as orc  # input data only


def loop(A, b, w, fsd, t, trust, rng):
    # trust = False: the label lists are fingerprinted in full on every call (an in-place edit of any entry is seen);
    # trust = True: the caller promises pt.touch_labels() after an edit, the caches are keyed on that version
    pt = ParallelTools()
    cfg = Config(pt, {"SOLVER": {"solver": "RIDGE"}, "RIDGE": {"alpha": 1e-8}})
    s = solver_factory.solver("RIDGE", pt, cfg)
    s.keep_resident = True
    s.trust_label_version = trust
    times = []
    for it in range(8):
        w_it = w * rng.uniform(0.5, 2.0)                       # a new candidate's weights ...
        w_tr = w_it[~t]                                        # ... one per training row, as the reference's GA passes them
        t0 = time.perf_counter()
        s.fit = None
        s.perform_fit(A, b, w_tr, fs_dict=fsd)
        t1 = time.perf_counter()
        s.error_analysis(A, b, w_it, fsd)
        rmse = s.errors.iloc[:, 2].to_numpy()[:3]
        t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
    pt.free()
    return times, rmse


for m, K, ngroups in ((15213, 31, 12), (1000000, 128, 40)):
    A, b, w = orc.synth_problem(m, K)
    rng = np.random.default_rng(3)
    groups = [f"g{g:02d}" for g in np.sort(rng.integers(0, ngroups, size=m))]
    testing = (rng.random(m) < 0.1).tolist()
    row_type = [("Energy", "Force", "Stress")[i % 3] for i in range(m)]
    fsd = {"Groups": groups, "Testing": testing, "Row_Type": row_type}
    t = np.asarray(testing)
    import pandas as pd
    fsd_np = {"Groups": pd.Categorical(groups), "Testing": np.asarray(testing), "Row_Type": pd.Categorical(row_type)}
    from fitsnap_amd.parallel_tools import LabelList
    # what this package's own producers hand out (Calculator.collect_distributed_lists, FitSnap.load_descriptors): lists that
    # count their in-place edits -- the whole-content fingerprint is O(1)
    fsd_ll = {k: LabelList(v) for k, v in fsd.items()}
    for trust, labels in ((False, fsd_ll), (False, fsd), (False, fsd_np), (True, fsd)):
        times, rmse = loop(A, b, w, labels, t, trust, rng)
        tt = np.array(times[2:])
        how = ("trusted by version (pt.touch_labels)" if trust else
               "LabelList (package-produced: a list that counts its edits)" if labels is fsd_ll else
               "Python lists, fingerprinted in full every call" if labels is fsd else
               "numpy bool array + pandas Categoricals, fingerprinted in full every call (xxh3 over their buffers)")
        print(f"{m} x {K}, {ngroups} groups, labels {how}: perform_fit {tt[:,0].mean()*1e3:.2f} ms, error_analysis "
              f"{tt[:,1].mean()*1e3:.2f} ms per candidate (first call: {times[0][0]*1e3:.1f} + {times[0][1]*1e3:.1f} ms); rmse {rmse}")

"""CPU: the parts of bench.py that do not need a GPU -- the HBM-traffic record is only reported for the kernel source,
shape and launch geometry it was measured for, and the committed record matches the committed kernel sources."""
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _records():
    recs = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    return [recs] if isinstance(recs, dict) else recs


def test_committed_traffic_record_belongs_to_the_committed_kernel_sources():
    bench = _bench()
    recs = _records()
    assert recs[0]["rows"] == 1000000 and recs[0]["K"] == 128          # the headline shape comes first
    for rec in recs:
        assert rec["source_sha256"] == bench.kernel_source_digest(rec["kernel"]), \
            "the kernel's source file / fsnap_device_common.h changed after the PMC passes: re-run scripts/pmc_record_all.sh"
        info = {"workgroups": rec["workgroups"], "threads": rec["threads"], "chunks_per_wave": rec["chunks_per_wave"]}
        traffic, source = bench.recorded_traffic(rec["rows"], rec["K"], info, rec["kernel"])
        assert traffic == rec["hbm_bytes_per_launch"] and "FETCH_SIZE" in source
    # headline kernel: 1.02 x the algorithmic bytes, the rows are read once
    rec = recs[0]
    assert 1.0 <= rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"] < 1.05


def test_traffic_is_null_for_any_other_shape_geometry_or_kernel():
    bench = _bench()
    rec = _records()[0]
    info = {"workgroups": rec["workgroups"], "threads": rec["threads"], "chunks_per_wave": rec["chunks_per_wave"]}
    for change in ({"rows": rec["rows"] + 1}, {"K": 96}, {"kernel": "fsnap_syrk_tiled"}):
        args = {"rows": rec["rows"], "K": rec["K"], "kernel": rec["kernel"], **change}
        traffic, why = bench.recorded_traffic(args["rows"], args["K"], info, args["kernel"])
        assert traffic is None and "recorded for" in why
    other = dict(info, chunks_per_wave=info["chunks_per_wave"] + 1)
    assert bench.recorded_traffic(rec["rows"], rec["K"], other, rec["kernel"])[0] is None

"""profiles/current.json: every registered rocprofv3 summary exists, parses and yields a roofline object (CPU only)."""
import json
import os

import pytest

from conftest import REPO
import bench


def registered():
    cur = json.load(open(os.path.join(REPO, "profiles", "current.json")))
    return [k for k in cur if not k.startswith("_")]


@pytest.mark.parametrize("key", registered())
def test_registered_profile_yields_a_roofline(key):
    cur, stats, pmc = bench.load_profile(key)
    assert cur and stats and pmc, key
    assert os.path.exists(os.path.join(REPO, "profiles", cur["kernel_stats"])) and os.path.exists(os.path.join(REPO, "profiles", cur["pmc"]))
    assert pmc.get("kernel_source_digest"), "a PMC summary records the kernel sources it was taken from"
    workload = "synthetic" if key.startswith("synthetic_") else key
    schedule = {"config2": "wavefront", "config4": "tile"}.get(key)
    names = " ".join(r["Name"] for r in stats)
    if schedule is None and key != "pathtracer":        # scenes read from HBM: whichever schedule the bench line adopted
        schedule = max(("tree", "whitted_tree_kernel"), ("tile", "wf_tile_kernel"), ("wavefront", "wf_secondary_kernel"),
                       key=lambda kv: sum(float(r.get("TotalDurationNs") or 0) for r in stats if kv[1] in r["Name"]))[0]
    live = {"schedule": schedule, "kernel_ms": 1.0, "frame_ms": 1.0, "frame_ms_in_flight": 1.0, "frames_in_flight": 4}
    r = bench.roofline_from_profiles(workload, live, key if key.startswith("synthetic_") else None)
    assert r["kernel"] and r["kernel"] in " ".join(bench._kname(n) for n in names.split("void ")) or r["kernel"] in names
    assert r["frac"] is not None and 0.0 < r["frac"] <= 1.0
    assert r["bound"] in ("valu_issue", "fetch_latency") and r["peak"] > 0 and r["kernel_us_profile"] > 0
    if key != "pathtracer":
        w = r["whole_frame"]
        assert 0.0 < w["frac"] and w["issue_mix"]["salu_wave_instr_per_frame"] > 0

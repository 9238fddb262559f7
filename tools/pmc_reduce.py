#!/usr/bin/env python3
"""Per-unit PMC constants of the STEADY STATE from a rocprofv3 --pmc pass over a bench.py run (round 6).

The pass sees the whole process: the pre-roll, in which every stream's BA window grows from 1 to 10 keyframes, included.  Dividing
the per-kernel totals by the whole process' unit counts (rounds 2-5) therefore mixes small and full-size problems.  The bench line
now also reports the unit counts and the launches per family of its first timed window (`units_timed_window`); with every further
leg switched off that window's launches are the LAST launches of the process, so the last `launches[f]` dispatches of each kernel of
family f, divided by the window's units, are the steady-state figure.  Used by tools/pmc_traffic.sh and tools/pmc_valu_step.sh.

  pmc_reduce.py <dir with *counter_collection.csv> <bench line json> <COUNTER> [...]  -> JSON on stdout:
     {family: {"unit", "units", "launches", COUNTER: total over the window's launches, ...}}"""
import collections
import csv
import glob
import json
import re
import sys

FAMILIES = {"local_ba": (["k_local_ba_t<0", "k_ba_build", "k_dmap_ba_gather", "k_dmap_ba_scatter"], "job", lambda u: u["ba_calls"]),
            "ba_solve": (["k_local_ba_t<0"], "job", lambda u: u["ba_calls"]),
            "lk": (["k_lk"], "point", lambda u: u["track_pts"] + u["right_pts"]),
            "pose_only": (["k_pose_only<"], "job", lambda u: u["frames"]),
            "pyramid": (["k_pyr_fused<"], "image", lambda u: u["pyr_left"] + u["pyr_right"]),
            "gftt": (["k_gftt_eig3<", "k_gftt_select2"], "image", lambda u: u["gftt_calls"]),
            "triangulate": (["k_triangulate"], "point", lambda u: u["tri_pts"]),
            "map": (["k_dmap_begin", "k_dmap_commit", "k_dmap_refresh", "k_dmap_stereo"], "keyframe", lambda u: u["keyframes"])}
LAUNCH_FAMILY = {"ba_solve": "local_ba", "map": "gftt"}        # whose launch count a family shares (one dispatch per family launch)


def reduce(csv_dir, line, counters):
    u = line["units_timed_window"]
    rows = collections.defaultdict(list)                        # kernel -> [(dispatch id, {counter: value})]
    per = collections.defaultdict(dict)
    for f in glob.glob(csv_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] not in counters:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            key = (name, int(r.get("Dispatch_Id") or r.get("Correlation_Id") or 0))
            per[key][r["Counter_Name"]] = per[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for (name, did), vals in per.items():
        rows[name].append((did, vals))
    out = {}
    for fam, (prefixes, unit, nunits) in FAMILIES.items():
        n_launch = int(u["launches"].get(LAUNCH_FAMILY.get(fam, fam), 0))
        tot = {c: 0.0 for c in counters}
        used = 0
        # a family launch dispatches each of its kernels once; template variants of one kernel (k_dmap_ba_gather<512> / <1024>)
        # share the launches: merge the variants of a prefix, take that prefix' last n_launch dispatches
        for p in prefixes:
            d = sorted((x for k, v in rows.items() if k.startswith(p) for x in v), key=lambda x: x[0])
            take = d[-n_launch * (2 if p == "k_dmap_stereo" else 1):] if n_launch else []      # (stereo_prep + stereo_finish share the prefix)
            used += len(take)
            for _, vals in take:
                for c in counters:
                    tot[c] += vals.get(c, 0.0)
        out[fam] = {"unit": unit, "units": int(nunits(u)), "launches": n_launch, "dispatches_used": used, **tot}
    return out


if __name__ == "__main__":
    line = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    json.dump(reduce(sys.argv[1], line, sys.argv[3:]), sys.stdout, indent=1)

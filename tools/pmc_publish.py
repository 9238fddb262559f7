#!/usr/bin/env python3
"""profiles/pmc_valu.json from the PMC summaries of tools/pmc_ba.sh (256 local-BA problems per launch) and tools/pmc_lk.sh
(512 x 150 points): VALU wave-instructions per unit, VALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x the kernel's
cycles, GRBM_GUI_ACTIVE / 8 XCDs), waves waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; stamped with svslam_build_info().

  python tools/pmc_publish.py <pmc_ba summary.txt> <pmc_lk summary.txt> <source tag> > profiles/pmc_valu.json   (on the GPU box)"""
import importlib
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def read(path):
    out = {}
    for ln in open(path):
        m = re.match(r"(\w+)\s+per-launch\s+([0-9.eE+-]+)", ln)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def main():
    ba, lk, tag = read(sys.argv[1]), read(sys.argv[2]), sys.argv[3]
    svs = importlib.import_module("stereovision-slam_amd")
    busy = lambda d: round(d["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * d["GRBM_GUI_ACTIVE"] / 8), 3) if d.get("GRBM_GUI_ACTIVE") else None
    wait = lambda d: round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3) if d.get("SQ_WAVE_CYCLES") else None
    out = {"_comment": __doc__.split("\n\n")[0].replace("\n", " "), "build_info": svs.load().svslam_build_info().decode(),
           "local_ba": {"unit": "problem", "valu_insts_per_unit": round(ba["SQ_INSTS_VALU"] / 256), "valu_busy": busy(ba), "waves_waiting": wait(ba),
                        "source": "%s (tools/pmc_ba.sh, 256 problems per launch)" % tag},
           "lk": {"unit": "point", "valu_insts_per_unit": round(lk["SQ_INSTS_VALU"] / (512 * 150)), "valu_busy": busy(lk), "waves_waiting": wait(lk),
                  "source": "%s (tools/pmc_lk.sh, 512 x 150 points)" % tag}}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()

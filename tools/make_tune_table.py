#!/usr/bin/env python3
"""Produce airv2x_perception_amd/tuned_gfx950.json: run the four models at 2..8 agents once on an MI355X so that the
autotuner sees every conv shape of the shipped configurations, and dump the picks.  Every candidate inside a numerics
class is bit-identical, so the table only saves first-frame tuning time (and pins speed, not results).
    AV2X_TUNE_CACHE=0 python tools/make_tune_table.py gpurun_out/tuned_gfx950.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AV2X_TUNE_CACHE"] = "0"
import torch  # noqa: E402

import bench  # noqa: E402


def main(out):
    dev = torch.device("cuda", 0)
    table = {}
    runs = [(model, agents, None) for model in ("where2com", "cobevt", "v2xvit", "when2com", "v2vnet")
            for agents in ((1, 2, 3, 4, 5, 8) if model == "where2com" else (4, 8))]
    runs.append(("where2com", 8, "cam,lidar"))          # BASELINE configs[4]: the camera trunk's shapes
    for model, agents, mods in runs:
            a = bench.parse(["--model", model, "--agents", str(agents)] + (["--modalities", mods] if mods else []))
            hy, args, dd, _, _ = bench.build_inputs(agents, a.points, dev, only=None, model=model, modalities=(tuple(mods.split(",")) if mods else ("lidar",)))
            m, eng, _ = bench.make_model(a, args, dev)
            for amp, split3 in (((False, False),) if mods else ((False, False), (True, False), (False, True))):
                eng.amp, m.amp, eng.split3 = amp, amp, split3
                m(dd)
                torch.cuda.synchronize()
            eng.amp, m.amp, eng.split3, eng.throughput_mode = False, False, False, True   # the frames-in-flight candidate set
            m(dd)
            torch.cuda.synchronize()
            for k, v in eng.tile_cache.items():
                table["|".join(str(x) for x in k)] = [int(v[0]), int(v[1])]
            print(model, agents, len(table), flush=True)
            del m, eng
            torch.cuda.empty_cache()
    json.dump(table, open(out, "w"), indent=0, sort_keys=True)
    print("wrote", out, len(table))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tuned_gfx950.json")

#!/usr/bin/env python3
"""cfg4 through map plans, HBM-cold (four lens pairs in turn), bilinear: the gather kernel against the LDS-staged kernel over its tile
options.  python tests/tools/bench_cfg4_stage.py [--steps 40]"""
import argparse
import json
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "360cam-pgm-3dgs-tools_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "tests" / "tools"))

import gs360  # noqa: E402
import bench_configs as bc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--interp", type=int, default=1, help="1 linear (the staged kernel's), 2 cubic (always the gather kernel)")
    ap.add_argument("--variants", default="0:32:0,1:32:0,1:32:2,1:16:0,1:8:0")
    args = ap.parse_args()
    ctx = gs360.Context(0, n_slots=1)
    for var in args.variants.split(","):
        stage, rows, wgs = (int(x) for x in var.split(":"))
        with ctx.options(table_stage=stage, table_stage_rows=rows, table_stage_wgs=wgs):
            for r in bc.cfg4_rows(ctx, args.steps, interps=((args.interp, "linear" if args.interp == 1 else "cubic"),)):
                if "plans" in r["key"]:
                    r.update(table_stage=stage, rows=rows, wgs=wgs, staged_jobs=ctx.get_option("last_table_kernel"))
                    print(json.dumps(r), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()

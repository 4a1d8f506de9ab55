"""Times only bench.py's MtlTabNet leg (BASELINE.json configs[4], table-structure half) on the bench's own pages and table regions: the loop for the
decoder's launch-chain work (PT_MTL_ROWFUSED=0|1 A/B, profiles/r05/mtl_rowfused.txt).  usage: [PT_MTL_LEG_MODES=bf16,bf16_kv8] python tools/mtl_leg.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    sys.argv = [sys.argv[0], "--stages", "layout,tsr"]
    args = bench.parse_args() if hasattr(bench, "parse_args") else None
    if args is None:
        raise SystemExit("bench.py has no parse_args()")
    runner = bench.HipRunner(args, 0, 0, 1, None)
    runner.run(2)
    modes = tuple(os.environ.get("PT_MTL_LEG_MODES", "bf16,bf16_kv8,bf16x3").split(","))
    leg = runner.mtl_tabnet_leg(steps=steps, warm=1, modes=modes)
    print(json.dumps({k: ({a: (round(b, 2) if isinstance(b, float) else b) for a, b in v.items() if a != "asserted_by"} if isinstance(v, dict) else v) for k, v in leg.items() if k not in ("note", "asserted_by")}))


if __name__ == "__main__":
    main()

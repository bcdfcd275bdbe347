#!/usr/bin/env python
"""Run the detector over a directory of .bin frames, like the reference executable's `-d` mode
(src/dsvt-ai-trt.cpp:1800-1960):

    python tools/detect.py --data data/bin --out data/outputs [--wts dsvt.wts] [--fp16] [--ref-caps]

One <frame>.txt per <frame>.bin, in the reference's save_txt layout (include/helper.h:441-468)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g                                   # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data", required=True, help="directory of .bin point clouds (float32 x, y, z, intensity)")
    ap.add_argument("--out", required=True, help="directory for the result .txt files")
    ap.add_argument("--wts", default=None, help="weights in the reference's .wts text format")
    ap.add_argument("--seed", type=int, default=1234, help="seed of the synthetic weights used without --wts")
    ap.add_argument("--fp32", action="store_true", help="the default since round 5: the reference's arithmetic is fp32 (fp32-grade split-precision frame)")
    ap.add_argument("--fp16", action="store_true", help="fp16 operands, fp32 accumulation (BASELINE configs[2]; boxes 2e-3 .. 4e-3 from the fp32 oracle on z / size)")
    ap.add_argument("--ref-caps", action="store_true", help="the reference's compile-time caps (50000 points, 10000 pillars)")
    args = ap.parse_args()
    pkg = g.load_package()
    caps = pkg.pipeline.Caps.reference() if args.ref_caps else None
    weights = pkg.detect.load_weights(args.wts, args.seed)
    pkg.detect.run_directory(args.data, args.out, weights, caps=caps, fp16=args.fp16 and not args.fp32)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""train_semi.py -- same CLI and YAML surface as the reference's train_semi.py (--config --local_rank --seed --port),
running the MI355X-native step (u2pl_amd).  Launch: python -m torch.distributed.run --nproc-per-node N
--master-addr 127.0.0.1 --master-port P train_semi.py --config=config.yaml --seed 2 --port P"""
import argparse
import os
import sys

import yaml

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from u2pl_amd.engine import run  # noqa: E402


def main():
    parser = argparse.ArgumentParser(description="Semi-Supervised Semantic Segmentation (MI355X-native U2PL)")
    parser.add_argument("--config", type=str, default="config.yaml")
    parser.add_argument("--local_rank", type=int, default=0)
    parser.add_argument("--seed", type=int, default=0)
    parser.add_argument("--port", default=None, type=int)
    args = parser.parse_args()
    cfg = yaml.load(open(args.config, "r"), Loader=yaml.Loader)
    return run(cfg, args, semi=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Entry point with the reference's CLI (reference run_exp.py:23-52): parse -> Experiment -> forward().

    python run_exp.py --model_type RadLIF --nb_hiddens 1024 --dataset_name ssc --synthetic 1 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 run_exp.py ...   (data parallel)
"""
import argparse

from sparch.exp import Experiment
from sparch.parsers.model_config import add_model_options
from sparch.parsers.training_config import add_training_options


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Model training on spiking speech commands datasets.")
    add_model_options(parser)
    add_training_options(parser)
    return parser.parse_args(argv)


def main(argv=None):
    Experiment(parse_args(argv)).forward()


if __name__ == "__main__":
    main()

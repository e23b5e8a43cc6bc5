"""Command line of the reference's entry points (upsnet/config/parse_args.py:17-31): `--cfg <experiment yaml>` is merged into the
global config before anything else is parsed; `--eval_only`, `--weight_path` as in the reference. Extra options of this
implementation (synthetic workload, steps) are accepted too, so that one parser serves both spellings of the entry point."""
import argparse

from .config import update_config


def parse_args(description='', argv=None):
    parser = argparse.ArgumentParser(description=description)
    parser.add_argument('--cfg', help='experiment configure file name', required=False, type=str, default='')
    parser.add_argument('--eval_only', help='if only eval existing results', action='store_true')
    parser.add_argument('--weight_path', help='manually specify model weights', type=str, default='')
    parser.add_argument('--workload', help='synthetic workload (upsnet_amd.upsnet_end2end_test.WORKLOADS): the config preset when --cfg is '
                        'absent (default upsnet50_cityscapes_1024x2048); with --cfg only its image sizes', type=str, default='')
    parser.add_argument('--synthetic_weights', action='store_true',
                        help='with --cfg and no --weight_path: seeded synthetic weights instead of the default checkpoint path')
    parser.add_argument('--steps', type=int, default=20)
    parser.add_argument('--warmup', type=int, default=10)
    parser.add_argument('--in-flight', type=int, default=1, help='images in flight per rank (1 = the reference loop)')
    args, rest = parser.parse_known_args(argv)
    if args.cfg:
        update_config(args.cfg)
    if rest:    # (the reference re-parses strictly, parse_args.py:32; an option neither parser knows is reported, not swallowed)
        parser.error('unrecognized arguments: %s' % ' '.join(rest))
    return args

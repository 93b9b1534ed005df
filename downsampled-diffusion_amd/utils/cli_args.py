"""Command-line surface of train.py (reference utils/cli_args.py:4-83): -m -d -e -bs -is -mute -downsample.

Superset flags (not in the reference): -T overrides the number of diffusion steps that the
reference fixes in train.py:26, --n_samples the size of the logged sample grid (reference: 25, which forces
batch_size >= 25, trainers/trainer.py:50-51).
"""
import argparse


def get_args(config: dict, data_names: list, model_names: list, argv=None) -> tuple:
    ap = argparse.ArgumentParser(description="Model training script.")
    ap.add_argument('-m', dest='model', type=str, default=model_names[0], choices=model_names,
                    help=f'Pick which model to train (default: {model_names[0]}).')
    ap.add_argument('-d', dest='dataset', type=str, default=data_names[0], choices=data_names,
                    help=f'Pick which dataset to fit to (default: {data_names[0]}).')
    ap.add_argument('-e', dest='n_steps', type=int, default=500,
                    help='Pick number of epochs/trainsteps to train over (default: 500).')
    ap.add_argument('-bs', dest='batch_size', type=int, default=32, help='Pick batch size of data.')
    ap.add_argument('-is', dest='image_size', type=int, default=32, help='Pick image size of data.')
    ap.add_argument('-mute', action='store_true', help='Mute tqdm and other print outputs.')
    if 'ddpm' in model_names:
        ap.add_argument('-downsample', dest='n_downsamples', type=int, default=0,
                        help='Determine how many downsamples (x2) to perform. When 0, run standard DDPM.')
    ap.add_argument('-T', dest='T_override', type=int, default=None, help='(extension) number of diffusion steps.')
    ap.add_argument('--n_samples', dest='n_samples', type=int, default=None,
                    help='(extension) images in the logged sample grid; must be a square number <= batch size.')
    args = ap.parse_args(argv)
    for key, value in vars(args).items():
        if key in ('mute', 'n_runs'):
            continue
        if key in ('T_override', 'n_samples') and value is None:
            continue
        config[key] = value
    if config['model'] != 'ddpm':
        config.pop('n_downsamples', None)
    return config, args.mute

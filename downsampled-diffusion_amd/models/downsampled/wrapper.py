"""Factories for the dDDPM resampling networks (reference models/downsampled/wrapper.py:6-59)."""
from .convblocks import ConvResNet

_ONLY = ("only 'convolutional_res' is built on the HIP path -- the mode train.py:34-35 selects; "
         "'deterministic' / 'convolutional' are outside the accelerated scope (SURVEY.md section 2)")


def _common(config, shape):
    assert shape[1] == shape[2]
    assert shape[0] == 1 or shape[0] == 3
    return shape[0], config['d_chans'], config['unet_in'], config['d_dropout'], config['n_downsamples']


def get_upsampling(config: dict, shape: tuple):
    """latent (unet_in channels) -> image (shape[0] channels), wrapper.py:6-30"""
    img_ch, dim, lat_ch, dropout, n_down = _common(config, shape)
    if config['u_mode'] != 'convolutional_res':
        raise NotImplementedError(f'Upsampling method "{config["u_mode"]}": {_ONLY}')
    return ConvResNet(dim, lat_ch, img_ch, n_down, upsample=True, dropout=dropout, n_blocks=config['u_n_blocks'])


def get_downsampling(config: dict, shape: tuple):
    """image -> latent, wrapper.py:33-59"""
    img_ch, dim, lat_ch, dropout, n_down = _common(config, shape)
    if config['d_mode'] != 'convolutional_res':
        raise NotImplementedError(f'Downsampling method "{config["d_mode"]}": {_ONLY}')
    return ConvResNet(dim, img_ch, lat_ch, n_down, upsample=False, dropout=dropout, n_blocks=config['d_n_blocks'])

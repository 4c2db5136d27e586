"""Model hyper-parameters -- same attribute names and as-shipped values as the reference's
``configs/models_config.py`` (:3-31), read by ``models.vae_gan`` at construction time.

The reference is switched between resolutions by (un)commenting blocks; here the same can be done
programmatically with ``use_px64()`` / ``use_px100()`` / ``use_px128()`` before the models are built.
"""

kernel_size = 5
stride = 2
padding = 2
dropout = 0.7

encoder_channels = [64, 128, 256]
decoder_channels = [256, 128, 32, 3]
discrim_channels = [32, 128, 256, 256, 512]

# paper settings (100 x 100, latent 512) -- the reference ships with these active
image_size = 100
fc_input = 13          # 8/13/14/16/28 for image_size = 64/100/112/128/224
fc_output = 1024
fc_input_gan = 7
fc_output_gan = 256
stride_gan = 2
latent_dim = 512
output_pad_dec = [False, True, True]
decoder_channels = [256, 128, 64, 3]


def _set(**kw):
    globals().update(kw)


def use_px100():
    _set(image_size=100, fc_input=13, fc_output=1024, fc_input_gan=7, fc_output_gan=256, stride_gan=2,
         latent_dim=512, output_pad_dec=[False, True, True], decoder_channels=[256, 128, 64, 3])


def use_px64():
    """The commented 'settings for resolution 64' block of the reference (BASELINE.json configs)."""
    _set(image_size=64, fc_input=8, fc_output=1024, fc_input_gan=8, fc_output_gan=512, stride_gan=1,
         latent_dim=128, output_pad_dec=[True, True, True], decoder_channels=[256, 128, 32, 3])


def use_px128():
    _set(image_size=128, fc_input=16, fc_output=1024, fc_input_gan=8, fc_output_gan=512, stride_gan=2,
         latent_dim=128, output_pad_dec=[True, True, True], decoder_channels=[256, 128, 32, 3])

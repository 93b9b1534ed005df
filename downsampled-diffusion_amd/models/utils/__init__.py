from .helpers import exists, default, extract, noise_like, get_ones_like, get_identity_like
from .losses import l1_loss, l2_loss, normal_kl, discretized_gaussian_log_likelihood

"""Differentiable (forward + backward HIP kernels) path for training.  Placeholder entry points: the
backward kernel set is not built yet, so these fail loudly rather than fall back."""
from ddk.lib import DDKError

_MSG = ("the training (autograd) path needs the HIP backward kernels, which are not built yet; "
        "wrap inference in torch.no_grad() / call .eval()")


def unet_forward_autograd(unet, x, time):
    raise DDKError("Unet: " + _MSG)


def sq_err_sum_autograd(eps, eps_hat):
    raise DDKError("loss: " + _MSG)


def resnet_forward_autograd(net, x, final_tanh):
    raise DDKError("ConvResNet: " + _MSG)

"""ctypes wrappers of include/dvs_train.h (loss gradient, SSIM, fused Adam) on torch CUDA tensors."""
import ctypes as C
import torch
from ._lib import lib, check, AdamGroup


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def l1_loss_grad(img, target):
    """-> (dL [like img], loss scalar tensor): mean |img - target| and its gradient."""
    dL = torch.empty_like(img)
    loss = torch.zeros(1, dtype=torch.float32, device=img.device)
    check(lib.dvs_l1_loss_grad(_st(), img.data_ptr(), target.data_ptr(), img.numel(), dL.data_ptr(), loss.data_ptr()), "dvs_l1_loss_grad")
    return dL, loss


class Ssim:
    """SSIM (11x11 Gaussian window, sigma 1.5, zero padding) of [3,H,W] images with reusable scratch maps."""

    def __init__(self, width, height, device):
        self.W, self.H = width, height
        self.maps = [torch.empty((3, height, width), dtype=torch.float32, device=device) for _ in range(3)]
        self.sum = torch.zeros(4096, dtype=torch.float32, device=device)      # DVS_SSIM_SLOTS partial sums

    def forward(self, img, target):
        """-> mean SSIM as a 1-element tensor (asynchronous)."""
        self.sum.zero_()
        check(lib.dvs_ssim_forward(_st(), img.data_ptr(), target.data_ptr(), self.W, self.H, self.maps[0].data_ptr(),
                                   self.maps[1].data_ptr(), self.maps[2].data_ptr(), self.sum.data_ptr()), "dvs_ssim_forward")
        return self.sum.sum().reshape(1) / (3.0 * self.W * self.H)

    def loss_backward(self, img, target, ssim_weight):
        """After forward(): dL of (1-w) mean|x-y| + w (1 - mean SSIM) in one pass (dvs_loss_l1_ssim_backward).
        -> (dL [3,H,W], l1 term as a 1-element tensor = (1-w) mean|x-y|)."""
        dL = torch.empty_like(img)
        l1 = torch.zeros(4096, dtype=torch.float32, device=img.device)            # DVS_SSIM_SLOTS
        check(lib.dvs_loss_l1_ssim_backward(_st(), img.data_ptr(), target.data_ptr(), self.W, self.H, self.maps[0].data_ptr(),
                                            self.maps[1].data_ptr(), self.maps[2].data_ptr(), float(ssim_weight), dL.data_ptr(), l1.data_ptr()),
              "dvs_loss_l1_ssim_backward")
        return dL, l1.sum().reshape(1)

    def backward(self, img, target, dL, scale, accumulate=True):
        """dL (+)= scale * d(mean SSIM)/d(img)."""
        check(lib.dvs_ssim_backward(_st(), img.data_ptr(), target.data_ptr(), self.W, self.H, self.maps[0].data_ptr(),
                                    self.maps[1].data_ptr(), self.maps[2].data_ptr(), float(scale), dL.data_ptr(), int(accumulate)),
              "dvs_ssim_backward")
        return dL


def adam_step(param, grad, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-15):
    check(lib.dvs_adam_step(_st(), param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(), float(lr), float(beta1),
                            float(beta2), float(eps), int(step)), "dvs_adam_step")


def adam_step_groups(groups, step, beta1=0.9, beta2=0.999, eps=1e-15, visible=None):
    """All parameter groups in one launch. groups: iterable of dicts {param, grad, m, v, lr, width[, tiled, active_chunks]};
    visible: optional int32 [n] radii tensor -> only splats with radii > 0 are updated (the reference's visibleAdam)."""
    arr = (AdamGroup * len(groups))()
    for a, g in zip(arr, groups):
        a.param, a.grad, a.m, a.v = g["param"].data_ptr(), g["grad"].data_ptr(), g["m"].data_ptr(), g["v"].data_ptr()
        a.count, a.lr, a.width = g["param"].numel(), float(g["lr"]), int(g["width"])
        a.layout, a.active_chunks = int(bool(g.get("tiled", False))), int(g.get("active_chunks", 0))
    check(lib.dvs_adam_step_groups(_st(), arr, len(groups), float(beta1), float(beta2), float(eps), int(step),
                                   visible.data_ptr() if visible is not None else None, int(visible.numel()) if visible is not None else 0),
          "dvs_adam_step_groups")

"""Oracle restatement of layers.py::SSIM (215-245), get_smooth_loss (199-212), and the SID helpers
utils.py::get_labels_sid (147-175) / get_depth_sid (106-133).  TEST INFRASTRUCTURE ONLY."""
import torch
import torch.nn.functional as F

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def ssim(x, y):
    """clamp((1 - SSIM)/2, 0, 1) with 3x3 mean windows over a 1-px reflection-padded image."""
    x = F.pad(x, (1, 1, 1, 1), mode="reflect")
    y = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(x, 3, 1)
    mu_y = F.avg_pool2d(y, 3, 1)
    sig_x = F.avg_pool2d(x * x, 3, 1) - mu_x ** 2
    sig_y = F.avg_pool2d(y * y, 3, 1) - mu_y ** 2
    sig_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    d = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def get_smooth_loss(disp, img):
    """Edge-aware first-order smoothness: mean(|d_x disp| e^{-mean_c|d_x img|}) + same in y."""
    gdx = (disp[:, :, :, :-1] - disp[:, :, :, 1:]).abs()
    gdy = (disp[:, :, :-1, :] - disp[:, :, 1:, :]).abs()
    gix = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True)
    giy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True)
    return (gdx * torch.exp(-gix)).mean() + (gdy * torch.exp(-giy)).mean()


def _sid_beta(dataset):
    if dataset == "kitti":
        return 80.999
    if dataset in ("nyu", "NYU"):
        return 10.999
    raise ValueError("undefined dataset %r" % (dataset,))


def get_labels_sid(depth, ordinal_c=71.0, dataset="kitti"):
    """int(K * log((d + 0.999)/1) / log(beta/1)), truncation toward zero, int32."""
    k = torch.tensor(float(ordinal_c))
    alpha = torch.tensor(1.0)
    beta = torch.tensor(_sid_beta(dataset))
    return (k * torch.log((depth + 0.999) / alpha) / torch.log(beta / alpha)).int()


def get_depth_sid(labels, ordinal_c=71.0, dataset="kitti"):
    """0.5 (beta^{l/K} + beta^{(l+1)/K}) - 0.999 as float32."""
    k = torch.tensor(float(ordinal_c))
    alpha = torch.tensor(1.0)
    beta = torch.tensor(_sid_beta(dataset))
    lf = labels.float()
    d = 0.5 * (alpha * (beta / alpha) ** (lf / k) + alpha * (beta / alpha) ** ((lf + 1.0) / k)) - 0.999
    return d.float()

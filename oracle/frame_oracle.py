"""CPU restatement of the dataset front-end's per-frame arithmetic — TEST INFRASTRUCTURE ONLY (imported by tests/).

Follows /root/reference/dataset_loaders/seven_scenes.py:324-352: `img/255` -> `cv2.resize(..., INTER_AREA)` ->
ToTensor -> Y = 0.299R + 0.587G + 0.114B (rgb_to_yuv) -> `torch.histc(bins, 0, 1)` -> percentages -> `torch.round`.
cv2 (opencv-python, requirements.txt) is third-party and absent from this image, so INTER_AREA is restated from its
definition — the mean of the source image over the output pixel's box, source pixels weighted by their covered
area — and is PARITY UNPINNED against cv2 itself; for integer factors that definition is the plain block mean."""
import numpy as np
import torch


def area_downscale(img_u8, H, W):
    """uint8 [h,w,3] -> float64 [H,W,3] in [0,1]: coverage-weighted box mean, explicit loops (small inputs only)."""
    h, w = img_u8.shape[:2]
    src = img_u8.astype(np.float64) / 255.0
    sy, sx = h / H, w / W
    out = np.zeros((H, W, 3))
    for y in range(H):
        y0, y1 = y * sy, (y + 1) * sy
        for x in range(W):
            x0, x1 = x * sx, (x + 1) * sx
            acc = np.zeros(3)
            for j in range(int(y0), min(h, int(np.ceil(y1)))):
                wy = min(y1, j + 1) - max(y0, j)
                for i in range(int(x0), min(w, int(np.ceil(x1)))):
                    acc += wy * (min(x1, i + 1) - max(x0, i)) * src[j, i]
            out[y, x] = acc / (sy * sx)
    return out


def luma_histogram(img_chw, bins=10):
    y = 0.299 * img_chw[0] + 0.587 * img_chw[1] + 0.114 * img_chw[2]
    h = torch.histc(y, bins=bins, min=0., max=1.)
    return torch.round(h / h.sum() * 100)

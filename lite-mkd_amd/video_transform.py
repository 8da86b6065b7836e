"""GPU side of the episode's frame transform.

Reference (host, per video, PIL): video_reader.py:92-112 builds Compose([Resize(256), RandomHorizontalFlip(), RandomCrop(224)])
for training and Compose([Resize(256), CenterCrop(224)]) for testing (96/84 for the small image size), then ToTensor per frame
and torch.stack (video_reader.py:377-385).  Here the decoded uint8 frames of ALL videos of an episode go to the GPU once
(a quarter of the bytes of the fp32 tensors) and three kernels do the rest: Pillow's BILINEAR resampler bit for bit
(lmkd_resize_pass_u8, two passes) and crop + flip + ToTensor straight into the stem's NHWC4 layout (lmkd_frames_u8_to_nhwc4).
The random draws use Python's `random` in the reference's order (per video: flip = random.random() < 0.5, then
x1 = randint(0, W - w), y1 = randint(0, H - h); video_transforms.py:45,168-169), so a seeded run picks the same augmentations."""
import random

import torch

from . import ops

_RESIZE_FOR = {84: 96, 224: 256}          # video_reader.py:96-101


class GpuFrameTransform:
    def __init__(self, img_size, device, resize=None):
        if resize is None:
            if img_size not in _RESIZE_FOR:
                raise ValueError("img size transforms not setup")      # video_reader.py:103-104 prints this and exits
            resize = _RESIZE_FOR[img_size]
        self.img_size, self.resize, self.device = img_size, resize, device

    def resized_hw(self, h, w):
        s = self.resize                                                # functional.py:45-52,66-73
        if (w <= h and w == s) or (h <= w and h == s):
            return h, w
        return (int(s * h / w), s) if w < h else (s, int(s * w / h))

    def draw(self, h, w, train):
        """-> (flip, x1, y1) for one video whose decoded frames are h x w, consuming `random` like the reference"""
        oh, ow = self.resized_hw(h, w)
        S = self.img_size
        if S > ow or S > oh:
            raise ValueError("Initial image size should be larger then cropped size but got cropped sizes : (%d, %d) while "
                             "initial image is (%d, %d)" % (S, S, ow, oh))
        if train:
            flip = random.random() < 0.5                               # RandomHorizontalFlip
            x1 = random.randint(0, ow - S)                             # RandomCrop
            y1 = random.randint(0, oh - S)
        else:
            flip = False
            x1 = int(round((ow - S) / 2.))                             # CenterCrop
            y1 = int(round((oh - S) / 2.))
        return flip, x1, y1

    def __call__(self, videos, train=True, params=None):
        """videos: list of uint8 tensors [L, H, W, 3] (decoded frames, host or device; resolutions may differ between videos).
        -> float32 NHWC4 [sum L, S, S, 4] in video order (what the backbones accept in place of [F, 3, S, S])."""
        if not videos:
            raise ValueError("empty episode")
        L = videos[0].shape[0]
        if params is None:
            params = [self.draw(v.shape[1], v.shape[2], train) for v in videos]
        S = self.img_size
        out = torch.empty((len(videos) * L, S, S, 4), dtype=torch.float32, device=self.device)
        groups = {}
        for i, v in enumerate(videos):
            if v.dtype != torch.uint8 or v.dim() != 4 or v.shape[0] != L or v.shape[3] != 3:
                raise RuntimeError("every video must be uint8 [%d, H, W, 3]" % L)
            groups.setdefault((v.shape[1], v.shape[2]), []).append(i)
        for (h, w), idx in groups.items():                            # one resize + one crop launch per source resolution
            frames = torch.cat([videos[i].to(self.device, non_blocking=True) for i in idx], 0).contiguous()
            r = ops.resize_frames_u8(frames, self.resize)
            i32 = lambda vals: torch.tensor(vals, dtype=torch.int32, device=self.device)      # noqa: E731
            ow = r.shape[2]
            # the reference flips the whole resized frame and THEN crops at x1; the kernel crops and then mirrors inside the
            # crop, so a flipped video reads columns [ow - S - x1, ow - x1)
            cx = [(ow - S - params[i][1]) if params[i][0] else params[i][1] for i in idx]
            x = ops.frames_u8_to_nhwc4(r, i32([params[i][2] for i in idx]), i32(cx), i32([int(params[i][0]) for i in idx]), S,
                                       frames_per_video=L)
            for j, i in enumerate(idx):
                out[i * L:(i + 1) * L] = x[j * L:(j + 1) * L]
        return out

    def batch(self, frames_u8, params, frames_per_video=8, out=None):
        """the same transform for an episode whose decoded frames share one resolution and already sit in ONE uint8 device tensor
        [F, H, W, 3] (one H2D copy of the whole episode): resize + crop / flip / ToTensor in one launch + the crop parameters.
        params: [(flip, x1, y1)] per video (draw()).  out: optional preallocated [F, S, S, 4] fp32 tensor (static input buffers)."""
        S = self.img_size
        _, ow = ops._resized_shape(frames_u8.shape[1], frames_u8.shape[2], self.resize)
        i32 = lambda vals: torch.tensor(vals, dtype=torch.int32).pin_memory().to(self.device, non_blocking=True)      # noqa: E731
        cx = [(ow - S - p[1]) if p[0] else p[1] for p in params]
        # one launch (round 5; before: two resize passes + the crop kernel, 1.08 ms of a streamed episode's copy-stream work)
        return ops.frames_resize_crop_nhwc4(frames_u8, self.resize, i32([p[2] for p in params]), i32(cx), i32([int(p[0]) for p in params]), S,
                                            frames_per_video=frames_per_video, out=out)


def load_teacher_feature(path):
    """video_reader.py:393-394: the fused teacher feature of one video, `feature.npy` [8, 2048] -> tensor"""
    import numpy as np
    return torch.from_numpy(np.load(path))

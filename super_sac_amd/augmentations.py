"""DrQ-family augmentations on the device (mirror of super_sac/augmentations.py:20-41,165-293,
489-503).  One randomisation per call is shared by every batch passed (s and s' get the same
shift).  The pixel work runs in the ``ssac_drq_shift`` kernel; when used through
``learning_utils.sample_move_and_augment`` the kernel reads the uint8 replay rows directly
(gather + uint8->fp32 + shift + aug_mix row selection fused, one pass over the pixels).
"""
import torch

from . import engine, rng
from ._lib import check, lib


class _ShiftAug:
    MODE = 0

    def __init__(self, batch_size, pad=4, noise=False, *_a, **_k):
        self.batch_size, self.pad, self.noise = batch_size, pad, noise
        self._shift_dev = None
        self.change_randomization_params()

    def _upload(self, shift_xy):
        self._shift_host = shift_xy.contiguous()
        self._shift_dev = None

    def _adopt_state(self):
        """an object built by the reference's class of the same name (adopt.adopt_augmenter swapped its class): take
        over the randomisation it currently holds, without drawing"""
        raise NotImplementedError

    def shift_device(self, device):
        if self._shift_dev is None or self._shift_dev.device != device:
            self._shift_dev = self._shift_host.to(device)
        return self._shift_dev

    def apply(self, src, idx, n, c, h, n_aug, dst, noise=None):
        assert n == self.batch_size
        check(lib.ssac_drq_shift(src.data_ptr(), 1 if src.dtype == torch.uint8 else 0,
                                 0 if idx is None else idx.data_ptr(), n, c, h, self.pad,
                                 self.shift_device(dst.device).data_ptr(), self.MODE,
                                 0 if noise is None else noise.data_ptr(), n_aug, dst.data_ptr(),
                                 engine.stream()))
        return dst

    def __call__(self, imgs):
        engine.require_gpu(imgs)
        n, c, h, w = imgs.shape
        assert h == w
        assert n == self.batch_size
        imgs = imgs.contiguous()
        noise = rng.draw_normal((n, c, h, w), imgs.device) if self.noise else None
        return self.apply(imgs, None, n, c, h, n, torch.empty_like(imgs, dtype=torch.float32), noise)


class Drqv2Aug(_ShiftAug):
    """replicate-pad + bilinear grid shift (augmentations.py:214-269)."""
    MODE = 0

    def change_randomization_params(self):
        self.shift = rng.draw_drqv2_shift(self.batch_size, self.pad)  # (B,1,1,2): x, y
        self._upload(self.shift.reshape(self.batch_size, 2).to(torch.int64))

    def _adopt_state(self):
        self._upload(self.shift.reshape(self.batch_size, 2).to(torch.int64))

    def __repr__(self):
        return "DrqV2"


class DrqAug(_ShiftAug):
    """reflection-pad + integer crop (+ N(0,1) noise) (augmentations.py:165-211)."""
    MODE = 1

    def __init__(self, batch_size, pad=4, noise=True, *_a, **_k):
        super().__init__(batch_size, pad, noise)

    def change_randomization_params(self):
        self.w1, self.h1 = rng.draw_drq_offsets(self.batch_size, self.pad)
        self._upload(torch.stack([self.w1, self.h1], dim=1).to(torch.int64))

    def _adopt_state(self):
        self._upload(torch.stack([self.w1, self.h1], dim=1).to(torch.int64))

    def __repr__(self):
        return "Drqv1"


class DrqNoNoiseAug(DrqAug):
    def __init__(self, batch_size, pad=4, noise=False, *_a, **_k):
        super().__init__(batch_size, pad, noise)

    def __repr__(self):
        return "Drqv1NoNoise"


class LargeDrqNoNoiseAug(DrqAug):
    def __init__(self, batch_size, pad=12, noise=False, *_a, **_k):
        super().__init__(batch_size, pad, noise)

    def __repr__(self):
        return "Drqv1LargeNoNoise"


class LargeDrqAug(DrqAug):
    def __init__(self, batch_size, pad=12, *_a, **_k):
        super().__init__(batch_size, pad)

    def __repr__(self):
        return "Drqv1Large"


class IdentityAug:
    def __init__(self, batch_size, *_a, **_k):
        self.batch_size = batch_size

    def __call__(self, imgs):
        return imgs

    def change_randomization_params(self):
        return

    def _adopt_state(self):
        return

    def __repr__(self):
        return "Identity"


class AugmentationSequence:
    def __init__(self, aug_list, keys=None):
        self.aug_list = aug_list
        self.keys = keys

    def is_identity(self):
        return all(isinstance(a, IdentityAug) for a in self.aug_list)

    def single_shift(self):
        """the one shift-type augmentation this sequence consists of (fusable), else None."""
        real = [a for a in self.aug_list if not isinstance(a, IdentityAug)]
        return real[0] if len(real) == 1 and isinstance(real[0], _ShiftAug) else None

    def change_randomization_params(self):
        for aug in self.aug_list:
            aug.change_randomization_params()

    def _augment_one(self, batch):
        """one observation dict through every augmentation, key by key; keys outside `self.keys` pass through as copies"""
        out = {}
        for name, value in batch.items():
            value = value.clone()
            if name in self.keys:
                for aug in self.aug_list:
                    value = aug(value)
            out[name] = value
        return out

    def __call__(self, *batches):
        """augmentations.py:20-38 as a contract: ONE randomisation per call, shared by every batch passed (s and s' get the
        same shift); the batches themselves are left untouched; one batch in -> one dict out, several -> a tuple."""
        if self.keys is None:   # (first call: every key of the first batch, remembered -- as the reference does)
            self.keys = batches[0].keys()
        self.change_randomization_params()
        augmented = tuple(self._augment_one(b) for b in batches)
        return augmented[0] if len(augmented) == 1 else augmented

    def __repr__(self):
        names = [repr(a) for a in self.aug_list]
        return f"AugmentationSequence: ({names})"

"""data_process/pipeline.py of the reference (:85-170): StandardAugmentationsPipeline with the same constructor keywords.  Training:
random scale (one of the discrete factors min..max step, aspect kept) -> pad bottom / right with the mean pixel / ignore label up to the
crop size -> random crop -> random flip -> random erasing; evaluation: pad only.  The reference maps these over a tf.data stream on the
host; here the host only draws the decisions (numpy Generator, seeded) and ONE kernel (csrc/augment.hip) gathers every output pixel of the
batch straight from the source images -- at > 1 000 images/s per GPU a host-side image pipeline would be the bottleneck."""
import numpy as np
import torch

from .. import kernels as K
from .. import nn
from .input_norm import norm_affine
from .input_norm_types import InputNormTypes


class StandardAugmentationsPipeline:
    def __init__(self, training=False, mean_pixel=[127.5, 127.5, 127.5], ignore_label=255, max_resize_height=None, max_resize_width=None,
                 crop_height=513, crop_width=513, eval_crop_height=None, eval_crop_width=None, prob_of_flip=0.5, prob_of_erase=0.5,
                 min_scale_factor=0.5, max_scale_factor=2.0, scale_factor_step_size=0.1, random_brightness=False,
                 photo_metric_distortions=False, random_erase=True, random_jepg_quality=False, random_noisy_eval_level=0, name=None,
                 input_norm_type=InputNormTypes.NONE, seed=0):
        if eval_crop_height is None:
            eval_crop_height = crop_height
        if eval_crop_width is None:
            eval_crop_width = crop_width
        if not training:
            crop_height, crop_width = eval_crop_height, eval_crop_width
        if random_jepg_quality:
            raise NotImplementedError("RandomJEPGQualityAugment (a JPEG codec round trip) is not part of the on-device pipeline "
                                      "(the standard recipe leaves it off)")
        # ResizeAugment (:118-119, augments/resize_augment.py): samples larger than the bound are resampled once in front of everything else
        self.max_resize = (max_resize_height, max_resize_width) if (max_resize_height or max_resize_width) else None
        # optional photometric augmentations (:129-134, :160-164): drawn on the host like every other decision, applied by the gather kernel
        self.random_brightness, self.photo_metric_distortions = bool(random_brightness), bool(photo_metric_distortions)
        self.random_noisy_eval_level = float(random_noisy_eval_level)
        if min_scale_factor < 0 or min_scale_factor > max_scale_factor:
            raise ValueError("Unexpected value of min_scale_factor.")
        self.training, self.name = training, name
        self.mean_pixel, self.ignore_label = [float(v) for v in mean_pixel], int(ignore_label)
        self.target_height, self.target_width = int(crop_height), int(crop_width)
        self.prob_of_flip, self.prob_of_erase, self.random_erase = prob_of_flip, prob_of_erase, random_erase
        self.min_scale_factor, self.max_scale_factor, self.scale_factor_step_size = min_scale_factor, max_scale_factor, scale_factor_step_size
        self.input_norm_type = input_norm_type
        self.rng = np.random.default_rng(seed)
        self._launches = 0

    # ---- the random decisions (host) -------------------------------------------------------------------------------------------------
    def get_random_scale(self):
        """utils.py:303-328: a uniform draw from linspace(min, max, num_steps); step 0 = continuous uniform"""
        lo, hi, step = self.min_scale_factor, self.max_scale_factor, self.scale_factor_step_size
        if lo == hi:
            return float(lo)
        if step == 0:
            return float(self.rng.uniform(lo, hi))
        num_steps = int((hi - lo) / step + 1)
        return float(np.linspace(np.float32(lo), np.float32(hi), num_steps, dtype=np.float32)[self.rng.integers(0, num_steps)])

    def draw(self, sizes):
        """per-sample parameter table [B, iseg_augment_params_ints()] for source sizes [(H, W), ...]"""
        n_int = K.augment_params_ints()
        tab = np.zeros((len(sizes), n_int), dtype=np.int32)
        ch, cw = self.target_height, self.target_width
        for b, (H, W) in enumerate(sizes):
            nH, nW, oy, ox, flip, rects = H, W, 0, 0, 0, []
            if self.training:
                s = self.get_random_scale()
                if s != 1.0:      # tf.cast(tf.cast(h, float32) * scale, int32)
                    nH, nW = int(np.float32(H) * np.float32(s)), int(np.float32(W) * np.float32(s))
                ph, pw = max(nH, ch), max(nW, cw)      # size after PadAugment
                oy = int(self.rng.integers(0, ph - ch + 1))
                ox = int(self.rng.integers(0, pw - cw + 1))
                flip = int(self.rng.random() <= self.prob_of_flip)
                if self.random_erase and self.rng.random() <= self.prob_of_erase:
                    # random_erasing_augment.py:64-106 with min_area_size 0, max_area_size 0.25, 1..5 areas (the pipeline's settings)
                    max_h, max_w = int(np.float32(ch) * np.float32(0.25)), int(np.float32(cw) * np.float32(0.25))
                    for _ in range(int(self.rng.integers(1, 5))):
                        ah = min(max(int(self.rng.integers(0, max(max_h, 1))), 1), ch)
                        aw = min(max(int(self.rng.integers(0, max(max_w, 1))), 1), cw)
                        rects.append((int(self.rng.integers(0, max(ch - ah, 1))), int(self.rng.integers(0, max(cw - aw, 1))), ah, aw))
            tab[b, :8] = [H, W, nH, nW, oy, ox, flip, len(rects)]
            for e, r in enumerate(rects):
                tab[b, 8 + 4 * e:12 + 4 * e] = r
        return tab

    def draw_photometric(self, n):
        """[n, iseg_augment_params_floats()] float32 or None: RandomBrightnessAugment(max_delta 32, p 0.5) (random_brightness_augment.py:12-28);
        RandomPhotoMetricDistortions = contrast U(0.75, 1.25) p 0.5 -> saturation U(0.75, 1.25) p 0.5 -> hue U(-0.1, 0.1) always
        (random_photo_metric_distortions.py:15-37); evaluation: RandomNoisyEvalAugment's stddev (random_noisy_eval_augment.py:12-30)"""
        train_any = self.training and (self.random_brightness or self.photo_metric_distortions)
        noisy = (not self.training) and self.random_noisy_eval_level > 1e-3
        if not (train_any or noisy):
            return None
        tab = np.zeros((n, K.augment_params_floats()), dtype=np.float32)
        tab[:, 1] = 1.0
        tab[:, 5] = 1.0
        for b in range(n):
            if self.training and self.random_brightness and self.rng.random() <= 0.5:
                tab[b, 0] = self.rng.uniform(-32.0, 32.0)
            if self.training and self.photo_metric_distortions:
                if self.rng.random() <= 0.5:
                    tab[b, 1] = self.rng.uniform(0.75, 1.25)
                if self.rng.random() <= 0.5:
                    tab[b, 5] = self.rng.uniform(0.75, 1.25)
                tab[b, 6] = self.rng.uniform(-0.1, 0.1)
            if noisy:
                tab[b, 7] = self.random_noisy_eval_level
        return tab

    def resize_target(self, height, width):
        """ResizeAugment.compuate_target_size (resize_augment.py:15-33) in its float32 arithmetic: never larger than the sample"""
        max_h, max_w = self.max_resize
        f = np.float32
        th = min(int(max_h), height) if max_h else height
        tw = int(f(width) * f(th) / f(height))
        tw = min(int(max_w), tw) if max_w else tw
        tw = min(width, tw)
        th = int(f(height) * f(tw) / f(width))
        return th, tw

    def _resize_to_bound(self, images, labels, sizes):
        """the optional first augmentation: bilinear image / nearest label, sample by sample (the targets differ), back into the padded batch
        buffer (float32 from here on)"""
        out = torch.zeros(images.shape, dtype=torch.float32, device=images.device)
        lab = None if labels is None else labels.clone()
        new_sizes = []
        for b, (H, W) in enumerate(sizes):
            th, tw = self.resize_target(H, W)
            src = images[b:b + 1, :H, :W].to(torch.float32).contiguous()
            if (th, tw) == (H, W):
                out[b, :H, :W] = src[0]
            else:
                out[b, :th, :tw] = K.resize_bilinear(src, th, tw)[0]
                if lab is not None:
                    lab[b, :th, :tw] = K.resize_nearest_i32(labels[b:b + 1, :H, :W, None].contiguous(), th, tw)[0, :, :, 0]
            new_sizes.append((th, tw))
        return out, lab, new_sizes

    # ---- the batch on the device -----------------------------------------------------------------------------------------------------
    def apply_batch(self, images, labels, sizes=None, params=None, photometric="draw"):
        """images [B, Hs, Ws, 3] uint8 / float32 (samples smaller than Hs x Ws sit in the top-left corner, `sizes` = their (H, W)),
        labels [B, Hs, Ws] int32 or None  ->  (float32 [B, crop_h, crop_w, 3] normalised, int32 [B, crop_h, crop_w] or None)"""
        B, Hs, Ws, _ = images.shape
        if sizes is None:
            sizes = [(Hs, Ws)] * B
        if self.max_resize is not None:      # (`params`, when given, were drawn for the sizes AFTER this step: resize_target)
            images, labels, sizes = self._resize_to_bound(images, labels, sizes)
        if params is None:
            params = self.draw(sizes)
        dev = images.device
        scale, shift = norm_affine(self.input_norm_type)
        self._launches += 1
        seed = (nn.seed() * 0x9E3779B97F4A7C15 + self._launches * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        if isinstance(photometric, str):
            photometric = self.draw_photometric(B)
        pdev = torch.from_numpy(params).to(dev)
        fdev = None
        if photometric is not None:
            fdev = torch.from_numpy(np.ascontiguousarray(photometric, dtype=np.float32)).to(dev)
            if bool((photometric[:, 1] != 1.0).any()):      # the contrast step needs the channel means of the scaled image
                K.augment_channel_means(images, pdev, fdev)
        return K.augment_crop_batch(images, labels, pdev, self.mean_pixel, scale, shift, self.ignore_label, self.target_height,
                                    self.target_width, seed, fparams=fdev)

    def __call__(self, ds):
        """dataset -> dataset of augmented samples (batch of one through the same kernel), for code written against the tf.data form"""
        if ds is None:
            return ds

        def one(image, label):
            img, lab = self.apply_batch(image.unsqueeze(0).to(nn.device()), label.unsqueeze(0).to(nn.device()).to(torch.int32))
            return img[0], lab[0]

        return ds.map(one)

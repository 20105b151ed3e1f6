"""Drop-in for the reference's swiftwatcher/segment_classification.py (SegmentClassifier :14-44,
setup_model :47-67): same constructor and call signature, same preprocessing chain, same
keep-if-argmax==1 rule and 1..k relabelling -- with the SqueezeNet-1.0 forward batched over all
segments of the call and run by PyTorch-ROCm (MIOpen picks the MFMA convolution kernels).

Differences from the reference, all deliberate:
  * the network is built in plain torch (torchvision is not needed) and is NOT fetched from the
    internet: setup_model's `pretrained=True` download is overwritten by model.pt anyway (:17, :51);
  * the model runs in eval() mode under no_grad: the reference never leaves train mode, so its
    Dropout(0.5) is live and its decisions are random; eval mode is the deterministic definition;
  * segments are classified in one batch instead of one 602 KB H2D copy + sync per segment.
"""
import numpy as np
import torch
from torch import nn

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # segment_classification.py:23
IMAGENET_STD = (0.229, 0.224, 0.225)
RESIZE = 24                                  # :20
PAD = (224 - 24) // 2                        # :21


class Fire(nn.Module):
    """torchvision.models.squeezenet.Fire: 1x1 squeeze, then 1x1 and 3x3 expands concatenated."""

    def __init__(self, inplanes, squeeze_planes, expand1x1_planes, expand3x3_planes):
        super().__init__()
        self.squeeze = nn.Conv2d(inplanes, squeeze_planes, kernel_size=1)
        self.squeeze_activation = nn.ReLU(inplace=True)
        self.expand1x1 = nn.Conv2d(squeeze_planes, expand1x1_planes, kernel_size=1)
        self.expand1x1_activation = nn.ReLU(inplace=True)
        self.expand3x3 = nn.Conv2d(squeeze_planes, expand3x3_planes, kernel_size=3, padding=1)
        self.expand3x3_activation = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.squeeze_activation(self.squeeze(x))
        return torch.cat([self.expand1x1_activation(self.expand1x1(x)),
                          self.expand3x3_activation(self.expand3x3(x))], 1)


class SqueezeNet10(nn.Module):
    """SqueezeNet 1.0 with the module names of torchvision's, so model.pt's state_dict
    (features.{0,3,4,5,7,8,9,10,12}.*, classifier.1.*) loads with strict=True."""

    def __init__(self, num_classes=2):
        super().__init__()
        self.num_classes = num_classes
        self.features = nn.Sequential(
            nn.Conv2d(3, 96, kernel_size=7, stride=2), nn.ReLU(inplace=True),
            nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True),
            Fire(96, 16, 64, 64), Fire(128, 16, 64, 64), Fire(128, 32, 128, 128),
            nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True),
            Fire(256, 32, 128, 128), Fire(256, 48, 192, 192), Fire(384, 48, 192, 192), Fire(384, 64, 256, 256),
            nn.MaxPool2d(kernel_size=3, stride=2, ceil_mode=True),
            Fire(512, 64, 256, 256))
        self.classifier = nn.Sequential(nn.Dropout(p=0.5), nn.Conv2d(512, num_classes, kernel_size=1),
                                        nn.ReLU(inplace=True), nn.AdaptiveAvgPool2d((1, 1)))

    def forward(self, x):
        return torch.flatten(self.classifier(self.features(x)), 1)


def setup_model(num_classes):
    """segment_classification.py:47-67 minus the ImageNet download: the 2-class head replaces
    classifier[1] there, and every weight is then overwritten by model.pt."""
    model = SqueezeNet10(num_classes)
    for p in model.parameters():
        p.requires_grad = False
    return model


def resize_segment(segment_image):
    """ToPILImage -> Resize((24, 24)) of the reference's transform list (:19-20): PIL's bilinear
    resampling with its support scaling, on the uint8 HxWx3 crop taken as RGB (:30-32 feed BGR crops
    as they are).  A 24x24 crop passes through unchanged, like PIL."""
    if segment_image.shape[0] == RESIZE and segment_image.shape[1] == RESIZE:
        return np.ascontiguousarray(segment_image)
    from PIL import Image
    bil = getattr(Image, "Resampling", Image).BILINEAR
    return np.asarray(Image.fromarray(np.ascontiguousarray(segment_image)).resize((RESIZE, RESIZE), bil))


class SegmentClassifier:
    """segment_classification.py:14-44."""

    def __init__(self, model_path, device=None, batch_size=1024):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("SegmentClassifier runs on the MI355X (PyTorch-ROCm); pass device='cpu' "
                                   "explicitly to run the torch CPU kernels instead")
            device = "cuda:0"
        self.device = torch.device(device)
        self.batch_size = batch_size
        self.model = setup_model(2)
        state = torch.load(model_path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(state, strict=True)
        self.model = self.model.to(self.device).eval()
        mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=self.device).view(1, 3, 1, 1)
        self._mean, self._std = mean, std
        # Pad(100) puts zeros around the 24x24 patch BEFORE ToTensor/Normalize: the border is (0-mean)/std
        self._border = ((0.0 - mean) / std).expand(1, 3, 224, 224).contiguous()

    def preprocess(self, segment_images):
        """(:18-24, :31-33) for a list of HxWx3 uint8 crops -> float32 (B, 3, 224, 224) on the device.
        On the GPU the whole chain (Pillow-exact resize, pad, /255, normalise) is one HIP kernel that writes the
        network's input tensor in place (swk_classifier_input); crops larger than 512 px or a CPU device use the
        torch / Pillow statements below."""
        if self.device.type == "cuda" and all(max(im.shape[0], im.shape[1]) <= 512 for im in segment_images):
            from . import _lib
            x = torch.empty((len(segment_images), 3, 224, 224), dtype=torch.float32, device=self.device)
            torch.cuda.synchronize(self.device)
            _lib.default_context(self.device.index or 0).classifier_input(segment_images, IMAGENET_MEAN, IMAGENET_STD,
                                                                          net_ptr=x.data_ptr())
            return x
        patches = np.stack([resize_segment(im) for im in segment_images])              # (B, 24, 24, 3) u8
        t = torch.from_numpy(patches).to(self.device).permute(0, 3, 1, 2).to(torch.float32).div_(255.0)   # ToTensor
        t = (t - self._mean) / self._std                                               # Normalize
        x = self._border.repeat(t.shape[0], 1, 1, 1)
        x[:, :, PAD:PAD + RESIZE, PAD:PAD + RESIZE] = t
        return x

    @torch.no_grad()
    def scores(self, segment_images):
        out = []
        for i in range(0, len(segment_images), self.batch_size):
            out.append(self.model(self.preprocess(segment_images[i:i + self.batch_size])))
        return torch.cat(out) if out else torch.zeros((0, 2), device=self.device)

    def __call__(self, segments):
        """:26-44: keep segments whose argmax is class 1 (ties / all-zero scores give 0 and are
        dropped, as torch.max does), relabel the kept ones 1..k."""
        if not segments:
            return []
        score = self.scores([s.segment_image for s in segments])
        pred = torch.max(score, 1)[1].cpu().numpy()
        kept = [s for s, y in zip(segments, pred) if y == 1]
        for i, s in enumerate(kept):
            s.label = i + 1
        return kept

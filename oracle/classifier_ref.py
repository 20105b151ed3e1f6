"""CPU oracle of the segment classifier -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates segment_classification.py:14-67 of the reference one segment at a time, batch 1, on the
CPU in float32, from the state_dict alone (functional convolutions; no nn.Module shared with the
product).  Deviations, all forced: torchvision is not installed, so SqueezeNet-1.0's topology is
written out (torchvision/models/squeezenet.py, version 1_0); `pretrained=True` would download
ImageNet weights that model.pt overwrites anyway; Dropout is the identity (eval mode) because
the reference's train-mode output is random.  PARITY: pinned only against model.pt's own key set
and shapes (tests/test_classifier.py, when /root/reference is present); no reference logits exist
because the reference module cannot be imported here (torchvision missing).
"""
import numpy as np
import torch
import torch.nn.functional as F

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
FIRES = {3: None, 4: None, 5: None, 7: None, 8: None, 9: None, 10: None, 12: None}


def transform(segment_image):
    """:18-24: ToPILImage, Resize((24,24)), Pad(100), ToTensor, Normalize -> (1, 3, 224, 224)."""
    from PIL import Image, ImageOps
    bil = getattr(Image, "Resampling", Image).BILINEAR
    img = Image.fromarray(np.ascontiguousarray(segment_image))           # ToPILImage (treated as RGB)
    img = img.resize((24, 24), bil)                                      # Resize
    img = ImageOps.expand(img, border=100, fill=0)                       # Pad
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)   # ToTensor
    t = (t - torch.from_numpy(MEAN).view(3, 1, 1)) / torch.from_numpy(STD).view(3, 1, 1)   # Normalize
    return t.unsqueeze(0)


def _fire(x, sd, i):
    p = "features.%d." % i
    x = F.relu(F.conv2d(x, sd[p + "squeeze.weight"], sd[p + "squeeze.bias"]))
    a = F.relu(F.conv2d(x, sd[p + "expand1x1.weight"], sd[p + "expand1x1.bias"]))
    b = F.relu(F.conv2d(x, sd[p + "expand3x3.weight"], sd[p + "expand3x3.bias"], padding=1))
    return torch.cat([a, b], 1)


def forward(sd, x):
    x = F.relu(F.conv2d(x, sd["features.0.weight"], sd["features.0.bias"], stride=2))
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    for i in (3, 4, 5):
        x = _fire(x, sd, i)
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    for i in (7, 8, 9, 10):
        x = _fire(x, sd, i)
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    x = _fire(x, sd, 12)
    x = F.relu(F.conv2d(x, sd["classifier.1.weight"], sd["classifier.1.bias"]))      # Dropout = identity
    return x.mean(dim=(2, 3))


def classify(sd, segment_images):
    """Per-segment scores (N, 2) float32 and the keep mask (argmax == 1; ties -> 0 -> dropped)."""
    sd = {k: v.float().cpu() for k, v in sd.items()}
    scores = []
    # a batch-1 forward of these small convolutions gets SLOWER with one thread per hardware thread (128 on the GPU box: tens of
    # milliseconds per segment against 6 with 8-16 threads, bench.py's cpu_baseline); the arithmetic does not depend on the count
    threads = torch.get_num_threads()
    torch.set_num_threads(min(threads, 16))
    try:
        with torch.no_grad():
            for im in segment_images:
                scores.append(forward(sd, transform(im))[0])
    finally:
        torch.set_num_threads(threads)
    s = torch.stack(scores) if scores else torch.zeros((0, 2))
    return s.numpy(), (torch.max(s, 1)[1] == 1).numpy() if len(scores) else np.zeros(0, bool)


EXPECTED_SHAPES = {
    "features.0.weight": (96, 3, 7, 7), "classifier.1.weight": (2, 512, 1, 1),
    "features.3.squeeze.weight": (16, 96, 1, 1), "features.12.expand3x3.weight": (256, 64, 3, 3),
}


def random_state_dict(seed):
    """Seeded random weights with SqueezeNet-1.0 (2-class head) shapes, scaled like Kaiming init and with a
    positive head bias so both classes occur."""
    g = torch.Generator().manual_seed(seed)
    plan = [("features.0", 96, 3, 7)]
    fires = {3: (96, 16, 64, 64), 4: (128, 16, 64, 64), 5: (128, 32, 128, 128), 7: (256, 32, 128, 128),
             8: (256, 48, 192, 192), 9: (384, 48, 192, 192), 10: (384, 64, 256, 256), 12: (512, 64, 256, 256)}
    for i, (inp, s, e1, e3) in fires.items():
        plan += [("features.%d.squeeze" % i, s, inp, 1), ("features.%d.expand1x1" % i, e1, s, 1),
                 ("features.%d.expand3x3" % i, e3, s, 3)]
    plan.append(("classifier.1", 2, 512, 1))
    sd = {}
    for name, co, ci, k in plan:
        fan = ci * k * k
        sd[name + ".weight"] = torch.randn((co, ci, k, k), generator=g) * (2.0 / fan) ** 0.5
        sd[name + ".bias"] = torch.randn((co,), generator=g) * 0.05
    return sd


def features(sd, x):
    x = F.relu(F.conv2d(x, sd["features.0.weight"], sd["features.0.bias"], stride=2))
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    for i in (3, 4, 5):
        x = _fire(x, sd, i)
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    for i in (7, 8, 9, 10):
        x = _fire(x, sd, i)
    x = F.max_pool2d(x, 3, 2, ceil_mode=True)
    return _fire(x, sd, 12)


def calibrate_head(sd, segment_images):
    """Random weights put every crop in the same class (89 % of the 224x224 input is constant padding).
    Shift the head biases so the median crop of `segment_images` sits on the decision boundary and both
    pre-activations stay positive (ReLU inactive): about half the crops are then kept."""
    with torch.no_grad():
        pooled = torch.stack([features(sd, transform(im))[0].mean(dim=(1, 2)) for im in segment_images])
        w = sd["classifier.1.weight"].view(2, 512)
        pre = pooled @ w.t()                                    # (N, 2) spatial means of the 1x1 conv
        d = pre[:, 1] - pre[:, 0]
        big = float(pre.abs().max()) + 5.0
        sd["classifier.1.bias"] = torch.tensor([big, big - float(d.median())])
    return sd

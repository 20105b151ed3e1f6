"""Drop-in for the reference's swiftwatcher/segment_classification.py (SegmentClassifier :14-44,
setup_model :47-67): same constructor and call signature, same preprocessing chain, same
keep-if-argmax==1 rule and 1..k relabelling -- with the SqueezeNet-1.0 forward batched over all
segments of the call and run by PyTorch-ROCm (MIOpen picks the MFMA convolution kernels).

Differences from the reference, all deliberate:
  * the network is built in plain torch (torchvision is not needed) and is NOT fetched from the
    internet: setup_model's `pretrained=True` download is overwritten by model.pt anyway (:17, :51);
  * the model runs in eval() mode under no_grad: the reference never leaves train mode, so its
    Dropout(0.5) is live and its decisions are random; eval mode is the deterministic definition;
  * segments are classified in one batch instead of one 602 KB H2D copy + sync per segment;
  * by default only the part of each feature map the 24x24 patch can influence is evaluated
    (CroppedSqueezeNet10, 5.7x fewer MACs: 0.128 instead of 0.733 G per segment, same arithmetic per output); cropped=False runs the full network.
"""
import ctypes
import os
import threading

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


def _affected(lo, hi, k, st, pad, out_size):
    """Outputs of a (k, stride, pad) window layer that see any input index in [lo, hi]."""
    a = -((-(lo + pad - k + 1)) // st)          # ceil
    b = (hi + pad) // st
    return max(a, 0), min(b, out_size - 1)


class CroppedSqueezeNet10:
    """Receptive-field cropped evaluation of SqueezeNet10 for this classifier's inputs.

    Every input is a 24x24 patch in the middle of a 224x224 image whose other 89 % is ONE constant (Pad(100) then
    Normalize, segment_classification.py:21-23).  Outside the patch's receptive field every activation is therefore
    the same for every segment: it is computed once, from a blank image, when the model is loaded.  Per batch only
    the affected square of each feature map is evaluated -- on a tile cut from that background map (which supplies
    the halo a 3x3 window needs) with the live values pasted in its middle and the layer run without padding:

        input 40x40 (rows 92..131) -> conv1 17x17 (46..62 of 109) -> pool 8x8 (23..30 of 54)
        -> fire2..4: 10, 12, 14 -> pool 8x8 (9..16 of 27) -> fire5..8: 10, 12, 14, 16 -> pool 9x9 (2..10 of 13)
        -> fire9 11x11 -> 1x1 head + ReLU on 11x11, plus the head's constant ring, / 169.

    Same arithmetic per output as the full network (0.128 instead of 0.733 GMAC per segment); results differ from
    it only by float32 summation order inside the convolution kernels and in the final average."""

    IN_LO, IN_HI = 92, 131            # rows/cols of the 224-pixel input the tile covers (patch at 100..123)

    def __init__(self, model, border):
        """model: SqueezeNet10 in eval mode on its device; border: (3,) tensor, the normalised pad value."""
        self.model = model
        dev = next(model.parameters()).device
        feats = list(model.features)
        with torch.no_grad():
            x = border.to(dev, torch.float32).view(1, 3, 1, 1).expand(1, 3, 224, 224).contiguous()
            maps = []                  # maps[i] = input of features[i] for the blank image
            for layer in feats:
                maps.append(x)
                x = layer(x)
            head = torch.relu(model.classifier[1](x))                 # (1, 2, 13, 13)
        lo, hi = PAD, PAD + RESIZE - 1
        size = 224
        self.plan = []                 # per features[i] from the first Fire on: (kind, tile, paste offset, paste size)
        # conv1 + ReLU + pool are run on the input tile directly
        assert isinstance(feats[0], nn.Conv2d) and feats[0].padding == (0, 0)
        out_size = maps[1].shape[-1]
        lo, hi = _affected(lo, hi, 7, 2, 0, out_size)
        need = (2 * lo, 2 * hi + 6)
        assert need[0] >= self.IN_LO and need[1] <= self.IN_HI
        assert (2 * lo - self.IN_LO) % 2 == 0
        self.conv_skip = (2 * lo - self.IN_LO) // 2          # conv outputs the tile yields before the first affected one
        conv_lo = lo - self.conv_skip
        conv_hi = conv_lo + (self.IN_HI - self.IN_LO + 1 - 7) // 2
        size = out_size
        # the first pool works on conv rows [conv_lo, conv_hi] (all computed from the tile, background included)
        out_size = maps[3].shape[-1]
        plo, phi = _affected(lo, hi, 3, 2, 0, out_size)
        assert 2 * plo >= conv_lo and 2 * phi + 2 <= conv_hi
        self.pool1_slice = (2 * plo - conv_lo, 2 * phi + 2 - conv_lo + 1)
        lo, hi, size = plo, phi, out_size
        for i in range(3, len(feats)):
            layer = feats[i]
            in_map = maps[i]
            if isinstance(layer, Fire):
                a, b = max(lo - 1, 0), min(hi + 1, size - 1)
                n0, n1 = a - 1, b + 1                       # inputs the 3x3 expand needs; may leave the map by one
                c0, c1 = max(n0, 0), min(n1, size - 1)
                tile = in_map[:, :, c0:c1 + 1, c0:c1 + 1].contiguous()
                self.plan.append(("fire", layer, tile, lo - c0, hi - lo + 1, (c0 - n0, n1 - c1), (a - c0, b - a + 1)))
                lo, hi = a, b
            else:                                            # MaxPool2d(3, 2, ceil_mode=True)
                out_size = maps[i + 1].shape[-1] if i + 1 < len(feats) else x.shape[-1]
                a, b = _affected(lo, hi, 3, 2, 0, out_size)
                n0, n1 = 2 * a, 2 * b + 2
                assert n0 >= 0 and n1 <= size - 1            # no clipped window among the affected ones
                tile = in_map[:, :, n0:n1 + 1, n0:n1 + 1].contiguous()
                self.plan.append(("pool", layer, tile, lo - n0, hi - lo + 1, None, None))
                lo, hi, size = a, b, out_size
        self.final = (lo, hi, size)
        self.final_bg = x[:, :, lo:hi + 1, lo:hi + 1].contiguous()          # the last Fire's output for the blank image, live square
        ring = head.clone()
        ring[:, :, lo:hi + 1, lo:hi + 1] = 0
        self.ring_sum = ring.sum(dim=(2, 3))                 # (1, 2): the head's input-independent positions
        self.n_pos = float(size * size)
        # input-independent squeeze output of every Fire tile (with the map's own zero padding where a tile reaches
        # the edge): the ring of the persistent squeeze buffers
        self.sq_bg = []
        with torch.no_grad():
            for kind, layer, tile, off, n, pad, crop in self.plan:
                if kind != "fire":
                    self.sq_bg.append(None)
                    continue
                sq = layer.squeeze_activation(layer.squeeze(tile))
                if pad[0] or pad[1]:
                    sq = torch.nn.functional.pad(sq, (pad[0], pad[1], pad[0], pad[1]))
                self.sq_bg.append(sq.contiguous())
        self._cap = 0
        self._buf = None
        # channels-last on the GPU: what the library's convolution kernels read and write
        self.memory_format = torch.channels_last if dev.type == "cuda" else torch.contiguous_format
        if self.memory_format == torch.channels_last:
            model.to(memory_format=torch.channels_last)
        # The two cross-checks the tests keep: own_kernels = False routes every convolution through MIOpen (+ the placement kernel),
        # winograd = False runs the 3x3 expands on the direct kernel (csrc/cnn_conv3x3.hip) instead of F(2x2, 3x3).
        self.own_kernels = os.environ.get("SWK_OWN_CNN_KERNELS", "1") == "1"
        self._head = None
        self.winograd = True
        # max-pool + the squeeze behind it as one kernel (csrc/cnn_poolsq.hip): the pooled tensor never goes to memory
        self.fuse_pool = os.environ.get("SWK_FUSE_POOL", "1") == "1"
        self._w1 = None
        self._wt3 = {}
        self._ww3 = {}

    def macs_per_segment(self):
        """Multiply-accumulates of one forward per segment: (executed, useful).  "useful" prices every convolution output the next
        layer reads as a direct convolution (the figure to compare networks and hardware with); "executed" is what the kernels
        really multiply: the Winograd F(2x2, 3x3) kernel does 16 instead of 36 per 2 x 2 outputs (tiles padded to even sizes), the
        MIOpen path (fused kernels off) runs the 1x1 expands over the whole squeeze tile.  The full 224 x 224 network does 0.7326 G."""
        m = self.model
        c1 = m.features[0]
        side = (self.IN_HI - self.IN_LO + 1 - 7) // 2 + 1
        executed = useful = side * side * c1.out_channels * c1.in_channels * 49
        last_n = None
        on_gpu = self.ring_sum.is_cuda and self.memory_format == torch.channels_last and self.own_kernels
        for kind, layer, tile, off, n, pad, crop in self.plan:
            if kind != "fire":
                continue
            t = tile.shape[2] + pad[0] + pad[1]
            sq, e1, e3 = layer.squeeze, layer.expand1x1, layer.expand3x3
            executed += n * n * sq.out_channels * sq.in_channels
            e1_side = n if on_gpu else t      # the fused kernel only computes the outputs that depend on the segment
            executed += e1_side * e1_side * e1.out_channels * e1.in_channels
            wino = on_gpu and self.winograd and e3.out_channels == 4 * e3.in_channels and e3.in_channels in (16, 32, 48, 64)
            if wino:
                tiles = ((t - 2 + 1) // 2) ** 2
                executed += tiles * 16 * e3.out_channels * e3.in_channels
            else:
                executed += (t - 2) ** 2 * e3.out_channels * e3.in_channels * 9
            useful += n * n * sq.out_channels * sq.in_channels
            useful += crop[1] ** 2 * (e1.out_channels * e1.in_channels + e3.out_channels * e3.in_channels * 9)
            last_n = crop[1]
        head = m.classifier[1]
        executed += last_n * last_n * head.out_channels * head.in_channels
        useful += last_n * last_n * head.out_channels * head.in_channels
        return int(executed), int(useful)

    def _buffers(self, batch):
        """Persistent per-layer tiles for up to `batch` segments.  Their rings hold the background values and are
        written once; a forward pass only overwrites the live centres."""
        if batch <= self._cap:
            return self._buf
        bufs = []
        for (kind, layer, tile, off, n, pad, crop), sq in zip(self.plan, self.sq_bg):
            if kind == "fire":
                bufs.append(sq.expand(batch, -1, -1, -1).contiguous(memory_format=self.memory_format))
            else:
                bufs.append(tile.expand(batch, -1, -1, -1).contiguous(memory_format=self.memory_format))
        # live (ring-free) inputs of the Fires that follow a Fire, and of the head
        live = []
        for j, (kind, layer, tile, off, n, pad, crop) in enumerate(self.plan):
            nxt = self.plan[j + 1][0] if j + 1 < len(self.plan) else "head"
            if kind == "fire" and nxt != "pool":
                # initialised with the blank image's values: the expand1x1 outputs on the ring of the square (squeeze values that do
                # not depend on the segment) are never recomputed, a forward writes the expand1x1 centre and all of expand3x3
                if nxt == "fire":
                    _, _, ntile, noff, nn, _, _ = self.plan[j + 1]
                    bg = ntile[:, :, noff:noff + nn, noff:noff + nn]
                else:
                    bg = self.final_bg
                assert bg.shape[2] == crop[1] and bg.shape[1] == layer.expand1x1.out_channels + layer.expand3x3.out_channels
                live.append(bg.expand(batch, -1, -1, -1).contiguous(memory_format=self.memory_format))
            else:
                live.append(None)
        self._buf, self._cap = (bufs, live), batch
        self.version = getattr(self, "version", 0) + 1          # new addresses: captured graphs of the old ones are void
        return self._buf

    def _gpu_path(self, tiles):
        return tiles.is_cuda and self.memory_format == torch.channels_last and all(
            kind != "fire" or not (pad[0] or pad[1]) for kind, _, _, _, _, pad, _ in self.plan)

    def reserve(self, batch):
        """Persistent tiles for `batch` segments, made now (on the current stream)."""
        self._buffers(batch)
        if self.ring_sum.is_cuda:
            self._aux_buffers(batch)

    @torch.no_grad()
    def __call__(self, tiles, row0=0):
        """tiles: (B, 3, 40, 40) float32 = rows/cols 92..131 of the normalised 224x224 input.  row0: first row of the persistent
        per-layer tiles this forward works in (two forwards on disjoint row ranges may run side by side on two streams)."""
        if self._gpu_path(tiles):
            with torch.cuda.device(tiles.device):            # the glue kernels go to this device's current stream
                return self._forward_hip_glue(tiles, row0)
        if row0:
            raise ValueError("row0 is a feature of the GPU path")
        m = self.model
        conv2d = torch.nn.functional.conv2d
        k = tiles.shape[0]
        bufs, live = self._buffers(k)
        x = torch.relu(m.features[0](tiles.contiguous(memory_format=self.memory_format)))
        a, b = self.pool1_slice
        x = m.features[2](x[:, :, a:b, a:b])
        for j, (kind, layer, tile, off, n, pad, crop) in enumerate(self.plan):
            if kind == "pool":
                x = layer(bufs[j][:k])                       # centre written by the Fire before it
                continue
            sq = bufs[j][:k]
            o = off + pad[0]
            torch.clamp_min(layer.squeeze(x), 0, out=sq[:, :, o:o + n, o:o + n])
            e3 = conv2d(sq, layer.expand3x3.weight, layer.expand3x3.bias)            # valid: the halo is in the tile
            e1 = conv2d(sq, layer.expand1x1.weight, layer.expand1x1.bias)
            c, cn = crop
            c1 = layer.expand1x1.out_channels
            if live[j] is not None:
                dest = live[j][:k]
            else:                                            # straight into the centre of the pool tile
                _, _, _, poff, pn, _, _ = self.plan[j + 1]
                assert pn == cn
                dest = bufs[j + 1][:k, :, poff:poff + pn, poff:poff + pn]
            cp = c + pad[0]
            torch.clamp_min(e1[:, :, cp:cp + cn, cp:cp + cn], 0, out=dest[:, :c1])
            torch.clamp_min(e3, 0, out=dest[:, c1:])
            x = dest
        s = torch.relu(m.classifier[1](x)).sum(dim=(2, 3))
        return (s + self.ring_sum) / self.n_pos

    def _forward_hip_glue(self, tiles, row0=0):
        """The same forward on the GPU with the convolutions alone left to MIOpen: bias + ReLU + placement into the
        next tile is one HIP kernel per convolution output (swk_nhwc_bias_relu_place), max-pooling another
        (swk_nhwc_maxpool3s2), both on PyTorch's current stream.  Same float32 operations per element as above."""
        from . import _lib
        lib = _lib.load()
        stream = ctypes.c_void_p(torch.cuda.current_stream(tiles.device).cuda_stream)
        m = self.model
        conv2d = torch.nn.functional.conv2d
        k = tiles.shape[0]
        bufs, live = self._buffers(row0 + k)
        rows = slice(row0, row0 + k)
        cl = torch.channels_last

        def nhwc(t):
            return t if t.is_contiguous(memory_format=cl) else t.contiguous(memory_format=cl)

        def place(e, bias, dest, crop, size, off, c_off):
            e = nhwc(e)
            rc = lib.swk_nhwc_bias_relu_place(stream, e.data_ptr(), k, e.shape[2], e.shape[3], e.shape[1], crop, crop, size, size,
                                              bias.data_ptr(), dest.data_ptr(), dest.shape[2], dest.shape[3], dest.shape[1],
                                              off, off, c_off)
            if rc:
                raise RuntimeError("swk_nhwc_bias_relu_place failed (%d)" % rc)

        def conv1x1(src, crop, size, conv, dest, off, c_off):
            src = nhwc(src)
            wgt = conv.weight.reshape(conv.out_channels, conv.in_channels)       # 1 x 1 kernel: same memory in both layouts
            rc = lib.swk_nhwc_conv1x1_bias_relu_place(stream, src.data_ptr(), k, src.shape[2], src.shape[3], src.shape[1], crop, crop,
                                                      size, size, wgt.data_ptr(), conv.bias.data_ptr(), conv.out_channels,
                                                      dest.data_ptr(), dest.shape[2], dest.shape[3], dest.shape[1], off, off, c_off)
            if rc:
                raise RuntimeError("swk_nhwc_conv1x1_bias_relu_place failed (%d)" % rc)

        def conv3x3(src, j, conv, dest, off, c_off):
            src = nhwc(src)
            cin, cout = conv.in_channels, conv.out_channels
            if self.winograd and cout == 4 * cin and cin in (16, 32, 48, 64) and k * src.shape[2] * src.shape[2] * cin * 4 < (1 << 32):
                ww = self._ww3.get(j)
                if ww is None:       # G g G^T in the kernel's operand layout, once per layer (host code of the library)
                    w = conv.weight.detach().to("cpu", torch.float32).contiguous()
                    out = torch.empty(16 * cin * cout, dtype=torch.float32)
                    if lib.swk_winograd_f2x2_3x3_weights(w.data_ptr(), cout, cin, out.data_ptr()):
                        raise RuntimeError("swk_winograd_f2x2_3x3_weights failed")
                    ww = self._ww3[j] = out.to(src.device)
                rc = lib.swk_nhwc_conv3x3_winograd_bias_relu_place(stream, src.data_ptr(), k, src.shape[2], cin, ww.data_ptr(),
                                                                   conv.bias.data_ptr(), cout, dest.data_ptr(), dest.shape[2],
                                                                   dest.shape[3], dest.shape[1], off, off, c_off)
                if rc:
                    raise RuntimeError("swk_nhwc_conv3x3_winograd_bias_relu_place failed (%d)" % rc)
                return
            wt = self._wt3.get(j)
            if wt is None:       # [co][ci][3][3] -> [tap][ci][co], once per layer
                wt = self._wt3[j] = conv.weight.detach().permute(2, 3, 1, 0).contiguous(memory_format=torch.contiguous_format)
            assert src.shape[2] == src.shape[3]
            rc = lib.swk_nhwc_conv3x3_bias_relu_place(stream, src.data_ptr(), k, src.shape[2], src.shape[1], wt.data_ptr(),
                                                      conv.bias.data_ptr(), conv.out_channels, dest.data_ptr(), dest.shape[2],
                                                      dest.shape[3], dest.shape[1], off, off, c_off)
            if rc:
                raise RuntimeError("swk_nhwc_conv3x3_bias_relu_place failed (%d)" % rc)

        def pool(src, dst):
            rc = lib.swk_nhwc_maxpool3s2(stream, src.data_ptr(), k, src.shape[2], src.shape[3], src.shape[1], dst.data_ptr())
            if rc:
                raise RuntimeError("swk_nhwc_maxpool3s2 failed (%d)" % rc)

        def pool_squeeze(src, conv, dest, off, ring=None, live=(0, 0)):
            # ring: the tile whose ring every tile of src shares (written once, _buffers): only the live square is read per segment
            wgt = conv.weight.reshape(conv.out_channels, conv.in_channels)
            rc = lib.swk_nhwc_maxpool3s2_conv1x1_bias_relu_place(stream, src.data_ptr(), k, src.shape[2], src.shape[1], wgt.data_ptr(),
                                                                 conv.bias.data_ptr(), conv.out_channels, dest.data_ptr(), dest.shape[2],
                                                                 dest.shape[3], dest.shape[1], off, off,
                                                                 None if ring is None else ring.data_ptr(), live[0], live[1])
            if rc:
                raise RuntimeError("swk_nhwc_maxpool3s2_conv1x1_bias_relu_place failed (%d)" % rc)

        aux = self._aux_buffers(row0 + k)
        conv1 = m.features[0]
        a, b = self.pool1_slice
        c1buf = aux["conv1"][rows]
        side = tiles.shape[2]
        if self.own_kernels and conv1.out_channels == 96 and side % 2 == 0 and 2 * (b - 1) + 8 <= side:
            # conv1 + bias + ReLU on the rows the first pool reads, one kernel (csrc/cnn_conv1.hip)
            if self._w1 is None:
                self._w1 = conv1.weight.detach().contiguous(memory_format=torch.contiguous_format).clone()
            xin = nhwc(tiles)
            rc = lib.swk_nhwc_conv7x7s2_bias_relu(stream, xin.data_ptr(), k, side, a, b - a, self._w1.data_ptr(), conv1.bias.data_ptr(),
                                                  conv1.out_channels, c1buf.data_ptr())
            if rc:
                raise RuntimeError("swk_nhwc_conv7x7s2_bias_relu failed (%d)" % rc)
        else:
            e = conv2d(nhwc(tiles), conv1.weight, None, stride=conv1.stride)
            place(e, conv1.bias, c1buf, a, b - a, 0, 0)
        fuse = self.fuse_pool and self.own_kernels
        to_pool = c1buf               # a tensor whose max-pool the next squeeze reads (fused into it), or None
        pool_ring, pool_live = None, (0, 0)        # conv1's output has no ring; the pool tiles behind fire4 and fire8 do
        x = aux["pool_in"][rows]
        if not fuse:
            pool(c1buf, x)
        pi = 0
        for j, (kind, layer, tile, off, n, pad, crop) in enumerate(self.plan):
            if kind == "pool":
                x = aux["pool_out"][pi][rows]
                if fuse:
                    to_pool, pool_ring, pool_live = bufs[j][rows], bufs[j][0:1], (off, n)
                else:
                    pool(bufs[j][rows], x)
                pi += 1
                continue
            sq = bufs[j][rows]
            c, cn = crop
            c1 = layer.expand1x1.out_channels
            if live[j] is not None:
                dest, doff = live[j][rows], 0
            else:
                dest, doff = bufs[j + 1][rows], self.plan[j + 1][3]
            if self.own_kernels:
                # squeeze and expand1x1 as ONE kernel each on the f32 matrix cores: convolution + bias + ReLU + placement
                # (csrc/cnn_conv1x1.hip)
                if fuse and to_pool is not None:
                    assert (to_pool.shape[2] - 3) // 2 + 1 == n
                    pool_squeeze(to_pool, layer.squeeze, sq, off, pool_ring, pool_live)
                    to_pool = None
                else:
                    conv1x1(x, 0, n, layer.squeeze, sq, off, 0)
                # expand1x1 only where the squeeze output depends on the segment (n x n inside the cn x cn square: the ring keeps the
                # blank image's values the buffers were created with)
                conv1x1(sq, off, n, layer.expand1x1, dest, doff + off - c, 0)
            else:
                place(conv2d(x, layer.squeeze.weight, None), layer.squeeze.bias, sq, 0, n, off, 0)
                e1 = conv2d(sq, layer.expand1x1.weight, None)
                place(e1, layer.expand1x1.bias, dest, c, cn, doff, 0)
            if self.own_kernels:
                # the 3x3 expand (valid convolution over the squeeze tile) likewise (csrc/cnn_conv3x3.hip)
                conv3x3(sq, j, layer.expand3x3, dest, doff, c1)
            else:
                place(conv2d(sq, layer.expand3x3.weight, None), layer.expand3x3.bias, dest, 0, cn, doff, c1)
            x = dest
        # the 512 -> 2 head (1 x 1 convolution + ReLU + spatial sum) as a plain matrix product over the channels-last pixels: no
        # convolution library on this path, hence no kernel search per batch shape
        head = m.classifier[1]
        if self.own_kernels and head.out_channels == 2 and x.shape[1] in (256, 512, 768, 1024) and x.is_contiguous(memory_format=cl):
            # the head as one kernel with a fixed summation order (csrc/cnn_aux.hip): scores that do not depend on the batch's row count
            if self._head is None:
                self._head = (head.weight.detach().reshape(2, -1).contiguous(), head.bias.detach().contiguous(),
                              self.ring_sum.reshape(-1).contiguous())
            hw, hb, ring = self._head
            out = torch.empty((k, 2), dtype=torch.float32, device=x.device)
            rc = lib.swk_nhwc_head2_relu_mean(stream, x.data_ptr(), k, x.shape[2] * x.shape[3], x.shape[1], hw.data_ptr(), hb.data_ptr(),
                                              ring.data_ptr(), self.n_pos, out.data_ptr())
            if rc:
                raise RuntimeError("swk_nhwc_head2_relu_mean failed (%d)" % rc)
            return out
        px = x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])                       # (B * live positions, 512): a view of the NHWC memory
        s = torch.relu(torch.nn.functional.linear(px, head.weight.view(head.out_channels, -1), head.bias))
        s = s.view(k, -1, head.out_channels).sum(dim=1)
        return (s + self.ring_sum) / self.n_pos

    def _aux_buffers(self, batch):
        if getattr(self, "_aux_cap", 0) >= batch:
            return self._aux
        dev = self.ring_sum.device
        cl = torch.channels_last

        def buf(c, h):
            return torch.empty((batch, c, h, h), dtype=torch.float32, device=dev).contiguous(memory_format=cl)

        a, b = self.pool1_slice
        conv1_c = self.model.features[0].out_channels
        aux = {"conv1": buf(conv1_c, b - a), "pool_in": buf(conv1_c, (b - a - 3) // 2 + 1), "pool_out": []}
        for kind, layer, tile, off, n, pad, crop in self.plan:
            if kind == "pool":
                aux["pool_out"].append(buf(tile.shape[1], (tile.shape[2] - 3) // 2 + 1))
        self._aux, self._aux_cap = aux, batch
        self.version = getattr(self, "version", 0) + 1
        return aux


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

    @classmethod
    def from_state_dict(cls, state, **kw):
        """Same as the constructor, from an in-memory state_dict (synthetic-weight benchmarks, tests)."""
        import io
        buf = io.BytesIO()
        torch.save(state, buf)
        buf.seek(0)
        return cls(buf, **kw)

    def __init__(self, model_path, device=None, batch_size=1024, cropped=True):
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("SegmentClassifier runs on the MI355X (PyTorch-ROCm); pass device='cpu' "
                                   "explicitly to run the torch CPU kernels instead")
            device = "cuda:0"
        self.device = torch.device(device)
        # MIOpen's exhaustive find instead of its immediate-mode heuristic for the convolutions left to it (the head; everything with
        # the fused kernels off): scoped to this classifier's forwards (torch.backends.cudnn.flags in _run), not set process-wide
        self._cudnn_benchmark = self.device.type == "cuda" and os.environ.get("SWK_CUDNN_BENCHMARK", "1") == "1"
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
        self.cropped = CroppedSqueezeNet10(self.model, ((0.0 - mean) / std).view(3)) if cropped else None
        # measurement hook (bench.py): with timing on, every network forward is bracketed by events on torch's
        # current stream; net_time() sums them
        self.timing = False
        self._events = []
        # device-side input buffers of scores_from_device / predict_last_batch: two slots, each with the event of the last forward
        # that read it (a slot is handed to the library's stream again only after that event)
        self._slots = None
        self._lock = threading.RLock()          # one scoring at a time (see _scores_device)
        # forwards of this many rows or more run as two chains on two streams (_forward_two_streams); 0 = never
        self._split_rows = int(os.environ.get("SWK_CNN_SPLIT_ROWS", "1024")) if self.device.type == "cuda" else 0
        self._side_stream = None
        self._graphs = {}
        self._use_graphs = self.device.type == "cuda" and os.environ.get("SWK_HIP_GRAPHS", "1") == "1"
        self._graph_error = None

    def preprocess(self, segment_images, window=False):
        """(:18-24, :31-33) for a list of HxWx3 uint8 crops -> float32 (B, 3, 224, 224) on the device, or with
        window=True only rows/cols 92..131 of it, (B, 3, 40, 40), which is all the cropped network reads.
        On the GPU the whole chain (Pillow-exact resize, pad, /255, normalise) is one HIP kernel that writes the
        network's input tensor in place (swk_classifier_input_window); crops larger than 512 px or a CPU device
        use the torch / Pillow statements below."""
        lo, hi = (CroppedSqueezeNet10.IN_LO, CroppedSqueezeNet10.IN_HI + 1) if window else (0, 224)
        if self.device.type == "cuda" and all(max(im.shape[0], im.shape[1]) <= 512 for im in segment_images):
            from . import _lib
            x = torch.empty((len(segment_images), 3, hi - lo, hi - lo), dtype=torch.float32, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()     # the library fills x on its own stream
            _lib.default_context(self.device.index or 0).classifier_input(segment_images, IMAGENET_MEAN, IMAGENET_STD,
                                                                          net_ptr=x.data_ptr(), pad=PAD - lo)
            # (x is a fresh allocation: the caching allocator may hand out a block whose last reader is still queued on
            # torch's stream -- hence the synchronize above, before the library's stream writes it)
            return x
        patches = np.stack([resize_segment(im) for im in segment_images])              # (B, 24, 24, 3) u8
        t = torch.from_numpy(patches).to(self.device).permute(0, 3, 1, 2).to(torch.float32).div_(255.0)   # ToTensor
        t = (t - self._mean) / self._std                                               # Normalize
        x = self._border[:, :, lo:hi, lo:hi].repeat(t.shape[0], 1, 1, 1)
        x[:, :, PAD - lo:PAD - lo + RESIZE, PAD - lo:PAD - lo + RESIZE] = t
        return x

    @torch.no_grad()
    def _bucket(self, k):
        """Batch size the network is run at for k inputs: a counting loop hands over a different number of segments every call, and
        a few fixed shapes keep the per-shape state small (persistent tiles are sized by the largest; with the fused kernels off
        MIOpen searches once per shape) -- so on the GPU the batch is padded to a multiple of 64 up to 512 rows, beyond that to a
        multiple of 512 up to batch_size.  The padding rows are scored and thrown away."""
        if self.device.type != "cuda":
            return k
        if k >= self.batch_size:
            return max(self.batch_size, k)
        if k <= 512:                         # a FrameQueue window's worth of segments: 64-row steps
            return min(self.batch_size, -(-k // 64) * 64)
        return min(self.batch_size, -(-k // 512) * 512)

    def _forward(self, x):
        # (nothing of the default GPU path is a MIOpen convolution: the setting only matters for the full network and the cross-check)
        if self._cudnn_benchmark and (self.cropped is None or not self.cropped.own_kernels):
            with torch.backends.cudnn.flags(enabled=True, benchmark=True):
                return self.cropped(x) if self.cropped is not None else self.model(x)
        if (self._split_rows and x.shape[0] >= self._split_rows and self.cropped is not None and self.cropped.own_kernels
                and self.cropped._gpu_path(x)):
            return self._forward_two_streams(x)
        return self.cropped(x) if self.cropped is not None else self.model(x)

    def _forward_two_streams(self, x):
        """A large forward as two forwards on disjoint rows of the persistent tiles, the second on a side stream.  The kernels of a
        forward alternate between the matrix pipe (3 x 3 expands, 70 % busy at 1 TB/s) and memory (squeezes, expand1x1: half the HBM
        rate), and every one of its 34 launches ends with a tail of part-filled CUs; two chains fill each other's gaps: 13.86 against
        14.66 ms per 8,192 rows (tools/r4/two_stream_forward.py; three chains 13.97, four 14.48; tools/r4/chunked_forward.py: 2,048 rows
        as 2 x 1,024 take 9 % less than as one chain, 1,024 as 2 x 512 8 % less; window-sized forwards gain nothing, see _forward_graphed).  Same kernels on the same rows, the
        head included (swk_nhwc_head2_relu_mean sums in an order fixed by the shapes): the scores are the single chain's bit for bit."""
        k = x.shape[0]
        half = -(-k // 2 // 512) * 512
        cur = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        self.cropped.reserve(k)                     # the tiles grow on THIS stream, before either half looks them up
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            hi = self.cropped(x[half:], row0=half)
        lo = self.cropped(x[:half])
        cur.wait_stream(side)
        hi.record_stream(cur)
        return torch.cat([lo, hi])

    def _forward_graphed(self, x):
        """A window's worth of segments (<= 512 rows) is some thirty kernels of a few microseconds each: launched one by one from
        Python they cost more host time than GPU time.  The forward on a persistent input slot is captured once per (slot, rows) as a
        HIP graph and replayed with one launch.  Anything that goes wrong while capturing switches this off for the classifier."""
        key = (x.data_ptr(), int(x.shape[0]), getattr(self.cropped, "version", 0))
        entry = self._graphs.get(key)
        if entry is None:
            try:
                self._forward(x)                                            # kernel attributes, persistent tiles, library handles
                torch.cuda.current_stream(self.device).synchronize()
                graph = torch.cuda.CUDAGraph()
                # thread-local capture mode: another thread may be inside the library meanwhile (a reader that segments ahead:
                # stream synchronisations, allocations) -- under the default, global mode those calls fail while this one captures
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    # (two halves on two streams inside the graph -- forwards on disjoint rows of the persistent tiles, row0 --
                    # were measured: 3.11 against 3.17 ms per window of the counting loop, no gain; one chain it is)
                    out = self._forward(x)
                if len(self._graphs) >= 24:
                    self._graphs.clear()
                entry = self._graphs[key] = (graph, out)
            except Exception as exc:                                        # noqa: BLE001
                self._use_graphs = False
                self._graph_error = repr(exc)
                torch.cuda.synchronize(self.device)
                return self._forward(x)
        entry[0].replay()
        return entry[1]

    def _run(self, x, k, persistent=False):
        """Scores of the first k rows of x (x has _bucket(k) rows).  persistent: x is one of the classifier's own input slots."""
        if persistent and self._use_graphs and x.shape[0] <= 512 and self.cropped is not None and not self.timing:
            return self._forward_graphed(x)[:k]
        if self.timing and self.device.type == "cuda":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self._forward(x)[:k]
            b.record()
            self._events.append((a, b, int(x.shape[0])))
            return out
        return self._forward(x)[:k]

    def net_time(self, reset=True):
        """(milliseconds, rows pushed through the network, forwards) of the timed forwards since the last reset."""
        if self._events:
            torch.cuda.synchronize(self.device)
        ms = sum(a.elapsed_time(b) for a, b, _ in self._events)
        rows, calls = sum(r for _, _, r in self._events), len(self._events)
        if reset:
            self._events = []
        return ms, rows, calls

    def _scores_unlocked(self, segment_images):
        out = []
        for i in range(0, len(segment_images), self.batch_size):
            chunk = segment_images[i:i + self.batch_size]
            k, b = len(chunk), self._bucket(len(chunk))
            x = self.preprocess(chunk, window=self.cropped is not None)
            if b != k:
                x = torch.cat([x, x[:1].expand(b - k, -1, -1, -1)])
            out.append(self._run(x, k))
        return torch.cat(out) if out else torch.zeros((0, 2), device=self.device)

    def _device_slots(self, side, nhwc):
        key = (side, nhwc, self.batch_size)
        if self._slots is None or self._slots[0] != key:
            fmt = torch.channels_last if nhwc else torch.contiguous_format
            xb = [torch.empty((self.batch_size, 3, side, side), dtype=torch.float32, device=self.device, memory_format=fmt) for _ in range(2)]
            fb = [torch.empty((self.batch_size,), dtype=torch.int32, device=self.device) for _ in range(2)]
            torch.cuda.current_stream(self.device).synchronize()          # fresh blocks: nothing queued on torch's stream may still read them
            self._slots = (key, xb, fb, [None, None])
        return self._slots[1:]

    def scores(self, segment_images):
        with self._lock:
            return self._scores_unlocked(segment_images)

    def _scores_device(self, cut):
        """_scores_device_unlocked under the classifier's lock: the input slots, their events, the captured graphs and the forward's
        buffers belong to the classifier, and two threads score through one classifier when a reader segments (and scores) ahead
        beside the counting loop (io_frames.PresegmentingReader, pipeline.py windows_per_call): one scoring at a time."""
        with self._lock:
            return self._scores_device_unlocked(cut)

    @torch.no_grad()
    def _scores_device_unlocked(self, cut):
        """Scores of a device-resident batch.  cut(net_ptr, frame_ptr, net_cap, first, pad, channels_last) -> (total, skipped) writes
        the network inputs of segments [first, first + net_cap) (a library call: its own stream, synchronous).  Two input slots: the
        library cuts and resamples chunk i + 1 while PyTorch's stream runs the network on chunk i; a slot is written again only after
        the forward that read it has finished (its event) -- also across calls, the slots and events belong to the classifier."""
        if self.device.type != "cuda":
            raise RuntimeError("device-resident scoring needs the GPU")
        pad = PAD - CroppedSqueezeNet10.IN_LO if self.cropped is not None else PAD
        side = RESIZE + 2 * pad
        bs = self.batch_size
        # the cropped network's convolution kernels read channels-last: the input is written that way (a copy per forward less)
        nhwc = self.cropped is not None and self.cropped.memory_format == torch.channels_last
        xb, fb, done = self._device_slots(side, nhwc)

        def produce(slot, first):
            if done[slot] is not None:
                done[slot].synchronize()
            total, skipped = cut(xb[slot].data_ptr(), fb[slot].data_ptr(), bs, first, pad, nhwc)
            if skipped:
                raise RuntimeError("%d segment boxes were empty or larger than 512 pixels" % skipped)
            return total

        scores, frames_of = [], []
        total = produce(0, 0)
        first, i = 0, 0
        while first < total:
            slot = i & 1
            k = min(bs, total - first)
            scores.append(self._run(xb[slot][:self._bucket(k)], k, persistent=True).clone())     # rows past k hold an earlier chunk: scored, dropped
            frames_of.append(fb[slot][:k].clone())
            done[slot] = torch.cuda.Event()
            done[slot].record(torch.cuda.current_stream(self.device))
            first += k
            i += 1
            if first < total:
                produce(i & 1, first)
        if not scores:
            return torch.zeros((0, 2), device=self.device), torch.zeros((0,), dtype=torch.int32, device=self.device)
        return torch.cat(scores), torch.cat(frames_of)

    def scores_from_device(self, ctx, inp, frame_hw, segs, nseg, seg_cap, min_seg_size=(24, 24)):
        """The classifier on a whole batch_run without leaving the GPU: inp is the swk_input of that call (device
        frames), segs / nseg its device region records (torch tensors).  The crops of extract_segment_images
        (image_filtering.py:338-369) are cut, resized and normalised by swk_segment_inputs into the network's input
        tensor, batch_size segments at a time.  Returns (scores (T, 2), frame index (T,) int32), both on the device,
        segments in frame order then ascending label -- the order FrameQueue hands them to __call__."""
        def cut(net_ptr, frame_ptr, cap, first, pad, nhwc):
            return ctx.segment_inputs(inp, frame_hw, segs.data_ptr(), nseg.data_ptr(), seg_cap, IMAGENET_MEAN, IMAGENET_STD, net_ptr, cap,
                                      first=first, pad=pad, min_seg_size=min_seg_size, seg_frame_ptr=frame_ptr, channels_last=nhwc)
        return self._scores_device(cut)

    def predict_last_batch(self, ctx, generation, total, min_seg_size=(24, 24)):
        """Predicted class of every segment of the batch `ctx` ran last (FrameQueue.segment_queue's window), in batch order: an int64
        device tensor (total,), not waited for.  The inputs are cut from what that batch left on the device (swk_segment_inputs_last);
        raises _lib.StaleBatch when the context has moved on."""
        def cut(net_ptr, frame_ptr, cap, first, pad, nhwc):
            return ctx.segment_inputs_last(generation, IMAGENET_MEAN, IMAGENET_STD, net_ptr, cap, first=first, pad=pad,
                                           min_seg_size=min_seg_size, seg_frame_ptr=frame_ptr, channels_last=nhwc, known_total=total)
        scores, _ = self._scores_device(cut)
        if scores.shape[0] != total:
            raise RuntimeError("the device holds %d segments, the window has %d" % (scores.shape[0], total))
        return torch.max(scores, 1)[1]                                   # :36-39

    def classify_frames(self, frames):
        """__call__ for many frames with ONE scoring batch: every frame's segments replaced by the kept ones,
        relabelled 1..k per frame (:41-42)."""
        segs = [s for fr in frames for s in fr.segments]
        if not segs:
            return
        pred = self._window_predictions(segs)                 # segments of one segment_windows / segment_queue call: device-resident
        if pred is None:
            pred = torch.max(self.scores([s.segment_image for s in segs]), 1)[1].cpu().numpy()
        i = 0
        for fr in frames:
            kept = []
            for s in fr.segments:
                if pred[i] == 1:
                    kept.append(s)
                i += 1
            for j, s in enumerate(kept):
                s.label = j + 1
            fr.segments = kept

    def _window_predictions(self, segments):
        """Segments that FrameQueue.segment_queue made carry their window (data_structures.WindowBatch) and their index in it:
        the first call of a window scores ALL its segments in one device-resident batch, the others look their rows up.  None
        when the segments are not such a group or the window's device state is gone (the caller then scores their images)."""
        if self.device.type != "cuda":
            return None
        batch = getattr(segments[0], "_batch", None)
        if batch is None or any(getattr(s, "_batch", None) is not batch for s in segments):
            return None
        if (self.device.index or 0) != batch.ctx.device:
            return None
        table = batch.predictions(self)
        if table is None:
            return None
        return [table[s._index] for s in segments]

    def __call__(self, segments):
        """:26-44: keep segments whose argmax is class 1 (ties / all-zero scores give 0 and are
        dropped, as torch.max does), relabel the kept ones 1..k."""
        if not segments:
            return []
        pred = self._window_predictions(segments)
        if pred is None:
            score = self.scores([s.segment_image for s in segments])
            pred = torch.max(score, 1)[1].cpu().numpy()
        kept = [s for s, y in zip(segments, pred) if y == 1]
        for i, s in enumerate(kept):
            s.label = i + 1
        return kept

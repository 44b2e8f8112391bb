"""Encoder with the reference's interface: `VOSNet(model)`; `forward((B,3,H,W)) -> (B,256,ceil(H/8),ceil(W/8))`.

Mirrors reference src/model/vos_net.py:9-54 and the truncated ResNet of src/model/backbone/resnet.py:99-150
(children [0:8]: stem + layer1..layer4, layer3/layer4 at stride 1 so the output stride is 8;
`adjust_dim` 1x1 conv + `bn256` for resnet50/101).  Module / state-dict names are identical
(`backbone.{0,1,4,5,6,7}.*`, `adjust_dim.weight`, `bn256.*`) so reference checkpoints load unchanged.

Differences, on purpose:
  * construction never touches the network (the reference calls model_zoo.load_url at vos_net.py:17,20,25);
    weights come from `load_model` / `--resume` only.  `model='facebook'` needs torch.hub -> clear error.
  * it runs as stock PyTorch-ROCm modules (MIOpen); `prepare_for_inference` switches to channels_last + bf16/f16,
    the MI355X-friendly layout for the 1x1/3x3 convolutions.  The propagation step, not the encoder, is the
    hand-written HIP part of this project.
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}     # VOSPROP_DT_*


def bias_act_(y, bias, residual=None, relu=True):
    """In place: y = act(y + bias[c] (+ residual)).  On the GPU one pass of the hand-written epilogue kernel
    (vosprop_bias_act, csrc/encoder_ops.h) over the channels-last tensor; anywhere else the same thing with torch ops."""
    if (y.is_cuda and y.dtype in _DT and y.dim() == 4 and y.shape[1] % 8 == 0 and bias.dtype == y.dtype
            and y.is_contiguous(memory_format=torch.channels_last)
            and (residual is None or (residual.dtype == y.dtype and residual.shape == y.shape
                                      and residual.is_contiguous(memory_format=torch.channels_last)))):
        from . import _native
        rc = _native.lib().vosprop_bias_act(
            ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
            ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
            y.shape[0] * y.shape[2] * y.shape[3], y.shape[1], int(bool(relu)), _DT[y.dtype],
            ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
        if rc != 0:
            raise _native.VospropError(f'vosprop_bias_act failed ({rc})')
        return y
    y.add_(bias.view(1, -1, 1, 1))
    if residual is not None:
        y.add_(residual)
    return y.relu_() if relu else y


def bias_relu_maxpool(y, bias, pool):
    """pool(relu(y + bias[c])) for the stem (3x3 / stride 2 / pad 1 max-pool): on the GPU one pass of vosprop_bias_relu_maxpool
    (csrc/encoder_ops.h) - the network's largest activation is read once; anywhere else the separate steps.  Same bits."""
    if (y.is_cuda and y.dtype in _DT and y.dim() == 4 and y.shape[1] % 8 == 0 and bias.dtype == y.dtype
            and y.is_contiguous(memory_format=torch.channels_last) and isinstance(pool, nn.MaxPool2d)
            and (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False)):
        from . import _native
        n, c, h, w = y.shape
        out = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=y.dtype, device=y.device,
                          memory_format=torch.channels_last)
        rc = _native.lib().vosprop_bias_relu_maxpool(
            ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(out.data_ptr()), n, h, w, c,
            _DT[y.dtype], ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
        if rc != 0:
            raise _native.VospropError(f'vosprop_bias_relu_maxpool failed ({rc})')
        return out
    return pool(bias_act_(y, bias))


def _conv_nobias(x, conv):
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


_POINTWISE_GEMM = os.environ.get('VOSPROP_POINTWISE', '1') != '0'     # dev switch: 0 = every convolution through MIOpen


def _is_pointwise(conv):
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1
            and conv.in_channels % 8 == 0 and conv.out_channels % 8 == 0)


def conv_bias_act(x, conv, bias, residual=None, relu=True):
    """act(conv(x) + bias (+ residual)) for one folded convolution.  A pointwise convolution over a channels-last GPU tensor
    is one GEMM with the epilogue inside (vosprop_pointwise_conv, csrc/pointwise.h: the output is written once); anything else
    is the library convolution followed by one pass of bias_act_."""
    if (_POINTWISE_GEMM and x.is_cuda and x.dtype in _DT and x.dim() == 4 and _is_pointwise(conv)
            and x.is_contiguous(memory_format=torch.channels_last) and conv.weight.dtype == x.dtype
            and (bias is None or bias.dtype == x.dtype)
            and (residual is None or (residual.dtype == x.dtype and residual.is_contiguous(memory_format=torch.channels_last)))):
        from . import _native
        n, _, h, w = x.shape
        y = torch.empty((n, conv.out_channels, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        assert residual is None or residual.shape == y.shape
        rc = _native.lib().vosprop_pointwise_conv(
            ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(conv.weight.data_ptr()),
            ctypes.c_void_p(bias.data_ptr()) if bias is not None else None,
            ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
            ctypes.c_void_p(y.data_ptr()), n * h * w, conv.in_channels, conv.out_channels, int(bool(relu)), _DT[x.dtype],
            ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
        if rc == 0:
            return y
        if rc != -4:          # VOSPROP_E_UNSUPPORTED (no library kernel for this shape) falls through to the convolution
            raise _native.VospropError(f'vosprop_pointwise_conv failed ({rc})')
    y = _conv_nobias(x, conv)
    if bias is None and residual is None and not relu:
        return y
    if bias is None:
        bias = torch.zeros(conv.out_channels, dtype=y.dtype, device=y.device)
    return bias_act_(y, bias, residual, relu)

# (block kind, blocks per stage) - reference resnet.py:159-216
_ARCH = {
    'resnet18': ('basic', (2, 2, 2, 2)),
    'resnet34': ('basic', (3, 4, 6, 3)),
    'resnet50': ('bottleneck', (3, 4, 6, 3)),
    'resnet101': ('bottleneck', (3, 4, 23, 3)),
}
_STAGE_PLANES = (64, 128, 256, 256)     # layer4 has planes=256, not 512 (reference resnet.py:112)
_STAGE_STRIDES = (1, 2, 1, 1)           # layer3 / layer4 keep stride 1 (reference resnet.py:111-112)


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class ResidualUnit(nn.Module):
    """Basic (3x3,3x3) or bottleneck (1x1,3x3,1x1 with x4 expansion) residual unit.  Sub-module names follow
    the reference blocks (resnet.py:28-95): conv1/bn1/conv2/bn2[/conv3/bn3], downsample.{0,1}."""

    def __init__(self, kind, cin, planes, stride):
        super().__init__()
        self.kind = kind
        if kind == 'basic':
            cout = planes
            self.conv1, self.bn1 = _conv(cin, planes, 3, stride), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3), nn.BatchNorm2d(planes)
        else:
            cout = planes * 4
            self.conv1, self.bn1 = _conv(cin, planes, 1), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3, stride), nn.BatchNorm2d(planes)
            self.conv3, self.bn3 = _conv(planes, cout, 1), nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout))
        self.out_channels = cout

    fused = False     # set by VOSNet.prepare_for_inference: BatchNorm folded, epilogues through bias_act_

    def forward_fused(self, x):
        """Same arithmetic with each convolution's bias / ReLU / residual add done in one pass (BatchNorm must be folded)."""
        if self.downsample is None:
            skip, b_out = x, (self.conv3 if self.kind != 'basic' else self.conv2).bias
        else:
            skip, b_out = conv_bias_act(x, self.downsample[0], None, None, relu=False), self.bias_out
        y = conv_bias_act(x, self.conv1, self.conv1.bias)
        if self.kind == 'basic':
            return conv_bias_act(y, self.conv2, b_out, skip)
        y = conv_bias_act(y, self.conv2, self.conv2.bias)
        return conv_bias_act(y, self.conv3, b_out, skip)

    def forward(self, x):
        if self.fused:
            return self.forward_fused(x)
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.kind != 'basic':
            y = self.bn3(self.conv3(self.relu(y)))
        return self.relu(y + skip)


def _build_backbone(model):
    kind, depths = _ARCH[model]
    mods = [nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False), nn.BatchNorm2d(64),
            nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2, padding=1)]
    cin = 64
    for planes, stride, depth in zip(_STAGE_PLANES, _STAGE_STRIDES, depths):
        units = []
        for i in range(depth):
            u = ResidualUnit(kind, cin, planes, stride if i == 0 else 1)
            cin = u.out_channels
            units.append(u)
        mods.append(nn.Sequential(*units))
    return nn.Sequential(*mods), cin


class VOSNet(nn.Module):
    def __init__(self, model='resnet50'):
        super().__init__()
        self.model = model
        if model == 'facebook':
            raise NotImplementedError("model='facebook' needs torch.hub.load of a remote repository "
                                      '(reference vos_net.py:30); not available offline')
        if model not in ('resnet18', 'resnet50', 'resnet101'):
            raise NotImplementedError(model)
        self.backbone, cout = _build_backbone(model)
        if model != 'resnet18':
            self.adjust_dim = nn.Conv2d(cout, 256, kernel_size=1, stride=1, padding=0, bias=False)
            self.bn256 = nn.BatchNorm2d(256)
        self._init_weights()

    def _init_weights(self):
        # He-normal convolutions, unit BatchNorm (reference resnet.py:116-122); adjust_dim/bn256 keep torch defaults
        for m in self.backbone.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                nn.init.normal_(m.weight, 0.0, (2.0 / n) ** 0.5)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    fused = False
    feature_dtype = None      # set by prepare_for_inference(feature_dtype=...): element type the features are handed over in

    def forward(self, x):
        if self.fused:
            bb = self.backbone
            x = bias_relu_maxpool(_conv_nobias(x, bb[0]), bb[0].bias, bb[3])      # stem conv; folded BN + ReLU + max-pool
            for stage in list(bb)[4:]:
                x = stage(x)
            if self.model != 'resnet18':
                x = conv_bias_act(x, self.adjust_dim, self.adjust_dim.bias, None, relu=False)
            if self.feature_dtype is not None and x.dtype != self.feature_dtype:
                x = x.to(self.feature_dtype)      # one element-wise pass over the batch, inside the captured graph
            return x
        x = self.backbone(x)
        if self.model != 'resnet18':
            x = self.bn256(self.adjust_dim(x))
        return x

    def freeze_feature_extraction(self):
        self.backbone.requires_grad_(False)

    def fold_batchnorm(self):
        """Inference-only: fold every eval-mode BatchNorm into the convolution in front of it (w' = w * g/sqrt(v+eps),
        b' = beta - mu * g/sqrt(v+eps), folded in f32) and replace it by Identity.  Removes ~50 element-wise launches per
        frame; outputs are equal up to f32 rounding.  Call after the checkpoint is loaded."""
        def fold(conv, bn):
            w = conv.weight.detach().float()
            scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
            fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                              conv.dilation, conv.groups, bias=True)
            fused.weight.data = (w * scale.view(-1, 1, 1, 1)).to(conv.weight.dtype)
            b0 = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(scale)
            fused.bias.data = (bn.bias.detach().float() + (b0 - bn.running_mean.detach().float()) * scale).to(conv.weight.dtype)
            return fused.to(conv.weight.device)

        bb = self.backbone
        bb[0], bb[1] = fold(bb[0], bb[1]), nn.Identity()
        for stage in list(bb)[4:]:
            for u in stage:
                u.conv1, u.bn1 = fold(u.conv1, u.bn1), nn.Identity()
                u.conv2, u.bn2 = fold(u.conv2, u.bn2), nn.Identity()
                if u.kind != 'basic':
                    u.conv3, u.bn3 = fold(u.conv3, u.bn3), nn.Identity()
                if u.downsample is not None:
                    u.downsample = nn.Sequential(fold(u.downsample[0], u.downsample[1]), nn.Identity())
        if self.model != 'resnet18':
            self.adjust_dim, self.bn256 = fold(self.adjust_dim, self.bn256), nn.Identity()
        return self

    def prepare_for_inference(self, device, dtype=torch.bfloat16, fold_bn=True, fuse_epilogue=True, miopen_find=False,
                              feature_dtype=None):
        """eval + folded BatchNorm + channels_last + reduced-precision weights on `device` (the reference runs the
        encoder under torch.cuda.amp.autocast = fp16 on GPU, inference_utils.py:35,52).  fuse_epilogue: bias + residual add +
        ReLU after every convolution as one pass (bias_act_) instead of two or three element-wise kernels.  miopen_find: let
        MIOpen time its solvers for every new convolution shape (torch.backends.cudnn.benchmark, a process-wide switch) instead
        of taking the immediate-mode pick - measured +16 % end to end at batch 32 on MI355X (1 225 -> 1 425 frames/s in
        bench.py), but the search costs 10-20 s per new batch shape in every process (it is not cached across processes on this
        stack), more than a whole DAVIS-sized job takes: off by default, on in bench.py (search in the untimed warm-up) and
        with `main.py inference --miopen-find` for long jobs.  feature_dtype (fused path only): the features leave the encoder in
        this type (one element-wise pass over the batch inside the captured graph, 3.8 us per 480p frame measured).  Not used by
        the CLI or the bench any more: the engine reads channels-last f16 as well as bf16 features in place (vosprop_step)."""
        self.feature_dtype = feature_dtype
        if miopen_find and torch.device(device).type == 'cuda':
            torch.backends.cudnn.benchmark = True
        self.eval().to(device)
        if fold_bn:
            self.fold_batchnorm()
        if dtype is not None and dtype != torch.float32:
            self.to(dtype)
        self.to(memory_format=torch.channels_last)
        if fold_bn and fuse_epilogue:
            for stage in list(self.backbone)[4:]:
                for u in stage:
                    if u.downsample is not None:      # the shortcut's bias joins the last convolution's: one combined add
                        last = u.conv3 if u.kind != 'basic' else u.conv2
                        u.bias_out = (last.bias.detach().float() + u.downsample[0].bias.detach().float()).to(last.bias.dtype)
                    u.fused = True
            self.fused = True
        return self


class GraphedEncoder:
    """The encoder forward captured in HIP graphs (torch.cuda.CUDAGraph), one per input shape (up to `max_graphs`: the full
    look-ahead batch, the tail batch of a video, ...): one graph launch instead of ~330 eager kernel launches per batch - at
    16-32 frames per call the eager launches cost the host about the GPU time of the whole batch, so the loop is host-bound
    without this.  Shapes beyond the limit, CPU tensors and non-channels-last inputs run the eager module.
    The output buffer of a shape is reused by the next call with that shape: consume (enqueue the readers of) one batch before
    asking for the next, on the stream the graph is replayed on - which is how the frame loops use it."""

    def __init__(self, net, warmup=3, max_graphs=4):
        self.net = net
        self.warmup = warmup
        self.max_graphs = max_graphs
        self.graphs = {}           # (shape, dtype) -> (graph, static input, static output)
        self.failed = False

    @property
    def graph(self):               # any captured graph (tests / introspection)
        return next(iter(self.graphs.values()))[0] if self.graphs else None

    def _capture(self, x):
        s = torch.cuda.Stream(x.device)
        s.wait_stream(torch.cuda.current_stream(x.device))
        xs = torch.empty_like(x)           # keeps memory format (channels_last)
        xs.copy_(x)
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(self.warmup):   # MIOpen's algorithm search and workspace growth happen here, outside the capture
                self.net(xs)
        torch.cuda.current_stream(x.device).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        # thread_local: calls made by other threads of the process (e.g. RCCL's watchdog under torch.distributed) must not
        # invalidate the capture
        with torch.cuda.graph(g, capture_error_mode='thread_local'), torch.no_grad():
            ys = self.net(xs)
        self.graphs[(tuple(x.shape), x.dtype)] = (g, xs, ys)

    def __call__(self, x):
        if not x.is_cuda or self.failed or not x.is_contiguous(memory_format=torch.channels_last):
            return self.net(x)
        key = (tuple(x.shape), x.dtype)
        if key not in self.graphs:
            if len(self.graphs) >= self.max_graphs:
                return self.net(x)
            try:
                self._capture(x)
            except Exception as e:         # capture is an optimisation: never a reason to fail the run - but never silent
                self.failed = True
                torch.cuda.synchronize()
                import sys
                print(f'[vosprop] HIP-graph capture of the encoder failed for input {tuple(x.shape)} {x.dtype} '
                      f'({type(e).__name__}: {e}); running it eagerly from here on (~2x more host time per batch)',
                      file=sys.stderr, flush=True)
                return self.net(x)
        g, xs, ys = self.graphs[key]
        xs.copy_(x)
        g.replay()
        return ys

    def eval(self):
        return self

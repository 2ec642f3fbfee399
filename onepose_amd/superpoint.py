"""MI355X-native SuperPoint extractor: host-side mirror of the reference module.

Drop-in for ``src/models/extractors/SuperPoint/superpoint.py::SuperPoint`` (reference :96-197): same
constructor (``SuperPoint(config)``), same parameter names / shapes (``conv1a.weight`` ...
``convDb.bias``; a reference ``superpoint_v1.pth`` loads with ``strict=True``), same ``forward(image)``
contract -- ``{'keypoints': [ [n,2] (x,y) float ], 'scores': [ [n] ], 'descriptors': [ [256,n] ]}``, one
list entry per image -- with every stage running as hand-written HIP kernels behind the C ABI of
``include/superpoint.h``.  The modules below are parameter containers; there is no PyTorch compute
path and no CPU fallback.
"""
from __future__ import annotations

import ctypes

import torch
from torch import nn

from . import _native_spp
from ._native import NativeError

LAYERS = (  # (name, out, in, k) -- reference :115-133
    ("conv1a", 64, 1, 3), ("conv1b", 64, 64, 3), ("conv2a", 64, 64, 3), ("conv2b", 64, 64, 3),
    ("conv3a", 128, 64, 3), ("conv3b", 128, 128, 3), ("conv4a", 128, 128, 3), ("conv4b", 128, 128, 3),
    ("convPa", 256, 128, 3), ("convPb", 65, 256, 1), ("convDa", 256, 128, 3), ("convDb", 256, 256, 1),
)


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _on_device(fn):
    """Run an engine method with the CURRENT HIP device set to the device of its first tensor argument (the C ABI takes a
    stream handle but launches on the current device)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = next((a.device for a in args if torch.is_tensor(a)), None)
        if dev is None or dev.type != "cuda":
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    return wrapper


class SuperPointEngine:
    """Owns the packed weights and workspaces of one module on one device; workspaces are cached per (shape, device,
    stream), so one module can be used from several streams at once."""

    def __init__(self, module):
        self.module = module
        self.lib = _native_spp.load()
        self._packed = None
        self._packed_key = None
        self._packed_event = None      # recorded on the packing stream right after spp_pack_weights
        self._packed_stream = None
        self._ws = {}

    def _params(self):
        m = self.module
        return [getattr(m, n).weight for n, *_ in LAYERS], [getattr(m, n).bias for n, *_ in LAYERS]

    def packed_weights(self, device):
        ws, bs = self._params()
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in ws + bs)
        if self._packed is not None and key == self._packed_key:
            cur = torch.cuda.current_stream(device)
            if cur.cuda_stream != self._packed_stream:     # another stream: order its reads behind the pack / split kernels
                cur.wait_event(self._packed_event)
            return self._packed
        if self._packed is not None:
            torch.cuda.synchronize(self._packed.device)    # re-pack: nobody may still be reading the blob that is dropped below
        for p in ws + bs:
            if not p.is_cuda:
                raise RuntimeError(f"onepose_amd.SuperPoint runs only on a ROCm GPU (a parameter is on {p.device}); "
                                   "there is no CPU fallback -- move the module to the GPU")
        keep_w = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in ws]
        keep_b = [p.detach().to(device=device, dtype=torch.float32).contiguous() for p in bs]
        raw = _native_spp.RawWeights()
        for i in range(_native_spp.NUM_LAYERS):
            raw.weight[i], raw.bias[i] = keep_w[i].data_ptr(), keep_b[i].data_ptr()
        packed = torch.empty(self.lib.spp_packed_weights_bytes() // 4, device=device, dtype=torch.float32)
        with torch.cuda.device(device):
            _native_spp.check(self.lib.spp_pack_weights(ctypes.byref(raw), packed.data_ptr(), _stream(device)), "spp_pack_weights")
            self._packed_event = torch.cuda.Event()
            self._packed_event.record(torch.cuda.current_stream(device))
            self._packed_stream = torch.cuda.current_stream(device).cuda_stream
        # keep_* may be released here: the caching allocator is stream-ordered and the packing kernels were
        # enqueued on this stream
        self._packed, self._packed_key = packed, key
        return packed

    def workspace(self, b, h, w, device):
        key = (b, h, w, str(device), torch.cuda.current_stream(device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.lib.spp_workspace_bytes(b, h, w)
            if nbytes == 0:
                raise NativeError("spp_workspace_bytes: " + self.lib.spp_last_error().decode())
            if len(self._ws) >= 6:
                self._ws.clear()
            ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
            self._ws[key] = ws
        return ws

    def flags(self):
        return _native_spp.PRECISIONS[self.module.precision]

    # ---- stages (tests) ----
    @_on_device
    def dense(self, image):
        b, _, h, w = image.shape
        dev = image.device
        ws = self.workspace(b, h, w, dev)
        score = torch.empty(b, h // 8 * 8, w // 8 * 8, device=dev, dtype=torch.float32)
        dense = torch.empty(b, 256, h // 8, w // 8, device=dev, dtype=torch.float32)
        _native_spp.check(self.lib.spp_dense(self.packed_weights(dev).data_ptr(), image.data_ptr(), b, h, w, score.data_ptr(),
                                             dense.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev), self.flags()), "spp_dense")
        return score, dense

    def _outputs(self, b, capacity, dev):
        return (torch.empty(b, capacity, 2, device=dev, dtype=torch.float32),
                torch.empty(b, capacity, device=dev, dtype=torch.float32),
                torch.empty(b, 256, capacity, device=dev, dtype=torch.float32),
                torch.empty(b, 2, device=dev, dtype=torch.int32))

    def _capacity(self, cfg, h, w, capacity):
        if capacity is not None:
            return int(capacity)
        mk = cfg["max_keypoints"]
        # -1 = keep everything: NMS survivors are more than `radius` apart (plateaus aside); retried at H*W on overflow
        return mk if mk >= 0 else max(1024, (h * w) // ((cfg["nms_radius"] + 1) ** 2))

    @_on_device
    def detect(self, score, dense, cfg, align_corners, capacity=None, return_nms=False):
        b, h, w = score.shape
        dev = score.device
        ws = self.workspace(b, h, w, dev)
        cap = self._capacity(cfg, h, w, capacity)
        kp, sc, de, cnt = self._outputs(b, cap, dev)
        nms = torch.empty_like(score) if return_nms else None
        _native_spp.check(self.lib.spp_detect(
            score.data_ptr(), dense.data_ptr(), b, h, w, cfg["nms_radius"], cfg["keypoint_threshold"], cfg["max_keypoints"],
            cfg["remove_borders"], int(align_corners), cap, kp.data_ptr(), sc.data_ptr(), de.data_ptr(), cnt.data_ptr(),
            nms.data_ptr() if return_nms else None, ws.data_ptr(), ws.numel(), _stream(dev)), "spp_detect")
        return kp, sc, de, cnt, nms

    @_on_device
    def forward(self, image, cfg, align_corners, capacity=None):
        b, _, h, w = image.shape
        dev = image.device
        ws = self.workspace(b, h, w, dev)
        cap = self._capacity(cfg, h, w, capacity)
        kp, sc, de, cnt = self._outputs(b, cap, dev)
        _native_spp.check(self.lib.spp_forward(
            self.packed_weights(dev).data_ptr(), image.data_ptr(), b, h, w, cfg["nms_radius"], cfg["keypoint_threshold"],
            cfg["max_keypoints"], cfg["remove_borders"], int(align_corners), cap, kp.data_ptr(), sc.data_ptr(), de.data_ptr(),
            cnt.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev), self.flags()), "spp_forward")
        return kp, sc, de, cnt


class SuperPoint(nn.Module):
    """SuperPoint detector / descriptor (reference :96-197) on HIP kernels.

    ``align_corners``: the reference picks it from ``int(torch.__version__[2]) > 2`` (:87) -- True on the torch
    1.x builds OnePose pins (its environment.yaml), which is the default here; pass False to reproduce what the
    same line yields on torch >= 1.10 / 2.x.

    ``precision`` (keyword, not part of the reference signature; also settable as an attribute): arithmetic of the GEMM
    convolutions -- ``"fp32"`` (default: exact fp32 MFMA, the reference's arithmetic) or ``"fp16x4"`` (two fp16 terms per operand,
    all four products: fp32-class results at a quarter of the matrix-pipe time; the matcher's mode of the same name).  It travels
    to the library as a bit of the ``flags`` argument of the C ABI; nothing is read from the environment."""

    default_config = {
        "descriptor_dim": 256,
        "nms_radius": 4,
        "keypoint_threshold": 0.005,
        "max_keypoints": -1,
        "remove_borders": 4,
    }

    def __init__(self, config=None, align_corners=True, precision="fp32"):
        super().__init__()
        self.precision = precision
        self.config = {**self.default_config, **(config or {})}
        if self.config["descriptor_dim"] != 256:
            raise ValueError("onepose_amd.SuperPoint supports descriptor_dim=256 (the reference default and the matcher's input)")
        for name, oc, ic, k in LAYERS:
            setattr(self, name, nn.Conv2d(ic, oc, kernel_size=k, stride=1, padding=k // 2))
        mk = self.config["max_keypoints"]
        if mk == 0 or mk < -1:
            raise ValueError('"max_keypoints" must be positive or "-1"')       # reference :135-137
        self.align_corners = bool(align_corners)
        self._engine = None

    @property
    def precision(self):
        return self._precision

    @precision.setter
    def precision(self, value):
        if value not in _native_spp.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_native_spp.PRECISIONS)} (got {value!r})")
        self._precision = value

    @property
    def engine(self):
        if self._engine is None:
            self._engine = SuperPointEngine(self)
        return self._engine

    def _check_image(self, inp):
        if not isinstance(inp, torch.Tensor) or inp.dim() != 4 or inp.shape[1] != 1:
            raise ValueError("expected a grayscale image batch [b, 1, H, W]")
        if not inp.is_cuda:
            raise RuntimeError(f"onepose_amd.SuperPoint runs only on a ROCm GPU (the image is on {inp.device}); "
                               "there is no CPU fallback")
        return inp.to(torch.float32).contiguous()

    @torch.no_grad()
    def forward_device(self, inp, capacity=None):
        """Batched outputs left on the GPU without a host round trip: (keypoints [b,cap,2], scores [b,cap],
        descriptors [b,256,cap], counts int32 [b,2]); only the first counts[i,0] slots of image i are defined."""
        img = self._check_image(inp)
        return self.engine.forward(img, self.config, self.align_corners, capacity)

    @torch.no_grad()
    def forward(self, inp):
        """Compute keypoints, scores, descriptors for image (reference :140-197)."""
        img = self._check_image(inp)
        kp, sc, de, cnt = self.engine.forward(img, self.config, self.align_corners)
        counts = cnt.cpu()                       # the reference synchronises here too (torch.nonzero, :165)
        cap = kp.shape[1]
        if self.config["max_keypoints"] < 0 and int(counts[:, 1].max()) > cap:
            kp, sc, de, cnt = self.engine.forward(img, self.config, self.align_corners, capacity=img.shape[2] * img.shape[3])
            counts = cnt.cpu()
        out = {"keypoints": [], "scores": [], "descriptors": []}
        for i in range(img.shape[0]):
            n = int(counts[i, 0])
            out["keypoints"].append(kp[i, :n])
            out["scores"].append(sc[i, :n])
            out["descriptors"].append(de[i, :, :n])
        return out

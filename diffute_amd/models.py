"""Host-side mirror of the diffusers classes the reference drives.

`UNet2DConditionModel` and `AutoencoderKL` keep the call surface train_diffute_v1.py / app.ipynb
use - `unet(sample, timestep, encoder_hidden_states).sample`, `vae.encode(x).latent_dist.sample()`,
`vae.decode(z).sample`, `vae(x)["sample"]`, `.config`, `.parameters()`, `state_dict()` with
diffusers keys, `from_pretrained / save_pretrained` - while every FLOP runs in the gfx950 C-ABI
library (diffute_amd/lib/libdiffute_hip.so).  torch is plumbing only: it owns device memory and
the stream.  There is no CPU or eager fallback.
"""
import ctypes
import json
import math
import os
from collections import OrderedDict
from types import SimpleNamespace

import torch
from torch import nn

from . import _cabi
from .init import init_param

SD2_INPAINT_UNET_CONFIG = dict(
    in_channels=9, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024, norm_num_groups=32,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    use_linear_projection=True, flip_sin_to_cos=True, freq_shift=0, norm_eps=1e-5, act_fn="silu",
    sample_size=64)

SD_VAE_CONFIG = dict(
    in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
    layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215, act_fn="silu", sample_size=512)


# diffusers config.json keys whose non-default values select arithmetic this library does not implement: a checkpoint that
# sets one of them differently is refused by from_pretrained / the constructor instead of being run as something else.
_UNET_ONLY_SUPPORTED = dict(
    act_fn="silu", center_input_sample=False, conv_in_kernel=3, conv_out_kernel=3, downsample_padding=1, dual_cross_attention=False,
    flip_sin_to_cos=True, mid_block_scale_factor=1, mid_block_type="UNetMidBlock2DCrossAttn", only_cross_attention=False,
    resnet_out_scale_factor=1.0, resnet_skip_time_act=False, resnet_time_scale_shift="default", time_embedding_type="positional",
    upcast_attention=False, use_linear_projection=True, class_embeddings_concat=False)
_UNET_MUST_BE_NONE = ("addition_embed_type", "class_embed_type", "cross_attention_norm", "encoder_hid_dim", "mid_block_only_cross_attention",
                      "num_class_embeds", "projection_class_embeddings_input_dim", "time_cond_proj_dim", "time_embedding_act_fn",
                      "time_embedding_dim", "timestep_post_act")
_VAE_ONLY_SUPPORTED = dict(act_fn="silu")


def _check_supported(kind, cfg, only, must_be_none=()):
    for k, want in only.items():
        if k in cfg and cfg[k] is not None and cfg[k] != want:
            raise NotImplementedError(f"{kind}: config {k}={cfg[k]!r} is not implemented (only {want!r})")
    for k in must_be_none:
        if cfg.get(k) is not None:
            raise NotImplementedError(f"{kind}: config {k}={cfg[k]!r} is not implemented (only null)")


class _Config(SimpleNamespace):
    def __getitem__(self, k):
        return getattr(self, k)

    def to_dict(self):
        return {k: (list(v) if isinstance(v, tuple) else v) for k, v in vars(self).items()}


class UNet2DConditionOutput(SimpleNamespace):
    """`.sample` (train_diffute_v1.py:913, app.ipynb:814)."""


class DecoderOutput(SimpleNamespace):
    """`.sample` (app.ipynb:819)."""


class DiagonalGaussianDistribution:
    """`vae.encode(x).latent_dist` (app.ipynb:781,793): sample() / mode() run dmx_gaussian_sample."""

    def __init__(self, parameters):
        self.parameters = parameters                     # moments NCHW fp32 [B, 2C, h, w]
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)

    def _run(self, noise):
        lib = _cabi.lib()
        B, C2, h, w = self.parameters.shape
        out = torch.empty(B, C2 // 2, h, w, dtype=torch.float32, device=self.parameters.device)
        _cabi.check(lib.dmx_gaussian_sample(_cabi.ptr(self.parameters), _cabi.ptr(noise), _cabi.ptr(out),
                                            B, C2 // 2, h * w, 1.0, _cabi.current_stream()), "gaussian_sample")
        return out

    def sample(self, generator=None, noise=None):
        if noise is None:
            B, C2, h, w = self.parameters.shape
            noise = torch.randn(B, C2 // 2, h, w, generator=generator, device=self.parameters.device, dtype=torch.float32)
        return self._run(noise.to(torch.float32).contiguous())

    def mode(self):
        return self._run(None)


class AutoencoderKLOutput(SimpleNamespace):
    """`.latent_dist`."""


class _Node(nn.Module):
    """Plain container; children / parameters are attached by dotted state-dict key."""


def _attach(root: nn.Module, key: str, param: nn.Parameter):
    parts = key.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, _Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class _HipModel(nn.Module):
    """Shared plumbing: parameter tree from the C library's table, packed-weights arena, workspace."""
    _kind = None         # "unet" | "vae"

    def _create_handle(self, elem):
        l = _cabi.lib(elem)
        h = getattr(l, f"dmx_{self._kind}_create")(ctypes.byref(self._cstruct))
        if not h:
            raise ValueError(f"{type(self).__name__}: " + l.dmx_last_error().decode())
        return h

    @property
    def _lib(self):
        """the build of the C-ABI library this model runs on: bf16 (default) or the fp16 build after `.to(dtype=torch.float16)`"""
        return _cabi.lib(self._elem)

    @property
    def compute_dtype(self):
        """torch dtype of the 16-bit storage / MFMA operands (accumulation is fp32 either way)"""
        return _cabi.torch_elem(self._elem)

    def _switch_build(self, elem):
        """re-create the handle in the other build of the library; Parameters (fp32 masters) are untouched, the packed arena,
        workspaces, context caches and captured graphs belong to the old handle and are dropped"""
        if elem == self._elem:
            return
        if getattr(self, "_fused", None) is not None or getattr(self, "_tb", None) is not None:
            raise NotImplementedError(f"{type(self).__name__}: changing the compute type after training state exists is not supported")
        torch.cuda.synchronize() if torch.cuda.is_initialized() else None
        new_h = self._create_handle(elem)                 # (first the new handle: a missing build / failed create leaves the model as it was)
        old_h, old_elem = self._h, self._elem
        self._h, self._elem = new_h, elem
        getattr(_cabi.lib(old_elem), f"dmx_{self._kind}_destroy")(old_h)
        self._arena = None; self._packed_sig = None; self._ws = None
        for attr in ("_masters32", "_train"):
            if hasattr(self, attr):
                setattr(self, attr, None)
        if hasattr(self, "_slots"):
            self._slots = {}

    def _setup(self, handle, seed, device):
        self._h = handle
        self._elem = "bf16"
        self._dtype = torch.float32
        self._arena = None
        self._packed_sig = None
        self._ws = None
        lib = self._lib
        n = getattr(lib, f"dmx_{self._kind}_param_count")(self._h)
        self._keys = []
        name = ctypes.c_char_p()
        shape = (ctypes.c_int * 4)()
        for i in range(n):
            _cabi.check(getattr(lib, f"dmx_{self._kind}_param_info")(self._h, i, ctypes.byref(name), ctypes.byref(shape)), "param_info")
            key = name.value.decode()
            shp = tuple(int(s) for s in shape if s > 0)
            self._keys.append(key)
            _attach(self, key, nn.Parameter(init_param(key, shp, seed=seed, device=device), requires_grad=True))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                getattr(_cabi.lib(self._elem), f"dmx_{self._kind}_destroy")(self._h)
                self._h = None
        except Exception:
            pass

    # ---- reference-visible knobs that are no-ops here
    def enable_xformers_memory_efficient_attention(self, *a, **k):
        """train_diffute_v1.py:657 - attention is always the fused flash-style HIP kernel."""

    def enable_gradient_checkpointing(self):
        """train_diffute_v1.py:696 - unnecessary with 288 GB of HBM; accepted for compatibility."""

    def register_to_config(self, **kw):
        for k, v in kw.items():
            setattr(self.config, k, v)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return self._dtype

    def to(self, *args, **kwargs):
        """`.to(device, dtype=weight_dtype)` (train_diffute_v1.py:789-797): device moves are honoured; master parameters stay
        fp32 and accumulation is fp32 whatever the dtype.  The dtype selects the BUILD of the library that computes:
        torch.float16 -> the fp16 build (fp16 storage, v_mfma_f32_32x32x16_f16; BASELINE configs[4]); torch.bfloat16 /
        torch.float32 -> the bf16 build.  It is also the dtype low-precision inputs get their outputs back in."""
        dtype = kwargs.pop("dtype", None)
        args = list(args)
        for a in list(args):
            if isinstance(a, torch.dtype):
                dtype = a; args.remove(a)
        if dtype is not None:
            self._dtype = dtype
            self._switch_build(_cabi.elem_of(dtype))
        if args or kwargs:
            super().to(*args, **kwargs)
            self._packed_sig = None
        return self

    # ---- packed weights
    def _signature(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def mark_parameters_changed(self):
        """Tell the model its Parameters were rewritten in a way torch does not record.  The packed weights arena (and a
        fused optimizer's fp32 masters) follow the Parameters through (data_ptr, _version); an in-place write through
        `p.data` (`p.data.copy_(...)`, diffusers' `EMAModel.copy_to`, `dist.broadcast(p.data)`) bumps neither, so the next
        forward would still run the old weights.  After such a write call this once; the next forward re-packs the arena and
        re-imports the masters.  Writes on the Parameter itself under `torch.no_grad()`, `load_state_dict`, `.to()` and
        optimizer steps are detected without it."""
        self._packed_sig = None

    def _ensure_packed(self):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("diffute_amd: model parameters must be on the GPU (call .cuda()); there is no CPU path")
        sig = self._signature()
        if self._packed_sig == sig and self._arena is not None:
            return                                   # (a fused optimizer updates the arena in place and leaves the Parameters alone)
        lib = self._lib
        k = self._kind
        nbytes = getattr(lib, f"dmx_{k}_arena_bytes")(self._h)
        if self._arena is None or self._arena.device != dev:
            self._arena = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        _cabi.check(getattr(lib, f"dmx_{k}_bind_arena")(self._h, _cabi.ptr(self._arena), nbytes), "bind_arena")
        st = _cabi.current_stream()
        sd = dict(self.named_parameters())
        for key in self._keys:
            src = sd[key].detach().to(torch.float32).contiguous()
            _cabi.check(getattr(lib, f"dmx_{k}_load_param")(self._h, key.encode(), _cabi.ptr(src), st), f"load_param({key})")
        self._finalize(st)
        self._packed_sig = sig
        fused = getattr(self, "_fused", None)
        if fused is not None:                        # the Parameters changed under a fused optimizer (load_state_dict, an in-place
            fused.reimport_masters()                 # write under no_grad, mark_parameters_changed()): they are the new master copy

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def state_dict(self, *args, **kwargs):
        f = getattr(self, "_fused", None)
        if f is not None:
            f.sync_to_model()
        return super().state_dict(*args, **kwargs)

    # ---- (de)serialisation in the diffusers directory layout (train_diffute_v1.py:664-690)
    def save_pretrained(self, save_directory):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(dict(self.config.to_dict(), _class_name=type(self).__name__), f, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()},
                  os.path.join(save_directory, "diffusion_pytorch_model.safetensors"))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, revision=None, **kw):
        d = pretrained_model_name_or_path if subfolder is None else os.path.join(pretrained_model_name_or_path, subfolder)
        with open(os.path.join(d, "config.json")) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        model = cls(**cfg)
        st = os.path.join(d, "diffusion_pytorch_model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "diffusion_pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(cls._convert_legacy_keys(sd))
        return model

    @staticmethod
    def _convert_legacy_keys(sd):
        """Linear weights some exporters store as 1x1 convolutions ([C_out, C_in, 1, 1]) are squeezed."""
        out = OrderedDict()
        for k, v in sd.items():
            if v.ndim == 4 and v.shape[2:] == (1, 1) and (k.endswith("proj_in.weight") or k.endswith("proj_out.weight")):
                v = v[:, :, 0, 0]
            out[k] = v
        return out


class UNet2DConditionModel(_HipModel):
    """SD2-inpainting UNet (reference: train_diffute_v1.py:633-635, app.ipynb:551-553)."""
    _kind = "unet"

    def __init__(self, seed=1234, device="cpu", **config):
        super().__init__()
        cfg = dict(SD2_INPAINT_UNET_CONFIG); cfg.update(config)
        _check_supported("UNet2DConditionModel", cfg, _UNET_ONLY_SUPPORTED, _UNET_MUST_BE_NONE)
        for bt in tuple(cfg["down_block_types"]) + tuple(cfg["up_block_types"]):
            if bt not in ("CrossAttnDownBlock2D", "DownBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"):
                raise NotImplementedError(f"UNet2DConditionModel: block type {bt!r} is not implemented")
        if isinstance(cfg["attention_head_dim"], int):       # diffusers accepts one int for every level
            cfg["attention_head_dim"] = (cfg["attention_head_dim"],) * len(cfg["block_out_channels"])
        for k in ("block_out_channels", "attention_head_dim", "down_block_types", "up_block_types"):
            cfg[k] = tuple(cfg[k])
        self.config = _Config(**cfg)
        if len(cfg["block_out_channels"]) != 4:
            raise ValueError("UNet2DConditionModel: exactly 4 resolution levels are supported")
        c = _cabi.UNetConfig()
        c.in_channels = cfg["in_channels"]; c.out_channels = cfg["out_channels"]
        c.layers_per_block = cfg["layers_per_block"]; c.cross_attention_dim = cfg["cross_attention_dim"]
        c.norm_num_groups = cfg["norm_num_groups"]
        for i in range(4):
            c.block_out_channels[i] = cfg["block_out_channels"][i]
            c.heads[i] = cfg["attention_head_dim"][i]
            c.down_has_attn[i] = int(cfg["down_block_types"][i].startswith("CrossAttn"))
            c.up_has_attn[i] = int(cfg["up_block_types"][i].startswith("CrossAttn"))
        self._cstruct = c
        self._slots = {}
        self._setup(self._create_handle("bf16"), seed, device)

    def _finalize(self, st):
        half = self.config.block_out_channels[0] // 2
        # diffusers get_timestep_embedding: exp(-ln(10000) * arange(half) / (half - freq_shift))
        exponent = -math.log(10000) * torch.arange(start=0, end=half, dtype=torch.float32)
        freq = torch.exp(exponent / (half - self.config.freq_shift)).contiguous()
        self._freq_host = freq
        _cabi.check(self._lib.dmx_unet_finalize(self._h, ctypes.c_void_p(freq.data_ptr()), st), "unet_finalize")
        for sl in self._slots.values():
            sl["ctx_key"] = None

    # ---- execution slots: independent (workspace, context cache) pairs so that several micro-batches can be in
    # flight on different streams at once (pipeline.denoise(..., micro_batches=n)); slot 0 serves the plain API.
    def _slot(self, i):
        sl = self._slots.get(i)
        if sl is None:
            sl = self._slots[i] = dict(ws=None, ctx_cache=None, ctx_key=None, ctx_shape=None, ws_need=None)
        return sl

    def _slot_workspace(self, sl, nbytes):
        if sl["ws"] is None or sl["ws"].numel() < nbytes or sl["ws"].device != self.device:
            sl["ws"] = None
            sl["ws"] = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return sl["ws"]

    # ---- glyph-context K/V cache (constant across denoise steps, app.ipynb:776,814)
    def set_context(self, encoder_hidden_states, slot=0):
        self._ensure_packed()
        sl = self._slot(slot)
        ctx = encoder_hidden_states
        _cabi.require_cuda(ctx)
        if ctx.dtype not in (torch.float32, self.compute_dtype):
            ctx = ctx.to(torch.float32)
        ctx = ctx.contiguous()
        B, S, D = ctx.shape
        if D != self.config.cross_attention_dim:
            raise ValueError(f"encoder_hidden_states last dim {D} != cross_attention_dim {self.config.cross_attention_dim}")
        lib = self._lib
        nb = lib.dmx_unet_context_bytes(self._h, B, S)
        if sl["ctx_cache"] is None or sl["ctx_cache"].numel() < nb or sl["ctx_cache"].device != ctx.device:
            sl["ctx_cache"] = torch.empty(nb, dtype=torch.uint8, device=ctx.device)
        ws = self._slot_workspace(sl, lib.dmx_unet_workspace_bytes(self._h, B, 8, 8, S))
        _cabi.check(lib.dmx_unet_set_context(self._h, _cabi.ptr(ctx), int(ctx.dtype == self.compute_dtype), B, S,
                                             _cabi.ptr(sl["ctx_cache"]), sl["ctx_cache"].numel(),
                                             _cabi.ptr(ws), ws.numel(), _cabi.current_stream()), "unet_set_context")
        # identity + version of the tensor object the K/V were projected from.  The strong reference keeps the allocator from
        # handing the same address to a different tensor (a key on data_ptr would then silently reuse stale K/V).
        sl["ctx_key"] = (encoder_hidden_states, encoder_hidden_states._version)
        sl["ctx_shape"] = (B, S)

    def temb_table(self, timesteps_dev):
        """[T][sum of resnet widths] fp32: the time-embedding MLP + every resnet's time_emb_proj for ALL the (scalar) timesteps of a
        denoise loop in one batched pass (include/diffute_hip.h dmx_unet_temb_table); forward_parts(..., temb=(table, step_index))
        then fetches row *step_index instead of recomputing four small layers per step.  Rows are bit-identical to the per-step path."""
        self._ensure_packed()
        lib = self._lib
        T = int(timesteps_dev.numel())
        table = torch.empty(int(lib.dmx_unet_temb_table_floats(self._h, T)), dtype=torch.float32, device=timesteps_dev.device)
        ws = torch.empty(int(lib.dmx_unet_temb_table_workspace_bytes(self._h, T)), dtype=torch.uint8, device=timesteps_dev.device)
        _cabi.check(lib.dmx_unet_temb_table(self._h, _cabi.ptr(timesteps_dev), T, _cabi.ptr(table), _cabi.ptr(ws), ws.numel(),
                                            _cabi.current_stream()), "unet_temb_table")
        return table.view(T, -1)

    def forward_parts(self, parts, timesteps_dev, out=None, graph=False, slot=0, temb=None):
        """Hot-loop entry: `parts` = list of (NCHW fp32 cuda tensor) whose channels sum to in_channels
        (fuses the torch.cat of app.ipynb:811); timesteps_dev = int64 cuda tensor [1] or [B];
        context must have been set with set_context() on the same slot.  graph=True replays a captured hipGraph
        when the same buffers are passed again (needs a non-default current stream).  temb = (temb_table(...), int32 cuda
        tensor [1] holding the step's row): the time-embedding projections come from the table (scalar timestep only)."""
        lib = self._lib
        sl = self._slot(slot)
        x0 = parts[0]
        B, _, H, W = x0.shape
        if sl["ctx_key"] is None or sl["ctx_shape"][0] != B:
            raise RuntimeError("UNet2DConditionModel: set_context() must be called with a batch-matching context first")
        ps = [(p, p.shape[1]) for p in parts] + [(None, 0)] * (3 - len(parts))
        if out is None:
            out = torch.empty(B, self.config.out_channels, H, W, dtype=torch.float32, device=x0.device)
        key = (B, H, W, sl["ctx_shape"][1], int(lib.dmx_plan_epoch()))      # (every plan switch changes the walk, hence the workspace it needs)
        if sl["ws_need"] is None or sl["ws_need"][0] != key:
            sl["ws_need"] = (key, lib.dmx_unet_workspace_bytes(self._h, B, H, W, sl["ctx_shape"][1]))
        ws = self._slot_workspace(sl, sl["ws_need"][1])
        fwd = lib.dmx_unet_forward_graph if graph else lib.dmx_unet_forward
        if temb is not None:
            _cabi.check(lib.dmx_unet_use_temb_table(self._h, _cabi.ptr(temb[0]), _cabi.ptr(temb[1])), "unet_use_temb_table")
        try:
            _cabi.check(fwd(self._h, _cabi.ptr(ps[0][0]), ps[0][1], _cabi.ptr(ps[1][0]), ps[1][1],
                            _cabi.ptr(ps[2][0]), ps[2][1], _cabi.ptr(timesteps_dev), timesteps_dev.numel(),
                            _cabi.ptr(sl["ctx_cache"]), sl["ctx_shape"][1], _cabi.ptr(out), B, H, W,
                            _cabi.ptr(ws), ws.numel(), _cabi.current_stream()), "unet_forward")
        finally:
            if temb is not None:
                lib.dmx_unet_use_temb_table(self._h, None, None)
        return out

    # ---- validation / debugging (tests only): per-block taps of the product path, and the fp32 instantiation of the graph
    TAP_NAMES = ("conv_in", "down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3")

    def _tap_buffers(self, B, H, W):
        boc = self.config.block_out_channels
        n = B * H * W * boc[0] * 4 + 16 * B * H * W * max(boc)           # generous: every tap is <= B*H*W*max(C) floats
        return (torch.empty(n, dtype=torch.float32, device=self.device), (ctypes.c_int * 64)(), ctypes.c_int(0))

    @staticmethod
    def _split_taps(buf, shapes, n):
        out, off = OrderedDict(), 0
        for i in range(n.value):
            b, c, h, w = (int(shapes[4 * i + k]) for k in range(4))
            out[UNet2DConditionModel.TAP_NAMES[i]] = buf[off:off + b * c * h * w].reshape(b, c, h, w).clone()
            off += b * c * h * w
        return out

    @torch.no_grad()
    def forward_taps(self, sample, timestep, encoder_hidden_states):
        """the product (bf16) forward plus the block outputs conv_in, down0..3, mid, up0..3 as NCHW fp32 tensors"""
        lib = self._lib
        self._ensure_packed()
        self.set_context(encoder_hidden_states)
        sl = self._slot(0)
        x = sample.to(torch.float32).contiguous()
        B, _, H, W = x.shape
        t = torch.as_tensor(timestep).reshape(-1).to(device=x.device, dtype=torch.int64)
        out = torch.empty(B, self.config.out_channels, H, W, dtype=torch.float32, device=x.device)
        ws = self._slot_workspace(sl, lib.dmx_unet_workspace_bytes(self._h, B, H, W, sl["ctx_shape"][1]))
        buf, shapes, n = self._tap_buffers(B, H, W)
        _cabi.check(lib.dmx_unet_forward_taps(self._h, _cabi.ptr(x), x.shape[1], None, 0, None, 0, _cabi.ptr(t), t.numel(),
                                              _cabi.ptr(sl["ctx_cache"]), sl["ctx_shape"][1], _cabi.ptr(out), B, H, W, _cabi.ptr(ws), ws.numel(),
                                              _cabi.ptr(buf), buf.numel(), shapes, ctypes.byref(n), _cabi.current_stream()), "unet_forward_taps")
        return out, self._split_taps(buf, shapes, n)

    @torch.no_grad()
    def forward_fp32(self, parts, timestep, encoder_hidden_states, taps=False):
        """VALIDATION ONLY: the same graph on fp32 activations, fp32 master weights and plain fp32 kernels (ref_f32.hip) -
        north_star's "within 1e-3 rel fp32" check against the fp32 reference path.  parts: one NCHW tensor or the list
        [latents, mask, masked_latents]; returns eps (and the block taps when taps=True).  Slow; never on the product path."""
        lib = self._lib
        self._ensure_packed()
        if torch.is_tensor(parts):
            parts = [parts]
        parts = [p.to(torch.float32).contiguous() for p in parts]
        B, _, H, W = parts[0].shape
        dev = parts[0].device
        m = getattr(self, "_masters32", None)
        if m is None or m[0] != self._packed_sig:
            arena = torch.zeros(lib.dmx_unet_grad_bytes(self._h) // 4, dtype=torch.float32, device=dev)
            st = _cabi.current_stream()
            for k, p in zip(self._keys, self._param_list()):
                src = p.detach().to(torch.float32).contiguous()
                _cabi.check(lib.dmx_unet_master_import(self._h, _cabi.ptr(arena), k.encode(), _cabi.ptr(src), st), f"master_import({k})")
            m = self._masters32 = (self._packed_sig, arena)
        ctx = encoder_hidden_states.to(torch.float32).contiguous()
        S = ctx.shape[1]
        t = torch.as_tensor(timestep).reshape(-1).to(device=dev, dtype=torch.int64)
        out = torch.empty(B, self.config.out_channels, H, W, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.dmx_unet_workspace_bytes_f32(self._h, B, H, W, S), dtype=torch.uint8, device=dev)
        ps = [(p, p.shape[1]) for p in parts] + [(None, 0)] * (3 - len(parts))
        buf, shapes, n = self._tap_buffers(B, H, W) if taps else (None, None, None)
        _cabi.check(lib.dmx_unet_forward_f32(self._h, _cabi.ptr(m[1]), _cabi.ptr(ps[0][0]), ps[0][1], _cabi.ptr(ps[1][0]), ps[1][1],
                                             _cabi.ptr(ps[2][0]), ps[2][1], _cabi.ptr(t), t.numel(), _cabi.ptr(ctx), S, _cabi.ptr(out), B, H, W,
                                             _cabi.ptr(ws), ws.numel(), _cabi.ptr(buf) if taps else None, buf.numel() if taps else 0,
                                             shapes if taps else None, ctypes.byref(n) if taps else None, _cabi.current_stream()), "unet_forward_f32")
        return (out, self._split_taps(buf, shapes, n)) if taps else out

    # ---- training (train_diffute_v1.py:913-925): forward that keeps activations + hand-written HIP backward
    def _train_buffers(self):
        """transposed-weights arena (data-gradient operands, refreshed when the weights change) and the fp32 gradient arena"""
        lib = self._lib
        tb = getattr(self, "_tb", None)
        if tb is None or tb["wt"].device != self.device:
            tb = self._tb = dict(wt=torch.empty(lib.dmx_unet_train_wt_bytes(self._h), dtype=torch.uint8, device=self.device),
                                 grads=torch.zeros(lib.dmx_unet_grad_bytes(self._h) // 4, dtype=torch.float32, device=self.device),      # (zeros: slots no backward writes - derived weights - are summed / exchanged with the rest)
                                 wt_sig=None, ws=None, events=None, plan=None)
        sig = (self._packed_sig, getattr(self, "_arena_version", 0))
        if tb["wt_sig"] != sig:
            _cabi.check(lib.dmx_unet_train_prepare(self._h, _cabi.ptr(tb["wt"]), tb["wt"].numel(), _cabi.current_stream()), "unet_train_prepare")
            tb["wt_sig"] = sig
        return tb

    def set_gradient_sync(self, dist=None, group=None, mode="rs_ag", accumulate_steps=1):
        """Average gradients over the ranks of `dist` (torch.distributed; RCCL on GPUs) INSIDE the backward: each of the
        11 gradient buckets is exchanged on a side stream as soon as the backward has finished it - as an
        in-place reduce-scatter + all-gather of the arena slices (mode "rs_ag") or one all-reduce per slice ("all_reduce").
        dist=None switches the exchange off (single GPU, or a wrapping torch DDP does it).

        accumulate_steps=n (`--gradient_accumulation_steps`, `accelerator.accumulate(unet)`, train_diffute_v1.py:873,926): only every n-th
        backward exchanges - the gradient ACCUMULATED over the window (diffute_amd.dist.GradientAccumulator); the others keep their
        gradient local, like DDP under `no_sync()`.  `with unet.no_sync():` does the same for the backwards inside the block.  With a torch
        optimizer the non-boundary backwards leave `.grad` untouched (the window's sum lives in the packed arena) and the boundary backward
        delivers the exchanged sum of the whole window; with FusedAdamW nothing changes for the caller."""
        if mode not in ("rs_ag", "all_reduce"):
            raise ValueError(f"set_gradient_sync: unknown mode {mode!r}")
        from .dist import GradientAccumulator
        self._sync = None if dist is None else dict(dist=dist, group=group, world=dist.get_world_size(group), stream=None, mode=mode,
                                                    exposed=None, acc=GradientAccumulator(accumulate_steps))
        # the exchange runs collective kernels on a side stream while the library's launches run: those hold CUs, so plans whose blocks need
        # co-resident peers (the in-kernel K split of dmx_conv3x3_gn, e.g. in the VAE encodes of the training step) are off for this process
        shared = self._sync is not None and self._sync["world"] > 1
        if shared and getattr(self, "_excl_before_sync", None) is None:
            self._excl_before_sync = (_cabi.set_exclusive_device(False),)
        elif not shared and getattr(self, "_excl_before_sync", None) is not None:
            _cabi.set_exclusive_device(self._excl_before_sync[0])      # what the caller had before the exchange was switched on
            self._excl_before_sync = None

    def no_sync(self):
        """`with unet.no_sync():` - the backwards inside the block keep their gradient local (torch DDP's contract, what
        `accelerator.accumulate` uses on non-boundary micro-steps, train_diffute_v1.py:873); the first backward after the block exchanges
        the accumulated gradient.  A no-op without set_gradient_sync."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            sync = getattr(self, "_sync", None)
            if sync is not None:
                sync["acc"].skip_ctx += 1
            try:
                yield
            finally:
                if sync is not None:
                    sync["acc"].skip_ctx -= 1
        return ctx()

    def exposed_exchange_ms(self):
        """how long the last backward's main stream sat waiting for the gradient exchange after its own kernels were done
        (the NON-overlapped part of D1); None before the first synchronised backward"""
        sync = getattr(self, "_sync", None)
        if not sync or not sync.get("exposed"):
            return None
        a, b = sync["exposed"]
        b.synchronize()
        return a.elapsed_time(b)

    def _sync_plan(self, tb):
        if tb["plan"] is None:
            from .dist import plan_buckets
            lib = self._lib
            b, e = ctypes.c_size_t(), ctypes.c_size_t()
            prs = []
            for k in self._keys:
                _cabi.check(lib.dmx_unet_grad_range(self._h, k.encode(), ctypes.byref(b), ctypes.byref(e)), "grad_range")
                prs.append((b.value, e.value))
            nb = lib.dmx_unet_train_bucket_count(self._h)
            spans = []
            for i in range(nb):
                _cabi.check(lib.dmx_unet_train_bucket_range(self._h, i, ctypes.byref(b), ctypes.byref(e)), "bucket_range")
                spans.append([(b.value, e.value)])
            _cabi.check(lib.dmx_unet_train_tail_range(self._h, ctypes.byref(b), ctypes.byref(e)), "tail_range")
            spans[-1].append((b.value, e.value))
            tb["plan"] = plan_buckets(prs, spans)
        return tb["plan"]

    def _train_forward(self, sample, timestep, ctx):
        lib = self._lib
        self._ensure_packed()
        tb = self._train_buffers()
        B, _, H, W = sample.shape
        S = ctx.shape[1]
        need = lib.dmx_unet_train_workspace_bytes(self._h, B, H, W, S)
        if tb["ws"] is None or tb["ws"].numel() < need:
            tb["ws"] = None
            tb["ws"] = torch.empty(int(need), dtype=torch.uint8, device=sample.device)
        pred = torch.empty(B, self.config.out_channels, H, W, dtype=torch.float32, device=sample.device)
        _cabi.check(lib.dmx_unet_train_forward(self._h, _cabi.ptr(tb["wt"]), _cabi.ptr(sample), sample.shape[1], None, 0, None, 0,
                                               _cabi.ptr(timestep), timestep.numel(), _cabi.ptr(ctx), int(ctx.dtype == self.compute_dtype), S,
                                               _cabi.ptr(pred), B, H, W, _cabi.ptr(tb["ws"]), tb["ws"].numel(), _cabi.current_stream()),
                    "unet_train_forward")
        tb["fwd_stream"] = torch.cuda.current_stream(sample.device)
        return pred

    def _train_backward(self, dpred):
        """-> list of parameter gradients (torch layouts, fp32) in self._keys order"""
        lib = self._lib
        tb = self._tb
        sync = getattr(self, "_sync", None)
        if sync is not None:
            dpred = dpred / sync["world"]                      # SUM over ranks below -> mean gradient
        dpred = dpred.to(torch.float32).contiguous()
        ev_arr, n_ev = None, 0
        acc = sync["acc"] if sync is not None else None
        fused = getattr(self, "_fused", None)
        # gradient-accumulation window with the exchange on: local sum in acc.acc, one exchange at the boundary (dist.GradientAccumulator)
        windowed = acc is not None and (not acc.boundary() or acc.acc_n > 0)
        if windowed and acc.acc_n == 0 and fused is not None and fused._pending > 0:
            raise RuntimeError("set_gradient_sync: a no_sync() / accumulate_steps window cannot start after a backward of the same optimizer step was already exchanged")
        exchange = sync is not None and acc.boundary()
        if exchange:
            n_ev = lib.dmx_unet_train_bucket_count(self._h)
            if tb["events"] is None:
                tb["events"] = [torch.cuda.Event() for _ in range(n_ev)]
                for ev in tb["events"]:
                    ev.record()                                # materialise the hipEvent_t handles
            ev_arr = (ctypes.c_void_p * n_ev)(*[ev.cuda_event for ev in tb["events"]])
        with torch.cuda.stream(tb["fwd_stream"]):
            if fused is not None and not windowed:
                fused.before_backward(tb["grads"])             # gradient accumulation: stash what earlier backwards left
            _cabi.check(lib.dmx_unet_train_backward(self._h, _cabi.ptr(tb["grads"]), _cabi.ptr(dpred), ev_arr, n_ev, _cabi.current_stream()),
                        "unet_train_backward")
            if sync is not None and not exchange:
                acc.stash(tb["grads"])                         # a non-boundary micro-step: its gradient stays local (no_sync)
        if exchange:
            from .dist import reduce_buckets
            if sync["stream"] is None:
                # (a HIGH-PRIORITY stream: HIP multiplexes ordinary streams onto a few hardware queues, and an exchange stream that shares the backward's
                # queue would run behind it instead of beside it; priority streams get their own queue - and the collectives are the latency-critical part)
                sync["stream"] = torch.cuda.Stream(device=dpred.device, priority=-1)
            side = sync["stream"]
            plan = self._sync_plan(tb)

            def bucket_ready(i):                               # on the exchange stream: the bucket is complete, then + the window's local sum
                side.wait_event(tb["events"][i])
                acc.pre_add(tb["grads"], plan[i])
            with torch.cuda.stream(side):                      # (acc.acc was completed by earlier backwards on the forward's stream: every bucket event orders behind them)
                reduce_buckets(tb["grads"], plan, sync["dist"], group=sync["group"], wait_bucket=bucket_ready, mode=sync["mode"])
            acc.exchanged()
            if sync["exposed"] is None:
                sync["exposed"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            sync["exposed"][0].record(tb["fwd_stream"])                 # the backward's own kernels end here ...
            tb["fwd_stream"].wait_stream(side)
            sync["exposed"][1].record(tb["fwd_stream"])                 # ... and here the exchange has caught up
        if fused is not None:                              # the fused optimizer reads the gradient arena directly
            with torch.cuda.stream(tb["fwd_stream"]):
                if windowed:
                    fused._pending += 1                        # (the window's sum is in acc / was added before the exchange: nothing to add back)
                else:
                    fused.after_backward(tb["grads"])
            torch.cuda.current_stream(dpred.device).wait_stream(tb["fwd_stream"])
            return [None] * len(self._keys)
        if sync is not None and not exchange:              # torch optimizer, non-boundary micro-step: `.grad` stays as it is; the boundary delivers the window's sum
            torch.cuda.current_stream(dpred.device).wait_stream(tb["fwd_stream"])
            return [None] * len(self._keys)
        out = []
        with torch.cuda.stream(tb["fwd_stream"]):
            st = _cabi.current_stream()
            for k, p in zip(self._keys, self._param_list()):
                g = torch.empty(p.shape, dtype=torch.float32, device=dpred.device)
                _cabi.check(lib.dmx_unet_grad_export(self._h, _cabi.ptr(tb["grads"]), k.encode(), _cabi.ptr(g), st), "grad_export")
                out.append(g if p.dtype == torch.float32 else g.to(p.dtype))
        torch.cuda.current_stream(dpred.device).wait_stream(tb["fwd_stream"])
        return out

    def _param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in self._keys]

    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True, **unused):
        """unet(sample[B,9,h,w], timestep (int / 0-d / [B] tensor), encoder_hidden_states[B,S,1024])."""
        _cabi.require_cuda(sample, encoder_hidden_states)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            B = sample.shape[0]
            if not torch.is_tensor(timestep):
                timestep = torch.tensor([int(timestep)], dtype=torch.int64)
            t = timestep.reshape(-1).to(device=sample.device, dtype=torch.int64)
            if t.numel() not in (1, B):
                raise ValueError(f"timestep must have 1 or {B} elements, got {t.numel()}")
            # (the fp16 build trains too - `--mixed_precision fp16`, train_diffute_v1.py:267,583,790: activation gradients are stored in
            # fp16 there, so run the backward under a loss scale: diffute_amd.GradScaler / torch.amp.GradScaler, training.train_step(scaler=))
            ctx = encoder_hidden_states if encoder_hidden_states.dtype in (torch.float32, self.compute_dtype) else encoder_hidden_states.float()
            out = _UNetTrainFn.apply(self, sample.detach().to(torch.float32).contiguous(), t, ctx.detach().contiguous(), *self._param_list())
            if self._dtype != torch.float32 and sample.dtype != torch.float32:
                out = out.to(sample.dtype)
            return UNet2DConditionOutput(sample=out) if return_dict else (out,)
        self._ensure_packed()
        ck = self._slot(0)["ctx_key"]
        if ck is None or ck[0] is not encoder_hidden_states or ck[1] != encoder_hidden_states._version:
            self.set_context(encoder_hidden_states)
        B = sample.shape[0]
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([int(timestep)], dtype=torch.int64)
        t = timestep.reshape(-1).to(device=sample.device, dtype=torch.int64)
        if t.numel() not in (1, B):
            raise ValueError(f"timestep must have 1 or {B} elements, got {t.numel()}")
        x = sample.to(torch.float32).contiguous()
        out = self.forward_parts([x], t)
        if self._dtype != torch.float32 and sample.dtype != torch.float32:
            out = out.to(sample.dtype)
        return UNet2DConditionOutput(sample=out) if return_dict else (out,)


class _UNetTrainFn(torch.autograd.Function):
    """Glue between torch autograd (loss.backward(), DDP hooks on the parameters) and the HIP training graph: the
    parameters are inputs so that their .grad is filled by autograd from the gradients the HIP backward exports."""

    @staticmethod
    def forward(ctx, model, sample, t, ehs, *params):
        ctx.model = model
        return model._train_forward(sample, t, ehs)

    @staticmethod
    def backward(ctx, dpred):
        grads = ctx.model._train_backward(dpred)
        return (None, None, None, None) + tuple(grads)


class _VAETrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, *params):
        ctx.model = model
        return model._train_forward(x)

    @staticmethod
    def backward(ctx, drecon):
        return (None, None) + tuple(ctx.model._train_backward(drecon))


def mse_loss(pred, target):
    """F.mse_loss(pred.float(), target.float(), reduction="mean") (train_diffute_v1.py:918) as a HIP kernel with its
    gradient: two-stage fixed-order fp32 reduction."""
    return _MSEFn.apply(pred, target)


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        lib = _cabi.lib()
        p = pred.detach().to(torch.float32).contiguous(); t = target.detach().to(torch.float32).contiguous()
        _cabi.require_cuda(p, t)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        dp = torch.empty_like(p)
        ws = torch.empty(lib.dmx_mse_loss_workspace_bytes(), dtype=torch.uint8, device=p.device)
        _cabi.check(lib.dmx_mse_loss(_cabi.ptr(p), _cabi.ptr(t), p.numel(), _cabi.ptr(loss), _cabi.ptr(dp), 1.0,
                                     _cabi.ptr(ws), ws.numel(), _cabi.current_stream()), "mse_loss")
        ctx.save_for_backward(dp)
        ctx.in_dtype = pred.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        (dp,) = ctx.saved_tensors
        return (dp * g).to(ctx.in_dtype), None


class AutoencoderKL(_HipModel):
    """SD VAE (reference: train_diffute_v1.py:632, app.ipynb:550, train_vae.py:516)."""
    _kind = "vae"

    def __init__(self, seed=4321, device="cpu", **config):
        super().__init__()
        cfg = dict(SD_VAE_CONFIG); cfg.update(config)
        _check_supported("AutoencoderKL", cfg, _VAE_ONLY_SUPPORTED)
        for k, want in (("down_block_types", "DownEncoderBlock2D"), ("up_block_types", "UpDecoderBlock2D")):
            if k in cfg:
                cfg[k] = tuple(cfg[k])
                if any(b != want for b in cfg[k]) or len(cfg[k]) != len(cfg["block_out_channels"]):
                    raise NotImplementedError(f"AutoencoderKL: {k}={cfg[k]!r} is not implemented (only {want!r} per level)")
        cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
        if cfg["block_out_channels"][-1] not in (128, 256, 512):
            # the fused single-head attention of the mid block (attention_wide.hip) is built for head widths 128 / 256 / 512
            raise NotImplementedError(f"AutoencoderKL: mid-block width block_out_channels[-1]={cfg['block_out_channels'][-1]} "
                                      "is not implemented (only 128, 256 or 512)")
        self.config = _Config(**cfg)
        c = _cabi.VAEConfig()
        c.in_channels = cfg["in_channels"]; c.out_channels = cfg["out_channels"]; c.latent_channels = cfg["latent_channels"]
        c.layers_per_block = cfg["layers_per_block"]; c.norm_num_groups = cfg["norm_num_groups"]
        for i in range(4):
            c.block_out_channels[i] = cfg["block_out_channels"][i]
        self._cstruct = c
        h = self._create_handle("bf16")
        self._setup(h, seed, device)

    def _finalize(self, st):
        _cabi.check(self._lib.dmx_vae_finalize(self._h, st), "vae_finalize")

    @staticmethod
    def _convert_legacy_keys(sd):
        """Old SD VAE checkpoints name the mid-block attention query/key/value/proj_attn."""
        ren = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
        out = OrderedDict()
        for k, v in sd.items():
            parts = k.split(".")
            if "attentions" in parts and len(parts) >= 2 and parts[-2] in ren:
                parts[-2] = ren[parts[-2]]
                k = ".".join(parts)
                if v.ndim == 4:
                    v = v[:, :, 0, 0]
            out[k] = v
        return out

    def _no_grad_check(self):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("diffute_amd: the HIP VAE is forward-only in this build; use torch.no_grad() / requires_grad_(False)")

    def encode(self, x, return_dict=True):
        self._no_grad_check()
        _cabi.require_cuda(x)
        self._ensure_packed()
        lib = self._lib
        x = x.to(torch.float32).contiguous()
        B, C, H, W = x.shape
        f = 2 ** (len(self.config.block_out_channels) - 1)
        moments = torch.empty(B, 2 * self.config.latent_channels, H // f, W // f, dtype=torch.float32, device=x.device)
        ws = self._workspace(lib.dmx_vae_workspace_bytes(self._h, B, H, W, 0))
        _cabi.check(lib.dmx_vae_encode(self._h, _cabi.ptr(x), _cabi.ptr(moments), B, H, W, _cabi.ptr(ws), ws.numel(),
                                       _cabi.current_stream()), "vae_encode")
        dist = DiagonalGaussianDistribution(moments)
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)

    def decode(self, z, return_dict=True):
        self._no_grad_check()
        _cabi.require_cuda(z)
        self._ensure_packed()
        lib = self._lib
        z = z.to(torch.float32).contiguous()
        B, C, h, w = z.shape
        f = 2 ** (len(self.config.block_out_channels) - 1)
        img = torch.empty(B, self.config.out_channels, h * f, w * f, dtype=torch.float32, device=z.device)
        ws = self._workspace(lib.dmx_vae_workspace_bytes(self._h, B, h, w, 1))
        _cabi.check(lib.dmx_vae_decode(self._h, _cabi.ptr(z), _cabi.ptr(img), B, h, w, _cabi.ptr(ws), ws.numel(),
                                       _cabi.current_stream()), "vae_decode")
        return DecoderOutput(sample=img) if return_dict else (img,)

    # ---- fp32 VALIDATION path (tests only): the same graphs on fp32 activations, fp32 master weights, plain fp32 kernels
    def _masters_fp32(self):
        lib = self._lib
        self._ensure_packed()
        m = getattr(self, "_masters32", None)
        if m is None or m[0] != self._packed_sig:
            arena = torch.zeros(lib.dmx_vae_grad_bytes(self._h) // 4, dtype=torch.float32, device=self.device)
            st = _cabi.current_stream()
            for k, p in zip(self._keys, self._param_list()):
                src = p.detach().to(torch.float32).contiguous()
                _cabi.check(lib.dmx_vae_master_import(self._h, _cabi.ptr(arena), k.encode(), _cabi.ptr(src), st), f"vae_master_import({k})")
            m = self._masters32 = (self._packed_sig, arena)
        return m[1]

    @torch.no_grad()
    def encode_fp32(self, x):
        """VALIDATION ONLY: `encode(x).latent_dist.parameters` (the moments) through the fp32 instantiation of the graph -
        north_star's "within 1e-3 rel fp32" check at model level.  Slow; images up to ~384 px."""
        lib = self._lib
        m = self._masters_fp32()
        x = x.to(torch.float32).contiguous()
        B, C, H, W = x.shape
        f = 2 ** (len(self.config.block_out_channels) - 1)
        moments = torch.empty(B, 2 * self.config.latent_channels, H // f, W // f, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.dmx_vae_workspace_bytes_f32(self._h, B, H, W, 0), dtype=torch.uint8, device=x.device)
        _cabi.check(lib.dmx_vae_encode_f32(self._h, _cabi.ptr(m), _cabi.ptr(x), _cabi.ptr(moments), B, H, W, _cabi.ptr(ws), ws.numel(),
                                           _cabi.current_stream()), "vae_encode_f32")
        return moments

    @torch.no_grad()
    def decode_fp32(self, z):
        """VALIDATION ONLY: `decode(z).sample` through the fp32 instantiation of the graph."""
        lib = self._lib
        m = self._masters_fp32()
        z = z.to(torch.float32).contiguous()
        B, C, h, w = z.shape
        f = 2 ** (len(self.config.block_out_channels) - 1)
        img = torch.empty(B, self.config.out_channels, h * f, w * f, dtype=torch.float32, device=z.device)
        ws = torch.empty(lib.dmx_vae_workspace_bytes_f32(self._h, B, h, w, 1), dtype=torch.uint8, device=z.device)
        _cabi.check(lib.dmx_vae_decode_f32(self._h, _cabi.ptr(m), _cabi.ptr(z), _cabi.ptr(img), B, h, w, _cabi.ptr(ws), ws.numel(),
                                           _cabi.current_stream()), "vae_decode_f32")
        return img

    # ---- training (train_vae.py:716-736): recon = decode(encode(x).mode()) with a HIP backward
    def _train_buffers(self):
        lib = self._lib
        tb = getattr(self, "_tb", None)
        if tb is None or tb["wt"].device != self.device:
            tb = self._tb = dict(wt=torch.empty(lib.dmx_vae_train_wt_bytes(self._h), dtype=torch.uint8, device=self.device),
                                 grads=torch.zeros(lib.dmx_vae_grad_bytes(self._h) // 4, dtype=torch.float32, device=self.device),
                                 wt_sig=None, ws=None)
        if tb["wt_sig"] != self._packed_sig:
            _cabi.check(lib.dmx_vae_train_prepare(self._h, _cabi.ptr(tb["wt"]), tb["wt"].numel(), _cabi.current_stream()), "vae_train_prepare")
            tb["wt_sig"] = self._packed_sig
        return tb

    def _param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in self._keys]

    def _train_forward(self, x):
        lib = self._lib
        self._ensure_packed()
        tb = self._train_buffers()
        B, _, H, W = x.shape
        need = lib.dmx_vae_train_workspace_bytes(self._h, B, H, W)
        if tb["ws"] is None or tb["ws"].numel() < need:
            tb["ws"] = None
            tb["ws"] = torch.empty(int(need), dtype=torch.uint8, device=x.device)
        recon = torch.empty(B, self.config.out_channels, H, W, dtype=torch.float32, device=x.device)
        _cabi.check(lib.dmx_vae_train_forward(self._h, _cabi.ptr(tb["wt"]), _cabi.ptr(x), _cabi.ptr(recon), B, H, W,
                                              _cabi.ptr(tb["ws"]), tb["ws"].numel(), _cabi.current_stream()), "vae_train_forward")
        tb["fwd_stream"] = torch.cuda.current_stream(x.device)
        return recon

    def _train_backward(self, drecon):
        lib = self._lib
        tb = self._tb
        drecon = drecon.to(torch.float32).contiguous()
        out = []
        with torch.cuda.stream(tb["fwd_stream"]):
            st = _cabi.current_stream()
            _cabi.check(lib.dmx_vae_train_backward(self._h, _cabi.ptr(tb["grads"]), _cabi.ptr(drecon), st), "vae_train_backward")
            for k, p in zip(self._keys, self._param_list()):
                g = torch.empty(p.shape, dtype=torch.float32, device=drecon.device)
                _cabi.check(lib.dmx_vae_grad_export(self._h, _cabi.ptr(tb["grads"]), k.encode(), _cabi.ptr(g), st), "vae_grad_export")
                out.append(g if p.dtype == torch.float32 else g.to(p.dtype))
        torch.cuda.current_stream(drecon.device).wait_stream(tb["fwd_stream"])
        return out

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None):
        """`vae(x)["sample"]` (train_vae.py:721-722): decode(encode(x).latent_dist.mode())."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if sample_posterior:
                raise NotImplementedError("diffute_amd: the differentiable VAE path is vae(x) with the posterior MODE (train_vae.py:721)")
            _cabi.require_cuda(sample)
            if sample.shape[2] % 64 or sample.shape[3] % 64:
                raise ValueError("AutoencoderKL training: image sides must be multiples of 64")
            # (fp16 build: run the backward under a loss scale - torch.amp.GradScaler / diffute_amd.GradScaler; train_vae.py's --mixed_precision)
            dec = _VAETrainFn.apply(self, sample.detach().to(torch.float32).contiguous(), *self._param_list())
            return {"sample": dec} if return_dict else (dec,)
        post = self.encode(sample).latent_dist
        z = post.sample(generator=generator) if sample_posterior else post.mode()
        dec = self.decode(z).sample
        return {"sample": dec} if return_dict else (dec,)


TROCR_LARGE_VIT_CONFIG = dict(image_size=384, patch_size=16, num_channels=3, hidden_size=1024, num_hidden_layers=24,
                              num_attention_heads=16, intermediate_size=4096, qkv_bias=False, layer_norm_eps=1e-12, hidden_act="gelu")


class BaseModelOutput(SimpleNamespace):
    """`.last_hidden_state` (app.ipynb:776, train_diffute_v1.py:871)."""


class TrOCREncoder(_HipModel):
    """The glyph encoder: ViT encoder of TrOCR (reference: `trocr_model = VisionEncoderDecoderModel.from_pretrained(
    'microsoft/trocr-large-printed').encoder`, train_diffute_v1.py:630-631, app.ipynb:546-548;
    `trocr_model(pixel_values).last_hidden_state`, :868-871 / :773-776).  Forward-only (frozen in the reference, :638)."""
    _kind = "vit"

    def __init__(self, seed=777, device="cpu", **config):
        super().__init__()
        cfg = dict(TROCR_LARGE_VIT_CONFIG); cfg.update(config)
        if cfg.get("hidden_act", "gelu") != "gelu":
            raise ValueError("TrOCREncoder: only the exact GELU activation is implemented")
        self.config = _Config(**cfg)
        c = _cabi.ViTConfig()
        c.image_size = cfg["image_size"]; c.patch_size = cfg["patch_size"]; c.num_channels = cfg["num_channels"]
        c.hidden_size = cfg["hidden_size"]; c.num_layers = cfg["num_hidden_layers"]; c.num_heads = cfg["num_attention_heads"]
        c.intermediate_size = cfg["intermediate_size"]; c.qkv_bias = int(bool(cfg["qkv_bias"])); c.layer_norm_eps = float(cfg["layer_norm_eps"])
        self._cstruct = c
        h = self._create_handle("bf16")
        self._setup(h, seed, device)
        self.requires_grad_(False)

    def _finalize(self, st):
        _cabi.check(self._lib.dmx_vit_finalize(self._h, st), "vit_finalize")

    @staticmethod
    def _convert_legacy_keys(sd):
        """VisionEncoderDecoderModel checkpoints prefix the encoder with `encoder.`; the ViTModel pooler is unused."""
        if any(k.startswith("encoder.embeddings.") for k in sd):
            sd = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
        return {k: v for k, v in sd.items() if not k.startswith("pooler.")}

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, revision=None, **kw):
        """transformers directory layout: config.json (a ViT config, or a VisionEncoderDecoder config whose `encoder` entry
        is one) + model.safetensors / pytorch_model.bin."""
        d = pretrained_model_name_or_path if subfolder is None else os.path.join(pretrained_model_name_or_path, subfolder)
        with open(os.path.join(d, "config.json")) as f:
            cfg = json.load(f)
        cfg = cfg.get("encoder", cfg)
        keep = {k: cfg[k] for k in TROCR_LARGE_VIT_CONFIG if k in cfg}
        model = cls(**keep)
        st = os.path.join(d, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(d, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict(cls._convert_legacy_keys(sd))
        return model

    def forward(self, pixel_values, return_dict=True, **unused):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("diffute_amd: the glyph encoder is forward-only (frozen in the reference, train_diffute_v1.py:638)")
        _cabi.require_cuda(pixel_values)
        self._ensure_packed()
        lib = self._lib
        x = pixel_values.to(torch.float32).contiguous()
        B, C, H, W = x.shape
        if (C, H, W) != (self.config.num_channels, self.config.image_size, self.config.image_size):
            raise ValueError(f"pixel_values must be [B,{self.config.num_channels},{self.config.image_size},{self.config.image_size}], got {tuple(x.shape)}")
        n = (self.config.image_size // self.config.patch_size) ** 2 + 1
        out = torch.empty(B, n, self.config.hidden_size, dtype=torch.float32, device=x.device)
        ws = self._workspace(lib.dmx_vit_workspace_bytes(self._h, B))
        _cabi.check(lib.dmx_vit_forward(self._h, _cabi.ptr(x), _cabi.ptr(out), B, _cabi.ptr(ws), ws.numel(), _cabi.current_stream()), "vit_forward")
        if self._dtype != torch.float32 and pixel_values.dtype != torch.float32:
            out = out.to(pixel_values.dtype)
        return BaseModelOutput(last_hidden_state=out) if return_dict else (out,)

    @torch.no_grad()
    def forward_fp32(self, pixel_values):
        """VALIDATION ONLY: `last_hidden_state` through the fp32 instantiation of the graph (fp32 activations, fp32 master
        weights, plain fp32 kernels) - compared with transformers' ViTModel at north_star's 1e-3.  Slow."""
        lib = self._lib
        self._ensure_packed()
        x = pixel_values.to(torch.float32).contiguous()
        B = x.shape[0]
        m = getattr(self, "_masters32", None)
        if m is None or m[0] != self._packed_sig:
            arena = torch.zeros(lib.dmx_vit_master_bytes(self._h) // 4, dtype=torch.float32, device=x.device)
            st = _cabi.current_stream()
            sd = dict(self.named_parameters())
            for k in self._keys:
                src = sd[k].detach().to(torch.float32).contiguous()
                _cabi.check(lib.dmx_vit_master_import(self._h, _cabi.ptr(arena), k.encode(), _cabi.ptr(src), st), f"vit_master_import({k})")
            m = self._masters32 = (self._packed_sig, arena)
        n = (self.config.image_size // self.config.patch_size) ** 2 + 1
        out = torch.empty(B, n, self.config.hidden_size, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.dmx_vit_workspace_bytes_f32(self._h, B), dtype=torch.uint8, device=x.device)
        _cabi.check(lib.dmx_vit_forward_f32(self._h, _cabi.ptr(m[1]), _cabi.ptr(x), _cabi.ptr(out), B, _cabi.ptr(ws), ws.numel(),
                                            _cabi.current_stream()), "vit_forward_f32")
        return out

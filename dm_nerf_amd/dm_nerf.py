"""Field model and encoders with the reference's names, constructor arguments and
``state_dict`` layout (networks/dm_nerf.py:8-106), evaluated by the fused gfx950 kernel.

Unlike the reference module, importing this file has no global side effect (the reference
switches on autograd anomaly detection at import, networks/dm_nerf.py:5)."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from . import ops

POS_CH, DIR_CH = 63, 27


class Embedder:
    """``Embedder(**kwargs).embed(x)`` -> ``[x, sin(2^0 x), cos(2^0 x), ...]``
    (networks/dm_nerf.py:8-38).  Only the configuration the reference ever builds is
    supported on the GPU: 3 input dims, input included, log-sampled bands, sin+cos."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs.get("input_dims", 3)
        if d != 3 or not kwargs.get("include_input", True) or not kwargs.get("log_sampling", True):
            raise NotImplementedError("dm_nerf_amd encodes 3-vectors with include_input and log sampling only")
        self.num_freqs = int(kwargs["num_freqs"])
        if int(kwargs.get("max_freq_log2", self.num_freqs - 1)) != self.num_freqs - 1:
            raise NotImplementedError("bands must be 2^0 .. 2^(num_freqs-1)")
        self.out_dim = d + 2 * d * self.num_freqs

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        return ops.encode(inputs, self.num_freqs)

    __call__ = embed


def get_embedder(multires: int, i: int = 0):
    """networks/dm_nerf.py:41-55.  ``i == -1`` (identity) is not used by any reference config
    and is not supported by the fused field kernel."""
    if i == -1:
        raise NotImplementedError("identity embedding (i_embed=-1) is not supported by the gfx950 field kernel")
    embedder = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                        log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return embedder, embedder.out_dim


class DM_NeRF(nn.Module):
    """8x256 field with skip at layer 4, density head, view-dependent colour branch and detached
    instance-code branch.  Same parameter names and shapes as the reference module
    (networks/dm_nerf.py:59-78), so checkpoints load unchanged (train_dmsr.py:82-87).

    ``forward(embedded[..., 90])`` runs the fused kernel on caller-made encodings
    (networks/manipulator.py:125-131); the render path uses ``blob()`` and lets the kernel
    build the encodings itself."""

    def __init__(self, D=8, W=256, input_ch_pts=3, input_ch_views=3, skips=[4], ins_num=None):
        super().__init__()
        if (D, W, input_ch_pts, input_ch_views, list(skips)) != (8, 256, POS_CH, DIR_CH, [4]):
            raise NotImplementedError(
                "the gfx950 field kernel is specialised for D=8, W=256, 63+27 input channels, skips=[4] "
                "(every configuration the reference ships: config.py:23-25,56-59,153)")
        if ins_num is None or not (0 <= ins_num < ops.MAX_INS_CHANNELS):
            raise ValueError(f"ins_num must be in [0, {ops.MAX_INS_CHANNELS - 1}]")
        self.skips = list(skips)
        self.input_ch_pts, self.input_ch_views, self.ins_num = input_ch_pts, input_ch_views, ins_num
        self.mlps = nn.ModuleList(
            [nn.Linear(input_ch_pts, W)]
            + [nn.Linear(W + input_ch_pts, W) if i in skips else nn.Linear(W, W) for i in range(D - 1)])
        self.rgb_feature_linear = nn.Linear(W, W)
        self.ins_feature_linear = nn.Linear(W, W)
        self.rgb_feature_linears = nn.ModuleList([nn.Linear(W + input_ch_views, W // 2)])
        self.ins_feature_linears = nn.ModuleList([nn.Linear(W, W // 2)])
        self.density_linear = nn.Linear(W, 1)
        self.ins_linear = nn.Linear(W // 2, ins_num + 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self.precision = ops.DEFAULT_PRECISION      # arithmetic of forward(); the render path passes its own
        self._blobs = {}                 # precision -> (key, blob)
        self._indices = {}               # (flavour, device) -> gather table
        self._flat = None                # (key, parameters as one flat vector) of the last blob packed
        self._grad_flat = None           # persistent flat gradient buffer (parallel.GradBucket hands out slices)
        self._grad_lent = False          # the buffer is the output of a backward node of the running backward pass

    # ---- packed weights -------------------------------------------------------------------
    def _ordered_params(self) -> List[torch.Tensor]:
        return [p for _, p in self.named_parameters()]      # registration order = state_dict order

    def blob(self, precision: str = "fp32") -> torch.Tensor:
        """Device blob consumed by the field kernel of the given precision ("bwd" / "bwd_x6" / "bwd_x3": the
        transposed-weight blobs of the exact, the bf16x6 and the bf16x3 backward-data kernel); re-gathered only when a parameter changed
        (optimizer steps, ``load_state_dict`` and every other in-place op bump the tensors' version counters).
        Writes THROUGH ``p.data`` (``p.data.mul_(..)``, legacy init / EMA / clipping code) bypass the version counter by
        design of autograd: call ``invalidate_blobs()`` after such a write."""
        if precision not in ("bwd", "bwd_x3", "bwd_x6"):
            ops.precision_code(precision)
        params = self._ordered_params()
        dev = params[0].device
        key = (dev, tuple((p.data_ptr(), p._version) for p in params))
        cached = self._blobs.get(precision)
        if cached is not None and cached[0] == key:
            return cached[1]
        if precision in ("fp32", "bwd", "bwd_x3", "bwd_x6"):
            flavour = precision
        elif precision in ops.X6_PRECISIONS:       # inference on the 16x16x32 kernel where it exists, else the 32x32x16 one
            flavour = "x6w" if ops.uses_x6w(self.ins_num, precision) else "x6"
        else:
            flavour = "x3"
        index = self._indices.get((flavour, dev))
        if index is None:
            make = {"fp32": ops.pack_index, "bwd": ops.bwd_pack_index, "bwd_x3": ops.bwd_pack_index_x3,
                    "bwd_x6": ops.bwd_pack_index_x6, "x3": ops.pack_index_x3, "x6": ops.pack_index_x6,
                    "x6w": ops.pack_index_x6w}[flavour]
            index = self._indices[(flavour, dev)] = make(self.ins_num, dev)
        if self._flat is not None and self._flat[0] == key:      # the backward's blob follows the forward's: one gather
            flat = self._flat[1]
        else:
            with torch.no_grad():
                flat = torch.cat([p.detach().reshape(-1).float() for p in params])
            self._flat = (key, flat)
        if flavour == "x3":
            blob = ops.pack_blob_x3(flat, index, self.ins_num, precision)
        elif flavour in ("x6", "x6w"):             # same gather kernel: the plane is the chunk's position modulo 3
            blob = ops.pack_blob_x6(flat, index, self.ins_num)
        elif flavour == "bwd_x3":
            blob = ops.pack_bwd_blob_x3(flat, index, self.ins_num)
        elif flavour == "bwd_x6":
            blob = ops.pack_bwd_blob_x6(flat, index, self.ins_num)
        else:
            blob = ops.pack_blob(flat, index)
        self._blobs[precision] = (key, blob)
        return blob

    def adopt_grad_buffer(self, flat: torch.Tensor) -> None:
        """Use ``flat`` (fp32, one float per parameter, state_dict order) as the destination of the weight-gradient
        kernel: after ``backward()`` every ``p.grad`` is a view of it (``parallel.GradBucket``)."""
        n = sum(p.numel() for p in self.parameters())
        if flat.dtype != torch.float32 or flat.numel() != n or not flat.is_contiguous():
            raise ValueError(f"gradient buffer must be {n} contiguous float32")
        self._grad_flat = flat
        self._grad_lent = False

    def grad_buffer(self):
        """The adopted flat gradient buffer if the kernel may write it now, else None: no parameter holds a gradient
        yet (one that lives in the buffer or would have to be accumulated into), and no other backward node of the
        running backward pass has been handed the buffer.  The second condition matters when this module is applied
        twice in one graph (two ray batches summed into one loss): the engine sums the two nodes' gradients only after
        both have run, ``p.grad`` is still None when the second one asks, and a second in-place write would replace the
        first node's result inside the engine's input buffer (g2 + g2 instead of g1 + g2).  The lending ends with the
        backward pass (``lend_grad_buffer``)."""
        flat = self._grad_flat
        if flat is None or self._grad_lent or flat.device != self._ordered_params()[0].device:
            return None
        if any(p.grad is not None for p in self.parameters()):
            return None
        return flat

    def lend_grad_buffer(self):
        """``grad_buffer()`` for a backward node that is going to return views of it: marks the buffer as handed out
        until the engine has finished the running backward pass (by then the gradients have been accumulated into
        ``p.grad``, which is what ``grad_buffer`` looks at from there on)."""
        flat = self.grad_buffer()
        if flat is not None:
            self._grad_lent = True
            torch.autograd.Variable._execution_engine.queue_callback(self._return_grad_buffer)
        return flat

    def _return_grad_buffer(self):
        self._grad_lent = False

    def invalidate_blobs(self) -> None:
        """Drop the packed-weight cache (needed only after parameter writes through ``.data``, see ``blob``)."""
        self._blobs.clear()
        self._flat = None

    def wgrad_index(self) -> torch.Tensor:
        """Scatter table of the weight-gradient kernel (ops.field_backward) on the parameters' device."""
        dev = self._ordered_params()[0].device
        index = self._indices.get(("wgrad", dev))
        if index is None:
            index = self._indices[("wgrad", dev)] = ops.wgrad_index(self.ins_num, dev)
        return index

    def _refuse_grad(self):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(
                "dm_nerf_amd: DM_NeRF.forward(embedded) is an inference entry point (manipulator / mesh tools, "
                "networks/manipulator.py:125-131); call it under torch.no_grad().  Training goes through dm_nerf().")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._refuse_grad()
        return ops.field_embedded(x, self.blob(self.precision), self.ins_num, self.precision)
